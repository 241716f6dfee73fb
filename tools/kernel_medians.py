#!/usr/bin/env python3
"""Median / mean / count of every kernel's duration from a rocprofv3 kernel trace (`*_kernel_trace.csv`), written as CSV.
The `--stats` summary of rocprofv3 only has means, and the means of this library's kernels include the launches that
exit early once `ctrl->done` is set (the natural solve in front of the timed solves ends with a few of them).

    python tools/kernel_medians.py gpurun_out/prof_<tag> [out.csv]"""
import csv
import glob
import os
import statistics
import sys


def main():
    d = sys.argv[1]
    f = d if d.endswith(".csv") else sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    dur = {}
    for r in csv.DictReader(open(f)):
        dur.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(((sum(v), k, v) for k, v in dur.items()), reverse=True)
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "MedianNs", "MeanNs", "MinNs", "MaxNs"])
    for _, k, v in rows:
        w.writerow([k, len(v), int(statistics.median(v)), int(statistics.fmean(v)), min(v), max(v)])


if __name__ == "__main__":
    main()
