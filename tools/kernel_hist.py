#!/usr/bin/env python3
"""Distribution of ONE kernel's launch durations in a rocprofv3 kernel trace, and the launch-by-launch sequence around it
(a median hides a second mode: e.g. the iterations behind a rejected LM step).

    python tools/kernel_hist.py gpurun_out/prof_<tag> <substring of the kernel's name> [launches to list]"""
import csv
import glob
import os
import sys


def main():
    d, pat = sys.argv[1], sys.argv[2]
    n_list = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    f = d if d.endswith(".csv") else sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    v = sorted(e - s for s, e, k in rows if pat in k)
    if not v:
        raise SystemExit("no launch of a kernel named *%s*" % pat)
    q = lambda p: v[min(len(v) - 1, int(p * len(v)))] / 1e3
    print(f"{pat}: {len(v)} launches  min {v[0] / 1e3:.2f}  p10 {q(.1):.2f}  p25 {q(.25):.2f}  median {q(.5):.2f}  p75 {q(.75):.2f}  p90 {q(.9):.2f}  max {v[-1] / 1e3:.2f} us")
    if n_list:
        idx = [i for i, r in enumerate(rows) if pat in r[2]]
        mid = idx[len(idx) // 2]
        print("launch by launch from the middle of the trace: start [us], duration [us], gap to the launch before [us], kernel")
        for i in range(mid, min(len(rows), mid + n_list)):
            s, e, k = rows[i]
            print(f"  {(s - rows[mid][0]) / 1e3:9.2f}  {(e - s) / 1e3:7.2f}  {(s - rows[i - 1][1]) / 1e3:6.2f}  {k[:60]}")


if __name__ == "__main__":
    main()
