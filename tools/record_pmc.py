#!/usr/bin/env python3
"""Turns the two PMC passes of tools/pmc.sh (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs of `bench.py --steps 10
--warmup 0 --no-cpu-baseline`) into profiles/pmc_eval_gram.json / pmc_side_kernels.json entries.

    python tools/record_pmc.py gpurun_out/pmc_<fetch tag> gpurun_out/pmc_<write tag> [config] [gpurun_out/pmc_<sq pass> ...]

Further pass directories (SQ_* / GRBM_* counters) are averaged over the k_eval_gram dispatches into the entry's
`sq_round<N>` block (N = TSCM_ROUND, default 5).

Units and corrections as MI355X_MICROARCH.md prescribes and tools/calib_fetch.hip confirmed on these boxes: both counters
report KB; FETCH_SIZE counts half of the bytes of this library's 8 B/lane and 16 B/lane loads (x2), WRITE_SIZE is exact.
The entry is stamped with the hash of the kernel sources it was measured on (bench.py only reports `roofline.traffic`
while that hash matches the sources it runs)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ROUND = int(os.environ.get("TSCM_ROUND", "6"))       # which round's evidence this is (keys `sq_round<N>`, `..._all_kernels_round<N>`)


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)      # (the newest pass: gpurun merges every call's files into the directory)
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tscm::", "")
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: tot[k] / n[k] for k in tot}, dict(n)


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    config = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    sq_dirs = sys.argv[4:]
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    sha = bench._kernel_src_sha()
    kernels = {}
    for k in sorted(fetch):
        rd, wr = 2.0 * fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
        kernels[k] = {"dispatches": nf[k], "FETCH_SIZE_KB_per_launch": fetch[k], "WRITE_SIZE_KB_per_launch": write.get(k, 0.0),
                      "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    path = os.path.join(ROOT, "profiles", "pmc_eval_gram.json")
    doc = json.load(open(path))
    # records of past rounds are append-only: a block of THIS round may be re-recorded, one of another round never
    for key in (f"config{config}_all_kernels_round{ROUND}",):
        if key in doc and doc[key].get("round", ROUND) != ROUND:
            raise SystemExit(f"{key} belongs to another round: refusing to overwrite it (set TSCM_ROUND)")
    if sq_dirs and f"sq_round{ROUND}" in doc.get(f"config{config}", {}) and doc[f"config{config}"][f"sq_round{ROUND}"].get("round", ROUND) != ROUND:
        raise SystemExit(f"config{config}.sq_round{ROUND} belongs to another round: refusing to overwrite it")
    ev = next(v for k, v in kernels.items() if k.startswith("k_eval_gram"))
    n_corners = {4: 2160000, 5: 8640000}.get(config)
    keep = {k: v for k, v in doc.get(f"config{config}", {}).items() if k.startswith("sq")}       # SQ counter blocks are recorded separately
    doc[f"config{config}"] = {
        "kernel_src_sha": sha, "round": ROUND,
        "FETCH_SIZE_KB_per_launch": ev["FETCH_SIZE_KB_per_launch"], "WRITE_SIZE_KB_per_launch": ev["WRITE_SIZE_KB_per_launch"],
        "read_bytes_corrected": ev["read_bytes_corrected"], "hbm_bytes_per_launch": ev["hbm_bytes_per_launch"],
        "algorithmic_bytes_per_launch": n_corners * bench.BYTES_PER_CORNER if n_corners else None,
        "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/pmc.sh), averaged over the k_eval_gram dispatches; "
                "hbm = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes). Writes are the kernel's OUTPUT: 102 doubles per view of Schur records (132 in round 2) + per-workgroup camera tiles.",
    }
    doc[f"config{config}"].update(keep)
    if sq_dirs:
        sq = {}
        for d in sq_dirs:
            f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)      # (the newest pass: gpurun merges every call's files into the directory)
            tot, n = collections.defaultdict(float), collections.Counter()
            for r in csv.DictReader(open(f)):
                if "k_eval_gram" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            for c in tot:
                sq[c] = tot[c] / n[c]
        sq["round"] = ROUND
        sq["passes"] = "tools/pmc.sh, one rocprofv3 --pmc pass per directory: " + ", ".join(os.path.basename(d) for d in sq_dirs) + "; per k_eval_gram dispatch"
        doc[f"config{config}"][f"sq_round{ROUND}"] = sq
    doc[f"config{config}_all_kernels_round{ROUND}"] = {"kernel_src_sha": sha, "round": ROUND, "per_launch": kernels}
    json.dump(doc, open(path, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:34s} read {v['read_bytes_corrected'] / 1e6:7.2f} MB  write {v['write_bytes'] / 1e6:7.2f} MB")


if __name__ == "__main__":
    main()
