#!/usr/bin/env python3
"""Measurement of the remap-table path (tscm_build_maps) on one MI355X.

Workload: the undistort_chessboard table (TS.cpp:308-330: 450 x 315 pixels for the 9x6 / 45 mm
board) of `--views` views of BASELINE config 4's camera 0 -- main.cpp:65 builds one per view
between the two mono solves -- as one launch.  Prints ONE JSON line: pixels/s of the map kernel
(HIP events, outputs resident in HBM), the HBM-write roofline and the CPU oracle on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tscm_calib_amd import maps, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0
BYTES_PER_PIXEL = 8            # two float32 tables; the descriptors are 224 B per 141,750 pixels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2000)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--cpu-maps", type=int, default=40)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    p = synth.make_problem(4, a.views, 20244)
    views = np.nonzero(p.view_camera == 0)[0][: a.views]
    descs, off = [], 0
    for v in views:
        rt = p.meta["gt_board_rt"][p.view_board[v]]
        R = synth.rodrigues(rt[:3])
        d = maps.chessboard_desc(p.meta["gt_intr"][0], np.stack([R[:, 0], R[:, 1], rt[3:]], axis=1), 9, 6, 45.0, out_offset=off)
        descs.append(d)
        off += (d.width * d.height + 3) // 4 * 4       # every table 16-byte aligned, like separately allocated cv::Mat
    out = dict(metric="remap_table_pixels_per_second", unit="pixels/s", n_gpus=1, higher_is_better=True, dtype="f64->f32",
               data="synthetic", config=dict(workload=f"{len(descs)} undistort_chessboard tables of 450x315 pixels ({off} pixels)"))
    for exact in (False, True):
        maps.build_maps(descs, off, exact=exact)
        sec = min(maps.build_maps(descs, off, exact=exact)[2] for _ in range(a.repeats))
        t0 = time.perf_counter()
        maps.build_maps(descs, off, exact=exact)
        wall = time.perf_counter() - t0
        key = "exact" if exact else "fast"
        out[key] = dict(seconds_kernel=sec, pixels_per_second=off / sec, seconds_call_incl_pcie=wall)
    out["value"] = out["fast"]["pixels_per_second"]
    gbs = out["value"] * BYTES_PER_PIXEL / 1e9
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_side_kernels.json")))["k_build_maps"]
        if a.views == 2000:
            traffic = pm["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    out["roofline"] = dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, traffic=traffic,
                           exact_variant_GBs=out["exact"]["pixels_per_second"] * BYTES_PER_PIXEL / 1e9)
    if not a.no_cpu:
        from oracle import pyoracle as orc   # cpu_baseline leg only
        sample = descs[: a.cpu_maps]
        n = sample[-1].out_offset + sample[-1].width * sample[-1].height
        t0 = time.perf_counter()
        orc.build_maps(sample, n)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=n / dt, unit="pixels/s", cores=1, kind="port", sample=f"{len(sample)} of the {len(descs)} tables")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
