#!/usr/bin/env python3
"""Measurement of the corner-candidate path (tscm_detect_corners, SURVEY 8f rank 4 first stage) on one MI355X.

Workload: one 1280 x 1080 synthetic fisheye image of a 9 x 6 chessboard (the reference's image size), sigma = 4.
Prints ONE JSON line: images/s from the device time of the kernels, an HBM roofline entry for the per-pixel
kernels, and the CPU oracle timed on the same image."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tscm_calib_amd import corners, synth  # noqa: E402

# algorithmic HBM bytes per pixel of the per-pixel kernels (corner candidates): grey 1 r (extremes) + 1 r (row
# pass); row pass 8 w; column pass 8 r + 8 w; metric 8 r + 16 w; suppression 8 r.  (The edge-angle / gradient planes of the
# reference are evaluated on demand around the maxima and never stored.)
BYTES_PER_PIXEL = 58
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--repeats", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--batch", type=int, default=32, help="images per tscm_detect_corners_batch call (second measurement)")
    a = ap.parse_args()
    p = synth.make_problem(1, 6, 3, noise_px=0.0, perturb=False)
    img = synth.render_chessboard(p.meta["gt_intr"][0], p.meta["gt_board_rt"][0], 9, 6, 45.0, a.width, a.height, supersample=2)
    corners.detect_corners(img)                              # warm-up
    dev, wall = [], []
    for _ in range(a.repeats):
        t0 = time.perf_counter()
        d = corners.detect_corners(img)
        wall.append(time.perf_counter() - t0)
        dev.append(d["seconds"])
    sec = float(np.median(dev))
    npix = a.width * a.height
    out = dict(metric="corner_detection_images_per_second", value=1.0 / sec, unit="images/s", n_gpus=1, higher_is_better=True, dtype="f64",
               data="synthetic", config=dict(workload=f"{a.width} x {a.height} grey image, 9 x 6 chessboard through the Triple Sphere model, sigma = 4"),
               device_ms=1e3 * sec, call_ms_incl_alloc_and_copies=1e3 * float(np.median(wall)), candidates=int(d["n"]), maxima=int(d["n_maxima"]),
               roofline=dict(bound="hbm", achieved=BYTES_PER_PIXEL * npix / sec / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                             frac=BYTES_PER_PIXEL * npix / sec / 1e9 / PEAK_HBM_GBS, traffic=None))
    if a.batch > 1:
        imgs = [img] * a.batch
        corners.detect_corners_batch(imgs)
        bdev, bwall = [], []
        for _ in range(max(3, a.repeats // 6)):
            t0 = time.perf_counter()
            r = corners.detect_corners_batch(imgs)
            bwall.append(time.perf_counter() - t0)
            bdev.append(sum(x["seconds"] for x in r))
        bsec = float(np.median(bdev)) / a.batch
        out["batched"] = dict(images_per_call=a.batch, value=1.0 / bsec, unit="images/s", device_ms_per_image=1e3 * bsec,
                              call_ms_per_image_incl_copies=1e3 * float(np.median(bwall)) / a.batch,
                              roofline=dict(bound="hbm", achieved=BYTES_PER_PIXEL * npix / bsec / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                                            frac=BYTES_PER_PIXEL * npix / bsec / 1e9 / PEAK_HBM_GBS, traffic=None))
    if not a.no_cpu:
        from oracle import pyoracle as orc
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            o = orc.detect_corners(img)
        cpu = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = dict(value=1.0 / cpu, unit="images/s", cores=1, kind="port", sample=f"the same image, {reps} runs")
        out["matches_cpu"] = bool(int((o["score"] >= 0.01).sum()) == d["n"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
