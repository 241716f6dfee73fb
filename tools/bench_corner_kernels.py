#!/usr/bin/env python3
"""Per-kernel device time of the corner path for a batch of images (rocprofv3-free: runs the batch N times and prints
the library's own device time).  usage: python tools/bench_corner_kernels.py [--batch 32]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tscm_calib_amd import corners, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
p = synth.make_problem(1, 6, 3, noise_px=0.0, perturb=False)
img = synth.render_chessboard(p.meta["gt_intr"][0], p.meta["gt_board_rt"][0], 9, 6, 45.0, 1280, 1080, supersample=1)
imgs = [img] * a.batch
corners.detect_corners_batch(imgs)
t = [sum(x["seconds"] for x in corners.detect_corners_batch(imgs)) for _ in range(5)]
print('{"batch": %d, "device_us_per_image": %.2f}' % (a.batch, 1e6 * float(np.median(t)) / a.batch))
