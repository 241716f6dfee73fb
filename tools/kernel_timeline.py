#!/usr/bin/env python3
"""Workgroup timeline of the six kernels of one LM iteration, from a -DTSCM_WAVE_TIMELINE build installed as
libtscm_hip.so (make variant VARIANT=T EXTRA=-DTSCM_WAVE_TIMELINE): every workgroup's thread 0 stamps its start and
end (s_memrealtime, 10 ns) during LM iteration 5 of a forced 10-iteration solve; tscm_debug_kernel_timeline copies the
stamps out.  Per kernel, relative to the end of the kernel before it: when its first / last workgroup started (launch
gap, dispatch ramp), when the first / last one ended (tail), and the median workgroup duration.

    python tools/kernel_timeline.py [--config 4]        (GPU box)"""
import ctypes
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tscm_calib_amd import api, lib, synth  # noqa: E402

NAMES = ["k_eval_gram4", "k_reduce_control", "(k_finalize_eval: comm path only)", "k_schur_gram", "k_solve_reduced", "k_backsub_prep"]
ORDER = [3, 4, 5, 0, 1, 2]          # launch order inside an LM iteration
GROUPS = 2048


def main():
    config = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    p = synth.make_config(config).normalised()
    with api.Solver(p) as s:
        s.upload_params()
        s.solve_resident(reset=True)
        for rep in range(3):        # the last of three solves: a device at its sustained clocks
            s.solve_resident(reset=True, max_num_iterations=10, function_tolerance=-1.0, parameter_tolerance=-1.0,
                             gradient_tolerance=-1.0, min_trust_region_radius=0.0, check_every=255)
        buf = np.zeros(2 * len(NAMES) * GROUPS, dtype=np.int64)
        n = lib.lib().tscm_debug_kernel_timeline(buf.ctypes.data_as(ctypes.c_void_p), GROUPS)
        if n <= 0:
            raise SystemExit(f"tscm_debug_kernel_timeline: {n} (not a -DTSCM_WAVE_TIMELINE build?)")
        xs = np.zeros(32, dtype=np.int64)
        if hasattr(lib.lib(), "tscm_debug_control_stamps"):
            lib.lib().tscm_debug_control_stamps(xs.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(len(NAMES), GROUPS, 2)
    prev_end, t_first = None, None
    print(f"{'kernel':18s} {'WGs':>5s} {'gap':>6s} {'ramp':>6s} {'first end':>9s} {'last end':>8s} {'WG median':>9s} {'kernel':>7s}   [us]")
    for k in ORDER:
        rows = t[k][t[k][:, 1] > 0]
        if not len(rows):
            continue
        s0, s1, e0, e1 = rows[:, 0].min(), rows[:, 0].max(), rows[:, 1].min(), rows[:, 1].max()
        if t_first is None:
            t_first = s0
        gap = (s0 - prev_end) / 100.0 if prev_end is not None else 0.0
        print(f"{NAMES[k]:18s} {len(rows):5d} {gap:6.2f} {(s1 - s0) / 100.0:6.2f} {(e0 - s0) / 100.0:9.2f} {(e1 - s0) / 100.0:8.2f} "
              f"{statistics.median((rows[:, 1] - rows[:, 0]).tolist()) / 100.0:9.2f} {(e1 - (prev_end if prev_end is not None else s0)) / 100.0:7.2f}")
        prev_end = e1
    nc = p.n_cameras * 16
    rc = t[1]
    cam, brd = rc[:nc][rc[:nc, 1] > 0], rc[nc:][rc[nc:, 1] > 0]
    if len(cam) and len(brd):
        k0 = rc[rc[:, 1] > 0][:, 0].min()
        print(f"k_reduce_control: camera-tile workgroups ({len(cam)}) end {np.median(cam[:, 1] - k0) / 100.0:.2f} us after the first start "
              f"(max {(cam[:, 1].max() - k0) / 100.0:.2f}), board-statistics workgroups ({len(brd)}) {np.median(brd[:, 1] - k0) / 100.0:.2f} "
              f"(max {(brd[:, 1].max() - k0) / 100.0:.2f}); median durations {np.median(cam[:, 1] - cam[:, 0]) / 100.0:.2f} / {np.median(brd[:, 1] - brd[:, 0]) / 100.0:.2f}")
    if xs.any():
        k0 = t[1][t[1][:, 1] > 0][:, 0].min()
        lab = {0: "last arrival known", 1: "acquired", 2: "scalars reduced", 3: "H_stage written", 4: "control: norms", 5: "control: reduced",
               6: "control: scalars read", 8: "control done"}
        print("workgroup of the control step, us after the first start of its kernel: " +
              ", ".join(f"{lab[i]} {(xs[i] - k0) / 100.0:.2f}" for i in sorted(lab) if xs[i] > 0))
    print(f"iteration (first start of k_schur_gram -> last end of k_reduce_control): {(prev_end - t_first) / 100.0:.1f} us")
    print("gap: last end of the kernel before -> first start; ramp: first -> last workgroup start; kernel: last end before -> last end")


if __name__ == "__main__":
    main()
