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

NAMES = ["k_eval_gram4", "k_reduce_stats / k_reduce_control", "(k_finalize_eval: communicator path)", "k_schur_gram (+ control step in its head)",
         "k_solve_reduced (solver + T reduction)", "back-substitution workgroups (same launch)"]
ORDER = [3, 4, 5, 0, 1]             # launch order inside an LM iteration
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
    # (the stamps are taken while ctrl->iteration == 5: since the control step moved into k_schur_gram's head, the Schur /
    # solve / back-substitution stamps are of one LM iteration and the evaluation / reduction stamps of the next one)
    st = {}
    for k in ORDER:
        rows = t[k][t[k][:, 1] > 0]
        if len(rows):
            st[k] = (rows[:, 0].min(), rows[:, 0].max(), rows[:, 1].min(), rows[:, 1].max(), len(rows),
                     statistics.median((rows[:, 1] - rows[:, 0]).tolist()))
    print(f"{'launch / role':44s} {'WGs':>5s} {'gap':>6s} {'ramp':>6s} {'first end':>9s} {'last end':>8s} {'WG median':>9s}   [us, from the role's first start]")
    for k in ORDER:
        if k not in st:
            continue
        s0, s1, e0, e1, n, med = st[k]
        before = {1: 0}.get(k)                # the gap that the stamps of ONE iteration give: evaluation -> reductions
        gap = f"{(s0 - st[before][3]) / 100.0:6.2f}" if before in st else "     -"
        if k == 5 and 4 in st:
            gap = f"{(s0 - st[4][0]) / 100.0:+6.2f}"          # same launch: start relative to the solver's
        print(f"{NAMES[k]:44s} {n:5d} {gap} {(s1 - s0) / 100.0:6.2f} {(e0 - s0) / 100.0:9.2f} {(e1 - s0) / 100.0:8.2f} {med / 100.0:9.2f}")
    if 4 in st and 5 in st:
        print(f"solve launch: solver workgroup ends {(t[4][0][1] - st[4][0]) / 100.0:.2f} us after its start, the last back-substitution workgroup {(st[5][3] - st[4][0]) / 100.0:.2f}")
    nc = p.n_cameras * 16
    rc = t[1]
    cam, brd = rc[:nc][rc[:nc, 1] > 0], rc[nc:][rc[nc:, 1] > 0]
    if len(cam) and len(brd):
        k0 = rc[rc[:, 1] > 0][:, 0].min()
        print(f"reductions: camera-tile workgroups ({len(cam)}) end {np.median(cam[:, 1] - k0) / 100.0:.2f} us after the first start "
              f"(max {(cam[:, 1].max() - k0) / 100.0:.2f}), board-statistics workgroups ({len(brd)}) {np.median(brd[:, 1] - k0) / 100.0:.2f} "
              f"(max {(brd[:, 1].max() - k0) / 100.0:.2f}); median durations {np.median(cam[:, 1] - cam[:, 0]) / 100.0:.2f} / {np.median(brd[:, 1] - brd[:, 0]) / 100.0:.2f}")
    if xs.any() and xs[0] > 0:
        lab = {1: "counted in", 2: "scalars reduced", 3: "H formed", 4: "norms", 5: "reduced", 6: "scalars read", 8: "done"}
        print("k_reduce_control's last workgroup (the initial evaluation's control step), us after it knew it was last: " +
              ", ".join(f"{lab[i]} {(xs[i] - xs[0]) / 100.0:.2f}" for i in sorted(lab) if xs[i] > 0))
    print("gap: last end of the launch before -> first start (where one iteration's stamps give it); ramp: first -> last workgroup start")


if __name__ == "__main__":
    main()
