#!/usr/bin/env python3
"""Per-workgroup phase stamps of the two record-streaming kernels, k_schur_gram and k_backsub_prep, from a
-DTSCM_WAVE_TIMELINE build (make variant VARIANT=T EXTRA=-DTSCM_WAVE_TIMELINE, installed as libtscm_hip.so): one 10-iteration
solve, the launches of LM iteration 5.  Per kernel: when its workgroups start and end, how long each phase takes by the
order in which the workgroups started (the grid's rounds), and how many workgroups are in their LOAD phase against time -- a
kernel streams if that number is constant, it runs in lock step if it is a square wave.

    python tools/phase_timeline.py [--config 5]        (GPU box)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tscm_calib_amd import api, lib, synth        # noqa: E402


def report(name, st, phases, n_cam_blocks=0):
    st_all = st
    st = st[st[:, 0] > 0]
    if not len(st):
        print(f"{name}: no stamps")
        return
    red = st[(st[:, 7] >> 32) > 0]
    if len(red):      # k_schur_gram<NV, true>: the first workgroups take a reduction block of the evaluation in front of their chunk
        t0r = st[:, 0].min()
        rd = (red[:, 7] >> 32) / 100.0
        print(f"{name}: {len(red)} workgroups with a reduction block in front: block done {rd.mean():.2f} us after their start (max {rd.max():.2f}; "
              f"last {((red[:, 0] - t0r) / 100.0 + rd).max():.2f} us after the launch's first start)")
        if n_cam_blocks:
            allr = (st_all[:, 7] >> 32) / 100.0
            camr, brdr = allr[1:1 + n_cam_blocks], allr[1 + n_cam_blocks:]
            camr, brdr = camr[camr > 0], brdr[brdr > 0]
            if len(camr) and len(brdr):
                print(f"  camera-tile blocks: done after {camr.mean():.2f} us (max {camr.max():.2f}); board-statistics blocks: {brdr.mean():.2f} (max {brdr.max():.2f})")
        w = ((st[:, 7] & 0xffffffff) >> 1) / 100.0
        print(f"  the count was seen complete {w.mean():.2f} us after a workgroup's start (min {w.min():.2f}, max {w.max():.2f}); "
              f"that is {((st[:, 0] - t0r) / 100.0 + w).mean():.2f} us after the launch's first start (max {((st[:, 0] - t0r) / 100.0 + w).max():.2f})")
    t0 = st[:, 0].min()
    us = (st[:, :6] - t0) / 100.0
    order = np.argsort(us[:, 0], kind="stable")
    us = us[order]
    n = len(us)
    print(f"{name}: {n} workgroups, first start 0.00, last start {us[:, 0].max():.2f}, last end {us[:, 5].max():.2f} us; boards per workgroup {int(st[:, 6].min())}..{int(st[:, 6].max())}")
    q = max(1, n // 8)
    print("  started as   " + "  ".join(f"{p:>12s}" for p in ["start"] + phases + ["whole"]))
    for k in range(0, n, q):
        sl = us[k:k + q]
        d = [sl[:, 0].mean()] + [(sl[:, i + 1] - sl[:, i]).mean() for i in range(len(phases))] + [(sl[:, len(phases)] - sl[:, 0]).mean()]
        print(f"  {k:5d}-{min(n, k + q) - 1:5d}  " + "  ".join(f"{x:12.2f}" for x in d))
    # workgroups in each phase against time
    tend = us[:, len(phases)].max()
    print("  t [us]   " + "  ".join(f"{p:>10s}" for p in phases) + "   (workgroups in that phase)")
    for t in np.arange(0.0, tend, max(1.0, tend / 24)):
        cnt = [int(((us[:, i] <= t) & (t < us[:, i + 1])).sum()) for i in range(len(phases))]
        print(f"  {t:6.1f}   " + "  ".join(f"{c:10d}" for c in cnt))


def main():
    cfg = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 5
    p = synth.make_config(cfg).normalised()
    with api.Solver(p) as s:
        s.solve(max_num_iterations=10, function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, min_trust_region_radius=0.0)
        G = 2048
        buf = np.zeros(3 * 8 * G, dtype=np.int64)
        w = lib.lib().tscm_debug_phase_stamps(buf.ctypes.data_as(ctypes.c_void_p), G)
        if w <= 0:
            raise SystemExit(f"tscm_debug_phase_stamps: {w} (needs a -DTSCM_WAVE_TIMELINE build)")
        st = buf.reshape(3, G, w)
        rb = st[2][st[2][:, 0] > 0]
        if len(rb):
            t0 = st[0][st[0][:, 0] > 0][:, 0].min()
            for kind, name in ((1, "camera-tile blocks"), (0, "board-statistics blocks")):
                r = rb[rb[:, 4] == kind]
                if len(r):
                    a = (r[:, :4] - t0) / 100.0
                    print(f"{name} ({len(r)}), us after the launch's first start: start {a[:, 0].mean():.2f}, results computed and stores issued {a[:, 1].mean():.2f} (max {a[:, 1].max():.2f}), "
                          f"stores acknowledged {a[:, 2].mean():.2f} (max {a[:, 2].max():.2f}), counted in {a[:, 3].mean():.2f} (max {a[:, 3].max():.2f})")
        report("k_schur_gram", st[0], ["head/control", "records+E sums", "factor", "Gram", "tiles"], n_cam_blocks=16 * p.n_cameras)
        report("k_backsub_prep / riders", st[1][:, [0, 1, 2, 3, 4, 5, 6, 7]][:, [1, 2, 3, 4, 5, 5, 6, 7]], ["W.yhat", "board solve", "view constants"])


if __name__ == "__main__":
    main()
