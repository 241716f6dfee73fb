#!/usr/bin/env python3
"""Wave timeline of k_eval_gram from a -DTSCM_WAVE_TIMELINE build (make variant VARIANT=T EXTRA=-DTSCM_WAVE_TIMELINE,
installed as libtscm_hip.so): runs one 10-iteration solve of a BASELINE config, reads the per-wave records (hardware
slot, start, end in 10 ns ticks) of the evaluation of LM iteration 5 through tscm_debug_wave_timeline, and prints how
the waves that shared a SIMD finished: the k-th wave to finish on its SIMD, averaged over all SIMDs.

    python tools/wave_timeline.py [--config 4]        (GPU box)
    python tools/wave_timeline.py file.log ...        (`EGW block wave hw_id xcc_id t0 t1` lines)"""
import collections
import re
import statistics
import sys


def record(config):
    import ctypes
    import os
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tscm_calib_amd import api, lib, synth
    p = synth.make_config(config).normalised()
    with api.Solver(p) as s:
        s.solve(max_num_iterations=10, function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0,
                min_trust_region_radius=0.0)
        buf = np.zeros(4 * 8192, dtype=np.int64)
        n = lib.lib().tscm_debug_wave_timeline(buf.ctypes.data_as(ctypes.c_void_p), 8192)
        if n <= 0:
            raise SystemExit(f"tscm_debug_wave_timeline: {n}")
        ph = np.zeros(5 * 8192, dtype=np.int64)
        if hasattr(lib.lib(), "tscm_debug_wave_phases") and lib.lib().tscm_debug_wave_phases(ph.ctypes.data_as(ctypes.c_void_p), 8192) > 0 and ph.any():
            ph = ph.reshape(-1, 5)[:n]
            ph = ph[ph.sum(axis=1) > 0]
            nv = p.n_views / max(1, len(ph))
            names = ["geometry", "MFMA u-rows", "v-row copy", "MFMA v-rows", "epilogue"]
            print("shader clocks per view and wave (mean over the waves, k_eval_gram4): " +
                  ", ".join(f"{nm} {ph[:, k].mean() / nv:.0f}" for k, nm in enumerate(names)) + f"; total {ph.sum(axis=1).mean() / nv:.0f}")
    rows = []
    for w in range(n):
        hw, xcc, t0, t1 = (int(x) for x in buf[4 * w:4 * w + 4])
        if t1 > 0:
            rows.append((w // 4, w % 4, hw, xcc, t0, t1))
    return rows


def main():
    if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
        lines = []
        for f in sys.argv[1:]:
            lines += re.findall(r"EGW (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", open(f).read())
        rows = [tuple(int(x) for x in a) for a in lines]
    else:
        rows = record(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    if not rows:
        print("no EGW lines")
        return
    # launches are separated in time: split where t0 jumps by more than 20 us
    rows.sort(key=lambda r: r[4])
    launches, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[4] - cur[0][4] > 5000:
            launches.append(cur)
            cur = []
        cur.append(r)
    launches.append(cur)
    print(f"{len(rows)} waves in {len(launches)} launch(es); analysing the last one ({len(launches[-1])} waves)")
    L = launches[-1]
    tmin = min(r[4] for r in L)
    simd = collections.defaultdict(list)
    for blk, wave, hw, xcc, t0, t1 in L:
        # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        key = (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
        simd[key].append(((t0 - tmin) / 100.0, (t1 - tmin) / 100.0, blk, hw & 15))
    occ = collections.Counter(len(v) for v in simd.values())
    print("SIMDs in use", len(simd), " waves per SIMD:", dict(sorted(occ.items())))
    kmax = max(occ)
    for k in range(kmax):
        ends = [sorted(v, key=lambda x: x[1])[k][1] for v in simd.values() if len(v) > k]
        starts = [sorted(v, key=lambda x: x[1])[k][0] for v in simd.values() if len(v) > k]
        print(f"  {k + 1}. wave to finish on its SIMD: end mean {statistics.fmean(ends):6.1f} us  (min {min(ends):6.1f}  max {max(ends):6.1f}),"
              f" start mean {statistics.fmean(starts):5.1f} us")
    ends = [r[5] for r in L]
    print(f"kernel (first start -> last end) {(max(ends) - tmin) / 100.0:.1f} us; wave duration mean "
          f"{statistics.fmean((r[5] - r[4]) / 100.0 for r in L):.1f} us")
    # does the finishing order follow the dispatch order (block index)?
    agree = sum(1 for v in simd.values() if [x[2] for x in sorted(v, key=lambda x: x[1])] == sorted(x[2] for x in v))
    print(f"SIMDs whose waves finish in block-index order: {agree} of {len(simd)}")


if __name__ == "__main__":
    main()
