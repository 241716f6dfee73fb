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
        if hasattr(lib.lib(), "tscm_debug_wave_views"):
            tv = np.zeros(32 * 8192, dtype=np.int64)
            w = lib.lib().tscm_debug_wave_views(tv.ctypes.data_as(ctypes.c_void_p), 8192)
            if w > 0:
                fixed_cost(buf.reshape(-1, 4)[:n], tv[:w * 8192].reshape(-1, w)[:n])
    rows = []
    for w in range(n):
        hw, xcc, t0, t1 = (int(x) for x in buf[4 * w:4 * w + 4])
        if t1 > 0:
            rows.append((w // 4, w % 4, hw, xcc, t0, t1))
    return rows


def fixed_cost(tl, tv):
    """Where the time of a launch goes that is not per-view work (k_eval_gram4, -DTSCM_WAVE_TIMELINE): per wave the head
    (start -> view loop -> first MFMA), every view's duration by its position in the chunk, the tail (last record stored ->
    end of the loop -> camera tile handed on), and per SIMD the number of waves still in their view loop over the last
    microseconds of the launch.  All stamps s_memrealtime (10 ns)."""
    import numpy as np
    ok = (tl[:, 3] > 0) & (tv[:, 0] > 0)
    tl, tv = tl[ok], tv[ok]
    t0 = tl[:, 2].min()
    us = lambda a: (a - t0) / 100.0
    start, loop, mfma1, end2, lastrec, end1 = us(tl[:, 2]), us(tv[:, 0]), us(tv[:, 1]), us(tv[:, 2]), us(tv[:, 3]), us(tl[:, 3])
    views = tv[:, 4:]
    nv = (views > 0).sum(axis=1)
    def q(a):
        return f"mean {a.mean():6.2f}  p10 {np.percentile(a, 10):6.2f}  median {np.median(a):6.2f}  p90 {np.percentile(a, 90):6.2f}  max {a.max():6.2f}"
    print(f"fixed cost of the launch, {len(tl)} waves, views per wave {nv.min()}..{nv.max()}   [us]")
    print(f"  kernel: first wave start 0.00, last wave start {start.max():.2f}, last view loop end {end1.max():.2f}, last wave end {end2.max():.2f}")
    print(f"  wave start                          {q(start)}")
    print(f"  head: start -> view loop            {q(loop - start)}")
    print(f"  head: view loop -> first MFMA       {q(mfma1 - loop)}")
    print(f"  first MFMA at                       {q(mfma1)}")
    nmax = int(nv.max())
    for i in range(nmax):
        have = nv > i
        nxt = np.where(nv > i + 1, views[:, min(i + 1, views.shape[1] - 1)], tl[:, 3])
        if i + 1 >= views.shape[1]:
            break
        d = (nxt[have] - views[have, i]) / 100.0
        print(f"  view {i:2d} duration ({have.sum():4d} waves)      {q(d)}")
    print(f"  last record stored at               {q(lastrec)}")
    print(f"  view loop ends at                   {q(end1)}")
    print(f"  tail: loop end -> tile handed on    {q(end2 - end1)}")
    print(f"  wave: in the view loop              {q(end1 - loop)}")
    print(f"  wave: whole                         {q(end2 - start)}")
    # waves of a SIMD still in their view loop, against time
    import collections
    simd = collections.defaultdict(list)
    for i in range(len(tl)):
        hw, xcc = int(tl[i, 0]), int(tl[i, 1])
        simd[(xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)].append(i)
    tend = end1.max()
    print("  waves of a SIMD still in their view loop (mean over the SIMDs in use; a full SIMD holds 4):")
    for dt in (10, 8, 6, 5, 4, 3, 2, 1.5, 1, 0.5):
        t = tend - dt
        cnt = [sum(1 for i in v if loop[i] <= t < end1[i]) for v in simd.values()]
        print(f"    {dt:4.1f} us before the last loop end: {np.mean(cnt):.2f}   (SIMDs with 0/1/2/3/4: {[cnt.count(k) for k in range(5)]})")
    print("  ... and from the start:")
    for t in (0.5, 1, 1.5, 2, 3, 4, 5, 6, 8):
        cnt = [sum(1 for i in v if mfma1[i] <= t) for v in simd.values()]
        print(f"    {t:4.1f} us after the first wave start: waves past their first geometry {np.mean(cnt):.2f} per SIMD")


def placement(L, tmin):
    """Is the order in which the waves of a SIMD finish a function of the workgroup index?  (If it is, the chunk table could
    hand the wave that the arbiter favours more views.)  Prints, per rank of a workgroup on its CU by block index, when its
    waves end; how often that rank is blockIdx / 256 (round-robin placement over 8 XCDs x 32 CUs); and how often the waves of
    a SIMD finish in block order."""
    cu = collections.defaultdict(set)
    for blk, wave, hw, xcc, t0, t1 in L:
        cu[(xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].add(blk)
    rank = {}
    for v in cu.values():
        for k, b in enumerate(sorted(v)):
            rank[b] = k
    by_rank = collections.defaultdict(list)
    start_rank = collections.defaultdict(list)
    for blk, wave, hw, xcc, t0, t1 in L:
        by_rank[rank[blk]].append((t1 - tmin) / 100.0)
        start_rank[rank[blk]].append((t0 - tmin) / 100.0)
    print("  workgroups per CU:", dict(sorted(collections.Counter(len(v) for v in cu.values()).items())),
          " rank on the CU == blockIdx / 256 for %.1f %% of the workgroups;" % (100.0 * sum(1 for b, k in rank.items() if k == b // 256) / len(rank)),
          " XCD == blockIdx %% 8 for %.1f %% of the waves" % (100.0 * sum(1 for r in L if (r[3] & 15) == r[0] % 8) / len(L)))
    for k in sorted(by_rank):
        print(f"  waves of the {k + 1}. workgroup of a CU (by block index): start mean {statistics.fmean(start_rank[k]):5.2f}, end mean {statistics.fmean(by_rank[k]):6.2f} us"
              f"  (p10 {sorted(by_rank[k])[len(by_rank[k]) // 10]:6.2f}  p90 {sorted(by_rank[k])[9 * len(by_rank[k]) // 10]:6.2f})")
    simd = collections.defaultdict(list)
    for blk, wave, hw, xcc, t0, t1 in L:
        simd[(xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)].append((t1, blk))
    full = [v for v in simd.values() if len(v) == 4]
    in_order = sum(1 for v in full if [b for _, b in sorted(v)] == sorted(b for _, b in v))
    rev_order = sum(1 for v in full if [b for _, b in sorted(v)] == sorted((b for _, b in v), reverse=True))
    pos = collections.Counter()
    for v in full:
        order = [b for _, b in sorted(v)]
        blks = sorted(order)
        for k, b in enumerate(order):
            pos[(blks.index(b), k)] += 1
    print(f"  SIMDs with four waves: {len(full)}; they finish in block order on {in_order}, in reverse block order on {rev_order}")
    print("  rows: rank by block index; columns: 1st..4th to finish")
    for r in range(4):
        print("   ", [pos[(r, k)] for k in range(4)])


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
    placement(L, tmin)
    ends = [r[5] for r in L]
    print(f"kernel (first start -> last end) {(max(ends) - tmin) / 100.0:.1f} us; wave duration mean "
          f"{statistics.fmean((r[5] - r[4]) / 100.0 for r in L):.1f} us")
    # does the finishing order follow the dispatch order (block index)?
    agree = sum(1 for v in simd.values() if [x[2] for x in sorted(v, key=lambda x: x[1])] == sorted(x[2] for x in v))
    print(f"SIMDs whose waves finish in block-index order: {agree} of {len(simd)}")


if __name__ == "__main__":
    main()
