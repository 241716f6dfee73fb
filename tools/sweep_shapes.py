"""Ad-hoc GPU <-> oracle sweep over board shapes and camera counts for every kernel family (LM solve, rig init,
focal estimate, planar PnP).  Prints one line per case and a failure count; exits non-zero on any mismatch.
Run on a GPU box: python tools/sweep_shapes.py"""
import sys
import traceback
import numpy as np
sys.path.insert(0, ".")
from oracle import pyoracle as orc
from tscm_calib_amd import api, rig, synth
from tests import helpers as H

shapes = [(9, 6), (11, 8), (10, 6), (5, 4), (4, 3), (7, 5), (8, 8), (13, 5), (9, 7), (12, 9), (6, 5), (3, 3), (14, 10)]
fails = 0


def check(name, fn):
    global fails
    try:
        fn()
        print("ok  ", name, flush=True)
    except Exception as e:          # noqa: BLE001
        if getattr(e, "code", None) == -5:          # TSCM_E_UNSUPPORTED: a documented limit (board width 4..32 for the focal estimate)
            print("skip", name, str(e)[:120], flush=True)
            return
        fails += 1
        print("FAIL", name, repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)


def solve_case(C, cols, rows, seed):
    def f():
        p = synth.make_problem(C, 8, seed, cols=cols, rows=rows, pitch=360.0 / max(cols, rows))
        pg, po = p.copy().normalised(), p.copy().normalised()
        with api.Solver(pg) as s:
            gs = s.solve(max_num_iterations=3)
        os_ = orc.solve(po, max_num_iterations=3)
        assert gs["num_iterations"] == os_["num_iterations"]
        for a, b in zip(gs["iterations"], os_["iterations"]):
            assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"]), (a["cost"], b["cost"])
        d = H.param_rel_err(pg, po)
        assert max(d.values()) < 1e-7, d
    return f


def rig_case(C, cols, rows, seed):
    def f():
        inp = synth.make_rig_input(synth.make_problem(C, 8, seed, cols=cols, rows=rows, pitch=360.0 / max(cols, rows)))
        g, o = rig.rig_init(inp, 0), orc.rig_init(inp)
        assert np.max(np.abs(g["cam_rt"] - o["cam_rt"])) < 1e-9
        assert np.max(np.abs(g["board_t"] - o["board_t"])) < 1e-10
        assert np.max(np.abs(g["board_R"] - o["board_R"])) < 1e-13
    return f


def init_case(cols, rows, seed):
    def f():
        p = synth.make_problem(1, 12, seed, noise_px=0.05, perturb=False, cols=cols, rows=rows, pitch=360.0 / max(cols, rows))
        V, n = p.n_views, cols * rows
        pu, pv = p.obs_u.reshape(V, n), p.obs_v.reshape(V, n)
        count = np.full(V, n, dtype=np.int32)
        fo, no, rc = orc.estimate_focal(pu, pv, count, cols, rows, 639.5, 539.5)
        fg, ng = rig.estimate_focal(pu, pv, count, cols, rows, 639.5, 539.5, 0)
        assert ng == no, (ng, no)
        if no:
            assert abs(fg - fo) < 1e-8 * fo, (fg, fo)
        W = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
        intr = p.meta["gt_intr"][0]
        Ro, ko = orc.estimate_extrinsic(intr, pu, pv, count, W, cols)
        Rg, kg = rig.estimate_extrinsic(intr, pu, pv, count, W, cols, 0)
        assert kg == ko
        assert np.max(np.abs(Rg[:, :, :2] - Ro[:, :, :2])) < 1e-7
        assert np.max(np.abs(Rg[:, :, 2] - Ro[:, :, 2])) < 1e-6 * np.max(np.abs(Ro[:, :, 2]))
    return f


for i, (c, r) in enumerate(shapes):
    for C in (1, 2, 3, 5, 7, 8):
        check(f"solve C={C} {c}x{r}", solve_case(C, c, r, 40 + i))
    for C in (2, 3, 5):
        check(f"rig   C={C} {c}x{r}", rig_case(C, c, r, 70 + i))
    check(f"init  {c}x{r}", init_case(c, r, 90 + i))
for (c, r) in [(25, 20), (40, 30)]:
    check(f"solve C=2 {c}x{r}", solve_case(2, c, r, 300 + c))
    check(f"solve C=1 {c}x{r}", solve_case(1, c, r, 310 + c))
    check(f"rig   C=3 {c}x{r}", rig_case(3, c, r, 320 + c))
    check(f"init  {c}x{r}", init_case(c, r, 330 + c))
for C in (9, 12, 20):
    check(f"rig   C={C} 9x6", rig_case(C, 9, 6, 400 + C))
print("failures:", fails)
sys.exit(1 if fails else 0)
