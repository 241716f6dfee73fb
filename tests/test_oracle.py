"""CPU tests of the oracle (oracle/tscm_oracle.c): known-answer values, Jacobians, LM behaviour.
Parity is unpinned against real Ceres (not installable here); these are the independent pins
listed in oracle/tscm_oracle.h."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import synth
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLD, "kat_ts.json")))
INTR = np.array(KAT["intrinsics"] + [0.0, 0.0])


def test_project_kat():
    for c in KAT["project"]:
        uv = orc.project(INTR, c["P"])
        assert abs(uv[0] - float(c["u"])) < 1e-12 * abs(float(c["u"]))
        assert abs(uv[1] - float(c["v"])) < 1e-12 * abs(float(c["v"]))


def test_unproject_is_inverse_of_project():
    for c in KAT["project"]:
        P = np.array(c["P"])
        ray = orc.unproject(INTR, [float(c["u"]), float(c["v"])])
        assert np.max(np.abs(ray - P / np.linalg.norm(P))) < 1e-13
        assert abs(np.linalg.norm(ray) - 1.0) < 1e-13


def test_functor_kat():
    for c in KAT["functor"]:
        if c["kind"] == "mono":
            r = orc.mono_residual(INTR, c["rt"], [0.0, 0.0], c["board_pt"])
        else:
            r = orc.multi_residual(c["cam_rt"], c["board_rt"], INTR, [0.0, 0.0], c["board_pt"])
        assert abs(-r[0] - float(c["u"])) < 1e-12 * float(c["u"])
        assert abs(-r[1] - float(c["v"])) < 1e-12 * float(c["v"])


def test_skew_terms_in_plain_projection_only():
    I = INTR.copy()
    I[7], I[8] = 0.7, -0.4
    P = np.array([120.0, -80.0, 300.0])
    uv0, uv1 = orc.project(INTR, P), orc.project(I, P)
    assert abs(uv0[0] - uv1[0]) > 1e-3          # TS.cpp:341-342 carry b, c
    r0 = orc.multi_residual(np.zeros(6), np.r_[0, 0, 0, P], INTR, [0, 0], [0, 0])
    r1 = orc.multi_residual(np.zeros(6), np.r_[0, 0, 0, P], I, [0, 0], [0, 0])
    assert np.all(r0 == r1)                     # functors ignore them (multi_calib.h:175-176)
    back = orc.unproject(I, uv1)
    assert np.max(np.abs(back - P / np.linalg.norm(P))) < 1e-12


def test_rotation_branches():
    pt = np.array([3.0, -2.0, 5.0])
    assert np.all(orc.rotate([0, 0, 0], pt) == pt)
    w = np.array([1.0e-9, -2e-9, 3e-9])          # theta^2 < DBL_EPSILON: pt + w x pt
    assert np.max(np.abs(orc.rotate(w, pt) - (pt + np.cross(w, pt)))) == 0.0
    w = np.array([0.3, -1.1, 0.7])
    R = orc.rodrigues(w)
    assert np.max(np.abs(R @ R.T - np.eye(3))) < 1e-15
    assert np.max(np.abs(orc.rotate(w, pt) - R @ pt)) < 1e-14
    assert np.max(np.abs(synth.rodrigues(w) - R)) < 1e-15
    assert np.max(np.abs(synth.rotmat_to_aa(R) - w)) < 1e-14


def _num_jac(f, x, h):
    J = np.zeros((2, x.size))
    for i in range(x.size):
        d = np.zeros_like(x); d[i] = h[i]
        J[:, i] = (f(x + d) - f(x - d)) / (2 * h[i])
    return J


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_autodiff_vs_central_differences(seed):
    rng = np.random.default_rng(seed)
    p = H.small_rig(4, 4, seed=seed + 40)
    v = int(rng.integers(p.n_views)); j = int(rng.integers(p.n_points))
    m, b = int(p.view_camera[v]), int(p.view_board[v])
    cam = p.meta["gt_cam_rt"][m].copy() if m else np.array([0.02, -0.01, 0.03, 5.0, -3.0, 2.0])
    brd, I = p.meta["gt_board_rt"][b], p.meta["gt_intr"][m]
    obs = [p.obs_u[p.view_offset[v] + j], p.obs_v[p.view_offset[v] + j]]
    bp = p.board_xy[j]
    res, Jc, Jb, Ji = orc.multi_autodiff(cam, brd, I, obs, bp)
    assert np.all(res == orc.multi_residual(cam, brd, I, obs, bp) ) or np.max(np.abs(res - orc.multi_residual(cam, brd, I, obs, bp))) < 1e-11
    hp = np.array([1e-6] * 3 + [1e-3] * 3)
    Jc_n = _num_jac(lambda x: orc.multi_residual(x, brd, I, obs, bp), cam, hp)
    Jb_n = _num_jac(lambda x: orc.multi_residual(cam, x, I, obs, bp), brd, hp)
    hi = np.array([1e-3, 1e-3, 1e-3, 1e-3, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6])
    Ji_n = _num_jac(lambda x: orc.multi_residual(cam, brd, x, obs, bp), I, hi)
    for a, n in ((Jc, Jc_n), (Jb, Jb_n), (Ji, Ji_n)):
        assert np.max(np.abs(a - n)) <= 2e-6 * max(1.0, np.abs(n).max())
    assert np.all(Ji[:, 7:] == 0.0)
    # mono functor = multi functor with an identity camera
    rm, Jim, Jrm = orc.mono_autodiff(I, brd, obs, bp)
    r0, _, Jb0, Ji0 = orc.multi_autodiff(np.zeros(6), brd, I, obs, bp)
    assert np.max(np.abs(rm - r0)) < 1e-12 and np.max(np.abs(Jim - Ji0)) < 1e-9 and np.max(np.abs(Jrm - Jb0)) < 1e-9


def test_lm_golden_traces():
    gold = json.load(open(os.path.join(GOLD, "lm_traces.json")))
    for name, p in (("config1_mono", synth.make_config(1)), ("rig4x6", synth.make_problem(4, 6, 11))):
        g = gold[name]
        q = p.copy().normalised()
        s = orc.solve(q)
        assert s["message"] == g["message"] and s["num_iterations"] == g["num_iterations"]
        assert np.allclose([it["cost"] for it in s["iterations"]], g["costs"], rtol=1e-9)
        assert np.allclose([it["trust_region_radius"] for it in s["iterations"]], g["radii"], rtol=1e-6)
        assert np.allclose(q.intr, g["intr"], rtol=1e-7, atol=1e-12)
        assert np.allclose(q.cam_rt, g["cam_rt"], rtol=1e-7, atol=1e-9)
        assert abs(orc.rmse(q) - g["rmse"]) < 1e-9


def test_lm_config1_behaviour():
    p = synth.make_config(1)
    q = p.copy().normalised()
    s = orc.solve(q)
    assert s["termination_type"] == 0
    assert s["message"] == "Function tolerance reached."
    costs = [it["cost"] for it in s["iterations"]]
    assert all(b < a for a, b in zip(costs, costs[1:]))
    assert abs(orc.rmse(q) - 0.1 * np.sqrt(2)) < 0.01
    assert np.all(q.intr[:, 7:] == 0.0)
    assert np.all(q.cam_rt == p.cam_rt)
    g, per = orc.mean_reprojection_error(q)
    assert 0.08 < g < 0.2 and abs(per[0] - g) < 1e-12


def test_lm_zero_noise_reaches_zero():
    p = synth.make_problem(1, 20, 99, noise_px=0.0)
    q = p.copy().normalised()
    orc.solve(q)
    assert orc.rmse(q) < 1e-4


def test_lm_matches_scipy_optimum():
    """Independent optimiser (MINPACK via SciPy) reaches the same optimum COST / RMSE from the
    same start.  (Not the same intrinsics: fx/xi/lambda/alpha trade off along a near-flat
    valley -- SURVEY H1 -- which is why parameter parity needs trajectory parity.)"""
    from scipy.optimize import least_squares
    p = synth.make_config(1)
    q = p.copy().normalised()
    s = orc.solve(q)
    B = p.n_boards

    def unpack(x, pr):
        pr.intr[0, :7] = x[:7]
        pr.board_rt[:] = x[7:].reshape(B, 6)

    work = p.copy().normalised()

    def fun(x):
        unpack(x, work)
        return orc.evaluate(work, jets=False)[1].ravel()

    def jac(x):
        unpack(x, work)
        _, _, _, Jb, Ji = orc.evaluate(work, jets=True)
        N = Jb.shape[0]
        J = np.zeros((2 * N, 7 + 6 * B))
        J[:, :7] = Ji[:, :, :7].reshape(2 * N, 7)
        rows = np.arange(2 * N)
        board = np.repeat(work.view_board, work.view_count)
        for k in range(6):
            J[rows, 7 + 6 * np.repeat(board, 2) + k] = Jb[:, :, k].reshape(2 * N)
        return J

    x0 = np.r_[p.intr[0, :7], p.board_rt.ravel()]
    # (MINPACK's own "lm" crawls along the valley for thousands of evaluations; the
    #  Jacobian-scaled dogleg gets there in ~200.)  The valley floor keeps sinking by ~3e-5
    #  relative in cost long after function_tolerance=1e-6 has fired, hence the 1e-4 band.
    r = least_squares(fun, x0, jac=jac, method="dogbox", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=300)
    assert abs(r.cost - s["final_cost"]) <= 1e-4 * s["final_cost"], (r.cost, s["final_cost"], r.status)
    assert r.cost <= s["final_cost"] * (1 + 1e-9) or abs(r.cost - s["final_cost"]) <= 1e-4 * s["final_cost"]


def test_constant_camera_and_inactive_blocks():
    p = H.small_rig(4, 6, seed=9)
    cnt = p.view_count.copy()
    cnt[0] = 0; cnt[1] = 0            # board 0 unseen
    p.view_count = cnt
    q = p.copy().normalised()
    s = orc.solve(q)
    assert s["termination_type"] == 0
    assert np.all(q.cam_rt[0] == p.cam_rt[0])
    assert np.all(q.board_rt[0] == p.board_rt[0])
    assert np.any(q.board_rt[1] != p.board_rt[1])


def test_invalid_steps_fail_like_ceres():
    """NaN observations -> every step invalid -> FAILURE after 5 consecutive invalid steps."""
    p = synth.make_config(1)
    q = p.copy().normalised()
    q.obs_u[3] = np.nan
    s = orc.solve(q)
    assert s["termination_type"] == 2


def test_openmp_passes_agree_with_the_sequential_checker():
    """orc_set_num_threads(n > 1) (the all-cores CPU baseline of bench.py, and the config-5 parity test) re-associates
    sums per thread: same iterations and decisions, parameters equal to rounding."""
    L = orc.lib()
    for p in (synth.make_problem(4, 40, 5), synth.make_problem(1, 60, 6)):
        a, b = p.copy().normalised(), p.copy().normalised()
        sa = orc.solve(a)
        L.orc_set_num_threads(4)
        try:
            sb = orc.solve(b)
        finally:
            L.orc_set_num_threads(1)
        assert sa["num_iterations"] == sb["num_iterations"] and sa["message"] == sb["message"]
        for x, y in zip(sa["iterations"], sb["iterations"]):
            assert x["step_is_successful"] == y["step_is_successful"]
            assert abs(x["cost"] - y["cost"]) <= 1e-11 * abs(y["cost"])
        assert np.max(np.abs(a.intr - b.intr) / np.maximum(np.abs(a.intr), 1e-3)) < 1e-9
        assert np.max(np.abs(a.board_rt - b.board_rt)) < 1e-7


def test_constant_board_poses_in_the_oracle():
    """tscm_problem.board_pose_constant (the poses-fixed form of BASELINE config 2): constant blocks leave the program --
    they are returned untouched, their residuals still count, and with exact data and exact poses the intrinsics are
    recovered."""
    p = synth.make_problem(1, 30, 5, noise_px=0.0)
    p.board_rt = p.meta["gt_board_rt"].copy()
    p.board_pose_constant = np.ones(p.n_boards, dtype=np.uint8)
    q = p.copy().normalised()
    s = orc.solve(q)
    assert s["termination_type"] == 0 and s["final_cost"] < 1e-12
    assert np.array_equal(q.board_rt, p.board_rt)
    assert np.max(np.abs(q.intr[0, :7] - p.meta["gt_intr"][0, :7]) / np.abs(p.meta["gt_intr"][0, :7])) < 1e-5
    # a rig with every second board constant: those stay, the others move, and the cost is the plain sum over ALL corners
    p = synth.make_problem(4, 12, 6)
    c = np.zeros(p.n_boards, dtype=np.uint8)
    c[::2] = 1
    p.board_pose_constant = c
    q = p.copy().normalised()
    s = orc.solve(q)
    assert s["termination_type"] == 0
    assert np.array_equal(q.board_rt[::2], p.board_rt[::2]) and np.abs(q.board_rt[1::2] - p.board_rt[1::2]).max() > 0
    cost, _ = orc.evaluate(q, jets=False)
    assert abs(cost - s["final_cost"]) <= 1e-12 * cost
    # an all-zero flag array is the unconstrained problem
    a, b = p.copy().normalised(), p.copy().normalised()
    a.board_pose_constant = None
    b.board_pose_constant = np.zeros(p.n_boards, dtype=np.uint8)
    sa, sb = orc.solve(a), orc.solve(b)
    assert sa["iterations"] == sb["iterations"] and np.array_equal(a.intr, b.intr)
