"""GPU parity test of tscm_estimate_focal (TS.cpp:110-168) against the CPU oracle."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import lib, rig, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("views,cols,rows,seed", [(20, 9, 6, 20241), (300, 9, 6, 4), (40, 11, 8, 9)])
def test_estimate_focal_matches_oracle(hip_device, views, cols, rows, seed):
    p = synth.make_problem(1, views, seed, cols=cols, rows=rows, pitch=30.0 if cols == 11 else 45.0)
    V, n = p.n_views, cols * rows
    pu, pv = p.obs_u.reshape(V, n), p.obs_v.reshape(V, n)
    count = np.full(V, n, dtype=np.int32)
    count[::7] = 0
    fo, no, rc = orc.estimate_focal(pu, pv, count, cols, rows, 639.5, 539.5)
    fg, ng = rig.estimate_focal(pu, pv, count, cols, rows, 639.5, 539.5, hip_device)
    assert rc == 0 and ng == no and ng > 0
    # the circle fits are ill-conditioned (singular values span 1e6): two SVD algorithms agree to ~1e-11
    assert abs(fg - fo) < 1e-9 * fo


def test_estimate_focal_edge_cases(hip_device):
    p = synth.make_problem(1, 6, 3)
    pu, pv = p.obs_u.reshape(6, 54), p.obs_v.reshape(6, 54)
    assert rig.estimate_focal(pu, pv, np.zeros(6, dtype=np.int32), 9, 6, 639.5, 539.5, hip_device) == (0.0, 0)
    assert rig.estimate_focal(np.zeros((0, 54)), np.zeros((0, 54)), np.zeros(0, dtype=np.int32), 9, 6, 639.5, 539.5, hip_device) == (0.0, 0)
    with pytest.raises(lib.TscmError) as e:
        rig.estimate_focal(pu[:, :18], pv[:, :18], np.full(6, 18, dtype=np.int32), 3, 6, 639.5, 539.5, hip_device)
    assert e.value.code == -5


@pytest.mark.parametrize("views,cols,rows,seed,noise", [(12, 9, 6, 5, 0.0), (200, 9, 6, 6, 0.2), (30, 11, 8, 7, 0.1)])
def test_estimate_extrinsic_matches_oracle(hip_device, views, cols, rows, seed, noise):
    p = synth.make_problem(1, views, seed, noise_px=noise, perturb=False, cols=cols, rows=rows, pitch=30.0 if cols == 11 else 45.0)
    V, n = p.n_views, cols * rows
    pu, pv = p.obs_u.reshape(V, n), p.obs_v.reshape(V, n)
    W = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
    count = np.full(V, n, dtype=np.int32)
    count[::5] = 0
    for intr in (p.meta["gt_intr"][0], np.array([470.0, 470.0, 639.5, 539.5, 0.0, 0.0, 0.5, 0.0, 0.0])):
        Ro, ko = orc.estimate_extrinsic(intr, pu, pv, count, W, cols)
        Rg, kg = rig.estimate_extrinsic(intr, pu, pv, count, W, cols, hip_device)
        assert kg == ko == int((count > 0).sum())
        assert np.all(Rg[count == 0] == 0.0)
        # two independent Gauss-Newton implementations (analytic vs central-difference Jacobian) converged to
        # the same minimiser: 1e-7 relative on the translation scale, 1e-8 on the rotation columns
        assert np.max(np.abs(Rg[:, :, :2] - Ro[:, :, :2])) < 1e-8
        assert np.max(np.abs(Rg[:, :, 2] - Ro[:, :, 2])) < 1e-7 * np.max(np.abs(Ro[:, :, 2]))


def test_mono_calibration_from_raw_corners(hip_device):
    """The whole TripleSphereCamera::calibrate flow (TS.cpp:30-105) from corner lists only: principal point at
    the image centre, xi = lambda = 0, alpha = 0.5 (:43-47), estimate_focal, estimate_extrinsic, [r1 r2 t] ->
    poses, refinement -- and the LM reaches the noise floor."""
    from tscm_calib_amd import api
    from tscm_calib_amd.problem import Problem
    p = synth.make_problem(1, 40, 11, noise_px=0.1)
    V, n = p.n_views, 54
    pu, pv = p.obs_u.reshape(V, n), p.obs_v.reshape(V, n)
    W = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
    count = np.full(V, n, dtype=np.int32)
    intr = np.array([0.0, 0.0, 1280 / 2 - 0.5, 1080 / 2 - 0.5, 0.0, 0.0, 0.5, 0.0, 0.0])
    focal, used = rig.estimate_focal(pu, pv, count, 9, 6, intr[2], intr[3], hip_device)
    assert used > 0
    intr[0] = intr[1] = focal
    Rt, k = rig.estimate_extrinsic(intr, pu, pv, count, W, 9, hip_device)
    assert k == V
    rt = rig.poses_from_Rt(Rt)
    q = Problem(1, V, p.board_xy, p.view_camera, p.view_board, p.view_offset, p.view_count, p.obs_u, p.obs_v,
                np.zeros((1, 6)), intr[None, :].copy(), rt, p.cam_pose_constant, True).normalised()
    ok, s = api.refinement(q, hip_device)
    assert ok and s["rmse"] < 0.2                      # sigma = 0.1 px per coordinate -> 0.141
    assert np.max(np.abs(q.intr[0, 2:4] - p.meta["gt_intr"][0, 2:4])) < 1.0          # principal point, px
