"""CPU tests of the rig-initialisation oracle (oracle/tscm_oracle_rig.c restating
multi_calib.cpp:6-153) against independent numpy restatements and ground truth."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import synth

from tests.helpers import mixed_visibility_rig, np_Rt_to_R_t, np_rig_stage, rig_with_unseen_boards


def test_rodrigues_inverse_round_trip():
    rng = np.random.default_rng(3)
    for _ in range(200):
        aa = rng.normal(size=3)
        aa *= rng.uniform(1e-3, np.pi - 1e-3) / np.linalg.norm(aa)
        got = orc.rodrigues_inverse(orc.rodrigues(aa))
        assert np.max(np.abs(got - aa)) < 1e-9 * max(1.0, np.linalg.norm(aa))
    assert np.all(orc.rodrigues_inverse(np.eye(3)) == 0.0)
    # tiny angles go through the generic branch down to s = 1e-5, below that rvec = 0 (cv::Rodrigues)
    got = orc.rodrigues_inverse(orc.rodrigues(np.array([3e-5, 0.0, 0.0])))
    assert abs(got[0] - 3e-5) < 1e-10             # theta = acos(c): ill-conditioned near 0, as in OpenCV
    assert np.all(orc.rodrigues_inverse(orc.rodrigues(np.array([1e-6, 0.0, 0.0]))) == 0.0)


@pytest.mark.parametrize("axis", [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, -2, 3], [-1, 2, 0.5]])
def test_rodrigues_inverse_at_pi(axis):
    k = np.array(axis, dtype=np.float64)
    k /= np.linalg.norm(k)
    R = 2.0 * np.outer(k, k) - np.eye(3)
    r = orc.rodrigues_inverse(R)
    assert abs(np.linalg.norm(r) - np.pi) < 1e-7          # acos(c) near c = -1 is only sqrt(eps) accurate
    # the axis sign is a convention at exactly pi; the rotation must be the same
    assert np.max(np.abs(orc.rodrigues(r) - R)) < 1e-7


def test_rodrigues_inverse_orthonormalises_first():
    """cv::Rodrigues replaces R by U V^T of its SVD before extracting the axis."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        aa = rng.normal(size=3)
        R = orc.rodrigues(aa) + 1e-4 * rng.normal(size=(3, 3))
        U, _, Vt = np.linalg.svd(R)
        Q = U @ Vt
        want = synth.rotmat_to_aa(Q)
        got = orc.rodrigues_inverse(R)
        assert np.max(np.abs(got - want)) < 1e-9


def test_Rt_to_R_t_keeps_float32_columns():
    rng = np.random.default_rng(11)
    Rt = rng.normal(size=(3, 3))
    R, t = orc.Rt_to_R_t(Rt)
    Rn, tn = np_Rt_to_R_t(Rt)
    assert np.array_equal(R, Rn)                 # bit-exact: float32 casts and a float32 cross product
    assert np.array_equal(t, Rt[:, 2]) and np.array_equal(t, tn)
    assert np.array_equal(R[:, 0], Rt[:, 0].astype(np.float32).astype(np.float64))


@pytest.mark.parametrize("C,V,seed", [(4, 12, 7), (2, 10, 3), (8, 6, 21)])
def test_rig_init_matches_numpy_restatement(C, V, seed):
    p = synth.make_problem(C, V, seed)
    inp = synth.make_rig_input(p)
    o = orc.rig_init(inp)
    assert o["rc"] == 0
    Rp, tp = np.eye(3), np.zeros(3)
    for i in range(1, C):
        common, Rs, ts, err = np_rig_stage(inp, i, Rp, tp)
        j = int(np.argmin(err))
        assert o["cam_choice"][i] == j
        assert abs(o["cam_min_error"][i] - err[j]) < 1e-10 * err[j]
        assert np.max(np.abs(o["cam_R"][i] - Rs[j])) < 1e-13
        assert np.max(np.abs(o["cam_t"][i] - ts[j])) < 1e-10
        e2 = orc.rig_hypothesis_errors(inp, i, Rp, tp, Rs, ts)
        assert np.max(np.abs(e2 - err) / err) < 1e-10
        Rp, tp = Rs[j], ts[j]
    assert np.max(np.abs(o["cam_rt"][:, 3:] - o["cam_t"])) == 0.0
    for i in range(C):
        assert np.max(np.abs(orc.rodrigues(o["cam_rt"][i, :3]) - o["cam_R"][i])) < 1e-6   # float32 columns


def test_rig_init_exact_inputs_recover_ground_truth():
    p = synth.make_problem(4, 10, 5, noise_px=0.0, perturb=False)
    inp = synth.make_rig_input(p, rot_sigma=0.0, t_sigma=0.0)
    o = orc.rig_init(inp)
    assert o["rc"] == 0 and o["board_initial"].all()
    gt_cam, gt_board = p.meta["gt_cam_rt"], p.meta["gt_board_rt"]
    # float32 rotation columns limit the agreement to ~1e-7 rad, 1e-4 mm
    assert np.max(np.abs(o["cam_rt"][:, :3] - gt_cam[:, :3])) < 1e-6
    assert np.max(np.abs(o["cam_rt"][:, 3:] - gt_cam[:, 3:])) < 1e-3
    assert np.max(np.abs(o["board_rt"][:, :3] - gt_board[:, :3])) < 1e-6
    assert np.max(np.abs(o["board_rt"][:, 3:] - gt_board[:, 3:])) < 1e-3
    per_projection = o["cam_min_error"][1:] / (2 * 54 * 5)
    assert np.all(per_projection < 1e-3)         # pixels


def test_rig_init_board_poses_minimise_their_own_error():
    p = rig_with_unseen_boards(mixed_visibility_rig(seed=9), extra=3)
    inp = synth.make_rig_input(p)
    o = orc.rig_init(inp)
    assert o["rc"] == 0
    from tests.helpers import np_project_skew
    seen = inp.has.sum(axis=0)
    assert np.array_equal(o["board_initial"], (seen > 0).astype(np.uint8))
    assert np.all(o["board_rt"][seen == 0] == 0.0) and np.all(o["board_R"][seen == 0] == 0.0)
    for b in np.nonzero(seen > 0)[0]:
        cams = np.nonzero(inp.has[:, b])[0]
        R, t = np_Rt_to_R_t(inp.Rt[cams, b])
        Rs = np.swapaxes(o["cam_R"][cams], 1, 2) @ R
        ts = np.einsum("kji,kj->ki", o["cam_R"][cams], t - o["cam_t"][cams])
        errs = []
        for q in range(cams.size):
            e = 0.0
            for m in cams:
                P = inp.worlds @ (o["cam_R"][m] @ Rs[q]).T + o["cam_R"][m] @ ts[q] + o["cam_t"][m]
                u, v = np_project_skew(inp.intr[m], P)
                e += np.sqrt((inp.pix_u[m, b] - u) ** 2 + (inp.pix_v[m, b] - v) ** 2).sum()
            errs.append(e)
        q = 0 if cams.size == 1 else int(np.argmin(errs))
        assert np.max(np.abs(o["board_R"][b] - Rs[q])) < 1e-12
        assert np.max(np.abs(o["board_t"][b] - ts[q])) < 1e-9


def test_rig_init_reports_reference_undefined_behaviour():
    """Cameras 1 and 2 share no board: the reference indexes Rs[-1] (multi_calib.cpp:86)."""
    p = synth.make_problem(4, 8, 2)
    inp = synth.make_rig_input(p)
    drop = inp.has[1].astype(bool) & inp.has[2].astype(bool)
    inp.has[2, drop] = 0
    assert orc.rig_init(inp)["rc"] == -1
