"""CPU tests of the mono-initialisation pieces: oracle estimate_focal (TS.cpp:110-168) against
LAPACK's SVD, the model's own geometry, and tscm_poses_from_r1r2t (host-only, TS.cpp:62-74)."""
import numpy as np

from oracle import pyoracle as orc
from tscm_calib_amd import rig, synth


def np_focal(pu, pv, count, w, h, cx, cy):
    f, n = 0.0, 0
    for k in range(pu.shape[0]):
        if count[k] == 0:
            continue
        for i in range(h):
            x, y = pu[k, i * w:(i + 1) * w] - cx, pv[k, i * w:(i + 1) * w] - cy
            P = np.stack([x, y, 0.5 * np.ones(w), -0.5 * (x * x + y * y)], axis=1)
            c = np.linalg.svd(P)[2][-1]
            t = c[0] ** 2 + c[1] ** 2 + c[2] * c[3]
            if t < 0:
                continue
            d = np.sqrt(1 / t)
            nx, ny = c[0] * d, c[1] * d
            if nx * nx + ny * ny > 0.95:
                continue
            f += abs(c[2] * d / np.sqrt(1 - nx * nx - ny * ny))
            n += 1
    return (f / n if n else 0.0), n


def test_estimate_focal_matches_lapack_svd():
    p = synth.make_problem(1, 30, 20241)
    V = p.n_views
    pu, pv = p.obs_u.reshape(V, 54), p.obs_v.reshape(V, 54)
    count = np.full(V, 54, dtype=np.int32)
    count[[3, 17]] = 0                                     # images without a board are skipped
    f, n, rc = orc.estimate_focal(pu, pv, count, 9, 6, 639.5, 539.5)
    fn, nn = np_focal(pu, pv, count, 9, 6, 639.5, 539.5)
    assert rc == 0 and n == nn and 0 < n <= 28 * 6
    assert abs(f - fn) < 1e-9 * fn
    assert 300.0 < f < 650.0                               # a starting value for fx ~ 431, not an estimate of it


def test_estimate_focal_is_exact_for_the_model_it_assumes():
    """xi = lambda = 0, alpha = 0.5 (the initial values of TS.cpp:45-47): lines map to circles and
    every accepted row returns the focal length itself."""
    intr = np.array([400.0, 400.0, 639.5, 539.5, 0.0, 0.0, 0.5, 0.0, 0.0])
    p = synth.make_problem(1, 12, 5, noise_px=0.0, perturb=False)
    V = p.n_views
    bt = p.meta["gt_board_rt"]
    P3 = np.concatenate([p.board_xy, np.zeros((54, 1))], axis=1)
    pu, pv = np.zeros((V, 54)), np.zeros((V, 54))
    for k in range(V):
        Pc = P3 @ synth.rodrigues(bt[k, :3]).T + bt[k, 3:]
        pu[k], pv[k], _ = synth.ts_project(intr, Pc)
    f, n, rc = orc.estimate_focal(pu, pv, np.full(V, 54, dtype=np.int32), 9, 6, 639.5, 539.5)
    assert rc == 0 and n > 0
    assert abs(f - 400.0) < 1e-6
    assert orc.estimate_focal(pu, pv, np.zeros(V, dtype=np.int32), 9, 6, 639.5, 539.5)[:2] == (0.0, 0)   # "focal estimation failed"
    assert orc.estimate_focal(pu[:, :18], pv[:, :18], np.full(V, 18, dtype=np.int32), 3, 6, 639.5, 539.5)[2] == -1


def test_poses_from_Rt_matches_oracle():
    rng = np.random.default_rng(8)
    aa = rng.normal(size=(40, 3))
    R = synth.rodrigues(aa)
    t = rng.normal(size=(40, 3)) * 300.0
    Rt = np.stack([R[:, :, 0], R[:, :, 1], t], axis=2)
    has = np.ones(40, dtype=np.uint8)
    has[5] = 0
    got = rig.poses_from_Rt(Rt, has)
    for i in range(40):
        if not has[i]:
            assert np.all(got[i] == 0.0)
            continue
        want = orc.Rt_to_rt(Rt[i])
        assert np.max(np.abs(got[i] - want)) < 1e-9
        assert np.array_equal(got[i, 3:], t[i])
        assert np.max(np.abs(got[i, :3] - synth.rotmat_to_aa(R[i]))) < 1e-6      # float32 columns


def _mono_views(seed, V=12, noise=0.0):
    p = synth.make_problem(1, V, seed, noise_px=noise, perturb=False)
    n = p.n_points
    W = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
    return p, p.obs_u.reshape(V, n), p.obs_v.reshape(V, n), W


def test_estimate_extrinsic_recovers_exact_poses():
    """Exact intrinsics + exact corners: the planar PnP returns the generating poses."""
    p, pu, pv, W = _mono_views(5)
    count = np.full(p.n_views, 54, dtype=np.int32)
    count[4] = 0
    Rt, k = orc.estimate_extrinsic(p.meta["gt_intr"][0], pu, pv, count, W, 9)
    assert k == p.n_views - 1 and np.all(Rt[4] == 0.0)
    gt = p.meta["gt_board_rt"]
    R = synth.rodrigues(gt[:, :3])
    ok = count > 0
    assert np.max(np.abs(Rt[ok][:, :, 0] - R[ok][:, :, 0])) < 1e-10 and np.max(np.abs(Rt[ok][:, :, 1] - R[ok][:, :, 1])) < 1e-10
    assert np.max(np.abs(Rt[ok][:, :, 2] - gt[ok][:, 3:])) < 1e-8


def test_estimate_extrinsic_minimises_the_normalised_reprojection_error():
    """With noisy corners and rough intrinsics the result is the least-squares planar pose in the rotated
    normalised plane: perturbing it increases the error the reference's PnP minimises."""
    p, pu, pv, W = _mono_views(8, V=6, noise=0.2)
    I0 = np.array([470.0, 470.0, 639.5, 539.5, 0.0, 0.0, 0.5, 0.0, 0.0])      # TS.cpp:43-47 + a focal estimate
    Rt, k = orc.estimate_extrinsic(I0, pu, pv, np.full(6, 54, dtype=np.int32), W, 9)
    assert k == 6

    def cost(Rt_k, u, v):
        ref = 54 // 2 - 9 // 2 - 1
        q = orc.unproject(I0, np.array([u[ref], v[ref]]))
        a, b = np.arctan2(q[0], q[2]), np.arcsin(q[1])
        R1 = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]])
        R2 = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        T = R2 @ R1
        e = 0.0
        for i in range(54):
            ray = T @ orc.unproject(I0, np.array([u[i], v[i]]))
            P = T @ (Rt_k[:, 0] * W[i, 0] + Rt_k[:, 1] * W[i, 1] + Rt_k[:, 2])
            e += (P[0] / P[2] - ray[0] / ray[2]) ** 2 + (P[1] / P[2] - ray[1] / ray[2]) ** 2
        return e
    rng = np.random.default_rng(0)
    for k in range(6):
        base = cost(Rt[k], pu[k], pv[k])
        for _ in range(5):
            d = Rt[k].copy()
            d[:, 2] += rng.normal(size=3) * 0.5                  # move the translation by ~0.5 mm
            assert cost(d, pu[k], pv[k]) > base
