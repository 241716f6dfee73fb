"""Deterministic synthetic chessboard observations for the TSCM LM solver.

Image 1280x1080 (EpipolarRectify/test_img.jpg is a 2x2 mosaic of such frames),
board 9x6 corners at 45 mm pitch in the point order of main.cpp:12-18, ground-truth
intrinsics / rig extrinsics are the example result EpipolarRectify/calib.yaml:3-69.
Counter-based RNG (splitmix64 -> Box-Muller in fp64) so the same (seed, shape) always
gives bit-identical data on any box.  Seed convention: 20240 + config index.
"""
from __future__ import annotations

import numpy as np

from .problem import Problem

IMG_W, IMG_H = 1280.0, 1080.0

# EpipolarRectify/calib.yaml:3-10, 16-23, 34-41, 52-59  (cam0..cam3: fx fy cx cy xi lambda alpha b c)
CALIB_INTR = np.array([
    [4.3129641731951233e+02, 4.3077528857601646e+02, 6.4653015901902177e+02, 5.2120451427825685e+02,
     -2.7125775332873053e-01, -8.7861849854000834e-02, 5.6023435889162265e-01, 0., 0.],
    [4.3366730337860304e+02, 4.3377366718652252e+02, 6.5043289767844408e+02, 5.3217610796339648e+02,
     -2.5567341708788405e-01, -8.0998645840408265e-02, 5.6043293184809229e-01, 0., 0.],
    [4.4342294254852777e+02, 4.4269548663571004e+02, 6.5012232252239130e+02, 5.1864631548858017e+02,
     -2.3275919129762454e-01, -8.7007852953879805e-02, 5.6302432477866149e-01, 0., 0.],
    [4.3725205336966712e+02, 4.3738251105641092e+02, 6.4148306394889755e+02, 5.5309342913742341e+02,
     -2.6287894613485679e-01, -8.5693153628330507e-02, 5.6177801764159951e-01, 0., 0.],
])

# EpipolarRectify/calib.yaml:11-15, 24-33, 42-51, 60-69  (Twc0..Twc3 = [R|t] of cameras_[i].R(), t())
CALIB_TWC = np.array([
    [[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.]],
    [[5.0160892202284401e-03, -1.2446352011191332e-02, 9.9990995953163087e-01, 3.1111069091426958e+02],
     [-4.9652802236104215e-02, 9.9868604260337857e-01, 1.2680202652341772e-02, -3.2581972269830493e+00],
     [-9.9875394271013351e-01, -4.9711936502369769e-02, 4.3915020377227887e-03, -3.0250006677005149e+02]],
    [[-9.9912757728632307e-01, -4.1543088687854141e-02, 4.2727143873527804e-03, -4.5684542332524316e+00],
     [-4.1723107975334954e-02, 9.9738123070453755e-01, -5.9075061567300073e-02, -3.5570993658832819e+01],
     [-1.8073646121761103e-03, -5.9201794065508177e-02, -9.9824439943962817e-01, -6.1759896466830685e+02]],
    [[-1.0309658021738319e-02, -5.9375126415255344e-02, -9.9818250100602735e-01, -3.0116931471069142e+02],
     [5.1486531371731037e-02, 9.9687992136190828e-01, -5.9829419793145176e-02, -2.9127716209064634e+01],
     [9.9862047247129027e-01, -5.2009775510466122e-02, -7.2204717690991169e-03, -3.0393000777920400e+02]],
])


# ----------------------------------------------------------------------------- RNG
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser applied to a uint64 counter array."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


class CounterRNG:
    """Stateless stream: value k of stream s is a pure function of (seed, s, k)."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed)

    def _bits(self, stream: int, idx: np.ndarray) -> np.ndarray:
        with np.errstate(over="ignore"):
            base = splitmix64(np.array([self.seed ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))], dtype=np.uint64))[0]
            return splitmix64(base + idx.astype(np.uint64) * np.uint64(0x2545F4914F6CDD1D))

    def uniform(self, stream: int, idx: np.ndarray) -> np.ndarray:
        """U(0,1) with 53 random bits, never exactly 0."""
        return ((self._bits(stream, idx) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)

    def normal(self, stream: int, idx: np.ndarray) -> np.ndarray:
        u1 = self.uniform(2 * stream, idx)
        u2 = self.uniform(2 * stream + 1, idx)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ----------------------------------------------------------------------------- geometry
def board_points(cols: int = 9, rows: int = 6, size: float = 45.0) -> np.ndarray:
    """main.cpp:12-18: for u in rows: for v in cols: (v*size, u*size, 0)."""
    v, u = np.meshgrid(np.arange(cols), np.arange(rows))
    return np.stack([v.ravel() * size, u.ravel() * size], axis=1).astype(np.float64)


def rodrigues(aa: np.ndarray) -> np.ndarray:
    """angle-axis [...,3] -> rotation matrices [...,3,3]."""
    aa = np.asarray(aa, dtype=np.float64)
    th = np.linalg.norm(aa, axis=-1)[..., None, None]
    k = aa / np.maximum(th[..., 0], 1e-300)
    K = np.zeros(aa.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    I = np.broadcast_to(np.eye(3), K.shape)
    return I + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def rotmat_to_aa(R: np.ndarray) -> np.ndarray:
    """rotation matrices [...,3,3] -> angle-axis [...,3] via quaternions (robust near pi)."""
    R = np.asarray(R, dtype=np.float64)
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    q = np.empty(R.shape[:-2] + (4,))
    q[..., 0] = np.sqrt(np.maximum(0.0, 1.0 + m00 + m11 + m22)) / 2.0
    q[..., 1] = np.sqrt(np.maximum(0.0, 1.0 + m00 - m11 - m22)) / 2.0
    q[..., 2] = np.sqrt(np.maximum(0.0, 1.0 - m00 + m11 - m22)) / 2.0
    q[..., 3] = np.sqrt(np.maximum(0.0, 1.0 - m00 - m11 + m22)) / 2.0
    # pick the largest component as pivot for sign recovery
    piv = np.argmax(q, axis=-1)
    out = np.empty_like(q)
    r = lambda i, j: R[..., i, j]
    cand = [
        np.stack([q[..., 0], (r(2, 1) - r(1, 2)) / (4 * np.maximum(q[..., 0], 1e-300)),
                  (r(0, 2) - r(2, 0)) / (4 * np.maximum(q[..., 0], 1e-300)),
                  (r(1, 0) - r(0, 1)) / (4 * np.maximum(q[..., 0], 1e-300))], -1),
        np.stack([(r(2, 1) - r(1, 2)) / (4 * np.maximum(q[..., 1], 1e-300)), q[..., 1],
                  (r(0, 1) + r(1, 0)) / (4 * np.maximum(q[..., 1], 1e-300)),
                  (r(0, 2) + r(2, 0)) / (4 * np.maximum(q[..., 1], 1e-300))], -1),
        np.stack([(r(0, 2) - r(2, 0)) / (4 * np.maximum(q[..., 2], 1e-300)),
                  (r(0, 1) + r(1, 0)) / (4 * np.maximum(q[..., 2], 1e-300)), q[..., 2],
                  (r(1, 2) + r(2, 1)) / (4 * np.maximum(q[..., 2], 1e-300))], -1),
        np.stack([(r(1, 0) - r(0, 1)) / (4 * np.maximum(q[..., 3], 1e-300)),
                  (r(0, 2) + r(2, 0)) / (4 * np.maximum(q[..., 3], 1e-300)),
                  (r(1, 2) + r(2, 1)) / (4 * np.maximum(q[..., 3], 1e-300)), q[..., 3]], -1),
    ]
    for i in range(4):
        m = piv == i
        out[m] = cand[i][m]
    out = out / np.linalg.norm(out, axis=-1, keepdims=True)
    out = np.where(out[..., :1] < 0, -out, out)
    sin_half = np.linalg.norm(out[..., 1:], axis=-1)
    angle = 2.0 * np.arctan2(sin_half, out[..., 0])
    scale = np.where(sin_half > 1e-12, angle / np.maximum(sin_half, 1e-300), 2.0)
    return out[..., 1:] * scale[..., None]


def ts_project(intr: np.ndarray, P: np.ndarray):
    """Functor-form projection (no skew): TS.h:117-125.  intr [...,9], P [...,3] -> u, v, ksai."""
    X, Y, Z = P[..., 0], P[..., 1], P[..., 2]
    fx, fy, cx, cy, xi, lam, al = (intr[..., i] for i in range(7))
    d1 = np.sqrt(X * X + Y * Y + Z * Z)
    d2 = np.sqrt(X * X + Y * Y + (Z + xi * d1) ** 2)
    d3 = np.sqrt(X * X + Y * Y + (Z + xi * d1 + lam * d2) ** 2)
    ksai = Z + xi * d1 + lam * d2 + al / (1 - al) * d3
    return fx * X / ksai + cx, fy * Y / ksai + cy, ksai


def rig(n_cameras: int):
    """Ground-truth intrinsics [C,9] and camera poses [C,6]."""
    if n_cameras == 1:
        return CALIB_INTR[:1].copy(), np.zeros((1, 6))
    if n_cameras == 4:
        R, t = CALIB_TWC[:, :, :3], CALIB_TWC[:, :, 3]
    else:
        # reuse the four intrinsic sets, cameras every 360/C degrees on a ~430 mm ring
        ang = -2.0 * np.pi * np.arange(n_cameras) / n_cameras
        R = rodrigues(np.stack([np.zeros_like(ang), ang, np.zeros_like(ang)], 1))
        radius = 430.0
        centre = np.array([0.0, 0.0, -radius])
        axes = np.einsum("cji,j->ci", R, np.array([0.0, 0.0, 1.0]))   # R^T e_z
        cpos = centre + radius * axes
        t = -np.einsum("cij,cj->ci", R, cpos)
    aa = rotmat_to_aa(R)
    aa[0] = 0.0
    tt = t.copy()
    tt[0] = 0.0
    intr = CALIB_INTR[np.arange(n_cameras) % 4].copy()
    return intr, np.concatenate([aa, tt], axis=1)


def _sample_board_poses(rng: CounterRNG, frame_ids: np.ndarray, attempt: int, cams_a, cams_b,
                        cam_R, cam_t, bxy):
    """One attempt at a board pose for every frame in frame_ids; returns (aa, t)."""
    n = frame_ids.shape[0]
    ctr = frame_ids.astype(np.uint64) * np.uint64(64) + np.uint64(attempt)
    nrm = lambda s: rng.normal(s, ctr)
    uni = lambda s: rng.uniform(s, ctr)
    Ra, Rb = cam_R[cams_a], cam_R[cams_b]
    ca = -np.einsum("nji,nj->ni", Ra, cam_t[cams_a])
    cb = -np.einsum("nji,nj->ni", Rb, cam_t[cams_b])
    ax = Ra[:, 2, :] + Rb[:, 2, :]                  # R^T e_z = third row of R
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    # orthonormal frame around the bisector
    up = np.tile(np.array([0.0, 1.0, 0.0]), (n, 1))
    xb = np.cross(up, ax)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    yb = np.cross(ax, xb)
    dist = 300.0 + 600.0 * uni(10)
    lat = 0.35 * dist
    pos = 0.5 * (ca + cb) + ax * dist[:, None] + xb * (lat * nrm(11))[:, None] + yb * (0.6 * lat * nrm(12))[:, None]
    Rface = np.stack([xb, yb, ax], axis=2)          # columns = board x, y, normal
    pert = rodrigues(0.4 * np.stack([nrm(13), nrm(14), nrm(15)], axis=1))
    Rboard = pert @ Rface
    centre = np.array([0.5 * (bxy[:, 0].max() + bxy[:, 0].min()), 0.5 * (bxy[:, 1].max() + bxy[:, 1].min()), 0.0])
    tboard = pos - Rboard @ centre
    return Rboard, tboard


def _visible(intr, R, t, Rboard, tboard, bxy, margin=2.0):
    """All board corners project inside the image with a valid (positive) ksai."""
    P3 = np.concatenate([bxy, np.zeros((bxy.shape[0], 1))], axis=1)      # [n,3]
    Pw = np.einsum("fij,nj->fni", Rboard, P3) + tboard[:, None, :]
    Pc = np.einsum("fij,fnj->fni", R, Pw) + t[:, None, :]
    u, v, ks = ts_project(intr[:, None, :], Pc)
    ok = (ks > 1e-3) & (u > margin) & (u < IMG_W - margin) & (v > margin) & (v < IMG_H - margin)
    return ok.all(axis=1), u, v


def make_problem(n_cameras: int, views_per_cam: int, seed: int, *, noise_px: float = 0.1,
                 perturb: bool = True, dense: bool = False, cols: int = 9, rows: int = 6,
                 pitch: float = 45.0) -> Problem:
    """Synthetic problem of SURVEY 8d.

    n_cameras == 1 -> mono problem (TS.cpp:247-282) with `views_per_cam` images.
    n_cameras  > 1 -> rig; B = C*V/2 frames, frame f seen by cameras f%C and (f+1)%C
    (adjacency requirement of multi_calib.cpp:36), or with dense=True B = V frames seen
    by cameras pairs cycling so that each frame is still seen by two adjacent cameras.
    The returned Problem holds the *initial guess* in cam_rt/intr/board_rt and the ground
    truth in meta["gt_*"].
    """
    rng = CounterRNG(seed)
    bxy = board_points(cols, rows, pitch)
    npts = bxy.shape[0]
    intr_gt, cam_gt = rig(n_cameras)
    C = n_cameras
    mono = C == 1
    cam_R = rodrigues(cam_gt[:, :3])
    cam_t = cam_gt[:, 3:]
    if mono:
        B = views_per_cam
        cams_a = np.zeros(B, dtype=np.int64)
        cams_b = cams_a
    else:
        B = (C * views_per_cam) // 2
        f = np.arange(B)
        cams_a, cams_b = f % C, (f + 1) % C
    frame_ids = np.arange(B)
    Rb = np.zeros((B, 3, 3))
    tb = np.zeros((B, 3))
    ua = np.zeros((B, npts)); va = np.zeros((B, npts)); ub = np.zeros((B, npts)); vb = np.zeros((B, npts))
    todo = frame_ids.copy()
    for attempt in range(64):
        if todo.size == 0:
            break
        R_try, t_try = _sample_board_poses(rng, todo, attempt, cams_a[todo], cams_b[todo], cam_R, cam_t, bxy)
        if mono:
            # mono: spread the board over the field of view (direction cone around the axis)
            pass
        ok_a, u_a, v_a = _visible(intr_gt[cams_a[todo]], cam_R[cams_a[todo]], cam_t[cams_a[todo]], R_try, t_try, bxy)
        ok_b, u_b, v_b = _visible(intr_gt[cams_b[todo]], cam_R[cams_b[todo]], cam_t[cams_b[todo]], R_try, t_try, bxy)
        ok = ok_a & ok_b
        sel = todo[ok]
        Rb[sel], tb[sel] = R_try[ok], t_try[ok]
        ua[sel], va[sel], ub[sel], vb[sel] = u_a[ok], v_a[ok], u_b[ok], v_b[ok]
        todo = todo[~ok]
    if todo.size:
        raise RuntimeError(f"could not place {todo.size} boards inside the images")
    board_gt = np.concatenate([rotmat_to_aa(Rb), tb], axis=1)

    # views: frame-major, camera a then camera b (mono: one view per frame)
    if mono:
        view_board = frame_ids.astype(np.int32)
        view_camera = np.zeros(B, dtype=np.int32)
        u = ua
        v = va
    else:
        view_board = np.repeat(frame_ids, 2).astype(np.int32)
        view_camera = np.stack([cams_a, cams_b], axis=1).ravel().astype(np.int32)
        u = np.stack([ua, ub], axis=1).reshape(2 * B, npts)
        v = np.stack([va, vb], axis=1).reshape(2 * B, npts)
    V = view_board.shape[0]
    view_count = np.full(V, npts, dtype=np.int32)
    view_offset = (np.arange(V) * npts).astype(np.int32)
    obs_u = u.ravel().copy()
    obs_v = v.ravel().copy()
    if noise_px > 0:
        k = np.arange(obs_u.shape[0])
        obs_u += noise_px * rng.normal(20, k)
        obs_v += noise_px * rng.normal(21, k)

    intr0, cam0, board0 = intr_gt.copy(), cam_gt.copy(), board_gt.copy()
    if perturb:
        ci = np.arange(C * 7)
        intr0[:, :7] *= 1.0 + 0.02 * rng.normal(30, ci).reshape(C, 7)
        cb = np.arange(B * 6)
        board0[:, :3] += 0.01 * rng.normal(31, cb).reshape(B, 6)[:, :3]
        sgn = np.where(rng.uniform(32, cb).reshape(B, 6)[:, 3:] < 0.5, -1.0, 1.0)
        board0[:, 3:] += sgn * (3.0 + 2.0 * rng.uniform(33, cb).reshape(B, 6)[:, 3:])
        if C > 1:
            cc = np.arange(C * 6)
            cam0[1:, :3] += 0.01 * rng.normal(34, cc).reshape(C, 6)[1:, :3]
            sg = np.where(rng.uniform(35, cc).reshape(C, 6)[1:, 3:] < 0.5, -1.0, 1.0)
            cam0[1:, 3:] += sg * (3.0 + 2.0 * rng.uniform(36, cc).reshape(C, 6)[1:, 3:])
    const = np.zeros(C, dtype=np.uint8)
    const[0] = 1   # multi_calib.cpp:186 (camera 0); mono: there is no camera-pose block at all
    p = Problem(C, B, bxy, view_camera, view_board, view_offset, view_count, obs_u, obs_v,
                cam0, intr0, board0, const, mono,
                meta=dict(seed=seed, gt_intr=intr_gt, gt_cam_rt=cam_gt, gt_board_rt=board_gt,
                          noise_px=noise_px, views_per_cam=views_per_cam))
    return p.normalised()


# BASELINE.json configs (index -> (cameras, views per camera)); seed = 20240 + index
CONFIGS = {1: (1, 20), 2: (1, 2000), 3: (4, 500), 4: (4, 10000), 5: (8, 20000)}


def make_config(index: int, *, poses_fixed: bool = False, **kw) -> Problem:
    """BASELINE.json config `index`.  poses_fixed: the "intrinsics-only" form SURVEY 8d asks for next to config 2 --
    every board / view pose block is held constant (tscm_problem.board_pose_constant) at its ground-truth value, as if
    the poses were known from an earlier calibration, and only the intrinsics (and the free camera poses) are solved."""
    C, V = CONFIGS[index]
    p = make_problem(C, V, 20240 + index, **kw)
    p.meta["config"] = index
    if poses_fixed:
        p.board_rt = p.meta["gt_board_rt"].copy()
        p.board_pose_constant = np.ones(p.n_boards, dtype=np.uint8)
        p.meta["poses_fixed"] = True
    return p


def make_rig_input(p: Problem, *, rot_sigma: float = 0.01, t_sigma: float = 3.0, seed: int | None = None):
    """What MultiCalib::MultiCalib reads from the mono-calibrated cameras (multi_calib.cpp:6-153)
    for the synthetic rig `p`: per-camera board poses Rt(j) = [r1 r2 t] of the ground-truth
    view pose perturbed by `rot_sigma` rad / `t_sigma` mm (a mono calibration is only that good),
    the (perturbed) intrinsics p.intr and the observed pixels."""
    from .rig import RigInput
    if p.mono:
        raise ValueError("rig initialisation needs a multi-camera problem")
    rng = CounterRNG((p.meta.get("seed", 0) if seed is None else seed) + 7919)
    C, B, n = p.n_cameras, p.n_boards, p.n_points
    gt_cam, gt_board = p.meta["gt_cam_rt"], p.meta["gt_board_rt"]
    cam_R, cam_t = rodrigues(gt_cam[:, :3]), gt_cam[:, 3:]
    vc, vb = p.view_camera.astype(np.int64), p.view_board.astype(np.int64)
    V = vc.shape[0]
    Rb, tb = rodrigues(gt_board[vb, :3]), gt_board[vb, 3:]
    R = cam_R[vc] @ Rb
    t = np.einsum("vij,vj->vi", cam_R[vc], tb) + cam_t[vc]
    k = np.arange(V)
    pert = rodrigues(rot_sigma * np.stack([rng.normal(40, k), rng.normal(41, k), rng.normal(42, k)], axis=1))
    R = pert @ R
    t = t + t_sigma * np.stack([rng.normal(43, k), rng.normal(44, k), rng.normal(45, k)], axis=1)
    has = np.zeros((C, B), dtype=np.uint8)
    Rt = np.zeros((C, B, 3, 3))
    pu = np.zeros((C, B, n))
    pv = np.zeros((C, B, n))
    full = p.view_count == n
    if not full.all():
        raise ValueError("rig initialisation needs complete boards (cv::findChessboardCorners is all-or-nothing)")
    has[vc, vb] = 1
    Rt[vc, vb, :, 0] = R[:, :, 0]
    Rt[vc, vb, :, 1] = R[:, :, 1]
    Rt[vc, vb, :, 2] = t
    idx = p.view_offset.astype(np.int64)[:, None] + np.arange(n)[None, :]
    pu[vc, vb] = p.obs_u[idx]
    pv[vc, vb] = p.obs_v[idx]
    worlds = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
    return RigInput(worlds, p.intr.copy(), has, Rt, pu, pv,
                    meta=dict(p.meta, rot_sigma=rot_sigma, t_sigma=t_sigma)).normalised()


def unproject_pixels_np(intr, u, v):
    """Vectorised inverse Triple Sphere model (TS.cpp get_unit_sphere_coordinate): pixel arrays -> unit rays."""
    fx, fy, cx, cy, xi, lam, alpha, b, c = [float(t) for t in np.asarray(intr, dtype=np.float64).ravel()]
    x, y = u - cx, v - cy
    den = fx * fy - b * c
    mx, my = (fy * x - b * y) / den, (-c * x + fx * y) / den
    ksai = alpha / (1 - alpha)
    r2 = mx * mx + my * my
    gamma = (ksai + np.sqrt(np.maximum(1 + (1 - ksai * ksai) * r2, 0.0))) / (r2 + 1)      # (outside the image circle: clamped)
    g = gamma - ksai
    yita = lam * g + np.sqrt((g * g - 1) * lam * lam + 1)
    mz = yita * g
    mu = xi * (mz - lam) + np.sqrt(xi * xi * ((mz - lam) ** 2 - 1) + 1)
    return np.stack([mu * yita * gamma * mx, mu * yita * gamma * my, mu * (mz - lam) - xi], axis=-1)


def render_chessboard(intr, board_rt, cols: int, rows: int, pitch: float, width: int, height: int,
                      supersample: int = 3, background: int = 110, dark: int = 25, bright: int = 230) -> np.ndarray:
    """Synthetic grey image (uint8, height x width) of a chessboard with cols x rows INNER corners at
    (i * pitch, j * pitch, 0), i.e. (cols + 1) x (rows + 1) squares, seen through the Triple Sphere model with the
    board pose board_rt (angle-axis + translation, board -> camera).  Every pixel is the mean of supersample^2 rays
    intersected with the board plane.  Test / demo data for the corner detector."""
    if isinstance(board_rt, tuple):                    # (R, t) given directly
        R, t = np.asarray(board_rt[0], dtype=np.float64), np.asarray(board_rt[1], dtype=np.float64)
    else:
        R = rodrigues(np.asarray(board_rt[:3], dtype=np.float64))
        t = np.asarray(board_rt[3:], dtype=np.float64)
    n = R[:, 2]                                        # board normal in camera coordinates
    ss = supersample
    offs = (np.arange(ss) + 0.5) / ss - 0.5
    acc = np.zeros((height, width), dtype=np.float64)
    jj, ii = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    for oy in offs:
        for ox in offs:
            d = unproject_pixels_np(intr, jj + ox, ii + oy)
            den = d @ n
            s = (t @ n) / np.where(np.abs(den) < 1e-12, 1e-12, den)
            P = d * s[..., None] - t                   # point on the plane, relative to the board origin (camera axes)
            X, Y = P @ R[:, 0], P @ R[:, 1]
            qx, qy = np.floor(X / pitch + 1.0), np.floor(Y / pitch + 1.0)      # square index, -1 outside on the low side
            inside = (s > 0) & (qx >= 0) & (qx <= cols) & (qy >= 0) & (qy <= rows)
            col = np.where(((qx + qy) % 2) == 0, dark, bright)
            acc += np.where(inside, col, background)
    return np.clip(np.rint(acc / (ss * ss)), 0, 255).astype(np.uint8)
