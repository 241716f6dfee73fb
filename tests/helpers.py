"""Shared helpers of the parity tests (test infrastructure; may call the oracle)."""
from __future__ import annotations

import numpy as np

from oracle import pyoracle as orc
from tscm_calib_amd import synth
from tscm_calib_amd.problem import Problem


def small_rig(n_cameras=4, views_per_cam=12, seed=7, **kw) -> Problem:
    return synth.make_problem(n_cameras, views_per_cam, seed, **kw)


def oracle_normal_equations(p: Problem) -> dict:
    """Schur-form normal equations from the oracle's dual-number Jacobians (unscaled)."""
    cost, res, Jc, Jb, Ji = orc.evaluate(p, jets=True)
    C, B, V = p.n_cameras, p.n_boards, p.n_views
    out = dict(board_gram=np.zeros((B, 6, 6)), board_grad=np.zeros((B, 6)), view_cross=np.zeros((V, 6, 15)),
               cam_gram=np.zeros((C, 15, 15)), cam_grad=np.zeros((C, 15)), cost=cost)
    k = 0
    for v in range(V):
        n = int(p.view_count[v])
        m, b = int(p.view_camera[v]), int(p.view_board[v])
        E = Jb[k:k + n].reshape(2 * n, 6)
        F = np.concatenate([Jc[k:k + n].reshape(2 * n, 6), Ji[k:k + n].reshape(2 * n, 9)], axis=1)
        r = res[k:k + n].reshape(2 * n)
        out["board_gram"][b] += E.T @ E
        out["board_grad"][b] += E.T @ r
        out["view_cross"][v] = E.T @ F
        out["cam_gram"][m] += F.T @ F
        out["cam_grad"][m] += F.T @ r
        k += n
    return out


def rel_err(a, b, floor=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def param_rel_err(p: Problem, q: Problem) -> dict:
    """Relative parameter differences in the sense of north_star (1e-6 fp64): each block is
    compared relative to the block's own magnitude (max-norm), which is how intrinsics
    (fx ~ 430, xi ~ 0.27) and poses (rad / mm) can share one threshold."""
    d = {}
    d["intr"] = float(np.max(np.abs(p.intr[:, :7] - q.intr[:, :7]) / np.maximum(np.abs(q.intr[:, :7]), 1e-3)))
    pose = lambda a, b: float(max(
        np.max(np.abs(a[:, :3] - b[:, :3])) / max(np.max(np.abs(b[:, :3])), 1e-12),
        np.max(np.abs(a[:, 3:] - b[:, 3:])) / max(np.max(np.abs(b[:, 3:])), 1e-12))) if a.size else 0.0
    d["cam_rt"] = pose(p.cam_rt, q.cam_rt) if not p.mono else 0.0
    d["board_rt"] = pose(p.board_rt, q.board_rt)
    return d


def mixed_visibility_rig(seed=5, n_frames=24, n_cameras=4, noise_px=0.05, **board) -> Problem:
    """Rig whose frames are seen by 1..C cameras (random subsets).  Observations are exact
    projections of the ground truth (image bounds ignored: the solver does not care), plus noise."""
    base = synth.make_problem(n_cameras, 2 * n_frames // n_cameras * 2, seed, noise_px=0.0, **board)
    rng = np.random.default_rng(seed)
    C, B = n_cameras, min(n_frames, base.n_boards)
    intr, cam, brd = base.meta["gt_intr"], base.meta["gt_cam_rt"], base.meta["gt_board_rt"][:B]
    npts = base.n_points
    P3 = np.concatenate([base.board_xy, np.zeros((npts, 1))], axis=1)
    vc, vb, u, v = [], [], [], []
    for b in range(B):
        k = 1 + (b % C)                                  # 1, 2, 3, 4, 1, ... cameras
        cams = np.sort(rng.choice(C, size=k, replace=False))
        Rb = synth.rodrigues(brd[b, :3])
        Pw = P3 @ Rb.T + brd[b, 3:]
        for m in cams:
            Pc = Pw @ synth.rodrigues(cam[m, :3]).T + cam[m, 3:]
            uu, vv, ks = synth.ts_project(intr[m], Pc)
            if np.any(ks <= 1e-3):
                continue
            vc.append(m); vb.append(b); u.append(uu); v.append(vv)
    V = len(vc)
    obs_u = np.concatenate(u) + noise_px * rng.normal(size=V * npts)
    obs_v = np.concatenate(v) + noise_px * rng.normal(size=V * npts)
    p = Problem(C, B, base.board_xy, np.array(vc, dtype=np.int32), np.array(vb, dtype=np.int32),
                (np.arange(V) * npts).astype(np.int32), np.full(V, npts, dtype=np.int32), obs_u, obs_v,
                base.cam_rt.copy(), base.intr.copy(), base.board_rt[:B].copy(), base.cam_pose_constant.copy(), False,
                meta=dict(gt_intr=intr, gt_cam_rt=cam, gt_board_rt=brd))
    return p.normalised()


def rig_with_pairs(n_cameras, pairs, frames_per_pair=6, seed=11, noise_px=0.05) -> Problem:
    """Rig whose camera-pair graph is exactly `pairs` (list of (a, b)): frame f is seen by the two cameras of pair
    f % len(pairs).  Observations are exact projections of the ground truth plus noise (image bounds ignored, like
    mixed_visibility_rig): chains, complete graphs, stars -- the shapes the reduced solver's elimination plan is built for."""
    C = n_cameras
    B = frames_per_pair * len(pairs)
    base = synth.make_problem(C, max(2, (2 * B + C - 1) // C), seed, noise_px=0.0)
    assert base.n_boards >= B
    rng = np.random.default_rng(seed)
    intr, cam, brd = base.meta["gt_intr"], base.meta["gt_cam_rt"], base.meta["gt_board_rt"][:B]
    npts = base.n_points
    P3 = np.concatenate([base.board_xy, np.zeros((npts, 1))], axis=1)
    vc, vb, u, v = [], [], [], []
    for b in range(B):
        Rb = synth.rodrigues(brd[b, :3])
        Pw = P3 @ Rb.T + brd[b, 3:]
        for m in sorted(pairs[b % len(pairs)]):
            Pc = Pw @ synth.rodrigues(cam[m, :3]).T + cam[m, 3:]
            uu, vv, ks = synth.ts_project(intr[m], Pc)
            assert np.all(ks > 1e-3), (b, m)
            vc.append(m); vb.append(b); u.append(uu); v.append(vv)
    V = len(vc)
    obs_u = np.concatenate(u) + noise_px * rng.normal(size=V * npts)
    obs_v = np.concatenate(v) + noise_px * rng.normal(size=V * npts)
    p = Problem(C, B, base.board_xy, np.array(vc, dtype=np.int32), np.array(vb, dtype=np.int32),
                (np.arange(V) * npts).astype(np.int32), np.full(V, npts, dtype=np.int32), obs_u, obs_v,
                base.cam_rt.copy(), base.intr.copy(), base.board_rt[:B].copy(), base.cam_pose_constant.copy(), False,
                meta=dict(gt_intr=intr, gt_cam_rt=cam, gt_board_rt=brd))
    return p.normalised()


# ----------------------------------------------------------------------------- rig initialisation
def np_project_skew(I, P):
    """TS.cpp:332-344 in numpy (with the skew terms b, c)."""
    X, Y, Z = P[..., 0], P[..., 1], P[..., 2]
    fx, fy, cx, cy, xi, lam, al, b, c = I
    d1 = np.sqrt(X * X + Y * Y + Z * Z)
    d2 = np.sqrt(X * X + Y * Y + (Z + xi * d1) ** 2)
    d3 = np.sqrt(X * X + Y * Y + (Z + xi * d1 + lam * d2) ** 2)
    ks = Z + xi * d1 + lam * d2 + al / (1 - al) * d3
    return fx * X / ks + b * Y / ks + cx, c * X / ks + fy * Y / ks + cy


def np_Rt_to_R_t(Rt):
    """multi_calib.h:130-137 in numpy float32."""
    r1 = Rt[..., :, 0].astype(np.float32)
    r2 = Rt[..., :, 1].astype(np.float32)
    r3 = np.cross(r1, r2).astype(np.float32)
    return np.stack([r1, r2, r3], axis=-1).astype(np.float64), Rt[..., :, 2].copy()


def np_rig_stage(inp, i, Rp, tp):
    """multi_calib.cpp:25-85 for camera i in numpy: hypotheses and the full error matrix summed per
    hypothesis (pairwise summation, so only ~1e-13 relative agreement with a sequential loop)."""
    common = np.nonzero(inp.has[i - 1].astype(bool) & inp.has[i].astype(bool))[0]
    Ri, ti = np_Rt_to_R_t(inp.Rt[i, common])
    Rk, tk = np_Rt_to_R_t(inp.Rt[i - 1, common])
    Rik = Ri @ np.swapaxes(Rk, 1, 2)
    tik = ti - np.einsum("kij,kj->ki", Rik, tk)
    Rs = Rik @ Rp
    ts = Rik @ tp + tik
    J = common.size
    err = np.zeros(J)
    W = inp.worlds
    for j in range(J):
        A1 = Rp @ Rs[j].T
        a1 = tp - A1 @ ts[j]
        Rv, tv = A1 @ Ri, ti @ A1.T + a1
        P = np.einsum("kij,nj->kni", Rv, W) + tv[:, None, :]
        u, v = np_project_skew(inp.intr[i - 1], P)
        e = np.sqrt((inp.pix_u[i - 1, common] - u) ** 2 + (inp.pix_v[i - 1, common] - v) ** 2).sum()
        A2 = Rs[j] @ Rp.T
        a2 = ts[j] - A2 @ tp
        Rv, tv = A2 @ Rk, tk @ A2.T + a2
        P = np.einsum("kij,nj->kni", Rv, W) + tv[:, None, :]
        u, v = np_project_skew(inp.intr[i], P)
        e += np.sqrt((inp.pix_u[i, common] - u) ** 2 + (inp.pix_v[i, common] - v) ** 2).sum()
        err[j] = e
    return common, Rs, ts, err


def rig_with_unseen_boards(p: Problem, extra: int = 2) -> Problem:
    """Append `extra` boards no camera sees (is_initial() stays false: multi_calib.cpp:98-103)."""
    q = Problem(p.n_cameras, p.n_boards + extra, p.board_xy, p.view_camera, p.view_board, p.view_offset, p.view_count,
                p.obs_u, p.obs_v, p.cam_rt, p.intr, np.concatenate([p.board_rt, np.zeros((extra, 6))]),
                p.cam_pose_constant, False,
                meta=dict(p.meta, gt_board_rt=np.concatenate([p.meta["gt_board_rt"], np.zeros((extra, 6))])))
    return q.normalised()
