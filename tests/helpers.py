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


def mixed_visibility_rig(seed=5, n_frames=24, n_cameras=4, noise_px=0.05) -> Problem:
    """Rig whose frames are seen by 1..C cameras (random subsets).  Observations are exact
    projections of the ground truth (image bounds ignored: the solver does not care), plus noise."""
    base = synth.make_problem(n_cameras, 2 * n_frames // n_cameras * 2, seed, noise_px=0.0)
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
