"""Image-level flow of the reference's main.cpp on top of the library: monocular_calib (main.cpp:8-130: corner
detection, TripleSphereCamera::calibrate, the refinement pass on the remapped chessboards with the flip rule, second
calibrate) and the rig calibration that follows it (MultiCalib constructor + calibrate, main.cpp:196-319).
Host orchestration only: every numeric step is a call into the C ABI (GPU), nothing is computed here."""
from __future__ import annotations

import numpy as np

from . import api, corners, maps, rig, synth
from .problem import Problem


def board_points(cols: int, rows: int, pitch: float) -> np.ndarray:
    """main.cpp:12-18: (v * size, u * size, 0), v fastest."""
    v, u = np.meshgrid(np.arange(cols), np.arange(rows))
    return np.stack([v.ravel() * pitch, u.ravel() * pitch, np.zeros(cols * rows)], axis=1).astype(np.float64)


def calibrate_camera(pu, pv, has, cols, rows, pitch, img_size, device=0, init_intr=None):
    """TripleSphereCamera::calibrate (TS.cpp:30-105).  Without an initial guess (has_init_guess_ false, :41-51):
    principal point at the image centre, xi = lambda = 0, alpha = 0.5, estimate_focal.  With init_intr (the member
    intrinsic_ after a converged earlier refinement set has_init_guess_, :78) those steps are skipped and only the
    extrinsics are re-estimated (:52).  Then estimate_extrinsic, refinement.
    Returns (intr[9], Rt[V,3,3] = [r1 r2 t], summary)."""
    n = cols * rows
    W = board_points(cols, rows, pitch)
    count = (np.asarray(has, dtype=np.int32) * n).astype(np.int32)
    if init_intr is None:
        intr = np.array([0.0, 0.0, img_size[0] / 2 - 0.5, img_size[1] / 2 - 0.5, 0.0, 0.0, 0.5, 0.0, 0.0])
        intr[0] = intr[1] = rig.estimate_focal(pu, pv, count, cols, rows, intr[2], intr[3], device)[0]
    else:
        intr = np.array(init_intr, dtype=np.float64).reshape(9).copy()
    Rt0, _ = rig.estimate_extrinsic(intr, pu, pv, count, W, cols, device)
    sel = np.flatnonzero(count > 0)
    V = sel.shape[0]
    q = Problem(1, V, W[:, :2].copy(), np.zeros(V, dtype=np.int32), np.arange(V, dtype=np.int32), (np.arange(V) * n).astype(np.int32),
                np.full(V, n, dtype=np.int32), pu[sel].ravel().copy(), pv[sel].ravel().copy(), np.zeros((1, 6)), intr[None, :].copy(),
                rig.poses_from_Rt(Rt0[sel]), np.ones(1, dtype=np.uint8), True).normalised()
    _, summary = api.refinement(q, device)
    R = synth.rodrigues(q.board_rt[:, :3])                                   # TS.cpp:88-102
    Rt = np.zeros((count.shape[0], 3, 3))
    Rt[sel] = np.stack([R[:, :, 0], R[:, :, 1], q.board_rt[:, 3:]], axis=2)
    return q.intr[0].copy(), Rt, summary


def _top_left_is_bright(board_img, pitch):
    """main.cpp:76-85: grey values at the centres of the squares (0,0), (1,0), (1,1), (0,1) of the remapped board."""
    g = lambda x, y: int(board_img[int(y), int(x)])
    return g(pitch / 2, pitch / 2) + g(pitch * 3 / 2, pitch * 3 / 2) > g(pitch * 3 / 2, pitch / 2) + g(pitch / 2, pitch * 3 / 2)


def monocular_calib(images, cols: int, rows: int, pitch: float, sigma: int = 4, device: int = 0) -> dict:
    """main.cpp:8-130 for one camera.  images: list of (H, W) uint8 (or (H, W, 3) BGR) arrays, one per frame.
    Returns intr, Rt [V,3,3], has [V], pix_u / pix_v [V, n] (refined, flip rule applied), the two LM summaries."""
    n, V = cols * rows, len(images)
    grey = [im if im.ndim == 2 else maps.remap(im, *np.meshgrid(np.arange(im.shape[1], dtype=np.float32), np.arange(im.shape[0], dtype=np.float32)),
                                                to_gray=True, device=device) for im in images]
    img_size = (grey[0].shape[1], grey[0].shape[0])
    has = np.zeros(V, dtype=np.uint8)
    pu, pv = np.zeros((V, n)), np.zeros((V, n))
    for i, pts in enumerate(corners.find_chessboards(grey, cols, rows, sigma=sigma, device=device)):     # :24-50, one batch
        if pts is not None:
            has[i], pu[i], pv[i] = 1, pts[:, 0], pts[:, 1]
    intr, Rt, first = calibrate_camera(pu, pv, has, cols, rows, pitch, img_size, device)          # :57
    seen = np.flatnonzero(has)                                                # :59-126 refinement pass
    board_imgs = []
    for i in seen:
        desc = maps.chessboard_desc(intr, Rt[i], cols, rows, pitch)
        mx, my, _ = maps.build_maps([desc], desc.width * desc.height, device)
        board_imgs.append(maps.remap(images[i], mx.reshape(desc.height, desc.width), my.reshape(desc.height, desc.width), to_gray=images[i].ndim == 3,
                                     device=device))
    # (main.cpp:67: the remapped board is rejected only if NO board came out or board 0 has the wrong shape)
    for i, board_img, pts in zip(seen, board_imgs, corners.find_chessboards(board_imgs, cols, rows, sigma=sigma, device=device, first_board_only=True)):
        if pts is not None:                                                   # :92-105 back through [r1 r2 t] and project()
            P = (Rt[i] @ np.concatenate([pts - pitch, np.ones((n, 1))], axis=1).T).T
            uv = api.project(intr, P, device)
            pu[i], pv[i] = uv[:, 0], uv[:, 1]
        if _top_left_is_bright(board_img, pitch):                             # :72-89 / :107-121 flip rule
            pu[i], pv[i] = pu[i][::-1].copy(), pv[i][::-1].copy()
    # :127 -- the second calibrate() of the same object: a converged first refinement left has_init_guess_ set (TS.cpp:78),
    # so it starts from the first-pass intrinsics and only re-estimates the extrinsics
    warm = intr if first["termination_type"] == 0 else None
    intr, Rt, second = calibrate_camera(pu, pv, has, cols, rows, pitch, img_size, device, init_intr=warm)
    return dict(intr=intr, Rt=Rt, has=has, pix_u=pu, pix_v=pv, first=first, second=second)


def calibrate_rig(images_by_camera, cols: int, rows: int, pitch: float, sigma: int = 4, device: int = 0) -> dict:
    """main.cpp:196-303: monocular_calib per camera, MultiCalib(cameras, worlds), calibrate().  images_by_camera[m][f] is
    the image of frame f in camera m.  Returns the joint problem (intr, cam_rt, board_rt), the per-camera results
    and the LM summary; write the YAML with calib_io.write_calib_yaml."""
    mono = [monocular_calib(imgs, cols, rows, pitch, sigma, device) for imgs in images_by_camera]
    W = board_points(cols, rows, pitch)
    inp = rig.RigInput(W, np.stack([m["intr"] for m in mono]), np.stack([m["has"] for m in mono]), np.stack([m["Rt"] for m in mono]),
                       np.stack([m["pix_u"] for m in mono]), np.stack([m["pix_v"] for m in mono])).normalised()
    g = rig.rig_init(inp, device)
    problem = rig.problem_from_rig(inp, g)
    summary = api.calibrate(problem, device)
    return dict(problem=problem, mono=mono, rig_init=g, summary=summary)
