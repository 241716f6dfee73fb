"""Rig initialisation: host-side mirror of the constructor MultiCalib::MultiCalib(cameras, worlds)
(multi_calib.cpp:6-153), the step that produces the initial guess calibrate() starts from.

RigInput is what the constructor reads from the mono-calibrated cameras:
cameras_[m].intrinsic(), has_chessboard(j), Rt(j), pixels()[j] and the board points.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import lib as _lib
from .problem import Problem


@dataclass
class RigInput:
    worlds: np.ndarray      # [n,3]   board points (main.cpp:12-18)
    intr: np.ndarray        # [C,9]
    has: np.ndarray         # [C,B]   uint8
    Rt: np.ndarray          # [C,B,3,3]  [r1 r2 t] (TS.cpp:150-204 output, TS.h:33)
    pix_u: np.ndarray       # [C,B,n]
    pix_v: np.ndarray       # [C,B,n]
    meta: dict = field(default_factory=dict)

    @property
    def n_cameras(self) -> int:
        return self.intr.shape[0]

    @property
    def n_boards(self) -> int:
        return self.has.shape[1]

    @property
    def n_points(self) -> int:
        return self.worlds.shape[0]

    def normalised(self) -> "RigInput":
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        return RigInput(f(self.worlds), f(self.intr), np.ascontiguousarray(self.has, dtype=np.uint8), f(self.Rt),
                        f(self.pix_u), f(self.pix_v), self.meta)


def rig_init(inp: RigInput, device: int = 0) -> dict:
    """tscm_rig_init.  Returns cam_R/cam_t/cam_rt, board_R/board_t/board_rt, board_initial,
    cam_choice, cam_min_error and the device timings."""
    inp = inp.normalised()
    Cn, B, n = inp.n_cameras, inp.n_boards, inp.n_points
    if inp.has.shape != (Cn, B) or inp.Rt.shape != (Cn, B, 3, 3) or inp.pix_u.shape != (Cn, B, n) \
            or inp.pix_v.shape != (Cn, B, n) or inp.worlds.shape != (n, 3) or inp.intr.shape != (Cn, 9):
        raise ValueError("inconsistent RigInput shapes")
    q = _lib.CRigInput()
    q.n_cameras, q.n_boards, q.n_points = Cn, B, n
    for name in ("worlds", "intr", "has", "Rt", "pix_u", "pix_v"):
        setattr(q, name, getattr(inp, name).ctypes.data)
    out = dict(cam_R=np.zeros((Cn, 3, 3)), cam_t=np.zeros((Cn, 3)), cam_rt=np.zeros((Cn, 6)),
               board_R=np.zeros((B, 3, 3)), board_t=np.zeros((B, 3)), board_rt=np.zeros((B, 6)),
               board_initial=np.zeros(B, dtype=np.uint8), cam_choice=np.zeros(Cn, dtype=np.int32),
               cam_min_error=np.zeros(Cn))
    r = _lib.CRigResult()
    for name in ("cam_R", "cam_t", "cam_rt", "board_R", "board_t", "board_rt", "board_initial", "cam_choice",
                 "cam_min_error"):
        setattr(r, name, out[name].ctypes.data)
    _lib.check(_lib.lib().tscm_rig_init(C.byref(q), device, C.byref(r)))
    out.update(seconds_hypotheses=r.seconds_hypotheses, seconds_total=r.seconds_total,
               n_projections=r.n_projections)
    return out


def problem_from_rig(inp: RigInput, init: dict) -> Problem:
    """The ceres::Problem MultiCalib::calibrate() builds from the constructor's result
    (multi_calib.cpp:157-207): one view per (camera, initialised board) with pixels."""
    Cn, B, n = inp.n_cameras, inp.n_boards, inp.n_points
    cams, boards = np.nonzero(inp.has.astype(bool) & init["board_initial"].astype(bool)[None, :])
    order = np.lexsort((cams, boards))       # board-major like the reference loop (:164-168)
    cams, boards = cams[order], boards[order]
    V = cams.shape[0]
    const = np.zeros(Cn, dtype=np.uint8)
    const[0] = 1
    p = Problem(Cn, B, inp.worlds[:, :2].copy(), cams.astype(np.int32), boards.astype(np.int32),
                (np.arange(V) * n).astype(np.int32), np.full(V, n, dtype=np.int32),
                inp.pix_u[cams, boards].ravel().copy(), inp.pix_v[cams, boards].ravel().copy(),
                init["cam_rt"].copy(), inp.intr.copy(), init["board_rt"].copy(), const, False, meta=dict(inp.meta))
    return p.normalised()


def estimate_focal(pix_u, pix_v, count, board_w: int, board_h: int, cx: float, cy: float, device: int = 0):
    """tscm_estimate_focal (TripleSphereCamera::estimate_focal, TS.cpp:110-168) -> focal, accepted rows."""
    pix_u = np.ascontiguousarray(pix_u, dtype=np.float64)
    pix_v = np.ascontiguousarray(pix_v, dtype=np.float64)
    count = np.ascontiguousarray(count, dtype=np.int32)
    if pix_u.size != count.shape[0] * board_w * board_h or pix_v.size != pix_u.size:
        raise ValueError("pix_u / pix_v must hold n_views * board_w * board_h values")
    focal, used = C.c_double(0.0), C.c_int(0)
    _lib.check(_lib.lib().tscm_estimate_focal(_lib.dptr(pix_u), _lib.dptr(pix_v), count.ctypes.data_as(C.POINTER(C.c_int)),
                                              count.shape[0], board_w, board_h, cx, cy, device,
                                              C.cast(C.byref(focal), C.POINTER(C.c_double)), C.byref(used)))
    return focal.value, used.value


def poses_from_Rt(Rt, has=None) -> np.ndarray:
    """tscm_poses_from_r1r2t (TS.cpp:62-74): [n,3,3] [r1 r2 t] -> [n,6] angle-axis + t (host-only)."""
    Rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(-1, 3, 3)
    rt = np.zeros((Rt.shape[0], 6))
    h = None if has is None else np.ascontiguousarray(has, dtype=np.uint8)
    _lib.check(_lib.lib().tscm_poses_from_r1r2t(_lib.dptr(Rt), None if h is None else h.ctypes.data, Rt.shape[0], _lib.dptr(rt)))
    return rt


def estimate_extrinsic(intr, pix_u, pix_v, count, worlds, board_w: int, device: int = 0):
    """tscm_estimate_extrinsic (TripleSphereCamera::estimate_extrinsic, TS.cpp:170-203, with a deterministic
    planar PnP in place of cv::solvePnPRansac) -> Rt [V,3,3] = [r1 r2 t] per image, number of poses."""
    intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(9)
    pix_u = np.ascontiguousarray(pix_u, dtype=np.float64)
    pix_v = np.ascontiguousarray(pix_v, dtype=np.float64)
    count = np.ascontiguousarray(count, dtype=np.int32)
    worlds = np.ascontiguousarray(worlds, dtype=np.float64).reshape(-1, 3)
    V, n = count.shape[0], worlds.shape[0]
    if pix_u.size != V * n or pix_v.size != V * n:
        raise ValueError("pix_u / pix_v must hold n_views * n_points values")
    Rt = np.zeros((V, 3, 3))
    done = C.c_int(0)
    _lib.check(_lib.lib().tscm_estimate_extrinsic(_lib.dptr(intr), _lib.dptr(pix_u), _lib.dptr(pix_v),
                                                  count.ctypes.data_as(C.POINTER(C.c_int)), V, _lib.dptr(worlds), n, board_w,
                                                  device, _lib.dptr(Rt), C.byref(done)))
    return Rt, done.value
