"""Host-side description of a TSCM calibration problem (plain numpy, no torch).

Mirrors the data the reference hands to Ceres:
  * mono:  TripleSphereCamera::refinement(pixels, worlds)      (TS.cpp:247-282)
           parameter blocks intrinsic_[9] (shared) and rt_[i][6] per image.
  * multi: MultiCalib::calibrate()                              (multi_calib.cpp:155-218)
           parameter blocks cameras_[m].rt_[6], chessboards_[i].rt_[6],
           cameras_[m].intrinsic_[9]; cameras_[0].rt_ constant (:186).

A *view* is one (camera, board/frame) pair that has corners: in the reference a view
without a detection is an empty `pixels[i]` (main.cpp:35-37) and contributes no
residual blocks (multi_calib.cpp:169).  Corner j of a view observes board point j.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class Problem:
    n_cameras: int
    n_boards: int
    board_xy: np.ndarray          # [n_points, 2] float64 (z forced to 0: TS.h:107-109)
    view_camera: np.ndarray       # [n_views] int32
    view_board: np.ndarray        # [n_views] int32
    view_offset: np.ndarray       # [n_views] int32 (first corner in obs_u/obs_v)
    view_count: np.ndarray        # [n_views] int32
    obs_u: np.ndarray             # [N] float64
    obs_v: np.ndarray             # [N] float64
    cam_rt: np.ndarray            # [C, 6] float64  angle-axis + translation (in/out)
    intr: np.ndarray              # [C, 9] float64  fx fy cx cy xi lambda alpha b c (in/out)
    board_rt: np.ndarray          # [B, 6] float64  (in/out)
    cam_pose_constant: np.ndarray # [C] uint8
    mono: bool = False
    meta: dict = field(default_factory=dict)
    board_pose_constant: "np.ndarray | None" = None   # [B] uint8, 1 = pose block held constant (None: all free, as in the reference)

    @property
    def n_points(self) -> int:
        return int(self.board_xy.shape[0])

    @property
    def n_views(self) -> int:
        return int(self.view_camera.shape[0])

    @property
    def n_corners(self) -> int:
        return int(self.view_count.sum())

    def copy(self) -> "Problem":
        return Problem(
            self.n_cameras, self.n_boards, self.board_xy.copy(), self.view_camera.copy(),
            self.view_board.copy(), self.view_offset.copy(), self.view_count.copy(),
            self.obs_u.copy(), self.obs_v.copy(), self.cam_rt.copy(), self.intr.copy(),
            self.board_rt.copy(), self.cam_pose_constant.copy(), self.mono, dict(self.meta),
            None if self.board_pose_constant is None else self.board_pose_constant.copy())

    def normalised(self) -> "Problem":
        """Contiguous, correctly typed arrays (what the C ABI expects)."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        return Problem(
            int(self.n_cameras), int(self.n_boards), f(self.board_xy).reshape(-1, 2), i(self.view_camera),
            i(self.view_board), i(self.view_offset), i(self.view_count), f(self.obs_u), f(self.obs_v),
            f(self.cam_rt).reshape(-1, 6), f(self.intr).reshape(-1, 9), f(self.board_rt).reshape(-1, 6),
            np.ascontiguousarray(self.cam_pose_constant, dtype=np.uint8), bool(self.mono), dict(self.meta),
            None if self.board_pose_constant is None else np.ascontiguousarray(self.board_pose_constant, dtype=np.uint8))

    def validate(self) -> None:
        C, B, V = self.n_cameras, self.n_boards, self.n_views
        if self.cam_rt.shape != (C, 6) or self.intr.shape != (C, 9) or self.board_rt.shape != (B, 6):
            raise ValueError("parameter array shapes do not match n_cameras/n_boards")
        if self.mono and C != 1:
            raise ValueError("mono problem needs exactly one camera")
        if self.board_pose_constant is not None and self.board_pose_constant.shape != (B,):
            raise ValueError("board_pose_constant must have n_boards entries")
        for a in (self.view_board, self.view_offset, self.view_count):
            if a.shape != (V,):
                raise ValueError("view arrays must all have n_views entries")
        if V and (self.view_camera.min() < 0 or self.view_camera.max() >= C):
            raise ValueError("view_camera out of range")
        if V and (self.view_board.min() < 0 or self.view_board.max() >= B):
            raise ValueError("view_board out of range")
        if V and (self.view_count.min() < 0 or self.view_count.max() > self.n_points):
            raise ValueError("view_count must be in [0, n_points]")
        if V and int((self.view_offset + self.view_count).max()) > self.obs_u.shape[0]:
            raise ValueError("view_offset+view_count exceeds observation arrays")
        if self.obs_u.shape != self.obs_v.shape:
            raise ValueError("obs_u / obs_v length mismatch")


def shard_frames(p: Problem, rank: int, world: int) -> Problem:
    """Frame-sharded sub-problem for `rank` of `world`: contiguous ranges of
    board indices balanced by corner count; every view of a frame stays on one rank so
    that the board's 6x6 Schur block is rank-local.  Camera parameters are replicated,
    board poses are kept full-length (only the owned boards have views, hence residuals).
    """
    if world == 1:
        return p
    B = p.n_boards
    per_board = np.bincount(p.view_board, weights=p.view_count.astype(np.float64), minlength=B)
    csum = np.concatenate([[0.0], np.cumsum(per_board)])
    total = csum[-1]
    # board b belongs to rank floor(world * (corners before b) / total), clipped
    owner = np.minimum((csum[:-1] * world / max(total, 1.0)).astype(np.int64), world - 1)
    sel = owner[p.view_board] == rank
    q = p.copy()
    q.view_camera = p.view_camera[sel]
    q.view_board = p.view_board[sel]
    q.view_count = p.view_count[sel]
    # compact the observations of the selected views
    idx = np.concatenate([np.arange(o, o + c) for o, c in zip(p.view_offset[sel], p.view_count[sel])]) \
        if sel.any() else np.zeros(0, dtype=np.int64)
    q.obs_u = p.obs_u[idx]
    q.obs_v = p.obs_v[idx]
    q.view_offset = (np.cumsum(q.view_count) - q.view_count).astype(np.int32)
    q.meta = dict(p.meta, rank=rank, world=world, owned_boards=np.nonzero(owner == rank)[0])
    return q
