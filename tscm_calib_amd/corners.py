"""Chessboard-corner candidates of a grey image on the GPU: host-side mirror of findCorner()'s first stage
(DetectCorner/findCorner.cpp:7-66 + :492-541) over tscm_detect_corners.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _l


def detect_corners(gray, sigma: int = 4, min_score: float = 0.01, device: int = 0) -> dict:
    """gray: (H, W) uint8.  Returns the kept candidates in suppression order: x, y (pixel of the maximum), v1, v2
    (edge directions), score, sub (sub-pixel position), plus n_maxima and the device time in seconds."""
    g = np.asarray(gray)
    if g.ndim != 2 or g.dtype != np.uint8:
        raise ValueError("detect_corners expects a 2-D uint8 image (convert BGR to grey first, findCorner.cpp:9-10)")
    if g.strides[1] != 1 or g.strides[0] < g.shape[1]:          # rows must be contiguous; a row stride > width is passed through
        g = np.ascontiguousarray(g)
    h, w = g.shape
    out = _l.CCornerCandidates()
    f = _l.lib().tscm_detect_corners
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(_l.CCornerCandidates)]
    _l.check(f(g.ctypes.data, w, h, g.strides[0], int(sigma), float(min_score), int(device), C.byref(out)))
    try:
        n = out.n
        arr = lambda p, k: np.ctypeslib.as_array(p, shape=(n * k,)).copy().reshape(n, k) if n else np.zeros((0, k))
        return dict(n=n, n_maxima=out.n_maxima, x=arr(out.x, 1)[:, 0], y=arr(out.y, 1)[:, 0], v1=arr(out.v1, 2), v2=arr(out.v2, 2),
                    score=arr(out.score, 1)[:, 0], sub=arr(out.sub, 2), seconds=out.seconds)
    finally:
        fr = _l.lib().tscm_corner_candidates_free
        fr.restype = None
        fr.argtypes = [C.POINTER(_l.CCornerCandidates)]
        fr(C.byref(out))


def _unpack(out) -> dict:
    n = out.n
    arr = lambda p, k: np.ctypeslib.as_array(p, shape=(n * k,)).copy().reshape(n, k) if n else np.zeros((0, k))
    return dict(n=n, n_maxima=out.n_maxima, x=arr(out.x, 1)[:, 0], y=arr(out.y, 1)[:, 0], v1=arr(out.v1, 2), v2=arr(out.v2, 2),
                score=arr(out.score, 1)[:, 0], sub=arr(out.sub, 2), seconds=out.seconds)


def detect_corners_batch(images, sigma: int = 4, min_score: float = 0.01, device: int = 0) -> list:
    """tscm_detect_corners_batch: a list of (H, W) uint8 images of one size -> list of candidate dicts, one pass of the
    kernels for all of them."""
    imgs = [np.ascontiguousarray(g) for g in images]
    if not imgs:
        return []
    h, w = imgs[0].shape
    if any(g.ndim != 2 or g.dtype != np.uint8 or g.shape != (h, w) for g in imgs):
        raise ValueError("detect_corners_batch expects 2-D uint8 images of one size")
    n = len(imgs)
    ptrs = (C.c_void_p * n)(*[g.ctypes.data for g in imgs])
    outs = (_l.CCornerCandidates * n)()
    f = _l.lib().tscm_detect_corners_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    _l.check(f(ptrs, n, w, h, w, int(sigma), float(min_score), int(device), outs))
    fr = _l.lib().tscm_corner_candidates_free
    fr.restype = None
    fr.argtypes = [C.POINTER(_l.CCornerCandidates)]
    try:
        return [_unpack(outs[i]) for i in range(n)]
    finally:
        for i in range(n):
            fr(C.byref(outs[i]))


def chessboards_from_corners(x, y, v1, v2) -> list:
    """chessboardsFromCorners (DetectCorner/chessboard.cpp:3-103): list of index matrices (rows x cols, cols >= rows)
    into the candidate list.  Host logic of the library (no device needed)."""
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    v1, v2 = np.ascontiguousarray(v1, dtype=np.float64), np.ascontiguousarray(v2, dtype=np.float64)
    out = _l.CChessboards()
    f = _l.lib().tscm_chessboards_from_corners
    f.restype = C.c_int
    f.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.POINTER(_l.CChessboards)]
    _l.check(f(int(x.shape[0]), x.ctypes.data, y.ctypes.data, v1.ctypes.data, v2.ctypes.data, C.byref(out)))
    try:
        res = []
        for q in range(out.n_boards):
            r, c, o = out.rows[q], out.cols[q], out.offset[q]
            res.append(np.array([out.cells[o + k] for k in range(r * c)], dtype=np.int32).reshape(r, c))
        return res
    finally:
        fr = _l.lib().tscm_chessboards_free
        fr.restype = None
        fr.argtypes = [C.POINTER(_l.CChessboards)]
        fr(C.byref(out))


def find_chessboard(gray, cols: int, rows: int, sigma: int = 4, device: int = 0):
    """The acceptance test of main.cpp:32-46: corner candidates, structure recovery, and -- when exactly one board of
    cols x rows inner corners came out -- its sub-pixel corners as a (rows * cols, 2) array in board order; else None."""
    d = detect_corners(gray, sigma=sigma, device=device)
    boards = chessboards_from_corners(d["x"], d["y"], d["v1"], d["v2"])
    if len(boards) != 1 or boards[0].shape != (rows, cols):
        return None
    return d["sub"][boards[0].ravel()]


def find_chessboards(images, cols: int, rows: int, sigma: int = 4, device: int = 0, first_board_only: bool = False) -> list:
    """find_chessboard for a list of images of one size: ONE pass of the detection kernels for all of them
    (tscm_detect_corners_batch), then the structure recovery per image.  Entries are (rows * cols, 2) arrays or None.
    Acceptance: exactly one board of rows x cols corners (main.cpp:40-46, the input images); first_board_only: at least
    one board and board 0 has the right shape (main.cpp:67, the remapped chessboards of the refinement pass)."""
    out = []
    for d in detect_corners_batch(images, sigma=sigma, device=device):
        boards = chessboards_from_corners(d["x"], d["y"], d["v1"], d["v2"])
        ok = (len(boards) >= 1 if first_board_only else len(boards) == 1) and boards[0].shape == (rows, cols)
        out.append(d["sub"][boards[0].ravel()] if ok else None)
    return out
