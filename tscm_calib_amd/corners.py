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
