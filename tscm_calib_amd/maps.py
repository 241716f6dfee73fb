"""Remap tables: host-side mirror of TripleSphereCamera::undistort (TS.cpp:284-306), the table of
undistort_chessboard (TS.cpp:308-330) and Remap::init_remap (EpipolarRectify/rectify.cpp:86-199),
all built by tscm_build_maps on the device."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import lib as _lib


@dataclass
class MapDesc:
    intr: np.ndarray            # [9] camera that is sampled
    R: np.ndarray               # [3,3]
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int
    offset_x: float = 0.0
    offset_y: float = 0.0
    out_offset: int = 0
    out_stride: int = 0         # 0 -> width
    check_w2: int = 0
    w2: float = 0.42399         # rectify.cpp:7

    def __post_init__(self):
        if self.out_stride == 0:
            self.out_stride = self.width


def undistort_desc(intr, fx, fy, cx, cy, width, height, **kw) -> MapDesc:
    """TripleSphereCamera::undistort(fx, fy, cx, cy, img_size, mapx, mapy)."""
    return MapDesc(np.asarray(intr, dtype=np.float64), np.eye(3), fx, fy, cx, cy, width, height, **kw)


def chessboard_desc(intr, Rt, chessboard_cols, chessboard_rows, chessboard_size, **kw) -> MapDesc:
    """The table of undistort_chessboard(src, index, chessboard, chessboard_size): Rt = Rt_[index];
    output (cols+1)*size x (rows+1)*size, P = Rt * (j - size, i - size, 1)."""
    w, h = int((chessboard_cols + 1) * chessboard_size), int((chessboard_rows + 1) * chessboard_size)
    return MapDesc(np.asarray(intr, dtype=np.float64), np.asarray(Rt, dtype=np.float64).reshape(3, 3), 1.0, 1.0,
                   float(chessboard_size), float(chessboard_size), w, h, **kw)


def rectify_pair_rotation(t1, t2) -> np.ndarray:
    """Remap::calc_R (rectify.cpp:234-248)."""
    x = np.asarray(t2, dtype=np.float64) - np.asarray(t1, dtype=np.float64)
    n = np.sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2])
    if n != 0:
        x = x / n
    z = np.array([-x[2], 0.0, x[0]])
    n = np.sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2])
    if n != 0:
        z = z / n
    y = np.array([-z[2] * x[1] + z[1] * x[2], z[2] * x[0] - z[0] * x[2], -z[1] * x[0] + z[0] * x[1]])
    n = np.sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2])
    if n != 0:
        y = y / n
    return np.stack([x, y, z], axis=1)


def rectify_descs(intr4, Twc4, size=400, mosaic_w=1280.0, mosaic_h=1080.0):
    """The eight tables of Remap::init_remap for cameras (front, right, rear, left) = cam0..3 of the
    calibration file.  Returns (descs, n_elems): elements [0, 4*size*size) are left_map (400 x 1600),
    [4*size*size, 8*size*size) right_map, each four size x size blocks stacked vertically."""
    intr4 = np.asarray(intr4, dtype=np.float64).reshape(4, 9)
    Twc4 = np.asarray(Twc4, dtype=np.float64).reshape(4, 3, 4)
    Rc, tc = Twc4[:, :, :3], Twc4[:, :, 3]
    origin = [(0.0, 0.0), (mosaic_w, 0.0), (0.0, mosaic_h), (mosaic_w, mosaic_h)]   # front, right, rear, left in the mosaic
    block = size * size
    descs = []
    for k in range(4):                       # pairs (front,right), (right,rear), (rear,left), (left,front)
        a, b = k, (k + 1) % 4
        Rp = rectify_pair_rotation(tc[a], tc[b])
        for cam, side in ((a, 0), (b, 1)):   # side 0 -> left_map, 1 -> right_map   (rectify.cpp:95-118)
            descs.append(MapDesc(intr4[cam], Rc[cam].T @ Rp, size / 2.0, size / 2.0, size / 2.0, size / 2.0, size, size,
                                 offset_x=origin[cam][0], offset_y=origin[cam][1],
                                 out_offset=side * 4 * block + k * block, check_w2=1))
    return descs, 8 * block


def _c_descs(descs):
    arr = (_lib.CMapDesc * len(descs))()
    for m, d in zip(arr, descs):
        m.intr[:] = list(np.asarray(d.intr, dtype=np.float64).ravel())
        m.R[:] = list(np.asarray(d.R, dtype=np.float64).ravel())
        for k in ("fx", "fy", "cx", "cy", "offset_x", "offset_y", "width", "height", "out_stride", "check_w2", "out_offset", "w2"):
            setattr(m, k, getattr(d, k))
    return arr


def build_maps(descs, n_elems: int | None = None, device: int = 0, exact: bool = True):
    """tscm_build_maps -> mapx, mapy (float32, flat), seconds_kernel"""
    if n_elems is None:
        n_elems = max((d.out_offset + (d.height - 1) * d.out_stride + d.width for d in descs if d.width and d.height), default=0)
    mapx, mapy = np.zeros(n_elems, dtype=np.float32), np.zeros(n_elems, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    sec = C.c_double(0.0)
    arr = _c_descs(descs)
    _lib.check(_lib.lib().tscm_build_maps(arr, len(descs), device, 1 if exact else 0, mapx.ctypes.data_as(fp),
                                           mapy.ctypes.data_as(fp), n_elems, C.cast(C.byref(sec), C.POINTER(C.c_double))))
    return mapx, mapy, sec.value


def remap(src, mapx, mapy, to_gray: bool = False, device: int = 0) -> np.ndarray:
    """cv::remap(src, dst, mapx, mapy, INTER_LINEAR) on the GPU (TS.cpp:304, :329) for uint8 images (H, W) or
    (H, W, 3) and float32 tables of the output size; to_gray: BGR2GRAY of the result (findCorner.cpp:9-10)."""
    import ctypes as C
    from . import lib as _l
    src = np.ascontiguousarray(src, dtype=np.uint8)
    ch = 1 if src.ndim == 2 else src.shape[2]
    mapx, mapy = np.ascontiguousarray(mapx, dtype=np.float32), np.ascontiguousarray(mapy, dtype=np.float32)
    if mapx.shape != mapy.shape or mapx.ndim != 2:
        raise ValueError("mapx and mapy must be 2-D tables of the same shape")
    mh, mw = mapx.shape
    out_ch = 1 if (to_gray or ch == 1) else ch
    dst = np.zeros((mh, mw) if out_ch == 1 else (mh, mw, out_ch), dtype=np.uint8)
    f = _l.lib().tscm_remap
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    _l.check(f(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], ch, mapx.ctypes.data, mapy.ctypes.data, mw, mh, mw, int(bool(to_gray)), int(device),
               dst.ctypes.data, dst.strides[0] if mh else mw * out_ch))
    return dst
