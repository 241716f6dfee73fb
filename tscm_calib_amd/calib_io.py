"""Result I/O: the calibration YAML of main.cpp:305-319 (cv::FileStorage layout), read back by
EpipolarRectify/rectify.cpp:262-270.  ctypes mirror of tscm_yaml_* (host-only entry points)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def _arrays(intr, cam_R, cam_t):
    intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(-1, 9)
    cam_R = np.ascontiguousarray(cam_R, dtype=np.float64).reshape(-1, 3, 3)
    cam_t = np.ascontiguousarray(cam_t, dtype=np.float64).reshape(-1, 3)
    if not (intr.shape[0] == cam_R.shape[0] == cam_t.shape[0]):
        raise ValueError("intr, cam_R, cam_t disagree on the number of cameras")
    return intr, cam_R, cam_t


def format_calib_yaml(intr, cam_R, cam_t) -> str:
    intr, cam_R, cam_t = _arrays(intr, cam_R, cam_t)
    L = _lib.lib()
    need = C.c_size_t(0)
    _lib.check(L.tscm_yaml_format(intr.shape[0], _lib.dptr(intr), _lib.dptr(cam_R), _lib.dptr(cam_t), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _lib.check(L.tscm_yaml_format(intr.shape[0], _lib.dptr(intr), _lib.dptr(cam_R), _lib.dptr(cam_t), buf, need.value, None))
    return buf.value.decode()


def write_calib_yaml(path: str, intr, cam_R, cam_t) -> None:
    intr, cam_R, cam_t = _arrays(intr, cam_R, cam_t)
    _lib.check(_lib.lib().tscm_yaml_write(path.encode(), intr.shape[0], _lib.dptr(intr), _lib.dptr(cam_R), _lib.dptr(cam_t)))


def _parse(call, arg):
    L = _lib.lib()
    n = C.c_int(0)
    _lib.check(call(arg, 0, C.byref(n), None, None))
    intr, Twc = np.zeros((n.value, 9)), np.zeros((n.value, 3, 4))
    _lib.check(call(arg, n.value, C.byref(n), _lib.dptr(intr), _lib.dptr(Twc)))
    return intr, Twc


def parse_calib_yaml(text: str):
    """-> intr [C,9], Twc [C,3,4]"""
    return _parse(_lib.lib().tscm_yaml_parse, text.encode())


def read_calib_yaml(path: str):
    return _parse(_lib.lib().tscm_yaml_read, path.encode())


def write_corners(path: str, has, pix_u, pix_v, board_cols: int, board_rows: int, pitch: float, image_size=(1280, 1080)) -> None:
    """tscm_corners_write: has [C,B], pix_u / pix_v [C,B,cols*rows] (the layout of rig.RigInput)."""
    has = np.ascontiguousarray(has, dtype=np.uint8)
    pix_u = np.ascontiguousarray(pix_u, dtype=np.float64)
    pix_v = np.ascontiguousarray(pix_v, dtype=np.float64)
    Cn, B = has.shape
    if pix_u.shape != (Cn, B, board_cols * board_rows) or pix_v.shape != pix_u.shape:
        raise ValueError("pix_u / pix_v must be [C, B, cols*rows]")
    cs = _lib.CCornerSet(Cn, B, board_cols, board_rows, float(pitch), int(image_size[0]), int(image_size[1]),
                         has.ctypes.data, pix_u.ctypes.data, pix_v.ctypes.data)
    _lib.check(_lib.lib().tscm_corners_write(path.encode(), C.byref(cs)))


def read_corners(path: str) -> dict:
    """tscm_corners_read -> dict(has [C,B], pix_u, pix_v [C,B,n], board_cols, board_rows, pitch, image_size)."""
    cs = _lib.CCornerSet()
    L = _lib.lib()
    _lib.check(L.tscm_corners_read(path.encode(), C.byref(cs)))
    try:
        Cn, B, n = cs.n_cameras, cs.n_boards, cs.board_cols * cs.board_rows
        def arr(ptr, ctype, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).astype(dtype, copy=True)
        has = arr(cs.has, C.c_ubyte, Cn * B, np.uint8).reshape(Cn, B)
        pu = arr(cs.pix_u, C.c_double, Cn * B * n, np.float64).reshape(Cn, B, n)
        pv = arr(cs.pix_v, C.c_double, Cn * B * n, np.float64).reshape(Cn, B, n)
        return dict(has=has, pix_u=pu, pix_v=pv, board_cols=cs.board_cols, board_rows=cs.board_rows, pitch=cs.pitch,
                    image_size=(cs.image_width, cs.image_height))
    finally:
        L.tscm_corners_free(C.byref(cs))
