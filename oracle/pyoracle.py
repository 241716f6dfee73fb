"""ctypes binding of the CPU oracle (oracle/tscm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (tscm_calib_amd/) must never import
this module.  Parity is unpinned against real Ceres -- see oracle/tscm_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("tscm_oracle.c", "tscm_oracle_rig.c", "tscm_oracle_maps.c", "tscm_oracle_focal.c", "tscm_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _LIB_PATH


class OrcProblem(C.Structure):
    _fields_ = [
        ("n_cameras", C.c_int), ("n_boards", C.c_int), ("n_points", C.c_int), ("n_views", C.c_int),
        ("board_xy", C.c_void_p), ("view_camera", C.c_void_p), ("view_board", C.c_void_p),
        ("view_offset", C.c_void_p), ("view_count", C.c_void_p), ("obs_u", C.c_void_p), ("obs_v", C.c_void_p),
        ("cam_rt", C.c_void_p), ("intr", C.c_void_p), ("board_rt", C.c_void_p),
        ("cam_pose_constant", C.c_void_p), ("mono", C.c_int), ("board_pose_constant", C.c_void_p),
    ]


class OrcOptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int), ("jacobi_scaling", C.c_int),
    ]


class OrcIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


class OrcSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int), ("num_iterations", C.c_int), ("num_successful_steps", C.c_int),
        ("num_unsuccessful_steps", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("n_residual_blocks", C.c_int), ("iterations", OrcIteration * 256), ("message", C.c_char * 128),
        ("seconds_total", C.c_double), ("seconds_jacobian", C.c_double), ("seconds_linear", C.c_double),
    ]


class OrcRigInput(C.Structure):
    _fields_ = [
        ("n_cameras", C.c_int), ("n_boards", C.c_int), ("n_points", C.c_int),
        ("worlds", C.c_void_p), ("intr", C.c_void_p), ("has", C.c_void_p), ("Rt", C.c_void_p),
        ("pix_u", C.c_void_p), ("pix_v", C.c_void_p),
    ]


class OrcMapDesc(C.Structure):
    _fields_ = [
        ("intr", C.c_double * 9), ("R", C.c_double * 9), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
        ("cy", C.c_double), ("offset_x", C.c_double), ("offset_y", C.c_double), ("width", C.c_int), ("height", C.c_int),
        ("out_stride", C.c_int), ("check_w2", C.c_int), ("out_offset", C.c_longlong), ("w2", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # TSCM_ORACLE_LIB: another build of the same sources (tests/test_oracle_sanitizers.py loads liboracle_asan.so
        # into an interpreter started with the sanitizer runtime preloaded)
        path = os.environ.get("TSCM_ORACLE_LIB")
        if not path:
            build()
            path = _LIB_PATH
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.orc_default_options.argtypes = [C.POINTER(OrcOptions), C.c_int]
        L.orc_project.argtypes = [dp, dp, dp]
        L.orc_unproject.argtypes = [dp, dp, dp]
        L.orc_angle_axis_rotate_point.argtypes = [dp, dp, dp]
        L.orc_rodrigues.argtypes = [dp, dp]
        L.orc_mono_residual.argtypes = [dp] * 5
        L.orc_multi_residual.argtypes = [dp] * 6
        L.orc_mono_autodiff.argtypes = [dp] * 7
        L.orc_multi_autodiff.argtypes = [dp] * 9
        L.orc_evaluate.argtypes = [C.POINTER(OrcProblem), C.c_int, dp, dp, dp, dp]
        L.orc_evaluate.restype = C.c_double
        L.orc_solve.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcOptions), C.POINTER(OrcSummary)]
        L.orc_solve.restype = C.c_int
        L.orc_mean_reprojection_error.argtypes = [C.POINTER(OrcProblem), dp]
        L.orc_mean_reprojection_error.restype = C.c_double
        L.orc_rmse.argtypes = [C.POINTER(OrcProblem)]
        L.orc_rmse.restype = C.c_double
        L.orc_rig_init.argtypes = [C.POINTER(OrcRigInput), dp, dp, dp, dp, dp, dp, C.c_void_p, C.c_void_p, dp]
        L.orc_rig_init.restype = C.c_int
        L.orc_rodrigues_inverse.argtypes = [dp, dp]
        L.orc_rig_hypothesis_errors.argtypes = [C.POINTER(OrcRigInput), C.c_int, dp, dp, dp, dp, C.c_int, dp]
        L.orc_rig_hypothesis_errors.restype = None
        L.orc_Rt_to_R_t.argtypes = [dp, dp, dp]
        L.orc_estimate_focal.argtypes = [dp, dp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp,
                                         C.POINTER(C.c_int)]
        L.orc_estimate_focal.restype = C.c_int
        L.orc_focal_sample.argtypes = [dp, dp, C.c_int, C.c_double, C.c_double, dp]
        L.orc_focal_sample.restype = C.c_int
        L.orc_Rt_to_rt.argtypes = [dp, dp]
        L.orc_Rt_to_rt.restype = None
        L.orc_estimate_extrinsic.argtypes = [dp, dp, dp, C.POINTER(C.c_int), C.c_int, dp, C.c_int, C.c_int, dp]
        L.orc_estimate_extrinsic.restype = C.c_int
        L.orc_build_map.argtypes = [C.POINTER(OrcMapDesc), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_build_map.restype = None
        L.orc_rectify_pair_rotation.argtypes = [dp, dp, dp]
        L.orc_rectify_pair_rotation.restype = None
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def c_problem(p) -> OrcProblem:
    """Wrap a tscm_calib_amd.problem.Problem (arrays are referenced, not copied: keep `p` alive)."""
    q = OrcProblem()
    q.n_cameras, q.n_boards, q.n_points, q.n_views = p.n_cameras, p.n_boards, p.n_points, p.n_views
    for name in ("board_xy", "view_camera", "view_board", "view_offset", "view_count", "obs_u", "obs_v",
                 "cam_rt", "intr", "board_rt", "cam_pose_constant"):
        arr = getattr(p, name)
        assert arr.flags["C_CONTIGUOUS"]
        setattr(q, name, arr.ctypes.data)
    q.mono = 1 if p.mono else 0
    bpc = getattr(p, "board_pose_constant", None)
    if bpc is not None:
        assert bpc.dtype == np.uint8 and bpc.flags["C_CONTIGUOUS"]
        q.board_pose_constant = bpc.ctypes.data
    return q


def default_options(mono: bool, **over) -> OrcOptions:
    o = OrcOptions()
    lib().orc_default_options(C.byref(o), 1 if mono else 0)
    for k, v in over.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def summary_dict(s: OrcSummary) -> dict:
    its = []
    for i in range(min(s.num_iterations, 256)):
        it = s.iterations[i]
        its.append({k: getattr(it, k) for k, _ in OrcIteration._fields_})
    return dict(termination_type=s.termination_type, num_iterations=s.num_iterations,
                num_successful_steps=s.num_successful_steps, num_unsuccessful_steps=s.num_unsuccessful_steps,
                initial_cost=s.initial_cost, final_cost=s.final_cost, n_residual_blocks=s.n_residual_blocks,
                iterations=its, message=s.message.decode(), seconds_total=s.seconds_total,
                seconds_jacobian=s.seconds_jacobian, seconds_linear=s.seconds_linear)


def solve(p, **option_overrides) -> dict:
    """Run the oracle LM in place on `p` (a normalised Problem)."""
    q = c_problem(p)
    o = default_options(p.mono, **option_overrides)
    s = OrcSummary()
    rc = lib().orc_solve(C.byref(q), C.byref(o), C.byref(s))
    d = summary_dict(s)
    d["rc"] = rc
    return d


def evaluate(p, jets: bool = True):
    """-> cost, residuals [N,2], J_cam [N,2,6], J_board [N,2,6], J_intr [N,2,9] (jets only)."""
    q = c_problem(p)
    N = p.n_corners
    res = np.zeros((N, 2))
    if jets:
        Jc, Jb, Ji = np.zeros((N, 2, 6)), np.zeros((N, 2, 6)), np.zeros((N, 2, 9))
        cost = lib().orc_evaluate(C.byref(q), 1, _dp(res), _dp(Jc), _dp(Jb), _dp(Ji))
        return cost, res, Jc, Jb, Ji
    cost = lib().orc_evaluate(C.byref(q), 0, _dp(res), None, None, None)
    return cost, res


def project(intr, P) -> np.ndarray:
    intr, P = _f(intr), _f(P)
    out = np.zeros(2)
    lib().orc_project(_dp(intr), _dp(P), _dp(out))
    return out


def unproject(intr, uv) -> np.ndarray:
    intr, uv = _f(intr), _f(uv)
    out = np.zeros(3)
    lib().orc_unproject(_dp(intr), _dp(uv), _dp(out))
    return out


def rotate(aa, pt) -> np.ndarray:
    aa, pt = _f(aa), _f(pt)
    out = np.zeros(3)
    lib().orc_angle_axis_rotate_point(_dp(aa), _dp(pt), _dp(out))
    return out


def rodrigues(aa) -> np.ndarray:
    aa = _f(aa)
    out = np.zeros(9)
    lib().orc_rodrigues(_dp(aa), _dp(out))
    return out.reshape(3, 3)


def mono_residual(intr, rt, obs, board_pt) -> np.ndarray:
    a = [_f(x) for x in (intr, rt, obs, board_pt)]
    out = np.zeros(2)
    lib().orc_mono_residual(*[_dp(x) for x in a], _dp(out))
    return out


def multi_residual(cam_rt, board_rt, intr, obs, board_pt) -> np.ndarray:
    a = [_f(x) for x in (cam_rt, board_rt, intr, obs, board_pt)]
    out = np.zeros(2)
    lib().orc_multi_residual(*[_dp(x) for x in a], _dp(out))
    return out


def mono_autodiff(intr, rt, obs, board_pt):
    a = [_f(x) for x in (intr, rt, obs, board_pt)]
    res, Ji, Jr = np.zeros(2), np.zeros((2, 9)), np.zeros((2, 6))
    lib().orc_mono_autodiff(*[_dp(x) for x in a], _dp(res), _dp(Ji), _dp(Jr))
    return res, Ji, Jr


def multi_autodiff(cam_rt, board_rt, intr, obs, board_pt):
    a = [_f(x) for x in (cam_rt, board_rt, intr, obs, board_pt)]
    res, Jc, Jb, Ji = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 6)), np.zeros((2, 9))
    lib().orc_multi_autodiff(*[_dp(x) for x in a], _dp(res), _dp(Jc), _dp(Jb), _dp(Ji))
    return res, Jc, Jb, Ji


def mean_reprojection_error(p):
    q = c_problem(p)
    per = np.zeros(p.n_cameras)
    g = lib().orc_mean_reprojection_error(C.byref(q), _dp(per))
    return g, per


def rmse(p) -> float:
    q = c_problem(p)
    return lib().orc_rmse(C.byref(q))


def rodrigues_inverse(R) -> np.ndarray:
    R = _f(R)
    out = np.zeros(3)
    lib().orc_rodrigues_inverse(_dp(R), _dp(out))
    return out


def Rt_to_R_t(Rt):
    Rt = _f(Rt)
    R, t = np.zeros(9), np.zeros(3)
    lib().orc_Rt_to_R_t(_dp(Rt), _dp(R), _dp(t))
    return R.reshape(3, 3), t


def c_rig_input(inp) -> OrcRigInput:
    q = OrcRigInput()
    q.n_cameras, q.n_boards, q.n_points = inp.n_cameras, inp.n_boards, inp.n_points
    for name in ("worlds", "intr", "has", "Rt", "pix_u", "pix_v"):
        arr = getattr(inp, name)
        assert arr.flags["C_CONTIGUOUS"]
        setattr(q, name, arr.ctypes.data)
    return q


def rig_hypothesis_errors(inp, i, Rp, tp, Rs, ts) -> np.ndarray:
    """multi_calib.cpp:52-78 for the given hypotheses of camera i."""
    q = c_rig_input(inp)
    Rp, tp, Rs, ts = _f(Rp), _f(tp), _f(Rs), _f(ts)
    nj = Rs.size // 9
    err = np.zeros(nj)
    lib().orc_rig_hypothesis_errors(C.byref(q), int(i), _dp(Rp), _dp(tp), _dp(Rs), _dp(ts), nj, _dp(err))
    return err


def rig_init(inp) -> dict:
    """MultiCalib::MultiCalib (multi_calib.cpp:6-153) on a tscm_calib_amd.rig.RigInput."""
    Cn, B = inp.n_cameras, inp.n_boards
    q = c_rig_input(inp)
    out = dict(cam_R=np.zeros((Cn, 3, 3)), cam_t=np.zeros((Cn, 3)), cam_rt=np.zeros((Cn, 6)),
               board_R=np.zeros((B, 3, 3)), board_t=np.zeros((B, 3)), board_rt=np.zeros((B, 6)),
               board_initial=np.zeros(B, dtype=np.uint8), cam_choice=np.zeros(Cn, dtype=np.int32),
               cam_min_error=np.zeros(Cn))
    rc = lib().orc_rig_init(C.byref(q), _dp(out["cam_R"]), _dp(out["cam_t"]), _dp(out["cam_rt"]),
                            _dp(out["board_R"]), _dp(out["board_t"]), _dp(out["board_rt"]),
                            out["board_initial"].ctypes.data, out["cam_choice"].ctypes.data,
                            _dp(out["cam_min_error"]))
    out["rc"] = rc
    return out


def build_maps(descs, n_elems: int):
    """TS.cpp:284-330 / rectify.cpp:86-199 for a list of tscm_calib_amd.maps.MapDesc -> mapx, mapy (float32)."""
    mapx, mapy = np.zeros(n_elems, dtype=np.float32), np.zeros(n_elems, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    for d in descs:
        m = OrcMapDesc()
        m.intr[:] = list(np.asarray(d.intr, dtype=np.float64).ravel())
        m.R[:] = list(np.asarray(d.R, dtype=np.float64).ravel())
        for k in ("fx", "fy", "cx", "cy", "offset_x", "offset_y", "width", "height", "out_stride", "check_w2", "out_offset", "w2"):
            setattr(m, k, getattr(d, k))
        lib().orc_build_map(C.byref(m), mapx.ctypes.data_as(fp), mapy.ctypes.data_as(fp))
    return mapx, mapy


def rectify_pair_rotation(t1, t2) -> np.ndarray:
    t1, t2 = _f(t1), _f(t2)
    R = np.zeros(9)
    lib().orc_rectify_pair_rotation(_dp(t1), _dp(t2), _dp(R))
    return R.reshape(3, 3)


def estimate_focal(pix_u, pix_v, count, width, height, cx, cy):
    """TS.cpp:110-168 -> focal, number of accepted rows, rc"""
    pix_u, pix_v = _f(pix_u), _f(pix_v)
    count = np.ascontiguousarray(count, dtype=np.int32)
    focal, total = C.c_double(0.0), C.c_int(0)
    rc = lib().orc_estimate_focal(_dp(pix_u), _dp(pix_v), count.ctypes.data_as(C.POINTER(C.c_int)), count.shape[0], width, height,
                                  cx, cy, C.cast(C.byref(focal), C.POINTER(C.c_double)), C.byref(total))
    return focal.value, total.value, rc


def focal_sample(pu, pv, cx, cy):
    pu, pv = _f(pu), _f(pv)
    g = C.c_double(0.0)
    ok = lib().orc_focal_sample(_dp(pu), _dp(pv), pu.shape[0], cx, cy, C.cast(C.byref(g), C.POINTER(C.c_double)))
    return (g.value if ok else None)


def Rt_to_rt(Rt) -> np.ndarray:
    Rt = _f(Rt)
    out = np.zeros(6)
    lib().orc_Rt_to_rt(_dp(Rt), _dp(out))
    return out


def estimate_extrinsic(intr, pix_u, pix_v, count, worlds, board_w):
    """TS.cpp:170-203 (deterministic planar PnP) -> Rt [V,3,3] ([r1 r2 t]), number of poses"""
    intr, pix_u, pix_v, worlds = _f(intr), _f(pix_u), _f(pix_v), _f(worlds)
    count = np.ascontiguousarray(count, dtype=np.int32)
    V, n = count.shape[0], worlds.shape[0]
    Rt = np.zeros((V, 3, 3))
    k = lib().orc_estimate_extrinsic(_dp(intr), _dp(pix_u), _dp(pix_v), count.ctypes.data_as(C.POINTER(C.c_int)), V,
                                     _dp(worlds), n, board_w, _dp(Rt))
    return Rt, k


def detect_corners(gray, sigma: int = 4, cap: int = 8192, planes: bool = False) -> dict:
    """findCorner.cpp:7-46 (+ the sub-pixel fit of :84 for every candidate): candidates of a grey uint8 image in the
    order the non-maximum suppression finds them."""
    g = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = g.shape
    px, py, score = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    v, sub = np.zeros((cap, 4)), np.zeros((cap, 2))
    metric = np.zeros((h, w)) if planes else None
    ixy = np.zeros((h, w)) if planes else None
    f = lib().orc_detect_corners
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    n = f(g.ctypes.data, w, h, w, sigma, cap, px.ctypes.data, py.ctypes.data, v.ctypes.data, score.ctypes.data, sub.ctypes.data,
          metric.ctypes.data if planes else None, ixy.ctypes.data if planes else None)
    if n < 0:
        raise RuntimeError(f"orc_detect_corners failed ({n})")
    m = min(n, cap)
    out = dict(n=n, x=px[:m], y=py[:m], v1=v[:m, :2], v2=v[:m, 2:], score=score[:m], sub=sub[:m])
    if planes:
        out.update(metric=metric, ixy=ixy)
    return out


def chessboards_from_corners(x, y, v1, v2, max_boards: int = 16, max_cells: int = 4096) -> list:
    """DetectCorner/chessboard.cpp:3-103 -> list of index matrices (rows x cols, cols >= rows)."""
    x, y, v1, v2 = _f(x), _f(y), _f(v1), _f(v2)
    n = x.shape[0]
    rows, cols = np.zeros(max_boards, dtype=np.int32), np.zeros(max_boards, dtype=np.int32)
    cells = np.zeros(max_boards * max_cells, dtype=np.int32)
    f = lib().orc_chessboards_from_corners
    f.restype = C.c_int
    f.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 3
    nb = f(n, x.ctypes.data, y.ctypes.data, v1.ctypes.data, v2.ctypes.data, max_boards, max_cells, rows.ctypes.data, cols.ctypes.data, cells.ctypes.data)
    return [cells[q * max_cells:q * max_cells + rows[q] * cols[q]].reshape(rows[q], cols[q]).copy() for q in range(min(nb, max_boards))]


def remap(src, mapx, mapy, to_gray: bool = False) -> np.ndarray:
    """cv::remap(src, dst, mapx, mapy, INTER_LINEAR) for uint8 images (H, W) or (H, W, 3); to_gray: BGR2GRAY of the result."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    ch = 1 if src.ndim == 2 else src.shape[2]
    mapx, mapy = np.ascontiguousarray(mapx, dtype=np.float32), np.ascontiguousarray(mapy, dtype=np.float32)
    mh, mw = mapx.shape
    out_ch = 1 if (to_gray or ch == 1) else ch
    dst = np.zeros((mh, mw) if out_ch == 1 else (mh, mw, out_ch), dtype=np.uint8)
    f = lib().orc_remap_bilinear
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    f(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], ch, mapx.ctypes.data, mapy.ctypes.data, mw, mh, mw, int(bool(to_gray)), dst.ctypes.data, dst.strides[0])
    return dst
