"""ctypes binding of the C ABI in include/tscm/tscm.h (libtscm_hip.so).

There is no CPU fallback: if the shared library is missing, or no HIP device is
present, compute calls raise.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libtscm_hip.so")
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 80
MAX_ITERATIONS = 255
# tscm_options.exec_flags (tscm.h)
EXEC_SEPARATE_T_REDUCE = 1
EXEC_KEEP_SINGLE_RANK_COMM = 2
EXEC_GRAM_16X16 = 4
EXEC_SEPARATE_BACKSUB = 8
EXEC_SEPARATE_CONTROL = 16
EXEC_DENSE_REDUCED_ORDER = 32
EXEC_GRAPH_REDUCED_ORDER = 64
EXEC_SEPARATE_STATS = 128
EXEC_MFMA_REDUCED_SOLVE = 256
EXEC_ONE_VIEW_PER_PASS = 512
EXPERIMENT_SCHUR_CHUNK_32 = 0
EXPERIMENT_GRAM_STREAM = 1

E_NAMES = {0: "TSCM_OK", -1: "TSCM_E_INVALID", -2: "TSCM_E_NO_DEVICE", -3: "TSCM_E_HIP",
           -4: "TSCM_E_RCCL", -5: "TSCM_E_UNSUPPORTED", -6: "TSCM_E_NOMEM", -7: "TSCM_E_PEER"}
TERMINATION = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}


class TscmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{E_NAMES.get(code, code)}: {message}")
        self.code = code


class CProblem(C.Structure):
    _fields_ = [
        ("n_cameras", C.c_int), ("n_boards", C.c_int), ("n_points", C.c_int), ("n_views", C.c_int),
        ("board_xy", C.c_void_p), ("view_camera", C.c_void_p), ("view_board", C.c_void_p),
        ("view_offset", C.c_void_p), ("view_count", C.c_void_p), ("obs_u", C.c_void_p), ("obs_v", C.c_void_p),
        ("cam_rt", C.c_void_p), ("intr", C.c_void_p), ("board_rt", C.c_void_p),
        ("cam_pose_constant", C.c_void_p), ("mono", C.c_int), ("board_pose_constant", C.c_void_p),
    ]


class COptions(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t), ("max_num_iterations", C.c_int), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int), ("jacobi_scaling", C.c_int), ("check_every", C.c_int),
        ("jacobian_fp32", C.c_int), ("exec_flags", C.c_int),
    ]


class CIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


class CSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int), ("num_iterations", C.c_int), ("num_successful_steps", C.c_int),
        ("num_unsuccessful_steps", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("n_residual_blocks", C.c_int), ("lm_iterations", C.c_int),
        ("iterations", CIteration * (MAX_ITERATIONS + 1)), ("message", C.c_char * 128),
        ("seconds_solve", C.c_double), ("seconds_total", C.c_double), ("rmse", C.c_double),
    ]


class CRigInput(C.Structure):
    _fields_ = [
        ("n_cameras", C.c_int), ("n_boards", C.c_int), ("n_points", C.c_int),
        ("worlds", C.c_void_p), ("intr", C.c_void_p), ("has", C.c_void_p), ("Rt", C.c_void_p),
        ("pix_u", C.c_void_p), ("pix_v", C.c_void_p),
    ]


class CRigResult(C.Structure):
    _fields_ = [
        ("cam_R", C.c_void_p), ("cam_t", C.c_void_p), ("cam_rt", C.c_void_p),
        ("board_R", C.c_void_p), ("board_t", C.c_void_p), ("board_rt", C.c_void_p),
        ("board_initial", C.c_void_p), ("cam_choice", C.c_void_p), ("cam_min_error", C.c_void_p),
        ("seconds_hypotheses", C.c_double), ("seconds_total", C.c_double), ("n_projections", C.c_longlong),
    ]


class CMapDesc(C.Structure):
    _fields_ = [
        ("intr", C.c_double * 9), ("R", C.c_double * 9), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
        ("cy", C.c_double), ("offset_x", C.c_double), ("offset_y", C.c_double), ("width", C.c_int), ("height", C.c_int),
        ("out_stride", C.c_int), ("check_w2", C.c_int), ("out_offset", C.c_longlong), ("w2", C.c_double),
    ]


class CCornerSet(C.Structure):
    _fields_ = [
        ("n_cameras", C.c_int), ("n_boards", C.c_int), ("board_cols", C.c_int), ("board_rows", C.c_int), ("pitch", C.c_double),
        ("image_width", C.c_int), ("image_height", C.c_int), ("has", C.c_void_p), ("pix_u", C.c_void_p), ("pix_v", C.c_void_p),
    ]


# every symbol include/tscm/tscm.h declares
EXPORTS = [
    "tscm_abi_version", "tscm_last_error", "tscm_device_count", "tscm_device_synchronize", "tscm_device_peak_fp64", "tscm_device_peak_fp64_ex", "tscm_device_peak_fp32_mfma", "tscm_default_options",
    "tscm_solver_create", "tscm_solver_create_timing", "tscm_debug_layout_order", "tscm_debug_gram_plan", "tscm_debug_experiment", "tscm_solver_set_comm", "tscm_solver_debug_withhold_handoff", "tscm_solver_debug_perturb_exchange", "tscm_solver_reruns", "tscm_solver_solve", "tscm_solver_upload_params",
    "tscm_solver_solve_resident", "tscm_solver_download_params", "tscm_solver_destroy",
    "tscm_solver_kernel_time", "tscm_solver_exchange_time", "tscm_solve_multi", "tscm_solve_mono", "tscm_eval_functor",
    "tscm_eval_normal_equations", "tscm_project_points", "tscm_unproject_pixels",
    "tscm_reprojection_error", "tscm_comm_unique_id", "tscm_comm_create", "tscm_comm_destroy",
    "tscm_shard_frames", "tscm_solver_create_sharded", "tscm_comm_create_local", "tscm_comm_ipc_open", "tscm_comm_ipc_connect", "tscm_solver_solve_group",
    "tscm_solver_gather_boards", "tscm_comm_info", "tscm_rig_init", "tscm_yaml_format", "tscm_yaml_write", "tscm_yaml_parse",
    "tscm_yaml_read", "tscm_build_maps", "tscm_estimate_focal", "tscm_poses_from_r1r2t",
    "tscm_estimate_extrinsic", "tscm_corners_write", "tscm_corners_read", "tscm_corners_free",
    "tscm_detect_corners", "tscm_detect_corners_batch", "tscm_corner_candidates_free", "tscm_chessboards_from_corners", "tscm_chessboards_free", "tscm_remap",
]


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    import glob
    srcs = [f for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(CSRC, pat))]     # same list as the Makefile
    srcs.append(os.path.join(_HERE, "..", "include", "tscm", "tscm.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(f) > os.path.getmtime(LIB_PATH) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TscmError(-2, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.tscm_abi_version.restype = C.c_int
    L.tscm_last_error.restype = C.c_char_p
    L.tscm_device_count.restype = C.c_int
    L.tscm_device_synchronize.argtypes = [C.c_int]
    L.tscm_device_peak_fp64.argtypes = [C.c_int, dp, dp]
    L.tscm_device_peak_fp64_ex.argtypes = [C.c_int, dp]
    L.tscm_device_peak_fp32_mfma.argtypes = [C.c_int, dp]
    L.tscm_default_options.argtypes = [C.POINTER(COptions), C.c_int]
    L.tscm_default_options.restype = None
    L.tscm_solver_create.argtypes = [C.POINTER(CProblem), C.c_int, C.POINTER(vp)]
    L.tscm_solver_create_sharded.argtypes = [C.POINTER(CProblem), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.tscm_comm_create_local.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.tscm_comm_ipc_open.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_ubyte)]
    L.tscm_comm_ipc_connect.argtypes = [vp, C.POINTER(C.c_ubyte)]
    L.tscm_solver_solve_group.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(COptions), C.POINTER(CSummary), C.c_int]
    L.tscm_solver_gather_boards.argtypes = [vp, dp]
    L.tscm_comm_info.argtypes = [vp, ip, ip, ip]
    L.tscm_solver_set_comm.argtypes = [vp, vp]
    L.tscm_solver_create_timing.argtypes = [vp, dp]
    L.tscm_debug_layout_order.argtypes = [C.POINTER(CProblem), C.c_int, C.c_int, ip, ip, ip, ip]
    L.tscm_debug_gram_plan.argtypes = [C.c_int, ip, ip, ip, ip]
    L.tscm_debug_experiment.argtypes = [C.c_int, C.c_int]
    L.tscm_solver_debug_withhold_handoff.argtypes = [vp, C.c_int]
    L.tscm_solver_debug_perturb_exchange.argtypes = [vp, C.c_int, C.c_int]
    L.tscm_solver_reruns.argtypes = [vp]
    L.tscm_solver_solve.argtypes = [vp, C.POINTER(COptions), C.POINTER(CSummary)]
    L.tscm_solver_upload_params.argtypes = [vp, dp, dp, dp]
    L.tscm_solver_solve_resident.argtypes = [vp, C.POINTER(COptions), C.POINTER(CSummary), C.c_int]
    L.tscm_solver_download_params.argtypes = [vp, dp, dp, dp]
    L.tscm_solver_destroy.argtypes = [vp]
    L.tscm_solver_destroy.restype = None
    L.tscm_solver_kernel_time.argtypes = [vp, C.c_int, ip, dp]
    L.tscm_solver_exchange_time.argtypes = [vp, ip, dp, ip, dp]
    L.tscm_solve_multi.argtypes = [C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(CSummary)]
    L.tscm_solve_mono.argtypes = [C.POINTER(CProblem), C.POINTER(COptions), C.POINTER(CSummary)]
    L.tscm_eval_functor.argtypes = [C.POINTER(CProblem), C.c_int, dp, dp, dp, dp, dp]
    L.tscm_eval_normal_equations.argtypes = [C.POINTER(CProblem), C.c_int, dp, dp, dp, dp, dp, dp]
    L.tscm_project_points.argtypes = [dp, dp, C.c_int, C.c_int, dp]
    L.tscm_unproject_pixels.argtypes = [dp, dp, C.c_int, C.c_int, dp]
    L.tscm_reprojection_error.argtypes = [C.POINTER(CProblem), C.c_int, dp, dp, dp]
    L.tscm_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
    L.tscm_comm_create.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.tscm_comm_destroy.argtypes = [vp]
    L.tscm_comm_destroy.restype = None
    L.tscm_shard_frames.argtypes = [C.POINTER(CProblem), C.c_int, ip]
    L.tscm_rig_init.argtypes = [C.POINTER(CRigInput), C.c_int, C.POINTER(CRigResult)]
    L.tscm_build_maps.argtypes = [C.POINTER(CMapDesc), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.c_size_t, dp]
    L.tscm_estimate_focal.argtypes = [dp, dp, ip, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, ip]
    L.tscm_poses_from_r1r2t.argtypes = [dp, C.c_void_p, C.c_int, dp]
    L.tscm_estimate_extrinsic.argtypes = [dp, dp, dp, ip, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp, ip]
    L.tscm_corners_write.argtypes = [C.c_char_p, C.POINTER(CCornerSet)]
    L.tscm_corners_read.argtypes = [C.c_char_p, C.POINTER(CCornerSet)]
    L.tscm_corners_free.argtypes = [C.POINTER(CCornerSet)]
    L.tscm_corners_free.restype = None
    L.tscm_yaml_format.argtypes = [C.c_int, dp, dp, dp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tscm_yaml_write.argtypes = [C.c_char_p, C.c_int, dp, dp, dp]
    L.tscm_yaml_parse.argtypes = [C.c_char_p, C.c_int, ip, dp, dp]
    L.tscm_yaml_read.argtypes = [C.c_char_p, C.c_int, ip, dp, dp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise TscmError(rc, lib().tscm_last_error().decode(errors="replace"))


def dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def c_problem(p) -> CProblem:
    """Wrap a normalised Problem. Arrays are referenced, not copied: keep `p` alive."""
    q = CProblem()
    q.n_cameras, q.n_boards, q.n_points, q.n_views = p.n_cameras, p.n_boards, p.n_points, p.n_views
    for name in ("board_xy", "view_camera", "view_board", "view_offset", "view_count", "obs_u", "obs_v",
                 "cam_rt", "intr", "board_rt", "cam_pose_constant"):
        arr = getattr(p, name)
        if not arr.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{name} must be C-contiguous (use Problem.normalised())")
        setattr(q, name, arr.ctypes.data)
    q.mono = 1 if p.mono else 0
    bpc = getattr(p, "board_pose_constant", None)
    if bpc is not None:
        if bpc.dtype != np.uint8 or not bpc.flags["C_CONTIGUOUS"]:
            raise ValueError("board_pose_constant must be C-contiguous uint8 (use Problem.normalised())")
        q.board_pose_constant = bpc.ctypes.data
    return q


def default_options(mono: bool, **over) -> COptions:
    o = COptions()
    lib().tscm_default_options(C.byref(o), 1 if mono else 0)
    for k, v in over.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def summary_dict(s: CSummary) -> dict:
    its = []
    for i in range(min(s.num_iterations, MAX_ITERATIONS + 1)):
        it = s.iterations[i]
        its.append({k: getattr(it, k) for k, _ in CIteration._fields_})
    return dict(termination_type=s.termination_type, termination=TERMINATION.get(s.termination_type, "?"),
                num_iterations=s.num_iterations, num_successful_steps=s.num_successful_steps,
                num_unsuccessful_steps=s.num_unsuccessful_steps, initial_cost=s.initial_cost,
                final_cost=s.final_cost, n_residual_blocks=s.n_residual_blocks, lm_iterations=s.lm_iterations,
                iterations=its, message=s.message.decode(), seconds_solve=s.seconds_solve,
                seconds_total=s.seconds_total, rmse=s.rmse)


class CCornerCandidates(C.Structure):
    """tscm_corner_candidates (tscm.h)"""
    _fields_ = [("n", C.c_int), ("n_maxima", C.c_int),
                ("x", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
                ("v1", C.POINTER(C.c_double)), ("v2", C.POINTER(C.c_double)),
                ("score", C.POINTER(C.c_double)), ("sub", C.POINTER(C.c_double)),
                ("seconds", C.c_double)]


class CChessboards(C.Structure):
    """tscm_chessboards (tscm.h)"""
    _fields_ = [("n_boards", C.c_int), ("rows", C.POINTER(C.c_int)), ("cols", C.POINTER(C.c_int)),
                ("offset", C.POINTER(C.c_int)), ("cells", C.POINTER(C.c_int))]
