"""Host-side mirror of the reference's interface for the LM hot path, on top of the C ABI.

Reference                                              here
---------------------------------------------------    -----------------------------------------
TripleSphereCamera::refinement (TS.cpp:247-282)        refinement(problem)  -> bool, summary
MultiCalib::calibrate (multi_calib.cpp:155-232)        calibrate(problem)   -> summary
ReprojectionError functors via AutoDiffCostFunction    evaluate_functor(problem)
(TS.h:93-134, multi_calib.h:138-199)
TripleSphereCamera::project (TS.cpp:332-344)           project(intr, points)
get_unit_sphere_coordinate (TS.h:39-57)                unproject(intr, pixels)
error report (multi_calib.cpp:233-283)                 reprojection_error(problem)

Everything computes on the GPU through libtscm_hip.so; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _l
from .problem import Problem


class Solver:
    """RAII wrapper of tscm_solver (problem uploaded once, parameters in/out per solve)."""

    def __init__(self, problem: Problem, device: int = 0, rank: int = 0, world: int = 1):
        """rank / world: frame-sharded solve -- every rank passes the same WHOLE problem and keeps the boards it owns
        (tscm_solver_create_sharded); attach a communicator with set_comm() before solving."""
        self.problem = problem.normalised() if not _is_normalised(problem) else problem
        self.problem.validate()
        self._cp = _l.c_problem(self.problem)
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        _l.check(_l.lib().tscm_solver_create_sharded(C.byref(self._cp), device, rank, world, C.byref(self._h)))
        self._comm = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _l.lib().tscm_solver_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_comm(self, comm: "Comm | None"):
        self._comm = comm
        _l.check(_l.lib().tscm_solver_set_comm(self._h, comm._h if comm else None))

    def debug_withhold_handoff(self, on=True):
        """Tests only: one producer of the device-side hand-off of the NEXT solve never reports in (tscm_solver_debug_withhold_handoff);
        on = 2: ... and the library must not run that solve again on separate launches; on = 3: like 1 for a reduction that rides in
        the Schur-complement launch."""
        _l.check(_l.lib().tscm_solver_debug_withhold_handoff(self._h, int(on)))

    def create_timing(self) -> dict:
        """Where the wall time of this solver's creation went, seconds (tscm_solver_create_timing)."""
        out = (C.c_double * 5)()
        _l.check(_l.lib().tscm_solver_create_timing(self._h, out))
        return dict(zip(("runtime_init", "host_layout", "gather", "h2d", "kernel_setup"), (float(x) for x in out)))

    def debug_perturb_exchange(self, iteration: int, ulps: int = 1):
        """Tests only (rank-divergence guard): in the NEXT solve the Schur-complement tiles this rank RECEIVES from the all-reduce of
        LM iteration `iteration` are moved by `ulps` units in the last place (tscm_solver_debug_perturb_exchange)."""
        _l.check(_l.lib().tscm_solver_debug_perturb_exchange(self._h, int(iteration), int(ulps)))

    def reruns(self) -> int:
        """Solves of this solver that were run again on separate launches after a late device-side hand-off."""
        n = _l.lib().tscm_solver_reruns(self._h)
        if n < 0:
            _l.check(n)
        return n

    def solve(self, **options) -> dict:
        """ceres::Solve equivalent: in/out through problem.cam_rt / intr / board_rt."""
        o = _l.default_options(self.problem.mono, **options)
        s = _l.CSummary()
        _l.check(_l.lib().tscm_solver_solve(self._h, C.byref(o), C.byref(s)))
        return _l.summary_dict(s)

    def upload_params(self, cam_rt=None, intr=None, board_rt=None):
        p = self.problem
        cam_rt = p.cam_rt if cam_rt is None else np.ascontiguousarray(cam_rt, dtype=np.float64)
        intr = p.intr if intr is None else np.ascontiguousarray(intr, dtype=np.float64)
        board_rt = p.board_rt if board_rt is None else np.ascontiguousarray(board_rt, dtype=np.float64)
        _l.check(_l.lib().tscm_solver_upload_params(self._h, _l.dptr(cam_rt), _l.dptr(intr), _l.dptr(board_rt)))

    def solve_resident(self, reset: bool = True, **options) -> dict:
        o = _l.default_options(self.problem.mono, **options)
        s = _l.CSummary()
        _l.check(_l.lib().tscm_solver_solve_resident(self._h, C.byref(o), C.byref(s), 1 if reset else 0))
        return _l.summary_dict(s)

    def download_params(self):
        p = self.problem
        cam, intr, board = np.zeros_like(p.cam_rt), np.zeros_like(p.intr), np.zeros_like(p.board_rt)
        if p.mono:
            cam[:] = p.cam_rt
        _l.check(_l.lib().tscm_solver_download_params(self._h, _l.dptr(cam), _l.dptr(intr), _l.dptr(board)))
        return cam, intr, board

    def gather_boards(self):
        """Complete board_rt on every rank after a sharded solve_resident (tscm_solver_gather_boards)."""
        board = self.problem.board_rt.copy()
        _l.check(_l.lib().tscm_solver_gather_boards(self._h, _l.dptr(board)))
        return board

    def kernel_time(self, enable=True):
        """(timed launches, total_ms) of the dominant kernel since the last call; (re)arms the HIP-event timers:
        enable = False / 0 off, True / 1 every launch, n every n-th launch."""
        n, ms = C.c_int(0), C.c_double(0.0)
        _l.check(_l.lib().tscm_solver_kernel_time(self._h, int(enable), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def exchange_time(self):
        """((timed, total_ms) of the all-reduce of T, (timed, total_ms) of the all-reduce of H_stage) since the last call;
        sampled at kernel_time()'s rate.  Zeros without a communicator."""
        nt, nh, mt, mh = C.c_int(0), C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        _l.check(_l.lib().tscm_solver_exchange_time(self._h, C.byref(nt), C.byref(mt), C.byref(nh), C.byref(mh)))
        return (nt.value, mt.value), (nh.value, mh.value)


class Comm:
    """Communicator of a frame-sharded solve: RCCL (one process per GPU: the constructor), the in-process group
    (`local_group`), or the IPC back-end (`ipc`: one process per rank, ranks may share a device)."""

    def __init__(self, unique_id: bytes | None, rank: int, world: int, device: int, _handle=None):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        if _handle is not None:
            self._h = _handle
            return
        buf = (C.c_ubyte * _l.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _l.check(_l.lib().tscm_comm_create(buf, rank, world, device, C.byref(self._h)))

    @staticmethod
    def local_group(world: int, device: int = 0) -> "list[Comm]":
        """The communicators of an in-process group on one device (tscm_comm_create_local)."""
        hs = (C.c_void_p * world)()
        _l.check(_l.lib().tscm_comm_create_local(world, device, hs))
        return [Comm(None, r, world, device, _handle=C.c_void_p(hs[r])) for r in range(world)]

    @staticmethod
    def ipc(rank: int, world: int, device: int, allgather, n_cameras: int = 32) -> "Comm":
        """Communicator of the IPC back-end (tscm_comm_ipc_open / _connect): one process per rank, the ranks may
        share a device.  `allgather(bytes) -> list[bytes]` is the caller's side channel (every rank calls it once, with
        its TSCM_IPC_HANDLE_BYTES-byte handle; it returns all ranks' handles in rank order).  `n_cameras`: upper bound of the rigs this
        communicator will serve (sizes the exchange slots: 256 doubles per camera-pair block)."""
        h = C.c_void_p()
        mine = (C.c_ubyte * _l.IPC_HANDLE_BYTES)()
        max_doubles = 256 * max(n_cameras * (n_cameras + 1) // 2, n_cameras) + 8 + 2 * world
        _l.check(_l.lib().tscm_comm_ipc_open(rank, world, device, max_doubles, C.byref(h), mine))
        note = _l.lib().tscm_last_error().decode(errors="replace")      # (which kind of memory the exchange buffer got)
        c = Comm(None, rank, world, device, _handle=h)
        c.note = note
        handles = allgather(bytes(mine))
        if len(handles) != world or any(len(x) != _l.IPC_HANDLE_BYTES for x in handles):
            raise RuntimeError(f"allgather must return one {_l.IPC_HANDLE_BYTES}-byte handle per rank")
        buf = (C.c_ubyte * (_l.IPC_HANDLE_BYTES * world)).from_buffer_copy(b"".join(handles))
        _l.check(_l.lib().tscm_comm_ipc_connect(c._h, buf))
        return c

    def backend_ranks(self) -> int:
        """Number of ranks the exchange back-end itself reports (ncclCommCount for RCCL)."""
        n = C.c_int(0)
        _l.check(_l.lib().tscm_comm_info(self._h, None, None, C.byref(n)))
        return n.value

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * _l.UNIQUE_ID_BYTES)()
        _l.check(_l.lib().tscm_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _l.lib().tscm_comm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class Group:
    """`world` shards of one problem on ONE device, solved in lock step with the in-process exchange
    (tscm_comm_create_local + tscm_solver_solve_group): the frame-sharded solver without RCCL, e.g. on a one-GPU box."""

    def __init__(self, problem: Problem, world: int, device: int = 0):
        self.problem = problem.normalised() if not _is_normalised(problem) else problem
        self.world = world
        self.comms = Comm.local_group(world, device)
        self.solvers = [Solver(self.problem, device, r, world) for r in range(world)]
        for s, c in zip(self.solvers, self.comms):
            s.set_comm(c)

    def solve(self, **options) -> "list[dict]":
        """In/out through the problem's arrays, like Solver.solve; returns one summary per rank."""
        for s in self.solvers:
            s.upload_params()
        out = self.solve_resident(reset=True, **options)
        p = self.problem
        cam, intr, _ = self.solvers[0].download_params()
        if not p.mono:
            p.cam_rt[:] = cam
        p.intr[:] = intr
        for s in self.solvers:                                  # every rank writes the boards it owns
            _l.check(_l.lib().tscm_solver_download_params(s._h, None, None, _l.dptr(p.board_rt)))
        return out

    def solve_resident(self, reset: bool = True, **options) -> "list[dict]":
        o = _l.default_options(self.problem.mono, **options)
        hs = (C.c_void_p * self.world)(*[s._h for s in self.solvers])
        sums = (_l.CSummary * self.world)()
        _l.check(_l.lib().tscm_solver_solve_group(hs, self.world, C.byref(o), sums, 1 if reset else 0))
        return [_l.summary_dict(sums[r]) for r in range(self.world)]

    def close(self):
        for s in self.solvers:
            s.close()
        for c in self.comms:
            c.close()
        self.solvers, self.comms = [], []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _is_normalised(p: Problem) -> bool:
    ok = lambda a, dt: isinstance(a, np.ndarray) and a.dtype == dt and a.flags["C_CONTIGUOUS"]
    return (all(ok(getattr(p, n), np.float64) for n in ("board_xy", "obs_u", "obs_v", "cam_rt", "intr", "board_rt"))
            and all(ok(getattr(p, n), np.int32) for n in ("view_camera", "view_board", "view_offset", "view_count"))
            and ok(p.cam_pose_constant, np.uint8)
            and (p.board_pose_constant is None or ok(p.board_pose_constant, np.uint8)))


def calibrate(problem: Problem, device: int = 0, **options) -> dict:
    """The Ceres block of MultiCalib::calibrate() (multi_calib.cpp:157-218): joint LM over
    camera poses, board poses and intrinsics, in place on `problem`.  Like the reference,
    the outcome is not turned into an error -- inspect summary['termination']."""
    if problem.mono:
        raise ValueError("calibrate() is the multi-camera solve; use refinement() for a mono problem")
    assert _is_normalised(problem), "use Problem.normalised()"
    o = _l.default_options(False, **options)
    s = _l.CSummary()
    cp = _l.c_problem(problem)
    _select(device)
    _l.check(_l.lib().tscm_solve_multi(C.byref(cp), C.byref(o), C.byref(s)))
    return _l.summary_dict(s)


def refinement(problem: Problem, device: int = 0, **options):
    """TripleSphereCamera::refinement (TS.cpp:247-282): returns (converged, summary) where
    converged == (termination_type == CONVERGENCE), the reference's return value (:281)."""
    if not problem.mono:
        raise ValueError("refinement() is the mono solve")
    assert _is_normalised(problem), "use Problem.normalised()"
    o = _l.default_options(True, **options)
    s = _l.CSummary()
    cp = _l.c_problem(problem)
    _select(device)
    _l.check(_l.lib().tscm_solve_mono(C.byref(cp), C.byref(o), C.byref(s)))
    d = _l.summary_dict(s)
    return d["termination_type"] == 0, d


def _select(device: int):
    # the one-shot entry points run on the calling thread's current HIP device (device 0 unless
    # the host program selected another one); use Solver(problem, device) to pick a device.
    if device != 0:
        raise ValueError("one-shot calibrate()/refinement() use the current device; use Solver(problem, device)")


def evaluate_functor(problem: Problem, device: int = 0, jacobians: bool = True):
    """-> cost, residuals [N,2], (J_cam [N,2,6], J_board [N,2,6], J_intr [N,2,9])."""
    assert _is_normalised(problem)
    N = problem.n_corners
    res = np.zeros((N, 2))
    cost = C.c_double(0.0)
    cp = _l.c_problem(problem)
    if jacobians:
        Jc, Jb, Ji = np.zeros((N, 2, 6)), np.zeros((N, 2, 6)), np.zeros((N, 2, 9))
        _l.check(_l.lib().tscm_eval_functor(C.byref(cp), device, _l.dptr(res), _l.dptr(Jc), _l.dptr(Jb), _l.dptr(Ji), C.byref(cost)))
        return cost.value, res, Jc, Jb, Ji
    _l.check(_l.lib().tscm_eval_functor(C.byref(cp), device, _l.dptr(res), None, None, None, C.byref(cost)))
    return cost.value, res


def normal_equations(problem: Problem, device: int = 0) -> dict:
    assert _is_normalised(problem)
    Cn, B, V = problem.n_cameras, problem.n_boards, problem.n_views
    out = dict(board_gram=np.zeros((B, 6, 6)), board_grad=np.zeros((B, 6)), view_cross=np.zeros((V, 6, 15)),
               cam_gram=np.zeros((Cn, 15, 15)), cam_grad=np.zeros((Cn, 15)))
    cost = C.c_double(0.0)
    cp = _l.c_problem(problem)
    _l.check(_l.lib().tscm_eval_normal_equations(
        C.byref(cp), device, _l.dptr(out["board_gram"]), _l.dptr(out["board_grad"]), _l.dptr(out["view_cross"]),
        _l.dptr(out["cam_gram"]), _l.dptr(out["cam_grad"]), C.byref(cost)))
    out["cost"] = cost.value
    return out


def project(intr, points, device: int = 0) -> np.ndarray:
    intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(9)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((pts.shape[0], 2))
    _l.check(_l.lib().tscm_project_points(_l.dptr(intr), _l.dptr(pts), pts.shape[0], device, _l.dptr(out)))
    return out


def unproject(intr, pixels, device: int = 0) -> np.ndarray:
    intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(9)
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    out = np.zeros((px.shape[0], 3))
    _l.check(_l.lib().tscm_unproject_pixels(_l.dptr(intr), _l.dptr(px), px.shape[0], device, _l.dptr(out)))
    return out


def reprojection_error(problem: Problem, device: int = 0):
    """-> (per-camera mean pixel error [C], global mean, rmse)  (multi_calib.cpp:233-283)."""
    assert _is_normalised(problem)
    per = np.zeros(problem.n_cameras)
    g, r = C.c_double(0.0), C.c_double(0.0)
    cp = _l.c_problem(problem)
    _l.check(_l.lib().tscm_reprojection_error(C.byref(cp), device, _l.dptr(per), C.byref(g), C.byref(r)))
    return per, g.value, r.value


def shard_owner(problem: Problem, world: int) -> np.ndarray:
    assert _is_normalised(problem)
    owner = np.zeros(problem.n_boards, dtype=np.int32)
    cp = _l.c_problem(problem)
    _l.check(_l.lib().tscm_shard_frames(C.byref(cp), world, owner.ctypes.data_as(C.POINTER(C.c_int))))
    return owner
