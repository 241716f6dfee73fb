#!/usr/bin/env python3
"""Benchmark of the TSCM LM hot path on MI355X.

Metric (BASELINE.json): LM iterations/sec at 4 cams x 10k views (config 4: 20,000 frames,
40,000 views, 2,160,000 corners, 9x6 board), joint intrinsics + extrinsics, fp64.

A "step" is ONE Levenberg-Marquardt iteration of the whole job: e-block elimination,
Schur complement, reduced solve, back-substitution, and the fused residual + analytic
Jacobian + Gram evaluation at the candidate point, plus the accept/reject decision.  The
timed region runs K iterations as ceil(K/50) solves of <= 50 iterations each ("for a
stable rate run with termination tests disabled for a fixed 50 iterations"), every solve
restarting from the same perturbed initial guess that is already resident in HBM; the
iteration-0 evaluation of every solve is inside the timed region but not counted as a step.
Termination tests are disabled for the timed solves (tolerances < 0) so that exactly K
iterations run: at config 4 a 50-iteration solve is 45 accepted and 5 rejected steps, all valid --
every one of them the full work of an LM iteration.  Parity on THIS trajectory (tests/test_gpu_trajectory.py, GPU against
the oracle over the same 50 forced iterations): cost 1e-9, poses 1e-8, but the intrinsics only 1e-4 -- past convergence
the iterates crawl along the fx / xi / lambda / alpha valley, where rounding decides the direction; the
NATURAL solve, which stops at the function tolerance, holds every parameter to 1e-6 (`natural_solve`, `cpu_baseline`).  (Round 1 and the first half of round 2 used
10 iterations per solve, i.e. one extra evaluation per 10 steps: TSCM_BENCH_ITERS_PER_SOLVE=10.)

N > 1: one process per GPU, frames sharded across the ranks (strong scaling: the job is fixed), two RCCL
all-reduces per iteration.  The ranks are either started by `python -m torch.distributed.run ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or by this script itself: `python bench.py --gpus N`
without WORLD_SIZE spawns N fresh child processes BEFORE anything touches the GPU.  No torch in any of them: the
ncclUniqueId, the barriers and the max-over-ranks travel over a TCP side channel (tscm_calib_amd/rendezvous.py), so
the only HIP runtime and RCCL in the process are the ones libtscm_hip.so links.  (Under a launcher whose ranks do NOT share a
parent process -- a wrapper shell per rank, several nodes -- set TSCM_RDZV_NONCE to one value for all ranks of the launch: without it the
side channel's handshake token contains the parent's process id, which keeps stale ranks of an earlier launch out.)
TSCM_BENCH_EXCHANGE=ipc replaces RCCL by the library's IPC exchange back-end (tscm_comm_ipc_open): the rank processes may then
share a device (rank r on device r mod n_devices) -- the multi-process path on a one-GPU box; the line says "exchange": "ipc" and
that it is not a scaling measurement.  Without it, more ranks than devices is an error.

Before anything is timed (all of it untimed, none of it skipping work inside the timed region): the device's fp64
ceilings are measured (three dense kernels, ~70 ms: they are part of the roofline block anyway and leave the device
at sustained clocks), the natural solve runs, and the hot path itself is run for TSCM_BENCH_PREHEAT_MS (60 ms) -- then
the W warmup steps of the contract, the barrier, and exactly K timed steps.  With 5 warmup steps alone the first
timed steps ran on a device still ramping from idle (dominant kernel 62 us instead of 55-57).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel k_eval_gram4, timed with HIP events on the
solver's own stream (the event pair rides in the kernel's dispatch: hipExtLaunchKernelGGL); `cpu_baseline` is the CPU
oracle (a plain-C port of the reference's Ceres path) on the full workload, 1 thread like the reference, plus an
all-cores figure.
"""
import argparse
import hashlib
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (the library takes at most TSCM_MAX_ITERATIONS = 255 iterations per solve)
ITERS_PER_SOLVE = min(255, max(1, int(os.environ.get("TSCM_BENCH_ITERS_PER_SOLVE", "50"))))
PREHEAT_MS = float(os.environ.get("TSCM_BENCH_PREHEAT_MS", "60"))
# algorithmic work of one k_eval_gram launch (DESIGN.md "roofline accounting")
FLOP_MFMA_PER_CORNER = 836           # Gram contraction 2P(P+1)+4P with P=19
FLOP_VALU_PER_CORNER = 600           # hand-structured residual + analytic Jacobian
FLOP_PER_CORNER = FLOP_MFMA_PER_CORNER + FLOP_VALU_PER_CORNER
BYTES_PER_CORNER = 16.0 + 168.0 / 54.0   # 9x6 board; main() recomputes it for --board (168 B of pose / intrinsic reads per VIEW)
FP64_PEAK_TFLOPS = 78.6              # MI355X FP64 vector = matrix peak (AMD datasheet; not in MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3             # MI355X FP32 vector (packed) = FP32 matrix peak, same datasheet
HBM_PEAK_GBS = 8000.0
BENCH_OPTS = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0,
                  min_trust_region_radius=0.0, check_every=ITERS_PER_SOLVE)
CPU_BASELINE_ITERS = 6               # 1 thread: 4 s per iteration of the full config 4 on the boxes seen so far -- ~24 s of CPU work
EVENT_STRIDE_MAX = 8                 # an event pair holds the stream for ~11 us (6.4 in front of the launch, 4.6 behind it: kernel
                                     # trace of the driver's command, profiles/r05_eval_fixed_cost.txt): every 8th launch is timed
MIN_TIMED_LAUNCHES = 4               # (TSCM_BENCH_EVENTS_IN_TIMED=1 only) short runs time more of them, so that at least this many are measured
# Round 6: the dominant kernel is timed during the untimed PREHEAT loop -- the same launches on the same data, ~70 of them -- and
# the timed region carries no event pair at all (rounds 1-5: five pairs = 55 us inside the driver's 20 timed steps, 2.75 us per
# step of self-inflicted measurement).  `roofline.timed_in` says which.  TSCM_BENCH_EVENTS_IN_TIMED=1 restores the old protocol.
EVENTS_IN_TIMED = os.environ.get("TSCM_BENCH_EVENTS_IN_TIMED") == "1"


def event_stride(steps: int) -> int:
    """Launches of the dominant kernel in the timed region: one per LM iteration plus the iteration-0 evaluation of
    every solve.  Every k-th is bracketed by HIP events, k as large as EVENT_STRIDE_MAX allows while still timing
    MIN_TIMED_LAUNCHES of them."""
    launches = steps + math.ceil(steps / ITERS_PER_SOLVE)
    return max(1, min(EVENT_STRIDE_MAX, launches // MIN_TIMED_LAUNCHES))


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """Start n rank processes of this script (one per GPU) and wait for them.  Runs in a parent that has not loaded
    the library or touched HIP; the children are fresh interpreters (subprocess, never exec after GPU init).
    MASTER_PORT is only the base of the side channel's port search (rank 0 binds the first free port above it and the
    others probe the same list, authenticated by a token that contains this launch's random id), so it does not matter
    if somebody else takes the probed port between here and there.  Whatever ends the launcher -- a rank that fails, an
    exception, SIGINT / SIGTERM -- ends the remaining ranks too."""
    import signal
    import uuid
    port = _free_port()
    run = uuid.uuid4().hex
    procs = []

    def _raise(signum, _frame):
        raise KeyboardInterrupt(f"signal {signum}")

    old = {sig: signal.signal(sig, _raise) for sig in (signal.SIGINT, signal.SIGTERM)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), TSCM_RDZV_RUN=run)
            env.pop("TSCM_RDZV_PORT", None)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:             # a rank died: its peers would wait for it forever
                        q.terminate()
            time.sleep(0.05)
    except BaseException:
        rc = rc or 130
        raise
    finally:
        for p in procs:                           # nothing outlives the launcher
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


# ------------------------------------------------------------------------------------------------ stub (CPU tests)
class _StubSolver:
    """TSCM_BENCH_STUB=1: stands in for the GPU solver so that the launcher, the side channel, the timing protocol and
    the JSON line can be exercised by the CPU test suite (tests/test_bench_launcher.py).  Never used otherwise."""

    def __init__(self, problem, device=0, rank=0, world=1):
        self.problem, self.rank, self.world = problem, rank, world

    def set_comm(self, comm):
        pass

    def upload_params(self):
        pass

    def solve_resident(self, reset=True, max_num_iterations=50, **kw):
        time.sleep(1e-3 * max_num_iterations * (1 + self.rank))      # rank r is (r + 1) x slower: max-over-ranks is visible
        return dict(lm_iterations=max_num_iterations, num_iterations=max_num_iterations + 1, message="stub", rmse=0.0,
                    seconds_solve=0.0, seconds_total=0.0)

    def kernel_time(self, enable=True):
        return 0, 0.0

    def exchange_time(self):
        return (0, 0.0), (0, 0.0)

    def close(self):
        pass


def _kernel_src_sha() -> str:
    """Identity of the hot kernel's sources (comments and white space do not count): profiles/pmc_eval_gram.json
    records it with the traffic measurement."""
    import re
    h = hashlib.sha256()
    for f in ("tscm_kernels.h", "tscm_eval_gram4.h", "tscm_math.h", "tscm_fastmath.h", "tscm_eval_f32.h"):
        with open(os.path.join(ROOT, "tscm_calib_amd", "csrc", f), "r") as fh:
            code = re.sub(r"//[^\n]*", "", fh.read())
            h.update(re.sub(r"\s+", "", code).encode())
    return h.hexdigest()[:16]


def run_iterations(solver, n_iter, **extra):
    """Run exactly n_iter LM iterations as solves of <= ITERS_PER_SOLVE iterations."""
    done = 0
    s = None
    while done < n_iter:
        k = min(ITERS_PER_SOLVE, n_iter - done)
        s = solver.solve_resident(reset=True, max_num_iterations=k, **BENCH_OPTS, **extra)
        if s["lm_iterations"] != k:
            raise RuntimeError(f"expected {k} LM iterations, device ran {s['lm_iterations']} ({s['message']})")
        done += k
    return s


def _grid_delta(api, intr_gpu, intr_cpu, device):
    """Rays of a 25 x 21 grid of pixels (1280 x 1080 image) under the GPU-fitted model, projected with the CPU-fitted
    one: the largest pixel displacement, over all cameras."""
    import numpy as np
    uu, vv = np.meshgrid(np.linspace(40, 1240, 25), np.linspace(40, 1040, 21))
    px = np.stack([uu.ravel(), vv.ravel()], axis=1)
    worst = 0.0
    for a, b in zip(intr_gpu, intr_cpu):
        rays = api.unproject(a, px, device)
        ok = np.all(np.isfinite(rays), axis=1)
        d = api.project(b, rays[ok], device) - px[ok]
        worst = max(worst, float(np.max(np.hypot(d[:, 0], d[:, 1]))))
    return worst


def _usable_cores() -> int:
    """Host cores this process may actually use: the affinity mask AND the cgroup CPU quota (a container that shows 256
    CPUs with cpu.max = "1600000 100000" gets 16 cores' worth of time; more threads than that only add contention)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(full_problem, device, iters=CPU_BASELINE_ITERS):
    """The CPU oracle (port of the reference's Ceres DENSE_SCHUR LM) on the FULL workload, termination tests disabled,
    a fixed number of iterations: once with 1 thread (what the reference uses: num_threads is never set) and once
    with all host cores (OpenMP over residual blocks / e-blocks: SURVEY 8d's generous baseline).  The GPU runs the
    same iterations from the same start: that is the "RMSE delta vs CPU" of the metric."""
    import numpy as np
    from oracle import pyoracle as orc
    from tscm_calib_amd import api
    opts = dict(max_num_iterations=iters, function_tolerance=-1.0, parameter_tolerance=-1.0,
                gradient_tolerance=-1.0, min_trust_region_radius=0.0)
    pg = full_problem.copy().normalised()
    with api.Solver(pg, device=device) as gs:
        g = gs.solve(**opts)
    L = orc.lib()
    po = full_problem.copy().normalised()
    L.orc_set_num_threads(1)
    t0 = time.time()
    s = orc.solve(po, **opts)
    wall1 = time.time() - t0
    n_it = s["num_iterations"] - 1
    cores = min(int(L.orc_max_threads()), _usable_cores())
    pm = full_problem.copy().normalised()
    L.orc_set_num_threads(cores)
    t0 = time.time()
    sm = orc.solve(pm, **opts)
    wallm = time.time() - t0
    L.orc_set_num_threads(1)
    n = full_problem.n_corners
    rmse_cpu = math.sqrt(2.0 * s["final_cost"] / n)
    return {
        "value": n_it / s["seconds_total"], "unit": "LM iterations/s", "cores": 1, "kind": "port",
        "sample": f"the full workload ({n} corners), {n_it} LM iterations with the termination tests off: "
                  f"{s['seconds_total']:.1f} s in the minimiser (wall {wall1:.1f} s), 1 thread as in the reference",
        "all_cores": {"value": (sm["num_iterations"] - 1) / sm["seconds_total"], "cores": cores, "host_cpus_visible": os.cpu_count(),
                      "seconds": sm["seconds_total"], "wall_seconds": wallm,
                      "rmse_rel_delta_vs_1_thread": abs(math.sqrt(2.0 * sm["final_cost"] / n) - rmse_cpu) / rmse_cpu},
        "rmse_px_cpu": rmse_cpu, "rmse_px_gpu_same_iterations": g["rmse"],
        "rmse_rel_delta": abs(g["rmse"] - rmse_cpu) / rmse_cpu,
        "max_rel_intrinsics_delta": float(np.max(np.abs(pg.intr[:, :7] - po.intr[:, :7]) / np.abs(po.intr[:, :7]))),
        # SURVEY 8d parity procedure: pixel-space difference of the two fitted models over a 25 x 21 image grid
        "max_pixel_delta_25x21_grid": _grid_delta(api, pg.intr, po.intr, device),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json config index (4 = headline)")
    ap.add_argument("--poses-fixed", action="store_true",
                    help="hold every board / view pose block constant (the intrinsics-only form of config 2)")
    ap.add_argument("--board", default="9x6", metavar="CxR",
                    help="inner corners of the chessboard (default 9x6: BASELINE's workload and the headline; 11x8 is the reference's own "
                         "board, main.cpp:190-191).  The board keeps its 360 mm extent, the views / frames of --config are unchanged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exec-flags", type=int, default=0,
                    help="tscm_options.exec_flags for every solve (A/B runs: 4 = TSCM_EXEC_GRAM_16X16, 1 = separate T reduction)")
    ap.add_argument("--jacobian-fp32", action="store_true",
                    help="north_star's 1e-3 tier: fp32 derivatives + fp32 MFMA contraction (default: all fp64, the headline)")
    args = ap.parse_args()

    force_dist = os.environ.get("TSCM_BENCH_FORCE_DIST") == "1"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        # not under a launcher: become one.  Nothing in this process has loaded the library or initialised the GPU.
        sys.exit(launch_ranks(max(1, args.gpus), sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("TSCM_BENCH_STUB") == "1"
    if stub and os.environ.get("TSCM_BENCH_STUB_FAIL_RANK") == str(rank):
        sys.exit(7)                                            # tests/test_bench_launcher.py: a rank that dies early
    # TSCM_BENCH_FORCE_DIST=1 runs the multi-process code path (launcher, side channel, RCCL communicator) with a
    # single rank, so that it can be exercised on a one-GPU box
    multi = world > 1 or force_dist
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}: running {world} rank(s)", file=sys.stderr)

    from tscm_calib_amd import synth
    chan = None
    if multi:
        from tscm_calib_amd.rendezvous import SideChannel
        chan = SideChannel(rank, world)

    if stub:
        full = synth.make_problem(4, 8, 123)
        api = lib = None
        Solver = _StubSolver
        device, exchange, n_dev = local_rank, "stub", world
    else:
        from tscm_calib_amd import api, lib
        cols, rows = (int(x) for x in args.board.lower().split("x"))
        board_kw = {} if (cols, rows) == (9, 6) else dict(cols=cols, rows=rows, pitch=360.0 / (max(cols, rows) - 1))
        full = synth.make_config(args.config, poses_fixed=args.poses_fixed, **board_kw)
        Solver = api.Solver
        n_dev = lib.lib().tscm_device_count()
        exchange = os.environ.get("TSCM_BENCH_EXCHANGE", "rccl" if n_dev >= world else "").lower()
        if n_dev <= local_rank and exchange != "ipc":
            raise SystemExit(f"rank {rank}: HIP device {local_rank} does not exist ({n_dev} visible): "
                             f"--gpus {world} needs {world} GPUs (one process per GPU; RCCL refuses two ranks on one device). "
                             f"TSCM_BENCH_EXCHANGE=ipc runs the {world} rank processes on the visible device(s) with the library's IPC exchange")
        device = local_rank % max(1, n_dev)

    if not stub and os.environ.get("TSCM_BENCH_EXPERIMENT"):              # A/B runs of the layout experiments: "which=value[,which=value]"
        for kv in os.environ["TSCM_BENCH_EXPERIMENT"].split(","):
            k, v = kv.split("=")
            lib.check(lib.lib().tscm_debug_experiment(int(k), int(v)))
    t_create = time.perf_counter()
    solver = Solver(full, device=device, rank=rank, world=world)          # H2D of this rank's observations + layout build
    t_create = time.perf_counter() - t_create
    comm = None
    # rccl_ranks: what ncclCommCount reports -- 0 unless RCCL carries the exchange (no communicator on one GPU, none under
    # TSCM_BENCH_EXCHANGE=ipc); exchange_ranks: the ranks of the exchange whatever the back-end
    rccl_ranks, exchange_ranks = 0, (world if stub else 1)
    if multi and not stub:
        if exchange == "ipc":
            comm = api.Comm.ipc(rank, world, device, chan.allgather_bytes, n_cameras=full.n_cameras)
        else:
            uid = chan.bcast(api.Comm.unique_id() if rank == 0 else None)
            comm = api.Comm(uid, rank, world, device)
        solver.set_comm(comm)
        exchange_ranks = comm.backend_ranks()
        rccl_ranks = exchange_ranks if exchange == "rccl" else 0
        # RCCL writes its version banner to the C stdout buffer at communicator creation; push it out now so that the
        # JSON line stays the LAST line of this program's output
        import ctypes
        ctypes.CDLL(None).fflush(None)
    solver.upload_params()

    def barrier():
        # device fence (the job of torch.cuda.synchronize() in the contract) + process barrier
        if not stub:
            lib.check(lib.lib().tscm_device_synchronize(device))
        if chan:
            chan.barrier()

    # measured fp64 ceilings of THIS device (three dense kernels, ~50 ms), before anything is timed: they also take the
    # device from its idle state to sustained clocks -- with 5 warmup steps (0.7 ms) alone the first timed steps run
    # on a device that is still ramping (k_eval_gram4 62 us instead of 57)
    pk = None
    if not stub:
        import ctypes
        pk = (ctypes.c_double * 3)()
        lib.check(lib.lib().tscm_device_peak_fp64_ex(device, pk))
        pk32 = ctypes.c_double(0.0)
        if args.jacobian_fp32:
            lib.check(lib.lib().tscm_device_peak_fp32_mfma(device, ctypes.byref(pk32)))
    # natural solve (reference options, termination tests on): untimed, doubles as warmup
    extra = dict(jacobian_fp32=1) if args.jacobian_fp32 else {}
    if args.exec_flags:
        extra["exec_flags"] = args.exec_flags
    natural = solver.solve_resident(reset=True, **extra)
    # the hot path itself for PREHEAT_MS before the W warmup steps (untimed): the fp64 ceilings above leave the device
    # at sustained clocks, this keeps it there through the host-side work in between
    # HIP events around every stride-th launch of the dominant kernel -- and, with a communicator, around every stride-th all-reduce
    # of each kind -- DURING THE PREHEAT (round 6), not inside the timed region
    no_events = bool(os.environ.get("TSCM_BENCH_NO_EVENTS"))
    stride = 0 if no_events else (event_stride(args.steps) if EVENTS_IN_TIMED else EVENT_STRIDE_MAX)
    if not EVENTS_IN_TIMED:
        solver.kernel_time(enable=stride)
        solver.exchange_time()
    t_pre = time.perf_counter()
    while not stub and (time.perf_counter() - t_pre) * 1e3 < PREHEAT_MS:
        run_iterations(solver, ITERS_PER_SOLVE, **extra)
    if not EVENTS_IN_TIMED:
        launches, kms = solver.kernel_time(enable=False)
        (n_xt, ms_xt), (n_xh, ms_xh) = solver.exchange_time()
    if args.warmup > 0:
        run_iterations(solver, args.warmup, **extra)
    if EVENTS_IN_TIMED:
        solver.kernel_time(enable=stride)
        solver.exchange_time()
    barrier()
    t0 = time.perf_counter()
    run_iterations(solver, args.steps, **extra)
    barrier()
    elapsed_local = time.perf_counter() - t0
    if EVENTS_IN_TIMED:
        launches, kms = solver.kernel_time(enable=False)
        (n_xt, ms_xt), (n_xh, ms_xh) = solver.exchange_time()
    elapsed = chan.allreduce_max(elapsed_local) if chan else elapsed_local
    if stub:
        n_local = int(full.n_corners / world)
    else:
        from tscm_calib_amd.problem import shard_frames
        n_local = full.n_corners if world == 1 else shard_frames(full, rank, world).n_corners
    # what every rank did, gathered on rank 0 (N > 1): the first run on real hardware has to explain itself
    mine = {"rank": rank, "corners": n_local, "us_per_step": 1e6 * elapsed_local / args.steps,
            "eval_kernel_us": 1e3 * kms / max(launches, 1), "eval_launches_timed": launches,
            "allreduce_T_us": 1e3 * ms_xt / max(n_xt, 1), "allreduce_H_us": 1e3 * ms_xh / max(n_xh, 1),
            "allreduces_timed": [n_xt, n_xh]}
    per_rank = chan.gather(mine) if chan else [mine]

    # what ONE call of the drop-in costs its caller (multi_calib.cpp:157-218: problem build + ceres::Solve): a warm
    # tscm_solve_multi / tscm_solve_mono -- create (layout + H2D of the observations) + natural solve + write-back + destroy --
    # behind the timed region, and where the creation of the bench's own solver (the first of the process) spent its time
    one_shot = create_split = None
    if rank == 0 and world == 1 and not stub:
        create_split = solver.create_timing()
        shots = []
        for _ in range(2):                                   # (the first call also faults in the pages of its freshly copied input arrays)
            q1 = full.copy().normalised()
            t1 = time.perf_counter()
            r1 = api.refinement(q1, device, **extra)[1] if full.mono else api.calibrate(q1, device, **extra)
            shots.append(time.perf_counter() - t1)
        one_shot = {"seconds": shots[1], "first_call_seconds": shots[0], "solve_seconds": r1["seconds_total"], "iterations": r1["num_iterations"] - 1,
                    "note": "warm call of tscm_solve_multi / _mono: create + H2D + natural solve + write-back + destroy (next to the bench's own resident solver)"}

    if rank == 0:
        avg_ms = kms / max(launches, 1)
        flops = n_local * FLOP_PER_CORNER
        bytes_per_corner = 16.0 + 168.0 / full.n_points
        # k_eval_gram4<KS, MULTI>: ceil(n / 56) passes per view of KS k-steps each (tscm_eval_gram4.h: g4_plan)
        if stub:
            g4_passes, g4_ks, g4_views = 1, 14, 1
        else:                                                   # the library's own pass plan (g4_plan, tscm_kernels.h)
            import ctypes as _C
            _pl = [_C.c_int(0) for _ in range(4)]
            lib.check(lib.lib().tscm_debug_gram_plan(full.n_points, *[_C.byref(x) for x in _pl]))
            g4_passes, g4_ks, g4_views = _pl[0].value, _pl[2].value, _pl[3].value
        g4_name = f"k_eval_gram4<{g4_ks},{'true' if g4_passes > 1 else 'false'}>"
        if g4_views > 1 and not (args.exec_flags & 512):       # boards of up to 32 corners: M views share a pass
            g4_name = f"k_eval_gram4p<{g4_ks},{g4_views}>"
        achieved_tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        peak = FP32_PEAK_TFLOPS if args.jacobian_fp32 else FP64_PEAK_TFLOPS
        roof = {
            "kernel": "k_eval_gram_f32" if args.jacobian_fp32 else ("k_eval_gram (16x16x4 MFMA)" if (args.exec_flags & 4) else g4_name), "bound": "mfma",
            "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tf / peak,
            "traffic": None, "launches": launches, "timed_launches": launches, "launches_timed_every": stride, "avg_launch_ms": avg_ms,
            # where the event pairs sat: "preheat" = the untimed loop in front of the W warmup steps (same launches, same data; the
            # K timed steps carry no event), "steps" = inside the timed region (rounds 1-5)
            "timed_in": "steps" if EVENTS_IN_TIMED else "preheat",
            # share of the step the dominant kernel accounts for; the iteration-0 evaluation of every solve is in the
            # timed region (and among the timed launches) but is not a step
            "share_of_step": (avg_ms * (args.steps + math.ceil(args.steps / ITERS_PER_SOLVE)) / (1e3 * elapsed)) if elapsed > 0 else 0.0,
            "event_pairs_in_timed_region": launches if EVENTS_IN_TIMED else 0,
            "iteration0_evals_in_timed_region": math.ceil(args.steps / ITERS_PER_SOLVE),
            "alg_flop_per_launch": flops, "alg_bytes_per_launch": n_local * bytes_per_corner,
            "ps_per_corner": 1e9 * avg_ms / max(n_local, 1),
            "hbm_frac_if_bandwidth_bound": (n_local * bytes_per_corner / (avg_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if avg_ms > 0 else 0.0,
            # the whole STEP against the same peak: the algorithmic flop of one LM iteration of the whole job (one fused
            # evaluation of every corner; the Schur / solve / back-substitution launches add < 1 % and are not counted)
            # over the wall time of a step on all N GPUs -- LM iterations/s follows this figure, not the kernel's `frac`
            "iteration_frac": (full.n_corners * FLOP_PER_CORNER / (elapsed / args.steps) / 1e12) / (peak * world) if elapsed > 0 else 0.0,
        }
        if not stub and args.jacobian_fp32:
            # fp32-Jacobian tier: the contraction (836 flop per corner) runs on v_mfma_f32_16x16x4, the projection and
            # residual in fp64 and the derivatives in packed fp32 on the VALU.  `peak` above is the fp32 datasheet figure
            # (what BASELINE's "fp64 vs fp32" asks to see); the floor prices the 600 VALU flop at the measured fp64 VALU
            # rate (an upper bound of their cost: part of them is fp32) and the contraction at the measured fp32 MFMA rate
            floor_ms = 1e3 * n_local * (FLOP_MFMA_PER_CORNER / (pk32.value * 1e12) + FLOP_VALU_PER_CORNER / (pk[2] * 1e12))
            roof.update(peak_measured_mfma_f32_16x16x4=pk32.value, peak_measured_valu_f64=pk[2], measured_floor_ms=floor_ms,
                        frac_of_measured_ceiling=floor_ms / avg_ms if avg_ms > 0 else 0.0)
        if not stub and not args.jacobian_fp32:
            # measured ceilings of THIS device (taken before the warmup).  fp64 MFMA and fp64 VALU share the DP pipe
            # (no overlap: tools/ubench_fp64.hip), so the kernel's floor is the SUM of its two parts at their own rates
            # the kernel contracts with v_mfma_f64_4x4x4_4b (TSCM_EXEC_GRAM_16X16: with v_mfma_f64_16x16x4): its ceiling
            pm = pk[0] if (args.exec_flags & 4) else pk[1]
            pv = pk[2]
            floor_ms = 1e3 * n_local * (FLOP_MFMA_PER_CORNER / (pm * 1e12) + FLOP_VALU_PER_CORNER / (pv * 1e12))
            roof.update(peak_measured_mfma_f64=pm, peak_measured_mfma_f64_16x16x4=pk[0], peak_measured_mfma_f64_4x4x4=pk[1],
                        peak_measured_valu_f64=pv, measured_floor_ms=floor_ms,
                        frac_of_measured_ceiling=floor_ms / avg_ms if avg_ms > 0 else 0.0)
            pmc = os.path.join(ROOT, "profiles", "pmc_eval_gram.json")
            if os.path.exists(pmc) and world == 1 and args.board.lower() == "9x6":
                try:
                    rec = json.load(open(pmc)).get(f"config{args.config}", {})
                    # the PMC passes are separate runs: their figure only describes THIS kernel if the sources are unchanged
                    if rec.get("kernel_src_sha") == _kernel_src_sha():
                        roof["traffic"] = rec.get("hbm_bytes_per_launch")
                        # what the total consists of: reads (2 x FETCH_SIZE) against the algorithmic INPUT bytes above, and
                        # the kernel's own output -- 102 doubles of Schur records per view (+ 4 KB of camera tile per workgroup)
                        roof["traffic_read"] = rec.get("read_bytes_corrected")
                        roof["traffic_written"] = (rec.get("hbm_bytes_per_launch") - rec.get("read_bytes_corrected")
                                                   if rec.get("hbm_bytes_per_launch") and rec.get("read_bytes_corrected") else None)
                        roof["alg_output_record_bytes_per_launch"] = 8.0 * 102 * full.n_views
                        roof["traffic_source"] = "profiles/pmc_eval_gram.json (same kernel sources)"
                    else:
                        roof["traffic_source"] = "none: profiles/pmc_eval_gram.json was measured on other kernel sources"
                except Exception:
                    pass
        out = {
            "metric": "LM iterations/sec at 4 cams x 10k views (joint intrinsics+extrinsics, fp64)" if args.config == 4 and not stub and args.board.lower() == "9x6"
                      else f"LM iterations/sec, BASELINE config {args.config}, {args.board} board (not the headline workload)",
            "value": args.steps / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64 (fp32 Jacobian + fp32 MFMA contraction)" if args.jacobian_fp32 else "f64",
            "data": "synthetic",
            "rccl_ranks": rccl_ranks, "exchange_ranks": exchange_ranks,
            "config": {"workload": f"BASELINE config {args.config}: {full.n_cameras} cams x "
                                   f"{full.meta.get('views_per_cam')} views/cam, {full.n_boards} frames, "
                                   f"{full.n_corners} corners ({args.board.lower()} board, sigma=0.1 px, seed {full.meta.get('seed')})"
                                   + (", all board poses constant" if args.poses_fixed else ""),
                       "iterations_per_solve": ITERS_PER_SOLVE, "parallelism": f"frames sharded over {world} GPU(s)" if exchange != "ipc" or n_dev >= world
                                      else f"frames sharded over {world} rank processes on {n_dev} GPU(s) (IPC exchange: a run of the multi-process path, not a scaling measurement)"},
            "exchange": exchange if multi else None, "n_devices": None if stub else min(world, n_dev),
            # which RCCL algorithm / protocol the operator pinned, if any (the redundant control step needs an all-reduce whose
            # result does not depend on the rank -- ring and tree sums are; the rank-divergence guard stops the solve if one ever is not)
            "rccl_env": ({k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO")} if multi and exchange == "rccl" else None),
            # (the IPC back-end between DEVICES -- fine-grained buffers, explicit peer access: ABI 6 -- has never met a second device)
            "exchange_note": ("IPC exchange across devices: correct by construction, never measured; RCCL is the production back-end"
                              if multi and exchange == "ipc" and not stub and min(world, n_dev) > 1 else None),
            "natural_solve": {"termination": natural["message"], "iterations": natural["num_iterations"] - 1,
                              "rmse_px": natural["rmse"], "seconds": natural["seconds_total"],
                              "device_seconds": natural["seconds_solve"],
                              "create_seconds_incl_H2D_of_observations": t_create,
                              # (the bench's solver is the first of its process: runtime_init carries HIP's start-up)
                              "create_split": create_split, "one_shot": one_shot},
            "roofline": roof,
        }
        if world > 1 or multi:
            # per iteration every rank runs its share of the kernels plus two sum all-reduces (T after the Schur
            # complement, H_stage after the evaluation); an all-reduce's HIP-event time includes the wait for the
            # slowest peer, so rank_compute_us = the rank's step time minus its two collectives is a lower bound of
            # the time its own kernels need
            out["corners_per_rank"] = [r["corners"] for r in per_rank]
            out["rank_compute_us"] = [r["us_per_step"] - r["allreduce_T_us"] - r["allreduce_H_us"] for r in per_rank]
            out["allreduce_ms"] = {
                "T": max(r["allreduce_T_us"] for r in per_rank) * 1e-3, "H_stage": max(r["allreduce_H_us"] for r in per_rank) * 1e-3,
                "per_step": max(r["allreduce_T_us"] + r["allreduce_H_us"] for r in per_rank) * 1e-3,
                "timed_every": stride, "note": "HIP events on the solver stream around the all-reduce (ncclAllReduce, or the IPC back-end's kernel: the wait for the slowest peer included), max over ranks of each rank's mean",
            }
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(full, local_rank)
        print(json.dumps(out), flush=True)
    if chan:
        chan.barrier()
    solver.close()
    if comm:
        comm.close()
    if chan:
        chan.close()


if __name__ == "__main__":
    main()
