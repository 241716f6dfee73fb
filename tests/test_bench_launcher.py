"""bench.py's multi-process entry point without a GPU: the launcher (`--gpus N` spawns N rank processes), the TCP side
channel that replaces torch.distributed (ncclUniqueId broadcast, barrier, max-over-ranks) and the timing protocol, with
the solver replaced by a stub (TSCM_BENCH_STUB=1).  The real thing runs in tests/test_gpu_bench.py."""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TSCM_RDZV_PORT", "TSCM_RDZV_RUN",
                        "TSCM_BENCH_FORCE_DIST")}
    env.update(extra)
    return env


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks, one JSON line, max-over-ranks timing."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "0"],
                         env=_clean_env(TSCM_BENCH_STUB="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "strong"
    # the line says what carried the exchange: the stub has no RCCL in it (rccl_ranks = ncclCommCount, nothing else)
    assert d["exchange_ranks"] == 2 and d["rccl_ranks"] == 0 and d["exchange"] == "stub"
    # ... and what the whole step achieves against the peak next to the dominant kernel's own fraction
    assert {"frac", "iteration_frac"} <= set(d["roofline"]) and 0 < d["roofline"]["iteration_frac"] < 1
    # the N > 1 line explains itself: what each rank had, what it computed, what the two collectives cost
    assert len(d["corners_per_rank"]) == 2 and sum(d["corners_per_rank"]) > 0
    assert len(d["rank_compute_us"]) == 2 and {"T", "H_stage", "per_step", "timed_every"} <= set(d["allreduce_ms"])
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and all("eval_kernel_us" in r and "allreduce_T_us" in r for r in d["per_rank"])
    assert d["roofline"]["timed_launches"] == d["roofline"]["launches"] and d["roofline"]["iteration0_evals_in_timed_region"] == 1
    # the stub's rank 1 takes 2 ms per iteration, rank 0 takes 1 ms: the reported time is the slower rank's
    assert 1.9 <= d["ms_per_step"] < 4.0, d["ms_per_step"]
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6


def test_bench_under_a_torchrun_style_environment():
    """The driver's launch: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set by the launcher, whose own
    store owns MASTER_PORT -- the side channel must find another port (here MASTER_PORT itself is kept busy)."""
    port = _free_port()
    busy = socket.socket()
    busy.bind(("127.0.0.1", port))
    busy.listen(1)
    try:
        procs = []
        for r in range(3):
            env = _clean_env(TSCM_BENCH_STUB="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3",
                             MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "3", "--steps", "10", "--warmup", "0"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=120) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
        d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 3
        assert 2.9 <= d["ms_per_step"] < 6.0                       # rank 2: 3 ms per iteration
        assert not any(l.startswith("{") for o in outs[1:] for l in o[0].splitlines())    # only rank 0 prints
    finally:
        busy.close()


def test_bench_under_the_real_torchrun():
    """Exactly the driver's command line for N > 1: python -m torch.distributed.run ... bench.py --gpus N (torch only in the
    launcher process; the ranks themselves never import it)."""
    import pytest
    pytest.importorskip("torch")                 # only the launcher needs it: nothing this repository ships does
    for attempt in range(3):                     # (the probed port can be taken by the time torchrun's store binds it)
        port = _free_port()
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "10", "--warmup", "0"],
                             env=_clean_env(TSCM_BENCH_STUB="1"), capture_output=True, text=True, timeout=300)
        if out.returncode == 0 or "in use" not in out.stderr:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and 1.9 <= d["ms_per_step"] < 4.0


def test_single_process_default_is_untouched():
    """No --gpus, no launcher environment: one process, no side channel."""
    out = subprocess.run([sys.executable, BENCH, "--steps", "5", "--warmup", "0"], env=_clean_env(TSCM_BENCH_STUB="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and 0.9 <= d["ms_per_step"] < 2.5


def test_a_failing_rank_takes_the_job_down():
    """A rank that dies must not leave its peers (and the launcher) waiting for it forever."""
    # (the real case is a rank whose device does not exist; the stub's rank 1 exits before the rendezvous instead)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "0"],
                         env=_clean_env(TSCM_BENCH_STUB="1", TSCM_BENCH_STUB_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


# ---------------------------------------------------------------------------------------------- side channel
def _chan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("TSCM_RDZV_PORT", None)
    sys.path.insert(0, ROOT)
    from tscm_calib_amd.rendezvous import SideChannel
    ch = SideChannel(rank, world, timeout=60)
    uid = ch.bcast(bytes(range(128)) if rank == 0 else None)
    ch.barrier()
    mx = ch.allreduce_max(10.0 + rank)
    g = ch.gather({"rank": rank})
    ag = ch.allgather_bytes(bytes([rank]) * 64)          # (the IPC back-end's memory handles travel like this)
    ch.barrier()
    ch.close()
    q.put((rank, uid, mx, g, ag))


def test_side_channel_collectives_world3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chan_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, uid, mx, g, ag in res:
        assert uid == bytes(range(128)) and mx == 12.0
        assert ag == [bytes([r]) * 64 for r in range(3)]
        assert (g == [{"rank": 0}, {"rank": 1}, {"rank": 2}]) if rank == 0 else (g is None)


def test_event_stride_times_enough_launches_without_owning_the_timed_region():
    """The driver runs `--steps 20`: 21 launches of the dominant kernel.  At a fixed stride of 8 only three were timed (round 2);
    at every second launch (rounds 3-4) the eleven event pairs were 6 us per step of measurement inside the timed region (an
    event pair holds the stream for ~11 us).  Now: at least four, every fifth launch at --steps 20."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    for steps in (1, 5, 20, 50, 100, 1000):
        k = bench.event_stride(steps)
        launches = steps + -(-steps // bench.ITERS_PER_SOLVE)
        assert 1 <= k <= bench.EVENT_STRIDE_MAX
        assert launches // k >= min(launches, bench.MIN_TIMED_LAUNCHES), (steps, k)
    assert bench.event_stride(20) == 5 and bench.event_stride(1000) == 8 and bench.MIN_TIMED_LAUNCHES >= 4
    assert 1 <= bench.ITERS_PER_SOLVE <= 255
