"""bench.py on the GPU: the multi-process entry point (launcher -> rank process -> TCP side channel -> RCCL communicator)
with ONE rank, which is all a one-GPU box can run (RCCL refuses two ranks on a device), and the plain single-process
line next to it.  Config 3 keeps it short."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(env_extra, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "TSCM_RDZV_PORT", "TSCM_BENCH_FORCE_DIST", "TSCM_BENCH_STUB")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, BENCH, "--config", "3", "--steps", "20", "--warmup", "10", "--no-cpu-baseline", *args],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_spawned_single_rank_matches_the_plain_run(hip_device):
    plain = _run({})
    dist = _run({"TSCM_BENCH_FORCE_DIST": "1"}, "--gpus", "1")
    assert plain["n_gpus"] == 1 and dist["n_gpus"] == 1
    assert dist["rccl_ranks"] == 1 and dist["exchange_ranks"] == 1      # ncclCommCount of the communicator the rank created
    assert plain["rccl_ranks"] == 0 and plain["exchange"] is None      # no communicator, no RCCL in the exchange
    assert 0 < plain["roofline"]["iteration_frac"] < plain["roofline"]["frac"]
    for d in (plain, dist):
        r = d["roofline"]
        assert r["launches"] > 0 and r["avg_launch_ms"] > 0
        assert r["peak_measured_mfma_f64"] > 5 and r["peak_measured_valu_f64"] > 5      # TFLOP/s, measured in the same process
        assert 0 < r["frac"] < 1 and 0 < r["frac_of_measured_ceiling"] < 1.5
        assert d["natural_solve"]["termination"] in ("Function tolerance reached.", "Parameter tolerance reached.", "Gradient tolerance reached.")
    # same solve, same decisions
    assert plain["natural_solve"]["iterations"] == dist["natural_solve"]["iterations"]
    assert abs(plain["natural_solve"]["rmse_px"] - dist["natural_solve"]["rmse_px"]) < 1e-12
    # the communicator path (separate control kernel + all-reduce of one rank) costs a few microseconds per iteration
    assert dist["value"] > 0.6 * plain["value"]


def test_more_ranks_than_gpus_fails_loudly(hip_device):
    from tscm_calib_amd import lib
    n = lib.lib().tscm_device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, BENCH, "--config", "3", "--gpus", str(n + 1), "--steps", "5", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "needs" in out.stderr and "GPUs" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_two_rank_processes_on_one_device_over_the_ipc_exchange(hip_device):
    """`bench.py --gpus 2` end to end -- launcher, side channel, communicator, barriers, max over ranks, rank 0's JSON line -- with
    both rank processes on THIS device: TSCM_BENCH_EXCHANGE=ipc selects the library's IPC exchange back-end (RCCL refuses two
    ranks on one device).  Same natural solve as the plain run; the line says what it is."""
    plain = _run({})
    two = _run({"TSCM_BENCH_EXCHANGE": "ipc"}, "--gpus", "2")
    assert two["n_gpus"] == 2 and two["exchange"] == "ipc" and two["exchange_ranks"] == 2
    assert two["rccl_ranks"] == 0                                   # RCCL carried nothing in this run: the line must not say it saw 2 ranks
    assert "rank processes on" in two["config"]["parallelism"]
    assert len(two["per_rank"]) == 2 and len(two["corners_per_rank"]) == 2
    assert two["natural_solve"]["iterations"] == plain["natural_solve"]["iterations"]
    assert abs(two["natural_solve"]["rmse_px"] - plain["natural_solve"]["rmse_px"]) < 1e-9
    assert two["value"] > 0


def test_ipc_rank_processes_reproduce_the_local_group_bit_for_bit(hip_device):
    """tools/ipc_check.py: 2 and 4 rank PROCESSES on this device (rendezvous, handle exchange, hipIpcOpenMemHandle, arrival
    flags between kernels of different processes, the board gather through the communicator) against the same shards in one
    process through the LOCAL group: iteration log and final parameters, bit for bit."""
    tool = os.path.join(os.path.dirname(BENCH), "tools", "ipc_check.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, tool, "--world", "2,4", "--config", "3", "--iterations", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [l["world"] for l in lines] == [2, 4]
    for l in lines:
        assert l["identical"] and l["iterations"] == 10
        assert set(l["fingerprint_ipc_ranks"]) == {l["fingerprint_local_group"]}


def test_the_drivers_torchrun_command_with_two_ranks_on_this_device(hip_device):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`
    -- the command the driver uses for N > 1 -- with both ranks on this device over the IPC exchange: one JSON line, from rank 0."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TSCM_BENCH_EXCHANGE"] = "ipc"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), BENCH, "--gpus", "2", "--config", "3", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["exchange"] == "ipc" and len(d["per_rank"]) == 2
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6


def test_a_rank_that_is_gone_is_reported_not_waited_for(hip_device):
    """Failure detection of the IPC exchange: one of two rank processes leaves before the solve; the other's exchange kernel gives
    up after its bound (10 s) and the solve returns TSCM_E_PEER (ABI 6: its own code, RCCL had no part in it) -- the exchanges enqueued
    behind the failed one return at once, and the communicator is unusable afterwards."""
    tool = os.path.join(os.path.dirname(BENCH), "tools", "ipc_check.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, tool, "--world", "2", "--config", "1", "--iterations", "10", "--die-rank", "1"], env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout + out.stderr)[-3000:]
    assert lines[0]["peer_failure_detected"] and lines[0]["code"] == -7 and 5 < lines[0]["seconds"] < 60
    assert lines[0]["again_code"] == -7 and lines[0]["again_seconds"] < 2.0


def test_rank_processes_that_receive_different_bits_stop_in_the_same_step(hip_device):
    """The rank-divergence guard across PROCESSES (IPC back-end, 3 ranks on this device): rank 1's received copy of the
    Schur-complement tiles is moved by one unit in the last place at LM iteration 3; every rank's solve must end with
    TSCM_E_PEER "ranks disagree ... at iteration 3" (not a hang, not three different answers), and the next solve on the
    dead communicator fails at once."""
    tool = os.path.join(os.path.dirname(BENCH), "tools", "ipc_check.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, tool, "--world", "3", "--config", "3", "--iterations", "10", "--perturb-rank", "1", "--perturb-at", "3"],
                         env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout + out.stderr)[-3000:]
    ranks = lines[0]["ranks"]
    assert len(ranks) == 3
    for r in ranks:
        assert r["disagreement_detected"] and r["code"] == -7 and "at iteration 3" in r["message"], r
        assert r["again_code"] == -7 and r["seconds"] < 30
