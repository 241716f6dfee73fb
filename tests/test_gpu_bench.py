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
    assert dist["rccl_ranks"] == 1                                  # ncclCommCount of the communicator the rank created
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
