#!/usr/bin/env python3
"""Benchmark of the TSCM LM hot path on MI355X.

Metric (BASELINE.json): LM iterations/sec at 4 cams x 10k views (config 4: 20,000 frames,
40,000 views, 2,160,000 corners, 9x6 board), joint intrinsics + extrinsics, fp64.

A "step" is ONE Levenberg-Marquardt iteration of the whole job: e-block elimination,
Schur complement, reduced solve, back-substitution, and the fused residual + analytic
Jacobian + Gram evaluation at the candidate point, plus the accept/reject decision.  The
timed region runs K iterations as ceil(K/10) solves of <= 10 iterations each, every solve
restarting from the same perturbed initial guess that is already resident in HBM (so each
iteration does productive LM work; the iteration-0 evaluation of every solve is inside the
timed region but not counted as a step).  Termination tests are disabled for the timed
solves (tolerances < 0) so that exactly K iterations run.

N > 1 (torchrun): frames are sharded across ranks (strong scaling: the job is fixed), one
process per GPU, the reduced camera system is all-reduced with RCCL twice per iteration.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel k_eval_gram, timed
with HIP events on the solver's own stream; `cpu_baseline` is the CPU oracle (a plain-C
port of the reference's Ceres path, 1 thread like the reference) on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tscm_calib_amd import api, synth  # noqa: E402
from tscm_calib_amd.problem import shard_frames  # noqa: E402

ITERS_PER_SOLVE = 10
# algorithmic work of one k_eval_gram launch (DESIGN.md "roofline accounting")
FLOP_PER_CORNER = 836 + 600          # Gram contraction 2P(P+1)+4P with P=19, + hand-structured geometry
BYTES_PER_CORNER = 16.0 + 168.0 / 54.0
FP64_PEAK_TFLOPS = 78.6              # MI355X FP64 vector = matrix peak (AMD datasheet; not in MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3             # MI355X FP32 vector (packed) = FP32 matrix peak, same datasheet
HBM_PEAK_GBS = 8000.0
BENCH_OPTS = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0,
                  min_trust_region_radius=0.0, check_every=ITERS_PER_SOLVE)


def run_iterations(solver, n_iter, **extra):
    """Run exactly n_iter LM iterations as solves of <= ITERS_PER_SOLVE iterations."""
    done = 0
    while done < n_iter:
        k = min(ITERS_PER_SOLVE, n_iter - done)
        s = solver.solve_resident(reset=True, max_num_iterations=k, **BENCH_OPTS, **extra)
        if s["lm_iterations"] != k:
            raise RuntimeError(f"expected {k} LM iterations, device ran {s['lm_iterations']} ({s['message']})")
        done += k
    return s


def _grid_delta(intr_gpu, intr_cpu, device):
    """Rays of a 25 x 21 grid of pixels (1280 x 1080 image) under the GPU-fitted model, projected with the CPU-fitted
    one: the largest pixel displacement, over all cameras."""
    uu, vv = np.meshgrid(np.linspace(40, 1240, 25), np.linspace(40, 1040, 21))
    px = np.stack([uu.ravel(), vv.ravel()], axis=1)
    worst = 0.0
    for a, b in zip(intr_gpu, intr_cpu):
        rays = api.unproject(a, px, device)
        ok = np.all(np.isfinite(rays), axis=1)
        d = api.project(b, rays[ok], device) - px[ok]
        worst = max(worst, float(np.max(np.hypot(d[:, 0], d[:, 1]))))
    return worst


def cpu_baseline(full_problem, device):
    """Oracle (port of the reference's Ceres DENSE_SCHUR LM, 1 thread) on the first 1/8 of the
    frames of the same workload, termination disabled, scaled to the full corner count.  The GPU
    runs the same sample with the same options: that is the "RMSE delta vs CPU" of the metric."""
    from oracle import pyoracle as orc
    frac = 8
    sub = shard_frames(full_problem, 0, frac).normalised()
    iters = 8
    opts = dict(max_num_iterations=iters, function_tolerance=-1.0, parameter_tolerance=-1.0,
                gradient_tolerance=-1.0, min_trust_region_radius=0.0)
    gsub = sub.copy().normalised()
    with api.Solver(gsub, device=device) as gs:
        g = gs.solve(**opts)
    t0 = time.time()
    s = orc.solve(sub, **opts)
    wall = time.time() - t0
    n_it = s["num_iterations"] - 1
    scale = sub.n_corners / full_problem.n_corners
    rmse_cpu = math.sqrt(2.0 * s["final_cost"] / sub.n_corners)
    return {
        "value": n_it / s["seconds_total"] * scale, "unit": "LM iterations/s", "cores": 1, "kind": "port",
        "sample": f"first 1/{frac} of the config-4 frames ({sub.n_corners} corners), {n_it} LM iterations in "
                  f"{s['seconds_total']:.1f} s (wall {wall:.1f} s), scaled by corner count to 2.16 M corners",
        "rmse_px_cpu": rmse_cpu, "rmse_px_gpu_same_sample": g["rmse"],
        "rmse_rel_delta": abs(g["rmse"] - rmse_cpu) / rmse_cpu,
        "max_rel_intrinsics_delta": float(np.max(np.abs(gsub.intr[:, :7] - sub.intr[:, :7]) / np.abs(sub.intr[:, :7]))),
        # SURVEY 8d parity procedure: pixel-space difference of the two fitted models over a 25 x 21 image grid
        "max_pixel_delta_25x21_grid": _grid_delta(gsub.intr, sub.intr, device),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json config index (4 = headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--jacobian-fp32", action="store_true",
                    help="north_star's 1e-3 tier: fp32 derivatives + fp32 MFMA contraction (default: all fp64, the headline)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # TSCM_BENCH_FORCE_DIST=1 runs the multi-process code path (gloo side channel + RCCL communicator) with a
    # single rank, so that it can be exercised on a one-GPU box
    multi = world > 1 or os.environ.get("TSCM_BENCH_FORCE_DIST") == "1"
    if multi:
        # torch is used for the CPU-side rendezvous only (gloo) and is imported BEFORE the library is first loaded
        # (api.Solver below): the torch wheel carries its own HIP / RCCL runtime, and the library then binds to the
        # copies that are already in the process.  torch.cuda is deliberately never touched -- device selection and
        # fencing go through the C ABI instead.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    full = synth.make_config(args.config)
    prob = shard_frames(full, rank, world).normalised() if world > 1 else full
    t_create = time.perf_counter()
    solver = api.Solver(prob, device=local_rank)          # H2D of the observations + layout build
    t_create = time.perf_counter() - t_create
    comm = None
    if multi:
        uid = [api.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = api.Comm(uid[0], rank, world, local_rank)
        solver.set_comm(comm)
    solver.upload_params()

    def barrier():
        # device fence (the job of torch.cuda.synchronize() in the contract) + process barrier
        from tscm_calib_amd import lib as _lib
        _lib.check(_lib.lib().tscm_device_synchronize(local_rank))
        if multi:
            dist.barrier()

    # natural solve (reference options, termination tests on): untimed, doubles as warmup
    extra = dict(jacobian_fp32=1) if args.jacobian_fp32 else {}
    natural = solver.solve_resident(reset=True, **extra)
    # warmup (untimed)
    if args.warmup > 0:
        run_iterations(solver, args.warmup, **extra)
    solver.kernel_time(enable=os.environ.get("TSCM_BENCH_NO_EVENTS") is None)
    barrier()
    t0 = time.perf_counter()
    last = run_iterations(solver, args.steps, **extra)
    barrier()
    elapsed = time.perf_counter() - t0
    launches, kms = solver.kernel_time(enable=False)
    if multi:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_local = prob.n_corners
        avg_ms = kms / max(launches, 1)
        flops = n_local * FLOP_PER_CORNER
        achieved_tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_eval_gram.json")
        if os.path.exists(pmc) and world == 1:
            try:
                traffic = json.load(open(pmc)).get(f"config{args.config}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "LM iterations/sec at 4 cams x 10k views (joint intrinsics+extrinsics, fp64)",
            "value": args.steps / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64 (fp32 Jacobian + fp32 MFMA contraction)" if args.jacobian_fp32 else "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {full.n_cameras} cams x "
                                   f"{full.meta['views_per_cam']} views/cam, {full.n_boards} frames, "
                                   f"{full.n_corners} corners (9x6 board, sigma=0.1 px, seed {full.meta['seed']})",
                       "iterations_per_solve": ITERS_PER_SOLVE, "parallelism": f"frames sharded over {world} GPU(s)"},
            "natural_solve": {"termination": natural["message"], "iterations": natural["num_iterations"] - 1,
                              "rmse_px": natural["rmse"], "seconds": natural["seconds_solve"],
                              "create_seconds_incl_H2D_of_observations": t_create},
            "roofline": {
                "kernel": "k_eval_gram_f32" if args.jacobian_fp32 else "k_eval_gram", "bound": "mfma",
                "achieved": achieved_tf, "peak": FP32_PEAK_TFLOPS if args.jacobian_fp32 else FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved_tf / (FP32_PEAK_TFLOPS if args.jacobian_fp32 else FP64_PEAK_TFLOPS),
                "traffic": None if args.jacobian_fp32 else traffic,
                "launches": launches, "avg_launch_ms": avg_ms,
                "alg_flop_per_launch": flops, "alg_bytes_per_launch": n_local * BYTES_PER_CORNER,
                "hbm_frac_if_bandwidth_bound": (n_local * BYTES_PER_CORNER / (avg_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if avg_ms > 0 else 0.0,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(full, local_rank)
        print(json.dumps(out))
    if multi:
        dist.barrier()
        solver.close()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
