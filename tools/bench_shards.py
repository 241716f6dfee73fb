#!/usr/bin/env python3
"""What ONE rank of an N-GPU run computes per LM iteration, measured on a single GPU: the N shards of the job run in one
process in lock step (api.Group: tscm_comm_create_local + tscm_solver_solve_group), all on one stream, so the time of an
iteration of the group divided by N is a rank's compute time -- every kernel at 1/N of the boards plus the replicated
reduced solve and control -- WITHOUT the two RCCL all-reduces (here: two small summing kernels).  It bounds the strong
scaling of DESIGN.md section 6 from below; the collectives and xGMI come on top.

    python tools/bench_shards.py [--config 4] [--worlds 1,2,4,8] [--iters 50]
Under rocprofv3 (tools/prof_tool.sh) the kernel stats give the per-shard kernel durations."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tscm_calib_amd import api, synth  # noqa: E402

OPTS = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, min_trust_region_radius=0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true", help="(accepted for tools/prof_tool.sh)")
    args = ap.parse_args()
    p = synth.make_config(args.config).normalised()
    out = {"config": args.config, "corners": p.n_corners, "iters": args.iters, "per_world": {}}
    for w in [int(x) for x in args.worlds.split(",")]:
        with api.Group(p, w) as g:
            for s in g.solvers:
                s.upload_params()
            g.solve_resident(reset=True, max_num_iterations=10, check_every=10, **OPTS)          # warm-up
            t0 = time.perf_counter()
            r = g.solve_resident(reset=True, max_num_iterations=args.iters, check_every=args.iters, **OPTS)
            dt = time.perf_counter() - t0
        assert r[0]["lm_iterations"] == args.iters, r[0]["message"]
        out["per_world"][w] = {"group_us_per_iteration": 1e6 * dt / args.iters, "rank_us_per_iteration": 1e6 * dt / args.iters / w,
                               "final_cost": r[0]["final_cost"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
