#!/usr/bin/env python3
"""LM iteration rate of rigs with more than 8 cameras (reduced system factored in global memory by
k_solve_reduced_big).  Prints ONE JSON line.  usage: python tools/bench_many_cams.py [--cameras 16] [--views 2000]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tscm_calib_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cameras", type=int, default=16)
    ap.add_argument("--views", type=int, default=2000)
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    p = synth.make_problem(a.cameras, a.views, 77).normalised()
    with api.Solver(p.copy()) as s:
        s.solve(max_num_iterations=5)
    q = p.copy()
    with api.Solver(q) as s:
        t0 = time.perf_counter()
        r = s.solve(max_num_iterations=a.iterations, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        dt = time.perf_counter() - t0
    n = r["num_iterations"]
    print(json.dumps(dict(metric="lm_iterations_per_second", value=n / dt, unit="it/s", cameras=a.cameras,
                          views=int(p.n_views), corners=int(p.view_count.sum()), iterations=n, ms_per_iteration=1e3 * dt / max(n, 1),
                          final_cost=r["final_cost"])))


if __name__ == "__main__":
    main()
