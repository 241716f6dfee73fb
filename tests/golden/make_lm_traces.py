#!/usr/bin/env python3
"""Regenerates tests/golden/lm_traces.json: per-iteration cost / radius traces and final
parameters of the CPU oracle (oracle/tscm_oracle.c) on two small problems.  The reference
itself cannot run here (Ceres/OpenCV absent -> "parity unpinned"); this fixture pins the
ORACLE against silent changes, and the GPU path is compared with the same numbers.
Run from the repo root:  python tests/golden/make_lm_traces.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pyoracle as orc  # noqa: E402
from tscm_calib_amd import synth  # noqa: E402

out = {}
for name, p in (("config1_mono", synth.make_config(1)), ("rig4x6", synth.make_problem(4, 6, 11))):
    q = p.copy().normalised()
    s = orc.solve(q)
    out[name] = dict(message=s["message"], num_iterations=s["num_iterations"],
                     costs=[it["cost"] for it in s["iterations"]],
                     radii=[it["trust_region_radius"] for it in s["iterations"]],
                     final_cost=s["final_cost"], intr=q.intr.tolist(), cam_rt=q.cam_rt.tolist(),
                     board_rt_head=q.board_rt[:3].tolist(), rmse=orc.rmse(q))
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lm_traces.json"), "w"), indent=1)
