#!/usr/bin/env python3
"""Bit-level fingerprint of the HIP solver's results on a fixed set of problems: run it with two builds of
libtscm_hip.so (before / after a change that must not move a bit -- layout changes, scheduling changes) and compare the
lines.  Per problem: iterations, accept/reject pattern, SHA-256 of the per-iteration costs and of the final parameters.

    python tools/regress_bits.py            (GPU box)
    python tools/regress_bits.py --check    ... and compare with tools/regress_bits.expected: exit code 1 and the differing lines on a mismatch"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers                                    # noqa: E402  (problem builders only; the oracle is not called)
from tscm_calib_amd import api, synth                        # noqa: E402


def fp(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()[:16]


def run(name, p, **opts):
    p = p.copy().normalised()
    s = api.refinement(p, **opts)[1] if p.mono else api.calibrate(p, **opts)
    its = s["iterations"]
    pattern = "".join("A" if it["step_is_successful"] else ("r" if it["step_is_valid"] else "x") for it in its[1:])
    costs = [it["cost"] for it in its] + [s["final_cost"]]
    line = f"{name:28s} it {s['num_iterations']:3d} {pattern:52s} costs {fp(costs)} params {fp(p.intr, p.cam_rt, p.board_rt)}"
    LINES.append(line)
    if "--check" not in sys.argv:
        print(line, flush=True)


LINES = []


def main():
    forced = dict(max_num_iterations=30, function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0,
                  min_trust_region_radius=0.0)
    run("config1 mono", synth.make_config(1))
    run("config2 mono 2000", synth.make_config(2))
    run("config2 poses fixed", synth.make_config(2, poses_fixed=True))
    run("config3 rig", synth.make_config(3))
    run("config3 forced 30", synth.make_config(3), **forced)
    run("config4", synth.make_config(4))
    run("rig 4x12", helpers.small_rig())
    run("rig 8x10", helpers.small_rig(8, 10, 11))
    run("rig 3x16 88 corners", helpers.small_rig(3, 16, 5, cols=11, rows=8))
    run("mixed visibility (1-4 cams)", helpers.mixed_visibility_rig())
    run("mixed visibility forced", helpers.mixed_visibility_rig(seed=9), **forced)
    run("rig 12 cams", helpers.small_rig(12, 8, 3))
    run("config3 fp32 jacobian", synth.make_config(3), jacobian_fp32=1)
    if "--check" in sys.argv:
        want = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "regress_bits.expected")).read().splitlines()
        bad = [(a, b) for a, b in zip(LINES, want) if a.rstrip() != b.rstrip()]
        for a, b in bad:
            print("got      " + a + "\nexpected " + b)
        print(f"regress_bits: {len(LINES) - len(bad)} of {len(want)} fingerprints as expected")
        sys.exit(1 if bad or len(LINES) != len(want) else 0)


if __name__ == "__main__":
    main()
