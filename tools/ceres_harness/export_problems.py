#!/usr/bin/env python3
"""Writes the fixture problems the Ceres harness runs (problem.bin format of examples/dropin_demo.cpp) into a directory.

    python tools/ceres_harness/export_problems.py out_dir
    for f in out_dir/*.bin; do n=$(basename $f .bin); build/ceres_harness/ceres_harness $f tests/golden/ceres_$n.json $n; done

The problems are regenerated from seeds (tscm_calib_amd/synth.py, counter-based RNG: bit-identical everywhere), so the
JSON a machine with Ceres produces can be compared here with the oracle and the HIP path on the same inputs
(tests/test_ceres_golden.py)."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tscm_calib_amd import synth  # noqa: E402

# name -> problem: the reference's own CPU-runnable case (BASELINE config 1), small rigs, config 3
FIXTURES = {
    "config1_mono": lambda: synth.make_config(1),
    "rig4x6": lambda: synth.make_problem(4, 6, 11),
    "rig4x30": lambda: synth.make_problem(4, 30, 21),
    "mono40": lambda: synth.make_problem(1, 40, 17),
    "config3": lambda: synth.make_config(3),
}


def write_problem(p, path):
    p = p.normalised()
    with open(path, "wb") as f:
        f.write(struct.pack("6i", p.n_cameras, p.n_boards, p.n_points, p.n_views, p.n_corners, int(p.mono)))
        for a in (p.board_xy, p.view_camera, p.view_board, p.view_offset, p.view_count, p.obs_u, p.obs_v,
                  p.cam_rt, p.intr, p.board_rt, p.cam_pose_constant):
            f.write(np.ascontiguousarray(a).tobytes())


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "ceres_problems"
    os.makedirs(out, exist_ok=True)
    for name, make in FIXTURES.items():
        write_problem(make(), os.path.join(out, name + ".bin"))
        print("wrote", os.path.join(out, name + ".bin"))
