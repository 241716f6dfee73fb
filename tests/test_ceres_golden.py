"""Consumes tests/golden/ceres_<name>.json -- results of the REAL Ceres on the fixture problems, produced by
tools/ceres_harness where libceres-dev is installed -- and compares the CPU oracle (and, with -m gpu, the HIP path)
with them: iteration count, accept / reject pattern, per-iteration cost, final parameters (1e-6, north_star's fp64 bar).
Skipped while no such file exists: Ceres cannot be installed in the build containers, parity is UNPINNED until then."""
import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "ceres_harness"))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ceres_*.json")))


def _problem(name):
    from export_problems import FIXTURES
    return FIXTURES[name]().normalised()


def _compare(gold, summary, p):
    assert summary["num_iterations"] == gold["num_iterations"], (summary["message"], gold["message"])
    costs = [it["cost"] for it in summary["iterations"]]
    assert np.allclose(costs, gold["costs"], rtol=1e-6, atol=0)
    assert [int(it["step_is_successful"]) for it in summary["iterations"]] == [int(x) for x in gold["step_is_successful"]]
    gi, gc, gb = np.array(gold["intr"]), np.array(gold["cam_rt"]), np.array(gold["board_rt"])
    assert np.max(np.abs(p.intr[:, :7] - gi[:, :7]) / np.maximum(np.abs(gi[:, :7]), 1e-3)) < 1e-6
    if not p.mono:
        assert np.max(np.abs(p.cam_rt - gc)) <= 1e-6 * max(np.max(np.abs(gc)), 1.0)
    assert np.max(np.abs(p.board_rt - gb)) <= 1e-6 * max(np.max(np.abs(gb)), 1.0)
    rm = np.sqrt(2.0 * summary["final_cost"] / p.n_corners)
    assert abs(rm - gold["rmse"]) <= 1e-6 * gold["rmse"]


@pytest.mark.skipif(not FILES, reason="no tests/golden/ceres_*.json: run tools/ceres_harness where Ceres is installed (parity unpinned)")
@pytest.mark.parametrize("path", FILES or ["none"])
def test_oracle_against_real_ceres(path):
    from oracle import pyoracle as orc
    gold = json.load(open(path))
    p = _problem(gold["name"])
    s = orc.solve(p)
    _compare(gold, s, p)


@pytest.mark.gpu
@pytest.mark.skipif(not FILES, reason="no tests/golden/ceres_*.json: run tools/ceres_harness where Ceres is installed (parity unpinned)")
@pytest.mark.parametrize("path", FILES or ["none"])
def test_hip_path_against_real_ceres(hip_device, path):
    from tscm_calib_amd import api
    gold = json.load(open(path))
    p = _problem(gold["name"])
    with api.Solver(p) as s:
        summary = s.solve()
    _compare(gold, summary, p)


PINNED_FIXTURES = {        # sha256 of what tools/ceres_harness/export_problems.py writes (tests/golden/README.md)
    "config1_mono": "7b9d375e973957dc919f88684e6e4fd44a8bc4be0671f77abd262d685c594985",
    "rig4x6": "9e7e4945bc9897ef1acf562d38d5aa149a3e1867bcfc2215f6021b9aab2c8116",
    "rig4x30": "53b5953c90f2af9987b6d0c5ca6a6bef43e5819ca5f80e60b3727375887455e5",
    "mono40": "fd464e6cf01292fe7a317bf6854753b44ca9ddf603e684f788c930b6a3061e90",
    "config3": "3447fb5d2f624ae21026bb3320aab6218440db5701c8b17f8bf8bb35480b6a4b",
}


def test_exported_fixture_problems_are_the_pinned_ones(tmp_path):
    """A Ceres golden made on another machine is only worth something if it was made from the same inputs: the fixture
    problems are regenerated from seeds, and their bytes are pinned here and in tests/golden/README.md."""
    import hashlib
    from export_problems import FIXTURES, write_problem
    assert set(FIXTURES) == set(PINNED_FIXTURES)
    for name, make in FIXTURES.items():
        f = tmp_path / (name + ".bin")
        write_problem(make(), str(f))
        assert hashlib.sha256(f.read_bytes()).hexdigest() == PINNED_FIXTURES[name], name
