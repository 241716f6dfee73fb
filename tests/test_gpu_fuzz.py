"""Randomised GPU <-> oracle parity: many small rigs of random shape (1-6 cameras, ragged and empty
views, boards seen by a random subset of the cameras, unseen boards, 54-, 88-corner or random-shape boards),
three LM iterations each.  Short runs stay clear of the flat valley of the model, so the whole
iteration trace can be compared tightly."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import api, synth
from tscm_calib_amd.problem import Problem
from tests import helpers as H

pytestmark = pytest.mark.gpu


def random_rig(seed: int) -> Problem:
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(1, 7))
    shape = int(rng.integers(0, 3))
    if shape == 0:
        kw = {}
    elif shape == 1:
        kw = dict(cols=11, rows=8, pitch=30.0)
    else:                       # any board from 3x2 to 14x10 corners, about the same physical size
        cols, rows = int(rng.integers(3, 15)), int(rng.integers(2, 11))
        kw = dict(cols=cols, rows=rows, pitch=float(360.0 / max(cols, rows)))
    if C == 1:
        p = synth.make_problem(1, int(rng.integers(4, 25)), 500 + seed, **kw)
    elif rng.integers(0, 2):
        p = synth.make_problem(C if C > 1 else 2, int(rng.integers(4, 16)), 500 + seed, **kw)
    else:
        p = H.mixed_visibility_rig(seed=500 + seed, n_frames=int(rng.integers(6, 30)), n_cameras=max(C, 2), **kw)
    # ragged / empty views
    cnt = p.view_count.copy()
    k = rng.integers(0, max(1, p.n_views // 3))
    idx = rng.choice(p.n_views, size=int(k), replace=False)
    cnt[idx] = rng.integers(0, p.n_points + 1, size=idx.shape[0])
    q = p.copy()
    q.view_count[:] = cnt
    if not p.mono and rng.integers(0, 2):
        q = H.rig_with_unseen_boards(q, extra=int(rng.integers(1, 4))) if "gt_board_rt" in q.meta else q
    return q.normalised()


@pytest.mark.parametrize("seed", range(40))
def test_random_rig_three_iterations(hip_device, seed):
    p = random_rig(seed)
    pg, po = p.copy().normalised(), p.copy().normalised()
    with api.Solver(pg) as s:
        gs = s.solve(max_num_iterations=3)
    os_ = orc.solve(po, max_num_iterations=3)
    assert gs["num_iterations"] == os_["num_iterations"]
    assert gs["termination_type"] == os_["termination_type"]
    for a, b in zip(gs["iterations"], os_["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"]
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"])
        assert abs(a["gradient_max_norm"] - b["gradient_max_norm"]) <= 1e-7 * max(b["gradient_max_norm"], 1e-12)
        assert abs(a["step_norm"] - b["step_norm"]) <= 1e-7 * max(b["step_norm"], 1e-12)
        assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-7 * b["trust_region_radius"]
    d = H.param_rel_err(pg, po)
    assert max(d.values()) < 1e-7, d


def large_rig(seed):
    rng = np.random.default_rng(7000 + seed)
    C = int(rng.integers(7, 15))
    if rng.integers(0, 2):
        p = H.mixed_visibility_rig(seed=700 + seed, n_frames=int(rng.integers(2 * C, 5 * C)), n_cameras=C)
        if rng.integers(0, 2):
            p = H.rig_with_unseen_boards(p, extra=2)
    else:
        p = synth.make_problem(C, int(rng.integers(4, 10)), 700 + seed)
    cnt = p.view_count.copy()
    idx = rng.choice(p.n_views, size=max(1, p.n_views // 5), replace=False)
    cnt[idx] = rng.integers(0, p.n_points + 1, size=idx.shape[0])
    q = p.copy()
    q.view_count[:] = cnt
    return q.normalised()


@pytest.mark.parametrize("seed", range(12))
def test_random_large_rig_three_iterations(hip_device, seed):
    """7..14 cameras (register/LDS solver up to 8, k_solve_reduced_big beyond), boards seen by 1..C cameras (the
    explicit pair-list Schur path), unseen boards, ragged views."""
    q = large_rig(seed)
    pg, po = q.copy().normalised(), q.copy().normalised()
    with api.Solver(pg) as s:
        gs = s.solve(max_num_iterations=3)
    os_ = orc.solve(po, max_num_iterations=3)
    assert gs["num_iterations"] == os_["num_iterations"]
    for a, b in zip(gs["iterations"], os_["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"]
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"])
        assert abs(a["step_norm"] - b["step_norm"]) <= 1e-7 * max(b["step_norm"], 1e-12)
    d = H.param_rel_err(pg, po)
    assert max(d.values()) < 1e-7, d
