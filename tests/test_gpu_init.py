"""GPU parity test of tscm_estimate_focal (TS.cpp:110-168) against the CPU oracle."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import lib, rig, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("views,cols,rows,seed", [(20, 9, 6, 20241), (300, 9, 6, 4), (40, 11, 8, 9)])
def test_estimate_focal_matches_oracle(hip_device, views, cols, rows, seed):
    p = synth.make_problem(1, views, seed, cols=cols, rows=rows, pitch=30.0 if cols == 11 else 45.0)
    V, n = p.n_views, cols * rows
    pu, pv = p.obs_u.reshape(V, n), p.obs_v.reshape(V, n)
    count = np.full(V, n, dtype=np.int32)
    count[::7] = 0
    fo, no, rc = orc.estimate_focal(pu, pv, count, cols, rows, 639.5, 539.5)
    fg, ng = rig.estimate_focal(pu, pv, count, cols, rows, 639.5, 539.5, hip_device)
    assert rc == 0 and ng == no and ng > 0
    # the circle fits are ill-conditioned (singular values span 1e6): two SVD algorithms agree to ~1e-11
    assert abs(fg - fo) < 1e-9 * fo


def test_estimate_focal_edge_cases(hip_device):
    p = synth.make_problem(1, 6, 3)
    pu, pv = p.obs_u.reshape(6, 54), p.obs_v.reshape(6, 54)
    assert rig.estimate_focal(pu, pv, np.zeros(6, dtype=np.int32), 9, 6, 639.5, 539.5, hip_device) == (0.0, 0)
    assert rig.estimate_focal(np.zeros((0, 54)), np.zeros((0, 54)), np.zeros(0, dtype=np.int32), 9, 6, 639.5, 539.5, hip_device) == (0.0, 0)
    with pytest.raises(lib.TscmError) as e:
        rig.estimate_focal(pu[:, :18], pv[:, :18], np.full(6, 18, dtype=np.int32), 3, 6, 639.5, 539.5, hip_device)
    assert e.value.code == -5
