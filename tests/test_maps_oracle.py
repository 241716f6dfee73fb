"""CPU tests of the remap-table oracle (oracle/tscm_oracle_maps.c restating TS.cpp:284-330 and
EpipolarRectify/rectify.cpp:86-248) against numpy restatements and the camera model's inverse."""
import os

import numpy as np

from oracle import pyoracle as orc
from tscm_calib_amd import calib_io, maps, synth
from tests import helpers as H

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_calib.yaml")


def np_map(d):
    j, i = np.meshgrid(np.arange(d.width, dtype=np.float64), np.arange(d.height, dtype=np.float64))
    x0, y0 = (j - d.cx) / d.fx, (i - d.cy) / d.fy
    R = np.asarray(d.R)
    P = np.stack([R[r, 0] * x0 + R[r, 1] * y0 + R[r, 2] for r in range(3)], axis=-1)
    u, v = H.np_project_skew(d.intr, P)
    if d.check_w2:
        bad = P[..., 2] <= -d.w2 * np.sqrt((P * P).sum(-1))
        u, v = np.where(bad, -1.0, u), np.where(bad, -1.0, v)
    return (u + d.offset_x).astype(np.float32), (v + d.offset_y).astype(np.float32)


def _same_floats(a, b):
    """identical float32 values up to a handful of 1-ulp differences (numpy evaluates x**2, sums in another order)"""
    a, b = np.asarray(a), np.asarray(b)
    diff = a != b
    assert diff.mean() < 1e-4
    assert np.all(np.abs(a[diff] - b[diff]) <= np.spacing(np.abs(b[diff])))


def test_undistort_table_matches_numpy():
    intr = synth.CALIB_INTR[0].copy()
    intr[7:] = [0.3, -0.2]                         # skew terms are live in project()
    d = maps.undistort_desc(intr, 300.0, 310.0, 639.5, 539.5, 321, 203)
    mx, my = orc.build_maps([d], 321 * 203)
    nx, ny = np_map(d)
    _same_floats(mx.reshape(203, 321), nx)
    _same_floats(my.reshape(203, 321), ny)
    # the table is the forward model: unprojecting the table entry gives back the pinhole ray
    i, j = 57, 200
    ray = orc.unproject(intr, np.array([float(mx.reshape(203, 321)[i, j]), float(my.reshape(203, 321)[i, j])]))
    want = np.array([(j - 639.5) / 300.0, (i - 539.5) / 310.0, 1.0])
    assert np.max(np.abs(ray / ray[2] - want)) < 1e-4          # float32 table entries


def test_chessboard_table_matches_numpy_and_hits_the_corners():
    p = synth.make_problem(1, 4, 3, noise_px=0.0, perturb=False)
    rt = p.meta["gt_board_rt"][2]
    R = synth.rodrigues(rt[:3])
    Rt = np.stack([R[:, 0], R[:, 1], rt[3:]], axis=1)
    d = maps.chessboard_desc(p.intr[0], Rt, 9, 6, 45.0)
    assert (d.width, d.height) == (450, 315)
    mx, my = orc.build_maps([d], 450 * 315)
    nx, ny = np_map(d)
    _same_floats(mx.reshape(315, 450), nx)
    _same_floats(my.reshape(315, 450), ny)
    # table entry (i, j) = (45 + y, 45 + x) is the image of board point (x, y, 0): the observed corner
    obs_u, obs_v = p.obs_u[2 * 54:3 * 54], p.obs_v[2 * 54:3 * 54]
    for c in (0, 8, 30, 53):
        x, y = p.board_xy[c]
        assert abs(mx.reshape(315, 450)[int(45 + y), int(45 + x)] - obs_u[c]) < 1e-3
        assert abs(my.reshape(315, 450)[int(45 + y), int(45 + x)] - obs_v[c]) < 1e-3


def test_rectification_tables_of_the_reference_calibration():
    intr, Twc = calib_io.read_calib_yaml(GOLDEN)
    descs, n = maps.rectify_descs(intr, Twc)
    assert len(descs) == 8 and n == 8 * 400 * 400
    mx, my = orc.build_maps(descs, n)
    for d in descs:
        nx, ny = np_map(d)
        sl = slice(d.out_offset, d.out_offset + 160000)
        _same_floats(mx[sl].reshape(400, 400), nx)
        _same_floats(my[sl].reshape(400, 400), ny)
    # every table entry is either the (-1,-1)+offset sentinel or lies in the mosaic quadrant of its camera
    left_x = mx[:640000].reshape(4, 400, 400)
    assert np.all((left_x[0] >= -1.0) & (left_x[0] < 1400.0))
    # the pair rotation is a rotation whose x axis joins the two camera centres
    t = Twc[:, :, 3]
    R = orc.rectify_pair_rotation(t[0], t[1])
    assert np.max(np.abs(R - maps.rectify_pair_rotation(t[0], t[1]))) < 1e-15
    assert np.max(np.abs(R.T @ R - np.eye(3))) < 1e-14 and abs(np.linalg.det(R) - 1) < 1e-14
    assert np.max(np.abs(R[:, 0] - (t[1] - t[0]) / np.linalg.norm(t[1] - t[0]))) < 1e-15
    assert R[1, 2] == 0.0
