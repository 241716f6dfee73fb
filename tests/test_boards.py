"""Chessboard structure recovery (tscm_chessboards_from_corners, DetectCorner/chessboard.cpp): the library's host
implementation against the CPU oracle on random candidate sets (jittered, rotated, perspective-warped grids with
clutter), and on the candidates of rendered images.  Host logic: runs without a GPU."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import corners, synth


def _random_candidates(seed):
    rng = np.random.default_rng(seed)
    gw, gh = int(rng.integers(3, 12)), int(rng.integers(3, 9))
    ang, step = rng.uniform(0, 2 * np.pi), rng.uniform(18, 60)
    c, r = np.meshgrid(np.arange(gw), np.arange(gh))
    gx, gy = step * (c * np.cos(ang) - r * np.sin(ang)), step * (c * np.sin(ang) + r * np.cos(ang))
    k = rng.uniform(0, 4e-4)                                           # mild barrel-like warp
    rr = gx * gx + gy * gy
    gx, gy = gx * (1 - k * np.sqrt(rr)), gy * (1 - k * np.sqrt(rr))
    pts = np.stack([gx.ravel() + 400, gy.ravel() + 400], axis=1) + rng.uniform(-0.8, 0.8, size=(gw * gh, 2))
    pts = np.rint(pts)                                                 # the candidates are integer maxima
    v1 = np.tile([np.cos(ang), np.sin(ang)], (gw * gh, 1))
    v2 = np.tile([-np.sin(ang), np.cos(ang)], (gw * gh, 1))
    nc = int(rng.integers(0, 15))
    cl = np.rint(rng.uniform(0, 900, size=(nc, 2)))
    a = rng.uniform(0, 2 * np.pi, size=nc)
    pts, v1, v2 = np.concatenate([pts, cl]), np.concatenate([v1, np.stack([np.cos(a), np.sin(a)], 1)]), np.concatenate([v2, np.stack([-np.sin(a), np.cos(a)], 1)])
    perm = rng.permutation(pts.shape[0])
    return pts[perm, 0].copy(), pts[perm, 1].copy(), v1[perm].copy(), v2[perm].copy(), (gh, gw)


@pytest.mark.parametrize("seed", range(60))
def test_library_matches_oracle_on_random_candidate_sets(seed):
    x, y, v1, v2, _ = _random_candidates(seed)
    a = orc.chessboards_from_corners(x, y, v1, v2)
    b = corners.chessboards_from_corners(x, y, v1, v2)
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_clean_grids_are_recovered_completely():
    found = 0
    for seed in range(100, 130):
        x, y, v1, v2, (gh, gw) = _random_candidates(seed)
        b = corners.chessboards_from_corners(x, y, v1, v2)
        if len(b) == 1 and sorted(b[0].shape) == sorted((gh, gw)):
            found += 1
            assert b[0].shape[1] >= b[0].shape[0]
            idx = b[0]
            # neighbours along a row are one grid step apart, rows are ordered consistently
            d = np.hypot(np.diff(x[idx], axis=1), np.diff(y[idx], axis=1))
            assert d.std() / d.mean() < 0.2
    assert found >= 24          # candidate 0 is the reference's "empty" marker: grids that contain it cannot be completed


def test_degenerate_inputs():
    assert corners.chessboards_from_corners([], [], np.zeros((0, 2)), np.zeros((0, 2))) == []
    x = np.arange(8.0)
    assert corners.chessboards_from_corners(x, x, np.tile([1.0, 0.0], (8, 1)), np.tile([0.0, 1.0], (8, 1))) == []     # fewer than 9
    x = np.full(20, 7.0)
    a = corners.chessboards_from_corners(x, x, np.tile([1.0, 0.0], (20, 1)), np.tile([0.0, 1.0], (20, 1)))
    assert a == orc.chessboards_from_corners(x, x, np.tile([1.0, 0.0], (20, 1)), np.tile([0.0, 1.0], (20, 1))) == []


@pytest.mark.parametrize("view", [0, 1, 2])
def test_boards_of_rendered_images(view):
    p = synth.make_problem(1, 6, 3, noise_px=0.0, perturb=False)
    img = synth.render_chessboard(p.meta["gt_intr"][0], p.meta["gt_board_rt"][view], 9, 6, 45.0, 1280, 1080, supersample=2)
    d = orc.detect_corners(img)
    keep = d["score"] >= 0.01
    b = corners.chessboards_from_corners(d["x"][keep], d["y"][keep], d["v1"][keep], d["v2"][keep])
    assert len(b) == 1 and b[0].shape == (6, 9)
    o = p.view_offset[view]
    uv = np.stack([p.obs_u[o:o + 54], p.obs_v[o:o + 54]], axis=1)
    det = d["sub"][keep][b[0].ravel()]
    # the board comes out in one of its two 180-degree orderings
    err = min(np.abs(det - uv).max(), np.abs(det[::-1] - uv).max())
    assert err < 0.3
