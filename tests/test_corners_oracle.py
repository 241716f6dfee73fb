"""CPU tests of the corner-candidate oracle (oracle/tscm_oracle_corners.c; DetectCorner/findCorner.cpp).
The reference holds no images or expected corner lists, and OpenCV is not available to generate any, so the
restatement is pinned by what the detector is for: on synthetic fisheye images of a chessboard rendered through
the Triple Sphere model, the candidates that survive the reference's score filter are exactly the inner corners,
at their projected positions."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import synth


def _scene(seed, view, width=1280, height=1080, cols=9, rows=6):
    p = synth.make_problem(1, 6, seed, noise_px=0.0, perturb=False, cols=cols, rows=rows, pitch=45.0 if cols == 9 else 360.0 / max(cols, rows))
    intr, rt = p.meta["gt_intr"][0], p.meta["gt_board_rt"][view]
    n = cols * rows
    o = p.view_offset[view]
    uv = np.stack([p.obs_u[o:o + n], p.obs_v[o:o + n]], axis=1)
    pitch = 45.0 if cols == 9 else 360.0 / max(cols, rows)
    return synth.render_chessboard(intr, rt, cols, rows, pitch, width, height), uv


@pytest.mark.parametrize("seed,view", [(3, 0), (3, 2), (8, 1)])
def test_kept_candidates_are_the_inner_corners(seed, view):
    img, uv = _scene(seed, view)
    d = orc.detect_corners(img)
    keep = d["score"] >= 0.01
    assert d["n"] > keep.sum() == uv.shape[0]                        # spurious maxima exist and are all rejected by the score
    sub = d["sub"][keep]
    dist = np.sqrt(((uv[:, None, :] - sub[None, :, :]) ** 2).sum(-1))
    assert np.all(dist.min(axis=1) < 0.25) and len(set(dist.argmin(axis=1))) == uv.shape[0]
    assert np.median(dist.min(axis=1)) < 0.1
    # integer maxima within a pixel of the truth, directions are unit vectors about 90 degrees apart in the image centre
    raw = np.stack([d["x"][keep], d["y"][keep]], axis=1)
    assert np.all(np.abs(raw - sub).max(axis=1) <= 2.0)
    n1, n2 = np.linalg.norm(d["v1"][keep], axis=1), np.linalg.norm(d["v2"][keep], axis=1)
    assert np.allclose(n1, 1.0) and np.allclose(n2, 1.0)


def test_planes_and_building_blocks():
    full, uv = _scene(3, 0)
    x0, y0 = int(uv[:, 0].min()) - 60, int(uv[:, 1].min()) - 50
    img = np.ascontiguousarray(full[y0:y0 + 240, x0:x0 + 320])      # a window on the board
    d = orc.detect_corners(img, planes=True)
    m = d["metric"]
    assert m.shape == (240, 320) and m.min() == 0.0 and m.max() > 0.07 and d["n"] > 0
    # the suppression only returns local maxima of the metric above tau, away from the border
    for x, y in zip(d["x"].astype(int), d["y"].astype(int)):
        assert 9 <= x < 320 - 9 and 9 <= y < 240 - 9
        assert m[y, x] >= 0.07 and m[y, x] == m[y - 4:y + 4, x - 4:x + 4].max()
    # a constant image has no corners (0 / 0 normalisation -> NaN metric, nothing passes the threshold)
    flat = np.full((64, 64), 128, dtype=np.uint8)
    assert orc.detect_corners(flat)["n"] == 0
    # Gaussian taps: symmetric, normalised, 29 of them for sigma = 4
    import ctypes as C
    k = np.zeros(29)
    orc.lib().orc_gaussian_kernel(4, k.ctypes.data_as(C.POINTER(C.c_double)))
    assert abs(k.sum() - 1.0) < 1e-15 and np.array_equal(k, k[::-1]) and k.argmax() == 14
    # the 6 x 25 least-squares operator reproduces a quadratic exactly
    X = np.zeros(150)
    orc.lib().orc_subpixel_operator(X.ctypes.data_as(C.POINTER(C.c_double)))
    X = X.reshape(6, 25)
    xs, ys = np.meshgrid(np.arange(-2, 3), np.arange(-2, 3), indexing="ij")          # row index = (x + 2) * 5 + y + 2
    coef = np.array([0.3, -0.2, 0.5, 0.1, 0.05, 2.0])
    patch = (coef[0] * xs * xs + coef[1] * ys * ys + coef[2] * xs + coef[3] * ys + coef[4] * xs * ys + coef[5]).ravel()
    assert np.allclose(X @ patch, coef, atol=1e-12)


def test_odd_sigma_is_refused():
    img = np.zeros((64, 64), dtype=np.uint8)
    with pytest.raises(RuntimeError):
        orc.detect_corners(img, sigma=3)


def _golden():
    import base64
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "corners_small.json")))
    img = np.frombuffer(base64.b64decode(g["image_b64"]), dtype=np.uint8).reshape(g["height"], g["width"]).copy()
    return g, img


def test_oracle_reproduces_the_committed_fixture():
    """tests/golden/corners_small.json (made by make_corners_golden.py): maxima, directions, boards exact; scores and
    sub-pixel positions to the last digits (libm differences between machines stay below 1e-12)."""
    g, img = _golden()
    d = orc.detect_corners(img)
    assert d["n"] == g["n_maxima"]
    assert np.array_equal(d["x"], g["x"]) and np.array_equal(d["y"], g["y"])
    assert np.allclose(d["v1"], g["v1"], atol=1e-15) and np.allclose(d["v2"], g["v2"], atol=1e-15)
    assert np.allclose(d["score"], g["score"], rtol=1e-12, atol=1e-16) and np.allclose(d["sub"], g["sub"], atol=1e-11)
    keep = d["score"] >= 0.01
    boards = orc.chessboards_from_corners(d["x"][keep], d["y"][keep], d["v1"][keep], d["v2"][keep])
    assert [b.tolist() for b in boards] == g["boards"]
    truth = np.array(g["truth"])
    det = d["sub"][keep][np.array(g["boards"][0]).ravel()]
    assert min(np.abs(det - truth).max(), np.abs(det[::-1] - truth).max()) < 0.2
