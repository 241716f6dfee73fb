"""GPU parity tests of tscm_detect_corners (DetectCorner/findCorner.cpp:7-66, :492-541) against the CPU oracle,
through the C ABI, on synthetic Triple Sphere renderings of a chessboard."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import corners, lib, synth
from tests.test_corners_oracle import _scene

pytestmark = pytest.mark.gpu


def _compare(g, o, min_score=0.01):
    keep = ~(o["score"] < min_score)
    assert g["n_maxima"] == o["n"] and g["n"] == int(keep.sum())
    # same maxima in the same order (the metric plane is bit-identical, so this is exact)
    assert np.array_equal(g["x"], o["x"][keep]) and np.array_equal(g["y"], o["y"][keep])
    # directions come from the same 32-bin table: exact
    assert np.array_equal(g["v1"], o["v1"][keep]) and np.array_equal(g["v2"], o["v2"][keep])
    # scores: wave-parallel sums instead of sequential ones; sub-pixel fit: same order of operations
    assert np.allclose(g["score"], o["score"][keep], rtol=1e-10, atol=1e-14)
    assert np.allclose(g["sub"], o["sub"][keep], rtol=0, atol=1e-9)


@pytest.mark.parametrize("seed,view", [(3, 0), (3, 2), (8, 1), (11, 4)])
def test_candidates_match_the_oracle(hip_device, seed, view):
    img, uv = _scene(seed, view)
    g = corners.detect_corners(img, device=hip_device)
    o = orc.detect_corners(img)
    _compare(g, o)
    assert g["n"] == uv.shape[0] and g["seconds"] > 0
    dist = np.sqrt(((uv[:, None, :] - g["sub"][None, :, :]) ** 2).sum(-1))
    assert np.all(dist.min(axis=1) < 0.25)


def test_all_maxima_without_the_score_filter(hip_device):
    img, _ = _scene(3, 1)
    g = corners.detect_corners(img, min_score=-1.0, device=hip_device)
    o = orc.detect_corners(img)
    assert g["n"] == g["n_maxima"] == o["n"]
    _compare(g, o, min_score=-1.0)


@pytest.mark.parametrize("w,h", [(333, 247), (64, 48), (20, 20), (1, 1), (1921, 130)])
def test_odd_sizes_strides_and_noise(hip_device, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    full, uv = _scene(3, 0)
    x0, y0 = max(0, int(uv[:, 0].min()) - 40), max(0, int(uv[:, 1].min()) - 40)
    big = np.zeros((h, w + 7), dtype=np.uint8)                          # row stride > width
    crop = full[y0:y0 + h, x0:x0 + w]
    big[:crop.shape[0], :crop.shape[1]] = crop
    big[:, :w] = np.clip(big[:, :w].astype(int) + rng.integers(-6, 7, size=(h, w)), 0, 255).astype(np.uint8)
    view = big[:, :w]
    g = corners.detect_corners(view, min_score=-1.0, device=hip_device)
    o = orc.detect_corners(np.ascontiguousarray(view))
    _compare(g, o, min_score=-1.0)


def test_argument_checks(hip_device):
    with pytest.raises(ValueError):
        corners.detect_corners(np.zeros((4, 4, 3), dtype=np.uint8))
    with pytest.raises(lib.TscmError) as e:
        corners.detect_corners(np.zeros((32, 32), dtype=np.uint8), sigma=3)
    assert e.value.code == -5
    flat = corners.detect_corners(np.full((64, 64), 77, dtype=np.uint8), device=hip_device)
    assert flat["n"] == 0 and flat["n_maxima"] == 0
