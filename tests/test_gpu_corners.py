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


def test_find_chessboard_matches_the_oracle_pipeline(hip_device):
    img, uv = _scene(3, 2)
    pts = corners.find_chessboard(img, 9, 6, device=hip_device)
    assert pts is not None and pts.shape == (54, 2)
    o = orc.detect_corners(img)
    keep = o["score"] >= 0.01
    ob = orc.chessboards_from_corners(o["x"][keep], o["y"][keep], o["v1"][keep], o["v2"][keep])
    assert len(ob) == 1
    assert np.allclose(pts, o["sub"][keep][ob[0].ravel()], rtol=0, atol=1e-9)
    assert min(np.abs(pts - uv).max(), np.abs(pts[::-1] - uv).max()) < 0.3
    assert corners.find_chessboard(img, 8, 6, device=hip_device) is None            # main.cpp:33: wrong size -> image skipped
    assert corners.find_chessboard(np.full((200, 300), 90, dtype=np.uint8), 9, 6, device=hip_device) is None


def test_mono_calibration_from_rendered_images(hip_device):
    """monocular_calib (main.cpp:8-57) from pixels: render -> corner candidates (GPU) -> board -> focal, poses (GPU) ->
    LM refinement (GPU); the calibration reproduces the camera that rendered the images."""
    from tscm_calib_amd import api, rig
    from tscm_calib_amd.problem import Problem
    p = synth.make_problem(1, 10, 21, noise_px=0.0, perturb=False)
    intr_gt = p.meta["gt_intr"][0]
    V, n = p.n_views, 54
    pu, pv, count = np.zeros((V, n)), np.zeros((V, n)), np.zeros(V, dtype=np.int32)
    for k in range(V):
        img = synth.render_chessboard(intr_gt, p.meta["gt_board_rt"][k], 9, 6, 45.0, 1280, 1080, supersample=2)
        pts = corners.find_chessboard(img, 9, 6, device=hip_device)
        if pts is None:
            continue
        pu[k], pv[k], count[k] = pts[:, 0], pts[:, 1], n
    assert (count > 0).sum() >= V - 3                      # steep or border-cut boards are skipped, like main.cpp:33-37 does
    W = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
    intr = np.array([0.0, 0.0, 1280 / 2 - 0.5, 1080 / 2 - 0.5, 0.0, 0.0, 0.5, 0.0, 0.0])
    focal, used = rig.estimate_focal(pu, pv, count, 9, 6, intr[2], intr[3], hip_device)
    assert used > 0
    intr[0] = intr[1] = focal
    Rt, kk = rig.estimate_extrinsic(intr, pu, pv, count, W, 9, hip_device)
    rt = rig.poses_from_Rt(Rt)
    sel = np.flatnonzero(count > 0)
    off = np.arange(sel.shape[0], dtype=np.int32) * n
    q = Problem(1, sel.shape[0], p.board_xy, np.zeros(sel.shape[0], dtype=np.int32), np.arange(sel.shape[0], dtype=np.int32), off,
                np.full(sel.shape[0], n, dtype=np.int32), pu[sel].ravel(), pv[sel].ravel(), np.zeros((1, 6)), intr[None, :].copy(), rt[sel],
                p.cam_pose_constant, True).normalised()
    ok, s = api.refinement(q, hip_device)
    assert s["rmse"] < 0.4                                    # sub-pixel fit on the sigma = 4 blurred, distorted pattern: a few tenths of a pixel
    # the projection of the calibrated model agrees with the true camera over the image area the boards covered
    rays = np.stack([[np.cos(a) * np.sin(t), np.sin(a) * np.sin(t), np.cos(t)] for a in (0.0, 2.0, 4.0) for t in (0.2, 0.5, 0.8)])
    uv_gt = np.stack([orc.project(intr_gt, r) for r in rays])
    uv_fit = np.stack([orc.project(q.intr[0], r) for r in rays])
    # (measured: rmse 0.27 px, fx 432.3 vs 431.3, principal point within 0.35 px, projections within 0.65 px)
    assert abs(q.intr[0][0] / intr_gt[0] - 1) < 0.01 and np.abs(q.intr[0][2:4] - intr_gt[2:4]).max() < 1.0
    assert np.abs(uv_gt - uv_fit).max() < 2.0, np.abs(uv_gt - uv_fit).max()


def test_cpp_mirror_find_corner_on_a_pgm(hip_device, tmp_path):
    """examples/find_corners_demo.cpp: tscm::findCorner of the C++ mirror on a PGM file, same corners as the Python path."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "tscm_calib_amd", "csrc")
    exe = str(tmp_path / "find_corners_demo")
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "find_corners_demo.cpp"),
                           "-L", csrc, "-ltscm_hip", "-Wl,-rpath," + csrc, "-o", exe])
    img, _ = _scene(3, 0)
    with open(tmp_path / "board.pgm", "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())
    out = subprocess.check_output([exe, str(tmp_path / "board.pgm")]).decode().splitlines()
    assert out[0].startswith("candidates 54 boards 1")
    got = np.array([[float(t) for t in line.split()] for line in out[1:]])
    want = corners.find_chessboard(img, 9, 6, device=hip_device)
    assert got.shape == (54, 2) and np.allclose(got, want, atol=2e-6)


def test_second_pass_on_the_remapped_chessboard(hip_device):
    """The refinement pass of monocular_calib (main.cpp:59-105): with the board pose known, undistort_chessboard
    (TS.cpp:308-330: table + cv::remap) turns the view into a fronto-parallel chessboard, findCorner runs on it, and the
    refined corners go back through [r1 r2 t] and project().  On the rendering they land closer to the truth."""
    from tscm_calib_amd import maps
    p = synth.make_problem(1, 6, 3, noise_px=0.0, perturb=False)
    intr = p.meta["gt_intr"][0]
    k = 0
    rt = p.meta["gt_board_rt"][k]
    img, uv = _scene(3, k)
    first = corners.find_chessboard(img, 9, 6, device=hip_device)
    assert first is not None
    R = synth.rodrigues(rt[:3])
    Rt = np.stack([R[:, 0], R[:, 1], rt[3:]], axis=1)                       # TS.cpp:316: [r1 r2 t]
    desc = maps.chessboard_desc(intr, Rt, 9, 6, 45.0)
    mx, my, _ = maps.build_maps([desc], 450 * 315, hip_device)
    board_img = maps.remap(img, mx.reshape(315, 450), my.reshape(315, 450), device=hip_device)
    assert np.array_equal(board_img, orc.remap(img, mx.reshape(315, 450), my.reshape(315, 450)))
    second = corners.find_chessboard(board_img, 9, 6, device=hip_device)
    assert second is not None
    # fronto-parallel image: the inner corners sit on the 45-pixel grid starting at (45, 45) (main.cpp:99-100)
    grid = np.stack(np.meshgrid(np.arange(9), np.arange(6)), axis=-1).reshape(-1, 2) * 45.0 + 45.0
    fwd, rev = np.abs(second - grid).max(), np.abs(second[::-1] - grid).max()
    assert min(fwd, rev) < 0.6
    if rev < fwd:
        second = second[::-1]
    # back to the image: P = Rt (x - 45, y - 45, 1), project (main.cpp:99-102)
    P = (Rt @ np.concatenate([second - 45.0, np.ones((54, 1))], axis=1).T).T
    back = np.stack([orc.project(intr, q) for q in P])
    e1 = min(np.abs(first - uv).max(), np.abs(first[::-1] - uv).max())
    e2 = np.abs(back - uv).max()
    assert e2 < 0.25 and e2 <= e1 + 0.05, (e1, e2)


def test_gpu_reproduces_the_committed_fixture(hip_device):
    from tests.test_corners_oracle import _golden
    g, img = _golden()
    d = corners.detect_corners(img, min_score=-1.0, device=hip_device)
    assert d["n_maxima"] == g["n_maxima"]
    assert np.array_equal(d["x"], g["x"]) and np.array_equal(d["y"], g["y"])
    assert np.allclose(d["v1"], g["v1"], atol=1e-15) and np.allclose(d["v2"], g["v2"], atol=1e-15)
    assert np.allclose(d["score"], g["score"], rtol=1e-10, atol=1e-14) and np.allclose(d["sub"], g["sub"], atol=1e-9)
    keep = d["score"] >= 0.01
    boards = corners.chessboards_from_corners(d["x"][keep], d["y"][keep], d["v1"][keep], d["v2"][keep])
    assert [b.tolist() for b in boards] == g["boards"]


def test_batch_equals_single_calls(hip_device):
    rng = np.random.default_rng(5)
    full = [_scene(3, v)[0] for v in (0, 1, 2)]
    imgs = [np.ascontiguousarray(f[300:300 + 420, 380:380 + 560]) for f in full]
    imgs.append(np.full((420, 560), 99, dtype=np.uint8))                                       # nothing to find
    imgs.append(np.clip(imgs[0].astype(int) + rng.integers(-9, 10, size=imgs[0].shape), 0, 255).astype(np.uint8))
    batch = corners.detect_corners_batch(imgs, min_score=-1.0, device=hip_device)
    assert len(batch) == len(imgs)
    for g, im in zip(batch, imgs):
        s = corners.detect_corners(im, min_score=-1.0, device=hip_device)
        assert g["n"] == s["n"] and g["n_maxima"] == s["n_maxima"]
        for k in ("x", "y", "v1", "v2", "score", "sub"):
            assert np.array_equal(g[k], s[k]), k
    assert batch[3]["n"] == 0 and batch[0]["n"] > 0
    assert corners.detect_corners_batch([], device=hip_device) == []


@pytest.mark.parametrize("sigma", [2, 6])
def test_other_sigmas_use_the_generic_column_pass(hip_device, sigma):
    full, _ = _scene(3, 0)
    img = np.ascontiguousarray(full[200:200 + 500, 300:300 + 700])
    g = corners.detect_corners(img, sigma=sigma, min_score=-1.0, device=hip_device)
    o = orc.detect_corners(img, sigma=sigma)
    _compare(g, o, min_score=-1.0)
