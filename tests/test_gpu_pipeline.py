"""main.cpp end to end from pixels: rendered images of a two-camera rig -> monocular_calib per camera (detection,
calibration, refinement pass on the remapped chessboards, flip rule) -> MultiCalib constructor + calibrate -> YAML.
Every numeric step runs through the C ABI on the GPU."""
import numpy as np
import pytest

from tscm_calib_amd import calib_io, pipeline, synth

pytestmark = pytest.mark.gpu


def _render_rig(seed, n_frames):
    """Two fisheye cameras 0.6 rad apart with a 160 mm baseline, boards in the shared field of view."""
    rng = np.random.default_rng(seed)
    intr = synth.make_problem(2, 4, seed, noise_px=0.0, perturb=False).meta["gt_intr"]
    cam = np.array([[0, 0, 0, 0, 0, 0], [0.0, -0.6, 0.0, -160.0, 5.0, 30.0]])          # P_cam = R P_world + t
    boards = []
    for _ in range(n_frames):
        tilt = rng.uniform(-0.35, 0.35, size=3) + np.array([0.0, 0.3, 0.0])             # roughly facing both cameras
        centre = np.array([rng.uniform(80, 260), rng.uniform(-90, 90), rng.uniform(380, 560)])
        Rb = synth.rodrigues(tilt)
        tb = centre - Rb @ np.array([4 * 45.0, 2.5 * 45.0, 0.0])                        # board centre -> origin corner
        boards.append((Rb, tb))
    images = [[None] * n_frames for _ in range(2)]
    for m in range(2):
        Rc = synth.rodrigues(cam[m, :3])
        for b, (Rb, tb) in enumerate(boards):
            images[m][b] = synth.render_chessboard(intr[m], (Rc @ Rb, Rc @ tb + cam[m, 3:]), 9, 6, 45.0, 1280, 1080, supersample=1)
    return intr, cam, images


def test_two_camera_rig_from_rendered_images(hip_device, tmp_path):
    gt_intr, gt, images = _render_rig(31, 10)
    out = pipeline.calibrate_rig(images, 9, 6, 45.0, device=hip_device)
    mono, prob, s = out["mono"], out["problem"], out["summary"]
    for m in range(2):
        assert mono[m]["has"].sum() >= 7                                   # steep / cut boards are skipped (main.cpp:33-37)
        assert mono[m]["second"]["rmse"] < 0.45
        assert mono[m]["first"]["rmse"] < 0.45
    assert s["termination_type"] == 0 and s["rmse"] < 0.5
    # the second camera's pose relative to the first: within a few mrad / a few mm of the rig that rendered the images
    assert np.max(np.abs(prob.cam_rt[1, :3] - gt[1, :3])) < 8e-3, prob.cam_rt[1] - gt[1]
    assert np.max(np.abs(prob.cam_rt[1, 3:] - gt[1, 3:])) < 6.0, prob.cam_rt[1] - gt[1]
    # intrinsics: fx alone slides along the fx / xi / lambda / alpha valley with ten boards in one part of the image;
    # what the data determine is the projection over the covered field of view
    from tscm_calib_amd import api
    rays = np.array([[np.cos(a) * np.sin(t), np.sin(a) * np.sin(t), np.cos(t)] for a in np.linspace(0, 2 * np.pi, 7)[:-1] for t in (0.15, 0.35, 0.55)])
    for m in range(2):
        d = np.abs(api.project(prob.intr[m], rays, hip_device) - api.project(gt_intr[m], rays, hip_device)).max()
        assert d < 3.0, (m, d)
    # result file in the reference's format
    R = synth.rodrigues(prob.cam_rt[:, :3])
    calib_io.write_calib_yaml(str(tmp_path / "calib.yaml"), prob.intr, np.transpose(R, (0, 2, 1)), -np.einsum("cji,cj->ci", R, prob.cam_rt[:, 3:]))
    intr, Twc = calib_io.read_calib_yaml(str(tmp_path / "calib.yaml"))
    assert intr.shape == (2, 9) and np.allclose(Twc[0], np.eye(3, 4), atol=1e-12)
