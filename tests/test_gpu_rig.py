"""GPU parity tests of tscm_rig_init (MultiCalib::MultiCalib, multi_calib.cpp:6-153) against the
CPU oracle, through the C ABI."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tscm_calib_amd import api, lib, rig, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _compare(inp, g, o):
    assert o["rc"] == 0
    assert np.array_equal(g["cam_choice"], o["cam_choice"])
    assert np.array_equal(g["board_initial"], o["board_initial"])
    assert H.rel_err(g["cam_min_error"][1:], o["cam_min_error"][1:]) < 1e-12
    assert np.max(np.abs(g["cam_R"] - o["cam_R"])) < 1e-14
    assert np.max(np.abs(g["cam_t"] - o["cam_t"])) < 1e-11            # mm, |t| ~ 600
    assert np.max(np.abs(g["cam_rt"] - o["cam_rt"])) < 1e-9
    assert np.max(np.abs(g["board_R"] - o["board_R"])) < 1e-13
    assert np.max(np.abs(g["board_t"] - o["board_t"])) < 1e-10
    ang = np.linalg.norm(o["board_rt"][:, :3], axis=1)
    ok = (ang < np.pi - 1e-3) | (o["board_initial"] == 0)              # the axis sign flips at pi
    assert np.max(np.abs(g["board_rt"][ok] - o["board_rt"][ok])) < 1e-9


@pytest.mark.parametrize("C,V,seed", [(4, 12, 7), (2, 30, 3), (8, 10, 21), (4, 300, 17)])
def test_rig_init_matches_oracle(hip_device, C, V, seed):
    inp = synth.make_rig_input(synth.make_problem(C, V, seed))
    g = rig.rig_init(inp, hip_device)
    _compare(inp, g, orc.rig_init(inp))
    assert g["n_projections"] > 0 and g["seconds_hypotheses"] > 0


def test_rig_init_mixed_visibility_and_unseen_boards(hip_device):
    p = H.rig_with_unseen_boards(H.mixed_visibility_rig(seed=9), extra=3)
    inp = synth.make_rig_input(p)
    g = rig.rig_init(inp, hip_device)
    _compare(inp, g, orc.rig_init(inp))
    seen = inp.has.sum(axis=0)
    assert np.all(g["board_rt"][seen == 0] == 0.0) and not g["board_initial"][seen == 0].any()


def test_rig_init_88_corner_board(hip_device):
    inp = synth.make_rig_input(synth.make_problem(4, 16, 31, cols=11, rows=8, pitch=30.0))
    _compare(inp, rig.rig_init(inp, hip_device), orc.rig_init(inp))


def test_rig_init_rejects_reference_undefined_behaviour(hip_device):
    inp = synth.make_rig_input(synth.make_problem(4, 8, 2))
    drop = inp.has[1].astype(bool) & inp.has[2].astype(bool)
    inp.has[2, drop] = 0
    with pytest.raises(lib.TscmError) as e:
        rig.rig_init(inp, hip_device)
    assert e.value.code == -1 and "share no board" in str(e.value)


def test_rig_init_then_calibrate_matches_oracle_chain(hip_device):
    """constructor -> calibrate(): both chains (HIP, oracle) end at the same optimum."""
    p = synth.make_problem(4, 16, 13)
    inp = synth.make_rig_input(p)
    g = rig.rig_init(inp, hip_device)
    o = orc.rig_init(inp)
    pg = rig.problem_from_rig(inp, g)
    po = rig.problem_from_rig(inp, o)
    sg = api.calibrate(pg, hip_device)
    so = orc.solve(po)
    assert sg["termination_type"] == so["termination_type"] == 0
    assert abs(sg["final_cost"] - so["final_cost"]) < 1e-6 * so["final_cost"]
    assert sg["rmse"] < 0.2                                                    # 0.1 px noise
    d = H.param_rel_err(pg, po)
    assert max(d.values()) < 1e-6, d


def test_rig_init_full_size_properties(hip_device):
    """BASELINE config 4 (4 cameras x 10k views: 5000 common boards per camera pair, 2.7e9
    projections per pair): the winning hypothesis' error equals the oracle's for that hypothesis,
    and no hypothesis in an oracle-scored sample beats it."""
    p = synth.make_config(4)
    inp = synth.make_rig_input(p)
    g = rig.rig_init(inp, hip_device)
    assert g["n_projections"] >= 3 * 2 * 54 * 5000 * 5000
    rng = np.random.default_rng(0)
    for i in range(1, 4):
        Rp, tp = g["cam_R"][i - 1], g["cam_t"][i - 1]
        common = np.nonzero(inp.has[i - 1].astype(bool) & inp.has[i].astype(bool))[0]
        assert common.size == 5000
        # hypotheses (multi_calib.cpp:29-48) in numpy for the winner and a random sample
        js = np.concatenate([[g["cam_choice"][i]], rng.choice(common.size, size=12, replace=False)])
        Ri, ti = H.np_Rt_to_R_t(inp.Rt[i, common[js]])
        Rk, tk = H.np_Rt_to_R_t(inp.Rt[i - 1, common[js]])
        Rik = Ri @ np.swapaxes(Rk, 1, 2)
        Rs = Rik @ Rp
        ts = Rik @ tp + ti - np.einsum("kij,kj->ki", Rik, tk)
        err = orc.rig_hypothesis_errors(inp, i, Rp, tp, Rs, ts)
        assert abs(err[0] - g["cam_min_error"][i]) < 1e-11 * err[0]
        assert np.all(err[1:] >= err[0])
        assert np.max(np.abs(g["cam_R"][i] - Rs[0])) < 1e-13
    assert g["board_initial"].all()
    # boards: each is seen by two cameras; the chosen pose is one of the two hypotheses
    gt = p.meta["gt_board_rt"]
    assert np.median(np.abs(g["board_rt"][:, 3:] - gt[:, 3:])) < 30.0          # mm: an initial guess


def test_cpp_class_mirror_runs_the_reference_flow(hip_device, tmp_path):
    """examples/multicalib_demo.cpp: the reference's main.cpp flow (MultiCalib(cameras, worlds);
    calibrate(); YAML) written against include/tscm/tscm_calib.hpp, compared with the oracle chain."""
    import os, struct, subprocess
    from tscm_calib_amd import calib_io, maps
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "multicalib_demo")
    csrc = os.path.join(root, "tscm_calib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "multicalib_demo.cpp"),
                           "-L", csrc, "-ltscm_hip", "-Wl,-rpath," + csrc, "-o", exe])
    p = synth.make_problem(4, 16, 13)
    inp = synth.make_rig_input(p)
    C, B, n = inp.n_cameras, inp.n_boards, inp.n_points
    with open(tmp_path / "rig.bin", "wb") as f:
        f.write(struct.pack("5i", C, B, n, 9, 6))
        for a in (inp.worlds, inp.intr, inp.has, inp.Rt, inp.pix_u, inp.pix_v):
            f.write(np.ascontiguousarray(a).tobytes())
    out = subprocess.check_output([exe, str(tmp_path / "rig.bin"), str(tmp_path / "result.bin"), str(tmp_path / "calib.yaml")]).decode()
    assert "average reproject error" in out
    assert "mono calibration of camera 0 from raw corners: converged" in out
    raw = open(tmp_path / "result.bin", "rb").read()
    nd = 6 * C + 9 * C + 6 * B + C + 2
    vals = np.frombuffer(raw[:8 * nd], dtype=np.float64)
    cam_rt, intr, board_rt = vals[:6 * C].reshape(C, 6), vals[6 * C:15 * C].reshape(C, 9), vals[15 * C:15 * C + 6 * B].reshape(B, 6)
    cam_err, mean_err, focal0 = vals[15 * C + 6 * B:15 * C + 6 * B + C], vals[nd - 2], vals[nd - 1]
    term, iters = struct.unpack("2i", raw[8 * nd:8 * nd + 8])
    # oracle chain: constructor -> calibrate -> error report
    o = orc.rig_init(inp)
    po = rig.problem_from_rig(inp, o)
    so = orc.solve(po)
    assert term == so["termination_type"] == 0 and iters == so["num_iterations"]
    assert np.max(np.abs(intr[:, :7] - po.intr[:, :7]) / np.abs(po.intr[:, :7])) < 1e-6
    assert np.max(np.abs(cam_rt - po.cam_rt)) < 1e-6 * np.max(np.abs(po.cam_rt))
    assert np.max(np.abs(board_rt[:, 3:] - po.board_rt[:, 3:])) < 1e-6 * np.max(np.abs(po.board_rt[:, 3:]))
    g, per = orc.mean_reprojection_error(po)
    assert abs(mean_err - g) < 1e-6 * g and np.max(np.abs(cam_err - per)) < 1e-6 * g
    count = np.full(B, n, dtype=np.int32) * inp.has[0]
    fo, _, _ = orc.estimate_focal(inp.pix_u[0], inp.pix_v[0], count, 9, 6, 639.5, 539.5)
    assert abs(focal0 - fo) < 1e-9 * fo
    # the YAML holds the calibrated intrinsics and Twc = [R(cam_rt) | t]
    yi, yT = calib_io.read_calib_yaml(str(tmp_path / "calib.yaml"))
    assert np.array_equal(yi, intr)
    assert np.max(np.abs(yT[:, :, :3] - synth.rodrigues(cam_rt[:, :3]))) < 1e-12 and np.array_equal(yT[:, :, 3], cam_rt[:, 3:])
    # first 16 entries of camera 0's undistortion table
    mx = np.frombuffer(raw[8 * nd + 8:8 * nd + 8 + 64], dtype=np.float32)
    ox, _ = orc.build_maps([maps.undistort_desc(intr[0], 300.0, 300.0, 639.5, 539.5, 64, 48)], 64 * 48)
    assert np.array_equal(mx, ox[:16])


def test_full_pipeline_from_raw_corner_lists(hip_device):
    """main.cpp:196-319 after corner detection, with nothing but corner lists as input: per-camera mono
    calibration (focal estimate, planar PnP, refinement), rig initialisation, joint calibration.  The result
    must sit at the noise floor and reproduce the rig geometry the corners were generated from."""
    from tscm_calib_amd.problem import Problem
    p = synth.make_problem(4, 40, 77, noise_px=0.1)          # 80 frames, each seen by two adjacent cameras
    C, B, n = p.n_cameras, p.n_boards, p.n_points
    W = np.concatenate([p.board_xy, np.zeros((n, 1))], axis=1)
    has = np.zeros((C, B), dtype=np.uint8)
    pu, pv = np.zeros((C, B, n)), np.zeros((C, B, n))
    idx = p.view_offset.astype(np.int64)[:, None] + np.arange(n)[None, :]
    has[p.view_camera, p.view_board] = 1
    pu[p.view_camera, p.view_board] = p.obs_u[idx]
    pv[p.view_camera, p.view_board] = p.obs_v[idx]
    intr = np.zeros((C, 9))
    Rt = np.zeros((C, B, 3, 3))
    for m in range(C):                                        # TripleSphereCamera::calibrate per camera (TS.cpp:30-105)
        count = (has[m] * n).astype(np.int32)
        I = np.array([0.0, 0.0, 1280 / 2 - 0.5, 1080 / 2 - 0.5, 0.0, 0.0, 0.5, 0.0, 0.0])
        I[0] = I[1] = rig.estimate_focal(pu[m], pv[m], count, 9, 6, I[2], I[3], hip_device)[0]
        Rt_m, k = rig.estimate_extrinsic(I, pu[m], pv[m], count, W, 9, hip_device)
        assert k == int(has[m].sum())
        boards = np.nonzero(has[m])[0]
        V = boards.shape[0]
        q = Problem(1, V, p.board_xy, np.zeros(V, dtype=np.int32), np.arange(V, dtype=np.int32), (np.arange(V) * n).astype(np.int32),
                    np.full(V, n, dtype=np.int32), pu[m, boards].ravel().copy(), pv[m, boards].ravel().copy(),
                    np.zeros((1, 6)), I[None, :].copy(), rig.poses_from_Rt(Rt_m[boards]), np.ones(1, dtype=np.uint8), True).normalised()
        ok, s = api.refinement(q, hip_device)
        assert ok and s["rmse"] < 0.2
        intr[m] = q.intr[0]
        R = synth.rodrigues(q.board_rt[:, :3])                # TS.cpp:88-102: Rt_ from the refined poses
        Rt[m, boards] = np.stack([R[:, :, 0], R[:, :, 1], q.board_rt[:, 3:]], axis=2)
    inp = rig.RigInput(W, intr, has, Rt, pu, pv).normalised()
    g = rig.rig_init(inp, hip_device)                         # MultiCalib::MultiCalib
    pj = rig.problem_from_rig(inp, g)
    s = api.calibrate(pj, hip_device)                         # MultiCalib::calibrate
    assert s["termination_type"] == 0 and s["rmse"] < 0.16    # sigma = 0.1 px per coordinate -> 0.141
    gt = p.meta["gt_cam_rt"]
    # rig geometry: camera centres within a few mm, rotations within a few mrad of the generating rig
    assert np.max(np.abs(pj.cam_rt[:, :3] - gt[:, :3])) < 5e-3
    assert np.max(np.abs(pj.cam_rt[:, 3:] - gt[:, 3:])) < 5.0


def test_cli_from_corner_file_to_yaml(hip_device, tmp_path):
    """examples/calibrate_from_corners.cpp: corner file in, calibration YAML out -- the reference's main.cpp
    without the detector and the viewer."""
    import os, subprocess
    from tscm_calib_amd import calib_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "calibrate_from_corners")
    csrc = os.path.join(root, "tscm_calib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "calibrate_from_corners.cpp"),
                           "-L", csrc, "-ltscm_hip", "-Wl,-rpath," + csrc, "-o", exe])
    p = synth.make_problem(4, 30, 99, noise_px=0.1)
    inp = synth.make_rig_input(p)
    calib_io.write_corners(str(tmp_path / "corners.txt"), inp.has, inp.pix_u, inp.pix_v, 9, 6, 45.0)
    out = subprocess.check_output([exe, str(tmp_path / "corners.txt"), str(tmp_path / "calib.yaml")]).decode()
    assert out.count("converged") >= 4 and "NOT converged" not in out and "average reproject error" in out
    intr, Twc = calib_io.read_calib_yaml(str(tmp_path / "calib.yaml"))
    assert intr.shape == (4, 9) and np.array_equal(Twc[0], np.eye(3, 4))
    gt = p.meta["gt_cam_rt"]
    assert np.max(np.abs(Twc[:, :, 3] - gt[:, 3:])) < 6.0                                 # mm
    assert np.max(np.abs(Twc[:, :, :3] - synth.rodrigues(gt[:, :3]))) < 6e-3
    assert np.max(np.abs(intr[:, 2:4] - p.meta["gt_intr"][:, 2:4])) < 1.0                 # principal points, px


@pytest.mark.parametrize("C", [12, 20, 32])
def test_rig_init_and_calibrate_with_many_cameras(hip_device, C):
    """MultiCalib with more than 8 cameras: constructor (rig init) against the oracle, then calibrate()."""
    from tscm_calib_amd import api
    p = synth.make_problem(C, 6, 600 + C)
    inp = synth.make_rig_input(p)
    g = rig.rig_init(inp, hip_device)
    _compare(inp, g, orc.rig_init(inp))
    q = rig.problem_from_rig(inp, g)
    s = api.calibrate(q, hip_device)
    assert s["termination_type"] == 0 and s["rmse"] < 0.25


def test_cpp_class_mirror_sharded_branch(hip_device, tmp_path):
    """MultiCalib::set_sharding + the sharded branch of calibrate() (tscm_solver_create_sharded / set_comm / solve with a
    real RCCL communicator, here of one rank with TSCM_EXEC_KEEP_SINGLE_RANK_COMM so that the two all-reduces and the
    separate control step do run): compiled from the same header, the result is the plain calibrate()'s, bit for bit."""
    import os, struct, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "multicalib_demo")
    csrc = os.path.join(root, "tscm_calib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "multicalib_demo.cpp"),
                           "-L", csrc, "-ltscm_hip", "-Wl,-rpath," + csrc, "-o", exe])
    p = synth.make_problem(4, 12, 29)
    inp = synth.make_rig_input(p)
    C, B, n = inp.n_cameras, inp.n_boards, inp.n_points
    with open(tmp_path / "rig.bin", "wb") as f:
        f.write(struct.pack("5i", C, B, n, 9, 6))
        for a in (inp.worlds, inp.intr, inp.has, inp.Rt, inp.pix_u, inp.pix_v):
            f.write(np.ascontiguousarray(a).tobytes())
    res = {}
    for mode in ("plain", "sharded"):
        args = [exe, str(tmp_path / "rig.bin"), str(tmp_path / f"{mode}.bin"), str(tmp_path / f"{mode}.yaml")] + ([mode] if mode == "sharded" else [])
        out = subprocess.check_output(args).decode()
        assert "average reproject error" in out
        res[mode] = open(tmp_path / f"{mode}.bin", "rb").read()
    nd = 6 * C + 9 * C + 6 * B + C + 2
    assert res["plain"][:8 * nd + 8] == res["sharded"][:8 * nd + 8]          # parameters, error report, termination, iterations
