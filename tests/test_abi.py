"""CPU tests of the drop-in boundary: the C-ABI shared library loads without a GPU and exports
every symbol include/tscm/tscm.h declares; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tscm_calib_amd import api, lib, synth
from tscm_calib_amd.problem import shard_frames
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tscm", "tscm.h")).read()
    declared = set(re.findall(r"\b(tscm_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    L = lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.tscm_abi_version() == 6


def test_struct_layouts_match_header():
    # sizes the C side reports through a round trip of default options / a summary-sized buffer
    o = lib.default_options(False)
    assert (o.max_num_iterations, o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (50, 1e-6, 1e-10, 1e-8)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_trust_region_radius) == (1e4, 1e16, 1e-32)
    assert (o.min_relative_decrease, o.min_lm_diagonal, o.max_lm_diagonal) == (1e-3, 1e-6, 1e32)
    assert (o.max_num_consecutive_invalid_steps, o.jacobi_scaling, o.check_every) == (5, 1, 4)
    assert lib.default_options(True).max_num_iterations == 100       # TS.cpp:274
    # ABI 6: the struct says how long it is (the library reads only that much and refuses a size it does not know)
    assert o.struct_size == C.sizeof(lib.COptions) == 112
    assert C.sizeof(lib.CIteration) == 72 and C.sizeof(lib.CProblem) == 120        # ABI 2: + board_pose_constant


def test_no_gpu_means_loud_failure_not_fallback():
    if lib.lib().tscm_device_count() > 0:
        pytest.skip("a HIP device is present")
    p = synth.make_config(1)
    with pytest.raises(lib.TscmError) as e:
        api.Solver(p)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(lib.TscmError):
        api.project(synth.CALIB_INTR[0], np.zeros((1, 3)))
    with pytest.raises(lib.TscmError):
        api.refinement(p)


def test_argument_validation_happens_before_device_use():
    p = H.small_rig(4, 4, seed=1)
    bad = p.copy().normalised()
    bad.view_board = bad.view_board.copy()
    bad.view_board[0] = 10 ** 6
    cp = lib.c_problem(bad)
    h = C.c_void_p()
    assert lib.lib().tscm_solver_create(C.byref(cp), 0, C.byref(h)) == -1
    assert b"view_board" in lib.lib().tscm_last_error()
    big = synth.make_problem(4, 2, 3)
    big.n_cameras = 33                       # 9..32 cameras are served by k_solve_reduced_big
    cp = lib.c_problem(big)
    assert lib.lib().tscm_solver_create(C.byref(cp), 0, C.byref(h)) in (-1, -5)


def test_shard_frames_matches_c_abi_and_partitions_everything():
    p = synth.make_problem(4, 50, 77)
    for world in (1, 2, 3, 8):
        owner = api.shard_owner(p, world)
        assert owner.min() == 0 and owner.max() == world - 1
        assert np.all(np.diff(owner) >= 0)                       # contiguous frame ranges
        seen = np.zeros(p.n_views, dtype=int)
        corners = []
        for r in range(world):
            q = shard_frames(p, r, world)
            assert np.array_equal(np.nonzero(owner == r)[0], q.meta["owned_boards"]) if world > 1 else True
            assert set(np.unique(q.view_board)) <= set(np.nonzero(owner == r)[0]) if world > 1 else True
            corners.append(q.n_corners)
            # observations of the shard are exactly the selected views' observations
            sel = owner[p.view_board] == r if world > 1 else np.ones(p.n_views, bool)
            seen += sel
            idx = np.concatenate([np.arange(o, o + c) for o, c in zip(p.view_offset[sel], p.view_count[sel])])
            assert np.array_equal(q.obs_u, p.obs_u[idx]) and np.array_equal(q.obs_v, p.obs_v[idx])
        assert np.all(seen == 1) and sum(corners) == p.n_corners
        assert max(corners) - min(corners) <= 2 * 54 * 2          # balanced to within a frame or two


def test_synthetic_generator_is_deterministic_and_valid():
    a, b = synth.make_config(1), synth.make_config(1)
    assert np.array_equal(a.obs_u, b.obs_u) and np.array_equal(a.board_rt, b.board_rt)
    p = synth.make_problem(4, 20, 5, noise_px=0.0, perturb=False)
    assert p.n_boards == 40 and p.n_views == 80 and p.n_corners == 80 * 54
    assert p.obs_u.min() > 0 and p.obs_u.max() < synth.IMG_W and p.obs_v.min() > 0 and p.obs_v.max() < synth.IMG_H
    # board point order of main.cpp:12-18
    assert np.array_equal(p.board_xy[:3], [[0, 0], [45, 0], [90, 0]]) and np.array_equal(p.board_xy[9], [0, 45])
    # frame f is seen by cameras f%C and (f+1)%C (adjacency, multi_calib.cpp:36)
    assert np.array_equal(p.view_camera[:4], [0, 1, 1, 2])
    p8 = synth.make_problem(8, 4, 6)
    assert p8.n_cameras == 8 and p8.cam_pose_constant[0] == 1


@pytest.mark.parametrize("src", ["dropin_demo.cpp", "multicalib_demo.cpp", "calibrate_from_corners.cpp", "find_corners_demo.cpp", "calibrate_from_images.cpp"])
def test_cpp_hosts_compile_and_link_against_the_abi(tmp_path, src):
    """The C++11 hosts (the reference's language) -- the raw C ABI one and the class mirror
    include/tscm/tscm_calib.hpp -- build with plain g++ against libtscm_hip.so."""
    import subprocess
    csrc = os.path.join(ROOT, "tscm_calib_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src),
                           "-L", csrc, "-ltscm_hip", "-Wl,-rpath," + csrc, "-o", str(tmp_path / "a.out")])
