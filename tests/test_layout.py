"""Host-side layout of tscm_solver_create_sharded (no GPU): the device orders of views and boards are O(n) counting sorts
over small integer keys since round 6 (they were comparison sorts through lambdas: 8.8 ms of every one-shot call at config 5);
this pins them to the definition -- views camera-major and by device board, boards grouped by camera-set signature (number of
views, then the camera list), unseen boards last, ties in the caller's order -- on rigs with every kind of visibility, ragged and
empty views, and every shard of a sharded solver."""
import ctypes as C

import numpy as np
import pytest

from tscm_calib_amd import lib, synth
from tscm_calib_amd.problem import shard_frames
from tests import helpers as H


def _layout(p, rank=0, world=1):
    L = lib.lib()
    cp = lib.c_problem(p)
    n = C.c_int(0)
    b0 = C.c_int(0)
    dev2orig = np.zeros(max(p.n_views, 1), dtype=np.int32)
    perm = np.zeros(max(p.n_boards, 1), dtype=np.int32)
    ip = C.POINTER(C.c_int)
    lib.check(L.tscm_debug_layout_order(C.byref(cp), rank, world, C.byref(n), dev2orig.ctypes.data_as(ip), perm.ctypes.data_as(ip), C.byref(b0)))
    return n.value, dev2orig, perm, b0.value


def _expected(p, boards):
    """boards: the caller's boards this rank owns (a contiguous range)."""
    b0, b1 = boards[0], boards[-1] + 1
    views = [v for v in range(p.n_views) if p.view_count[v] > 0 and b0 <= p.view_board[v] < b1]
    cams = {b: sorted(int(p.view_camera[v]) for v in views if p.view_board[v] == b) for b in range(b0, b1)}
    key = lambda b: (len(cams[b]) == 0, len(cams[b]), tuple(cams[b]))
    perm = sorted(range(b0, b1), key=key)                      # (sorted is stable: ties in the caller's order)
    dev_board = {b: k for k, b in enumerate(perm)}
    order = sorted(views, key=lambda v: (int(p.view_camera[v]), dev_board[int(p.view_board[v])]))
    return order, [b - b0 for b in perm], b0


def _cases():
    yield "rig 4x12", H.small_rig()
    yield "mono", synth.make_problem(1, 30, 3)
    yield "mixed visibility", H.mixed_visibility_rig(seed=5, n_frames=40)
    yield "8 cameras", H.small_rig(8, 10, 11)
    p = H.small_rig(4, 10, seed=8)
    p.view_count[3] = 0
    p.view_count[4] = 0
    p.view_count[5] = 0
    p.view_count[7] = 31
    yield "ragged + unseen board", p
    yield "unseen boards appended", H.rig_with_unseen_boards(H.small_rig(3, 8, 2), 3)
    rng = np.random.default_rng(0)
    q = H.mixed_visibility_rig(seed=9, n_frames=64)
    shuffle = rng.permutation(q.n_views)                        # views in an arbitrary caller order
    q.view_camera, q.view_board, q.view_offset, q.view_count = (a[shuffle].copy() for a in (q.view_camera, q.view_board, q.view_offset, q.view_count))
    yield "shuffled views", q


@pytest.mark.parametrize("name,p", list(_cases()), ids=[n for n, _ in _cases()])
def test_device_orders_match_their_definition(name, p):
    p = p.copy().normalised()
    n, dev2orig, perm, b0 = _layout(p)
    order, want_perm, want_b0 = _expected(p, list(range(p.n_boards)))
    assert n == len(order) and b0 == want_b0 == 0
    assert dev2orig[:n].tolist() == order
    assert perm[:p.n_boards].tolist() == want_perm


@pytest.mark.parametrize("world", [2, 3, 8])
def test_device_orders_of_every_shard(world):
    p = H.mixed_visibility_rig(seed=5, n_frames=48).copy().normalised()
    L = lib.lib()
    owner = np.zeros(p.n_boards, dtype=np.int32)
    cp = lib.c_problem(p)
    lib.check(L.tscm_shard_frames(C.byref(cp), world, owner.ctypes.data_as(C.POINTER(C.c_int))))
    seen = 0
    for rank in range(world):
        boards = np.nonzero(owner == rank)[0].tolist()
        n, dev2orig, perm, b0 = _layout(p, rank, world)
        if not boards:
            assert n == 0
            continue
        order, want_perm, want_b0 = _expected(p, boards)
        assert b0 == want_b0 and n == len(order)
        assert dev2orig[:n].tolist() == order and perm[:len(boards)].tolist() == want_perm
        seen += n
    assert seen == int(np.count_nonzero(p.view_count > 0))


def _plan(n):
    out = [C.c_int(0) for _ in range(4)]
    lib.check(lib.lib().tscm_debug_gram_plan(n, *[C.byref(x) for x in out]))
    return tuple(x.value for x in out)          # passes, corners per pass, k-steps, views per pass


def test_pass_plan_of_the_gram_kernels():
    """g4_plan (round 6: the tuned Gram kernel serves every board size): known boards, and the invariants over 1..600 corners --
    a pass is a multiple of four corners and at most 64 rows, the passes cover the board and are balanced (no pass could be
    dropped), 64-row passes only where tile + board points leave four workgroups per CU their LDS."""
    assert _plan(54) == (1, 56, 14, 1)           # BASELINE's 9 x 6: the round-3 kernel's tile
    assert _plan(88) == (2, 44, 11, 1)           # the reference's 11 x 8 (main.cpp:190-191)
    assert _plan(64) == (1, 64, 16, 1) and _plan(63) == (1, 64, 16, 1) and _plan(57) == (1, 60, 15, 1)      # one pass of up to 64 rows
    assert _plan(48) == (1, 48, 12, 1) and _plan(42) == (1, 44, 11, 1)
    assert _plan(30) == (1, 32, 8, 2) and _plan(20) == (1, 20, 5, 3) and _plan(12) == (1, 12, 3, 4) and _plan(4) == (1, 4, 1, 4)
    assert _plan(70) == (2, 36, 9, 1) and _plan(108) == (2, 56, 14, 1) and _plan(117) == (2, 60, 15, 1)
    assert _plan(128) == (3, 44, 11, 1)          # two passes of 64 rows would not leave four workgroups per CU their LDS
    for n in range(1, 601):
        passes, per, ks, m = _plan(n)
        assert per == 4 * ks and 1 <= ks <= 16 and passes * per >= n and (passes - 1) * per < n
        assert m == (min(4, 16 // ks) if passes == 1 and ks <= 8 else 1) and m * ks <= 16
        if ks > 14:
            assert 32 * (max(68 * ks, 512) + 2 * n) <= 40 * 1024
        else:
            assert passes == -(-n // 56) or passes == -(-n // 64)
