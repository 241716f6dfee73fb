"""World-size-2 CPU test (gloo) of the frame-sharded LM step.

No GPU here, so the per-shard normal equations come from the oracle; what is tested is the
distributed PROTOCOL the HIP solver uses with RCCL: frames sharded so that every board's Schur
block is rank-local, ONE sum-all-reduce of the per-camera tiles {F^T F, F^T r, r^T r} after the
evaluation and ONE of the Schur complement sum_b Y_b^T Y_b, then a redundant reduced solve on
every rank and a purely local back-substitution.  The sharded step must equal the unsharded one.
"""
import os
import socket

import numpy as np
import pytest

from tscm_calib_amd import synth
from tscm_calib_amd.problem import shard_frames
from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _local_schur(ne, p, radius, owned):
    """Damped (no Jacobi scaling) Schur pieces of one rank: returns S contribution, rhs
    contribution and a closure for the local back-substitution."""
    C = p.n_cameras
    nf = 15 * C
    T = np.zeros((nf, nf))
    tr = np.zeros(nf)
    fac = {}
    for b in owned:
        views = np.nonzero((p.view_board == b) & (p.view_count > 0))[0]
        if views.size == 0:
            continue
        V = ne["board_gram"][b].copy()
        V[np.diag_indices(6)] += np.clip(np.diag(V), 1e-6, 1e32) / radius
        L = np.linalg.cholesky(V)
        W = np.zeros((6, nf))
        for v in views:
            m = int(p.view_camera[v])
            W[:, 15 * m:15 * m + 15] = ne["view_cross"][v]
        Y = np.linalg.solve(L, W)
        z = np.linalg.solve(L, ne["board_grad"][b])
        T += Y.T @ Y
        tr += Y.T @ z
        fac[b] = (L, Y, z)
    return T, tr, fac


def _reduced_solve(p, cam_gram, cam_grad, T, tr, radius):
    C = p.n_cameras
    nf = 15 * C
    Hc = np.zeros((nf, nf))
    g = np.zeros(nf)
    for m in range(C):
        Hc[15 * m:15 * m + 15, 15 * m:15 * m + 15] = cam_gram[m]
        g[15 * m:15 * m + 15] = cam_grad[m]
    active = np.ones(nf, bool)
    for m in range(C):
        active[15 * m + 13:15 * m + 15] = False            # b, c
        if p.cam_pose_constant[m]:
            active[15 * m:15 * m + 6] = False
    A = Hc - T
    A[np.diag_indices(nf)] += np.clip(np.diag(Hc), 1e-6, 1e32) / radius
    rhs = g - tr
    idx = np.nonzero(active)[0]
    y = np.zeros(nf)
    y[idx] = np.linalg.solve(A[np.ix_(idx, idx)], rhs[idx])
    return y


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = synth.make_problem(4, 10, 31)
        radius = 1e4
        shard = shard_frames(p, rank, world).normalised()
        owned = shard.meta["owned_boards"]
        ne = H.oracle_normal_equations(shard)
        # all-reduce #1: per-camera tiles + cost (what k_finalize_eval stages in H_stage)
        buf = torch.from_numpy(np.concatenate([ne["cam_gram"].ravel(), ne["cam_grad"].ravel(), [ne["cost"]]]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        C = p.n_cameras
        cam_gram = buf[:C * 225].numpy().reshape(C, 15, 15)
        cam_grad = buf[C * 225:C * 240].numpy().reshape(C, 15)
        cost = float(buf[-1])
        # all-reduce #2: Schur complement and its rhs (what k_T_reduce leaves in T)
        T, tr, fac = _local_schur(ne, shard, radius, owned)
        buf2 = torch.from_numpy(np.concatenate([T.ravel(), tr]))
        dist.all_reduce(buf2, op=dist.ReduceOp.SUM)
        nf = 15 * C
        T, tr = buf2[:nf * nf].numpy().reshape(nf, nf), buf2[nf * nf:].numpy()
        y_c = _reduced_solve(p, cam_gram, cam_grad, T, tr, radius)
        # local back-substitution for the owned boards
        y_b = {int(b): np.linalg.solve(L.T, z - Y @ y_c) for b, (L, Y, z) in fac.items()}
        # max-all-reduce of a rank-local statistic (the gradient max-norm travels this way)
        gm = torch.tensor([max(np.abs(ne["board_grad"][owned]).max(), 0.0)], dtype=torch.float64)
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)
        # the RCCL unique id is distributed like this in bench.py
        uid = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        q.put((rank, cost, y_c, y_b, float(gm), uid[0]))
    finally:
        dist.destroy_process_group()


def test_sharded_lm_step_equals_unsharded_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    results.sort(key=lambda t: t[0])

    # unsharded reference
    p = synth.make_problem(4, 10, 31)
    radius = 1e4
    ne = H.oracle_normal_equations(p)
    T, tr, fac = _local_schur(ne, p, radius, np.arange(p.n_boards))
    y_c = _reduced_solve(p, ne["cam_gram"], ne["cam_grad"], T, tr, radius)
    y_b = {int(b): np.linalg.solve(L.T, z - Y @ y_c) for b, (L, Y, z) in fac.items()}

    seen = set()
    for rank, cost, yc_r, yb_r, gm, uid in results:
        assert abs(cost - ne["cost"]) <= 1e-12 * ne["cost"]
        assert np.max(np.abs(yc_r - y_c)) <= 1e-9 * np.max(np.abs(y_c))     # every rank solves the same system
        assert uid == bytes(range(128))
        assert abs(gm - np.abs(ne["board_grad"]).max()) < 1e-9
        for b, y in yb_r.items():
            assert b not in seen
            seen.add(b)
            assert np.max(np.abs(y - y_b[b])) <= 1e-8 * max(1.0, np.max(np.abs(y_b[b])))
    assert seen == set(y_b.keys())                                          # every board stepped exactly once
    assert np.array_equal(results[0][2], results[1][2])                     # bitwise identical on both ranks
