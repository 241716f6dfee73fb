#!/usr/bin/env python3
"""The frame-sharded solve as N rank PROCESSES on ONE device, through the library's IPC exchange back-end
(tscm_comm_ipc_open / _connect), checked bit for bit against the same N shards solved in ONE process through the LOCAL
group (tscm_comm_create_local) -- which the GPU tests pin to the unsharded solve and to the oracle.

    python tools/ipc_check.py [--world 2,4,8] [--config 3] [--iterations 12]      (GPU box, via gpurun)

The launcher spawns the rank processes BEFORE anything in it touches the GPU (a process that has initialised the GPU must
not start others on this pool); rank 0 of every group runs the LOCAL reference after its own IPC solve and prints one
JSON line per world size.  What this exercises that the in-process group cannot: the rendezvous between processes, the
handle exchange, hipIpcOpenMemHandle, the arrival flags between kernels of DIFFERENT processes, tscm_solver_gather_boards
through the communicator, and per-rank timing of an exchange that is not a kernel of the same stream.  What it cannot:
xGMI -- all ranks share the device."""
import argparse
import json
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fingerprint(summary, arrays):
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for it in summary["iterations"]:
        h.update(np.asarray([it["cost"], it["cost_change"], it["gradient_max_norm"], it["step_norm"], it["trust_region_radius"], it["relative_decrease"]], dtype=np.float64).tobytes())
        h.update(bytes([it["step_is_successful"] & 1]))
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()[:16]


def rank_main(args):
    import numpy as np
    from tscm_calib_amd import api, synth
    from tscm_calib_amd.rendezvous import SideChannel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    chan = SideChannel(rank, world)
    full = synth.make_config(args.config)
    opts = dict(max_num_iterations=args.iterations, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0) if args.iterations else {}
    solver = api.Solver(full, device=0, rank=rank, world=world)
    comm = api.Comm.ipc(rank, world, 0, chan.allgather_bytes, n_cameras=full.n_cameras)
    solver.set_comm(comm)
    chan.barrier()
    if args.die_rank >= 0:
        # failure detection: one rank leaves before the solve; the others must come back with TSCM_E_PEER within the
        # exchange kernel's time bound instead of waiting for its flags for ever
        from tscm_calib_amd.lib import TscmError
        if rank == args.die_rank:
            os._exit(0)
        t0 = time.perf_counter()
        try:
            solver.solve(**opts)
            print(json.dumps({"rank": rank, "peer_failure_detected": False}), flush=True)
            sys.exit(4)
        except TscmError as e:
            t1 = time.perf_counter()
            try:                                   # the communicator is unusable from here on: the next solve fails at once, it does not wait again
                solver.solve(**opts)
                again = 0
            except TscmError as e2:
                again = e2.code
            print(json.dumps({"rank": rank, "peer_failure_detected": True, "code": e.code, "seconds": t1 - t0, "message": str(e)[:120],
                              "again_code": again, "again_seconds": time.perf_counter() - t1}), flush=True)
        os._exit(0)                                # (no collective clean-up with a peer that is gone)
    if args.perturb_rank >= 0:
        # rank-divergence guard: ONE rank's received copy of T is moved by one unit in the last place at iteration --perturb-at;
        # EVERY rank must come back with TSCM_E_PEER "ranks disagree" from that very iteration, and the communicator is unusable afterwards
        from tscm_calib_amd.lib import TscmError
        if rank == args.perturb_rank:
            solver.debug_perturb_exchange(args.perturb_at, 1)
        t0 = time.perf_counter()
        try:
            solver.solve(**opts)
            rec = {"rank": rank, "disagreement_detected": False}
        except TscmError as e:
            t1 = time.perf_counter()
            try:
                solver.solve(**opts)
                again = 0
            except TscmError as e2:
                again = e2.code
            rec = {"rank": rank, "disagreement_detected": True, "code": e.code, "seconds": t1 - t0, "message": str(e)[:160], "again_code": again}
        recs = chan.gather(rec)
        if rank == 0:
            print(json.dumps({"world": world, "perturbed_rank": args.perturb_rank, "at_iteration": args.perturb_at, "ranks": recs}), flush=True)
        chan.barrier()
        os._exit(0)                                # (the communicator is dead: no collective clean-up)
    t0 = time.perf_counter()
    s = solver.solve(**opts)                       # in/out through full.cam_rt / intr / board_rt; gathers the boards over the communicator
    wall = time.perf_counter() - t0
    mine = fingerprint(s, [full.cam_rt, full.intr, full.board_rt])
    prints = chan.gather(mine)
    walls = chan.gather(wall)
    dev = chan.gather(s["seconds_solve"])
    solver.close()
    comm.close()
    chan.barrier()
    if rank == 0:
        ref = synth.make_config(args.config)
        with api.Group(ref, world, 0) as g:
            rs = g.solve(**opts)
        local = fingerprint(rs[0], [ref.cam_rt, ref.intr, ref.board_rt])
        ok = all(p == local for p in prints)
        print(json.dumps({"config": args.config, "world": world, "iterations": s["num_iterations"] - 1, "termination": s["message"],
                          "final_cost": s["final_cost"], "fingerprint_ipc_ranks": prints, "fingerprint_local_group": local,
                          "identical": ok, "device_us_per_iteration_ipc": [1e6 * d / max(1, s["lm_iterations"]) for d in dev],
                          "device_us_per_iteration_local_group": 1e6 * rs[0]["seconds_solve"] / max(1, rs[0]["lm_iterations"]),
                          "wall_seconds_ipc": max(walls), "exchange_buffer": getattr(comm, "note", "")}), flush=True)
        if not ok:
            sys.exit(3)
    chan.close()


def launch(world, argv):
    from bench import _free_port
    port, run = _free_port(), uuid.uuid4().hex
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSCM_RDZV_RUN=run,
                   TSCM_IPC_CHECK_RANK="1")
        env.pop("TSCM_RDZV_PORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = time.time() + 600
    pending = list(procs)
    while pending and time.time() < deadline:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0:
                rc = rc or code
                for q in pending:
                    q.terminate()
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            p.kill()
            rc = rc or 124
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="2,4,8")
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--iterations", type=int, default=12, help="forced LM iterations (0: the natural solve)")
    ap.add_argument("--die-rank", type=int, default=-1, help="this rank exits before the solve: the others must report TSCM_E_PEER, not hang")
    ap.add_argument("--perturb-rank", type=int, default=-1, help="rank-divergence guard: this rank's received T is moved by one ulp ...")
    ap.add_argument("--perturb-at", type=int, default=2, help="... at this LM iteration: every rank must report TSCM_E_PEER 'ranks disagree'")
    args = ap.parse_args()
    if os.environ.get("TSCM_IPC_CHECK_RANK") == "1":
        return rank_main(args)
    rc = 0
    for w in [int(x) for x in args.world.split(",")]:
        rc = rc or launch(w, ["--config", str(args.config), "--iterations", str(args.iterations), "--world", str(w), "--die-rank", str(args.die_rank),
                                "--perturb-rank", str(args.perturb_rank), "--perturb-at", str(args.perturb_at)])
    sys.exit(rc)


if __name__ == "__main__":
    main()
