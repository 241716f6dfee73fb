"""What ONE call of the drop-in costs its caller (multi_calib.cpp:157-218: problem build + ceres::Solve): three creates of a solver on
BASELINE config <n> (the first is the process's cold one: HIP's own start-up sits in runtime_init), each with the split of
tscm_solver_create_timing, then one warm api.calibrate = tscm_solve_multi (create + H2D + natural solve + write-back + destroy).
GPU box: python tools/one_shot.py [config]   ->  profiles/r06_one_shot.json was assembled from its output."""
import sys, time, json
sys.path.insert(0, ".")
from tscm_calib_amd import api, synth
import numpy as np
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = synth.make_config(cfg)
for k in range(3):
    t0 = time.perf_counter()
    s = api.Solver(p)
    t1 = time.perf_counter()
    ct = s.create_timing()
    s.close()
    t2 = time.perf_counter()
    print(json.dumps({"create_ms": round(1e3 * (t1 - t0), 2), "destroy_ms": round(1e3 * (t2 - t1), 2), **{k_: round(1e3 * v, 2) for k_, v in ct.items()}}))
q = p.copy().normalised()
t0 = time.perf_counter(); r = api.calibrate(q); t1 = time.perf_counter()
print("one shot calibrate ms", round(1e3 * (t1 - t0), 2), "solve s", r["seconds_total"], r["seconds_solve"])
