"""Run-to-run bit stability of the in-launch hand-offs (riding reductions -> control step, T tiles -> reduced solve -> back-substitution):
thousands of resident solves of forced iterations per BASELINE config must give ONE log (a lost ordering or a stale read would show as a
second one).  GPU box: python tools/stress_handoffs.py"""
import sys
sys.path.insert(0, ".")
from tscm_calib_amd import api, synth
opts = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0, min_trust_region_radius=0.0, check_every=255)
for cfg, n, iters in ((4, 400, 12), (3, 1000, 25), (1, 2000, 25), (2, 1000, 25)):
    p = synth.make_config(cfg).normalised()
    with api.Solver(p) as s:
        s.upload_params()
        logs = {}
        for k in range(n):
            r = s.solve_resident(reset=True, max_num_iterations=iters, **opts)
            key = (tuple((i["cost"], i["step_is_successful"], i["trust_region_radius"], i["gradient_max_norm"]) for i in r["iterations"]), r["final_cost"])
            logs[key] = logs.get(key, 0) + 1
        rej = sum(1 for i in r["iterations"] if not i["step_is_successful"])
        print(f"config {cfg}: {n} resident solves of {iters} forced iterations ({rej} rejected steps each): {len(logs)} distinct log(s); reruns {s.reruns()}", flush=True)
        assert len(logs) == 1 and s.reruns() == 0
print("ok")
