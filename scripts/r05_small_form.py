#!/usr/bin/env python3
"""Where does the 32-workgroup one-hop form of the cooperative kernel stop paying?  One process per setting (CLC_COOP_SMALL_MAX_PPL is
read at upload): solve time and kernel time of single problems of n observations on the form the limit selects.
usage: python scripts/r05_small_form.py <n_poses> [pts]   (prints one JSON line)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
n_poses = int(sys.argv[1]); pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
rec = clc.flatten_observations(sd.sim_fixed_count(1000, n_poses, pts, noise_sigma=0.01), False)
sv.upload(rec)
for _ in range(5):
    r = sv.solve(x0, trace_cap=0)
t = []
for _ in range(7):
    t0 = time.perf_counter()
    for _ in range(20):
        r = sv.solve(x0, trace_cap=0)
    t.append((time.perf_counter() - t0) / 20)
o = clc.default_options(); o.profile_events = 2
k = min(sv.solve(x0, o, trace_cap=0).summary.eval_kernel_ms for _ in range(7))
pi = sv.path_info()
print(json.dumps({"limit": os.environ.get("CLC_COOP_SMALL_MAX_PPL", "default"), "obs": int(rec.shape[0]), "workgroups": pi.coop_workgroups, "ppl": pi.coop_points_per_lane,
                  "solve_ms": 1e3 * float(np.median(t)), "kernel_ms": k, "passes": int(r.summary.num_evaluations), "us_per_pass": 1e3 * k / r.summary.num_evaluations}))
