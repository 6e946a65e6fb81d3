#!/usr/bin/env python3
"""Problems one workgroup holds (the reference's C1 size, 1e4 observations) on the single-workgroup kernel (default) and on 32 co-resident
workgroups (clc_set_small_on_coop).  usage: python scripts/r05_small_on_coop.py [0|1]
Round 5, MI355X: C1 0.1325 -> 0.1156 ms per solve (5.30 -> 4.62 us per pass), 20 x 500 observations 0.1354 -> 0.1074 ms."""
import sys, time, json, os
sys.path.insert(0, '.')
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
if len(sys.argv) > 1:
    sv.set_small_on_coop(int(sys.argv[1]) != 0)  # problems one workgroup holds run on 32 co-resident workgroups first
for name, S in (("c1", sd.GenerateSimData(1, noise_sigma=0.01)), ("20x500", sd.sim_fixed_count(3, 20, 500, noise_sigma=0.01)), ("100x2 offline", sd.GenerateSimData(2, n_poses=100, noise_sigma=0.01))):
    rec = clc.flatten_observations(S, False)
    if name.startswith("100x2"): rec = clc.flatten_observations(S, True)
    sv.upload(rec)
    for _ in range(20): r = sv.solve(x0, trace_cap=0)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(100): r = sv.solve(x0, trace_cap=0)
        ts.append((time.perf_counter() - t0) / 100)
    pi = sv.path_info()
    print(json.dumps({"auto_paths": int(sys.argv[1]) if len(sys.argv) > 1 else 0, "problem": name, "obs": int(rec.shape[0]), "single": pi.single_resident, "coop": pi.coop_resident, "wgs": pi.coop_workgroups,
                      "coop_solves": pi.coop_solves, "ms": 1e3 * float(np.median(ts)), "passes": int(r.summary.num_evaluations), "us_per_pass": 1e6 * float(np.median(ts)) / r.summary.num_evaluations}))
