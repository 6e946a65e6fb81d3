#!/usr/bin/env python3
"""Soak test of the default solve path: thousands of solves over alternating problem sizes (uploads in between,
two handles), every result compared bit for bit with the first solve of the same problem."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
probs = [clc.flatten_observations(sd.sim_fixed_count(7 + i, p, q, noise_sigma=0.01), False)
         for i, (p, q) in enumerate(((50, 110), (3, 40), (200, 500), (2000, 500), (37, 333), (600, 500)))]
svs = [clc.Solver(0), clc.Solver(0)]
first = {}
bad = 0
n = 0
t0 = time.time()
for r in range(rounds):
    for k, rec in enumerate(probs):
        sv = svs[(r + k) & 1]
        sv.upload(rec)
        for rep in range(1 + (r + k) % 4):
            res = sv.solve(x0, trace_cap=0 if rep else 64)
            key = k
            sig = (res.pose.tobytes(), res.summary.final_cost, res.summary.num_iterations, res.summary.termination)
            if key not in first: first[key] = sig
            elif first[key] != sig:
                bad += 1
                print("MISMATCH round", r, "problem", k, "rep", rep, res.summary.num_iterations, res.summary.termination, flush=True)
            n += 1
print(f"{n} solves in {time.time()-t0:.1f} s, mismatches: {bad}")
