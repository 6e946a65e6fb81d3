import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
BASE = 2 | 16 | 32 | 128 | 256 | 512
sv = clc.Solver(0)
key = lambda s: (s.termination, s.num_iterations, s.num_evaluations, s.num_successful_steps, s.num_unsuccessful_steps)
for P, n_poses, K in ((24, 12, 97), (6, 20, 500), (64, 8, 200), (600, 12, 97)):
    rec, off, x0, gt = sd.sim_shard_records(21, 0, P, n_poses, K, 0.01)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    print(P, n_poses, K, "resident", sv.debug_resident())
    pr, sr = sv.solve_batched(x0)
    sv.set_launch(0, BASE | 2048 | 1024)
    pl, sl = sv.solve_batched(x0)
    bad = [k for k in range(P) if key(sr[k]) != key(sl[k])]
    print("  mismatching problems:", len(bad), bad[:40])
    for k in bad[:4]:
        print("   ", k, key(sr[k]), key(sl[k]), sr[k].initial_cost, sl[k].initial_cost, sr[k].final_cost, sl[k].final_cost)
