#!/usr/bin/env python3
"""C3 (BASELINE.json configs[2]): batch of independent T_cl problems on one MI355X.
python scripts/bench_batched.py [n_problems] [n_poses] [pts]   (default 1024 x 20 x 500 = 10k obs each)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_poses = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pts = int(sys.argv[3]) if len(sys.argv) > 3 else 500
t = time.time()
probs, gts = sd.sim_batch(4242, P, n_poses, pts, noise_sigma=0.01)
recs = [clc.flatten_observations(p, False) for p in probs]
off = np.zeros(P + 1, dtype=np.int64); off[1:] = np.cumsum([r.shape[0] for r in recs])
allrec = np.concatenate(recs)
print(f"generated {P} problems x {recs[0].shape[0]} obs in {time.time()-t:.1f}s ({allrec.nbytes/1e6:.0f} MB)", flush=True)
# start: the ground truth perturbed by (5 cm, ~3 deg)
x0 = np.stack([sd.pose7_from_T(g) for g in gts])
rng = np.random.default_rng(1)
sv = clc.Solver(0)
import ctypes as C
from camlasercalibratool_amd import _capi
d = rng.normal(size=(P, 6)) * np.array([0.05, 0.05, 0.05, 0.05, 0.05, 0.05])
x0 = sv.pose_plus(x0, d)
sv.upload_batched(allrec, off)
best = None
import itertools
# 18 = prefetch + compact, 82 = + deep pipeline, -1 = library default (deep chosen by size); repeated: the first config after an upload runs cold
for flags, grid, si in [(18, 0, 2), (82, 0, 2), (18, 0, 2), (82, 0, 2), (6, 0, 2), (-1, 0, 2)]:
    sv.set_launch(grid, flags)
    o = clc.default_options(); o.launch_ahead = si
    times = []
    for rep in range(12):
        t = time.perf_counter(); poses, sms = sv.solve_batched(x0, o); times.append(time.perf_counter() - t)
    for rep in range(1):
        dt = float(np.median(times))
        evals = sum(sms[k].num_evaluations * (off[k+1]-off[k]) for k in range(P))
        iters = [sms[k].num_iterations for k in range(P)]
        line = dict(flags=flags, grid=grid, launch_ahead=si, ms=dt*1e3, evals_per_s=evals/dt, problems_per_s=P/dt, iters_min=min(iters), iters_max=max(iters), iters_mean=float(np.mean(iters)),
                    GBps=64*evals/dt/1e9)
        if best is None or line["evals_per_s"] > best["evals_per_s"]: best = line
    print(json.dumps(line), flush=True)
print("BEST", json.dumps(best))
terms = {}
for k in range(P): terms[sms[k].termination] = terms.get(sms[k].termination, 0) + 1
print("terminations", terms)
err = max(np.abs(sd.T_from_pose7(poses[k]) - gts[k]).max() for k in range(P))
print("max |Tcl - gt| over problems (noise 1cm):", err)
# parity on a sample vs the oracle
import oracle
worst = (0, 0)
for k in list(range(0, P, max(1, P // 16))):
    ref = oracle.solve(recs[k], x0[k], linear_solver="qr")
    dT = np.abs(sd.T_from_pose7(poses[k]) - sd.T_from_pose7(ref.pose)).max(); dc = abs(sms[k].final_cost - ref.summary.final_cost)
    assert sms[k].num_iterations == ref.summary.num_iterations, (k, sms[k].num_iterations, ref.summary.num_iterations)
    worst = (max(worst[0], dT), max(worst[1], dc))
print("parity vs oracle on 16 problems: max dT %.2e, max dcost %.2e" % worst)
