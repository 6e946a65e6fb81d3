#!/usr/bin/env python3
"""One C4 shard on one GPU: where a bench step (solve_batched + RCCL gather) spends its time; wave64 vs wg256 batched kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, dist as cdist
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
ss = cdist.ShardSolver(P, device_index=0, rank=0, world=1)
ss.upload(rec, off); del rec
def med(f, n=12):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts[2:])) * 1e3
for rep in range(3):
    for name, fl in (("wave64", -1), ("wg256", 2 | 16 | 32 | 256 | 1024)):
        ss.solver.set_launch(0, fl)
        a = med(lambda: ss.solver.solve_batched(xb))
        b = med(lambda: ss.comm.gather_results(0, P, copy=False))
        c = med(lambda: ss.solve(xb, ordered=False, copy=False))
        k = min(ss.solver.time_batched_eval(xb, reps=10) for _ in range(3)) * 1e3
        print(f"P={P} {name}: solve_batched {a:.3f} ms, gather {b:.3f} ms, step {c:.3f} ms, eval kernel {k:.1f} us", flush=True)
ss.close()
