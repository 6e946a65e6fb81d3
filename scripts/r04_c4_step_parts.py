#!/usr/bin/env python3
"""Where one C4-shard step goes (8 192 problems x 1e4 observations, one GPU): start poses into the pinned buffer, clc_solve_batched
(in place), clc_gather_results (world size 1) — wall time of each part and the kernel time (HIP events) beside them."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import dist as cdist, simdata as sd
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ss = cdist.ShardSolver(P, device_index=0, rank=0, world=1)
rec, off, x0, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
ss.upload(rec, off)
del rec
for _ in range(3):
    ss.solve(x0, ordered=False, copy=False, inplace=True)
parts = {"poses_in": [], "solve_batched": [], "gather": [], "step": []}
for _ in range(20):
    t0 = time.perf_counter()
    pb, _ = ss.solver.batched_buffers()
    pb[:] = x0
    t1 = time.perf_counter()
    ss.solver.solve_batched_inplace(None)
    t2 = time.perf_counter()
    ss.comm.gather_results(ss.lo, ss.cap, copy=False)
    t3 = time.perf_counter()
    for k, v in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t3 - t0)):
        parts[k].append(1e3 * v)
o = clc.default_options(); o.profile_events = 1
km = []
for _ in range(7):
    pb, _ = ss.solver.batched_buffers(); pb[:] = x0
    _, sm = ss.solver.solve_batched_inplace(o)
    km.append(sm[0].eval_kernel_ms)
whole = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10):
        ss.solve(x0, ordered=False, copy=False, inplace=True)
    whole.append(1e2 * (time.perf_counter() - t0))
print(json.dumps({"lib": os.path.basename(os.environ.get("CLC_LIBRARY", "default")), "problems": P, **{k: float(np.median(v)) for k, v in parts.items()}, "kernel_ms_min": min(km), "kernel_ms_med": float(np.median(km)),
                  "ss_solve_ms": float(np.median(whole))}))
