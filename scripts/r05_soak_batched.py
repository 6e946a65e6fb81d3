#!/usr/bin/env python3
"""Soak of the round-5 step: N x clc_solve_batched_gather on a C3-size shard (and every 10th step the two-call form in between, which
rewrites this rank's segment of the gather buffer), every gathered array compared bit for bit with the first, totals with the first's;
the z form of the batched kernel likewise.  usage: python scripts/r05_soak_batched.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, dist as cdist
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
P = 1024
bad = 0
t0 = time.time()
for label, z in (("xy", False), ("z", True)):
    rec, off, x0, gt = sd.sim_shard_records(4242, 0, P, 20, 500, 0.01)
    if z:
        rec[:, 6] = np.random.default_rng(5).normal(size=rec.shape[0]) * 0.02
    with cdist.ShardSolver(P, device_index=0, rank=0, world=1) as ss:
        ss.upload(rec, off)
        first = ss.solve_gather(x0, ordered=False).copy()
        ev = ss.last_stats.evaluations
        for k in range(steps):
            if k % 10 == 9:
                two = ss.solve(x0, ordered=False)
                if not np.array_equal(two, first):
                    bad += 1; print("MISMATCH two-call", label, k, flush=True)
            out = ss.solve_gather(x0, ordered=False, copy=False)
            st = ss.last_stats
            if not np.array_equal(out, first) or st.evaluations != ev or st.problems != P or st.fused != 1:
                bad += 1; print("MISMATCH", label, k, st.problems, st.evaluations, ev, flush=True)
    print(f"{label}: {steps} steps, mismatches so far {bad}, {time.time() - t0:.1f} s", flush=True)
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
