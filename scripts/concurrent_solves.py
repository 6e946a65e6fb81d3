#!/usr/bin/env python3
"""Aggregate throughput of several independent C2-size solves running concurrently on ONE GPU (one handle = one
stream per host thread): the launch boundaries and controller prologues of one solve overlap the streaming of another."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
recs = [clc.flatten_observations(sd.sim_fixed_count(1000 + i, 2000, 500, noise_sigma=0.01), False) for i in range(4)]
for nthreads in (1, 2, 3, 4):
    solvers = [clc.Solver(0) for _ in range(nthreads)]
    for s, r in zip(solvers, recs): s.upload(r)
    for s in solvers:
        for _ in range(5): s.solve(x0, trace_cap=0)
    evals = [0] * nthreads
    def work(i):
        for _ in range(100):
            r = solvers[i].solve(x0, trace_cap=0)
            evals[i] += r.summary.num_evaluations * recs[i].shape[0]
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; dt = time.perf_counter() - t
    print(f"{nthreads} concurrent solves: {sum(evals)/dt:.3e} evals/s aggregate, {dt/100*1e3:.3f} ms per round of {nthreads} solves", flush=True)
    for s in solvers: s.close()
