#!/usr/bin/env python3
"""Batched evaluation / solve timing on C3 and one C4 shard for the library named by CLC_LIBRARY (A/B of kernel builds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0)
tag = os.path.basename(os.environ.get("CLC_LIBRARY", "default"))
for label, P in (("C3", 1024), ("C4shard", 8192)):
    rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
    sv.upload_batched(rec, off)
    del rec
    for name, fl in (("wave64", -1), ("wg256", 2 | 16 | 32 | 256 | 1024), ("wave64", -1), ("wg256", 2 | 16 | 32 | 256 | 1024)):
        sv.set_launch(0, fl)
        k = min(sv.time_batched_eval(xb, reps=10) for _ in range(5)) * 1e3
        ts = []
        for _ in range(10):
            t = time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter() - t)
        print(f"{tag} {label} {name}: eval {k:.2f} us, solve {np.median(ts[2:])*1e3:.4f} ms", flush=True)
    sv.set_launch(0, -1)
