#!/usr/bin/env python3
"""LineFittingCeres batched (clc_line_fit_batched) vs the CPU oracle, synthetic scans."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import camlasercalibratool_amd as clc
import oracle

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(0)
n = rng.integers(60, 180, S)
off = np.zeros(S + 1, dtype=np.int64); off[1:] = np.cumsum(n)
M = int(off[-1])
th = np.repeat(rng.uniform(-1.3, 1.3, S), n); c = np.repeat(rng.uniform(0.8, 5.0, S), n)
t = rng.uniform(-0.5, 0.5, M)
xy = np.stack([c * np.cos(th) - t * np.sin(th), c * np.sin(th) + t * np.cos(th)], 1) + rng.normal(size=(M, 2)) * 0.004
bad = rng.random(M) < 0.05
xy[bad] += rng.normal(size=(int(bad.sum()), 2)) * 0.3
lines0 = np.zeros((S, 2))
sv = clc.Solver(0)
for _ in range(2): sv.line_fit_batched(xy, off, lines0)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); lines, sms = sv.line_fit_batched(xy, off, lines0); ts.append(time.perf_counter() - t0)
dt = min(ts)
iters = np.array([sms[k].num_iterations for k in range(S)]); evals = np.array([sms[k].num_evaluations for k in range(S)])
print(json.dumps(dict(scans=S, points=M, gpu_ms_incl_copies=dt * 1e3, scans_per_s=S / dt, point_evals_per_s=float((evals * n).sum() / dt),
                      iters_mean=float(iters.mean()), iters_max=int(iters.max()))))
K = 3000
t0 = time.perf_counter()
worst = 0.0
for k in range(K):
    r = oracle.line_fit(xy[off[k]:off[k + 1]], lines0[k])
    worst = max(worst, np.abs(r.pose - lines[k]).max())
cpu = (time.perf_counter() - t0) / K
print(json.dumps(dict(cpu_us_per_scan=cpu * 1e6, cpu_scans_per_s=1 / cpu, speedup=S / dt * cpu, max_line_diff_vs_oracle=worst)))
