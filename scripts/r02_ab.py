#!/usr/bin/env python3
"""Round-2 A/B on one MI355X: per-point compact layout (flags 178) vs row layout (434) vs row layout with equal wave
shares (946): evaluation kernel alone (hipEvents around back-to-back launches), step_kernel period inside a solve, solve
wall time — single problem across sizes, then the batched solver on C3 and one C4 shard.
usage: r02_ab.py [out.csv] [--no-batched] [--max-obs N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else None
max_obs = int(sys.argv[sys.argv.index("--max-obs") + 1]) if "--max-obs" in sys.argv else 32_000_000
FLAGS = {"compact": 2 | 16 | 32 | 128, "rows": 2 | 16 | 32 | 128 | 256, "rows_eq": 2 | 16 | 32 | 128 | 256 | 512}
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
lines = ["workload,obs,layout,eval_us,eval_alg_GBs,step_us,solve_ms,passes,evals_per_s"]
for poses, pts in ((50, 110), (200, 500), (2000, 500), (8000, 500), (32000, 500), (64000, 500)):
    if poses * pts > max_obs:
        continue
    S = sd.sim_fixed_count(1000 if poses == 2000 else 7, poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    n = rec.shape[0]
    sv.upload(rec)
    del S
    for name, fl in FLAGS.items():
        sv.set_launch(0, fl)
        k = min(sv.time_eval(x0, reps=100 if n < 4e6 else 20) for _ in range(3)) * 1e3
        for _ in range(3):
            r = sv.solve(x0, trace_cap=0)
        reps = 30 if n < 4e6 else 8
        t = time.perf_counter()
        for _ in range(reps):
            r = sv.solve(x0, trace_cap=0)
        dt = (time.perf_counter() - t) / reps
        p = r.summary.num_evaluations
        try:
            st = min(sv.time_steps(x0, 2, p - 1)[0] for _ in range(3)) * 1e3 if p >= 4 else float("nan")
        except Exception as e:
            st = float("nan")
        lines.append(f"single,{n},{name},{k:.2f},{64 * n / k / 1e3:.0f},{st:.2f},{dt * 1e3:.4f},{p},{p * n / dt:.4e}")
        print(lines[-1], flush=True)
    sv.set_launch(0, -1)
    del rec
if "--no-batched" not in sys.argv:
    for label, P in (("C3", 1024), ("C4shard", 8192)):
        rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
        sv.upload_batched(rec, off)
        n = int(off[-1])
        del rec
        for name, fl in (("compact", 2 | 16 | 32), ("rows", 2 | 16 | 32 | 256), ("compact", 2 | 16 | 32), ("rows", 2 | 16 | 32 | 256)):
            sv.set_launch(0, fl)
            k = min(sv.time_batched_eval(xb, reps=10) for _ in range(3)) * 1e3
            ts = []
            for _ in range(10):
                t = time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter() - t)
            dt = float(np.median(ts[2:]))
            ev = sum(sms[i].num_evaluations for i in range(P)) * 10000
            lines.append(f"{label},{n},{name},{k:.2f},{64 * n / k / 1e3:.0f},nan,{dt * 1e3:.4f},{max(s.num_evaluations for s in sms)},{ev / dt:.4e}")
            print(lines[-1], flush=True)
        sv.set_launch(0, -1)
if out:
    open(out, "w").write("\n".join(lines) + "\n")
