#!/usr/bin/env python3
"""Round-1 side measurements: PCIe-inclusive rate of C2, C5 (10^6 obs + board-edge terms) parity and
timing, one C4 shard (8192 problems x 10^4 obs = 5.2 GB, what each of 8 GPUs holds)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
import oracle

sv = clc.Solver(0)
X0 = sd.pose7_from_T(np.eye(4))
# ---- C2 upload cost (host 64-byte records -> resident tiles + compact copy) ----
S = sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01)
rec = clc.flatten_observations(S, False)
sv.upload(rec)
ts = []
for _ in range(5):
    t = time.perf_counter(); sv.upload(rec); ts.append(time.perf_counter() - t)
up = min(ts)
for _ in range(5): r = sv.solve(X0, trace_cap=0)
ts = []
for _ in range(20):
    t = time.perf_counter(); r = sv.solve(X0, trace_cap=0); ts.append(time.perf_counter() - t)
so = min(ts)
ev = r.summary.num_evaluations * rec.shape[0]
print(json.dumps(dict(what="C2 PCIe-inclusive", upload_ms=up * 1e3, upload_GBps=rec.nbytes / up / 1e9, solve_ms=so * 1e3,
                      evals_per_s_resident=ev / so, evals_per_s_incl_upload=ev / (so + up))))
# ---- C5: 10^6 point residuals + 2 board-edge residuals per scan ----
t = time.perf_counter(); S5 = sd.sim_board_edges(5, 2000, 500, noise_sigma=0.002); gen = time.perf_counter() - t
rec5 = clc.flatten_observations(S5, True, True)
gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
x0 = oracle.pose_plus(gt, np.array([0.05, -0.04, 0.03, 0.05, -0.06, 0.04]))
sv.upload(rec5)
res = sv.solve(x0)
ts = []
for _ in range(10):
    t = time.perf_counter(); res = sv.solve(x0, trace_cap=0); ts.append(time.perf_counter() - t)
t = time.perf_counter(); ref = oracle.solve(rec5, x0, linear_solver="qr"); cpu = time.perf_counter() - t
dT = np.abs(sd.T_from_pose7(res.pose) - sd.T_from_pose7(ref.pose)).max()
print(json.dumps(dict(what="C5 full size", records=int(rec5.shape[0]), gen_s=gen, gpu_solve_ms=min(ts) * 1e3, iters=res.summary.num_iterations,
                      iters_oracle=ref.summary.num_iterations, dT_vs_oracle=float(dT), dcost=abs(res.summary.final_cost - ref.summary.final_cost),
                      cpu_oracle_s=cpu, err_vs_gt=float(np.abs(sd.T_from_pose7(res.pose) - sd.T_from_pose7(gt)).max()))))
# ---- C4 shard: 8192 problems x 10^4 obs ----
P = int(os.environ.get("C4_SHARD", "8192"))
t = time.perf_counter()
base_probs, base_gts = sd.sim_batch(99, 256, 20, 500, noise_sigma=0.01)   # 256 distinct problems, tiled 32x
recs = [clc.flatten_observations(p, False) for p in base_probs]
reps = P // 256
allrec = np.concatenate(recs * reps)
off = np.arange(P + 1, dtype=np.int64) * recs[0].shape[0]
gts = np.concatenate([base_gts] * reps)
rng = np.random.default_rng(1)
x0 = sv.pose_plus(np.stack([sd.pose7_from_T(g) for g in gts]), rng.normal(size=(P, 6)) * 0.05)
print("C4 shard generated: %.1f GB in %.1fs" % (allrec.nbytes / 1e9, time.perf_counter() - t), flush=True)
t = time.perf_counter(); sv.upload_batched(allrec, off); upb = time.perf_counter() - t
poses, sms = sv.solve_batched(x0)
ts = []
for _ in range(5):
    t = time.perf_counter(); poses, sms = sv.solve_batched(x0); ts.append(time.perf_counter() - t)
dt = float(np.median(ts))
evals = sum(sms[k].num_evaluations for k in range(P)) * recs[0].shape[0]
k = 123
ref = oracle.solve(recs[k % 256], x0[k], linear_solver="qr")
print(json.dumps(dict(what="C4 shard (1 of 8 GPUs)", problems=P, GB=allrec.nbytes / 1e9, upload_s=upb, solve_ms=dt * 1e3, problems_per_s=P / dt,
                      evals_per_s=evals / dt, iters_max=max(sms[i].num_iterations for i in range(P)),
                      dT_sample=float(np.abs(sd.T_from_pose7(poses[k]) - sd.T_from_pose7(ref.pose)).max()))))
