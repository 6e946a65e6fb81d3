#!/usr/bin/env python3
"""The five configurations of BASELINE.json on one MI355X, one line each: solve time (observations resident), passes,
evals/s, parity against the oracle (T_cl, final cost, iteration count) — gpurun_out/<CLC_CONFIGS_OUT, default r06_configs.json> (copied to
profiles/)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
import oracle

x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
out = []


def timed(f, n=30):
    for _ in range(3): r = f()
    best = 1e9
    for rep in range(4):
        t = time.perf_counter()
        for _ in range(n): r = f()
        best = min(best, (time.perf_counter() - t) / n)
    return best, r


def single(name, rec, start, oracle_check=True):
    sv.upload(rec)
    dt, r = timed(lambda: sv.solve(start, trace_cap=0))
    line = {"config": name, "observations": int(rec.shape[0]), "layout": "rows" if (sv.debug_rows()[0] and rec.shape[0] >= 200000) else "compact28",
            "path": ("cooperative one-launch solve" if sv.debug_coop()[0] else ("single-workgroup resident solve" if sv.debug_resident_single()[0] else "step chain")),
            "solve_ms": dt * 1e3, "passes": int(r.summary.num_evaluations), "lm_iterations": int(r.summary.num_iterations),
            "evals_per_s": r.summary.num_evaluations * rec.shape[0] / dt, "final_cost": r.summary.final_cost}
    if oracle_check:
        ref = oracle.solve(rec, start, linear_solver="qr")
        line.update({"T_cl_max_abs_err_vs_oracle": float(np.abs(sd.T_from_pose7(r.pose) - sd.T_from_pose7(ref.pose)).max()),
                     "final_cost_abs_err_vs_oracle": abs(r.summary.final_cost - ref.summary.final_cost),
                     "iterations_oracle": int(ref.summary.num_iterations)})
    out.append(line)
    print(json.dumps(line), flush=True)


single("C1 simulation_lasercamcal_node default (50 poses x 180 rays, sigma 0.01)", clc.flatten_observations(sd.GenerateSimData(1, noise_sigma=0.01), False), x0)
single("C2 single T_cl, 1e6 observations (2000 x 500, sigma 0.01)", clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False), x0)
gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
S5 = sd.sim_board_edges(5, 2000, 500, noise_sigma=0.002)
single("C5 1e6 obs + 2 board-edge residuals per scan (1 004 000 records)", clc.flatten_observations(S5, True, True),
       oracle.pose_plus(gt, np.array([0.05, -0.04, 0.03, 0.05, -0.06, 0.04])))
for name, P in (("C3 batch of 1024 problems x 1e4 observations", 1024), ("C4 shard: 8192 of 65536 problems x 1e4 observations (one GPU's share)", 8192)):
    rec, off, xb, gtb = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
    sv.upload_batched(rec, off)
    dt, (poses, sms) = timed(lambda: sv.solve_batched(xb), n=8)
    ev = sum(sms[k].num_evaluations for k in range(P)) * 10000
    worst_T, worst_c = 0.0, 0.0
    for k in range(0, P, P // 32):
        ref = oracle.solve(rec[off[k]:off[k + 1]], xb[k], linear_solver="qr")
        worst_T = max(worst_T, float(np.abs(sd.T_from_pose7(poses[k]) - sd.T_from_pose7(ref.pose)).max()))
        worst_c = max(worst_c, abs(sms[k].final_cost - ref.summary.final_cost))
    line = {"config": name, "problems": P, "observations": int(off[-1]), "layout": "rows", "ms_per_batch": dt * 1e3, "problems_per_s": P / dt,
            "evals_per_s": ev / dt, "lm_iterations_min_max": [int(min(s.num_iterations for s in sms)), int(max(s.num_iterations for s in sms))],
            "T_cl_max_abs_err_vs_oracle_32_problems": worst_T, "final_cost_abs_err_vs_oracle_32_problems": worst_c}
    out.append(line)
    print(json.dumps(line), flush=True)
    del rec
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", os.environ.get("CLC_CONFIGS_OUT", "r06_configs.json")), "w"), indent=1)
