#!/usr/bin/env python3
"""The default-path clc_solve of ONE problem larger than the chip holds (> 2.6e6 observations: the step chain, csrc/clc_kernels.hpp
step_kernel), measured as a WHOLE solve: wall time per solve, per-pass time (solve / evaluation passes), the steady-state launch
period (HIP events around launches 2..passes-2, hooks build), the evaluation kernel alone at the same size, bytes of the row layout,
parity against the oracle.  usage: large_solve_probe.py [n_poses ...]   (500 points per pose; 8000 -> 4e6, 64000 -> 3.2e7)
Prints one JSON line per size."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
import oracle

sizes = [int(a) for a in sys.argv[1:]] or [8000, 64000]
x0 = sd.pose7_from_T(np.eye(4))
for n_poses in sizes:
    t = time.perf_counter()
    rec = clc.flatten_observations(sd.sim_fixed_count(1000 + n_poses, n_poses, 500, noise_sigma=0.01), False)
    t_gen = time.perf_counter() - t
    sv = clc.Solver(0, library="hooks")
    t = time.perf_counter(); sv.upload(rec); t_up = time.perf_counter() - t
    pi = sv.path_info()
    _, n_rows, _, _ = sv.debug_rows()
    layout_bytes = n_rows * (64 * 16 + 64)
    out = {"observations": int(rec.shape[0]), "n_rows": int(n_rows), "row_layout_bytes": int(layout_bytes), "gen_s": t_gen, "upload_s": t_up,
           "coop_resident": int(pi.coop_resident), "single_resident": int(pi.single_resident)}
    for _ in range(3):
        res = sv.solve(x0, trace_cap=0)
    walls = []
    for _ in range(7):
        t = time.perf_counter(); res = sv.solve(x0, trace_cap=0); walls.append(1e3 * (time.perf_counter() - t))
    passes = int(res.summary.num_evaluations)
    out.update(solve_ms_median=float(np.median(walls)), solve_ms_min=float(min(walls)), passes=passes, iterations=int(res.summary.num_iterations),
               termination=res.termination)
    per_pass = np.median(walls) * 1e-3 / passes
    out["per_pass_us"] = 1e6 * per_pass
    out["frac_moved_whole_solve"] = layout_bytes / per_pass / 8e12
    ms, n = sv.time_steps(x0, 2, passes - 2)
    out["step_period_us"] = 1e3 * ms
    out["frac_moved_step_period"] = layout_bytes / (ms * 1e-3) / 8e12
    ev = [1e3 * sv.time_eval(x0, reps=10) for _ in range(30)]
    out["eval_alone_us_median"] = float(np.median(ev[10:]))
    out["frac_moved_eval_alone"] = layout_bytes / (np.median(ev[10:]) * 1e-6) / 8e12
    # sustained: solves back to back for ~50 ms (the clock transient of profiles/r05_large.md)
    t = time.perf_counter(); k = 0
    while time.perf_counter() - t < 0.3:
        sv.solve(x0, trace_cap=0); k += 1
    walls2 = []
    for _ in range(7):
        t = time.perf_counter(); sv.solve(x0, trace_cap=0); walls2.append(1e3 * (time.perf_counter() - t))
    out["solve_ms_sustained_median"] = float(np.median(walls2))
    out["frac_moved_whole_solve_sustained"] = layout_bytes / (np.median(walls2) * 1e-3 / passes) / 8e12
    # parity
    big = rec.shape[0] > 8_000_000
    t = time.perf_counter()
    ref = oracle.solve(rec, x0, linear_solver="ne" if big else "qr", threads=min(64, oracle.max_threads()) if big else 1)
    out["oracle_s"] = time.perf_counter() - t
    out["oracle"] = "ne, threads" if big else "qr"
    out["dT"] = float(np.abs(sd.T_from_pose7(res.pose) - sd.T_from_pose7(ref.pose)).max())
    out["dcost"] = float(abs(res.summary.final_cost - ref.summary.final_cost))
    out["iterations_oracle"] = int(ref.summary.num_iterations)
    out["termination_oracle"] = ref.termination
    print(json.dumps(out), flush=True)
    sv.close()
    del rec
