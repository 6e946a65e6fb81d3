#!/usr/bin/env python3
"""Round 4 quick timing on the GPU box: the cooperative C2 solve (wall + kernel), single-workgroup solves under both controllers
(hooks build, clc_debug_single_controller = the cooperative kernel's register-state controller instead of the LDS-state one), C3 and a C4 shard through the resident batched kernel.  One JSON line each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
out = {}
quick = "--quick" in sys.argv
# ---- cooperative solve over sizes ----
shapes = [(2000, 500)] if quick else [(40, 500), (400, 500), (2000, 500), (2000, 1000)]
for n_poses, pts in shapes:
    rec = clc.flatten_observations(sd.sim_fixed_count(1000, n_poses, pts, noise_sigma=0.01), False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    for _ in range(5):
        res = sv.solve(x0, trace_cap=0)
    t = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(20):
            res = sv.solve(x0, trace_cap=0)
        t.append((time.perf_counter() - t0) / 20)
    o = clc.default_options(); o.profile_events = 2
    k = [sv.solve(x0, o, trace_cap=0).summary.eval_kernel_ms for _ in range(9)]
    built, ppl, solves, aborts, off = sv.debug_coop()
    row = {"observations": int(rec.shape[0]), "ppl": ppl, "built": built, "aborts": aborts, "solve_ms_median": 1e3 * float(np.median(t)),
           "solve_ms_min": 1e3 * min(t), "kernel_ms_min": min(k), "kernel_ms_median": float(np.median(k)), "passes": int(res.summary.num_evaluations),
           "us_per_pass": 1e3 * min(k) / res.summary.num_evaluations, "termination": int(res.summary.termination), "iterations": int(res.summary.num_iterations)}
    print("coop", json.dumps(row), flush=True)
    out[f"coop_{rec.shape[0]}"] = row
# ---- single-workgroup solves, both controllers ----
for name, S in (("c1", sd.GenerateSimData(1, noise_sigma=0.01)), ("20x500", sd.sim_fixed_count(3, 20, 500, noise_sigma=0.01))):
    rec = clc.flatten_observations(S, False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    row = {"observations": int(rec.shape[0])}
    for label, mask in (("default_controller", 0), ("coop_controller", 4), ("default_controller", 0), ("coop_controller", 4)):
        sv.debug_single_controller(mask == 4)
        for _ in range(20):
            r = sv.solve(x0, trace_cap=0)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(100):
                r = sv.solve(x0, trace_cap=0)
            ts.append((time.perf_counter() - t0) / 100)
        row.setdefault(label + "_ms", []).append(1e3 * float(np.median(ts)))
        row["passes"] = int(r.summary.num_evaluations)
    sv.debug_single_controller(False)
    print("single", name, json.dumps(row), flush=True)
    out["single_" + name] = row
# ---- batched ----
for label, P in (("C3", 1024), ("C4shard", 8192)):
    rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter() - t0)
    o = clc.default_options(); o.profile_events = 1
    ks = [sv.solve_batched(xb, o)[1][0].eval_kernel_ms for _ in range(5)]
    row = {"problems": P, "solve_batched_ms_median": 1e3 * float(np.median(ts[2:])), "min": 1e3 * min(ts), "kernel_ms_min": min(ks),
           "passes_total": int(sum(s.num_evaluations for s in sms)), "resident": sv.debug_resident()}
    print("batched", label, json.dumps(row), flush=True)
    out["batched_" + label] = row
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r04_quick"
json.dump(out, open(os.path.join(ROOT, "gpurun_out", tag + ".json"), "w"), indent=1)
