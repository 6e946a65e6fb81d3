#!/usr/bin/env python3
"""Cooperative whole-GPU solve (csrc/clc_coop.hpp) beside the step chain, on the GPU box: clc_solve wall time (median of 15,
host clock around the C call) and the coop launch's own duration (HIP events, profile_events = 2) over problem sizes.
Prints one JSON line per size."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

STEP_CHAIN = 2 | 16 | 32 | 128 | 256 | 512
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
shapes = [(40, 500), (200, 500), (400, 500), (2000, 500), (2000, 1000)] if "--quick" not in sys.argv else [(2000, 500)]
for n_poses, pts in shapes:
    rec = clc.flatten_observations(sd.sim_fixed_count(1000, n_poses, pts, noise_sigma=0.01), False)
    out = {"observations": int(rec.shape[0])}
    for name, flags in (("coop", -1), ("step_chain", STEP_CHAIN)):
        sv.set_launch(0, flags)
        sv.upload(rec)
        for _ in range(3):
            res = sv.solve(x0)
        t = []
        for _ in range(15):
            t0 = time.perf_counter(); res = sv.solve(x0); t.append(time.perf_counter() - t0)
        out[name] = {"solve_ms_median": 1e3 * float(np.median(t)), "solve_ms_min": 1e3 * min(t), "summary_solve_ms": res.summary.solve_ms,
                     "iterations": res.summary.num_iterations, "evaluations": res.summary.num_evaluations}
        if name == "coop":
            built, ppl, solves, aborts, off = sv.debug_coop()
            o = clc.default_options(); o.profile_events = 2
            k = [sv.solve(x0, o).summary.eval_kernel_ms for _ in range(7)]
            out[name].update({"built": built, "points_per_lane": ppl, "aborts": aborts, "kernel_ms_min": min(k), "kernel_ms_median": float(np.median(k)),
                              "us_per_pass": 1e3 * min(k) / res.summary.num_evaluations})
    print(json.dumps(out), flush=True)
