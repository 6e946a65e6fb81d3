#!/usr/bin/env python3
"""Why does the same 557 MB launch of the row-layout evaluation kernel (3.2e7 observations, beyond the Infinity Cache) take 83 or 134 us?
(bench's best-of said 0.79 of the HBM peak, the rocprofv3 mean of the same run 0.64.)  Runs on the hooks build
(CLC_LIBRARY=.../libclc_hip_hooks.so), alone or under `rocprofv3 --kernel-trace`:
  phase A   30 groups of 10 back-to-back launches, group means by HIP events, IN ORDER (first groups after the upload: cold TLB? clocks?)
  phase B   the same after 200 ms of idle (clock ramp-down?)
  phase C   40 single launches, each followed by a stream synchronisation (launch-to-launch gaps of ~20 us)
  phase D   10 groups of 100 back-to-back launches (10 ms of sustained streaming each: power / thermal management?)
Prints one JSON line; scripts/summarize_kernels.py turns it and the kernel trace into profiles/r05_large.md."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
sv = clc.Solver(0, library="hooks")
x0 = sd.pose7_from_T(np.eye(4))
rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
big = np.ascontiguousarray(np.tile(rec, ((n + rec.shape[0] - 1) // rec.shape[0], 1))[:n])
t0 = time.perf_counter()
sv.upload(big)
t_up = time.perf_counter() - t0
del big
_, n_rows, _, _ = sv.debug_rows()
bytes_per_launch = n_rows * (64 * 16 + 64)
out = {"observations": n, "row_layout_bytes": bytes_per_launch, "upload_s": t_up}
mark = lambda: sv.pose_plus(x0[None, :], np.zeros((1, 6)))  # a tiny kernel: phase marker in the trace
mark()
out["A_first_groups_us"] = [1e3 * sv.time_eval(x0, reps=10) for _ in range(30)]
time.sleep(0.2)
mark()
out["B_after_idle_us"] = [1e3 * sv.time_eval(x0, reps=10) for _ in range(10)]
mark()
out["C_single_launches_us"] = [1e3 * sv.time_eval(x0, reps=1) for _ in range(40)]
mark()
out["D_sustained_us"] = [1e3 * sv.time_eval(x0, reps=100) for _ in range(10)]
mark()
for k in ("A_first_groups_us", "B_after_idle_us", "C_single_launches_us", "D_sustained_us"):
    v = np.array(out[k])
    out[k.replace("_us", "_stats")] = {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max()),
                                      "frac_of_8TBps_at_median": bytes_per_launch / (float(np.median(v)) * 1e-6) / 8e12}
print(json.dumps(out))
