#!/usr/bin/env python3
"""Small profiling targets (one per invocation) to run under rocprofv3 with a timeout:
   resident P      batched solve of P problems x 1e4 observations on the resident kernel (5 solves) and on the round-2 path (5 solves)
   step N          5 clc_solve of an N-observation problem through the step-kernel chain
   coop N [ITERS]  5 default clc_solve of an N-observation problem: one coop_solve_kernel launch each (csrc/clc_coop.hpp); ITERS caps
                   max_num_iterations (two pass counts give the instructions per pass)
   large N         30 default clc_solve of ONE N-observation problem beyond the chip's capacity (the step chain; the last 10 behind a marker kernel)
   eval N          20 launches of the row-layout evaluation kernel on N observations + 20 of the 64-byte-tile kernel (PMC calibration)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
what, n = sys.argv[1], int(sys.argv[2])
sv = clc.Solver(0)
x0 = sd.pose7_from_T(np.eye(4))
BASE = 2 | 16 | 32 | 128 | 256 | 512
out = {"what": what, "n": n}
if what == "resident":
    rec, off, xb, gt = sd.sim_shard_records(65536, 0, n, 20, 500, 0.01)
    sv.upload_batched(rec, off)
    ok, lanes, ppl, rows = sv.debug_resident()
    _, _, _, bn_rows = sv.debug_rows()
    out.update(resident=ok, lanes=lanes, ppl=ppl, lane_layout_bytes=rows * lanes * 16 + n * lanes * 8, row_layout_bytes=bn_rows * (64 * 16 + 64))
    for name, fl in (("resident", -1), ("round2", BASE | 4096)):
        sv.set_launch(0, fl)
        sv.pose_plus(x0[None, :], np.zeros((1, 6)))  # marker
        ts = []
        for _ in range(5):
            t = time.perf_counter(); p, s = sv.solve_batched(xb); ts.append(time.perf_counter() - t)
        out[name + "_ms"] = 1e3 * float(np.median(ts))
        out["passes_total"] = int(sum(q.num_evaluations for q in s))
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))
elif what in ("coop", "coopz"):
    rec = clc.flatten_observations(sd.sim_fixed_count(1000, n // 500, 500, noise_sigma=0.01), False)
    if what == "coopz":  # points off the lidar plane: the 24-byte-slot form of the kernel
        rec[:, 6] = np.random.default_rng(3).normal(size=rec.shape[0]) * 0.02
    sv.upload(rec)
    built, ppl, _, _, _ = sv.debug_coop()
    o = clc.default_options()
    if len(sys.argv) > 3:
        o.max_num_iterations = int(sys.argv[3])
    for _ in range(int(os.environ.get("CLC_PROBE_SOLVES", "5"))):  # (the trace run of scripts/profile_kernels.sh asks for 60: a steadier median)
        r = sv.solve(x0, o, trace_cap=0)
    _, _, solves, aborts, off = sv.debug_coop()
    out.update(workgroups=sv.path_info().coop_workgroups, z=sv.path_info().coop_points_carry_z, coop_built=built, points_per_lane=ppl, coop_solves=solves, aborts=aborts, passes=int(r.summary.num_evaluations), solve_ms=r.summary.solve_ms,
               lane_layout_bytes=256 * ppl * 256 * 16 + 256 * 256 * 8 + (n // 500) * 48)
elif what == "large":  # the DEFAULT clc_solve of one problem beyond the chip (step chain; n >= 2.7e6): 20 warm solves, a marker, 10 solves
    rec = clc.flatten_observations(sd.sim_fixed_count(1000 + n // 500, n // 500, 500, noise_sigma=0.01), False)
    sv.upload(rec)
    pi = sv.path_info()
    for _ in range(20):
        r = sv.solve(x0, trace_cap=0)
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))  # marker
    ts = []
    for _ in range(10):
        t = time.perf_counter(); r = sv.solve(x0, trace_cap=0); ts.append(time.perf_counter() - t)
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))
    out.update(observations=int(rec.shape[0]), coop_resident=int(pi.coop_resident), n_rows=int(pi.n_rows), row_layout_bytes=int(pi.n_rows) * (64 * 16 + 64),
               passes=int(r.summary.num_evaluations), iterations=int(r.summary.num_iterations), solve_ms_median=1e3 * float(np.median(ts)))
elif what == "step":
    rec = clc.flatten_observations(sd.sim_fixed_count(1000, n // 500, 500, noise_sigma=0.01), False)
    sv.set_launch(0, BASE)
    sv.upload(rec)
    for _ in range(5):
        r = sv.solve(x0, trace_cap=0)
    out.update(passes=int(r.summary.num_evaluations), solve_ms=r.summary.solve_ms)
else:
    rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
    reps = max(1, n // rec.shape[0])
    big = np.ascontiguousarray(np.tile(rec, (reps, 1)))
    sv.upload(big)
    _, n_rows, _, _ = sv.debug_rows()
    out.update(observations=int(big.shape[0]), row_layout_bytes=n_rows * (64 * 16 + 64))
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))
    out["rows_us"] = 1e3 * sv.time_eval(x0, reps=20)
    sv.set_launch(0, 6)
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))
    out["tiled64_us"] = 1e3 * sv.time_eval(x0, reps=20)
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))
print(json.dumps(out))
