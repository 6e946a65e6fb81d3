#!/usr/bin/env python3
"""Round-2 profiling workloads: every kernel of the path at known sizes, in a fixed order, separated by marker launches
(clc::plus_kernel with n = 1), so that a rocprofv3 kernel trace / PMC pass of this script can be cut into blocks
(scripts/summarize_r02.py).  Prints one JSON object: the plan (block labels, observations, layout bytes) and the
hipEvent / wall-clock timings taken inside the run.
usage: r02_workloads.py [--max-obs N] [--quick]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

max_obs = int(sys.argv[sys.argv.index("--max-obs") + 1]) if "--max-obs" in sys.argv else 32_000_000
quick = "--quick" in sys.argv
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
blocks = []
ROW_BYTES = 64 * 16 + 64


def marker():
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))


def block(label, **kw):
    marker()
    blocks.append(dict(label=label, **kw))
    return blocks[-1]


LAYOUTS = [("rows", 2 | 16 | 32 | 128 | 256), ("compact28", 2 | 16 | 32), ("tiled64", 6)]
for poses, pts in ((200, 500), (2000, 500), (8000, 500), (32000, 500), (64000, 500)):
    n = poses * pts
    if n > max_obs:
        continue
    S = sd.sim_fixed_count(1000 if poses == 2000 else 7, poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    del S, rec
    rows_ok, n_rows, _, _ = sv.debug_rows()
    for name, fl in LAYOUTS:
        sv.set_launch(0, fl)
        reps = 50 if n <= 4_000_000 else 12
        streamed = {"rows": n_rows * ROW_BYTES, "compact28": 28 * n, "tiled64": 64 * n}[name]
        b = block(f"eval {name} {n}", kind="eval", layout=name, obs=n, streamed_bytes=streamed, launches=reps + 3)
        b["hipevent_us"] = sv.time_eval(x0, reps=reps) * 1e3
    sv.set_launch(0, -1)
    if n in (1_000_000, 4_000_000):
        b = block(f"solve {n}", kind="solve", obs=n, streamed_bytes=n_rows * ROW_BYTES)
        for _ in range(3): r = sv.solve(x0, trace_cap=0)
        t = time.perf_counter()
        K = 30
        for _ in range(K): r = sv.solve(x0, trace_cap=0)
        b["solve_ms"] = (time.perf_counter() - t) / K * 1e3
        b["passes"] = int(r.summary.num_evaluations)
        b["step_hipevent_us"] = min(sv.time_steps(x0, 2, b["passes"] - 1)[0] for _ in range(3)) * 1e3
    if n == 1_000_000:
        b = block("closed form 1000000", kind="closed_form", obs=n, streamed_bytes=48 * n)
        t = time.perf_counter()
        for _ in range(20): sv.closed_form()
        b["call_ms"] = (time.perf_counter() - t) / 20 * 1e3
        b = block("information 1000000", kind="information", obs=n)
        for _ in range(5): sv.information(x0)
if not quick:
    for label, P in (("C3", 1024), ("C4shard", 8192)):
        if P * 10000 > max(max_obs, 10_240_000) * 3:
            continue
        rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
        sv.upload_batched(rec, off)
        n = int(off[-1])
        del rec
        _, _, brows_ok, bn_rows = sv.debug_rows()
        b = block(f"batched eval {label}", kind="batched_eval", obs=n, problems=P, streamed_bytes=bn_rows * ROW_BYTES, launches=12)
        b["hipevent_us"] = sv.time_batched_eval(xb, reps=10) * 1e3
        b = block(f"batched solve {label}", kind="batched_solve", obs=n, problems=P, streamed_bytes=bn_rows * ROW_BYTES)
        ts = []
        for _ in range(8):
            t = time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter() - t)
        b["ms_per_batch"] = float(np.median(ts[2:])) * 1e3
        b["passes_max"] = int(max(s.num_evaluations for s in sms))
    # front end: LineFittingCeres over 1e5 scans, TranScanToPoints over 4096 scans x 1080 rays
    rng = np.random.default_rng(3)
    S = 100_000
    cnt = rng.integers(60, 180, S)
    off = np.zeros(S + 1, dtype=np.int64); off[1:] = np.cumsum(cnt)
    th = rng.uniform(-1.2, 1.2, S); c = rng.uniform(0.8, 5.0, S)
    t = rng.uniform(-0.5, 0.5, off[-1]); sid = np.repeat(np.arange(S), cnt)
    xy = np.stack([c[sid] * np.cos(th[sid]) - t * np.sin(th[sid]), c[sid] * np.sin(th[sid]) + t * np.cos(th[sid])], 1) + rng.normal(size=(off[-1], 2)) * 0.004
    b = block("line fit 1e5 scans", kind="line_fit", scans=S, points=int(off[-1]), streamed_bytes=16 * int(off[-1]))
    t0 = time.perf_counter()
    for _ in range(3): sv.line_fit_batched(xy, off, np.zeros((S, 2)), want_summaries=False)
    b["call_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    n_sc, rays = 4096, 1080
    ranges = rng.uniform(0.3, 12.0, n_sc * rays).astype(np.float32)
    offs = np.arange(n_sc + 1, dtype=np.int64) * rays
    b = block("scan to points 4096 x 1080", kind="scan_to_points", rays=n_sc * rays, streamed_bytes=28 * n_sc * rays)
    t0 = time.perf_counter()
    for _ in range(3): sv.scan_to_points(ranges, offs, -2.35, 0.00436, 0.1)
    b["call_ms"] = (time.perf_counter() - t0) / 3 * 1e3
marker()
print(json.dumps({"blocks": blocks}))
