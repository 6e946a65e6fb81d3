#!/usr/bin/env python3
"""Round-3 profiling workloads in a fixed order, separated by marker launches (clc::plus_kernel with n = 1) so that a rocprofv3
kernel trace / PMC pass of this script can be cut into blocks (scripts/summarize_r03.py): the 64-byte-tile evaluation as the
PMC calibration (it reads exactly 64 B per observation), the row-layout evaluation and the step-kernel solve at C2, the
row-layout evaluation beyond the Infinity Cache, and the batched solves of C3 / one C4 shard on the resident kernel (default)
and on the round-2 paths (flag 4096).  Prints one JSON object: the plan + the timings taken inside the run.
usage: r03_workloads.py [--quick]   (--quick: no 3.2e7-observation block, C4 shard of 2 048 problems)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

quick = "--quick" in sys.argv
BASE = 2 | 16 | 32 | 128 | 256 | 512
ROW_BYTES = 64 * 16 + 64
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
blocks = []


def block(label, **kw):
    sv.pose_plus(x0[None, :], np.zeros((1, 6)))  # marker launch
    blocks.append(dict(label=label, **kw))
    return blocks[-1]


rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
n = rec.shape[0]
sv.upload(rec)
rows_ok, n_rows, _, _ = sv.debug_rows()
sv.set_launch(0, 6)
b = block("calib tiled64 1000000", kind="eval", obs=n, moved_bytes=64 * n, launches=23)
b["hipevent_us"] = sv.time_eval(x0, reps=20) * 1e3
sv.set_launch(0, -1)
b = block("eval rows 1000000", kind="eval", obs=n, moved_bytes=n_rows * ROW_BYTES, launches=53)
b["hipevent_us"] = sv.time_eval(x0, reps=50) * 1e3
b = block("solve 1000000", kind="solve", obs=n, moved_bytes=n_rows * ROW_BYTES)
for _ in range(3):
    r = sv.solve(x0, trace_cap=0)
t = time.perf_counter()
K = 30
for _ in range(K):
    r = sv.solve(x0, trace_cap=0)
b["solve_ms"] = (time.perf_counter() - t) / K * 1e3
b["passes"] = int(r.summary.num_evaluations)
b["step_hipevent_us"] = min(sv.time_steps(x0, 2, b["passes"] - 1)[0] for _ in range(3)) * 1e3
if not quick:
    big = np.ascontiguousarray(np.tile(rec, (32, 1)))
    sv.upload(big)
    nb = big.shape[0]
    del big
    _, nr, _, _ = sv.debug_rows()
    b = block(f"eval rows {nb}", kind="eval", obs=nb, moved_bytes=nr * ROW_BYTES, launches=15)
    b["hipevent_us"] = sv.time_eval(x0, reps=12) * 1e3
del rec
for label, P in (("c3", 1024), ("c4shard", 2048 if quick else 8192)):
    recb, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
    sv.set_launch(0, -1)
    sv.upload_batched(recb, off)
    del recb
    ok, lanes, ppl, rrows = sv.debug_resident()
    _, _, brows_ok, bn_rows = sv.debug_rows()
    for name, fl, moved in (("resident", -1, rrows * lanes * 16 + P * lanes * 8), ("round2", BASE | 4096, bn_rows * ROW_BYTES)):
        sv.set_launch(0, fl)
        for _ in range(2):
            sv.solve_batched(xb)
        b = block(f"{label} {name}", kind="batched", problems=P, obs=int(off[-1]), moved_bytes_per_pass_over_the_data=int(moved), solves=5,
                  resident=bool(ok and fl == -1), lanes=lanes, ppl=ppl)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter() - t)
        b["solve_ms"] = float(np.median(ts)) * 1e3
        b["passes_total"] = int(sum(s.num_evaluations for s in sms))
    sv.set_launch(0, -1)
block("end")
print(json.dumps({"blocks": blocks}))
