#!/usr/bin/env python3
"""Batches whose points carry z (p.z != 0): the on-chip 512-lane z form of resident_solve_kernel against the lockstep launches on the
rows that carry z (what such a batch ran before round 5).  C3 size (1 024 x 1e4 observations) and a C4 shard (8 192 x 1e4)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0)
BASE = 2 | 16 | 32 | 128 | 256 | 512
for P in (1024, 8192):
    rec, off, x0, gt = sd.sim_shard_records(4242, 0, P, 20, 500, 0.01)
    rec[:, 6] = np.random.default_rng(5).normal(size=rec.shape[0]) * 0.02
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    pi = sv.path_info()
    row = {"problems": P, "resident": pi.batched_resident, "z": pi.batched_points_carry_z, "lanes": pi.batched_lanes, "ppl": pi.batched_points_per_lane}
    for name, fl in (("on_chip_z", -1), ("lockstep_rows_z", BASE | 4096)):
        sv.set_launch(0, fl)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); p, s = sv.solve_batched(x0); ts.append(time.perf_counter() - t)
        row[name + "_ms"] = 1e3 * float(np.median(ts[2:]))
    o = clc.default_options(); o.profile_events = 1
    sv.set_launch(0, -1)
    row["on_chip_kernel_ms"] = min(sv.solve_batched(x0, o)[1][0].eval_kernel_ms for _ in range(3))
    print(json.dumps(row), flush=True)
    del rec
