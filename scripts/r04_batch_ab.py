import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0)
out = {"lib": os.path.basename(os.environ.get("CLC_LIBRARY", "default"))}
for label, P in (("C3", 1024), ("C4", 8192)):
    rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
    sv.upload_batched(rec, off); del rec
    for _ in range(3): sv.solve_batched(xb)
    o = clc.default_options(); o.profile_events = 1
    k = [sv.solve_batched(xb, o)[1][0].eval_kernel_ms for _ in range(9)]
    t = []
    for _ in range(9):
        t0 = time.perf_counter(); sv.solve_batched(xb); t.append(time.perf_counter() - t0)
    out[label] = {"kernel_ms_min": round(min(k), 4), "kernel_ms_med": round(float(np.median(k)), 4), "solve_ms": round(1e3 * float(np.median(t)), 4)}
print(json.dumps(out))
