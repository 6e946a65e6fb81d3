#!/usr/bin/env python3
"""Single small problems (the reference's own sizes): clc_solve in ONE single-workgroup launch of the resident kernel (default)
against the 256-workgroup step_kernel chain (explicit flags) — median wall time per solve."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0)
x0 = sd.pose7_from_T(np.eye(4))
BASE = 2 | 16 | 32 | 128 | 256 | 512
out = {}
for name, S in (("c1 (50 poses x ~114 pts)", sd.GenerateSimData(1, noise_sigma=0.01)), ("20 x 100", sd.sim_fixed_count(3, 20, 100, noise_sigma=0.01)),
                ("20 x 500", sd.sim_fixed_count(3, 20, 500, noise_sigma=0.01)), ("100 x 110", sd.sim_fixed_count(3, 100, 110, noise_sigma=0.01))):
    rec = clc.flatten_observations(S, False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    row = {"observations": int(rec.shape[0]), "resident_single": sv.debug_resident_single()}
    for label, fl in (("resident", -1), ("step_chain", BASE), ("resident", -1), ("step_chain", BASE)):
        sv.set_launch(0, fl)
        for _ in range(20):
            r = sv.solve(x0, trace_cap=0)
        ts = []
        for _ in range(7):
            t = time.perf_counter()
            for _ in range(100):
                r = sv.solve(x0, trace_cap=0)
            ts.append((time.perf_counter() - t) / 100)
        row.setdefault(label + "_ms", []).append(1e3 * float(np.median(ts)))
        row["passes"] = int(r.summary.num_evaluations)
    sv.set_launch(0, -1)
    out[name] = row
    print(name, json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_single.json", "w"), indent=1)
