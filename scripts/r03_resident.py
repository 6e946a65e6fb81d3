#!/usr/bin/env python3
"""Round 3: the on-chip resident batched solver (default) against the round-2 paths (flag 4096: whole-solve launch on the
row layout at C3, lockstep one-wave-per-problem at a C4 shard) — median clc_solve_batched wall time and agreement."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
BASE = 2 | 16 | 32 | 128 | 256 | 512
sv = clc.Solver(0)
out = {"device": sv.device_info()[0], "library": os.path.basename(os.environ.get("CLC_LIBRARY", "default"))}
sizes = [("C3", 1024), ("C4shard", 8192)]
if len(sys.argv) > 1:
    sizes = [(f"P{p}", int(p)) for p in sys.argv[1:]]
for label, P in sizes:
    rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
    t = time.perf_counter(); sv.upload_batched(rec, off); up = time.perf_counter() - t
    r = {"upload_s": up}
    res = {}
    for name, fl, up_fl in (("resident256", -1, -1), ("round2_default", BASE | 4096, None), ("resident512", -1, BASE | 8192),
                            ("resident256", -1, -1), ("round2_default", BASE | 4096, None), ("resident512", -1, BASE | 8192)):
        if up_fl is not None:  # the lane layout is chosen at upload
            sv.set_launch(0, up_fl)
            sv.upload_batched(rec, off)
            ok, lanes, ppl, rows = sv.debug_resident()
            r[name] = {"resident": ok, "lanes": lanes, "max_ppl": ppl, "rows": rows, "resident_bytes": rows * lanes * 16}
        sv.set_launch(0, fl)
        ts = []
        for _ in range(12):
            t = time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter() - t)
        ms = float(np.median(ts[2:]) * 1e3)
        r.setdefault(name + "_ms", []).append(ms)
        res[name] = (poses, [(s.termination, s.num_iterations, s.num_evaluations, s.final_cost) for s in sms])
        print(f"{label} {name}: solve_batched {ms:.4f} ms (min {min(ts)*1e3:.4f})", flush=True)
    sv.set_launch(0, -1)
    pa, sa = res["resident256"]; pb, sb = res["round2_default"]
    r["same_decisions"] = int(sum(a[:3] == b[:3] for a, b in zip(sa, sb)))
    r["max_pose_diff"] = float(np.abs(pa - pb).max())
    r["max_cost_diff"] = float(max(abs(a[3] - b[3]) for a, b in zip(sa, sb)))
    r["passes_total"] = int(sum(a[2] for a in sa))
    out[label] = r
    print(label, json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_resident.json", "w"), indent=1)
