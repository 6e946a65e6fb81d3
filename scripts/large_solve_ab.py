#!/usr/bin/env python3
"""A/B of the step chain's launch flags on ONE problem beyond the chip (default 64000 x 500 = 3.2e7 observations): sustained per-pass time
of a whole clc_solve.  usage: large_solve_ab.py [n_poses] [flags ...]   (-1 = library default)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
flag_sets = [int(a) for a in sys.argv[2:]] or [-1, 946 | 4, (946 | 4) & ~512]
x0 = sd.pose7_from_T(np.eye(4))
rec = clc.flatten_observations(sd.sim_fixed_count(1000 + n_poses, n_poses, 500, noise_sigma=0.01), False)
libs = os.environ.get("CLC_AB_LIBS", "hooks").split(",")
svs = [clc.Solver(0, library=l) for l in libs]
for s_ in svs:
    s_.upload(rec)
sv = svs[0]
_, n_rows, _, _ = sv.debug_rows()
layout_bytes = n_rows * (64 * 16 + 64)
for rnd in range(3):
  for li, sv in enumerate(svs):
    for fl in flag_sets:
        sv.set_launch(0, fl)
        t = time.perf_counter()
        while time.perf_counter() - t < 0.25:
            res = sv.solve(x0, trace_cap=0)
        w = []
        for _ in range(9):
            t = time.perf_counter(); res = sv.solve(x0, trace_cap=0); w.append(time.perf_counter() - t)
        passes = int(res.summary.num_evaluations)
        pp = float(np.median(w)) / passes
        print(json.dumps({"round": rnd, "lib": os.path.basename(libs[li]), "flags": fl, "passes": passes, "per_pass_us": 1e6 * pp, "frac_moved": layout_bytes / pp / 8e12,
                          "pose": [float(v) for v in res.pose[:3]]}), flush=True)
ev = [1e3 * sv.time_eval(x0, reps=10) for _ in range(30)]
print(json.dumps({"eval_alone_us": float(np.median(ev[10:])), "frac": layout_bytes / (np.median(ev[10:]) * 1e-6) / 8e12}))
for wl in (True, False):
    ev = [1e3 * sv.time_eval(x0, reps=10, with_loss=wl) for _ in range(30)]
    print(json.dumps({"with_loss": wl, "eval_alone_us": float(np.median(ev[10:])), "frac": layout_bytes / (np.median(ev[10:]) * 1e-6) / 8e12}))
