#!/usr/bin/env python3
"""Does occupancy limit streaming beyond the Infinity Cache?  The cost-only evaluation (no Jacobian) needs far
fewer registers, so it runs at higher occupancy over the same bytes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
for poses in (32000, 64000):
    S = sd.sim_fixed_count(7, poses, 500, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False); n = rec.shape[0]
    sv.upload(rec)
    for fl, grid in ((114, 0), (118, 0), (114, 0), (118, 0), (118, 512), (86, 512), (6, 0)):
        sv.set_launch(grid, fl)
        for jac in (True, False):
            k = min(sv.time_eval(x0, reps=50, with_jacobian=jac) for _ in range(3)) * 1e3
            print(f"N={n} flags={fl} grid={grid} jac={int(jac)}: {k:8.2f} us  streamed {28*n/k/1e3:7.0f} GB/s", flush=True)
