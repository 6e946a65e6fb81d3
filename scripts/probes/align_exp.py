import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0); x0 = sd.pose7_from_T(np.eye(4))
for poses, pts in ((2000,500),(2048,512),(2048,500),(2000,512)):
    S = sd.sim_fixed_count(1000, poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False); sv.upload(rec)
    for name, fl in (("rows", 2|16|32|128|256), ("rows_eq", 2|16|32|128|256|512)):
        sv.set_launch(0, fl)
        for _ in range(3): r = sv.solve(x0, trace_cap=0)
        p = r.summary.num_evaluations
        st = min(sv.time_steps(x0, 2, p-1)[0] for _ in range(5))*1e3
        ts=[]
        for _ in range(30):
            t=time.perf_counter(); r=sv.solve(x0, trace_cap=0); ts.append(time.perf_counter()-t)
        print(poses, pts, rec.shape[0], name, "step_us %.2f solve_ms %.4f passes %d" % (st, np.median(ts)*1e3, p), flush=True)
    sv.set_launch(0,-1)
