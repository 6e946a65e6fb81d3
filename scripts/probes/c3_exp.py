import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0)
cfgs = [(int(a), int(b), int(c)) for a, b, c in (x.split('x') for x in sys.argv[1:])] or [(1024, 20, 500)]
for P, n_poses, K in cfgs:
    rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, n_poses, K, 0.01)
    sv.upload_batched(rec, off); del rec
    for name, fl in (("default", -1), ("lockstep", 2|16|32|128|256|512|2048), ("default", -1), ("lockstep", 2|16|32|128|256|512|2048)):
        sv.set_launch(0, fl)
        ts=[]
        for _ in range(10):
            t=time.perf_counter(); poses, sms = sv.solve_batched(xb); ts.append(time.perf_counter()-t)
        print(P, n_poses*K, name, "ms_per_batch %.4f" % (np.median(ts[2:])*1e3), "iters", min(s.num_iterations for s in sms), max(s.num_iterations for s in sms), flush=True)
    sv.set_launch(0,-1)
