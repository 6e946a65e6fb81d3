import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, _capi
S = sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01)
rec = clc.flatten_observations(S, False); x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0); sv.upload(rec)
sv.set_launch(0, int(os.environ.get('CLC_FLAGS','-1')))
o = clc.default_options(); o.profile_events = int(os.environ.get('CLC_PROFILE', '2')); o.max_num_iterations = 5   # stop mid-way: the last lm_kernel ran a regular step
for _ in range(5):
    r = sv.solve(x0, o)
    p = (C.c_longlong * 8)()
    _capi.lib().clc_debug_lm_profile(sv._h, p)
    # step-kernel path: stamps of the leader workgroup, wave 0 (kernel entry, rows combined, controller start / end, mailbox)
    print("cycles: entry->rows combined %d, ->controller start %d | controller %d, mailbox %d (termination %s)"
          % (p[6]-p[0], p[1]-p[6], p[2]-p[1], p[3]-p[2], r.termination))
