#!/usr/bin/env python3
"""Where the time of the Session flow goes at C2 size (store / select / closed form / solve / analysis), per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
S = sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01)
sv = clc.Solver(0)
x0 = sd.pose7_from_T(np.eye(4))
def T(f, *a):
    t = time.perf_counter(); r = f(*a); return (time.perf_counter() - t) * 1e3, r
for rep in range(4):
    t_store, _ = T(sv.store_observations, S)
    t_sel1, _ = T(sv.select_observations, True, False)
    t_cf, _ = T(sv.closed_form)
    t_sel2, _ = T(sv.select_observations, False, False)
    t_solve, r = T(sv.solve, x0, None, 0)
    t_info, _ = T(sv.information, r.pose)
    t_flat, rec = T(clc.flatten_observations, S, False)
    t_up, _ = T(sv.upload, rec)
    print(f"rep {rep}: store {t_store:.3f} ms, select(linefit) {t_sel1:.3f}, closed_form {t_cf:.3f}, select(points) {t_sel2:.3f}, "
          f"solve {t_solve:.3f}, information {t_info:.3f} | host flatten {t_flat:.3f}, clc_upload(64 B records) {t_up:.3f}", flush=True)
