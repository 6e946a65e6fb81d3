#!/usr/bin/env python3
"""(needs the -DCLC_LEGACY_PATHS build: CLC_LIBRARY=camlasercalibratool_amd/csrc/libclc_hip_legacy.so)
Per-wave timeline of the default evaluation kernel: where do the ~9 us at 1e6 obs go, and how
balanced are the waves?  256- vs 512-thread workgroups."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, _capi
S = sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01)
rec = clc.flatten_observations(S, False); x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
def run(bt):
    st = np.zeros(8192 * 8, dtype=np.int64)
    g = _capi.lib().clc_debug_eval_timeline(sv._h, _capi.dptr(x0), C.c_double(0.05), st.ctypes.data_as(C.POINTER(C.c_longlong)), 8192, bt)
    assert g > 0, (g, _capi.lib().clc_last_error())
    return st[: g * 8].reshape(g, 8)
for mult in (1, 8):
    sv.upload(np.ascontiguousarray(np.tile(rec, (mult, 1))))
    for grid, bt in ((256, 256), (512, 256), (256, 512), (224, 512)):
        sv.set_launch(grid, -1)
        s = run(bt)
        t0 = s[:, 0].min()
        end = (s[:, 5] - t0) * 0.01; lend = (s[:, 1] - t0) * 0.01
        nw = bt // 64
        w = np.arange(s.shape[0]) % nw
        blk = np.arange(s.shape[0]) // nw
        loop = s[:, 3].astype(float)
        print(f"N={rec.shape[0]*mult} grid={grid} x {bt}thr: start spread {((s[:,0]-t0)*0.01).max():.2f} us, loop end first/median/last {lend.min():.2f}/{np.median(lend):.2f}/{lend.max():.2f} us, kernel end {end.max():.2f} us | "
              f"loop cycles by wave slot {[int(loop[w==i].mean()) for i in range(nw)]} | by block half {[int(loop[blk < grid//2].mean()), int(loop[blk >= grid//2].mean())]}")
