#!/usr/bin/env python3
"""Where a step_kernel launch spends its time, for EVERY workgroup: a -DCLC_STAMPS build of the library (built by
`python scripts/stamps_step.py --build` where hipcc is, loaded through CLC_LIBRARY) records 100 MHz wall-clock stamps
(entry, rows summed, barrier passed, controller done, stream done, end; first and last wave) per workgroup and launch.
usage (GPU box): CLC_LIBRARY=camlasercalibratool_amd/csrc/libclc_hip_stamps.so python scripts/stamps_step.py [obs_poses pts]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "camlasercalibratool_amd", "csrc", "libclc_hip_stamps.so")
if "--build" in sys.argv:
    from camlasercalibratool_amd import _build as b
    b.build_variant(STAMPS_LIB, ["-DCLC_TEST_HOOKS", "-DCLC_STAMPS"] + (["-DCLC_STAMPS_WARM"] if "--warm" in sys.argv else []))
    print("built", STAMPS_LIB)
    sys.exit(0)
import ctypes as C
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, _capi

args = [a for a in sys.argv[1:] if not a.startswith("--")]
poses, pts = (int(args[0]), int(args[1])) if len(args) >= 2 else (2000, 500)
L = _capi.lib()
sv = clc.Solver(0)
S = sd.sim_fixed_count(1000, poses, pts, noise_sigma=0.01)
rec = clc.flatten_observations(S, False)
sv.upload(rec)
x0 = sd.pose7_from_T(np.eye(4))
for _ in range(5):
    r = sv.solve(x0, trace_cap=0)
NL, NW, NS = 64, 512, 16
buf = np.zeros((NL, NW, NS), dtype=np.uint64)
L.clc_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.clc_debug_stamps(buf.ctypes.data, buf.nbytes) == 0  # clears
r = sv.solve(x0, trace_cap=0)
assert L.clc_debug_stamps(buf.ctypes.data, buf.nbytes) == 0
npass = r.summary.num_evaluations
t = buf.astype(np.int64)
used = (t[:, :, 0] > 0)
nwg = int(used[1].sum())
print(f"obs={rec.shape[0]} passes={npass} workgroups={nwg}  (all times in us, 10 ns resolution)")
us = lambda a: a / 100.0
print("launch | first entry after prev launch's last end | entry spread | rows summed | barrier | controller | stream | reduce+store | last end - first entry | end spread")
for k in range(1, min(npass, NL)):
    m = used[k]
    if not m.any() or t[k, m, 5].min() == 0:
        continue
    e0 = t[k, m, 0]; e1 = t[k, m, 1]; e2 = t[k, m, 2]; e3 = t[k, m, 3]; e4 = t[k, m, 4]; e5 = t[k, m, 5]; e6 = t[k, m, 6]; e7 = t[k, m, 7]
    prev_end = max(t[k - 1, used[k - 1], 5].max(), t[k - 1, used[k - 1], 7].max()) if k >= 1 and t[k - 1, used[k - 1], 5].min() > 0 else 0
    gap = us(e0.min() - prev_end) if prev_end else float("nan")
    print(f"{k:3d} | gap {gap:5.2f} | entry spread {us(e0.max() - e0.min()):5.2f} | rows {np.median(us(e1 - e0)):5.2f} (max {us((e1 - e0).max()):5.2f}) | "
          f"barrier {np.median(us(e2 - e1)):5.2f} (max {us((e2 - e1).max()):5.2f}) | ctrl {np.median(us(e3 - e2)):5.2f} (max {us((e3 - e2).max()):5.2f}) | "
          f"stream {np.median(us(e4 - e3)):5.2f} (max {us((e4 - e3).max()):5.2f}) | red {np.median(us(e5 - e4)):5.2f} | "
          f"total {us(max(e5.max(), e7.max()) - e0.min()):5.2f} | "
          f"end spread {us(e5.max() - e5.min()):5.2f}")
    e8 = t[k, m, 8]; e9 = t[k, m, 9]; e10 = t[k, m, 10]; e11 = t[k, m, 11]; e12 = t[k, m, 12]
    cyc = (t[k, m, 14] - t[k, m, 13]).astype(np.float64)
    print(f"      controller section: combine {np.median(us(e11 - e2)):5.2f} | lm_advance {np.median(us(e12 - e11)):5.2f} = {np.median(cyc):.0f} clock64 ticks "
          f"({np.median(cyc / np.maximum(us(e12 - e11), 0.01)):.0f} ticks/us) | publish+barrier {np.median(us(e3 - e12)):5.2f} || wave 7 stream end vs wave 0: {np.median(us(e8 - e4)):+5.2f} | "
          f"butterfly {np.median(us(e9 - e4)):5.2f} | barrier wait {np.median(us(e10 - e9)):5.2f} | sum+store {np.median(us(e5 - e10)):5.2f}")
    p0 = t[k, 5, 15]; p1 = t[k, 5, 6]
    seg = [(int(p0) >> (16 * i)) & 0xFFFF for i in range(4)] + [(int(p1) >> (16 * i)) & 0xFFFF for i in range(4)]
    print("      controller cycles (workgroup 5): load+prelude %d | Hs/A %d | cholesky %d | solves %d | model %d | plus+gmax %d | finalize %d | publish %d (then barrier + write-back, off the other waves' path)" % tuple(seg))
if "--dump" in sys.argv:
    np.save(os.path.join(ROOT, "gpurun_out", "r02_stamps.npy"), t)
