#!/usr/bin/env python3
"""Where a pass of coop_solve_kernel goes (C2 shape): a -DCLC_STAMPS build (python scripts/r03_coop_stamps.py --build where hipcc is)
stamps, per pass, on workgroups 0 and 7 (leaders), 8 and 255: pass start, wave totals in LDS, row published, group rows gathered,
group row published, the 8 group rows arrived, totals in LDS, controller done (shader clock, wave 0; wave 7 in the second row).
usage (GPU box): CLC_LIBRARY=camlasercalibratool_amd/csrc/libclc_hip_stamps.so python scripts/r03_coop_stamps.py [n_poses] [pts]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "camlasercalibratool_amd", "csrc", "libclc_hip_stamps.so")
if "--build" in sys.argv:
    from camlasercalibratool_amd import _build as b
    b.build_variant(STAMPS_LIB, ["-DCLC_TEST_HOOKS", "-DCLC_STAMPS"])
    print("built", STAMPS_LIB)
    sys.exit(0)
import ctypes as C
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, _capi
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_poses = int(args[0]) if args else 2000
pts = int(args[1]) if len(args) > 1 else 500
L = _capi.lib()
sv = clc.Solver(0)
rec = clc.flatten_observations(sd.sim_fixed_count(1000, n_poses, pts, noise_sigma=0.01), False)
sv.upload(rec)
x0 = sd.pose7_from_T(np.eye(4))
print("coop:", sv.debug_coop())
for _ in range(3):
    sv.solve(x0)
NP = 16
PP = 12
buf = np.zeros((4, 2, PP * NP + 2), dtype=np.int64)
L.clc_debug_coop_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.clc_debug_coop_stamps(buf.ctypes.data, buf.nbytes) == 0  # clears
o = clc.default_options(); o.profile_events = 2
r = sv.solve(x0, o)
assert L.clc_debug_coop_stamps(buf.ctypes.data, buf.nbytes) == 0
ne = r.summary.num_evaluations
print(f"evaluations {ne}, kernel {1e3 * r.summary.eval_kernel_ms:.1f} us")
names = ["pass (plane, pose, points, flush, butterfly)", "barrier + row published", "group rows gathered (leaders)", "group row published (leaders)",
         "8 group rows arrived", "barrier + totals", "controller (wave 0) / wait for it"]
wgs = ["wg 0 (leader)", "wg 7 (leader)", "wg 8", "wg 255"]
for w in range(4):
    t = buf[w, 0]
    print(f"--- {wgs[w]}: entry -> points in {t[PP*NP+1]-t[PP*NP]} cycles; first pass start {t[0]-t[PP*NP]} cycles after entry")
    per = []
    for p in range(min(ne, NP)):
        s = t[PP * p: PP * p + PP]
        d = [s[1] - s[0], s[2] - s[1], (s[3] - s[2]) if s[3] else 0, (s[4] - s[3]) if s[3] else 0, s[5] - (s[4] if s[3] else s[2]), s[6] - s[5], s[7] - s[6]]
        per.append(d)
    per = np.array(per, dtype=float)
    steady = per[2:] if len(per) > 3 else per
    for i, nm in enumerate(names):
        print(f"    {nm:48s} median {np.median(steady[:, i]):8.0f} cycles   (passes 0,1: {per[0, i]:.0f}, {per[1, i]:.0f})")
    tot = np.median(steady.sum(axis=1))
    full = np.median(np.diff(t[0:PP * min(ne, NP):PP]))
    inner = np.array([[t[PP*p+8]-t[PP*p], t[PP*p+9]-t[PP*p+8], t[PP*p+10]-t[PP*p+9], t[PP*p+1]-t[PP*p+10]] for p in range(2, min(ne, NP))], dtype=float)
    print('    inside the pass (median cycles): pose+plane %.0f, points %.0f, padding+expansion %.0f, butterfly %.0f' % tuple(np.median(inner, axis=0)))
    print(f"    sum {tot:.0f} cycles; pass start -> next pass start median {full:.0f} cycles")
span = buf[:, 0, PP * (min(ne, NP) - 1) + 7] - buf[:, 0, PP * NP]
print("entry -> last controller done (cycles):", span.tolist(), " => shader clock MHz if the kernel took all of it:", (span.max() / (1e3 * r.summary.eval_kernel_ms)))
