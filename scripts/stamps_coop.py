#!/usr/bin/env python3
"""Where a pass of coop_solve_kernel goes, round 4 (C2 shape): a -DCLC_STAMPS build (python scripts/stamps_coop.py --build where
hipcc is) stamps, per pass, wave 0 (and wave 3) of workgroups 0 and 7 (leaders), 8 and 255 with the shader clock — slots of
csrc/clc_coop.hpp: 0 pass start, 8 pose + plane set up, 9 points done, 10 expansion done, 1 partials in LDS, 2 row published (behind
barrier A), 3 group rows gathered / 4 group row published (leaders), 11 lmu_pre done (leaders: after 4), 5 the 8 group rows arrived,
6 totals in LDS, 7 lmu_post done (before barrier B).  Slots 0-2, 8-10 are point wave 0's, the others the controller wave's.
usage (GPU box): CLC_LIBRARY=camlasercalibratool_amd/csrc/libclc_hip_stamps.so python scripts/stamps_coop.py [n_poses] [pts]"""
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
    sv.solve(x0, trace_cap=0)
NP, PP = 16, 12
buf = np.zeros((4, 2, PP * NP + 2), dtype=np.int64)
L.clc_debug_coop_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.clc_debug_coop_stamps(buf.ctypes.data, buf.nbytes) == 0  # clears
o = clc.default_options(); o.profile_events = 2
r = sv.solve(x0, o, trace_cap=0)  # (no trace: its device stores would make workgroup 0 the slowest)
assert L.clc_debug_coop_stamps(buf.ctypes.data, buf.nbytes) == 0
ne = r.summary.num_evaluations
print(f"evaluations {ne}, kernel {1e3 * r.summary.eval_kernel_ms:.1f} us (stamped build)")
wgs = ["wg 0 (leader)", "wg 7 (leader)", "wg 8", "wg 255"]
names = ["pose + plane set-up", "points", "expansion (rows_flush)", "butterfly 28 -> 7 + LDS", "barrier A + row finished + published", "gather the group's 32 rows (leaders)",
         "group row summed + published (leaders)", "lmu_pre (leaders: behind their first look)", "8 group rows arrive (after lmu_pre)", "totals -> LDS", "lmu_post + barrier B", "-> next pass start"]
for w in range(4):
    t = buf[w, 0]
    t3 = buf[w, 1]
    print(f"--- {wgs[w]}: entry -> points in {t[PP*NP+1]-t[PP*NP]} cycles; first pass start {t[0]-t[PP*NP]} cycles after entry")
    per = []
    for p in range(1, min(ne, NP) - 1):
        s = t[PP * p: PP * p + PP]
        nxt = t[PP * (p + 1)]
        lead = s[3] != 0
        d = [s[8] - s[0], s[9] - s[8], s[10] - s[9], s[1] - s[10], s[2] - s[1], (s[3] - s[2]) if lead else 0, (s[4] - s[3]) if lead else 0,
             s[11] - (s[4] if lead else s[2]), s[5] - s[11], s[6] - s[5], s[7] - s[6], nxt - s[7]]
        per.append(d)
    per = np.array(per, dtype=float)
    for i, nm in enumerate(names):
        print(f"    {nm:44s} median {np.median(per[:, i]):8.0f} cycles   min {per[:, i].min():8.0f} max {per[:, i].max():8.0f}")
    full = np.median(np.diff(t[0:PP * min(ne, NP):PP]))
    w3 = [t3[PP * p + 1] - t3[PP * p] for p in range(1, min(ne, NP) - 1)]
    print(f"    sum {np.median(per.sum(axis=1)):.0f}; pass start -> next pass start median {full:.0f} cycles; wave 3 pass (start -> partials in LDS) median {np.median(w3):.0f}")
span = buf[:, 0, PP * (min(ne, NP) - 1) + 7] - buf[:, 0, PP * NP]
print("entry -> last controller done (cycles):", span.tolist(), " => shader clock MHz if the kernel took all of it:", (span.max() / (1e3 * r.summary.eval_kernel_ms)))
ck = np.zeros(16, dtype=np.int64)
L.clc_debug_lmregs_stamps.argtypes = [C.c_void_p, C.c_size_t]
if L.clc_debug_lmregs_stamps(ck.ctypes.data, ck.nbytes) == 0:
    nm = ["totals read (14 LDS), acceptance, selects, x / g / H at x", "scaling, damped diagonal", "Cholesky", "forward + backward substitution", "Plus (candidate)", "rotation + publish"]
    print("lmu_post of workgroup 8, a steady-state pass (cycles, each slot incl. its own stamp ~60):", ", ".join(f"{n} {ck[i+1]-ck[i]}" for i, n in enumerate(nm)), f"; total {ck[6]-ck[0]}")
# chronological timeline of one steady-state pass per stamped workgroup (cycles since that workgroup's pass start)
slot_names = {0: "pass start", 8: "plane set up", 9: "points done", 10: "expansion done", 1: "partials in LDS", 2: "row published", 3: "group rows gathered (leader)",
              4: "group row published (leader)", 11: "lmu_pre done", 5: "8 group rows arrived", 6: "totals in LDS", 7: "pose published (before barrier B)"}
p = min(6, ne - 2)
for w in range(4):
    t = buf[w, 0]
    s = t[PP * p: PP * p + PP]
    ev = sorted((int(s[k] - s[0]), slot_names[k]) for k in slot_names if s[k] != 0)
    nxt = int(t[PP * (p + 1)] - s[0])
    print(f"timeline {wgs[w]} pass {p}: " + " | ".join(f"{c} {n}" for c, n in ev) + f" | {nxt} next pass start")
