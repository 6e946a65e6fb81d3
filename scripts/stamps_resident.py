#!/usr/bin/env python3
"""Where a workgroup of resident_solve_kernel spends its time: a -DCLC_STAMPS build (python scripts/stamps_resident.py --build
where hipcc is; loaded through CLC_LIBRARY) stamps kernel entry / exit (100 MHz wall clock) and, per pass, pass start, end of
the point loop, end of the wave reduction, barrier 1 passed, totals done, controller done / barrier 2 passed (shader clock)
for waves 0 and 1 of the first 1 024 problems.
usage (GPU box): CLC_LIBRARY=camlasercalibratool_amd/csrc/libclc_hip_stamps.so python scripts/stamps_resident.py [P] [flags-at-upload]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "camlasercalibratool_amd", "csrc", "libclc_hip_stamps.so")
if "--build" in sys.argv:
    from camlasercalibratool_amd import _build as b
    for base in (0, 4096):
        out = STAMPS_LIB if base == 0 else STAMPS_LIB.replace(".so", f"_{base}.so")
        b.build_variant(out, ["-DCLC_TEST_HOOKS", "-DCLC_STAMPS", f"-DCLC_RES_STAMP_BASE={base}"])
        print("built", out)
    sys.exit(0)
import ctypes as C
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd, _capi
args = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(args[0]) if args else 8192
up_flags = int(args[1]) if len(args) > 1 else -1
L = _capi.lib()
sv = clc.Solver(0)
rec, off, xb, gt = sd.sim_shard_records(65536, 0, P, 20, 500, 0.01)
sv.set_launch(0, up_flags)
sv.upload_batched(rec, off)
del rec
sv.set_launch(0, -1)
print("resident:", sv.debug_resident())
for _ in range(3):
    sv.solve_batched(xb)
NWG, NPASS = 1024, 8
buf = np.zeros((NWG, 2, 4 + 6 * NPASS), dtype=np.int64)
L.clc_debug_res_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.clc_debug_res_stamps(buf.ctypes.data, buf.nbytes) == 0  # clears
poses, sms = sv.solve_batched(xb)
assert L.clc_debug_res_stamps(buf.ctypes.data, buf.nbytes) == 0
n = min(P, NWG)
t = buf[:n]
life = (t[:, 0, 1] - t[:, 0, 0]) * 0.01  # us
print(f"workgroup lifetime us: median {np.median(life):.2f}  p10 {np.percentile(life,10):.2f}  p90 {np.percentile(life,90):.2f}")
span = (t[:, 0, 1].max() - t[:, 0, 0].min()) * 0.01
print(f"first {n} workgroups: first entry -> last exit {span:.1f} us")
passes = np.array([sms[k].num_evaluations for k in range(n)])
mhz = 1e-6 * np.median((t[:, 0, 4 + 6 * 1 + 5] - t[:, 0, 2])[passes >= 2] / np.maximum(1e-9, 1e-8 * 1.0)) if False else None
def med(a):
    a = a[np.isfinite(a)]
    return float(np.median(a)) if a.size else float("nan")
print(f"entry -> loads landed (cycles): wave0 {med((t[:,0,3]-t[:,0,2]).astype(float)):.0f}  wave1 {med((t[:,1,3]-t[:,1,2]).astype(float)):.0f}")
names = ["point loop (incl. plane/pose fetch)", "flush+corr+butterfly", "barrier 1 wait", "totals (w0) / pass start -> plane+pose arrived (w1)", "controller (w0) / barrier 2 wait (w1)"]
for p in range(min(NPASS, int(passes.max()))):
    sel = passes > p
    if sel.sum() == 0:
        break
    b = 4 + 6 * p
    row = []
    for w in (0, 1):
        d = [med((t[sel, w, b + i + 1] - t[sel, w, b + i]).astype(float)) for i in range(5)]
        if w == 1:  # slot b + 4 of the non-controller wave: plane + pose arrived (start of the point arithmetic)
            d = d[:3] + [med((t[sel, w, b + 4] - t[sel, w, b]).astype(float)), med((t[sel, w, b + 5] - t[sel, w, b + 3]).astype(float))]
        row.append(d)
    nxt = med((t[sel & (passes > p + 1), 0, b + 6] - t[sel & (passes > p + 1), 0, b + 5]).astype(float)) if p + 1 < NPASS else float("nan")
    tot = med((t[sel, 1, b + 5] - t[sel, 1, b]).astype(float))
    print(f"pass {p} ({int(sel.sum())} wgs) cycles  w0: " + " ".join(f"{x:7.0f}" for x in row[0]) + "   w1: " + " ".join(f"{x:7.0f}" for x in row[1]) +
          f"   | pass total (w1) {tot:.0f}  w0 ctrl end -> next pass start {nxt:.0f}")
print("columns:", names)
cb = np.zeros((NWG, 16), dtype=np.uint64)
L.clc_debug_res_ctrl_stamps.argtypes = [C.c_void_p, C.c_size_t]
if L.clc_debug_res_ctrl_stamps(cb.ctypes.data, cb.nbytes) == 0:
    w0, w1 = cb[:n, 15], cb[:n, 6]
    ok = w0 > 0
    ph = [((w0[ok] >> np.uint64(16 * i)) & np.uint64(0xFFFF)).astype(float) for i in range(4)] + [((w1[ok] >> np.uint64(16 * i)) & np.uint64(0xFFFF)).astype(float) for i in range(4)]
    lab = ["loads + acceptance + early stores", "scaling / damping", "Cholesky", "triangular solves", "model cost change", "two Plus + gradient norm", "tests + publish", "(slow path) + barrier"]
    print("controller phases of a workgroup's last pass (cycles, median): " + "; ".join(f"{l} {np.median(p):.0f}" for l, p in zip(lab, ph)) + f"; sum {sum(np.median(p) for p in ph):.0f}")
# shader clock rate estimate: cycles between entry and exit stamps vs wall clock
