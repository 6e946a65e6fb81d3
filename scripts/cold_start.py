#!/usr/bin/env python3
"""Time to first result of the drop-in flow in a FRESH process — the reference's actual usage: main/calibr_offline.cpp:166-170
calls CamLaserCalClosedSolution + CamLaserCalibration exactly once per process, on O(10^2) board poses.
    python scripts/cold_start.py c1        simulation_lasercamcal_node's default problem (50 poses x ~114 points, 5.7e3 obs)
    python scripts/cold_start.py offline   the offline shape: 100 poses, every scan reduced to the 2 points of its fitted line
Prints ONE JSON line: wall time of clc_create (HIP runtime + context + stream + buffers), of the first closed form (first
upload + first kernel loads), of the first refinement + analysis pass, the same calls a second time (warm), and the CPU
oracle's time for the same calls.  bench.py runs this in a subprocess and reports it as `cold_start` (never part of `value`)."""
import json, os, sys, time
t_proc = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import calib, simdata as sd, _capi
shape = sys.argv[1] if len(sys.argv) > 1 else "c1"
want_cpu = "--no-cpu" not in sys.argv
linefit = shape == "offline"
if shape == "offline":
    S = sd.GenerateSimData(2, n_poses=100, noise_sigma=0.01)
else:
    S = sd.GenerateSimData(1, noise_sigma=0.01)
_capi.lib()  # dlopen of the extension (+ the HIP runtime it links): part of the cold path of a C++ caller too
t_loaded = time.perf_counter()
out = {"shape": shape, "poses": int(S.n_poses), "points": int(S.pts_off[-1]), "python_imports_and_data_s": t_loaded - t_proc}
t0 = time.perf_counter()
sv = clc.Solver(0)
t1 = time.perf_counter()
out["clc_create_ms"] = 1e3 * (t1 - t0)
def flow():
    ts = [time.perf_counter()]
    S2 = calib.points_on_fitted_lines(S, sv) if linefit else S  # calibr_offline.cpp:121-142 (GPU line fit per scan)
    ts.append(time.perf_counter())
    run = calib.Session(S2, sv)
    ts.append(time.perf_counter())
    Tlc = np.eye(4)
    run.CamLaserCalClosedSolution(Tlc, verbose=False)
    ts.append(time.perf_counter())
    Tcl = np.linalg.inv(Tlc)
    rep = run.CamLaserCalibration(Tcl, linefit, False, verbose=False)
    ts.append(time.perf_counter())
    d = np.diff(ts) * 1e3
    return {"line_fit_ms": float(d[0]), "store_ms": float(d[1]), "closed_form_ms": float(d[2]), "calibration_and_analysis_ms": float(d[3]),
            "total_ms": float(ts[-1] - ts[0]) * 1e3, "iterations": int(rep.result.summary.num_iterations)}, Tcl, S2
cold, Tcl_cold, S2 = flow()
warm, Tcl_warm, _ = flow()
out["first_call"] = cold
out["second_call_same_process"] = warm
out["time_to_first_result_ms"] = out["clc_create_ms"] + cold["total_ms"]
assert np.array_equal(Tcl_cold, Tcl_warm)
if want_cpu:
    import oracle
    rec_l = oracle.flatten(S2, True, False)
    rec = oracle.flatten(S2, linefit, False)
    t0 = time.perf_counter()
    T0, un, _ = oracle.closed_form(rec_l)
    t1 = time.perf_counter()
    x0 = sd.pose7_from_T(np.linalg.inv(T0))
    r = oracle.solve(rec, x0, linear_solver="qr")
    oracle.information(oracle.flatten(S2, linefit, False), r.pose)
    t2 = time.perf_counter()
    out["cpu_oracle"] = {"closed_form_ms": 1e3 * (t1 - t0), "calibration_and_analysis_ms": 1e3 * (t2 - t1), "total_ms": 1e3 * (t2 - t0),
                         "iterations": int(r.summary.num_iterations), "threads": 1,
                         "T_cl_max_abs_diff_vs_gpu": float(np.abs(sd.T_from_pose7(r.pose) - Tcl_cold).max())}
print(json.dumps(out))
