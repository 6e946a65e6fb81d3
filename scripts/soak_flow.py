#!/usr/bin/env python3
"""Soak test of the round-6 paths: the reference-size flow (store -> select -> closed form -> solve -> analysis, host-planned uploads and the
selection cache) over alternating shapes and selections, multi-start launches, and pipelined gather steps — every result compared bit for
bit with the first result of the same case.  usage: soak_flow.py [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import calib, dist as cdist, simdata as sd
from camlasercalibratool_amd.solver import Comm
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sv = clc.Solver(0)
shapes = [sd.GenerateSimData(1, noise_sigma=0.01), sd.GenerateSimData(2, n_poses=100, noise_sigma=0.01), sd.sim_board_edges(5, 40, 60, noise_sigma=0.002),
          sd.sim_fixed_count(3, 9, 700, noise_sigma=0.01), sd.sim_fixed_count(4, 30, 500, noise_sigma=0.01)]  # (the last one: 15 000 records, the generic pipeline)
first, bad, n = {}, 0, 0
t0 = time.time()

def check(key, sig):
    global bad, n
    n += 1
    if key not in first:
        first[key] = sig
    elif first[key] != sig:
        bad += 1
        print("MISMATCH", key, flush=True)

# multi-start + pipelined gather fixtures
rec_ms = clc.flatten_observations(sd.sim_fixed_count(6, 20, 500, noise_sigma=0.01), False)
x_true = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
starts = sv.pose_plus(np.tile(x_true, (300, 1)), np.random.default_rng(0).normal(size=(300, 6)) * 0.03)
sh = clc.Solver(0)
rec_b, off_b, x0_b, _ = sd.sim_shard_records(5, 0, 64, 7, 80, 0.01)
sh.upload_batched(rec_b, off_b)
comm = Comm(sh, cdist.exchange_unique_id(0, 1), 0, 1)
comm.set_root(0)
for r in range(rounds):
    for k, S in enumerate(shapes):
        ses = calib.Session(S, sv)
        for lf, bd in (((True, False), (False, False)) if k != 2 else ((True, False), (True, True), (False, False))):
            Tlc = np.eye(4)
            ses.CamLaserCalClosedSolution(Tlc, verbose=False)
            Tcl = np.linalg.inv(Tlc)
            rep = ses.CamLaserCalibration(Tcl, lf, bd, verbose=False)
            check(("flow", k, lf, bd), (Tlc.tobytes(), Tcl.tobytes(), rep.result.summary.num_iterations, rep.chi2, rep.H.tobytes()))
    if r % 5 == 0:
        sv.upload_batched(rec_ms, np.array([0, rec_ms.shape[0]], dtype=np.int64))
        p, sm = sv.solve_multistart(starts)
        check(("multistart",), (p.tobytes(), tuple(s.num_iterations for s in sm)))
    prev, _ = comm.solve_gather_pipelined(x0_b, 0, 64)
    if prev is not None:
        check(("pipelined",), (prev[:, :11].tobytes(),))
last, _ = comm.flush(64)
check(("pipelined",), (last[:, :11].tobytes(),))
print(f"{n} results in {time.time() - t0:.1f} s over {rounds} rounds, mismatches: {bad}; host-planned uploads: "
      f"{'n/a (product build)' if not hasattr(sv._L, 'clc_debug_fast_small') else sv.debug_fast_small()}")
comm.close(); sh.close(); sv.close()
sys.exit(1 if bad else 0)
