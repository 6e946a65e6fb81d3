#!/usr/bin/env python3
"""Freezes the ORACLE's answers for the SURVEY.md §8(d) input variants the GPU suite cannot afford to recompute on every run
 -> tests/golden/oracle_c4_shard.npz + tests/golden/oracle_variants.json:

  * C4: ALL 8 192 problems of the shard (seed 65536, global indices 3 x 8192 ...) that tests/test_gpu_parity.py::test_c4_full_size_shard
    solves on the GPU: {iterations, termination, final_cost, pose} of oracle.solve(DENSE_QR) per problem (~1 min on 8 cores);
  * C1: the simulation node's default (50 poses x 180 rays), seeds 0..9, noise-free and sigma = 0.01 m, init Tcl = I;
  * C2: 2 000 poses x 500 points, noise-free and sigma = 0.01 m, init Tcl = I and init = the closed form (the oracle's closed form:
    the start pose is stored, so the GPU solve starts from bit-identical numbers).

The inputs are pure functions of (seed, index) (camlasercalibratool_amd/simdata.py): only the outputs are stored.  The oracle is the
CPU restatement of the reference path (oracle/clc_oracle.cpp; Ceres' minimiser restated, unpinned — see its header).

    python tests/golden/make_c4_golden.py      # rewrite both fixtures
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from camlasercalibratool_amd import simdata as sd  # noqa: E402

C4 = dict(seed=65536, lo=3 * 8192, P=8192, n_poses=20, pts=500, sigma=0.01)
HERE = os.path.dirname(os.path.abspath(__file__))


def flatten(S, linefit=False):
    import oracle
    return oracle.flatten(S, linefit, False)


def c4_chunk(args):
    import oracle
    lo, hi = args
    rec, off, x0, gt = sd.sim_shard_records(C4["seed"], lo, hi, C4["n_poses"], C4["pts"], C4["sigma"])
    out = []
    for k in range(hi - lo):
        r = oracle.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
        out.append((r.summary.num_iterations, r.summary.termination, r.summary.final_cost, r.summary.initial_cost, r.pose))
    return out


def solve_record(rec, x0):
    import oracle
    r = oracle.solve(rec, x0, linear_solver="qr")
    return dict(start=[float(v) for v in x0], pose=[float(v) for v in r.pose], iterations=int(r.summary.num_iterations),
                termination=int(r.summary.termination), final_cost=float(r.summary.final_cost), initial_cost=float(r.summary.initial_cost),
                accepted=[int(t.step_is_successful) for t in r.trace], n=int(rec.shape[0]))


def main():
    import oracle
    oracle.build()
    lo, P = C4["lo"], C4["P"]
    chunks = [(a, min(a + 128, lo + P)) for a in range(lo, lo + P, 128)]
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = [r for part in ex.map(c4_chunk, chunks) for r in part]
    np.savez_compressed(os.path.join(HERE, "oracle_c4_shard.npz"),
                        iterations=np.array([r[0] for r in res], dtype=np.int16), termination=np.array([r[1] for r in res], dtype=np.int8),
                        final_cost=np.array([r[2] for r in res]), initial_cost=np.array([r[3] for r in res]), pose=np.array([r[4] for r in res]),
                        meta=np.array(json.dumps(dict(C4, oracle="oracle.solve(linear_solver='qr')", made_by="tests/golden/make_c4_golden.py"))))
    X0 = sd.pose7_from_T(np.eye(4))
    V = dict(provenance="oracle/clc_oracle.cpp (DENSE_QR restatement) on seeded simdata inputs; made by tests/golden/make_c4_golden.py", c1=[], c2=[])
    for seed in range(10):
        for sigma in (0.0, 0.01):
            rec = flatten(sd.GenerateSimData(seed, noise_sigma=sigma))
            V["c1"].append(dict(seed=seed, sigma=sigma, **solve_record(rec, X0)))
    for sigma in (0.0, 0.01):
        rec = flatten(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=sigma))
        V["c2"].append(dict(seed=1000, sigma=sigma, init="identity", **solve_record(rec, X0)))
        Tlc, unobs, _ = oracle.closed_form(rec)
        x_cf = sd.pose7_from_T(np.linalg.inv(Tlc))
        V["c2"].append(dict(seed=1000, sigma=sigma, init="closed_form", Tlc=[float(v) for v in Tlc.ravel()], unobservable=bool(unobs),
                            **solve_record(rec, x_cf)))
    json.dump(V, open(os.path.join(HERE, "oracle_variants.json"), "w"), indent=0)
    print("C4:", len(res), "problems, iterations", np.bincount([r[0] for r in res]).tolist(), "| C1:", len(V["c1"]), "| C2:", len(V["c2"]))


if __name__ == "__main__":
    main()
