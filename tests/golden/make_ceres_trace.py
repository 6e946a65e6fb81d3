#!/usr/bin/env python3
"""Hand-off for pinning Ceres' trust-region trajectory.

Ceres is not installed in this image, so the LM controller (camlasercalibratool_amd/csrc/clc_lm.hpp) and the oracle
(oracle/clc_oracle.cpp) restate ceres::Solve from its documented behaviour.  This script writes

    tests/golden/ceres_trace_expected.json   per-iteration trace of the ORACLE (DENSE_QR restatement) on seeded problems:
                                             cost, cost_change, trust_region_radius, relative_decrease, step_norm,
                                             gradient_max_norm, step_is_successful, plus summary fields, and
    <outdir>/<case>.records.bin              (optional, --write-inputs DIR) the flat 64-byte records of every case +
                                             the start pose, as little-endian doubles: [N, pose0[7], records[N*8]],

so that anyone with a real Ceres (1.14 / 2.0 / 2.1) can run tests/golden/dump_ceres_trace.cpp on the same inputs and
diff summary.iterations against this file (tests/golden/compare_ceres_trace.py).  The committed JSON is a regression
fixture of the restatement (tests/test_golden_regression.py), NOT a pin of Ceres: it becomes one the day somebody runs
the dump program and the traces agree.

    python tests/golden/make_ceres_trace.py [--write-inputs DIR]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from camlasercalibratool_amd import simdata as sd  # noqa: E402

CASES = [  # name, generator, kwargs, start
    ("c1_seed1_sigma0.01", "GenerateSimData", dict(seed=1, noise_sigma=0.01), "identity"),
    ("c1_seed7_sigma0.03", "GenerateSimData", dict(seed=7, noise_sigma=0.03), "identity"),
    ("c1_seed0_noise_free", "GenerateSimData", dict(seed=0, noise_sigma=0.0), "identity"),
    ("c2_1e6_sigma0.01", "sim_fixed_count", dict(seed=1000, n_poses=2000, pts_per_pose=500, noise_sigma=0.01), "identity"),
]


def problem(gen, kw):
    S = getattr(sd, gen)(**kw)
    return oracle.flatten(S, False, False)  # calibr_simulation.cpp:130: CamLaserCalibration(obs, Tcl, false)


def main():
    outdir = sys.argv[sys.argv.index("--write-inputs") + 1] if "--write-inputs" in sys.argv else None
    out = {"provenance": "oracle/clc_oracle.cpp DENSE_QR restatement of ceres::Solve with the options of src/LaseCamCalCeres.cpp:302-304 "
                         "(Ceres defaults otherwise); see tests/golden/make_ceres_trace.py", "cases": []}
    for name, gen, kw, start in CASES:
        rec = problem(gen, kw)
        x0 = sd.pose7_from_T(np.eye(4))
        r = oracle.solve(rec, x0, linear_solver="qr")
        out["cases"].append({
            "name": name, "generator": gen, "kwargs": kw, "start": start, "n_residuals": int(rec.shape[0]), "pose0": x0.tolist(),
            "termination": int(r.summary.termination), "num_iterations": int(r.summary.num_iterations),
            "num_successful_steps": int(r.summary.num_successful_steps), "num_unsuccessful_steps": int(r.summary.num_unsuccessful_steps),
            "initial_cost": r.summary.initial_cost, "final_cost": r.summary.final_cost, "pose": r.pose.tolist(),
            "iterations": [dict(iteration=t.iteration, cost=t.cost, cost_change=t.cost_change, gradient_max_norm=t.gradient_max_norm,
                                step_norm=t.step_norm, relative_decrease=t.relative_decrease, trust_region_radius=t.trust_region_radius,
                                step_is_valid=t.step_is_valid, step_is_successful=t.step_is_successful) for t in r.trace]})
        if outdir:
            os.makedirs(outdir, exist_ok=True)
            np.concatenate([[float(rec.shape[0])], x0, rec.reshape(-1)]).astype("<f8").tofile(os.path.join(outdir, name + ".records.bin"))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ceres_trace_expected.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, [(c["name"], c["num_iterations"]) for c in out["cases"]])


if __name__ == "__main__":
    main()
