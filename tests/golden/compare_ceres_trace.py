#!/usr/bin/env python3
"""Diff a trace printed by tests/golden/dump_ceres_trace.cpp (real Ceres) against tests/golden/ceres_trace_expected.json.

    python tests/golden/compare_ceres_trace.py c1_seed1_sigma0.01 c1_seed1_sigma0.01.ceres.json

Exit status 0 when Ceres took the same path as the restatement: same iteration count, same accept/reject sequence, costs
and radii equal to 1e-9 relative, final pose within 1e-6 and final cost within 1e-8 (the gates of BASELINE.json)."""
import json
import os
import sys


def main():
    name, path = sys.argv[1], sys.argv[2]
    exp = next(c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ceres_trace_expected.json")))["cases"] if c["name"] == name)
    got = json.load(open(path))
    bad = []
    if got["num_iterations"] != exp["num_iterations"]:
        bad.append(f"iterations: ceres {got['num_iterations']} vs restatement {exp['num_iterations']}")
    for a, b in zip(got["iterations"], exp["iterations"]):
        if a["step_is_successful"] != b["step_is_successful"]:
            bad.append(f"iteration {b['iteration']}: accepted {a['step_is_successful']} vs {b['step_is_successful']}")
        for k in ("cost", "trust_region_radius"):
            if abs(a[k] - b[k]) > 1e-9 * max(abs(b[k]), 1e-300):
                bad.append(f"iteration {b['iteration']}: {k} {a[k]!r} vs {b[k]!r}")
    dpose = max(abs(x - y) for x, y in zip(got["pose"], exp["pose"]))
    dcost = abs(got["final_cost"] - exp["final_cost"])
    print(f"ceres {got.get('ceres_version')}: |dpose| {dpose:.3e} (gate 1e-6), |dcost| {dcost:.3e} (gate 1e-8), "
          f"iterations {got['num_iterations']} vs {exp['num_iterations']}")
    for b in bad:
        print("DIFF", b)
    sys.exit(0 if not bad and dpose <= 1e-6 and dcost <= 1e-8 else 1)


if __name__ == "__main__":
    main()
