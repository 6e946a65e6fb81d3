#!/usr/bin/env python3
"""Why does the row-layout evaluation kernel lose at 3 waves/SIMD beyond the Infinity Cache?  
    python scripts/r03_occupancy.py --build          -DCLC_EVAL_VARIANTS build -> csrc/libclc_hip_variants.so (where hipcc is)
    python scripts/r03_occupancy.py --one            one evaluation block on 3.2e7 observations with the variant named by
                                                     CLC_EVAL_VARIANT (unset: 512 threads x 8 rows in flight = the default;
                                                     512x4: 2 waves/SIMD, 4 rows; 768x4: 3 waves/SIMD, 4 rows); prints JSON
scripts/profile_r03_occupancy.sh runs --one under rocprofv3 (kernel trace, then PMC passes) for the three variants."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "camlasercalibratool_amd", "csrc", "libclc_hip_variants.so")
if "--build" in sys.argv:
    from camlasercalibratool_amd import _build as b
    b.build_variant(LIB, ["-DCLC_TEST_HOOKS", "-DCLC_EVAL_VARIANTS"])
    print("built", LIB)
    sys.exit(0)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
big = np.ascontiguousarray(np.tile(rec, (32, 1)))
sv = clc.Solver(0)
sv.upload(big)
n = big.shape[0]
del big, rec
_, n_rows, _, _ = sv.debug_rows()
x0 = sd.pose7_from_T(np.eye(4))
ms = [sv.time_eval(x0, reps=12) for _ in range(3)]
moved = n_rows * (64 * 16 + 64)
print(json.dumps({"variant": os.environ.get("CLC_EVAL_VARIANT", "512x8 (default)"), "observations": n, "moved_bytes": moved, "hipevent_us": [1e3 * m for m in ms],
                  "frac_moved_best": moved / (min(ms) * 1e-3) / 8e12}))
