#!/usr/bin/env python3
"""Row share of a workgroup's wave 0 (the step kernel's controller wave) against the other seven: C2 / C1 / C5-like solve
time and step period for CLC_WAVE0_SHARE = a:b (one subprocess per setting: the library reads the variable once).
NEGATIVE RESULT (profiles/r03_wave0_share.md): the CLC_WAVE0_SHARE knob of wave_split_kernel was removed again after this
measurement (git history: commit "Fast per-point math in rows_point ... wave-0 share experiment (negative)"); kept as the record of how it was measured."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import camlasercalibratool_amd as clc
    from camlasercalibratool_amd import simdata as sd
    sv = clc.Solver(0)
    x0 = sd.pose7_from_T(np.eye(4))
    out = {"share": os.environ.get("CLC_WAVE0_SHARE", "default")}
    for name, poses, pts in (("c2", 2000, 500), ("n250k", 500, 500), ("c2_long_scans", 500, 2000)):
        rec = clc.flatten_observations(sd.sim_fixed_count(1000, poses, pts, noise_sigma=0.01), False)
        sv.upload(rec)
        for _ in range(20):
            r = sv.solve(x0, trace_cap=0)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            for _ in range(100):
                r = sv.solve(x0, trace_cap=0)
            ts.append((time.perf_counter() - t) / 100)
        passes = r.summary.num_evaluations
        step = min(sv.time_steps(x0, 2, passes - 1)[0] for _ in range(5))
        ev = min(sv.time_eval(x0, reps=200) for _ in range(3))
        out[name] = {"solve_ms": 1e3 * float(np.median(ts)), "step_us": 1e3 * step, "eval_alone_us": 1e3 * ev, "passes": int(passes)}
    print(json.dumps(out))
    sys.exit(0)
res = []
for share in sys.argv[1:] or ["6:6", "5:6", "4:6", "3:6", "2:6", "1:6", "0:6", "6:6"]:
    env = dict(os.environ, CLC_WAVE0_SHARE=share)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(share, "FAILED", p.stderr[-500:])
        continue
    d = json.loads(line[-1])
    res.append(d)
    print(share, " ".join(f"{k}: solve {v['solve_ms']:.4f} ms step {v['step_us']:.2f} us eval {v['eval_alone_us']:.2f} us" for k, v in d.items() if isinstance(v, dict)), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_wave0_share.json"), "w"), indent=1)
