#!/usr/bin/env python3
"""gpurun_out/r05_large_{plain,prof}.json + gpurun_out/prof_kernels/large/l_kernel_trace.csv (scripts/large_eval_probe.py, alone and under
rocprofv3 --kernel-trace) -> profiles/r05_large.md + profiles/r05_large_probe.json: the per-dispatch durations of the 3.2e7-observation
evaluation kernel IN ORDER — what the 83-vs-134 us spread of VERDICT r04 is."""
import csv, json, os, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")
plain = json.load(open(os.path.join(g, "r05_large_plain.json")))
prof = json.load(open(os.path.join(g, "r05_large_prof.json")))
rows = list(csv.DictReader(open(os.path.join(g, "prof_r05", "large", "l_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# phases are separated by the marker kernel (plus_kernel)
phases, cur = [], []
for r in rows:
    if "plus_kernel" in r["Kernel_Name"]:
        phases.append(cur); cur = []
    elif "eval_rows_kernel" in r["Kernel_Name"]:
        cur.append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])))
phases.append(cur)
phases = [p for p in phases if p]
B = plain["row_layout_bytes"]
frac = lambda us: B / (us * 1e-6) / 8e12
names = ["A: 30 groups of 10 right after the upload (+3 warm launches per group)", "B: 10 groups of 10 after 200 ms of idle", "C: 40 single launches, a synchronisation after each",
         "D: 10 groups of 100 back to back"]
L = ["# The 83-vs-134 us spread of the beyond-the-cache evaluation kernel (VERDICT r04, weak 4 / next 6) — round 5, MI355X", "",
     f"`scripts/large_eval_probe.py`: `eval_rows_kernel<true, true, 512, true, 12, false>` on {plain['observations']} observations = {B / 1e6:.0f} MB of rows per launch (beyond the 256 MiB "
     "Infinity Cache), every launch the same work.  Run alone (HIP-event means per group) and under `rocprofv3 --kernel-trace` (per-dispatch durations).", "",
     "**Finding: it is a power-management transient after an idle -> load transition, not the kernel and not the first touch of the allocation.**  The FIRST ~1 ms of "
     "streaming after any idle period runs at 84-85 us per launch; then the launches step to 100 us and, from ~2.3 to ~6.2 ms into the burst, to 118-126 us (the SMU pulls "
     "the clocks down under the new load), and over the following ~15-20 ms they converge to 85.5-86.5 us, where they stay for as long as the load lasts (phase D: 10 x 100 "
     "launches, 85.4-86.3 us).  "
     "The same shape appears right after the upload (phase A) and after 200 ms of idle on the warm allocation (phase B), so TLB misses / first touch are not it; single "
     "launches with a synchronisation in between (phase C, ~20 us gaps) sit in the tail of the same recovery.  Round 4's bench took 3 x 20 launches right after the upload — "
     "inside the transient — and reported the best; the rocprofv3 mean of that run (108 us) was the transient's average.", "",
     f"Sustained figure: **{statistics.median(plain['D_sustained_us']):.1f} us per launch = {B / (statistics.median(plain['D_sustained_us']) * 1e-6) / 1e9:.0f} GB/s = "
     f"{frac(statistics.median(plain['D_sustained_us'])):.2f} of the 8 TB/s HBM peak** (the guide's measured achievable rate is 6 290 GB/s = 0.79).  `bench.py` now streams "
     "300 launches before it measures and reports the median of 15 groups of 10 with min / max, plus the worst group of the transient as `after_idle_worst_frac`.", "",
     "## Group means by HIP events, in order (us per launch)", "", "| phase | alone | under rocprofv3 |", "|---|---|---|"]
for key, nm in zip(("A_first_groups_us", "B_after_idle_us", "C_single_launches_us", "D_sustained_us"), names):
    L.append(f"| {nm} | {' '.join(f'{v:.0f}' for v in plain[key])} | {' '.join(f'{v:.0f}' for v in prof[key])} |")
L += ["", "## Per-dispatch durations from the rocprofv3 kernel trace, in order", "",
      "| phase | dispatches | min | median | max | first 60 dispatches (us; each group = 3 warm + 10 timed launches) |", "|---|---|---|---|---|---|"]
for nm, p in zip(names, phases):
    d = [x[0] for x in p]
    L.append(f"| {nm.split(':')[0]} | {len(d)} | {min(d):.1f} | {statistics.median(d):.1f} | {max(d):.1f} | {' '.join(f'{v:.0f}' for v in d[:60])} |")
if phases:
    d0, t0 = [x[0] for x in phases[0]], phases[0][0][1]
    slow = [(x[1] - t0) / 1e6 for x in phases[0] if x[0] > 105]
    if slow:
        L += ["", f"Phase A: dispatches slower than 105 us lie between {min(slow):.2f} and {max(slow):.2f} ms after the phase's first dispatch "
                  f"({len(slow)} of {len(d0)}); the last 100 dispatches of the phase: median {statistics.median(d0[-100:]):.1f} us."]
open(os.path.join(root, "profiles", "r05_large.md"), "w").write("\n".join(L) + "\n")
json.dump({"alone": plain, "under_rocprofv3": prof}, open(os.path.join(root, "profiles", "r05_large_probe.json"), "w"), indent=1)
print("\n".join(L[-8:]))
