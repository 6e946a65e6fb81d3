#!/usr/bin/env python3
"""gpurun_out/prof_r03c/ (scripts/profile_r03c.sh) -> committed evidence under profiles/:
  r03_coop.md                 rocprofv3 durations of coop_solve_kernel at C2 + PMC HBM-side bytes and VALU counts per launch
  r03_coop_traffic.json       HBM bytes per launch of the dominant kernel of bench.py (coop_solve_kernel at C2), read by bench.py
  r03_bench_kernel_stats.csv  rocprofv3 --stats table of `bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cold-start`
PMC units as in scripts/summarize_r03.py: read bytes = 2 x FETCH_SIZE x 1024 (gfx950 half-counts 16-byte-per-lane reads), write = WRITE_SIZE x 1024."""
import collections, csv, json, os, shutil, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r03c")
dst = os.path.join(root, "profiles")


def info(name):
    for line in open(os.path.join(src, name + ".log")):
        if line.startswith("{"):
            return json.loads(line)
    return {}


def durations(name, frag):
    rows = list(csv.DictReader(open(os.path.join(src, name, "w_kernel_trace.csv"))))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if frag in r["Kernel_Name"]]


def counters(name, frag):
    p = os.path.join(src, name, "w_counter_collection.csv")
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if frag in r["Kernel_Name"]:
            disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = disp.setdefault(r["Dispatch_Id"], {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(disp.values())


inf = info("coop_trace")
d = durations("coop_trace", "coop_solve_kernel")
fetch = [c["FETCH_SIZE"] for c in counters("coop_fetch", "coop_solve_kernel")]
write = [c["WRITE_SIZE"] for c in counters("coop_write", "coop_solve_kernel")]
valu = counters("coop_valu", "coop_solve_kernel")
rd = 2 * 1024 * statistics.median(fetch)
wr = 1024 * statistics.median(write)
passes = inf.get("passes", 0)
L = ["# rocprofv3 evidence for the cooperative solve, round 3 (MI355X) — `scripts/profile_r03c.sh`", "",
     f"Target: `scripts/r03_prof_probe.py coop 1000000` — 5 default `clc_solve` of the C2 problem (1e6 observations, {inf.get('points_per_lane')} points per lane, "
     f"{passes} evaluation passes per solve), each ONE launch of `coop_solve_kernel`; launches that timed out: {inf.get('aborts')}.", "",
     "| | |", "|---|---|",
     f"| `coop_solve_kernel` dispatches in the trace | {len(d)} |",
     f"| duration per launch, rocprofv3 kernel trace (us) | median {statistics.median(d):.1f}, min {min(d):.1f}, max {max(d):.1f} |",
     f"| per evaluation pass (us) | {statistics.median(d) / max(1, passes):.2f} |",
     f"| HBM-side read per launch (PMC FETCH_SIZE x 2 x 1024) | {rd / 1e6:.2f} MB  (lane layout + descriptors + planes: {inf.get('lane_layout_bytes', 0) / 1e6:.2f} MB) |",
     f"| HBM-side write per launch (PMC WRITE_SIZE x 1024) | {wr / 1e6:.3f} MB |",
     f"| read / lane layout bytes | {rd / max(1, inf.get('lane_layout_bytes', 1)):.2f} x  (one pass over the data per SOLVE; the step chain reads 17.4 MB per PASS) |"]
if valu:
    keys = sorted(valu[0].keys())
    med = {k: statistics.median(v[k] for v in valu) for k in keys}
    L += [f"| {k} per launch | {med[k]:.4g} |" for k in keys]
    if "SQ_INSTS_VALU" in med and passes:
        L.append(f"| VALU wave-instructions per observation and pass | {med['SQ_INSTS_VALU'] * 64 / (1e6 * passes):.1f} lane-instructions "
                 f"(the per-lane expansion, reduction and the 256 redundant controllers included) |")
for extra in ("coop_wait", "coop_wait2"):
    if os.path.exists(os.path.join(src, extra, "w_counter_collection.csv")):
        cw = counters(extra, "coop_solve_kernel")
        if cw:
            for k in sorted(cw[0].keys()):
                L.append(f"| {k} per launch (chip total; wave-cycle counters count quad-cycles) | {statistics.median(v[k] for v in cw):.4g} |")
            if extra == "coop_wait2" and "SQ_WAIT_ANY" in cw[0]:
                wc = counters("coop_wait", "coop_solve_kernel")
                if wc and "SQ_WAVE_CYCLES" in wc[0]:
                    L.append(f"| waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | {statistics.median(v['SQ_WAIT_ANY'] for v in cw) / statistics.median(v['SQ_WAVE_CYCLES'] for v in wc):.2f} "
                             f"— the launch is a latency chain: three of four waves sit at the controller's barrier, the fourth polls or runs the serial controller |")
L += ["", "PMC passes serialise kernels and slow the polling kernel down; durations are taken from the kernel-trace run only.", ""]
open(os.path.join(dst, "r03_coop.md"), "w").write("\n".join(L))
json.dump({"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "lane_layout_bytes": inf.get("lane_layout_bytes"), "passes": passes,
           "kernel_us_median": statistics.median(d),
           "valu_wave_instructions_per_launch": (statistics.median(v["SQ_INSTS_VALU"] for v in valu) if valu and "SQ_INSTS_VALU" in valu[0] else None),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/r03_prof_probe.py coop 1000000, median of 5 launches; read = 2 x FETCH_SIZE x 1024 (gfx950)"},
          open(os.path.join(dst, "r03_coop_traffic.json"), "w"), indent=1)
for f in (os.listdir(os.path.join(src, "bench_trace")) if os.path.isdir(os.path.join(src, "bench_trace")) else []):
    if f.endswith("kernel_stats.csv"):
        shutil.copy(os.path.join(src, "bench_trace", f), os.path.join(dst, "r03_bench_kernel_stats.csv"))
print("\n".join(L))
