#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/profile_gpu.sh) into the committed evidence under
profiles/: the rocprofv3 --stats table, a per-kernel summary that separates real evaluation
launches from the no-op launches a finished solve leaves queued, and the PMC-derived HBM
traffic per launch (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the
bytes of 16-B/lane coalesced reads — MI355X_MICROARCH.md §HBM — so reads are doubled)."""
import collections, csv, json, os, shutil, statistics, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
name = sys.argv[2] if len(sys.argv) > 2 else tag

shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{name}_rocprofv3_kernel_stats.csv"))
rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = [f"# rocprofv3 summary `{name}` — `python bench.py --steps 50 --warmup 5 --no-cpu-baseline --large-obs 0`", "",
         "Kernel durations from `rocprofv3 --kernel-trace --stats` (us).  A finished solve leaves up to `lookahead` already-queued",
         "launches that exit on the device-side termination flag; they are listed separately (`no-op`: < 6 us for eval_kernel,",
         "< 2.5 us for lm_kernel, < 9 us for step_kernel<...,2> — that also sets aside each solve's final controller-only",
         "launch) so that the `real` average is the duration of launches that streamed the observations.", "",
         "| kernel | launches | avg us (all) | real launches | avg us (real) | median us (real) | min | max |", "|---|---|---|---|---|---|---|---|"]
summary = {}
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    thr = 6.0 if "eval_kernel" in k else (2.5 if "lm_kernel" in k else (9.0 if ("step_kernel" in k and k.rstrip().split("(")[0].rstrip().endswith("2>")) else 0.0))
    real = [x for x in v if x >= thr]
    short = k.split("(")[0].replace("void ", "")
    lines.append(f"| `{short}` | {len(v)} | {statistics.mean(v):.2f} | {len(real)} | {statistics.mean(real):.2f} | {statistics.median(real):.2f} | {min(real):.2f} | {max(real):.2f} |")
    summary[short] = {"launches": len(v), "avg_us_all": statistics.mean(v), "real": len(real), "avg_us_real": statistics.mean(real), "median_us_real": statistics.median(real)}

# PMC
def pmc(which):
    p = os.path.join(src, f"pmc_{which}", "pmc_counter_collection.csv")
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        out[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out
fetch, write = pmc("fetch"), pmc("write")
lines += ["", "## HBM traffic per launch from PMC (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes)", "",
          "`FETCH_SIZE`/`WRITE_SIZE` are KiB per dispatch.  HBM read bytes = 2 x FETCH_SIZE x 1024 (gfx950: wide coalesced reads are",
          "tallied at half size); write bytes = WRITE_SIZE x 1024.  Infinity-Cache hits are counted by these fabric-side counters.", "",
          "| kernel | class, layout | launches | FETCH_SIZE KiB (median) | read bytes (corrected) | WRITE_SIZE KiB | algorithmic bytes (64 B/obs) | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|"]
traffic = {}
for (k, cname), v in fetch.items():
    if "eval_kernel" not in k and "step_kernel" not in k:
        continue
    targs = [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
    batched = "batched_eval_kernel" in k
    step = "step_kernel" in k
    compact = True if step else ((len(targs) >= 2 and targs[1] == "true") if batched else (len(targs) >= 5 and targs[4] == "true"))
    groups = collections.defaultdict(list)
    for x in v:
        cls = "no-op" if x < 1000 else ("C3 batch, all 1024 problems running (1.024e7 obs)" if batched and x > 100000 else ("C3 batch, partly finished" if batched else ("1e6 obs" if x < 60000 else "8e6 obs")))
        groups[cls].append(x)
    wv = write.get((k, "WRITE_SIZE"), [0.0])
    for cls, g in sorted(groups.items()):
        if cls == "no-op":
            continue
        if cls == "C3 batch, partly finished":
            continue
        nobs = 1e6 if cls == "1e6 obs" else (1.024e7 if batched else 8e6)
        alg = 64 * nobs
        layout = (28 if compact else 64) * nobs
        rd = 2 * statistics.median(g) * 1024
        wr = statistics.median(wv) * 1024
        lines.append(f"| `{k}` | {cls}, {'compact 28 B/obs' if compact else 'tiles 64 B/obs'} | {len(g)} | {statistics.median(g):.1f} | {rd:.4g} | {statistics.median(wv):.1f} | {alg:.4g} | {(rd + wr) / alg:.3f} (vs layout bytes {layout:.3g}: {(rd + wr) / layout:.3f}) |")
        traffic[(cls, compact, "step" if step else ("batched" if batched else "eval"))] = {
            "read_bytes": rd, "write_bytes": wr, "algorithmic_bytes": alg, "layout_bytes": layout}
# instruction mix (optional pass)
vp = os.path.join(src, "pmc_valu", "pmc_counter_collection.csv")
if os.path.exists(vp):
    mix = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(vp)):
        mix[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["", "## Instruction mix per launch (`--pmc SQ_INSTS_VALU ..._FMA_F64 ..._ADD_F64 ..._MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS`, own pass)", "",
              "Wave-level instruction counts, median over the launches that streamed 10^6 observations (15 625 wave-tiles of 64 lanes x 2",
              "observations); `FP64 / obs` = (FMA + ADD + MUL F64) x 64 lanes / (2 x 10^6 observation slots) ... i.e. per observation and lane.", "",
              "| kernel | launches | VALU | FMA F64 | ADD F64 | MUL F64 | SALU | LDS | FP64 instructions per observation |", "|---|---|---|---|---|---|---|---|---|"]
    for kname, c in mix.items():
        if "step_kernel<true, false, 2>" not in kname and "eval_kernel<true, true, false, false, true, 512>" not in kname:
            continue
        valu = c.get("SQ_INSTS_VALU", [])
        big = [i for i, v in enumerate(valu) if v > 0.5 * max(valu)]  # launches that streamed (not the no-op / controller-only ones)
        med = lambda name: statistics.median([c[name][i] for i in big]) if c.get(name) else float("nan")
        f64 = med("SQ_INSTS_VALU_FMA_F64") + med("SQ_INSTS_VALU_ADD_F64") + med("SQ_INSTS_VALU_MUL_F64")
        lines.append(f"| `{kname}` | {len(big)} | {med('SQ_INSTS_VALU'):.4g} | {med('SQ_INSTS_VALU_FMA_F64'):.4g} | {med('SQ_INSTS_VALU_ADD_F64'):.4g} | "
                     f"{med('SQ_INSTS_VALU_MUL_F64'):.4g} | {med('SQ_INSTS_SALU'):.4g} | {med('SQ_INSTS_LDS'):.4g} | {f64 * 64 / 1e6 / 1.0:.1f} |")
bl = [l for l in open(os.path.join(src, "bench_plain.log")) if l.startswith("{")]
if bl:
    d = json.loads(bl[-1])
    lines += ["", "## bench line of the same command (un-profiled run on the same box)", "", "```json", json.dumps(d, indent=1), "```"]
    summary["bench"] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "roofline": d.get("roofline"), "roofline_large": d.get("roofline_large")}
open(os.path.join(dst, f"{name}_summary.md"), "w").write("\n".join(lines) + "\n")
# the dominant kernel of the bench command is step_kernel<true, false, 2>: its 1e6-obs launches come last in `fetch`
# only by accident of dict order, so pick them explicitly
key = next((k for k in (("1e6 obs", True, "step"), ("1e6 obs", True, "eval"), ("1e6 obs", False, "eval")) if k in traffic), None)
if key is not None:
    t = traffic[key]
    json.dump({"hbm_bytes_per_launch": t["read_bytes"] + t["write_bytes"], "read_bytes": t["read_bytes"], "write_bytes": t["write_bytes"],
               "layout": "compact" if key[1] else "tiled64", "kernel": "step_kernel" if key[2] == "step" else "eval_kernel",
               "source": f"profiles/{name}_summary.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024, median over launches",
               "all": {f"{c}/{'compact' if cp else 'tiled64'}/{kind}": v for (c, cp, kind), v in traffic.items()}},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, f"{name}_summary.json"), "w"), indent=1)
print("\n".join(lines[:40]))
