#!/usr/bin/env python3
"""gpurun_out/prof_r02/ (scripts/profile_r02.sh) -> committed evidence under profiles/:
  r02_size_sweep.{csv,md}   evaluation kernel per layout x size: rocprofv3 duration, in-run hipEvent period, PMC HBM bytes
  r02_kernels.md            every kernel of every workload block (rocprofv3 per-dispatch durations), PMC traffic, FP64 mix
  r02_bench_kernel_stats.csv / r02_workloads_kernel_stats.csv   rocprofv3 --stats tables as written by the tool
  pmc_traffic.json          HBM bytes per launch of the dominant kernel of bench.py (step_kernel at C2), read by bench.py
PMC units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB per dispatch, collected in separate
passes; on gfx950 FETCH_SIZE tallies 16-byte-per-lane coalesced reads at half size, so read bytes = 2 x FETCH_SIZE x 1024
(calibration inside this very run: the 64-byte-tile kernel reads exactly 64 B per observation)."""
import collections, csv, json, os, shutil, statistics, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r02")
dst = os.path.join(root, "profiles")
short = lambda k: k.split("(")[0].replace("void ", "").replace("clc::", "")


def plan_of(log):
    for line in open(log):
        if line.startswith('{"blocks"'):
            return json.loads(line)["blocks"]
    raise SystemExit(f"no plan in {log}")


def cut(rows, name_key, t_key):
    """dispatch rows sorted by time -> list of blocks (lists of rows) separated by plus_kernel markers"""
    rows = sorted(rows, key=lambda r: int(r[t_key]))
    out, cur = [], None
    for r in rows:
        if "plus_kernel" in r[name_key]:
            if cur is not None:
                out.append(cur)
            cur = []
            continue
        if cur is not None:
            cur.append(r)
    return out


plan = plan_of(os.path.join(src, "w_trace.log"))
trace = cut(list(csv.DictReader(open(os.path.join(src, "w_trace", "w_kernel_trace.csv")))), "Kernel_Name", "Start_Timestamp")
assert len(trace) == len(plan), (len(trace), len(plan))


def pmc_blocks(d):
    p = os.path.join(src, d, "w_counter_collection.csv")
    if not os.path.exists(p):
        return None
    rows = list(csv.DictReader(open(p)))
    disp = collections.OrderedDict()  # one row per (dispatch, counter): cut on dispatches
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        disp.setdefault(r["Dispatch_Id"], {"Kernel_Name": r["Kernel_Name"], "Start_Timestamp": r["Start_Timestamp"], "c": {}})["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return cut(list(disp.values()), "Kernel_Name", "Start_Timestamp")


fetch, write, valu = pmc_blocks("w_fetch"), pmc_blocks("w_write"), pmc_blocks("w_valu")
plan_valu = plan_of(os.path.join(src, "w_valu.log")) if valu is not None else []


def main_kernel(rows):
    tot = collections.defaultdict(float)
    for r in rows:
        tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return max(tot, key=tot.get) if tot else None


def streaming(values):
    """launches that did the full work: within 40 % of the block's maximum (sets aside the no-op launches a finished solve
    leaves queued and the partly finished launches of a batch)"""
    m = max(values)
    return [v for v in values if v >= 0.6 * m]


def med_counter(blocks, i, kernel, counter):
    if blocks is None or i >= len(blocks):
        return None
    v = [d["c"].get(counter) for d in blocks[i] if d["Kernel_Name"] == kernel and counter in d["c"]]
    v = [x for x in v if x is not None]
    return statistics.median(streaming(v)) if v else None


sweep = [["obs", "layout", "kernel", "streamed_bytes", "rocprof_avg_us", "rocprof_median_us", "hipevent_us", "algorithmic_GBs_64B",
          "streamed_GBs", "frac_of_8TBs_streamed", "pmc_read_bytes", "pmc_write_bytes", "pmc_traffic_over_streamed"]]
kern_lines = ["# rocprofv3 per-kernel durations by workload block — `scripts/r02_workloads.py` under `rocprofv3 --kernel-trace --stats`", "",
              "Blocks are cut at marker launches; `streaming` = dispatches within 40 % of the block's longest (sets aside no-op launches queued",
              "behind a finished solve and late, mostly-terminated launches of a batch).  PMC bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`",
              "passes of the same script, median over the streaming dispatches, read = 2 x FETCH_SIZE x 1024 (gfx950 half-counting of 16 B/lane reads).", ""]
traffic = {}
for i, (b, rows) in enumerate(zip(plan, trace)):
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    kern_lines += [f"## {b['label']}", "", "| kernel | dispatches | avg us | streaming dispatches | avg us | median us | min | max | PMC read B | PMC write B |", "|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        if k.startswith("__amd_rocclr") and len(v) < 3:
            continue
        sv_ = streaming(v)
        fr, wr = med_counter(fetch, i, k, "FETCH_SIZE"), med_counter(write, i, k, "WRITE_SIZE")
        rd = 2 * fr * 1024 if fr is not None else None
        wb = wr * 1024 if wr is not None else None
        kern_lines.append(f"| `{short(k)}` | {len(v)} | {statistics.mean(v):.2f} | {len(sv_)} | {statistics.mean(sv_):.2f} | {statistics.median(sv_):.2f} | {min(sv_):.2f} | {max(sv_):.2f} | "
                          f"{'' if rd is None else f'{rd:.4g}'} | {'' if wb is None else f'{wb:.4g}'} |")
    extra = {k: v for k, v in b.items() if k not in ("label", "kind")}
    kern_lines += ["", f"in-run measurements: `{json.dumps(extra)}`", ""]
    mk = main_kernel(rows)
    if b["kind"] in ("eval", "batched_eval") and mk:
        v = streaming(per[mk])
        fr, wr = med_counter(fetch, i, mk, "FETCH_SIZE"), med_counter(write, i, mk, "WRITE_SIZE")
        rd = 2 * fr * 1024 if fr is not None else float("nan")
        wb = wr * 1024 if wr is not None else float("nan")
        us = statistics.mean(v)
        sweep.append([b["obs"], b.get("layout", "rows (batched)"), short(mk), b["streamed_bytes"], f"{us:.2f}", f"{statistics.median(v):.2f}", f"{b['hipevent_us']:.2f}",
                      f"{64 * b['obs'] / us / 1e3:.0f}", f"{b['streamed_bytes'] / us / 1e3:.0f}", f"{b['streamed_bytes'] / us / 1e3 / 8000:.3f}",
                      f"{rd:.4g}", f"{wb:.4g}", f"{(rd + wb) / b['streamed_bytes']:.3f}"])
    if b["kind"] == "solve" and b["obs"] == 1_000_000:
        steps = [k for k in per if "step_kernel" in k and ", 2, " in k]
        if steps:
            k = max(steps, key=lambda k: sum(per[k]))
            fr, wr = med_counter(fetch, i, k, "FETCH_SIZE"), med_counter(write, i, k, "WRITE_SIZE")
            v = streaming(per[k])
            traffic = {"hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024, "read_bytes": 2 * fr * 1024, "write_bytes": wr * 1024,
                       "layout": "rows", "kernel": short(k), "rocprofv3_avg_us_streaming_launches": statistics.mean(v),
                       "rocprofv3_median_us_streaming_launches": statistics.median(v), "streaming_launches": len(v),
                       "streamed_bytes_by_layout": b["streamed_bytes"], "algorithmic_bytes": 64 * b["obs"],
                       "source": "profiles/r02_kernels.md (block `solve 1000000`): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024, median over the launches that streamed"}

# FP64 instruction mix of the evaluation kernels at 1e6 obs (own PMC pass)
if valu is not None:
    kern_lines += ["# Instruction mix at 10^6 observations (`--pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS`, own pass)", "",
                   "Wave-level instruction counts per dispatch (median over streaming dispatches); FP64 per observation = (FMA + ADD + MUL) x 64 lanes / observations.", "",
                   "| block | kernel | VALU | FMA F64 | ADD F64 | MUL F64 | SALU | LDS | FP64 instr / observation |", "|---|---|---|---|---|---|---|---|---|"]
    for i, b in enumerate(plan_valu):
        if b["kind"] not in ("eval", "solve") or b["obs"] != 1_000_000 or i >= len(valu):
            continue
        names = collections.Counter(d["Kernel_Name"] for d in valu[i])
        for k in names:
            if not any(t in k for t in ("eval_rows_kernel", "eval_kernel", "step_kernel")):
                continue
            m = lambda c: med_counter(valu, i, k, c) or 0.0
            f64 = m("SQ_INSTS_VALU_FMA_F64") + m("SQ_INSTS_VALU_ADD_F64") + m("SQ_INSTS_VALU_MUL_F64")
            kern_lines.append(f"| {b['label']} | `{short(k)}` | {m('SQ_INSTS_VALU'):.4g} | {m('SQ_INSTS_VALU_FMA_F64'):.4g} | {m('SQ_INSTS_VALU_ADD_F64'):.4g} | "
                              f"{m('SQ_INSTS_VALU_MUL_F64'):.4g} | {m('SQ_INSTS_SALU'):.4g} | {m('SQ_INSTS_LDS'):.4g} | {f64 * 64 / b['obs']:.1f} |")

os.makedirs(dst, exist_ok=True)
with open(os.path.join(dst, "r02_size_sweep.csv"), "w") as f:
    csv.writer(f).writerows(sweep)
md = ["# Evaluation kernel across sizes and layouts (MI355X, round 2)", "",
      "`scripts/r02_workloads.py` under rocprofv3 (`scripts/profile_r02.sh`).  One row = one block of back-to-back launches of the evaluation kernel",
      "(K1) on a resident array: `rows` = the default row layout (16 B point + 64 B descriptor per row of 64 points), `compact28` = round 1's",
      "per-point compact layout (24 B point + 4 B group id), `tiled64` = the 64-byte records as handed over the C-ABI.  `rocprof` = mean duration of",
      "the dispatches (kernel trace), `hipEvent` = launch period measured inside the run (events around the block, launch gaps included),",
      "`streamed` = bytes the layout moves per launch, `alg` = the contract's 64 B per evaluation.  The 256 MiB Infinity Cache holds the row layout up to",
      "1.5e7 observations; PMC bytes are fabric-side requests (Infinity-Cache hits included).  The `rows` blocks run with the 3:2 old/young wave",
      "shares (flags 2|16|32|128|256) — what the library picks for the evaluation kernel alone once a wave owns more than 16 rows; at 10^6 observations",
      "and below the default is the equal split cut at scan starts (flag 512), 6.1-6.2 us instead of 6.7-7.2 (`scripts/r02_ab.py`).", "",
      "| observations | layout | streamed MB | rocprof us | hipEvent us | streamed GB/s | of 8 TB/s | alg GB/s (64 B) | PMC read MB | PMC write KB | PMC / streamed |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in sweep[1:]:
    md.append(f"| {int(r[0]):,} | {r[1]} | {int(r[3]) / 1e6:.1f} | {r[4]} | {r[6]} | {r[8]} | {r[9]} | {r[7]} | {float(r[10]) / 1e6:.1f} | {float(r[11]) / 1e3:.0f} | {r[12]} |")
open(os.path.join(dst, "r02_size_sweep.md"), "w").write("\n".join(md) + "\n")
open(os.path.join(dst, "r02_kernels.md"), "w").write("\n".join(kern_lines) + "\n")
shutil.copy(os.path.join(src, "w_trace", "w_kernel_stats.csv"), os.path.join(dst, "r02_workloads_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "b_trace", "b_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "b_trace", "b_kernel_stats.csv"), os.path.join(dst, "r02_bench_kernel_stats.csv"))
if traffic:
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
bl = [l for l in open(os.path.join(src, "bench_plain.log")) if l.startswith("{")]
if bl:
    json.dump(json.loads(bl[-1]), open(os.path.join(dst, "r02_bench_line.json"), "w"), indent=1)
print("\n".join(md))
print(json.dumps(traffic, indent=1))
