#!/usr/bin/env python3
"""gpurun_out/prof_r03/ (scripts/profile_r03.sh) -> committed evidence under profiles/:
  r03_kernels.md            every kernel of every workload block of scripts/r03_workloads.py: rocprofv3 per-dispatch durations,
                            PMC HBM-side bytes (FETCH_SIZE / WRITE_SIZE, separate passes), VALU instruction counts
  r03_resident_traffic.json the batched solves: HBM-side bytes per batch on the resident kernel vs the round-2 paths
  r03_bench_kernel_stats.csv / r03_workloads_kernel_stats.csv   rocprofv3 --stats tables as written by the tool
  pmc_traffic.json          HBM bytes per launch of the dominant kernel of bench.py (step_kernel at C2), read by bench.py
PMC units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE tallies
16-byte-per-lane coalesced reads at half size, so read bytes = 2 x FETCH_SIZE x 1024 — calibrated inside this very run by the
64-byte-tile evaluation kernel, which reads exactly 64 B per observation (block `calib tiled64`)."""
import collections, csv, json, os, shutil, statistics

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r03")
dst = os.path.join(root, "profiles")
short = lambda k: k.split("(")[0].replace("void ", "").replace("clc::", "")


def plan_of(log):
    for line in open(log):
        if line.startswith('{"blocks"'):
            return json.loads(line)["blocks"]
    raise SystemExit(f"no plan in {log}")


def cut(rows, name_key, t_key):
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


def pmc_blocks(d):
    p = os.path.join(src, d, "w_counter_collection.csv")
    if not os.path.exists(p):
        return None
    disp = collections.OrderedDict()
    for r in sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"])):
        disp.setdefault(r["Dispatch_Id"], {"Kernel_Name": r["Kernel_Name"], "Start_Timestamp": r["Start_Timestamp"], "c": {}})["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return cut(list(disp.values()), "Kernel_Name", "Start_Timestamp")


plan = [b for b in plan_of(os.path.join(src, "w_trace.log")) if b["label"] != "end"]
trace = cut(list(csv.DictReader(open(os.path.join(src, "w_trace", "w_kernel_trace.csv")))), "Kernel_Name", "Start_Timestamp")
assert len(trace) == len(plan), (len(trace), len(plan))
fetch, write, valu = pmc_blocks("w_fetch"), pmc_blocks("w_write"), pmc_blocks("w_valu")
plan_valu = [b for b in plan_of(os.path.join(src, "w_valu.log")) if b["label"] != "end"] if valu is not None else []


def streaming(values):
    m = max(values)
    return [v for v in values if v >= 0.6 * m]


def sum_counter(blocks, i, counter, scale=1.0):
    """sum of a counter over ALL dispatches of block i (a batch's traffic is what every launch of it moved)"""
    if blocks is None or i >= len(blocks):
        return None
    return scale * sum(d["c"].get(counter, 0.0) for d in blocks[i])


def med_counter(blocks, i, kernel, counter):
    if blocks is None or i >= len(blocks):
        return None
    v = [d["c"].get(counter) for d in blocks[i] if d["Kernel_Name"] == kernel and counter in d["c"]]
    v = [x for x in v if x is not None]
    return statistics.median(streaming(v)) if v else None


lines = ["# rocprofv3 per-kernel durations by workload block — `scripts/r03_workloads.py` under `rocprofv3 --kernel-trace --stats` (MI355X, round 3)", "",
         "Blocks are cut at marker launches.  `streaming` = dispatches within 40 % of the block's longest.  PMC bytes: separate `--pmc FETCH_SIZE` /",
         "`--pmc WRITE_SIZE` passes of the same script; read = 2 x FETCH_SIZE x 1024 (gfx950 half-counting of 16 B/lane reads; block `calib tiled64` is the",
         "in-run calibration: it must read 64 B per observation), write = WRITE_SIZE x 1024.  For the batched blocks the PMC columns are per dispatch and the",
         "per-BATCH totals (all launches of one `clc_solve_batched`) are in the table at the end.", ""]
traffic, batched = {}, []
for i, (b, rows) in enumerate(zip(plan, trace)):
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines += [f"## {b['label']}", "", "| kernel | dispatches | avg us | streaming dispatches | avg us | median us | min | max | PMC read B | PMC write B |", "|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        if k.startswith("__amd_rocclr") and len(v) < 3:
            continue
        sv_ = streaming(v)
        fr, wr = med_counter(fetch, i, k, "FETCH_SIZE"), med_counter(write, i, k, "WRITE_SIZE")
        rd = 2 * fr * 1024 if fr is not None else None
        wb = wr * 1024 if wr is not None else None
        lines.append(f"| `{short(k)}` | {len(v)} | {statistics.mean(v):.2f} | {len(sv_)} | {statistics.mean(sv_):.2f} | {statistics.median(sv_):.2f} | {min(sv_):.2f} | {max(sv_):.2f} | "
                     f"{'' if rd is None else f'{rd:.4g}'} | {'' if wb is None else f'{wb:.4g}'} |")
    lines += ["", f"in-run measurements: `{json.dumps({k: v for k, v in b.items() if k not in ('label', 'kind')})}`", ""]
    if b["kind"] == "solve":
        steps = [k for k in per if "step_kernel" in k and ", 2, " in k]
        if steps:
            k = max(steps, key=lambda k: sum(per[k]))
            fr, wr = med_counter(fetch, i, k, "FETCH_SIZE"), med_counter(write, i, k, "WRITE_SIZE")
            v = streaming(per[k])
            if fr is not None and wr is not None:
                traffic = {"hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024, "read_bytes": 2 * fr * 1024, "write_bytes": wr * 1024,
                           "layout": "rows", "kernel": short(k), "rocprofv3_avg_us_streaming_launches": statistics.mean(v),
                           "rocprofv3_median_us_streaming_launches": statistics.median(v), "streaming_launches": len(v),
                           "streamed_bytes_by_layout": b["moved_bytes"], "algorithmic_bytes": 64 * b["obs"],
                           "source": "profiles/r03_kernels.md (block `solve 1000000`): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024, median over the launches that streamed"}
    if b["kind"] == "batched":
        n_solves = b["solves"]
        rd = sum_counter(fetch, i, "FETCH_SIZE", 2 * 1024.0)
        wb = sum_counter(write, i, "WRITE_SIZE", 1024.0)
        kernel_us = sum(sum(v) for k, v in per.items() if not k.startswith("__amd_rocclr"))
        mk = max(per, key=lambda k: sum(per[k]))
        batched.append({"block": b["label"], "problems": b["problems"], "observations": b["obs"], "resident": b["resident"],
                        "dominant_kernel": short(mk), "launches_per_batch": sum(len(v) for v in per.values()) / n_solves,
                        "kernel_time_us_per_batch": kernel_us / n_solves, "solve_batched_wall_ms": b["solve_ms"], "passes_per_problem": b["passes_total"] / b["problems"],
                        "bytes_of_one_pass_over_the_data": b["moved_bytes_per_pass_over_the_data"],
                        "pmc_read_bytes_per_batch": None if rd is None else rd / n_solves, "pmc_write_bytes_per_batch": None if wb is None else wb / n_solves,
                        "pmc_read_over_one_pass": None if rd is None else rd / n_solves / b["moved_bytes_per_pass_over_the_data"]})

lines += ["# Batched solves: HBM-side traffic per batch (sum over all launches of one `clc_solve_batched`)", "",
          "| block | problems | dominant kernel | launches / batch | kernel us / batch | wall ms | passes / problem | one pass over the data MB | PMC read MB / batch | read / one pass | PMC write MB / batch |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in batched:
    f = lambda x, s=1e6: "" if x is None else f"{x / s:.1f}"
    lines.append(f"| {r['block']} | {r['problems']} | `{r['dominant_kernel']}` | {r['launches_per_batch']:.1f} | {r['kernel_time_us_per_batch']:.0f} | {r['solve_batched_wall_ms']:.3f} | "
                 f"{r['passes_per_problem']:.2f} | {f(r['bytes_of_one_pass_over_the_data'])} | {f(r['pmc_read_bytes_per_batch'])} | "
                 f"{'' if r['pmc_read_over_one_pass'] is None else format(r['pmc_read_over_one_pass'], '.2f')} | {f(r['pmc_write_bytes_per_batch'])} |")
lines.append("")

if valu is not None:
    lines += ["# Instruction counts (`--pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS`, own pass, `--quick` sizes)", "",
              "Wave-level instruction counts, median per streaming dispatch; per observation-pass = x 64 lanes / (observations x passes).", "",
              "| block | kernel | VALU | FMA F64 | ADD F64 | MUL F64 | SALU | LDS | VALU / obs-pass | FP64 / obs-pass |", "|---|---|---|---|---|---|---|---|---|---|"]
    for i, b in enumerate(plan_valu):
        if i >= len(valu):
            continue
        names = collections.Counter(d["Kernel_Name"] for d in valu[i])
        for k in names:
            if not any(t in k for t in ("eval_rows_kernel", "eval_kernel", "step_kernel", "resident_solve_kernel", "batched_solve_kernel", "batched_rows_eval_kernel")):
                continue
            m = lambda c: med_counter(valu, i, k, c) or 0.0
            f64 = m("SQ_INSTS_VALU_FMA_F64") + m("SQ_INSTS_VALU_ADD_F64") + m("SQ_INSTS_VALU_MUL_F64")
            passes = b.get("passes_total", 0) / max(1, b.get("problems", 1)) if b["kind"] == "batched" and ("resident" in k or "batched_solve" in k) else 1.0
            denom = b["obs"] * max(passes, 1e-9)
            lines.append(f"| {b['label']} | `{short(k)}` | {m('SQ_INSTS_VALU'):.4g} | {m('SQ_INSTS_VALU_FMA_F64'):.4g} | {m('SQ_INSTS_VALU_ADD_F64'):.4g} | "
                         f"{m('SQ_INSTS_VALU_MUL_F64'):.4g} | {m('SQ_INSTS_SALU'):.4g} | {m('SQ_INSTS_LDS'):.4g} | {m('SQ_INSTS_VALU') * 64 / denom:.1f} | {f64 * 64 / denom:.1f} |")

os.makedirs(dst, exist_ok=True)
open(os.path.join(dst, "r03_kernels.md"), "w").write("\n".join(lines) + "\n")
json.dump(batched, open(os.path.join(dst, "r03_resident_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "w_trace", "w_kernel_stats.csv"), os.path.join(dst, "r03_workloads_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "b_trace", "b_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "b_trace", "b_kernel_stats.csv"), os.path.join(dst, "r03_bench_kernel_stats.csv"))
if traffic:
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines[-40:]))
print(json.dumps(traffic, indent=1))
