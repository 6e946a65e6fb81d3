#!/usr/bin/env python3
"""gpurun_out/prof_r03b/ (scripts/profile_r03b.sh: rocprofv3 runs of scripts/r03_prof_probe.py targets + the default bench command)
-> committed evidence under profiles/:
  r03_kernels.md             per-kernel rocprofv3 durations of every target, PMC HBM-side bytes (FETCH_SIZE / WRITE_SIZE, separate passes), VALU counts
  r03_resident_traffic.json  batched solves: HBM-side bytes per batch on the resident kernel vs the round-2 paths
  r03_bench_kernel_stats.csv rocprofv3 --stats table of `bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cold-start`
  pmc_traffic.json           HBM bytes per launch of the dominant kernel of bench.py (step_kernel at C2), read by bench.py
PMC units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE tallies 16-byte-per-lane
coalesced reads at half size, so read bytes = 2 x FETCH_SIZE x 1024 — calibrated in the same run by the 64-byte-tile evaluation kernel, which
reads exactly 64 B per observation (target `eval 32000000`, block tiled64)."""
import collections, csv, json, os, shutil, statistics

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r03b")
dst = os.path.join(root, "profiles")
short = lambda k: k.split("(")[0].replace("void ", "").replace("clc::", "")


def info(name):
    for line in open(os.path.join(src, name + ".log")):
        if line.startswith("{"):
            return json.loads(line)
    return {}


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
    if cur:
        out.append(cur)
    return out


def trace_blocks(name):
    p = os.path.join(src, name, "w_kernel_trace.csv")
    rows = list(csv.DictReader(open(p)))
    return cut(rows, "Kernel_Name", "Start_Timestamp"), rows


def pmc_blocks(name):
    p = os.path.join(src, name, "w_counter_collection.csv")
    if not os.path.exists(p):
        return None, None
    disp = collections.OrderedDict()
    for r in sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"])):
        disp.setdefault(r["Dispatch_Id"], {"Kernel_Name": r["Kernel_Name"], "Start_Timestamp": r["Start_Timestamp"], "c": {}})["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    allrows = list(disp.values())
    return cut(allrows, "Kernel_Name", "Start_Timestamp"), allrows


def table(rows):
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = ["| kernel | dispatches | total us | avg us | median us | min | max |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 0.002 * sum(sum(x) for x in per.values()) and len(per) > 6:
            continue
        out.append(f"| `{short(k)}` | {len(v)} | {sum(v):.1f} | {statistics.mean(v):.2f} | {statistics.median(v):.2f} | {min(v):.2f} | {max(v):.2f} |")
    return out, per


L = ["# rocprofv3 evidence, round 3 (MI355X) — `scripts/profile_r03b.sh`", "",
     "Every target is one run of `scripts/r03_prof_probe.py` under `rocprofv3 --kernel-trace --stats` and, separately, under `--pmc FETCH_SIZE`,",
     "`--pmc WRITE_SIZE` (and a VALU instruction pass for the batched solve).  Blocks inside a target are cut at marker launches (`plus_kernel`).",
     "PMC bytes: read = 2 x FETCH_SIZE x 1024 (gfx950 counts 16-byte-per-lane reads at half size — see the calibration block), write = WRITE_SIZE x 1024.", ""]
traffic_json = []

# ---- calibration + large evaluation ----
inf = info("eval_32000000_trace")
tb, _ = trace_blocks("eval_32000000_trace")
fb, _ = pmc_blocks("eval_32000000_fetch")
wb, _ = pmc_blocks("eval_32000000_write")
names = ["rows (default layout, 17.4 B / observation)", "tiled64 (64-byte records: the PMC calibration)"]
L += [f"## `eval 32000000`: the evaluation kernel beyond the Infinity Cache ({inf.get('observations')} observations)", ""]
calib = None
for i, nm in enumerate(names):
    if i >= len(tb):
        break
    t, per = table(tb[i])
    mk = max(per, key=lambda k: sum(per[k]))
    moved = inf["row_layout_bytes"] if i == 0 else 64 * inf["observations"]
    us = statistics.mean(per[mk])
    fr = statistics.median([d["c"]["FETCH_SIZE"] for d in fb[i] if d["Kernel_Name"] == mk]) if fb and i < len(fb) else None
    wr = statistics.median([d["c"]["WRITE_SIZE"] for d in wb[i] if d["Kernel_Name"] == mk]) if wb and i < len(wb) else None
    L += [f"### {nm}", ""] + t + ["", f"`{short(mk)}`: {us:.1f} us per launch (rocprofv3 mean), layout bytes {moved / 1e6:.1f} MB -> {moved / us / 1e3:.0f} GB/s moved = "
          f"**{moved / us / 1e3 / 8000:.3f} of the 8 TB/s HBM peak**; in-run hipEvent period {inf['rows_us' if i == 0 else 'tiled64_us']:.1f} us; "
          + (f"PMC read 2 x FETCH_SIZE x 1024 = {2 * fr * 1024 / 1e6:.1f} MB = {2 * fr * 1024 / moved:.3f} x the layout bytes" if fr is not None else "")
          + (f", write {wr * 1024 / 1e3:.0f} KB" if wr is not None else ""), ""]
    if i == 1 and fr is not None:
        calib = 2 * fr * 1024 / moved

# ---- step kernel at C2 ----
inf = info("step_1000000_trace")
_, rows = trace_blocks("step_1000000_trace")
t, per = table(rows)
L += ["## `step 1000000`: five `clc_solve` of C2 (10^6 observations) through the step-kernel chain", ""] + t + [""]
steps = [k for k in per if "step_kernel" in k and ", 2, " in k]
if steps:
    k = max(steps, key=lambda k: sum(per[k]))
    v = [x for x in per[k] if x >= 0.6 * max(per[k])]
    _, frows = pmc_blocks("step_1000000_fetch")
    _, wrows = pmc_blocks("step_1000000_write")
    med = lambda rows_, c: statistics.median(sorted(d["c"][c] for d in rows_ if d["Kernel_Name"] == k and d["c"].get(c, 0) >= 0.6 * max(e["c"].get(c, 0) for e in rows_ if e["Kernel_Name"] == k)))
    fr, wr = med(frows, "FETCH_SIZE"), med(wrows, "WRITE_SIZE")
    tj = {"hbm_bytes_per_launch": 2 * fr * 1024 + wr * 1024, "read_bytes": 2 * fr * 1024, "write_bytes": wr * 1024, "layout": "rows", "kernel": short(k),
          "rocprofv3_avg_us_streaming_launches": statistics.mean(v), "rocprofv3_median_us_streaming_launches": statistics.median(v), "streaming_launches": len(v),
          "streamed_bytes_by_layout": 17408000, "algorithmic_bytes": 64000000,
          "source": "profiles/r03_kernels.md (target `step 1000000`): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024, median over the launches that streamed"}
    json.dump(tj, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    L += [f"`{short(k)}` (steady-state launches that streamed, {len(v)}): mean {statistics.mean(v):.2f} us, median {statistics.median(v):.2f} us; PMC read {2 * fr * 1024 / 1e6:.2f} MB + write "
          f"{wr * 1024 / 1e3:.0f} KB per launch = {(2 * fr * 1024 + wr * 1024) / 17408000:.2f} x the 17.4 MB the row layout moves (+ 57 KB of partial rows per workgroup through L2).", ""]

# ---- batched solves ----
for target in ("resident_2048", "resident_8192"):
    inf = info(target + "_fetch")
    if not inf:
        continue
    P = inf["n"]
    L += [f"## `resident {P}`: {P} problems x 10^4 observations, 5 x `clc_solve_batched` on the resident kernel, then 5 x on the round-2 path (flag 4096)", "",
          f"lane layout {inf['lane_layout_bytes'] / 1e6:.1f} MB, row layout {inf['row_layout_bytes'] / 1e6:.1f} MB; {inf['passes_total'] / P:.2f} evaluation passes per problem", ""]
    tr = trace_blocks(target + "_trace")[0] if os.path.exists(os.path.join(src, target + "_trace")) else None
    fb, _ = pmc_blocks(target + "_fetch")
    wb, _ = pmc_blocks(target + "_write") if os.path.exists(os.path.join(src, target + "_write")) else (None, None)
    for i, (nm, once) in enumerate((("resident (default)", inf["lane_layout_bytes"]), ("round 2 (flag 4096)", inf["row_layout_bytes"]))):
        L += [f"### {nm}", ""]
        kernel_us = None
        if tr and i < len(tr):
            t, per = table(tr[i])
            L += t + [""]
            kernel_us = sum(sum(v) for k, v in per.items() if not k.startswith("__amd_rocclr")) / 5
        rd = 2 * 1024 * sum(d["c"].get("FETCH_SIZE", 0.0) for d in fb[i]) / 5 if fb and i < len(fb) else None
        wt = 1024 * sum(d["c"].get("WRITE_SIZE", 0.0) for d in wb[i]) / 5 if wb and i < len(wb) else None
        rec = {"problems": P, "path": nm, "bytes_of_one_pass_over_the_data": once, "pmc_read_bytes_per_batch": rd, "pmc_write_bytes_per_batch": wt,
               "pmc_read_over_one_pass": None if rd is None else rd / once, "kernel_us_per_batch": kernel_us, "passes_per_problem": inf["passes_total"] / P}
        traffic_json.append(rec)
        L += [f"per batch: " + (f"kernel time {kernel_us:.0f} us; " if kernel_us else "") + (f"**PMC read {rd / 1e6:.1f} MB = {rd / once:.2f} x one pass over the data**" if rd is not None else "")
              + (f", write {wt / 1e6:.2f} MB" if wt is not None else ""), ""]
    vb, _ = pmc_blocks(target + "_valu") if os.path.exists(os.path.join(src, target + "_valu")) else (None, None)
    if vb:
        L += ["### instruction counts (`--pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_LDS`)", "",
              "| block | kernel | dispatches | VALU | FMA F64 | ADD F64 | MUL F64 | SALU | LDS | VALU lane-instr / observation-pass | FP64 / observation-pass |", "|---|---|---|---|---|---|---|---|---|---|---|"]
        obs_passes = inf["passes_total"] * 10000.0
        for i, nm in enumerate(("resident", "round 2")):
            if i >= len(vb):
                break
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            cnt = collections.Counter()
            for d in vb[i]:
                cnt[d["Kernel_Name"]] += 1
                for c, x in d["c"].items():
                    agg[d["Kernel_Name"]][c] += x
            for k, c in agg.items():
                if not any(t in k for t in ("resident_solve", "batched_solve", "batched_rows_eval", "batched_lm")):
                    continue
                f64 = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]
                L.append(f"| {nm} | `{short(k)}` | {cnt[k]} | {c['SQ_INSTS_VALU']:.4g} | {c['SQ_INSTS_VALU_FMA_F64']:.4g} | {c['SQ_INSTS_VALU_ADD_F64']:.4g} | {c['SQ_INSTS_VALU_MUL_F64']:.4g} | "
                         f"{c['SQ_INSTS_SALU']:.4g} | {c['SQ_INSTS_LDS']:.4g} | {c['SQ_INSTS_VALU'] * 64 / 5 / obs_passes:.1f} | {f64 * 64 / 5 / obs_passes:.1f} |")
        L += ["", "(sums over the 5 batches of the block; per observation-pass = x 64 lanes / 5 / (problems x 10^4 x passes per problem))", ""]

os.makedirs(dst, exist_ok=True)
open(os.path.join(dst, "r03_kernels.md"), "w").write("\n".join(L) + "\n")
json.dump({"calibration_tiled64_read_over_bytes": calib, "batched": traffic_json}, open(os.path.join(dst, "r03_resident_traffic.json"), "w"), indent=1)
bs = os.path.join(src, "bench_trace", "b_kernel_stats.csv")
if os.path.exists(bs):
    shutil.copy(bs, os.path.join(dst, "r03_bench_kernel_stats.csv"))
print("\n".join(L))
