#!/usr/bin/env python3
"""gpurun_out/prof_kernels/ (scripts/profile_kernels.sh) -> committed evidence under profiles/ (TAG = the round, CLC_PROFILE_TAG, default r06):
  r05_coop.md                 rocprofv3 durations of coop_solve_kernel at C2, PMC HBM-side bytes, instruction counts per launch AND per pass
                              (two launches with different pass counts), wait counters; the stamp table of scripts/stamps_coop.py is
                              appended from gpurun_out/r05_stamps.txt when present
  r05_coop_traffic.json       what bench.py reads: HBM bytes per launch, VALU wave-instructions per workgroup and pass
  r05_resident.md / r05_resident_traffic.json   the resident batched kernel at the C4 shard (8 192 problems x 1e4 observations)
  r05_bench_kernel_stats.csv  rocprofv3 --stats table of `bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-cold-start`
PMC units (MI355X_MICROARCH.md, HBM / rocprofv3 section): read bytes = 2 x FETCH_SIZE x 1024 (gfx950 half-counts 16-byte-per-lane reads;
calibrated against the 64-byte-tile kernel in profiles/r03_kernels.md: 1.00008), write = WRITE_SIZE x 1024; separate --pmc passes."""
import collections, csv, json, os, shutil, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("CLC_PROFILE_TAG", "r06")
src = os.path.join(root, "gpurun_out", "prof_kernels")
dst = os.path.join(root, "profiles")


def info(name):
    p = os.path.join(src, name + ".log")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                return json.loads(line)
    return {}


def durations(name, frag):
    p = os.path.join(src, name, "w_kernel_trace.csv")
    if not os.path.exists(p):
        return []
    rows = list(csv.DictReader(open(p)))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if frag in r["Kernel_Name"]]


def counters(name, frag):
    p = os.path.join(src, name, "w_counter_collection.csv")
    if not os.path.exists(p):
        return []
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if frag in r["Kernel_Name"]:
            d = disp.setdefault(r["Dispatch_Id"], {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(disp.values())


med = lambda rows, k: statistics.median(v[k] for v in rows)
K = "coop_solve_kernel"
inf = info("coop_trace")
L = []
if inf:
    d = durations("coop_trace", K)
    fetch = [c["FETCH_SIZE"] for c in counters("coop_fetch", K)]
    write = [c["WRITE_SIZE"] for c in counters("coop_write", K)]
    valu, valu4 = counters("coop_valu", K), counters("coop_valu4", K)
    inf4 = info("coop_valu4")
    rd, wr = 2 * 1024 * statistics.median(fetch), 1024 * statistics.median(write)
    passes = inf.get("passes", 0)
    L = [f"# rocprofv3 evidence for the cooperative solve, {TAG} (MI355X) — `scripts/profile_kernels.sh`", "",
         f"Target: `scripts/prof_probe.py coop 1000000` — {len(d)} (trace run; 5 in the counter runs) default `clc_solve` of the C2 problem (1e6 observations, {inf.get('points_per_lane')} points per lane, "
         f"{passes} evaluation passes per solve), each ONE launch of `coop_solve_kernel` (5 waves per workgroup: 4 point waves + the controller wave of "
         f"`csrc/clc_lmuni.hpp`); launches that timed out: {inf.get('aborts')}.", "",
         f"| | {TAG} | round 3 (`profiles/r03_coop.md`; rounds 4, 5: `profiles/r04_coop.md`, `profiles/r05_coop.md`) |", "|---|---|---|",
         f"| `coop_solve_kernel` dispatches in the trace | {len(d)} | 5 |",
         f"| duration per launch, rocprofv3 kernel trace (us) | median {statistics.median(d):.1f}, min {min(d):.1f}, max {max(d):.1f} | median 88.5, min 88.1, max 91.4 |",
         f"| per evaluation pass (us) | {statistics.median(d) / max(1, passes):.2f} | 6.81 |",
         f"| HBM-side read per launch (PMC FETCH_SIZE x 2 x 1024) | {rd / 1e6:.2f} MB  (lane layout + descriptors + planes: {inf.get('lane_layout_bytes', 0) / 1e6:.2f} MB) | 20.53 MB |",
         f"| HBM-side write per launch (PMC WRITE_SIZE x 1024) | {wr / 1e6:.3f} MB | 3.075 MB |",
         f"| read / lane layout bytes | {rd / max(1, inf.get('lane_layout_bytes', 1)):.2f} x  (one pass over the data per SOLVE) | 1.18 x |"]
    r03 = {"SQ_INSTS_LDS": 3.709e5, "SQ_INSTS_SALU": 1.107e6, "SQ_INSTS_VALU": 1.177e7, "SQ_INSTS_VALU_ADD_F64": 1.993e6, "SQ_INSTS_VALU_FMA_F64": 3.576e6,
           "SQ_INSTS_VALU_MUL_F64": 1.996e6, "SQ_BUSY_CYCLES": 6.693e6, "SQ_WAIT_INST_ANY": 1.145e6, "SQ_WAVES": 1024, "SQ_WAVE_CYCLES": 5.344e7,
           "SQ_ACTIVE_INST_ANY": 1.594e7, "SQ_ACTIVE_INST_VALU": 1.324e7, "SQ_WAIT_ANY": 3.685e7}
    per_pass = None
    if valu:
        keys = sorted(valu[0].keys())
        m = {k: med(valu, k) for k in keys}
        L += [f"| {k} per launch | {m[k]:.4g} | {r03.get(k, float('nan')):.4g} |" for k in keys]
        if valu4 and inf4.get("passes") and inf4["passes"] != passes:
            m4 = {k: med(valu4, k) for k in keys}
            dp = passes - inf4["passes"]
            per_pass = {k: (m[k] - m4[k]) / dp for k in keys}
            fixed = {k: m[k] - passes * per_pass[k] for k in keys}
            L += [f"| ... with max_num_iterations = 4 ({inf4['passes']} passes): SQ_INSTS_VALU per launch | {m4['SQ_INSTS_VALU']:.4g} | |",
                  f"| **SQ_INSTS_VALU per pass** (difference of the two launches / {dp} passes), chip / per workgroup | {per_pass['SQ_INSTS_VALU']:.4g} / "
                  f"{per_pass['SQ_INSTS_VALU'] / 256:.0f} | (46 000 / 3 540 per workgroup: 11.77e6 / 13 / 256) |",
                  f"| FP64 (FMA + ADD + MUL) per pass and workgroup | {(per_pass['SQ_INSTS_VALU_FMA_F64'] + per_pass['SQ_INSTS_VALU_ADD_F64'] + per_pass['SQ_INSTS_VALU_MUL_F64']) / 256:.0f} | |",
                  f"| outside the passes (entry, point loads, exit), per workgroup | {fixed['SQ_INSTS_VALU'] / 256:.0f} | |"]
        L.append(f"| VALU lane-instructions per observation and pass | {m['SQ_INSTS_VALU'] * 64 / (1e6 * passes):.1f} | 58.0 |")
    wc, wc2 = counters("coop_wait", K), counters("coop_wait2", K)
    for cw in (wc, wc2):
        if cw:
            for k in sorted(cw[0].keys()):
                L.append(f"| {k} per launch (chip total; wave-cycle counters count quad-cycles) | {med(cw, k):.4g} | {r03.get(k, float('nan')):.4g} |")
    if wc and wc2 and "SQ_WAIT_ANY" in wc2[0] and "SQ_WAVE_CYCLES" in wc[0]:
        L.append(f"| waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | {med(wc2, 'SQ_WAIT_ANY') / med(wc, 'SQ_WAVE_CYCLES'):.2f} | 0.69 |")
    L += ["", "PMC passes serialise kernels and slow the polling kernel down; durations are taken from the kernel-trace run only.", ""]
    st = os.path.join(root, "gpurun_out", "coop_stamps.txt")
    if os.path.exists(st):
        L += ["## Where a pass goes: shader-clock stamps of a `-DCLC_STAMPS` build (`scripts/stamps_coop.py`; the stamps themselves cost ~10 % of a pass)", "",
              "```", open(st).read().rstrip(), "```", ""]
    open(os.path.join(dst, TAG + "_coop.md"), "w").write("\n".join(L))
    json.dump({"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "lane_layout_bytes": inf.get("lane_layout_bytes"), "passes": passes,
               "kernel_us_median": statistics.median(d),
               "valu_wave_instructions_per_launch": (med(valu, "SQ_INSTS_VALU") if valu else None),
               "valu_wave_instructions_per_workgroup_and_pass": (per_pass["SQ_INSTS_VALU"] / 256 if per_pass else None),
               "points_per_lane": inf.get("points_per_lane"),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (separate passes) of scripts/prof_probe.py coop 1000000 [4], median of 5 launches; "
                         "read = 2 x FETCH_SIZE x 1024 (gfx950); per pass = difference of a 13-pass and a shorter launch"},
              open(os.path.join(dst, TAG + "_coop_traffic.json"), "w"), indent=1)

# ---- resident batched kernel at the C4 shard ----
KR = "resident_solve_kernel"
infr = info("res_fetch") or info("res_trace")
if infr:
    P = infr["n"]
    once = infr["lane_layout_bytes"]
    dr = durations("res_trace", KR)
    fr = [c["FETCH_SIZE"] for c in counters("res_fetch", KR)]
    wt = [c["WRITE_SIZE"] for c in counters("res_write", KR)]
    vr = counters("res_valu", KR)
    R = [f"# rocprofv3 evidence for the resident batched kernel, {TAG} (MI355X) — `scripts/profile_kernels.sh`", "",
         f"Target: `scripts/prof_probe.py resident {P}` — {P} problems x 10^4 observations (the C4 shard), 5 x `clc_solve_batched` on the default path "
         f"(ONE launch of `resident_solve_kernel` each); lane layout {once / 1e6:.1f} MB; {infr['passes_total'] / P:.2f} evaluation passes per problem.", "",
         "| | |", "|---|---|"]
    rec = {"problems": P, "bytes_of_one_pass_over_the_data": once, "passes_per_problem": infr["passes_total"] / P}
    if dr:
        R.append(f"| `resident_solve_kernel` duration per launch, rocprofv3 kernel trace (us) | median {statistics.median(dr):.1f}, min {min(dr):.1f}, max {max(dr):.1f} ({len(dr)} launches) |")
        rec["kernel_us_median"] = statistics.median(dr)
    if fr:
        rdr = 2 * 1024 * statistics.median(fr)
        R.append(f"| HBM-side read per launch (PMC FETCH_SIZE x 2 x 1024) | {rdr / 1e6:.1f} MB = **{rdr / once:.3f} x one pass over the data** |")
        rec.update(pmc_read_bytes_per_batch=rdr, pmc_read_over_one_pass=rdr / once)
    if wt:
        R.append(f"| HBM-side write per launch (PMC WRITE_SIZE x 1024) | {1024 * statistics.median(wt) / 1e6:.2f} MB |")
        rec["pmc_write_bytes_per_batch"] = 1024 * statistics.median(wt)
    if vr:
        obs_passes = infr["passes_total"] * 10000.0
        for k in sorted(vr[0].keys()):
            R.append(f"| {k} per launch | {med(vr, k):.4g} |")
        R.append(f"| VALU lane-instructions per observation-pass | {med(vr, 'SQ_INSTS_VALU') * 64 / obs_passes:.1f} |")
        R.append(f"| VALU wave-instructions per wave and pass (4 waves per problem) | {med(vr, 'SQ_INSTS_VALU') / infr['passes_total'] / 4:.0f} |")
        rec["valu_wave_instructions_per_wave_and_pass"] = med(vr, "SQ_INSTS_VALU") / infr["passes_total"] / 4
        if dr:
            fracv = med(vr, "SQ_INSTS_VALU") * 64 / (statistics.median(dr) * 1e-6) / (256 * 4 * 16 * 2.4e9)
            R.append(f"| VALU issue fraction (lane-instructions / kernel time / 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz) | {fracv:.3f} |")
            rec["valu_issue_frac"] = fracv
    R.append("")
    rs = os.path.join(root, "gpurun_out", "res_stamps.txt")
    if os.path.exists(rs):
        R += ["## Where a workgroup's time goes: shader-clock stamps of a `-DCLC_STAMPS` build (`scripts/stamps_resident.py`, workgroups 4 096-5 119 of the C4 shard)", "",
              "```", open(rs).read().rstrip(), "```", "",
              "Rounds 4-5 (same script, `profiles/r05_resident.md`): load of the problem 11 000 cycles; per pass: point loop 6 000, expansion + padding correction + wave "
              "reduction 1 900, barrier + totals 800, controller 5 000-6 700; lifetime 57 us per problem.", ""]
    step = os.path.join(dst, TAG + "_c4_step.json")  # (hand-kept: the step forms measured during the round)
    if os.path.exists(step):
        sj = json.load(open(step))
        R += ["## The C4 step (8 192 problems x 1e4 observations per GPU + the gather of the result records)", "",
              "| step form | ms per step (median of 5 blocks of 20) | kernel (HIP events) |", "|---|---|---|"]
        for row in sj["rows"]:
            R.append(f"| {row['form']} | {row['ms_per_step']:.4f}  (blocks {', '.join(f'{b:.3f}' for b in row['blocks'])}) | {row['kernel_ms']:.4f} |")
        R += ["", sj.get("note", ""), ""]
    open(os.path.join(dst, TAG + "_resident.md"), "w").write("\n".join(R))
    json.dump(rec, open(os.path.join(dst, TAG + "_resident_traffic.json"), "w"), indent=1)
    L += R
bs = os.path.join(src, "bench_trace", "b_kernel_stats.csv")
if os.path.exists(bs):
    shutil.copy(bs, os.path.join(dst, TAG + "_bench_kernel_stats.csv"))
# ---- what bench.py prices the whole-solve kernels with: the instruction counts + the identity of the sources they were measured on ----
import subprocess, sys
sys.path.insert(0, root)
from camlasercalibratool_amd import _build
vc = {"csrc_sha16": _build.csrc_sha16(),
      # the commit the profile was MEASURED on: --head <sha> when summarising later than measuring (gpurun sends the working tree, which has no .git)
      "head": (sys.argv[sys.argv.index("--head") + 1] if "--head" in sys.argv else
               subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip() or None),
      "source": "rocprofv3 --pmc SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/prof_probe.py: scripts/profile_kernels.sh -> scripts/summarize_kernels.py "
                f"(profiles/{TAG}_coop.md, {TAG}_resident.md)", "coop": {}, "resident": {}}
try:
    cj = json.load(open(os.path.join(dst, TAG + "_coop_traffic.json")))
    if cj.get("valu_wave_instructions_per_workgroup_and_pass"):
        per = cj["valu_wave_instructions_per_workgroup_and_pass"]
        vc["coop"][str(cj["points_per_lane"])] = {"valu_per_workgroup_pass": per, "valu_per_workgroup_once": cj["valu_wave_instructions_per_launch"] / 256 - cj["passes"] * per,
                                                  "hbm_bytes_per_launch": cj["hbm_bytes_per_launch"], "passes": cj["passes"], "kernel_us_median_rocprofv3": cj["kernel_us_median"]}
except (OSError, KeyError, TypeError) as e:
    print("valu_counts: no cooperative counts:", e)
try:
    rj = json.load(open(os.path.join(dst, TAG + "_resident_traffic.json")))
    if rj.get("valu_wave_instructions_per_wave_and_pass"):
        vc["resident"]["42"] = {"valu_per_wave_pass": rj["valu_wave_instructions_per_wave_and_pass"], "valu_per_point": 21.5, "lanes": 256,
                                "pmc_read_bytes_per_batch": rj.get("pmc_read_bytes_per_batch"), "problems": rj["problems"], "kernel_us_median_rocprofv3": rj.get("kernel_us_median")}
except (OSError, KeyError, TypeError) as e:
    print("valu_counts: no resident counts:", e)
if vc["coop"] or vc["resident"]:
    json.dump(vc, open(os.path.join(dst, "valu_counts.json"), "w"), indent=1)
    print("profiles/valu_counts.json:", json.dumps(vc)[:400])
print("\n".join(L))
