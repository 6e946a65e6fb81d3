#!/usr/bin/env python3
"""bench.py — headline benchmark of the point-to-plane extrinsic LM path on MI355X.

Contract:  python bench.py --gpus N --steps K --warmup W
  N>1: launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
  --gpus N ...` — or WITHOUT a launcher: `python bench.py --gpus N` then starts the N ranks itself (it never measures fewer GPUs than
  it reports: --gpus != WORLD_SIZE is an error).

Output: ONE JSON line of at most 4 KB on stdout (compact_line: the contract's fields + config + roofline + cpu_baseline + parity +
scale_base + one-number summaries of the side measurements under `side`); the full result — every sub-object of the side measurements —
goes to bench_detail.json beside this file (`detail_file` on the line; --detail-file moves it) and to stderr.

Workload (BASELINE.json configs[1], "C2"): a single T_cl problem with 10^6 synthetic
point-to-plane observations (2000 board poses x 500 scan points, range noise sigma = 0.01 m,
seeded restatement of simulation_lasercamcal_node's generator), initial guess Tcl = I,
Cauchy loss, Ceres-default LM options — exactly what CamLaserCalibration() runs.
A "step" is one complete solve (clc_solve) with the observation array already resident in HBM.
This is the N=1 line: BASELINE.json quotes the metric on C2, and C2 is a single problem — it does not shard
("replicas only").

N>1 (BASELINE.json configs[3], "C4"): 65 536 independent T_cl problems x 10^4 observations sharded over
8 GPUs, i.e. 8 192 problems per GPU — weak scaling, so --gpus N runs 8 192 x N problems.  Every rank GENERATES
ONLY ITS OWN contiguous shard (problem k is a pure function of (seed, k)), keeps it resident in HBM
(clc_upload_batched), and a step is ONE call, clc_solve_batched_gather (include/clc.h): the on-chip batched solve of the shard, whose
epilogue writes every problem's 12-double result record into the gather buffer, + ONE RCCL all-gather (in place) over xGMI of the
records of all 8 192 x N problems + the copy of the other ranks' records to the host, inside the timed region.
No collective on the data path.  At N=1 the same shard workload (with a world-size-1 RCCL communicator) is reported
as the `batched_c4_shard` sub-object (detail file; `side.c4_*` on the line), the like-for-like base of the 1 -> 8 curve: the N=1 line
carries it at top level as `scale_base` {workload, value, ms_per_step}, the N>1 lines carry `per_gpu_value` (= value / N).

Timed region: K steps bracketed by barrier + torch.cuda.synchronize() on both sides, `--blocks` (5) times over; every block's time is
the MAX over ranks; ms_per_step is the MEDIAN block (min / max / first block ride along).

`roofline` is the dominant kernel of the timed region.  Both whole-solve kernels (coop_solve_kernel at C2, resident_solve_kernel
at C4) read the data from HBM once per SOLVE and are bound by FP64 VALU issue / latency: their object has "bound": "valu_f64"
and `frac` = VALU lane-instructions (measured counts of profiles/valu_counts.json — used only when they belong to the kernel sources
in the tree — x this run's passes) / this run's kernel time / issue peak; `frac_moved` (bytes moved / time / 8 TB/s, bounded by 1) and
`contract_64B_frac` (the contract's 64 algorithmic bytes per evaluation, not a bandwidth) are siblings.  The streaming kernels'
objects (detail file) have "bound": "hbm" with `frac` = `frac_moved`.

value = residual+Jacobian evaluations per second, whole job:
        sum over ranks of (observations x evaluation passes) / max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured achievable
BYTES_PER_EVAL = 64     # one clc_observation record read once per evaluation
COMPACT_BYTES_PER_EVAL = 28  # the compact layout streams 24 B point + 4 B group id per observation
ROW_BYTES = 64 * 16 + 64     # the default (row) layout streams 1 KiB of (x, y) + a 64 B descriptor per row of 64 points
MIN_BYTES_PER_POINT = 16     # what the reference's own container holds per observation: (x, y) of an Eigen::Vector3d whose z is 0 ...
MIN_BYTES_PER_SCAN = 40      # ... plus one plane (4 doubles) and one scale per Oberserve (include/LaseCamCalCeres.h:11-24)
INFINITY_CACHE_BYTES = 256 * 2**20
VALU_PEAK_LANE_INSTR = 256 * 4 * 16 * 2.4e9  # 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz: FP64 (and any other) VALU issue peak
# The whole-solve kernels are priced against FP64 VALU issue with instruction counts that are constants of a BUILD (the pass's
# instruction stream does not depend on the data): PMC SQ_INSTS_VALU of the profile scripts (scripts/profile_kernels.sh ->
# scripts/summarize_kernels.py), kept in profiles/valu_counts.json together with the identity of the sources they were measured on
# (`csrc_sha16` = camlasercalibratool_amd/_build.csrc_sha16(), `head`).  When the sources in the tree differ from that identity the
# counts are NOT used: `roofline.frac` is null and `valu_issue.current` false, until the profile is regenerated.
VALU_COUNTS_FILE = os.path.join(ROOT, "profiles", "valu_counts.json")


def valu_counts(path=None, sha=None):
    """(counts, info): the measured instruction counts when they belong to the sources in the tree, else (None, info) —
    info = {source, source_head, csrc_sha16, current} rides on the line as roofline.valu_issue."""
    from camlasercalibratool_amd import _build
    sha = sha or _build.csrc_sha16()
    try:
        d = json.load(open(path or VALU_COUNTS_FILE))
    except (OSError, ValueError) as e:
        return None, {"source": "profiles/valu_counts.json", "current": False, "csrc_sha16": sha, "error": repr(e)}
    info = {"source": "profiles/valu_counts.json (rocprofv3 --pmc SQ_INSTS_VALU)", "source_head": d.get("head"), "csrc_sha16": d.get("csrc_sha16"),
            "current": d.get("csrc_sha16") == sha}
    if not info["current"]:
        info["tree_csrc_sha16"] = sha
        print(f"bench.py: profiles/valu_counts.json was measured on sources {d.get('csrc_sha16')} but the tree is {sha}: VALU counts "
              "not used (roofline.frac = null); regenerate with scripts/profile_kernels.sh + scripts/summarize_kernels.py", file=sys.stderr)
        return None, info
    return d, info


def hbm_figures(moved_bytes, contract_bytes, sec):
    """The figures of an HBM-bound launch: `achieved` / `frac` price the bytes the layout actually MOVES (a bandwidth, bounded by 1);
    the contract's pricing — 64 algorithmic bytes per residual+Jacobian evaluation, which a lossless 3.7x smaller
    layout or data held on chip can exceed the peak with — is kept next to it as contract_64B_*."""
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": moved_bytes / sec / 1e9, "frac": moved_bytes / sec / 1e9 / HBM_PEAK_GBS,
            "frac_moved": moved_bytes / sec / 1e9 / HBM_PEAK_GBS, "moved_bytes_per_launch": int(moved_bytes),
            "algorithmic_bytes_per_launch": int(contract_bytes), "contract_64B_GBps": contract_bytes / sec / 1e9,
            "contract_64B_frac": contract_bytes / sec / 1e9 / HBM_PEAK_GBS}


def pricings(n_obs, n_scans, moved_bytes, seconds):
    """The three ways one pass over n_obs observations is priced, each as bytes, GB/s and fraction of the 8 TB/s HBM peak:
    `contract` = 64 B per residual+Jacobian evaluation (one clc_observation record); `reference_container_min`
    = the bytes the reference's own Oberserve container holds for them (16 B of (x, y) per point, z == 0, + 40 B per scan);
    `moved` = the bytes the layout streamed actually moves.  Only `moved` is a bandwidth; it is bounded by 1."""
    out = {}
    for name, b in (("contract_64B_per_eval", BYTES_PER_EVAL * n_obs),
                    ("reference_container_min", MIN_BYTES_PER_POINT * n_obs + MIN_BYTES_PER_SCAN * n_scans), ("moved", moved_bytes)):
        out[name] = {"bytes": int(b), "GBps": b / seconds / 1e9, "frac_of_hbm_peak": b / seconds / 1e9 / HBM_PEAK_GBS}
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--poses", type=int, default=2000)
    ap.add_argument("--pts", type=int, default=500)
    ap.add_argument("--noise", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-all-ranks", action="store_true", help="N>1 step: ncclAllGather + a host copy of every record on EVERY rank (default: gather to rank 0)")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1 step: the plain one-call step instead of the pipelined stream of steps")
    ap.add_argument("--kernel-events", action="store_true",
                    help="bracket every evaluation-kernel launch with HIP event pairs INSIDE the timed region "
                         "(perturbs the timed solves; default: events are used after the timed region)")
    ap.add_argument("--large-obs", type=int, default=32_000_000,
                    help="extra evaluation-kernel measurement on a working set beyond the Infinity Cache in the layout streamed "
                         "(3.2e7 obs = 544 MB of rows; 0 disables)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-batched", action="store_true", help="skip the extra C3 (1024 problems x 1e4 obs) measurement")
    ap.add_argument("--problems-per-gpu", type=int, default=8192, help="C4 shard: independent problems per GPU (N>1 workload)")
    ap.add_argument("--problems-total", type=int, default=0,
                    help="N>1 dry runs: total number of problems (default problems-per-gpu x N); a count not divisible by N gives "
                         "shards that differ by one and padding records in the gather")
    ap.add_argument("--shard-poses", type=int, default=20)
    ap.add_argument("--shard-pts", type=int, default=500)
    ap.add_argument("--shard-steps", type=int, default=20, help="timed steps of the batched_c4_shard sub-object at N=1")
    ap.add_argument("--no-c4-shard", action="store_true", help="N=1: skip the batched_c4_shard sub-object")
    ap.add_argument("--no-cold-start", action="store_true", help="N=1: skip the fresh-process cold_start measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group for N>1 (nccl = RCCL; gloo only for dry runs: the result records are then gathered "
                         "with torch.distributed instead of clc_gather_results)")
    ap.add_argument("--detail-file", default=None, help="where the full result goes (default: bench_detail.json beside bench.py)")
    ap.add_argument("--blocks", type=int, default=5,
                    help="the K-step timed region is repeated this many times (each bracketed by barrier + synchronize, max over ranks); "
                         "ms_per_step is the median block")
    ap.add_argument("--two-call-step", action="store_true",
                    help="C4 step as clc_solve_batched + clc_gather_results (the round-4 form) instead of the one-call clc_solve_batched_gather")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="testing only: map ranks onto the visible GPUs modulo their count (with --backend gloo)")
    return ap.parse_args()


C4_SEED = 65536


def hooks_solver(clc, device):
    """A second handle on the -DCLC_TEST_HOOKS build of the library (same translation units + clc_time_* / clc_debug_*), for the
    kernel-only measurements next to the timed region — the timed region itself runs on the product library, which does not
    export them.  None when that build is absent."""
    try:
        return clc.Solver(device, library="hooks")
    except Exception as e:  # the side measurements are optional
        print(f"bench.py: no hooks build ({e}); kernel-only sub-objects skipped", file=sys.stderr)
        return None


def resident_kernel_report(clc, solver, x_start, n_problems, n_obs_total, n_scans_total):
    """Roofline object of the batched solve when it runs as ONE launch of resident_solve_kernel (csrc/clc_resident.hpp): every
    problem is read from HBM once and solved on chip, so the launch is bound by VALU issue (FP64), not by HBM.  The kernel
    time is a HIP event pair around the launch on the solver's stream (clc_options.profile_events = 1), best of 5."""
    ok, lanes, ppl, rows = solver.debug_resident()
    if not ok:
        return None
    o = clc.default_options()
    o.profile_events = 1
    best, sms = None, None
    for _ in range(5):
        _, sm = solver.solve_batched(x_start, o)
        if best is None or sm[0].eval_kernel_ms < best:
            best, sms = sm[0].eval_kernel_ms, sm
    passes = int(sum(sms[k].num_evaluations for k in range(n_problems)))
    per = n_obs_total // max(1, n_problems)
    evals = float(sum(sms[k].num_evaluations for k in range(n_problems))) * per
    moved = rows * lanes * 16 + n_problems * lanes * 8 + n_scans_total * 48  # point rows + lane descriptors + plane table, once
    sec = best * 1e-3
    rep = {"bound": "valu_f64", "peak": VALU_PEAK_LANE_INSTR / 1e12, "unit": "T VALU lane-instructions/s", "achieved": None, "frac": None,
           "kernel": f"clc::resident_solve_kernel<loss=1, {lanes // 64} waves per problem> (whole LM solve of every problem in ONE launch; "
                     f"{lanes} lanes x {ppl} points per problem held in registers + LDS)",
           "launches_per_batch": 1, "avg_kernel_ms": best, "evaluation_passes": passes,
           "moved_bytes_per_launch": int(moved), "achieved_moved_GBps": moved / sec / 1e9, "frac_moved": moved / sec / 1e9 / HBM_PEAK_GBS,
           "served_from": "infinity_cache" if moved <= INFINITY_CACHE_BYTES else "hbm", "traffic_measured_in_run": False, "traffic": None,
           "hbm_passes_over_the_data_per_solve": 1,
           "algorithmic_bytes_per_launch": BYTES_PER_EVAL * evals, "contract_64B_GBps": BYTES_PER_EVAL * evals / sec / 1e9,
           "contract_64B_frac": BYTES_PER_EVAL * evals / sec / 1e9 / HBM_PEAK_GBS,
           "pricings": pricings(evals, n_scans_total * passes / max(1, n_problems), moved, sec),
           "timing": "hipEvent pair around the launch on the solver's stream (profile_events = 1), best of 5",
           "note": "the data crosses HBM once per SOLVE, not once per pass (`frac_moved`, bounded by 1); the launch is bound by FP64 VALU issue: "
                   "`frac` = VALU lane-instructions / kernel time / issue peak.  `contract_64B_frac` prices the launch at the contract's 64 algorithmic "
                   "bytes per evaluation x all passes and is not a bandwidth"}
    counts, vinfo = valu_counts()
    table = (counts or {}).get("resident") or {}
    rc, exact = table.get(str(ppl)), True
    if rc is None and table:  # another problem size: the nearest measured count, moved along its per-point slope (an estimate, flagged)
        near = min(table, key=lambda k: abs(int(k) - ppl))
        rc, exact = dict(table[near]), False
        rc["valu_per_wave_pass"] = float(rc["valu_per_wave_pass"]) + float(rc.get("valu_per_point", 21.5)) * (ppl - int(near))
    rep["valu_issue"] = dict(vinfo, exact=exact)
    if rc is not None and lanes == rc.get("lanes", 256):
        if exact and n_problems == rc.get("problems"):
            rep["traffic"] = rc.get("pmc_read_bytes_per_batch")
            rep["traffic_source"] = "profiles/valu_counts.json (rocprofv3 --pmc FETCH_SIZE of the builder's run on the same sources)"
        per_wave_pass = float(rc["valu_per_wave_pass"])
        lane_instr = passes * (lanes / 64.0) * per_wave_pass * 64
        rep["achieved"] = lane_instr / sec / 1e12
        rep["frac"] = lane_instr / sec / VALU_PEAK_LANE_INSTR
        rep["valu_issue"].update({"per_pass": per_wave_pass, "lane_instructions": lane_instr, "peak_lane_instructions_per_s": VALU_PEAK_LANE_INSTR,
                                  "note": "VALU wave-instructions per wave and pass (controller's share included) x waves x this run's passes x 64 lanes, "
                                          "over this run's kernel time, against 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz"})
    elif counts is not None:
        rep["valu_issue"]["current"] = False
        rep["valu_issue"]["error"] = f"no count for {ppl} points per lane x {lanes} lanes in profiles/valu_counts.json"
    return rep


COOP_WGS, COOP_LANES = 256, 256  # csrc/clc_coop.hpp: workgroups of the cooperative solve, lanes per workgroup


def coop_kernel_report(clc, solver, x0, n_obs, n_scans):
    """Roofline object of clc_solve when it runs as ONE launch of coop_solve_kernel (csrc/clc_coop.hpp): the problem is read from HBM
    once, dealt to 256 co-resident workgroups and solved on chip; a pass is the serial chain [moments of the lane's points ->
    exchange of the 28 totals across the chip -> LM controller].  Kernel time: HIP event pair around the launch on the solver's
    stream (clc_options.profile_events = 2), best of 7."""
    built, ppl, _, aborts, off = solver.debug_coop()
    if not built or off:
        return None
    o = clc.default_options()
    o.profile_events = 2
    best, res = None, None
    for _ in range(7):
        r = solver.solve(x0, o, trace_cap=0)
        if r.summary.eval_kernel_launches != 1:
            return None  # the launch timed out and the step chain took over
        if best is None or r.summary.eval_kernel_ms < best:
            best, res = r.summary.eval_kernel_ms, r
    passes = int(res.summary.num_evaluations)
    sec = best * 1e-3
    evals = float(passes) * n_obs
    # lane layout (16 B per point slot, zero-padded to ppl per lane) + lane descriptors + plane table, read once; + the exchange words
    moved = COOP_WGS * ppl * COOP_LANES * 16 + COOP_WGS * COOP_LANES * 8 + n_scans * 48
    exchange = passes * (COOP_WGS * 448 * 2 + 8 * 448 * (1 + COOP_WGS))  # rows written + read by the 8 leaders, group rows written + read by all
    counts, vinfo = valu_counts()
    cc = ((counts or {}).get("coop") or {}).get(str(ppl)) or {}
    vp, vo = cc.get("valu_per_workgroup_pass"), cc.get("valu_per_workgroup_once", 0)
    if counts is not None and not cc:
        vinfo = dict(vinfo, current=False, error=f"no count for {ppl} points per lane in profiles/valu_counts.json")
    lane_instr = 64.0 * COOP_WGS * (passes * vp + vo) if vp else None
    rep = {"bound": "valu_f64", "peak": VALU_PEAK_LANE_INSTR / 1e12, "unit": "T VALU lane-instructions/s",
           "achieved": lane_instr / sec / 1e12 if vp else None, "frac": lane_instr / sec / VALU_PEAK_LANE_INSTR if vp else None,
           "kernel": f"clc::coop_solve_kernel<loss=1> (the whole LM solve in ONE launch: {COOP_WGS} co-resident workgroups x ({COOP_LANES} point lanes x {ppl} points "
                     "per lane in registers + LDS, + one controller wave); per pass: lane moments -> two-level exchange of the 28 totals through tagged words -> "
                     "LM controller (csrc/clc_lmuni.hpp) on the controller wave of every workgroup)",
           "launches_per_solve": 1, "avg_kernel_ms": best, "evaluation_passes": passes, "us_per_pass": 1e3 * best / passes,
           "valu_issue": {**vinfo, "per_pass": vp, "once": vo,
                          "lane_instructions_per_launch": lane_instr, "peak_lane_instructions_per_s": VALU_PEAK_LANE_INSTR,
                          "note": "VALU wave-instructions per WORKGROUP and pass / once per launch (difference of a 13-pass and a shorter launch under --pmc "
                                  "SQ_INSTS_VALU); x 256 workgroups x this run's passes x 64 lanes, over this run's kernel time, against 256 CUs x 4 SIMDs x "
                                  "16 lanes/clk x 2.4 GHz.  A fifth of the issue peak: the launch waits more than it computes, see `limited_by`"},
           "limited_by": "latency: a pass is the serial chain [pose + lane moments + expansion + wave reduction, FP64 issue on one wave per SIMD, ~2.3 us at 16 points "
                         "per lane] -> [exchange: 32 rows -> 8 group leaders -> everyone, two store-to-load hops through device memory of ~1.1 us each] -> "
                         "[LM controller's critical part on one wave, ~1.5 us]; stamps of a -DCLC_STAMPS build in profiles/r04_coop.md",
           "algorithmic_bytes_per_launch": BYTES_PER_EVAL * evals, "contract_64B_GBps": BYTES_PER_EVAL * evals / sec / 1e9,
           "contract_64B_frac": BYTES_PER_EVAL * evals / sec / 1e9 / HBM_PEAK_GBS,
           "moved_bytes_per_launch": int(moved + exchange), "achieved_moved_GBps": (moved + exchange) / sec / 1e9,
           "frac_moved": (moved + exchange) / sec / 1e9 / HBM_PEAK_GBS,
           "hbm_passes_over_the_data_per_solve": 1, "exchange_bytes_per_launch": int(exchange),
           "served_from": "infinity_cache" if moved <= INFINITY_CACHE_BYTES else "hbm", "traffic_measured_in_run": False,
           "traffic": cc.get("hbm_bytes_per_launch") if cc.get("passes") == passes else None,
           "traffic_source": "profiles/valu_counts.json (rocprofv3 --pmc FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 of the builder's run on the same sources)",
           "pricings": pricings(evals, n_scans * passes, moved + exchange, sec),
           "timing": "hipEvent pair around the launch on the solver's stream (profile_events = 2), best of 7",
           "cooperative_launch_timeouts": aborts,
           "note": "the observations cross HBM once per SOLVE (then live in registers + LDS): `frac_moved` = bytes actually moved / time / 8 TB/s is the HBM "
                   "fraction (bounded by 1, and small: the launch is not bandwidth-bound); `contract_64B_frac` prices the launch at the contract's 64 "
                   "algorithmic bytes per evaluation x all passes and is not a bandwidth.  `roofline_step_chain` / `roofline_large` are the streaming kernels"}
    return rep


def run_c4_shard(args, torch, dist, rank, world, local_rank, steps, warmup, use_rccl=True):
    """BASELINE.json configs[3] on this rank: generate problems [lo, hi) of the 8 192 x world batch, keep them resident,
    then time `steps` x (clc_solve_batched + RCCL all-gather of every rank's result records).
    Returns the dict rank 0 reports (None on other ranks)."""
    import camlasercalibratool_amd as clc
    from camlasercalibratool_amd import dist as cdist, simdata as sd

    n_total = args.problems_total if args.problems_total > 0 else args.problems_per_gpu * world
    t_gen0 = time.perf_counter()
    # the gathered records are delivered to rank 0 (clc_comm_set_root: ncclGather, one host copy at the root) unless
    # --gather-all-ranks asks for the all-gather + a host copy on every rank
    root = 0 if (use_rccl and not args.gather_all_ranks) else None
    ss = cdist.ShardSolver(n_total, device_index=local_rank, rank=rank, world=world, use_rccl=use_rccl, root=root)
    rccl_ranks = None
    if ss.comm is not None:  # RCCL itself must span the job: a future SCALE record shows that it saw N ranks
        rccl_ranks = ss.comm.rccl_ranks
        assert rccl_ranks == world, f"RCCL communicator spans {rccl_ranks} ranks, WORLD_SIZE is {world}"
    rec, off, x0, gt = sd.sim_shard_records(C4_SEED, ss.lo, ss.hi, args.shard_poses, args.shard_pts, 0.01)
    t_gen = time.perf_counter() - t_gen0
    t_up0 = time.perf_counter()
    ss.upload(rec, off)
    t_up = time.perf_counter() - t_up0
    n_obs_local = int(off[-1])
    keep = {k: rec[off[k]:off[k + 1]].copy() for k in (0, len(off) - 2)} if rank == 0 else {}
    hs = hooks_solver(clc, local_rank) if rank == 0 else None  # (rank 0, outside the timed region: the streaming kernel's own timing)
    if hs is not None:
        hs.upload_batched(rec, off)
    del rec
    dev_name, n_cus = ss.solver.device_info()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The step: with the RCCL communicator ONE call, clc_solve_batched_gather (the kernel's epilogue writes the result records into the
    # gather buffer; all-gather in place; one copy to the host); in the gloo dry runs the two calls + torch.distributed.
    fused = ss.comm is not None and not args.two_call_step
    # a stream of steps: clc_solve_batched_gather_pipelined — call k enqueues step k and hands back step k-1; the root's copy of the other
    # ranks' records to its host runs behind step k's kernel.  The flush that completes the last step is INSIDE the timed region.
    pipelined = fused and not args.no_pipeline

    def step():
        if pipelined:
            return ss.solve_gather_pipelined(x0, ordered=False)
        if fused:
            return ss.solve_gather(x0, ordered=False, copy=False)
        # start poses into the handle's pinned buffer, solve this shard in place, all-gather of all ranks' records
        return ss.solve(x0, ordered=False, copy=False, inplace=True)

    def finish(last):
        return ss.flush(ordered=False) if pipelined else last

    for _ in range(warmup):
        step()
    finish(None)
    # the timed region: `steps` steps bracketed by barrier + synchronize on both sides, `blocks` times over; every block's time is
    # the MAX over ranks, ms_per_step the median block
    block_s = []
    for _ in range(max(1, args.blocks)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        out = finish(out)
        barrier()
        block_s.append(time.perf_counter() - t0)
    per = args.shard_poses * args.shard_pts
    if fused:  # the shard's totals came back with the records (clc_batch_stats)
        evals_step, iters_step = float(ss.last_stats.evaluations) * per, float(ss.last_stats.iterations)
        assert ss.last_stats.problems == ss.hi - ss.lo, (ss.last_stats.problems, ss.hi - ss.lo)
    else:
        sms = ss.last_summaries
        evals_step = float(sum(sms[k].num_evaluations for k in range(len(sms)))) * per
        iters_step = float(sum(sms[k].num_iterations for k in range(len(sms))))
    if dist is not None:
        cdev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"
        tmax = torch.tensor(block_s, dtype=torch.float64, device=cdev)
        t = torch.tensor([evals_step, iters_step], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        block_s, evals_step, iters_step = [float(v) for v in tmax], float(t[0]), float(t[1])
    elapsed = float(np.median(block_s))
    res = None
    ci = ss.comm.info() if ss.comm is not None else None
    if rank == 0:
        raw = np.array(out).reshape(-1, 12)
        full = cdist.order_records(raw, n_total)
        dt = elapsed / steps
        shards = [list(cdist.shard_problems(n_total, r, world)) for r in range(world)]
        its = full[:, 9]
        res = {
            "workload": f"C4 shard: {args.problems_per_gpu} independent T_cl problems x {per} observations per GPU "
                        f"({args.shard_poses} poses x {args.shard_pts} pts, sigma=0.01 m, start 5 cm / 3 deg off the truth), "
                        f"x{world} GPUs = {n_total} problems; step = solve of the shard + RCCL all-gather of all {n_total} result records",
            "problems": n_total, "problems_per_gpu": args.problems_per_gpu, "observations_per_gpu": n_obs_local,
            "record_bytes_per_gpu": 64 * n_obs_local,
            "ms_per_step": 1e3 * dt, "ms_per_step_blocks": [1e3 * b / steps for b in block_s],
            "problems_per_s": n_total / dt, "evals_per_s": evals_step / dt,
            "lm_iters_per_s": iters_step / dt,
            "lm_iterations_min_max": [int(its.min()), int(its.max())],
            "terminations": {clc.TERMINATION.get(int(c), str(int(c))): int((full[:, 10] == c).sum()) for c in np.unique(full[:, 10])},
            "step_call": (("clc_solve_batched_gather_pipelined (step k's launch + collective enqueued, step k-1's records handed back; " if pipelined else "clc_solve_batched_gather (")
                          + "one launch whose epilogue writes the records into the gather buffer + in-place collective + the host copy at the root, "
                          f"fused={int(ss.last_stats.fused)})" if fused else "clc_solve_batched + clc_gather_results" if use_rccl else "clc_solve_batched + torch.distributed all_gather"),
            "gather": {"collective": (("ncclGather to rank 0 (in place)" if (ci is not None and ci.rooted_collectives > 0) else "ncclAllGather (in place)")
                                      + (" via clc_solve_batched_gather" + ("_pipelined" if pipelined else "") if fused else " via clc_gather_results")) if use_rccl
                                     else "torch.distributed all_gather (dry run)",
                       "root": root, "host_copy_ranks": (1 if root is not None else world) if use_rccl else None, "pipelined": bool(pipelined),
                       "host_copy_bytes_per_step_at_root": (96 * ss.cap * (world - 1)) if use_rccl else None,
                       "rooted_collectives": int(ci.rooted_collectives) if ci is not None else None, "collectives": int(ci.collectives) if ci is not None else None,
                       "library": ss.comm.library if ss.comm is not None else None, "rccl_ranks": rccl_ranks,
                       "bytes_per_rank": 96 * ss.cap, "bytes_total": 96 * ss.cap * world,
                       "records_per_rank": ss.cap, "padding_records": int((raw[:, 11] < 0).sum()),
                       "first_global_index_per_rank": [lo_ for lo_, _ in shards], "shard_sizes": [hi_ - lo_ for lo_, hi_ in shards],
                       "first_record_index_of_each_rank_block": [float(raw[r * ss.cap, 11]) if raw.shape[0] == world * ss.cap else None for r in range(world)]},
            "setup_s": {"generate_shard": t_gen, "upload_shard": t_up},
            "device": dev_name, "compute_units": n_cus,
            "first_and_last_record": [full[0].tolist(), full[-1].tolist()],
        }
        # ground truth recovery of this rank's problems (sanity, outside the timed region)
        mine = full[ss.lo:ss.hi]
        err = max(float(np.abs(sd.T_from_pose7(mine[k, :7]) - gt[k]).max()) for k in range(0, ss.hi - ss.lo, max(1, (ss.hi - ss.lo) // 256)))
        res["max_abs_T_err_vs_ground_truth_sampled"] = err
        # dominant kernel of the step: ONE launch of the resident kernel; its report leads, the streaming evaluation launch of the
        # lockstep path (still what runs when a problem does not fit a workgroup) follows as `streaming_eval_kernel`
        res["resident"] = dict(zip(("built", "lanes_per_problem", "max_points_per_lane", "rows"), ss.solver.debug_resident()))
        rk = resident_kernel_report(clc, ss.solver, x0, ss.hi - ss.lo, n_obs_local, (ss.hi - ss.lo) * args.shard_poses)
        # batched_eval_kernel over the whole shard (all problems active)
        _, _, brows_ok, bn_rows = ss.solver.debug_rows()
        streamed = bn_rows * ROW_BYTES if brows_ok else COMPACT_BYTES_PER_EVAL * n_obs_local
        kms = min(hs.time_batched_eval(x0, reps=10) for _ in range(3)) if hs is not None else float("nan")
        res["roofline"] = dict(hbm_figures(streamed, BYTES_PER_EVAL * n_obs_local, kms * 1e-3), traffic=None,
                               kernel=("clc::batched_rows_eval_kernel<loss=1,nt=1>" if brows_ok else "clc::batched_eval_kernel<compact,deep>")
                                      + " (all problems of the shard active)",
                               avg_kernel_ms=kms, served_from="infinity_cache" if streamed <= INFINITY_CACHE_BYTES else "hbm",
                               traffic_measured_in_run=False, timing="hipEvent pair around 10 back-to-back launches on the solver's stream (best of 3)")
        if hs is None:
            res["roofline"] = None
        if rk is not None:
            rk["streaming_eval_kernel"] = res["roofline"]
            res["roofline"] = rk
        if not args.no_cpu_baseline:
            import oracle as _o
            w = 0.0
            for k, r_k in keep.items():
                rk = _o.solve(r_k, x0[k], linear_solver="qr")
                w = max(w, float(np.abs(sd.T_from_pose7(mine[k, :7]) - sd.T_from_pose7(rk.pose)).max()))
            res["T_cl_max_abs_err_vs_oracle_sample"] = w
            res["oracle_sample"] = f"{len(keep)} of {ss.hi - ss.lo} problems of rank 0 (tests/test_gpu_parity.py::test_c4_full_size_shard checks 512 of 8192)"
    if hs is not None:
        hs.close()
    ss.close()
    return res


_RESULT_FD = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries loaded below write banners to the process's stdout from C
    (RCCL prints its version block when a communicator is created), so file descriptor 1 is pointed at stderr for the
    duration of the run and the result line is written to the original stdout at the end."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


LINE_LIMIT = 4096  # bytes: the driver keeps a bounded tail of stdout; the r04 line (21.6 KB) was not parsed


def _short(text, n=160):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def _num(v, digits=6):
    """Numbers of the line: finite floats rounded to `digits` significant digits, everything non-finite -> None (strict JSON)."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if not np.isfinite(v):
            return None
        return float(f"{v:.{digits}g}")
    return v


def _pick(d, keys, digits=6, keep_none=False):
    return {k: _num(d[k], digits) for k in keys if isinstance(d, dict) and k in d and (keep_none or d[k] is not None)}


def compact_line(out, detail_file=None):
    """The ONE stdout line of the contract, built from the full result dict: the headline fields, `config`, `roofline`, `cpu_baseline`,
    `parity`, `scale_base` and one-number summaries of the side measurements — at most LINE_LIMIT bytes, strict JSON (no NaN / Infinity).
    Everything else (sub-objects of the other kernels, pricings, notes, cold-start breakdown, gather bookkeeping) goes to the detail
    file named in `detail_file` and to stderr.  tests/test_bench_line.py holds this to the limit."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"), 9, keep_none=True)
    for k in ("higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        line[k] = out.get(k)
    cfg = out.get("config") or {}
    line["config"] = {k: (_short(v, 240) if isinstance(v, str) else v) for k, v in cfg.items()
                      if k in ("workload", "observations", "problems", "observations_per_gpu", "parallelism", "device", "compute_units")}
    line.update(_pick(out, ("timed_blocks", "ms_per_step_min", "ms_per_step_max", "ms_per_step_first_block", "lm_iters_per_s", "problems_per_s",
                            "per_gpu_value", "lm_iterations_per_solve", "evaluation_passes_per_solve", "final_cost"), 9))
    if out.get("termination") is not None:
        line["termination"] = out["termination"]
    rf = out.get("roofline")
    if isinstance(rf, dict):
        r = _pick(rf, ("bound", "peak", "unit", "achieved", "frac", "avg_kernel_ms", "frac_moved", "contract_64B_frac", "traffic",
                       "evaluation_passes", "us_per_pass", "moved_bytes_per_launch", "algorithmic_bytes_per_launch",
                       "traffic_measured_in_run", "served_from"))
        for k in ("achieved", "frac", "traffic"):  # null = not measured (e.g. VALU counts that belong to other sources), never absent
            r.setdefault(k, None)
        r["kernel"] = _short(rf.get("kernel", ""), 120)
        vi = rf.get("valu_issue")
        if isinstance(vi, dict):
            r["valu_issue"] = _pick(vi, ("source", "source_head", "csrc_sha16", "current", "per_pass", "once"))
        line["roofline"] = r
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "host_cpus", "ms_per_solve"))
        c["sample"] = _short(cb.get("sample", ""), 200)
        line["cpu_baseline"] = c
    pa = out.get("parity")
    if isinstance(pa, dict):
        line["parity"] = _pick(pa, ("T_cl_max_abs_err_vs_oracle", "final_cost_abs_err_vs_oracle", "iterations_gpu", "iterations_oracle"), 3)
        line["parity"]["gates"] = pa.get("gates")
    sb = out.get("scale_base")
    if isinstance(sb, dict):
        line["scale_base"] = _pick(sb, ("workload", "value", "unit", "ms_per_step"))
    summ = {}
    c3, c4, rl, cs = out.get("batched_c3"), out.get("batched_c4_shard"), out.get("roofline_large"), out.get("cold_start")
    if isinstance(c3, dict) and "ms_per_batch" in c3:
        summ["c3_ms_per_batch"] = _num(c3["ms_per_batch"], 5)
        summ["c3_T_cl_err_vs_oracle"] = _num(c3.get("T_cl_max_abs_err_vs_oracle_sample"), 3)
    if isinstance(c4, dict) and "ms_per_step" in c4:
        summ["c4_shard_ms_per_step"] = _num(c4["ms_per_step"], 5)
        summ["c4_T_cl_err_vs_oracle"] = _num(c4.get("T_cl_max_abs_err_vs_oracle_sample"), 3)
        k4 = c4.get("roofline") or {}
        summ["c4_kernel_ms"], summ["c4_kernel_frac"] = _num(k4.get("avg_kernel_ms"), 5), _num(k4.get("frac"), 4)
        g = c4.get("gather") or {}
        summ["c4_rccl_ranks"] = g.get("rccl_ranks")
        summ["c4_gather"] = {"root": g.get("root"), "host_copy_ranks": g.get("host_copy_ranks"), "pipelined": g.get("pipelined")}
    elif isinstance(c4, dict) and "error" in c4:
        summ["c4_error"] = _short(c4["error"], 120)
    if isinstance(rl, dict):
        summ["large"] = _pick(rl, ("frac", "frac_min", "frac_max", "after_idle_worst_frac", "avg_kernel_ms", "kernel_ms_min", "kernel_ms_max", "observations", "launches"), 5)
        summ["large"]["stat"] = rl.get("stat")
    msv = out.get("multistart")
    if isinstance(msv, dict) and "ms_per_call" in msv:
        summ["multistart_1024_ms"] = _num(msv["ms_per_call"], 5)
        summ["multistart_vs_c3"] = _num(msv.get("vs_c3_batch_of_uploaded_copies"), 4)
    lsv = out.get("roofline_large_solve")
    if isinstance(lsv, dict) and "frac" in lsv:
        summ["large_solve_frac"] = _num(lsv["frac"], 4)
        summ["large_solve"] = _pick(lsv, ("us_per_pass", "evaluation_passes", "ms_per_solve", "step_period_us", "observations"), 5)
    if isinstance(cs, dict):
        for shape in ("c1", "offline"):
            d = cs.get(shape) or {}
            if "first_call" in d:
                summ[f"cold_{shape}_ms"] = [_num(d.get("clc_create_ms"), 4), _num(d["first_call"].get("total_ms"), 4),
                                            _num((d.get("cpu_oracle") or {}).get("total_ms"), 4)]
        summ["cold_fields"] = "[clc_create, first call, same calls on the CPU port]"
    if summ:
        line["side"] = summ
    if detail_file:
        line["detail_file"] = detail_file
    # belt and braces: whatever a future field adds, the line never outgrows the limit
    for victim in ("side", "parity", "scale_base", "detail_file"):
        if len(json.dumps(line, allow_nan=False)) < LINE_LIMIT - 64:
            break
        line.pop(victim, None)
    return line


def _jsonable(o):
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, (np.floating, float)):
        return float(o) if np.isfinite(o) else None
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.bool_):
        return bool(o)
    if isinstance(o, np.ndarray):
        return _jsonable(o.tolist())
    return o


def _emit(out, detail_path=None):
    """stdout gets the compact line (compact_line); the full result goes to `detail_path` (default bench_detail.json beside this file,
    and a copy under gpurun_out/ when that directory exists) and to stderr."""
    sys.stdout.flush()
    full = _jsonable(out)
    paths = [detail_path] if detail_path else [os.path.join(ROOT, "bench_detail.json")]
    if not detail_path and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(full, f, indent=1, allow_nan=False)
            written = written or pth
        except OSError as e:
            print(f"bench.py: could not write {pth}: {e}", file=sys.stderr)
    print("bench.py detail: " + json.dumps(full, allow_nan=False), file=sys.stderr)
    line = json.dumps(compact_line(full, os.path.relpath(written, ROOT) if written else None), allow_nan=False)
    assert len(line) < LINE_LIMIT, len(line)
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (line + "\n").encode())


def _self_launch(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves — one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 — instead of quietly measuring one GPU.  Replaces this process."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args.gpus)  # does not return
    _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:  # never measure a different number of GPUs than the line will claim
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or run `python bench.py --gpus {args.gpus}` "
                         "without a launcher: it starts them itself)")

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    if torch.cuda.device_count() < world and not args.oversubscribe:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; "
                         "--oversubscribe with --backend gloo is for dry runs)")
    if args.oversubscribe:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    import camlasercalibratool_amd as clc
    from camlasercalibratool_amd import simdata as sd

    if world > 1:
        # ---- N>1: the C4 workload (sharded batch + RCCL gather); see run_c4_shard ----
        c4 = run_c4_shard(args, torch, dist, rank, world, local_rank, args.steps, args.warmup, use_rccl=(args.backend == "nccl"))
        if rank == 0:
            out = {
                "metric": "residual+Jacobian evals/s (batched LM solves, 8192 independent 1e4-obs point-to-plane problems per GPU, RCCL gather)",
                "value": c4["evals_per_s"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": c4["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": c4["workload"], "step": "clc_solve_batched of the rank's resident shard + one RCCL all-gather of all result records",
                           "problems": c4["problems"], "observations_per_gpu": c4["observations_per_gpu"],
                           "parallelism": f"problem-sharded x{world} (contiguous shards, no data-path collective), clc_gather_results",
                           "device": c4["device"], "compute_units": c4["compute_units"]},
                "timed_blocks": len(c4["ms_per_step_blocks"]), "ms_per_step_min": min(c4["ms_per_step_blocks"]),
                "ms_per_step_max": max(c4["ms_per_step_blocks"]), "ms_per_step_first_block": c4["ms_per_step_blocks"][0],
                "problems_per_s": c4["problems_per_s"], "lm_iters_per_s": c4["lm_iters_per_s"],
                "roofline": c4["roofline"],
                "batched_c4_shard": c4,
                # the N=1 line's `value` is C2 (the configuration the metric is quoted on, a single problem that does not shard); the base
                # of THIS curve is the same line's top-level `scale_base.value` (= its batched_c4_shard.evals_per_s)
                "per_gpu_value": c4["evals_per_s"] / world,
                "scale_base_field": "scale_base.value of the `bench.py --gpus 1` line (this workload, one GPU's shard)",
                "single_gpu_base": "run `bench.py --gpus 1`: its `batched_c4_shard` sub-object is this workload on one GPU",
            }
            _emit(out, args.detail_file)
        dist.destroy_process_group()
        return

    # ---- N=1: C2, the configuration the metric is quoted on ----
    S = sd.sim_fixed_count(1000, args.poses, args.pts, noise_sigma=args.noise)
    rec = clc.flatten_observations(S, use_linefitting_data=False)  # calibr_simulation.cpp:130 flags
    n_obs = rec.shape[0]
    x0 = sd.pose7_from_T(np.eye(4))
    solver = clc.Solver(local_rank)
    dev_name, n_cus = solver.device_info()
    solver.upload(rec)

    opt = clc.default_options()
    opt.profile_events = 1 if args.kernel_events else 0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = solver.solve(x0, opt, trace_cap=0)
    # the timed region: K solves bracketed by barrier + synchronize on both sides, `blocks` times over (at the driver's --steps 20
    # one block is under 2 ms of wall time: a single scheduler hiccup would move it by percent); ms_per_step = the median block
    block_ms, evals, iters = [], 0, 0
    for _ in range(max(1, args.blocks)):
        barrier()
        t0 = time.perf_counter()
        evals = iters = 0
        for _ in range(args.steps):
            res = solver.solve(x0, opt, trace_cap=0)  # no iteration trace: nothing forces a stream sync
            evals += res.summary.num_evaluations * n_obs
            iters += res.summary.num_iterations
        barrier()
        block_ms.append(1e3 * (time.perf_counter() - t0) / args.steps)
    elapsed_max, evals_total, iters_total = 1e-3 * float(np.median(block_ms)) * args.steps, float(evals), float(iters)  # one rank

    out = None
    hs = None
    if rank == 0:
        value = evals_total / elapsed_max
        out = {
            "metric": "residual+Jacobian evals/s (LM solve, 1e6-obs synthetic point-to-plane problem per GPU)",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "timed_blocks": len(block_ms), "ms_per_step_blocks": block_ms, "ms_per_step_min": min(block_ms), "ms_per_step_max": max(block_ms),
            "ms_per_step_first_block": block_ms[0],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2: single T_cl, {n_obs} point-to-plane observations ({args.poses} poses x {args.pts} pts, "
                            f"sigma={args.noise} m), init Tcl=I, Cauchy loss, Ceres-default LM; one independent problem per GPU",
                "step": "one complete clc_solve (observations resident in HBM)",
                "observations": n_obs,
                "parallelism": "single GPU (a single problem does not shard: replicas only; --gpus N>1 runs the C4 sharded batch)",
                "device": dev_name,
                "compute_units": n_cus,
            },
            "lm_iters_per_s": iters_total / elapsed_max,
            "lm_iterations_per_solve": res.summary.num_iterations,
            "evaluation_passes_per_solve": res.summary.num_evaluations,
            "final_cost": res.summary.final_cost,
            "termination": clc.TERMINATION.get(res.summary.termination),
        }

        # ---- roofline of the dominant kernel ----
        # The timed region consists of step_kernel launches only: one per evaluation pass (every workgroup first runs
        # the LM controller on the previous launch's partial rows, then streams its share of the observations) plus
        # one final controller-only launch per solve.  Its launch period is measured with HIP events on the solver's
        # stream (below); the rocprofv3 average of the streaming launches of step_kernel<..., 2, ...> in profiles/ is the
        # per-dispatch figure it must agree with.  The wall time of the timed region per pass is reported next to it.
        passes = res.summary.num_evaluations
        wall_ms_per_pass = (1e3 * elapsed_max / args.steps) / passes
        # kernel-only measurements: on a second handle of the hooks build (same code + clc_time_*), same resident array and pose
        hs = hooks_solver(clc, local_rank)
        roof = None
        if hs is not None:
            hs.upload(rec)
            # HIP events on the solver's stream right before launch 2 and right after launch passes-1 of a solve: the
            # steady-state launches, every one of which consumed a pass and streamed the array
            step_ms = min(hs.time_steps(x0, 2, passes - 1)[0] for _ in range(5)) if passes >= 4 else wall_ms_per_pass
            # the streaming part alone: HIP events on the solver's stream around 200 back-to-back launches of the
            # evaluation kernel (same loop, same layout, no controller prologue) on the same resident array and pose
            b2b = min(hs.time_eval(x0, reps=200) for _ in range(3))
            rows_ok, n_rows, _, _ = solver.debug_rows()
            # The default layout is a lossless re-encoding of the 64-byte records: per scan one plane, per point (x, y)
            # [z == 0] in rows of 64 points + a 64-byte descriptor per row.  The bytes actually streamed per launch:
            streamed = (n_rows * ROW_BYTES) if rows_ok else COMPACT_BYTES_PER_EVAL * n_obs
            layout = ("rows (16 B point + 64 B descriptor per row of 64 points, per-scan moments)" if rows_ok
                      else "compact (24 B point + 4 B group id per observation; group table per scan)")
            roof = dict(hbm_figures(streamed, BYTES_PER_EVAL * n_obs, step_ms * 1e-3), traffic=None,
                        kernel=("clc::step_kernel<loss=1,nt=0,mode=2,layout=rows,weighted=0 (equal shares at scan starts)>" if rows_ok else "clc::step_kernel<loss=1,deep=0,mode=2,layout=compact>")
                               + " (controller prologue + streaming loop)",
                        avg_kernel_ms=step_ms,
                        timing=f"hipEvent pair on the solver's stream around launches 2..{passes - 1} (steady state, back to back) of one "
                               "clc_solve, / number of launches (best of 5)",
                        timed_region_wall_ms_per_pass=wall_ms_per_pass, layout=layout, streamed_bytes_per_launch=int(streamed))
            roof["streaming_alone"] = dict(hbm_figures(streamed, BYTES_PER_EVAL * n_obs, b2b * 1e-3),
                                           kernel="clc::eval_rows_kernel<loss=1,nt=0,512,weighted=0 (equal shares at scan starts)>" if rows_ok else "clc::eval_kernel<loss=1,jac=1,deep=0,nt=0,compact=1,512>",
                                           avg_kernel_ms=b2b, timing="hipEvent pair around 200 back-to-back launches on the solver's stream (best of 3)")
            roof["pricings"] = pricings(n_obs, args.poses, streamed, step_ms * 1e-3)
            roof["frac_algorithmic_min"] = roof["pricings"]["reference_container_min"]["frac_of_hbm_peak"]
            roof["served_from"] = "infinity_cache" if streamed <= INFINITY_CACHE_BYTES else "hbm"
            roof["traffic_measured_in_run"] = False  # `traffic` below is the rocprofv3 PMC figure of profiles/pmc_traffic.json (builder's box)
            # the same evaluation on the other layouts (A/B through clc_set_launch): compact 28 B/obs, 64-byte tiles
            hs.set_launch(0, 2 | 16 | 32)
            b2bc = min(hs.time_eval(x0, reps=200) for _ in range(3))
            hs.set_launch(0, 6)
            b2b64 = min(hs.time_eval(x0, reps=200) for _ in range(3))
            hs.set_launch(0, -1)
            roof["compact28"] = dict(hbm_figures(COMPACT_BYTES_PER_EVAL * n_obs, BYTES_PER_EVAL * n_obs, b2bc * 1e-3), avg_kernel_ms=b2bc)
            roof["tiled64"] = dict(hbm_figures(BYTES_PER_EVAL * n_obs, BYTES_PER_EVAL * n_obs, b2b64 * 1e-3), avg_kernel_ms=b2b64,
                                   note="64-byte records streamed as stored: bytes moved == the contract's algorithmic bytes")
            roof["note"] = (f"{streamed / 2**20:.0f} MiB working set fits the 256 MiB Infinity Cache: steady-state passes are served on-die, and "
                            "~6.5 us of every launch is fixed cost (launch boundary 1.2, the previous launch's 57 KB of partial rows 1.7, the LM controller 2.2, reductions 1.4: scripts/stamps_step.py); "
                            "`frac` (= `frac_moved`) is bytes moved / time / 8 TB/s and is bounded by 1; `contract_64B_frac` prices the launch at the contract's 64 algorithmic bytes per "
                            "evaluation; see roofline_large for a working set beyond the cache")
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    roof["traffic"] = tj.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = tj.get("source")
                except Exception:
                    pass
        # The default clc_solve at this size is ONE launch of the cooperative kernel (csrc/clc_coop.hpp): that is the dominant kernel of
        # the timed region; the step chain (what runs beyond 2.6e6 observations, with p.z != 0, or with explicit launch flags) keeps
        # its own object.
        coop = coop_kernel_report(clc, solver, x0, n_obs, args.poses)
        if coop is not None:
            coop["timed_region_wall_ms_per_solve"] = 1e3 * elapsed_max / args.steps
            out["roofline"] = coop
            if roof is not None:
                roof["note"] = "clc_solve as the step chain (clc_set_launch with explicit flags; the default beyond what the chip holds). " + roof["note"]
                out["roofline_step_chain"] = roof
        else:
            out["roofline"] = roof

    # ---- working set beyond the 256 MiB Infinity Cache IN THE LAYOUT STREAMED: evaluation kernel only (rank 0, N=1) ----
    if rank == 0 and world == 1 and args.large_obs > 0 and hs is not None:
        reps = (args.large_obs + n_obs - 1) // n_obs
        big = np.ascontiguousarray(np.tile(rec, (reps, 1))[: args.large_obs])
        nb = int(big.shape[0])
        hs.upload(big)
        rows_ok_l, n_rows_l, _, _ = hs.debug_rows()
        streamed_l = (n_rows_l * ROW_BYTES) if rows_ok_l else COMPACT_BYTES_PER_EVAL * nb
        # The same launch takes 85 us in steady state and up to ~125 us for a few milliseconds after an idle -> load transition (a
        # power-management transient, profiles/r05_large.md: r04's best-of-3 right after the upload sat inside it).  So: 300 launches
        # (~26 ms) of sustained streaming first, then 15 groups of 10 back-to-back launches — the MEDIAN group is the figure (`frac`),
        # min / max ride along —, then the transient itself: 200 ms of idle and 8 more groups, the slowest of which is reported too.
        hs.time_eval(x0, reps=300)
        groups = sorted(hs.time_eval(x0, reps=10) for _ in range(15))
        ms, ms_lo, ms_hi = groups[len(groups) // 2], groups[0], groups[-1]
        time.sleep(0.2)
        after_idle = [hs.time_eval(x0, reps=10) for _ in range(8)]
        hs.set_launch(0, 2 | 16 | 32)
        msc = min(hs.time_eval(x0, reps=10) for _ in range(3))
        hs.set_launch(0, 6)
        ms64 = min(hs.time_eval(x0, reps=10) for _ in range(3))
        hs.set_launch(0, -1)
        out["roofline_large"] = dict(
            hbm_figures(streamed_l, BYTES_PER_EVAL * nb, ms * 1e-3),
            observations=nb, algorithmic_bytes=int(BYTES_PER_EVAL * nb), streamed_bytes=int(streamed_l),
            beyond_infinity_cache=bool(streamed_l > 256 * 2**20),
            kernel="clc::eval_rows_kernel<loss=1,nt=1,512,weighted=1>" if rows_ok_l else "clc::eval_kernel (compact, deep)",
            avg_kernel_ms=ms, kernel_ms_min=ms_lo, kernel_ms_max=ms_hi, launches=150,
            stat="median of 15 groups of 10 back-to-back launches after 300 warm ones (frac_min / frac_max: slowest / fastest group)",
            after_idle_groups_ms=after_idle, after_idle_worst_frac=streamed_l / (max(after_idle) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            after_idle_note="8 groups of 10 launches right after 200 ms of idle: the clocks' transient under the new load (profiles/r05_large.md)",
            frac_min=streamed_l / (ms_hi * 1e-3) / 1e9 / HBM_PEAK_GBS, frac_max=streamed_l / (ms_lo * 1e-3) / 1e9 / HBM_PEAK_GBS,
            kernel_ms_groups=groups, evals_per_s=nb / (ms * 1e-3),
            pricings=pricings(nb, (nb + args.pts - 1) // args.pts, streamed_l, ms * 1e-3),
            served_from="infinity_cache" if streamed_l <= INFINITY_CACHE_BYTES else "hbm", traffic_measured_in_run=False,
            note="`frac` (= `frac_moved`) = bytes the layout moves / time / 8 TB/s is the HBM fraction; `contract_64B_frac` prices the same launch at the "
                 "contract's 64 algorithmic bytes per evaluation and exceeds 1 because the layout is a 3.8x lossless compression",
            compact28=dict(hbm_figures(COMPACT_BYTES_PER_EVAL * nb, BYTES_PER_EVAL * nb, msc * 1e-3), avg_kernel_ms=msc),
            tiled64=dict(hbm_figures(BYTES_PER_EVAL * nb, BYTES_PER_EVAL * nb, ms64 * 1e-3), avg_kernel_ms=ms64))
        # ---- the WHOLE default-path clc_solve at this size (SURVEY.md §8 row g): no on-chip form holds 3.2e7 observations, so the default
        # path is the step chain — one step_kernel launch per LM iteration (controller on the previous launch's partial rows in front, then
        # the rows re-streamed from HBM).  Per pass = wall time of a solve / its evaluation passes, controller and launch boundary included.
        try:
            pi_l = hs.path_info()
            for _ in range(20):  # ~30 ms of sustained load first (the clock transient above)
                rl_ = hs.solve(x0, trace_cap=0)
            walls = []
            for _ in range(9):
                t1 = time.perf_counter()
                rl_ = hs.solve(x0, trace_cap=0)
                walls.append(time.perf_counter() - t1)
            passes_l = int(rl_.summary.num_evaluations)
            wall = float(np.median(walls))
            per_pass = wall / passes_l
            period_ms, _ = hs.time_steps(x0, 2, passes_l - 2)  # HIP events around launches 2 .. passes-2 of one more solve
            ls = dict(hbm_figures(streamed_l, BYTES_PER_EVAL * nb, per_pass),
                      observations=nb, streamed_bytes_per_pass=int(streamed_l), evaluation_passes=passes_l,
                      lm_iterations=int(rl_.summary.num_iterations), termination=rl_.termination,
                      ms_per_solve=1e3 * wall, ms_per_solve_min=1e3 * min(walls), ms_per_solve_max=1e3 * max(walls), us_per_pass=1e6 * per_pass,
                      step_period_us=1e3 * period_ms, frac_step_period=streamed_l / (period_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      eval_kernel_alone_us=1e3 * ms, frac_eval_kernel_alone=streamed_l / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      default_path="step chain" if not (pi_l.coop_resident or pi_l.single_resident) else "on chip",
                      kernel="clc::step_kernel<loss=1,nt=1,mode 0|1|2,rows,equal shares> x passes (one launch per LM iteration)",
                      served_from="infinity_cache" if streamed_l <= INFINITY_CACHE_BYTES else "hbm", traffic_measured_in_run=False,
                      stat="median of 9 solves after 20 warm ones; per pass = wall / evaluation passes (launch boundary + partial rows + LM controller included)",
                      where_the_difference_goes="profiles/r06_large_solve.md: launch boundary 3.9 us + partial rows 2.3 + barrier 0.6 + controller 3.1 in "
                                                "front of ~88 us of streaming (stamped build); overlapping them (deeper launch queue, LDS-DMA prefetch) "
                                                "was measured slower: the part runs power-managed in this regime")
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic_large_solve.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    ls["traffic"] = tj.get("hbm_bytes_per_launch")
                    ls["traffic_source"] = tj.get("source")
                except Exception:
                    pass
            if not args.no_cpu_baseline:  # parity of this very solve (normal-equation oracle with threads: the dense-Jacobian QR is 1.5 GB here)
                import oracle
                ref_l = oracle.solve(big, x0, linear_solver="ne", threads=min(64, oracle.max_threads()))
                ls["parity"] = {"oracle": "oracle/clc_oracle.cpp, normal-equation form, threads (pinned against DENSE_QR on the CPU: tests/test_oracle_solver.py)",
                                "T_cl_max_abs_err": float(np.abs(sd.T_from_pose7(rl_.pose) - sd.T_from_pose7(ref_l.pose)).max()),
                                "final_cost_abs_err": float(abs(rl_.summary.final_cost - ref_l.summary.final_cost)),
                                "iterations_gpu": int(rl_.summary.num_iterations), "iterations_oracle": int(ref_l.summary.num_iterations)}
            out["roofline_large_solve"] = ls
        except Exception as e:  # (never fatal for the line)
            out["roofline_large_solve"] = {"error": repr(e)}
        del big
        hs.upload(rec[:64])  # (frees nothing: the handle keeps its capacity; the large array is simply no longer referenced)

    # ---- BASELINE.json configs[2] (C3): 1 024 independent T_cl problems x 10^4 observations ----
    if rank == 0 and world == 1 and not args.no_batched:
        Pb = 1024
        probs, gts = sd.sim_batch(4242, Pb, 20, 500, noise_sigma=0.01)
        recs = [clc.flatten_observations(p, False) for p in probs]
        offb = np.zeros(Pb + 1, dtype=np.int64)
        offb[1:] = np.cumsum([r.shape[0] for r in recs])
        xb = solver.pose_plus(np.stack([sd.pose7_from_T(g) for g in gts]), np.random.default_rng(1).normal(size=(Pb, 6)) * 0.05)
        solver.upload_batched(np.concatenate(recs), offb)
        tb = []
        pin_poses, _ = solver.batched_buffers()  # the handle's pinned arrays: start poses in, results out, no staging copies
        for _ in range(25):
            t1 = time.perf_counter()
            pin_poses[:] = xb
            pb, smb = solver.solve_batched_inplace()
            tb.append(time.perf_counter() - t1)
        dtb = float(np.median(tb[5:]))
        pb = pb.copy()
        evb = sum(smb[k].num_evaluations * int(offb[k + 1] - offb[k]) for k in range(Pb))
        out["batched_c3"] = {
            "workload": "C3: 1024 independent T_cl problems x 10000 observations (655 MB), start 5 cm / 3 deg off the truth",
            "ms_per_batch": 1e3 * dtb, "problems_per_s": Pb / dtb, "evals_per_s": evb / dtb,
            "lm_iterations_min_max": [min(s_.num_iterations for s_ in smb), max(s_.num_iterations for s_ in smb)],
            "max_abs_T_err_vs_ground_truth": float(max(np.abs(sd.T_from_pose7(pb[k]) - gts[k]).max() for k in range(Pb))),
        }
        out["batched_c3"]["resident"] = dict(zip(("built", "lanes_per_problem", "max_points_per_lane", "rows"), solver.debug_resident()))
        rk3 = resident_kernel_report(clc, solver, xb, Pb, int(offb[-1]), Pb * 20)
        if rk3 is not None:
            out["batched_c3"]["solve_kernel"] = rk3
        _, _, brows_ok, bn_rows = solver.debug_rows()
        sb = bn_rows * ROW_BYTES if brows_ok else COMPACT_BYTES_PER_EVAL * int(offb[-1])
        if hs is not None:
            hs.upload_batched(np.concatenate(recs), offb)
            kb = min(hs.time_batched_eval(xb, reps=20) for _ in range(3))
            out["batched_c3"]["eval_kernel"] = dict(hbm_figures(sb, BYTES_PER_EVAL * int(offb[-1]), kb * 1e-3), avg_kernel_ms=kb,
                                                served_from="infinity_cache" if sb <= INFINITY_CACHE_BYTES else "hbm",
                                                note="the streaming evaluation launch of the lockstep path (one pass over all problems), not what the default solve runs")
        if not args.no_cpu_baseline:
            import oracle as _o
            w = 0.0
            sample = (0, 511, 1023)
            for k in sample:
                rk = _o.solve(recs[k], xb[k], linear_solver="qr")
                w = max(w, float(np.abs(sd.T_from_pose7(pb[k]) - sd.T_from_pose7(rk.pose)).max()))
            out["batched_c3"]["T_cl_max_abs_err_vs_oracle_sample"] = w
            out["batched_c3"]["oracle_sample"] = f"{len(sample)} of {Pb} problems (tests/test_gpu_parity.py::test_c3_full_size_batch checks all 1024)"

        # ---- multi-hypothesis calibration on SHARED observations (north_star): 1 024 start poses on ONE problem of the C3 shape — one copy
        # of its observations on the device (clc_solve_multistart) against the C3 batch of 1 024 uploaded problems above ----
        try:
            x_true0 = sd.pose7_from_T(gts[0])
            starts = solver.pose_plus(np.tile(x_true0, (Pb, 1)), np.random.default_rng(3).normal(size=(Pb, 6)) * 0.05)
            solver.upload_batched(recs[0], np.array([0, recs[0].shape[0]], dtype=np.int64))
            tm = []
            for _ in range(15):
                t1 = time.perf_counter()
                pm, smm = solver.solve_multistart(starts)
                tm.append(time.perf_counter() - t1)
            dtm = float(np.median(tm[5:]))
            pi_m = solver.path_info()
            out["multistart"] = {
                "workload": f"{Pb} start poses (5 cm / 3 deg around the truth) on ONE problem of {recs[0].shape[0]} observations: clc_solve_multistart, one launch, a workgroup per start",
                "ms_per_call": 1e3 * dtm, "starts_per_s": Pb / dtm, "vs_c3_batch_of_uploaded_copies": dtm / dtb,
                "observation_bytes_on_device_lane_layout": int(pi_m.batched_lane_rows) * int(pi_m.batched_lanes) * 16,
                "observation_bytes_of_1024_uploaded_copies_lane_layout": int(out["batched_c3"]["resident"]["rows"]) * int(out["batched_c3"]["resident"]["lanes_per_problem"]) * 16,
                "lm_iterations_min_max": [min(s_.num_iterations for s_ in smm), max(s_.num_iterations for s_ in smm)],
                "max_abs_T_spread_between_starts": float(max(np.abs(sd.T_from_pose7(pm[k]) - sd.T_from_pose7(pm[0])).max() for k in range(Pb))),
                "resident": bool(pi_m.batched_resident)}
        except Exception as e:
            out["multistart"] = {"error": repr(e)}

    # ---- CPU baseline + parity (rank 0, N=1): the oracle's DENSE_QR Ceres restatement, 1 thread ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle

        ref = None
        cpu_t = 0.0
        cpu_evals = 0
        cpu_iters = 0
        n_rep = 0
        while cpu_t < args.cpu_seconds and n_rep < 64:
            t1 = time.perf_counter()
            ref = oracle.solve(rec, x0, linear_solver="qr", threads=1)
            cpu_t += time.perf_counter() - t1
            cpu_evals += ref.summary.num_residual_evaluations * n_obs
            cpu_iters += ref.summary.num_iterations
            n_rep += 1
        out["cpu_baseline"] = {
            "value": cpu_evals / cpu_t, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"{n_rep} full solves of the same 1e6-obs problem by oracle/clc_oracle.cpp (Ceres-LM restatement, dense "
                      f"N x 6 Jacobian + Householder QR, -O3, 1 thread; the reference runs Ceres with num_threads=1), "
                      f"{cpu_t:.1f} s of CPU time; counts cost-only and Jacobian passes alike",
            "lm_iters_per_s": cpu_iters / cpu_t,
            "ms_per_solve": 1e3 * cpu_t / n_rep,
            "host_cpus": os.cpu_count(),
        }
        # other CPU variants of BASELINE.md §3 (evaluation passes only, bounded): normal-equation
        # accumulation (the GPU's formulation) on 1 thread and on the host cores OpenMP gives us
        def _time_ne(threads, reps):
            t1 = time.perf_counter()
            for _ in range(reps):
                oracle.evaluate_ne(rec, x0, threads=threads)
            return n_obs * reps / (time.perf_counter() - t1)
        nthr = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0)), 64))
        out["cpu_baseline"]["variants"] = {
            "ne_1_thread_evals_per_s": _time_ne(1, 10),
            "ne_openmp_evals_per_s": _time_ne(nthr, 20), "ne_openmp_threads": nthr,
            "note": "residual+Jacobian+6x6 accumulation passes only (no QR, no LM control)",
        }
        # the reference's OWN sources (oracle/_ref/libref.so, built against stand-in Eigen/Ceres headers) on the
        # reference's own problem size (C1): reported for transparency, not used as the baseline — the stand-in
        # matrices live on the heap, which makes this several times slower than real Eigen would be
        try:
            import oracle.ref as oref
            if os.path.exists(oref.LIB_PATH):
                S1 = sd.GenerateSimData(1, noise_sigma=0.01)
                rec1 = clc.flatten_observations(S1, False)
                passes1 = oracle.solve(rec1, x0, linear_solver="qr").summary.num_residual_evaluations
                t1 = time.perf_counter()
                n1 = 0
                while time.perf_counter() - t1 < 1.0:
                    oref.calibration(S1, np.eye(4), False, False)
                    n1 += 1
                out["cpu_baseline"]["variants"]["reference_sources_standin_eigen_c1_evals_per_s"] = \
                    passes1 * rec1.shape[0] * n1 / (time.perf_counter() - t1)
                t1 = time.perf_counter()
                for _ in range(20):
                    oracle.solve(rec1, x0, linear_solver="qr")
                out["cpu_baseline"]["variants"]["port_c1_evals_per_s"] = passes1 * rec1.shape[0] * 20 / (time.perf_counter() - t1)
        except Exception as e:  # the library is test infrastructure and may be absent
            out["cpu_baseline"]["variants"]["reference_sources_note"] = f"oracle/_ref not available: {e}"
        dT = float(np.abs(sd.T_from_pose7(res.pose) - sd.T_from_pose7(ref.pose)).max())
        # what the unpinned part of the oracle (Ceres' minimiser, absent here) can cost: how far the result moves if the
        # solve ran on to the minimum instead of stopping at function_tolerance — the size of a +-1-iteration
        # disagreement with a real Ceres (tests/test_ceres_handoff.py, tests/golden/dump_ceres_trace.cpp)
        tight = clc.default_options()
        tight.function_tolerance = 1e-12
        rt = solver.solve(x0, tight, trace_cap=0)
        sens = {"final_cost_minus_converged_minimum": res.summary.final_cost - rt.summary.final_cost,
                "bound_function_tolerance_x_cost": 1e-6 * res.summary.final_cost,
                "T_cl_max_abs_shift_to_converged_minimum": float(np.abs(sd.T_from_pose7(res.pose) - sd.T_from_pose7(rt.pose)).max()),
                "iterations_default_vs_converged": [res.summary.num_iterations, rt.summary.num_iterations]}
        out["parity"] = {
            "stopping_rule_sensitivity": sens,
            "T_cl_max_abs_err_vs_oracle": dT,
            "final_cost_abs_err_vs_oracle": abs(res.summary.final_cost - ref.summary.final_cost),
            "iterations_gpu": res.summary.num_iterations, "iterations_oracle": ref.summary.num_iterations,
            "gates": {"T_cl": 1e-6, "final_cost": 1e-8},
            "oracle": "CPU restatement of the reference + Ceres LM (reference-owned arithmetic pinned against the reference's own sources, tests/test_ref_pin.py; Ceres minimiser restated, unpinned: no Ceres available)",
        }

    # ---- cold path (rank 0, N=1): what a user of the drop-in waits for — main/calibr_offline.cpp:166-170 calls the path once per
    # process — measured in FRESH processes (scripts/cold_start.py); never part of `value`
    if rank == 0 and world == 1 and not args.no_cold_start:
        import subprocess
        cs = {}
        for shape in ("c1", "offline"):
            try:
                cmd = [sys.executable, os.path.join(ROOT, "scripts", "cold_start.py"), shape] + (["--no-cpu"] if args.no_cpu_baseline else [])
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
                line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                cs[shape] = json.loads(line[-1]) if pr.returncode == 0 and line else {"error": (pr.stderr or pr.stdout)[-400:]}
            except Exception as e:  # never let the side measurement take the line down
                cs[shape] = {"error": repr(e)}
        cs["note"] = ("fresh process per shape: clc_create (HIP runtime + context + stream + buffers) + first closed form + first refinement and "
                      "analysis pass; `second_call_same_process` is the warm figure; cpu_oracle = the CPU port's time for the same calls")
        out["cold_start"] = cs

    # ---- BASELINE.json configs[3] (C4), one GPU's share: the N=1 base of the multi-GPU curve ----
    # Last, and under a watchdog: it is the one section of the N=1 run that creates an RCCL communicator (world size 1).  Should
    # that ever not return, the line — complete but for this sub-object — is still written.
    if rank == 0 and world == 1 and not args.no_c4_shard:
        import threading

        def _give_up():
            out["batched_c4_shard"] = {"error": "the C4-shard section did not finish within 300 s; everything else in this line was measured before it"}
            _emit(out, args.detail_file)
            os._exit(0)

        dog = threading.Timer(300.0, _give_up)
        dog.daemon = True
        dog.start()
        out["batched_c4_shard"] = run_c4_shard(args, torch, None, 0, 1, local_rank, args.shard_steps, 2)
        dog.cancel()
        # what the --gpus N>1 lines scale from: their `value` / N (`per_gpu_value`) against this figure, not against `value` above
        out["scale_base"] = {"workload": "C4 shard", "value": out["batched_c4_shard"]["evals_per_s"], "unit": "evals/s",
                             "ms_per_step": out["batched_c4_shard"]["ms_per_step"],
                             "note": "one GPU's share of the --gpus N>1 workload (8 192 problems x 1e4 observations, clc_solve_batched + result gather); "
                                     "the N>1 lines report `per_gpu_value` = value / N to set against it"}

    if rank == 0:
        _emit(out, args.detail_file)
    if hs is not None:
        hs.close()
    solver.close()


if __name__ == "__main__":
    main()
