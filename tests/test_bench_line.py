"""bench.py's ONE stdout line stays something the driver can parse: under 4 KB, strict JSON, the contract's fields present — whatever
the full result dict holds (the r04 line had grown to 21.6 KB and the driver recorded `parsed: null`).  Also: `--gpus N` without a
launcher starts N ranks itself or refuses, and the VALU instruction counts are refused when they were measured on other sources.
No GPU needed."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _fat(depth=0):
    """A sub-object of the kind the r04 line carried: long notes, nested pricings, lists."""
    d = {"note": "x" * 700, "frac": 0.5, "avg_kernel_ms": 0.123456789012, "kernel": "clc::kernel<" + "a, " * 80 + ">",
         "pricings": {k: {"bytes": 10 ** 9, "GBps": 1234.5678901, "frac_of_hbm_peak": 0.15} for k in ("a", "b", "c")},
         "list": list(range(200))}
    if depth < 2:
        d["child"] = _fat(depth + 1)
    return d


def canned(nan=False):
    bad = float("nan") if nan else 1.0
    roof = dict(_fat(), bound="valu_f64", peak=39.3216, unit="T VALU lane-instructions/s", achieved=8.6, frac=0.22, frac_moved=0.049,
                contract_64B_frac=1.25, traffic=23100160.0, evaluation_passes=13, us_per_pass=6.39,
                valu_issue={"source": "profiles/valu_counts.json", "source_head": "0123456789ab", "csrc_sha16": "f" * 16, "current": True,
                            "per_pass": 3298, "once": 1041, "note": "y" * 900})
    return {
        "metric": "residual+Jacobian evals/s (LM solve, 1e6-obs synthetic point-to-plane problem per GPU)", "value": 1.45e11 * bad, "unit": "evals/s",
        "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.0897, "timed_blocks": 5, "ms_per_step_min": 0.0891, "ms_per_step_max": 0.0931,
        "ms_per_step_blocks": [0.09] * 5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2: " + "w" * 400, "step": "s" * 300, "observations": 1000000, "parallelism": "p" * 100, "device": "AMD Instinct MI355X",
                   "compute_units": 256},
        "lm_iters_per_s": 1.2e5, "final_cost": 49.123456789, "termination": "CONVERGENCE (function tolerance)",
        "roofline": roof, "roofline_step_chain": _fat(), "roofline_large": dict(_fat(), frac=0.7, frac_min=0.5, frac_max=0.8, stat="median", observations=32000000),
        "batched_c3": dict(_fat(), ms_per_batch=0.15, T_cl_max_abs_err_vs_oracle_sample=float("inf") if nan else 1e-15),
        "cpu_baseline": {"value": 3.0e7, "unit": "evals/s", "cores": 1, "kind": "port", "sample": "z" * 500, "host_cpus": 256, "variants": _fat()},
        "parity": {"stopping_rule_sensitivity": _fat(), "T_cl_max_abs_err_vs_oracle": 4.4e-16, "final_cost_abs_err_vs_oracle": 9.3e-16,
                   "iterations_gpu": 11, "iterations_oracle": 11, "gates": {"T_cl": 1e-6, "final_cost": 1e-8}, "oracle": "o" * 300},
        "cold_start": {"c1": {"clc_create_ms": 160.0, "first_call": {"total_ms": 1.43}, "cpu_oracle": {"total_ms": 1.5}, "more": _fat()},
                       "offline": {"clc_create_ms": 235.0, "first_call": {"total_ms": 1.78}, "cpu_oracle": {"total_ms": 0.3}}, "note": "n" * 500},
        "batched_c4_shard": dict(_fat(), ms_per_step=0.9366, evals_per_s=4.9e11, T_cl_max_abs_err_vs_oracle_sample=1e-15,
                                 roofline={"avg_kernel_ms": 0.8314, "frac": 0.535}, gather={"rccl_ranks": 1, "shard_sizes": [8192]},
                                 first_and_last_record=[[0.0] * 12, [1.0] * 12]),
        "scale_base": {"workload": "C4 shard", "value": 4.9e11, "unit": "evals/s", "ms_per_step": 0.9366, "note": "q" * 300},
    }


def _no_nan(o):
    if isinstance(o, dict):
        return all(_no_nan(v) for v in o.values())
    if isinstance(o, list):
        return all(_no_nan(v) for v in o)
    return not (isinstance(o, float) and not math.isfinite(o))


@pytest.mark.parametrize("nan", [False, True])
def test_line_is_small_strict_json_with_the_contract_fields(nan):
    full = bench._jsonable(canned(nan))
    line = json.dumps(bench.compact_line(full, "bench_detail.json"), allow_nan=False)  # raises on NaN / Infinity
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    assert _no_nan(d)
    for k in CONTRACT:
        assert k in d, k
    assert d["config"]["workload"].startswith("C2") and len(d["config"]["workload"]) <= 240
    rf = d["roofline"]
    for k in ("bound", "peak", "unit", "achieved", "frac", "kernel", "avg_kernel_ms", "frac_moved", "contract_64B_frac", "traffic"):
        assert k in rf, k
    assert rf["valu_issue"]["source_head"] == "0123456789ab" and rf["valu_issue"]["current"] is True
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and len(cb["sample"]) <= 200
    assert d["parity"]["iterations_gpu"] == d["parity"]["iterations_oracle"] == 11
    assert d["scale_base"]["workload"] == "C4 shard" and d["scale_base"]["ms_per_step"] == pytest.approx(0.9366)
    side = d["side"]
    assert side["c4_shard_ms_per_step"] == pytest.approx(0.9366) and side["c4_rccl_ranks"] == 1 and side["c3_ms_per_batch"] == pytest.approx(0.15)
    assert side["large"]["frac"] == pytest.approx(0.7) and side["cold_c1_ms"] == [160.0, 1.43, 1.5]
    if nan:
        assert d["value"] is None and side["c3_T_cl_err_vs_oracle"] is None  # non-finite numbers become null, never NaN
    assert d["detail_file"] == "bench_detail.json"


def test_line_of_the_multi_gpu_run_and_of_a_bare_result():
    full = canned()
    full.update(n_gpus=8, per_gpu_value=full["value"] / 8, problems_per_s=8.7e6)
    full["config"] = {"workload": "C4 shard x8", "problems": 65536, "observations_per_gpu": 81920000, "parallelism": "problem-sharded x8", "device": "MI355X",
                      "compute_units": 256, "step": "t" * 200}
    d = json.loads(json.dumps(bench.compact_line(bench._jsonable(full)), allow_nan=False))
    assert d["n_gpus"] == 8 and d["config"]["problems"] == 65536 and d["per_gpu_value"] == pytest.approx(full["value"] / 8, rel=1e-6)
    # a result with nothing but the headline (every side measurement switched off) still makes a line
    bare = {k: canned()[k] for k in CONTRACT}
    d = json.loads(json.dumps(bench.compact_line(bare), allow_nan=False))
    assert set(CONTRACT) <= set(d) and "side" not in d and "roofline" not in d


def test_emit_writes_the_detail_file_and_one_short_line(tmp_path, capfd):
    detail = tmp_path / "detail.json"
    bench._RESULT_FD = None
    full = canned()
    full["array"] = np.arange(3.0)
    bench._emit(full, str(detail))
    out, err = capfd.readouterr()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    kept = json.load(open(detail))
    assert kept["roofline_step_chain"]["child"]["child"]["note"] == "x" * 700 and kept["array"] == [0.0, 1.0, 2.0]  # nothing is lost
    assert "bench.py detail: {" in err


def test_valu_counts_are_refused_when_measured_on_other_sources(tmp_path):
    from camlasercalibratool_amd import _build
    sha = _build.csrc_sha16()
    assert len(sha) == 16 and sha == _build.csrc_sha16()
    f = tmp_path / "valu_counts.json"
    f.write_text(json.dumps({"csrc_sha16": sha, "head": "abc", "coop": {"16": {"valu_per_workgroup_pass": 1}}}))
    counts, info = bench.valu_counts(str(f))
    assert counts is not None and info["current"] is True and info["source_head"] == "abc"
    f.write_text(json.dumps({"csrc_sha16": "0" * 16, "head": "abc", "coop": {}}))
    counts, info = bench.valu_counts(str(f))
    assert counts is None and info["current"] is False and info["tree_csrc_sha16"] == sha
    counts, info = bench.valu_counts(str(tmp_path / "missing.json"))
    assert counts is None and info["current"] is False


def test_gpus_n_without_a_launcher_starts_n_ranks_or_refuses():
    """`python bench.py --gpus 2` with no WORLD_SIZE must not quietly measure one GPU: it re-executes itself under
    torch.distributed.run with 2 ranks (seen here through the launcher's ranks failing for want of a GPU, each naming its rank), and a
    launcher whose WORLD_SIZE disagrees with --gpus is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--oversubscribe"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert "--gpus 2 without WORLD_SIZE: launching" in p.stderr and "--nproc-per-node 2" in p.stderr
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "no GPU visible" in p.stderr  # the ranks were started and refused to run without a device
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env2)
    assert p.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


def test_committed_valu_counts_belong_to_the_kernel_sources_in_the_tree():
    """profiles/valu_counts.json (the PMC instruction counts behind roofline.frac) was measured on exactly the kernel sources that are in
    the tree: a change to any csrc header or to the compiler flags without re-running scripts/profile_kernels.sh + scripts/summarize_kernels.py
    turns this red (and bench.py would report frac = null rather than a stale figure)."""
    counts, info = bench.valu_counts()
    assert counts is not None and info["current"] is True, info
    assert "16" in counts["coop"] and 3000 < counts["coop"]["16"]["valu_per_workgroup_pass"] < 3600   # C2: 16 points per lane
    assert "42" in counts["resident"] and 1300 < counts["resident"]["42"]["valu_per_wave_pass"] < 1700  # C3 / C4: 42 points per lane
    assert len(info["source_head"]) >= 7
