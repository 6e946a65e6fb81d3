"""The C-ABI library loads without a GPU, exports every symbol include/clc.h declares, and
refuses to compute without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import _build, _capi, simdata as sd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "clc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(clc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_capi.EXPORTED)
    L = _capi.load(_build.PRODUCT_LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/clc.h but not exported"
    assert L.clc_version() == 210


def _dynamic_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] == "T"}  # functions (kernel handles are objects)


def test_product_library_exports_the_header_and_nothing_else():
    """nm -D libclc_hip.so: exactly the functions include/clc.h declares — no clc_debug_* / clc_time_* hook, no helper shared
    between the translation units (-fvisibility=hidden); the hooks exist only in the -DCLC_TEST_HOOKS builds."""
    prod = _dynamic_symbols(_build.PRODUCT_LIB_PATH)
    ours = {s for s in prod if not s.startswith(("_init", "_fini", "__hip", "__bss", "_edata", "_end", "__cudaRegister"))}
    assert sum(1 for s in prod if "clc_debug" in s or "clc_time" in s) == 0  # nm -D libclc_hip.so | grep -c clc_debug == 0
    assert {s for s in ours if s.startswith("clc_")} == set(_capi.EXPORTED)
    stray = {s for s in ours if not s.startswith("clc_") and ("clc" in s or s.startswith("_Z"))}
    assert not stray, sorted(stray)[:10]
    hooks = _dynamic_symbols(_build.HOOKS_LIB_PATH)
    assert {s for s in hooks if s.startswith("clc_")} == set(_capi.EXPORTED) | set(_capi.HOOKS)


def hdr_text():
    return open(os.path.join(ROOT, "include", "clc.h")).read()


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_capi.Options) == 4 * 4 + 10 * 8 + 2 * 4
    assert C.sizeof(_capi.Iteration) == 16 + 6 * 8
    assert C.sizeof(_capi.Summary) == 16 + 8 + 4 * 8 + 8
    assert C.sizeof(_capi.PathInfo) == 16 * 4 + 5 * 8  # clc_path_info: 15 int32 + 1 reserved, 5 int64 (fields are only ever appended)
    assert C.sizeof(_capi.CommInfo) == 8 * 4 + 5 * 8  # clc_comm_info
    body = re.sub(r"/\*.*?\*/", "", re.search(r"typedef struct clc_comm_info \{(.*?)\} clc_comm_info;", hdr_text(), flags=re.S).group(1), flags=re.S)
    names = [n.strip() for m in re.finditer(r"(int32_t|int64_t)\s+([\w, ]+);", body) for n in m.group(2).split(",")]
    assert names == [f[0] for f in _capi.CommInfo._fields_]
    assert C.sizeof(_capi.BatchStats) == 4 * 8 + 2 * 4 + 2 * 8  # clc_batch_stats
    hdr = open(os.path.join(ROOT, "include", "clc.h")).read()
    body = re.sub(r"/\*.*?\*/", "", re.search(r"typedef struct clc_path_info \{(.*?)\} clc_path_info;", hdr, flags=re.S).group(1), flags=re.S)
    assert [m.group(2) for m in re.finditer(r"(int32_t|int64_t)\s+(\w+);", body)] == [f[0] for f in _capi.PathInfo._fields_]


def test_result_record_is_twelve_doubles():
    hdr = open(os.path.join(ROOT, "include", "clc.h")).read()
    body = re.search(r"typedef struct clc_result_record \{(.*?)\} clc_result_record;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    n = sum(int(m.group(2) or 1) * len(m.group(1).split(",")) if not m.group(2) else int(m.group(2))
            for m in re.finditer(r"double\s+([a-z_, ]+?)(?:\[(\d+)\])?;", body))
    from camlasercalibratool_amd import dist, solver
    assert n == 12 == dist.RECORD == solver.RESULT_RECORD


def test_host_side_flatten_needs_no_gpu_and_matches_oracle(oracle_mod):
    S = sd.sim_board_edges(3, 5, 12)
    for lf, bd in [(False, False), (True, False), (True, True)]:
        assert np.array_equal(clc.flatten_observations(S, lf, bd), oracle_mod.flatten(S, lf, bd))
    S.pts_off[:] = 0
    with pytest.raises(clc.ClcError) as e:
        clc.flatten_observations(S, True, True)
    assert e.value.code == -4  # reference: std::out_of_range at LaseCamCalCeres.cpp:278


def test_pose_plus_jacobian_constant():
    import ctypes as C
    J = np.empty(42)
    assert _capi.lib().clc_pose_plus_jacobian(None, _capi.dptr(J)) == 0
    assert np.array_equal(J.reshape(7, 6), np.vstack([np.eye(6), np.zeros((1, 6))]))


def test_no_cpu_fallback():
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(clc.ClcError) as e:
        clc.Solver()
    assert e.value.code == -7


def test_rccl_binding_without_a_gpu():
    """The multi-GPU entry points bind RCCL at run time (dlopen of the copy already mapped, else the system one).  Without
    a GPU, ncclGetUniqueId either works (PyTorch's RCCL: the id is host-side bootstrap state) or fails — then the C-ABI
    reports CLC_ERR_COMM with RCCL's message: no fallback, no crash.  A communicator itself needs a handle, i.e. a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    from camlasercalibratool_amd import solver
    try:
        uid = solver.comm_unique_id()
    except clc.ClcError as e:
        assert e.code == -8 and ("nccl" in str(e).lower() or "rccl" in str(e).lower())
    else:
        assert len(uid) == 128 and "rccl" in _capi.lib().clc_comm_library().decode()


def test_extension_is_in_tree_and_current():
    assert os.path.dirname(_build.LIB_PATH).endswith(os.path.join("camlasercalibratool_amd", "csrc"))
    assert os.path.exists(_build.LIB_PATH)
