"""ctypes binding of the C-ABI (include/clc.h) exported by csrc/libclc_hip.so.

There is no CPU fallback: if the HIP extension is missing this module raises on first use,
and every entry point that needs the device raises when no gfx950 GPU is present."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

CLC_OK = 0
ERRORS = {
    -1: "CLC_ERR_INVALID_ARG", -2: "CLC_ERR_HIP", -3: "CLC_ERR_NONFINITE", -4: "CLC_ERR_EMPTY_SCAN",
    -5: "CLC_ERR_NO_DATA", -6: "CLC_ERR_LINALG", -7: "CLC_ERR_NO_DEVICE", -8: "CLC_ERR_COMM",
}
TERMINATION = {
    0: "RUNNING", 1: "CONVERGENCE(gradient)", 2: "CONVERGENCE(parameter)", 3: "CONVERGENCE(function)",
    4: "CONVERGENCE(radius)", 5: "NO_CONVERGENCE", 6: "FAILURE",
}

# every symbol include/clc.h declares (checked by tests/test_abi_symbols.py)
EXPORTED = [
    "clc_version", "clc_last_error", "clc_options_default", "clc_create", "clc_destroy", "clc_set_stream",
    "clc_set_launch", "clc_set_auto_paths", "clc_set_small_on_coop", "clc_flatten_observations", "clc_upload", "clc_upload_device", "clc_num_observations",
    "clc_factor_evaluate", "clc_pose_plus", "clc_pose_plus_jacobian", "clc_eval", "clc_solve",
    "clc_information", "clc_closed_form", "clc_upload_batched", "clc_solve_batched", "clc_solve_multistart", "clc_num_problems",
    "clc_line_options_default", "clc_line_fit_batched", "clc_scan_to_points",
    "clc_comm_unique_id", "clc_comm_create", "clc_comm_destroy", "clc_comm_rank", "clc_comm_world",
    "clc_gather_results", "clc_comm_records", "clc_solve_batched_gather", "clc_comm_set_root", "clc_comm_get_info",
    "clc_solve_batched_gather_pipelined", "clc_gather_flush",
    "clc_store_observations", "clc_select_observations", "clc_upload_batched_device", "clc_line_fit_batched_device",
    "clc_scan_to_points_device", "clc_pinned_alloc", "clc_pinned_free", "clc_store_generation", "clc_batched_host_buffers",
    "clc_get_path_info", "clc_device_info", "clc_comm_library",
]
# test / profiling hooks: NOT in include/clc.h and not in the product library; exported by the -DCLC_TEST_HOOKS builds
# (csrc/libclc_hip_hooks.so) only (tests/test_abi_symbols.py checks both directions)
HOOKS = [
    "clc_debug_flatten_device", "clc_debug_math", "clc_debug_wave_reduce", "clc_debug_build_features", "clc_debug_rows",
    "clc_debug_wave_split", "clc_debug_resident", "clc_debug_resident_single", "clc_debug_coop", "clc_debug_coop_control",
    "clc_debug_coop_set_tag", "clc_debug_layout", "clc_debug_lm_profile", "clc_time_steps", "clc_time_batched_eval", "clc_time_eval",
    "clc_debug_comm_create_layout", "clc_debug_single_controller", "clc_debug_fast_small",
]


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("use_loss", C.c_int32),
        ("loss_scale_factor", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("launch_ahead", C.c_int32),
        ("profile_events", C.c_int32),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("step_is_valid", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("pad_", C.c_int32),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("num_evaluations", C.c_int64),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("solve_ms", C.c_double),
        ("eval_kernel_ms", C.c_double),
        ("eval_kernel_launches", C.c_int64),
    ]


class BatchStats(C.Structure):
    """clc_batch_stats (include/clc.h): the local shard's totals of one clc_solve_batched_gather."""
    _fields_ = [("problems", C.c_int64), ("evaluations", C.c_int64), ("iterations", C.c_int64), ("not_converged", C.c_int64),
                ("fused", C.c_int32), ("pad_", C.c_int32), ("kernel_ms", C.c_double), ("solve_ms", C.c_double)]


class CommInfo(C.Structure):
    """clc_comm_info (include/clc.h)."""
    _fields_ = [(n, C.c_int32) for n in ("struct_size", "rank", "world", "root", "rooted_collective_available", "copies_other_ranks_to_host",
                                         "step_in_flight", "pad_")] + \
               [(n, C.c_int64) for n in ("collectives", "rooted_collectives", "host_copies", "host_copy_bytes", "pipelined_steps")]


class PathInfo(C.Structure):
    """clc_path_info (include/clc.h)."""
    _fields_ = [(n, C.c_int32) for n in ("single_resident", "single_lanes", "single_points_per_lane", "coop_resident", "coop_points_per_lane",
                                         "coop_points_carry_z", "coop_resting", "coop_timeouts", "batched_resident", "batched_lanes", "batched_points_per_lane",
                                         "rows_layout", "batched_rows_layout", "coop_workgroups", "batched_points_carry_z", "reserved_")] + \
               [(n, C.c_int64) for n in ("coop_solves", "batched_lane_rows", "n_rows", "batched_n_rows", "coop_gate_waits_expired")]


class ClcError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str):
        super().__init__(f"{where}: {ERRORS.get(code, code)} — {detail}")
        self.code = code


_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def _preload_process_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's).  If this extension pulled in the system copy first, a later
    `import torch` would find no GPUs; so when torch is installed (it is the plumbing for
    device memory / streams / torch.distributed), its runtime is loaded first and the extension
    binds to it.  CLC_HIP_RUNTIME=system skips this."""
    if os.environ.get("CLC_HIP_RUNTIME", "").lower() == "system":
        return
    import sys
    if "torch" in sys.modules:
        return  # torch already brought its runtime in
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to whatever the dynamic loader resolves


_libs = {}


def load(path: str):
    """Load one build of the library (cached per path) and declare the non-int signatures."""
    path = os.path.abspath(path)
    L = _libs.get(path)
    if L is None:
        if not os.path.exists(path):
            raise RuntimeError(
                f"HIP extension {path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(needs hipcc).  camlasercalibratool_amd has no CPU fallback.")
        _preload_process_hip_runtime()
        L = C.CDLL(path)
        L.clc_last_error.restype = C.c_char_p
        L.clc_num_observations.restype = C.c_size_t
        L.clc_num_problems.restype = C.c_size_t
        L.clc_num_observations.argtypes = [C.c_void_p]
        L.clc_num_problems.argtypes = [C.c_void_p]
        L.clc_destroy.argtypes = [C.c_void_p]
        L.clc_store_generation.argtypes = [C.c_void_p]
        L.clc_store_generation.restype = C.c_int64
        L.clc_destroy.restype = None
        L.clc_comm_destroy.argtypes = [C.c_void_p]
        L.clc_comm_destroy.restype = None
        L.clc_comm_rank.argtypes = [C.c_void_p]
        L.clc_comm_world.argtypes = [C.c_void_p]
        L.clc_comm_library.restype = C.c_char_p
        L.clc_solve_batched_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_size_t, C.c_void_p, C.c_void_p]
        L.clc_solve_batched_gather.restype = C.c_int
        L.clc_solve_batched_gather_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_size_t, C.c_void_p, C.c_void_p]
        L.clc_solve_batched_gather_pipelined.restype = C.c_int
        L.clc_gather_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.clc_gather_flush.restype = C.c_int
        L.clc_comm_set_root.argtypes = [C.c_void_p, C.c_int]
        L.clc_comm_get_info.argtypes = [C.c_void_p, C.c_void_p]
        L.clc_comm_records.argtypes = [C.c_void_p]
        L.clc_comm_records.restype = C.POINTER(C.c_double)
        L.clc_get_path_info.argtypes = [C.c_void_p, C.c_void_p]
        L.has_hooks = hasattr(L, "clc_debug_build_features")
        _libs[path] = L
    return L


def lib():
    """The library the package runs on: csrc/libclc_hip.so (the product build) unless CLC_LIBRARY names another build."""
    global _lib
    if _lib is None:
        _lib = load(_build.LIB_PATH)
    return _lib


def hooks_lib():
    """The -DCLC_TEST_HOOKS build (product + clc_debug_* / clc_time_*): the default library itself when it has the hooks."""
    L = lib()
    return L if L.has_hooks else load(_build.HOOKS_LIB_PATH)


def check(rc: int, where: str) -> None:
    if rc != CLC_OK:
        # clc_last_error is per library (and thread): with more than one build loaded, take the message of the call that failed
        msgs = [L.clc_last_error().decode("utf-8", "replace") for L in _libs.values()]
        detail = next((m for m in msgs if m.startswith(where)), next((m for m in msgs if m), ""))
        raise ClcError(rc, where, detail)


def dptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "need C-contiguous float64"
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.int64 and a.flags["C_CONTIGUOUS"], "need C-contiguous int64"
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def default_options() -> Options:
    o = Options()
    lib().clc_options_default(C.byref(o))
    return o


def default_line_options() -> Options:
    """Ceres defaults + LineFittingCeres' settings (10 iterations, CauchyLoss(0.05))."""
    o = Options()
    lib().clc_line_options_default(C.byref(o))
    return o
