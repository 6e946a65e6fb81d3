"""Build of the HIP extension (in-tree, gfx950 only): the translation units csrc/abi_*.hip -> two libraries:
  csrc/libclc_hip.so         the product: exports include/clc.h and nothing else
  csrc/libclc_hip_hooks.so   -DCLC_TEST_HOOKS: + clc_debug_* / clc_time_* (tests, profiling scripts, bench.py's kernel-only legs)
The units are compiled in parallel (hipcc -c) and linked with hipcc -shared; objects live under csrc/.obj/<variant>/."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
PRODUCT_LIB_PATH = os.path.join(CSRC, "libclc_hip.so")
HOOKS_LIB_PATH = os.path.join(CSRC, "libclc_hip_hooks.so")
LIB_PATH = os.environ.get("CLC_LIBRARY") or PRODUCT_LIB_PATH  # CLC_LIBRARY: run the package on a different build
UNITS = ["abi_core.hip", "abi_layouts.hip", "abi_solve.hip", "abi_frontend.hip", "abi_batched.hip", "abi_comm.hip", "abi_debug.hip"]
HEADERS = ["clc_abi_internal.hpp", "clc_kernels.hpp", "clc_device.hpp", "clc_layouts.hpp", "clc_stream.hpp", "clc_controller.hpp", "clc_frontend.hpp",
           "clc_resident.hpp", "clc_coop.hpp", "clc_lmuni.hpp", "clc_rows.hpp", "clc_lm.hpp", "clc_math.hpp", "clc_host.hpp"]
SOURCES = UNITS + HEADERS
# -ffp-contract=on: FMA contraction only where the source spells one expression a*b+c (or fma()).  hipcc's default
# (fast) lets the backend fuse across statements, and it did so differently in different kernels that inline the
# same device functions — the step kernel and the [eval, lm] launch pair then differed in the last bits on 1 of 60
# random problems.  With `on` every path computes bit-identical results; same speed.
# -amdgpu-kernarg-preload-count=8: the command processor loads the first 8 kernel arguments into SGPRs while it
# dispatches the wave (gfx950 feature; kernels keep a fall-back preamble for firmware without it), so a launch does
# not begin with a kernel-argument fetch in front of its first loads (-1.5 % per solve; step_kernel orders its
# arguments for this).
# -fvisibility=hidden: the library exports what include/clc.h declares (clc_abi_internal.hpp includes it under
# `#pragma GCC visibility push(default)`) — the helpers shared between the units stay internal.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fvisibility=hidden",
               "-mllvm", "-amdgpu-kernarg-preload-count=8"]


def csrc_sha16() -> str:
    """Identity of what the KERNELS are built from: sha256 over the bytes of every csrc header (HEADERS order: all device code lives in
    the .hpp files; the abi_*.hip units are the host side — launchers, checks, RCCL calls) and the compiler flags.  Constants that were
    measured on a build (the VALU instruction counts bench.py prices the whole-solve kernels with, profiles/valu_counts.json) carry
    it, and bench.py refuses them when it differs from the sources in the tree."""
    import hashlib
    h = hashlib.sha256()
    for s in HEADERS:
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (no CPU fallback exists)")


def _is_stale(path: str) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "clc.h"),
                                                       os.path.abspath(__file__)]  # this file holds the compiler flags
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def is_stale() -> bool:
    return _is_stale(LIB_PATH)


def build_variant(out_path: str, defines=(), verbose: bool = False, units=None) -> str:
    """Compile every unit with the given -D flags (in parallel) and link them into out_path."""
    tag = os.path.splitext(os.path.basename(out_path))[0]
    objdir = os.path.join(CSRC, ".obj", tag)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(unit):
        obj = os.path.join(objdir, os.path.splitext(unit)[0] + ".o")
        cmd = [hipcc] + HIPCC_FLAGS + list(defines) + ["-c", os.path.join(CSRC, unit), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, units or UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out_path + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(out_path + ".tmp", out_path)
    return out_path


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """The product library (cross-compiles without a GPU)."""
    if not force and not _is_stale(PRODUCT_LIB_PATH):
        return PRODUCT_LIB_PATH
    return build_variant(PRODUCT_LIB_PATH, (), verbose)


def build_hooks_extension(force: bool = False, verbose: bool = False) -> str:
    """The product library + the test / profiling hooks (clc_debug_*, clc_time_*)."""
    if not force and not _is_stale(HOOKS_LIB_PATH):
        return HOOKS_LIB_PATH
    return build_variant(HOOKS_LIB_PATH, ("-DCLC_TEST_HOOKS",), verbose)
