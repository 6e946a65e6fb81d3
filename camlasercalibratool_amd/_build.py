"""Build of the HIP extension (in-tree, gfx950 only): csrc/clc_abi.hip -> csrc/libclc_hip.so."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("CLC_LIBRARY") or os.path.join(CSRC, "libclc_hip.so")  # CLC_LIBRARY: A/B a different build
SOURCES = ["clc_abi.hip", "clc_kernels.hpp", "clc_device.hpp", "clc_layouts.hpp", "clc_stream.hpp", "clc_controller.hpp", "clc_frontend.hpp",
           "clc_legacy.hpp", "clc_resident.hpp", "clc_coop.hpp", "clc_lmuni.hpp", "clc_rows.hpp", "clc_lm.hpp", "clc_math.hpp", "clc_host.hpp"]
# -ffp-contract=on: FMA contraction only where the source spells one expression a*b+c (or fma()).  hipcc's default
# (fast) lets the backend fuse across statements, and it did so differently in different kernels that inline the
# same device functions — the step kernel and the [eval, lm] launch pair then differed in the last bits on 1 of 60
# random problems.  With `on` every path computes bit-identical results; same speed.
# -amdgpu-kernarg-preload-count=8: the command processor loads the first 8 kernel arguments into SGPRs while it
# dispatches the wave (gfx950 feature; kernels keep a fall-back preamble for firmware without it), so a launch does
# not begin with a kernel-argument fetch in front of its first loads (-1.5 % per solve; step_kernel orders its
# arguments for this).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=on",
               "-mllvm", "-amdgpu-kernarg-preload-count=8"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (no CPU fallback exists)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "clc.h"),
                                                       os.path.abspath(__file__)]  # this file holds the compiler flags
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


LEGACY_LIB_PATH = os.path.join(CSRC, "libclc_hip_legacy.so")  # the same library + the paths of clc_legacy.hpp (tests, A/B)


def _is_stale(path: str) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "clc.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """Compile the extension with hipcc for gfx950 (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, "clc_abi.hip"), "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def build_legacy_extension(force: bool = False, verbose: bool = False) -> str:
    """The -DCLC_LEGACY_PATHS build: the default library plus the superseded paths (clc_legacy.hpp) that the bit-identity
    tests compare against (tests run with CLC_LIBRARY=<this file>)."""
    if not force and not _is_stale(LEGACY_LIB_PATH):
        return LEGACY_LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-DCLC_LEGACY_PATHS", os.path.join(CSRC, "clc_abi.hip"), "-o", LEGACY_LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LEGACY_LIB_PATH + ".tmp", LEGACY_LIB_PATH)
    return LEGACY_LIB_PATH
