"""The C++ drop-in header include/LaseCamCalCeres.h: compiled against a minimal Eigen stub
(Eigen is absent here) with a ROS-free port of the simulation node's main(), linked to the
C-ABI library.  CPU: it must compile, link and — without a GPU — fail loudly, not fall back.
GPU: it must recover the simulation's ground truth like the reference node does."""
import os
import re
import subprocess

import pytest

from camlasercalibratool_amd import _build

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(HERE, "dropin", "sim_main")


def _build_exe():
    src = os.path.join(HERE, "dropin", "sim_main.cpp")
    deps = [src, os.path.join(ROOT, "include", "LaseCamCalCeres.h"), os.path.join(ROOT, "include", "clc.h"), _build.LIB_PATH]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        lib_dir = os.path.dirname(_build.LIB_PATH)
        subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(HERE, "dropin", "eigen_stub"), src, "-o", EXE,
                               "-L", lib_dir, "-lclc_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_dropin_header_compiles_as_cxx11_and_links():
    """The reference builds with -std=c++11 (CMakeLists.txt:7-11)."""
    assert os.path.exists(_build_exe())


def test_dropin_reports_missing_gpu_loudly():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    p = subprocess.run([_build_exe()], capture_output=True, text=True, timeout=120)
    assert "no HIP device available" in p.stderr
    assert "Solver Summary" not in p.stdout  # nothing was computed on the CPU


@pytest.mark.gpu
@pytest.mark.parametrize("flow", ["simulation", "offline"])
def test_dropin_simulation_node_flow(flow):
    p = subprocess.run([_build_exe()] + ([flow] if flow == "offline" else []), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Solver Summary (MI355X HIP backend)" in p.stdout and "H singular values" in p.stdout and "recover chi2" in p.stdout
    if flow == "offline":
        assert "Closed-form solution Tlc" in p.stdout
    m = re.search(r"RESULT (\S+) (\S+)", p.stdout)
    et, eR = float(m.group(1)), float(m.group(2))
    tol = 1e-7 if flow == "simulation" else 2e-2   # noise-free: exact ground truth (calibr_simulation.cpp:15-20)
    assert et < tol and eR < tol, p.stdout[-600:]


@pytest.mark.gpu
def test_dropin_session_flow_at_c2_size():
    """clc_adapter::Session: closed form -> refinement -> analysis pass of main/calibr_offline.cpp:166-170 on ONE upload of the
    pose-major data (2 000 poses x ~550 points), residual blocks assembled on the device; prints the end-to-end time."""
    p = subprocess.run([_build_exe(), "session", "2000", "900"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert p.stdout.count("Closed-form solution Tlc") == 7 and p.stdout.count("recover chi2") == 7
    m = re.search(r"RESULT (\S+) (\S+)", p.stdout)
    assert float(m.group(1)) < 5e-3 and float(m.group(2)) < 5e-3, p.stdout[-600:]
    t = re.search(r"TIMING session flow: total (\S+) ms", p.stdout)
    assert t and float(t.group(1)) < 50.0
    print(re.search(r"TIMING.*", p.stdout).group(0))


@pytest.mark.gpu
def test_dropin_session_whose_scans_were_replaced_stores_them_again():
    """Two clc_adapter::Session objects alive at once: the second store replaces the first one's scans on the shared context; the
    first Session stores its own observations again (as calib.py's Session does) and recovers ITS noise-free ground truth; ok()
    reports that the calls ran."""
    p = subprocess.run([_build_exe(), "replaced"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    m = re.search(r"REPLACED ok_a=(\d) ok_b=(\d) (\S+) (\S+)", p.stdout)
    assert m and m.group(1) == "1" and m.group(2) == "1", p.stdout[-600:]
    assert float(m.group(3)) < 1e-7 and float(m.group(4)) < 1e-7, p.stdout[-600:]


@pytest.mark.gpu
def test_dropin_multistart_refines_many_initial_guesses_in_one_launch():
    """clc_adapter::Session::CalibrationFromStarts: 64 initial guesses on the same (noise-free) observations, ONE launch; the lowest final
    cost recovers the simulation's ground truth, several starts reach it, and the winner's start refined alone gives the same matrix."""
    p = subprocess.run([_build_exe(), "multistart"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    m = re.search(r"MULTISTART best=(\d+) cost=(\S+) reached=(\d+) of 64 same=(\S+) (\S+) (\S+)", p.stdout)
    assert m, p.stdout[-600:]
    assert float(m.group(2)) < 1e-12 and int(m.group(3)) >= 2 and float(m.group(4)) == 0.0
    assert float(m.group(5)) < 1e-7 and float(m.group(6)) < 1e-7, p.stdout[-300:]
