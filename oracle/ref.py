"""ctypes binding of oracle/_ref/libref.so — the REFERENCE'S OWN translation units
(src/LaseCamCalCeres.cpp, src/pose_local_parameterization.cpp, src/utilities.cpp, main/calibr_simulation.cpp) compiled from where
they lie under /root/reference against the stand-in Eigen / Ceres / sensor_msgs headers of
oracle/ref_shim/ (neither library is installed here; see oracle/ref_shim/mini_eigen.hpp).

TEST INFRASTRUCTURE ONLY: used by tests/ and by tests/golden/make_ref_golden.py to check the oracle's
restatement of the reference-owned arithmetic against the reference's code itself.  The minimiser
behind ceres::Solve in that build is the oracle's LM restatement, so iteration semantics of Ceres
are NOT pinned by this library — only what the reference's own sources compute.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref.so")
REF_ROOT = os.environ.get("CLC_REF_ROOT", "/root/reference")
_dp = C.POINTER(C.c_double)
_llp = C.POINTER(C.c_longlong)


def reference_present() -> bool:
    return os.path.exists(os.path.join(REF_ROOT, "src", "LaseCamCalCeres.cpp"))


def build(force: bool = False) -> str:
    """make -C oracle ref (needs the reference checkout; g++, a few seconds)."""
    if not reference_present():
        raise FileNotFoundError(f"reference sources not found under {REF_ROOT}")
    cmd = ["make", "-C", _HERE, f"REF_ROOT={REF_ROOT}", "ref"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def available() -> bool:
    """True if libref.so exists (prebuilt) or can be built here."""
    if reference_present():
        try:
            build()
        except Exception:
            return os.path.exists(LIB_PATH)
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if reference_present():
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def _pl(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_llp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def factor_evaluate(plane, point, scale, pose, want_jac=True):
    """PointInPlaneFactor::Evaluate (src/LaseCamCalCeres.cpp:43-66) -> (residual, jacobian[7] | None)."""
    r = np.empty(1)
    j = np.full(7, np.nan) if want_jac else None
    lib().ref_factor_evaluate(_p(_f64(plane)), _p(_f64(point)), C.c_double(scale), _p(_f64(pose)), _p(r), _p(j))
    return r[0], j


def pose_plus(x, delta) -> np.ndarray:
    out = np.empty(7)
    lib().ref_pose_plus(_p(_f64(x)), _p(_f64(delta)), _p(out))
    return out


def pose_plus_jacobian(x) -> np.ndarray:
    out = np.full(42, np.nan)
    lib().ref_pose_plus_jacobian(_p(_f64(x)), _p(out))
    return out.reshape(7, 6)


def pose_sizes():
    g, l = C.c_int(), C.c_int()
    lib().ref_pose_sizes(C.byref(g), C.byref(l))
    return g.value, l.value


def pi_from_ppp(x1, x2, x3) -> np.ndarray:
    pi = np.empty(4)
    lib().ref_pi_from_ppp(_p(_f64(x1)), _p(_f64(x2)), _p(_f64(x3)), _p(pi))
    return pi


def _obs_args(S):
    return (C.c_int(S.n_poses), _p(_f64(S.tag_q)), _p(_f64(S.tag_t)), _pl(S.pts_off), _p(S.pts), _pl(S.ptl_off), _p(S.ptl))


def closed_solution(S, Tlc0=None) -> np.ndarray:
    """CamLaserCalClosedSolution(obs, Tlc) (src/LaseCamCalCeres.cpp:112-203) -> Tlc[4,4]."""
    T = _f64(np.eye(4) if Tlc0 is None else Tlc0).reshape(16).copy()
    lib().ref_closed_solution(*_obs_args(S), _p(T))
    return T.reshape(4, 4)


def calibration(S, Tcl0, use_linefitting_data=True, use_boundary_constraint=False):
    """CamLaserCalibration(obs, Tcl, ...) (src/LaseCamCalCeres.cpp:213-383) ->
    (Tcl[4,4], record dict of the stand-in ceres::Solve, number of residual blocks added)."""
    T = _f64(Tcl0).reshape(16).copy()
    rec = np.empty(6)
    nb = C.c_longlong()
    rc = lib().ref_calibration(*_obs_args(S), _p(T), C.c_int(int(use_linefitting_data)), C.c_int(int(use_boundary_constraint)),
                               _p(rec), C.byref(nb))
    if rc != 0:
        raise IndexError("obi.points.at(0): std::out_of_range thrown by the reference (LaseCamCalCeres.cpp:278)")
    record = dict(termination=int(rec[0]), num_iterations=int(rec[1]), num_successful_steps=int(rec[2]),
                  num_unsuccessful_steps=int(rec[3]), initial_cost=rec[4], final_cost=rec[5])
    return T.reshape(4, 4), record, nb.value


def line_fitting(points_xyz, line0):
    """LineFittingCeres(Points, Line) (src/LaseCamCalCeres.cpp:401-433) -> (line[2], record)."""
    P = _f64(points_xyz).reshape(-1, 3)
    line = _f64(line0).copy()
    rec = np.empty(6)
    lib().ref_line_fitting(_p(P), C.c_longlong(P.shape[0]), _p(line), _p(rec))
    record = dict(termination=int(rec[0]), num_iterations=int(rec[1]), num_successful_steps=int(rec[2]),
                  num_unsuccessful_steps=int(rec[3]), initial_cost=rec[4], final_cost=rec[5])
    return line, record


def scan_to_points(ranges, angle_min, angle_increment, range_min) -> np.ndarray:
    """TranScanToPoints (src/utilities.cpp:181-215) -> points [n,3]."""
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    out = np.empty((r.shape[0], 3))
    lib().ref_scan_to_points(r.ctypes.data_as(C.POINTER(C.c_float)), C.c_longlong(r.shape[0]), C.c_float(angle_min),
                             C.c_float(angle_increment), C.c_float(range_min), _p(out))
    return out


def last_stdout() -> str:
    """What the last closed_solution() / calibration() call printed on std::cout (the reference reports its
    analysis pass — H singular values, null space, chi2 — only there, src/LaseCamCalCeres.cpp:365-381)."""
    L = lib()
    L.ref_last_stdout.restype = C.c_longlong
    n = L.ref_last_stdout(None, C.c_longlong(0))
    buf = C.create_string_buffer(int(n) + 1)
    L.ref_last_stdout(buf, C.c_longlong(n + 1))
    return buf.value.decode("utf-8", "replace")


def parse_analysis(text: str):
    """-> (singular values [6], chi2/2) from the analysis printout (default ostream precision: 6 digits)."""
    lines = text.splitlines()
    i = next(k for k, l in enumerate(lines) if "H singular values" in l)
    sv = [float(lines[i + 1 + k]) for k in range(6)]
    chi_half = float(next(l for l in lines if l.startswith("recover chi2:")).split(":")[1])
    return np.array(sv), chi_half


def generate_sim_data(seed: int):
    """GenerateSimData (main/calibr_simulation.cpp:8-108) with std::random_device pinned to `seed` ->
    (tag_q_wxyz [50,4], tag_t [50,3], counts [50], points [n,3])."""
    L = lib()
    L.ref_generate_sim_data.restype = C.c_longlong
    q = np.empty((50, 4)); t = np.empty((50, 3)); cnt = np.zeros(50, dtype=np.int64)
    cap = 50 * 180
    pts = np.empty((cap, 3))
    n = L.ref_generate_sim_data(C.c_uint(seed), _p(q), _p(t), _pl(cnt), _p(pts), C.c_longlong(cap))
    return q, t, cnt, pts[:n].copy()


def simulation_program(seed: int) -> str:
    """The reference's whole simulation node (its main(), main/calibr_simulation.cpp:110-165) on the pinned seed;
    returns everything it printed."""
    rc = lib().ref_simulation_program(C.c_uint(seed))
    if rc != 0:
        raise RuntimeError(f"reference main returned {rc}")
    return last_stdout()


def parse_simulation_tlc(text: str) -> np.ndarray:
    """The 4x4 the program prints after '----- Transform from Camera to Laser Tlc is: -----' (6 significant digits)."""
    lines = text.splitlines()
    i = next(k for k, l in enumerate(lines) if "Transform from Camera to Laser Tlc is" in l)
    rows = [l for l in lines[i + 1:] if l.strip()][:4]
    return np.array([[float(v) for v in r.split()] for r in rows])
