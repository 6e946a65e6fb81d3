"""ctypes binding of the CPU oracle (oracle/clc_oracle.cpp -> oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Parity: the reference-owned arithmetic is pinned against the reference's own
code (oracle/ref.py, tests/test_ref_pin.py); Ceres' minimiser is restated and UNPINNED (header of clc_oracle.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package camlasercalibratool_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

TERMINATION = {
    1: "CONVERGENCE(gradient)",
    2: "CONVERGENCE(parameter)",
    3: "CONVERGENCE(function)",
    4: "CONVERGENCE(radius)",
    5: "NO_CONVERGENCE",
    6: "FAILURE",
}


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
        ("num_residual_evaluations", C.c_int64),
        ("num_jacobian_evaluations", C.c_int64),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "clc_oracle.cpp")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "oracle_types.h")))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)
_llp = C.POINTER(C.c_longlong)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_evaluate.restype = C.c_double
        L.oracle_evaluate_ne.restype = C.c_double
        L.oracle_flatten.restype = C.c_longlong
        L.oracle_solve.restype = C.c_int
        L.oracle_closed_form.restype = C.c_int
        L.oracle_max_threads.restype = C.c_int
        L.oracle_line_fit.restype = C.c_int
        L.oracle_line_evaluate.restype = C.c_double
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def _pl(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_llp)


def default_options() -> Options:
    o = Options()
    lib().oracle_options_default(C.byref(o))
    return o


def quat_to_rot(q_xyzw: np.ndarray) -> np.ndarray:
    R = np.empty(9)
    lib().oracle_quat_to_rot(_p(np.ascontiguousarray(q_xyzw, dtype=np.float64)), _p(R))
    return R.reshape(3, 3)


def rot_to_quat(R: np.ndarray) -> np.ndarray:
    q = np.empty(4)
    lib().oracle_rot_to_quat(_p(np.ascontiguousarray(R, dtype=np.float64).reshape(9)), _p(q))
    return q


def factor_evaluate(plane, point, scale, pose, want_jac=True):
    r = np.empty(1)
    j = np.empty(7) if want_jac else None
    lib().oracle_factor_evaluate(
        _p(np.ascontiguousarray(plane, dtype=np.float64)),
        _p(np.ascontiguousarray(point, dtype=np.float64)),
        C.c_double(scale),
        _p(np.ascontiguousarray(pose, dtype=np.float64)),
        _p(r),
        _p(j),
    )
    return r[0], j


def factor_evaluate_batch(obs: np.ndarray, pose: np.ndarray, want_jac=True):
    N = obs.shape[0]
    r = np.empty(N)
    j = np.empty((N, 7)) if want_jac else None
    lib().oracle_factor_evaluate_batch(_p(obs), C.c_longlong(N), _p(np.ascontiguousarray(pose)), _p(r), _p(j))
    return r, j


def pose_plus(x, delta) -> np.ndarray:
    out = np.empty(7)
    lib().oracle_pose_plus(_p(np.ascontiguousarray(x, dtype=np.float64)), _p(np.ascontiguousarray(delta, dtype=np.float64)), _p(out))
    return out


def pose_plus_jacobian(x) -> np.ndarray:
    out = np.empty(42)
    lib().oracle_pose_plus_jacobian(_p(np.ascontiguousarray(x, dtype=np.float64)), _p(out))
    return out.reshape(7, 6)


def cauchy(a: float, s: float) -> np.ndarray:
    rho = np.empty(3)
    lib().oracle_cauchy(C.c_double(a), C.c_double(s), _p(rho))
    return rho


def pi_from_ppp(x1, x2, x3) -> np.ndarray:
    pi = np.empty(4)
    lib().oracle_pi_from_ppp(_p(np.ascontiguousarray(x1, dtype=np.float64)), _p(np.ascontiguousarray(x2, dtype=np.float64)),
                             _p(np.ascontiguousarray(x3, dtype=np.float64)), _p(pi))
    return pi


def flatten(obs_set, use_linefitting_data: bool, use_boundary_constraint: bool) -> np.ndarray:
    """Assembly loop LaseCamCalCeres.cpp:222-295 -> records [N,8]."""
    args = (
        C.c_int(obs_set.n_poses), _p(np.ascontiguousarray(obs_set.tag_q)), _p(np.ascontiguousarray(obs_set.tag_t)),
        _pl(obs_set.pts_off), _p(obs_set.pts), _pl(obs_set.ptl_off), _p(obs_set.ptl),
        C.c_int(int(use_linefitting_data)), C.c_int(int(use_boundary_constraint)),
    )
    n = lib().oracle_flatten(*args, None)
    if n < 0:
        raise IndexError("obi.points.at(0): empty scan in boundary mode (LaseCamCalCeres.cpp:278)")
    rec = np.empty((n, 8))
    lib().oracle_flatten(*args, _p(rec))
    return rec


def evaluate(obs: np.ndarray, pose: np.ndarray, with_loss=True, loss_factor=0.05, want_jac=False):
    """One Ceres evaluation pass.  Returns (cost, residuals[N], J[N,6] or None, g[6])."""
    N = obs.shape[0]
    res = np.empty(N)
    Jc = np.empty((6, N)) if want_jac else None  # column-major N x 6
    g = np.empty(6)
    cost = lib().oracle_evaluate(_p(obs), C.c_longlong(N), _p(np.ascontiguousarray(pose)), C.c_int(int(with_loss)),
                                 C.c_double(loss_factor), _p(res), _p(Jc), C.c_longlong(N), _p(g))
    return cost, res, (Jc.T if want_jac else None), g


def evaluate_ne(obs: np.ndarray, pose: np.ndarray, with_loss=True, loss_factor=0.05, threads=1):
    """Normal-equation evaluation: (cost, g[6], H[21] upper-tri row-major)."""
    g = np.empty(6)
    H = np.empty(21)
    cost = lib().oracle_evaluate_ne(_p(obs), C.c_longlong(obs.shape[0]), _p(np.ascontiguousarray(pose)),
                                    C.c_int(int(with_loss)), C.c_double(loss_factor), _p(g), _p(H), C.c_int(threads))
    return cost, g, H


@dataclass
class SolveResult:
    pose: np.ndarray
    summary: Summary
    trace: List[Iteration]

    @property
    def termination(self) -> str:
        return TERMINATION.get(self.summary.termination, "?")


def solve(obs: np.ndarray, pose0: np.ndarray, options: Optional[Options] = None, linear_solver: str = "qr",
          threads: int = 1, trace_cap: int = 256) -> SolveResult:
    """ceres::Solve restatement.  linear_solver 'qr' (reference: DENSE_QR) or 'ne'."""
    o = options or default_options()
    pose = np.array(pose0, dtype=np.float64).copy()
    s = Summary()
    tr = (Iteration * trace_cap)()
    lib().oracle_solve(_p(obs), C.c_longlong(obs.shape[0]), C.byref(o), _p(pose), C.byref(s), tr, C.c_int(trace_cap),
                       C.c_int(0 if linear_solver == "qr" else 1), C.c_int(threads))
    n = min(trace_cap, s.num_iterations + 1)
    return SolveResult(pose, s, [tr[i] for i in range(n)])


def information(obs: np.ndarray, pose: np.ndarray):
    """Analysis pass LaseCamCalCeres.cpp:316-381 -> (H[6,6], b[6], chi2, sv[6], V[6,6], n_null)."""
    H = np.empty(36); b = np.empty(6); chi = C.c_double(); sv = np.empty(6); V = np.empty(36); nn = C.c_int()
    lib().oracle_information(_p(obs), C.c_longlong(obs.shape[0]), _p(np.ascontiguousarray(pose)), _p(H), _p(b),
                             C.byref(chi), _p(sv), _p(V), C.byref(nn))
    return H.reshape(6, 6), b, chi.value, sv, V.reshape(6, 6), nn.value


def closed_form(obs: np.ndarray):
    """CamLaserCalClosedSolution, LaseCamCalCeres.cpp:112-203 -> (Tlc[4,4], unobservable, sv9)."""
    T = np.empty(16); un = C.c_int(); sv = np.empty(9)
    rc = lib().oracle_closed_form(_p(obs), C.c_longlong(obs.shape[0]), _p(T), C.byref(un), _p(sv))
    if rc != 0:
        raise np.linalg.LinAlgError("closed form: 9x9 solve failed")
    return T.reshape(4, 4), bool(un.value), sv


def max_threads() -> int:
    return lib().oracle_max_threads()


def default_line_options() -> Options:
    """Ceres defaults + max_num_iterations = 10 (src/LaseCamCalCeres.cpp:425)."""
    o = default_options()
    o.max_num_iterations = 10
    return o


def line_evaluate(xy: np.ndarray, line, with_loss=True, loss_a=0.05):
    """-> (cost, g[2], H3 = [H00, H01, H11]) of the line-fit problem at `line`."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    g = np.empty(2); H = np.empty(3)
    c = lib().oracle_line_evaluate(_p(xy), C.c_longlong(xy.shape[0]), _p(np.ascontiguousarray(line, dtype=np.float64)),
                                   C.c_int(int(with_loss)), C.c_double(loss_a), None, None, C.c_longlong(0), _p(g), _p(H))
    return c, g, H


def line_fit(xy: np.ndarray, line0, options: Optional[Options] = None, loss_a: float = 0.05,
             linear_solver: str = "qr", trace_cap: int = 64) -> SolveResult:
    """LineFittingCeres restatement (src/LaseCamCalCeres.cpp:401-433): result.pose = (m0, m1)."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    o = options or default_line_options()
    line = np.array(line0, dtype=np.float64).copy()
    s = Summary()
    tr = (Iteration * trace_cap)()
    lib().oracle_line_fit(_p(xy), C.c_longlong(xy.shape[0]), C.byref(o), C.c_double(loss_a), _p(line), C.byref(s), tr,
                          C.c_int(trace_cap), C.c_int(0 if linear_solver == "qr" else 1))
    n = min(trace_cap, s.num_iterations + 1)
    return SolveResult(line, s, [tr[i] for i in range(n)])


def scan_to_points(ranges: np.ndarray, angle_min: float, angle_increment: float, range_min: float) -> np.ndarray:
    """TranScanToPoints (src/utilities.cpp:181-215) for one scan -> points [n,3]."""
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    pts = np.empty((r.shape[0], 3))
    lib().oracle_scan_to_points(r.ctypes.data_as(C.POINTER(C.c_float)), C.c_longlong(r.shape[0]), C.c_float(angle_min),
                                C.c_float(angle_increment), C.c_float(range_min), _p(pts))
    return pts
