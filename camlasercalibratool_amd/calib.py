"""Host-side mirror of the reference's call surface (include/LaseCamCalCeres.h:27-28):

    CamLaserCalClosedSolution(obs, Tlc)                          src/LaseCamCalCeres.cpp:112
    CamLaserCalibration(obs, Tcl, use_linefitting_data=True,
                        use_boundary_constraint=False)           src/LaseCamCalCeres.cpp:213

Same names, argument meaning and in/out conventions (4x4 matrices are modified in place, like
the Eigen::Matrix4d& of the reference); the bodies flatten the observations and call the HIP
library through the C-ABI.  Diagnostics the reference prints to stdout (summary, singular
values of H, null space, "recover chi2") are printed too and also returned."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import simdata
from ._capi import Options, TERMINATION, default_line_options, default_options
from .simdata import Oberserve, ObservationSet
from .solver import SolveResult, Solver, flatten_observations

ObsLike = Union[ObservationSet, Sequence[Oberserve]]


_shared: Optional[Solver] = None


def _shared_solver() -> Solver:
    """One solver context per process for the calls below (like the C++ drop-in header): creating a
    context costs milliseconds, the calls themselves a few hundred microseconds."""
    global _shared
    if _shared is None or _shared._h is None:
        _shared = Solver()
    return _shared


def _as_set(obs: ObsLike) -> ObservationSet:
    return obs if isinstance(obs, ObservationSet) else ObservationSet.from_list(list(obs))


@dataclass
class CalibrationReport:
    result: SolveResult
    H: np.ndarray
    b: np.ndarray
    chi2: float
    singular_values: np.ndarray
    null_space: np.ndarray  # [6, n_null]


class Session:
    """The observations of one calibration run resident on the GPU: the pose-major data of std::vector<Oberserve> (tag
    poses + scan points) is uploaded ONCE; closed form, refinement and the analysis pass — the sequence of
    main/calibr_offline.cpp:166-170 — then select their residual blocks on the device (clc_select_observations) instead
    of re-flattening and re-uploading 64-byte records per call.  The module-level functions below are one-call sessions."""

    def __init__(self, obs: ObsLike, solver: Optional[Solver] = None):
        self.S = _as_set(obs)
        self.sv = solver or _shared_solver()
        self._store()

    def _store(self):
        self.sv.store_observations(self.S)
        self._generation = self.sv.store_generation

    def _ensure_stored(self):
        """The stored scans belong to the (possibly shared) solver handle: if somebody else — another Session, one of the
        module-level calls below — stored theirs since, put ours back before selecting from them."""
        if self.sv.store_generation != self._generation:
            self._store()

    def CamLaserCalClosedSolution(self, Tlc: np.ndarray, verbose: bool = True):
        """Closed-form initialiser; overwrites Tlc (camera->laser) like LaseCamCalCeres.cpp:198-200."""
        sv = self.sv
        self._ensure_stored()
        sv.select_observations(True, False)  # points_on_line only, :143
        T, unobservable, sv9 = sv.closed_form()
        if unobservable and verbose:  # :173-178
            print("\n~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~")
            print(" Notice Notice Notice: system unobservable !!!!!!!")
            print("~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~\n")
        Tlc[...] = T
        if verbose:
            print("------- Closed-form solution Tlc: -------\n", Tlc)  # :202
        return unobservable, sv9

    def CamLaserCalibration(self, Tcl: np.ndarray, use_linefitting_data: bool = True, use_boundary_constraint: bool = False,
                            options: Optional[Options] = None, verbose: bool = True) -> CalibrationReport:
        """Nonlinear refinement; Tcl (laser->camera) is in/out like LaseCamCalCeres.cpp:215,:311-314."""
        sv = self.sv
        self._ensure_stored()
        n_rec = sv.select_observations(use_linefitting_data, use_boundary_constraint)
        pose0 = simdata.pose7_from_T(np.asarray(Tcl, dtype=np.float64))  # :215-219
        res = sv.solve(pose0, options)
        if verbose:  # stands in for summary.FullReport(), :309: the per-iteration table, then the totals
            print("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  step")
            for it in res.trace:
                print(f"{it.iteration:4d}  {it.cost:.6e}  {it.cost_change:9.2e}  {it.gradient_max_norm:9.2e}  {it.step_norm:9.2e}  "
                      f"{it.relative_decrease:9.2e}  {it.trust_region_radius:9.2e}  "
                      f"{'invalid' if not it.step_is_valid else ('accepted' if it.step_is_successful else 'rejected')}")
            s = res.summary
            print(f"Solver Summary: iterations {s.num_iterations} (successful {s.num_successful_steps - 1}, "
                  f"unsuccessful {s.num_unsuccessful_steps}), initial cost {s.initial_cost:.6e}, "
                  f"final cost {s.final_cost:.6e}, termination {TERMINATION.get(s.termination)}, "
                  f"residuals {n_rec}, passes {s.num_evaluations}, time {s.solve_ms:.3f} ms")
        Tcl[...] = simdata.T_from_pose7(res.pose)  # :311-314
        # analysis pass: no loss, no boundary terms (:316-362) — re-selected on the device, nothing crosses PCIe
        if use_boundary_constraint and use_linefitting_data:
            sv.select_observations(use_linefitting_data, False)
        H, b, chi2, svals, V, n_null = sv.information(res.pose)
        null = V[:, 6 - n_null:] if n_null > 0 else np.zeros((6, 0))
        if verbose:  # :365-381
            print("----- H singular values--------:")
            print(svals)
            if n_null > 0:
                print("====== null space basis, it's means the unobservable direction for Tcl ======")
                print("       please note the unobservable direction is for Tcl, not for Tlc        ")
                print(null)
            print("\nrecover chi2: ", chi2 / 2.0)
        return CalibrationReport(res, H, b, chi2, svals, null)


def CamLaserCalibrationFromStarts(obs: ObsLike, Tcls: np.ndarray, use_linefitting_data: bool = True, use_boundary_constraint: bool = False,
                                  options: Optional[Options] = None, solver: Optional[Solver] = None):
    """Multi-hypothesis refinement (mirror of clc_adapter::Session::CalibrationFromStarts): every Tcl of Tcls [S, 4, 4] refined on the SAME
    observations by clc_solve_multistart — a workgroup per start on ONE copy of the data.  Tcls is overwritten with the refined matrices;
    -> (index of the lowest final cost, final costs [S], summaries)."""
    S = _as_set(obs)
    sv = solver or _shared_solver()
    rec = flatten_observations(S, use_linefitting_data, use_boundary_constraint)
    sv.upload_batched(rec, np.array([0, rec.shape[0]], dtype=np.int64))
    T = np.asarray(Tcls, dtype=np.float64).reshape(-1, 4, 4)
    poses, sms = sv.solve_multistart(np.stack([simdata.pose7_from_T(t) for t in T]), options)
    for k in range(T.shape[0]):
        T[k] = simdata.T_from_pose7(poses[k])
    np.asarray(Tcls)[...] = T.reshape(np.asarray(Tcls).shape)
    costs = np.array([s.final_cost for s in sms])
    return int(np.argmin(costs)), costs, sms


def CamLaserCalClosedSolution(obs: ObsLike, Tlc: np.ndarray, solver: Optional[Solver] = None, verbose: bool = True):
    """Closed-form initialiser; overwrites Tlc (camera->laser) like LaseCamCalCeres.cpp:198-200."""
    return Session(obs, solver).CamLaserCalClosedSolution(Tlc, verbose)


def CamLaserCalibration(obs: ObsLike, Tcl: np.ndarray, use_linefitting_data: bool = True,
                        use_boundary_constraint: bool = False, options: Optional[Options] = None,
                        solver: Optional[Solver] = None, verbose: bool = True) -> CalibrationReport:
    """Nonlinear refinement; Tcl (laser->camera) is in/out like LaseCamCalCeres.cpp:215,:311-314."""
    return Session(obs, solver).CamLaserCalibration(Tcl, use_linefitting_data, use_boundary_constraint, options, verbose)


def LineFittingCeres(Points: np.ndarray, Line: np.ndarray, solver: Optional[Solver] = None,
                     options: Optional[Options] = None) -> None:
    """Robust 2-parameter line fit of one scan, `Line` (m0, m1 of m0 x + m1 y + 1 = 0) in/out —
    mirror of LineFittingCeres(Points, Line), src/LaseCamCalCeres.cpp:401-433."""
    P = np.asarray(Points, dtype=np.float64).reshape(-1, 3)
    sv = solver or _shared_solver()
    lines, _ = sv.line_fit_batched(P[:, :2], np.array([0, P.shape[0]], dtype=np.int64),
                                   np.asarray(Line, dtype=np.float64).reshape(1, 2), options, want_summaries=False)

    Line[...] = lines[0]


def points_on_fitted_lines(obs_set: ObservationSet, solver: Optional[Solver] = None,
                           line0=(0.0, 0.0)) -> ObservationSet:
    """The scan front-end step of main/calibr_offline.cpp:121-142 for all scans at once: fit a line
    to every scan's `points` (batched on the GPU) and replace `points_on_line` by the two points of
    the fitted line at the first / last scan point's abscissa (or ordinate for near-vertical lines)."""
    S = obs_set.n_poses
    sv = solver or _shared_solver()
    lines, _ = sv.line_fit_batched(obs_set.pts[:, :2], obs_set.pts_off, np.tile(np.asarray(line0, dtype=np.float64), (S, 1)),
                                   want_summaries=False)

    # the two end points per scan, all scans at once (the per-scan arithmetic of :126-136, element for element)
    lo, hi = obs_set.pts_off[:-1], obs_set.pts_off[1:]
    keep = hi - lo >= 2
    first = np.where(keep, lo, 0)
    last = np.where(keep, hi - 1, 0)
    P = obs_set.pts if obs_set.pts.shape[0] else np.zeros((1, 3))
    xs, ys, xe, ye = P[first, 0].copy(), P[first, 1].copy(), P[last, 0].copy(), P[last, 1].copy()
    m0, m1 = lines[:, 0], lines[:, 1]
    horiz = np.abs(xe - xs) > np.abs(ye - ys)
    with np.errstate(divide="ignore", invalid="ignore"):
        ys = np.where(horiz, -(xs * m0 + 1) / m1, ys)   # :126-131
        ye = np.where(horiz, -(xe * m0 + 1) / m1, ye)
        xs = np.where(horiz, xs, -(ys * m1 + 1) / m0)   # :132-136
        xe = np.where(horiz, xe, -(ye * m1 + 1) / m0)
    ptl = np.zeros((2 * S, 3))
    ptl[0::2, 0], ptl[0::2, 1], ptl[1::2, 0], ptl[1::2, 1] = xs, ys, xe, ye
    ptl_off = np.zeros(S + 1, dtype=np.int64)
    ptl_off[1:] = np.cumsum(np.where(keep, 2, 0))
    ptl = ptl[np.repeat(keep, 2)]
    return ObservationSet(obs_set.tag_q, obs_set.tag_t, obs_set.pts_off, obs_set.pts, ptl_off, np.ascontiguousarray(ptl))
