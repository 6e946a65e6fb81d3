"""camlasercalibratool_amd — MI355X-native (gfx950) solver for CamLaserCalibraTool's
point-to-plane camera/2-D-lidar extrinsic least-squares path.

The package is a thin host layer over a C-ABI shared library of hand-written HIP kernels
(csrc/, include/clc.h).  It mirrors the reference's call surface
(CamLaserCalibration / CamLaserCalClosedSolution / Oberserve) and has no CPU fallback."""
from .simdata import GenerateSimData, Oberserve, ObservationSet  # noqa: F401
from ._capi import ClcError, Options, Summary, Iteration, TERMINATION, default_line_options, default_options  # noqa: F401
from .solver import Solver, SolveResult, flatten_observations  # noqa: F401
from .calib import (CamLaserCalibration, CamLaserCalClosedSolution, CamLaserCalibrationFromStarts, CalibrationReport,  # noqa: F401
                    LineFittingCeres, Session, points_on_fitted_lines)

__all__ = [
    "CamLaserCalibration", "CamLaserCalClosedSolution", "CamLaserCalibrationFromStarts", "LineFittingCeres", "Oberserve", "ObservationSet", "GenerateSimData",
    "Session", "Solver", "SolveResult", "Options", "default_options", "flatten_observations", "ClcError",
]
