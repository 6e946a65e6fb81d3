"""Solver: thin object wrapper over one clc_handle (include/clc.h)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import Iteration, Options, Summary, TERMINATION, check, default_line_options, default_options, dptr, iptr


_Pose7 = C.c_double * 7


@dataclass
class SolveResult:
    pose: np.ndarray  # [tx,ty,tz,qx,qy,qz,qw]
    summary: Summary
    trace: List[Iteration]

    @property
    def termination(self) -> str:
        return TERMINATION.get(self.summary.termination, "?")


def flatten_observations(obs_set, use_linefitting_data: bool = True, use_boundary_constraint: bool = False) -> np.ndarray:
    """Residual-block construction of CamLaserCalibration (src/LaseCamCalCeres.cpp:222-295)
    -> records [N,8] = {n(3), d, p(3), scale}.  Host-side; does not need a GPU."""
    L = _capi.lib()
    n = C.c_int64()
    args = (C.c_int(obs_set.n_poses), dptr(np.ascontiguousarray(obs_set.tag_q, dtype=np.float64)),
            dptr(np.ascontiguousarray(obs_set.tag_t, dtype=np.float64)), iptr(obs_set.pts_off), dptr(obs_set.pts),
            iptr(obs_set.ptl_off), dptr(obs_set.ptl), C.c_int(int(use_linefitting_data)),
            C.c_int(int(use_boundary_constraint)))
    check(L.clc_flatten_observations(*args, None, C.byref(n)), "clc_flatten_observations")
    rec = np.empty((n.value, 8))
    check(L.clc_flatten_observations(*args, dptr(rec), C.byref(n)), "clc_flatten_observations")
    return rec


RESULT_RECORD = 12  # doubles per clc_result_record: pose[7], final_cost, initial_cost, iterations, termination, global index
COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C-ABI: called by ONE rank, distributed to the others out of band."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(_capi.lib().clc_comm_unique_id(buf), "clc_comm_unique_id")
    return buf.raw


class Comm:
    """RCCL communicator bound to one Solver (clc_comm): collectives run on the solver's stream.
    Creating it is a collective call — every rank of the job must do so with the same id."""

    def __init__(self, solver: "Solver", uid: Optional[bytes], rank: int, world: int):
        """uid None (hooks build only, tests): a communicator laid out as `rank` of `world` without RCCL behind it — its all-gather
        moves this rank's segment into place and nothing else (clc_debug_comm_create_layout)."""
        self._L = solver._L
        self._c = C.c_void_p()
        self._solver = solver  # keep the handle alive
        if uid is None:
            check(solver._hook("clc_debug_comm_create_layout")(C.byref(self._c), solver._h, C.c_int(rank), C.c_int(world)), "clc_debug_comm_create_layout")
        else:
            assert len(uid) == COMM_ID_BYTES
            check(self._L.clc_comm_create(C.byref(self._c), solver._h, C.c_char_p(uid), C.c_int(rank), C.c_int(world)),
                  "clc_comm_create")
        self.rank, self.world = rank, world

    @property
    def library(self) -> str:
        return self._L.clc_comm_library().decode()

    @property
    def rccl_ranks(self) -> int:
        """Size of the communicator as RCCL reports it (ncclCommCount)."""
        return int(self._L.clc_comm_world(self._c))

    def gather_results(self, first_global_index: int, cap_per_rank: int, copy: bool = True) -> np.ndarray:
        """ncclAllGather of the result records of the solver's last solve_batched -> [world*cap_per_rank, 12]
        (rank-major; padding records have global index -1).  copy=False returns a view of the communicator's
        pinned host buffer (valid until the next gather) instead of a fresh array."""
        n = self.world * cap_per_rank
        if copy:
            out = np.empty((n, RESULT_RECORD))
            check(self._L.clc_gather_results(self._c, C.c_int64(first_global_index), C.c_size_t(cap_per_rank), dptr(out)),
                  "clc_gather_results")
            return out
        check(self._L.clc_gather_results(self._c, C.c_int64(first_global_index), C.c_size_t(cap_per_rank), None),
              "clc_gather_results")
        return np.ctypeslib.as_array(self._L.clc_comm_records(self._c), shape=(n, RESULT_RECORD))

    def solve_gather(self, poses0: np.ndarray, first_global_index: int, cap_per_rank: int, options: Optional[Options] = None,
                     copy: bool = True):
        """clc_solve_batched_gather: the solver's uploaded shard solved by ONE launch whose epilogue writes the result records into
        the gather buffer, all-gather in place, one copy to the host -> (records [world*cap_per_rank, 12], BatchStats of the local
        shard).  poses0: [P_local, 7] start poses (None: the handle's pinned buffer, batched_buffers()[0], already filled).
        copy=False: `records` is a view of the communicator's pinned host buffer (valid until the next gather)."""
        n = self.world * cap_per_rank
        st = _capi.BatchStats()
        o = options or default_options()
        if poses0 is None:
            pp = C.POINTER(C.c_double)()
            if self._solver.num_problems > 0:
                check(self._L.clc_batched_host_buffers(self._solver._h, C.byref(pp), None), "clc_batched_host_buffers")
            p_arg = C.cast(pp, C.c_void_p)
        else:
            poses = np.ascontiguousarray(poses0, dtype=np.float64)
            assert poses.size == 7 * self._solver.num_problems, "one start pose per local problem"
            p_arg = C.cast(dptr(poses), C.c_void_p)
        out = np.empty((n, RESULT_RECORD)) if copy else None
        check(self._L.clc_solve_batched_gather(self._c, C.byref(o), p_arg, C.c_int64(first_global_index), C.c_size_t(cap_per_rank),
                                               dptr(out) if copy else None, C.byref(st)), "clc_solve_batched_gather")
        if not copy:
            out = np.ctypeslib.as_array(self._L.clc_comm_records(self._c), shape=(n, RESULT_RECORD))
        return out, st

    def set_root(self, root: int = -1):
        """clc_comm_set_root: -1 = every rank receives (and copies to its host) every record; r >= 0 = only rank r does (ncclGather to r),
        the other ranks keep their own segment only.  Every rank must choose the same root."""
        check(self._L.clc_comm_set_root(self._c, C.c_int(root)), "clc_comm_set_root")

    def info(self) -> "_capi.CommInfo":
        ci = _capi.CommInfo()
        check(self._L.clc_comm_get_info(self._c, C.byref(ci)), "clc_comm_get_info")
        return ci

    def _records_view(self, ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n, RESULT_RECORD))

    def solve_gather_pipelined(self, poses0: np.ndarray, first_global_index: int, cap_per_rank: int, options: Optional[Options] = None):
        """clc_solve_batched_gather_pipelined: enqueue THIS step, get the PREVIOUS step's (records view [world*cap_per_rank, 12], BatchStats)
        — (None, None) on the first call.  The view stays valid until the NEXT pipelined call / flush; flush() returns the last step's."""
        n = self.world * cap_per_rank
        st = _capi.BatchStats()
        o = options or default_options()
        if poses0 is None:
            pp = C.POINTER(C.c_double)()
            if self._solver.num_problems > 0:
                check(self._L.clc_batched_host_buffers(self._solver._h, C.byref(pp), None), "clc_batched_host_buffers")
            p_arg = C.cast(pp, C.c_void_p)
        else:
            poses = np.ascontiguousarray(poses0, dtype=np.float64)
            assert poses.size == 7 * self._solver.num_problems, "one start pose per local problem"
            p_arg = C.cast(dptr(poses), C.c_void_p)
        prev = C.c_void_p()
        check(self._L.clc_solve_batched_gather_pipelined(self._c, C.byref(o), p_arg, C.c_int64(first_global_index), C.c_size_t(cap_per_rank),
                                                         C.byref(prev), C.byref(st)), "clc_solve_batched_gather_pipelined")
        if not prev.value:
            return None, None
        return self._records_view(prev, n), st

    def flush(self, cap_per_rank: int):
        """clc_gather_flush: complete the step in flight -> (records view, BatchStats), or (None, None) when none is."""
        st = _capi.BatchStats()
        rec = C.c_void_p()
        check(self._L.clc_gather_flush(self._c, C.byref(rec), C.byref(st)), "clc_gather_flush")
        if not rec.value:
            return None, None
        return self._records_view(rec, self.world * cap_per_rank), st

    def close(self):
        if getattr(self, "_c", None) is not None and self._c:
            self._L.clc_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Solver:
    """One solver context on one GPU (one HIP stream).  Not thread-safe per instance.
    library: None = the build the package runs on (the product library unless CLC_LIBRARY names another), "hooks" = the
    -DCLC_TEST_HOOKS build (the debug_* / time_* methods other than the layout reports need it), or a path."""

    def __init__(self, device: int = 0, library: Optional[str] = None):
        self._L = _capi.lib() if library is None else (_capi.hooks_lib() if library == "hooks" else _capi.load(library))
        self._h = C.c_void_p()
        check(self._L.clc_create(C.byref(self._h), C.c_int(device)), "clc_create")
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.clc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- configuration ----
    def set_stream(self, hip_stream: Optional[int]):
        check(self._L.clc_set_stream(self._h, C.c_void_p(hip_stream or 0)), "clc_set_stream")

    def set_auto_paths(self, disable_mask: int = 0):
        """clc_set_auto_paths: 1 = no cooperative one-launch solve, 2 = no single-workgroup on-chip solve, 8 = (at upload) not the
        32-workgroup one-hop form; 0 = library default.  Every bit disables a path."""
        check(self._L.clc_set_auto_paths(self._h, C.c_int(disable_mask)), "clc_set_auto_paths")

    def set_small_on_coop(self, enable: bool = True):
        """clc_set_small_on_coop (at upload): problems one workgroup holds also get the cooperative layout and run on 32 workgroups first."""
        check(self._L.clc_set_small_on_coop(self._h, C.c_int(int(enable))), "clc_set_small_on_coop")

    def debug_fast_small(self, enable: Optional[bool] = None) -> int:
        """Test hook: switch the host-planned upload of small problems on / off (None: leave it) -> uploads that took it so far."""
        n = C.c_longlong()
        check(self._hook("clc_debug_fast_small")(self._h, C.c_int(-1 if enable is None else int(enable)), C.byref(n)), "clc_debug_fast_small")
        return n.value

    def debug_single_controller(self, cooperative_kernels: bool):
        """Test hook: the single-workgroup solve runs the cooperative kernel's register-state LM controller instead of its own."""
        check(self._hook("clc_debug_single_controller")(self._h, C.c_int(int(cooperative_kernels))), "clc_debug_single_controller")

    def set_launch(self, grid_blocks: int = 0, flags: int = 0):
        check(self._L.clc_set_launch(self._h, C.c_int(grid_blocks), C.c_int(flags)), "clc_set_launch")

    def device_info(self) -> Tuple[str, int]:
        buf = C.create_string_buffer(256)
        n = C.c_int()
        check(self._L.clc_device_info(self._h, buf, C.c_int(256), C.byref(n)), "clc_device_info")
        return buf.value.decode(), n.value

    # ---- data ----
    def upload(self, records: np.ndarray):
        records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 8)
        check(self._L.clc_upload(self._h, dptr(records), C.c_size_t(records.shape[0])), "clc_upload")

    def upload_device(self, device_ptr: int, n: int):
        """records already in HBM as an [n,8] float64 AoS array (e.g. torch tensor .data_ptr())."""
        check(self._L.clc_upload_device(self._h, C.c_void_p(device_ptr), C.c_size_t(n)), "clc_upload_device")

    @property
    def num_observations(self) -> int:
        return int(self._L.clc_num_observations(self._h))

    # ---- resident scans: problem assembly on the device ----
    def store_observations(self, obs_set):
        """Upload the pose-major form of std::vector<Oberserve> (tag poses + scan points, 24 B per point) once; it stays
        resident for any number of select_observations calls."""
        S = obs_set
        check(self._L.clc_store_observations(self._h, C.c_int(S.n_poses), dptr(np.ascontiguousarray(S.tag_q, dtype=np.float64)),
                                             dptr(np.ascontiguousarray(S.tag_t, dtype=np.float64)), iptr(S.pts_off), dptr(S.pts),
                                             iptr(S.ptl_off), dptr(S.ptl)), "clc_store_observations")

    @property
    def store_generation(self) -> int:
        """Stamp of the scans currently stored on this handle (bumped by every store_observations)."""
        return int(self._L.clc_store_generation(self._h))

    def select_observations(self, use_linefitting_data: bool = True, use_boundary_constraint: bool = False) -> int:
        """Build the residual blocks of the selection on the device (src/LaseCamCalCeres.cpp:222-295) and make them the
        handle's observation array (as upload() would) -> number of records."""
        n = C.c_int64()
        check(self._L.clc_select_observations(self._h, C.c_int(int(use_linefitting_data)), C.c_int(int(use_boundary_constraint)),
                                              C.byref(n)), "clc_select_observations")
        return n.value

    def debug_flatten_device(self, use_linefitting_data: bool = True, use_boundary_constraint: bool = False) -> np.ndarray:
        """The device-built records of a selection, copied back (test hook)."""
        n = C.c_int64()
        check(self._hook("clc_debug_flatten_device")(self._h, C.c_int(int(use_linefitting_data)), C.c_int(int(use_boundary_constraint)),
                                               None, C.c_int64(0), C.byref(n)), "clc_debug_flatten_device")
        rec = np.empty((n.value, 8))
        if n.value:
            check(self._hook("clc_debug_flatten_device")(self._h, C.c_int(int(use_linefitting_data)), C.c_int(int(use_boundary_constraint)),
                                                   dptr(rec), C.c_int64(n.value), C.byref(n)), "clc_debug_flatten_device")
        return rec

    # ---- plug-in level ----
    def factor_evaluate(self, pose: np.ndarray, want_jacobian: bool = True):
        n = self.num_observations
        r = np.empty(n)
        j = np.empty((n, 7)) if want_jacobian else None
        check(self._L.clc_factor_evaluate(self._h, dptr(np.ascontiguousarray(pose, dtype=np.float64)), dptr(r), dptr(j)),
              "clc_factor_evaluate")
        return r, j

    def pose_plus(self, x: np.ndarray, delta: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 7)
        delta = np.ascontiguousarray(delta, dtype=np.float64).reshape(-1, 6)
        out = np.empty_like(x)
        check(self._L.clc_pose_plus(self._h, dptr(x), dptr(delta), dptr(out), C.c_size_t(x.shape[0])), "clc_pose_plus")
        return out

    # ---- evaluation / solve ----
    def eval(self, pose: np.ndarray, with_loss: bool = True, loss_scale_factor: float = 0.05, want_jacobian: bool = True):
        """-> (cost, g[6], H[21])  (g, H None for a cost-only pass)."""
        cost = C.c_double()
        g = np.empty(6) if want_jacobian else None
        H = np.empty(21) if want_jacobian else None
        check(self._L.clc_eval(self._h, dptr(np.ascontiguousarray(pose, dtype=np.float64)), C.c_int(int(with_loss)),
                               C.c_double(loss_scale_factor), C.byref(cost), dptr(g), dptr(H)), "clc_eval")
        return cost.value, g, H

    def solve(self, pose0: np.ndarray, options: Optional[Options] = None, trace_cap: int = 256) -> SolveResult:
        pose = np.array(pose0, dtype=np.float64).reshape(7)  # (a copy: in/out)
        s = Summary()
        tr = (Iteration * trace_cap)() if trace_cap > 0 else None
        o = options or default_options()
        # (a ctypes view of the array's buffer: numpy's .ctypes.data_as() costs 2.7 us per call, 3 % of a C2 solve)
        check(self._L.clc_solve(self._h, C.byref(o), _Pose7.from_buffer(pose), C.byref(s), tr, trace_cap), "clc_solve")
        n = max(0, min(trace_cap, s.num_iterations + 1))
        return SolveResult(pose, s, [tr[i] for i in range(n)])

    def information(self, pose: np.ndarray):
        """Analysis pass (src/LaseCamCalCeres.cpp:316-381) -> (H[6,6], b[6], chi2, sv[6], V[6,6], n_null)."""
        H = np.empty(36); b = np.empty(6); chi = C.c_double(); sv = np.empty(6); V = np.empty(36); nn = C.c_int()
        check(self._L.clc_information(self._h, dptr(np.ascontiguousarray(pose, dtype=np.float64)), dptr(H), dptr(b),
                                      C.byref(chi), dptr(sv), dptr(V), C.byref(nn)), "clc_information")
        return H.reshape(6, 6), b, chi.value, sv, V.reshape(6, 6), nn.value

    def closed_form(self):
        """CamLaserCalClosedSolution on the uploaded records -> (Tlc[4,4], unobservable, sv9)."""
        T = np.empty(16); un = C.c_int(); sv = np.empty(9)
        check(self._L.clc_closed_form(self._h, dptr(T), C.byref(un), dptr(sv)), "clc_closed_form")
        return T.reshape(4, 4), bool(un.value), sv

    # ---- batched ----
    def upload_batched(self, records: np.ndarray, offsets: np.ndarray):
        records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        check(self._L.clc_upload_batched(self._h, dptr(records), iptr(offsets), C.c_size_t(len(offsets) - 1)),
              "clc_upload_batched")

    def upload_batched_device(self, device_ptr: int, offsets: np.ndarray):
        """records already in HBM as an [N,8] float64 AoS array (ready on the solver's stream); offsets on the host."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        check(self._L.clc_upload_batched_device(self._h, C.c_void_p(device_ptr), iptr(offsets), C.c_size_t(len(offsets) - 1)),
              "clc_upload_batched_device")

    @property
    def num_problems(self) -> int:
        return int(self._L.clc_num_problems(self._h))

    def batched_buffers(self):
        """The handle's own pinned host arrays of the uploaded batch -> (poses [P,7] float64 view, summaries: ctypes array
        view of P Summary records).  Fill `poses` with the start poses and call solve_batched_inplace(): no staging copies."""
        P = self.num_problems
        pp, ps = C.POINTER(C.c_double)(), C.POINTER(Summary)()
        check(self._L.clc_batched_host_buffers(self._h, C.byref(pp), C.byref(ps)), "clc_batched_host_buffers")
        poses = np.ctypeslib.as_array(pp, shape=(P, 7))
        sms = C.cast(ps, C.POINTER(Summary * P)).contents
        return poses, sms

    def solve_batched_inplace(self, options: Optional[Options] = None):
        """clc_solve_batched on the handle's own buffers (batched_buffers()): start poses in, results out, in place."""
        P = self.num_problems
        pp, ps = C.POINTER(C.c_double)(), C.POINTER(Summary)()
        check(self._L.clc_batched_host_buffers(self._h, C.byref(pp), C.byref(ps)), "clc_batched_host_buffers")
        o = options or default_options()
        check(self._L.clc_solve_batched(self._h, C.byref(o), pp, ps), "clc_solve_batched")
        return np.ctypeslib.as_array(pp, shape=(P, 7)), C.cast(ps, C.POINTER(Summary * P)).contents

    def solve_batched(self, poses0: np.ndarray, options: Optional[Options] = None):
        """-> (poses[P,7], summaries[P])"""
        P = self.num_problems
        poses = np.ascontiguousarray(np.array(poses0, dtype=np.float64).reshape(P, 7)).copy()
        sm = (Summary * P)()
        o = options or default_options()
        check(self._L.clc_solve_batched(self._h, C.byref(o), dptr(poses), sm), "clc_solve_batched")
        return poses, sm

    def solve_multistart(self, poses0: np.ndarray, options: Optional[Options] = None):
        """clc_solve_multistart: S independent LM solves from S start poses [S, 7] on the ONE problem uploaded as a batch of one
        (upload_batched(records, [0, n])) — one copy of the observations on the device -> (poses [S, 7], summaries [S])."""
        poses = np.ascontiguousarray(np.array(poses0, dtype=np.float64).reshape(-1, 7)).copy()
        S = poses.shape[0]
        sm = (Summary * S)()
        o = options or default_options()
        check(self._L.clc_solve_multistart(self._h, C.byref(o), C.c_size_t(S), dptr(poses), sm), "clc_solve_multistart")
        return poses, sm

    # ---- scan line fitting ----
    def line_fit_batched(self, xy: np.ndarray, offsets: np.ndarray, lines0: np.ndarray,
                         options: Optional[Options] = None, want_summaries: bool = True):
        """LineFittingCeres for many scans: xy [M,2], CSR offsets [S+1], lines0 [S,2] -> (lines [S,2], summaries)."""
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        S = len(offsets) - 1
        lines = np.ascontiguousarray(np.array(lines0, dtype=np.float64).reshape(S, 2)).copy()
        sm = (Summary * S)() if want_summaries and S > 0 else None
        o = options or default_line_options()
        check(self._L.clc_line_fit_batched(self._h, C.byref(o), dptr(xy), iptr(offsets), C.c_size_t(S), dptr(lines), sm),
              "clc_line_fit_batched")
        return lines, sm

    def line_fit_batched_device(self, xy_ptr: int, offsets_ptr: int, n_scans: int, lines_ptr: int, summaries_ptr: int = 0,
                                options: Optional[Options] = None):
        """LineFittingCeres on device-resident arrays (data_ptr()s; ready on the solver's stream)."""
        o = options or default_line_options()
        check(self._L.clc_line_fit_batched_device(self._h, C.byref(o), C.c_void_p(xy_ptr), C.c_void_p(offsets_ptr), C.c_size_t(n_scans),
                                                  C.c_void_p(lines_ptr), C.c_void_p(summaries_ptr or 0)), "clc_line_fit_batched_device")

    def scan_to_points_device(self, ranges_ptr: int, offsets_ptr: int, n_scans: int, n_rays: int, angle_min_ptr: int,
                              angle_increment_ptr: int, range_min_ptr: int, points_ptr: int):
        """TranScanToPoints on device-resident arrays (data_ptr()s; ready on the solver's stream)."""
        check(self._L.clc_scan_to_points_device(self._h, C.c_void_p(ranges_ptr), C.c_void_p(offsets_ptr), C.c_size_t(n_scans),
                                                C.c_size_t(n_rays), C.c_void_p(angle_min_ptr), C.c_void_p(angle_increment_ptr),
                                                C.c_void_p(range_min_ptr), C.c_void_p(points_ptr)), "clc_scan_to_points_device")

    def scan_to_points(self, ranges: np.ndarray, offsets: np.ndarray, angle_min, angle_increment, range_min) -> np.ndarray:
        """TranScanToPoints (src/utilities.cpp:181-215) for many scans: ranges float32 [M], CSR offsets [S+1],
        per-scan angle_min / angle_increment / range_min -> points [M,3]."""
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        S = len(offsets) - 1
        f = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float32), (S,)))
        am, ai, rm = f(angle_min), f(angle_increment), f(range_min)
        pts = np.empty((r.shape[0], 3))
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        check(self._L.clc_scan_to_points(self._h, fp(r), iptr(offsets), C.c_size_t(S), fp(am), fp(ai), fp(rm), dptr(pts)),
              "clc_scan_to_points")
        return pts

    # ---- test / profiling hooks ----
    def debug_math(self, op: int, x: np.ndarray) -> np.ndarray:
        """Device branch of a scalar helper, element-wise (hooks build): 0 rsqrt_pos, 1 rcp_pos, 2 rcp_pos_safe, 3 sqrt_pos, 4 rcp_ge1,
        5 rcp_ge1_weight, 6 log via frexp_pos + log_mant_exp."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        out = np.empty_like(x)
        f = self._hook("clc_debug_math")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong]
        check(f(self._h, op, x.ctypes.data, out.ctypes.data, x.shape[0]), "clc_debug_math")
        return out

    def debug_wave_reduce(self, lanes: np.ndarray, reduce_mode: int) -> np.ndarray:
        lanes = np.ascontiguousarray(lanes, dtype=np.float64).reshape(64, 28)
        out = np.empty(28)
        check(self._hook("clc_debug_wave_reduce")(self._h, dptr(lanes), dptr(out), C.c_int(reduce_mode)), "clc_debug_wave_reduce")
        return out

    def _hook(self, name: str):
        f = getattr(self._L, name, None)
        if f is None:
            raise RuntimeError(f"{name} is a test hook: not in the product library — create the solver with library=\"hooks\" "
                               "(csrc/libclc_hip_hooks.so) or run with CLC_LIBRARY set to a hooks build")
        return f

    def path_info(self) -> "_capi.PathInfo":
        """clc_get_path_info: which layouts the last uploads built and how the cooperative path is doing."""
        pi = _capi.PathInfo()
        check(self._L.clc_get_path_info(self._h, C.byref(pi)), "clc_get_path_info")
        return pi

    def debug_rows(self):
        """Row layout report -> (rows_ok, n_rows, batched_rows_ok, batched_n_rows)."""
        pi = self.path_info()
        return bool(pi.rows_layout), pi.n_rows, bool(pi.batched_rows_layout), pi.batched_n_rows

    def rows_carry_z(self):
        """(single-problem rows, batched rows) carry z: some uploaded record has p.z != 0, so the rows are the 24-byte form
        (64 z after the 64 (x, y) pairs of every row, 14 moments per scan) instead of the 16-byte one."""
        pi = self.path_info()
        return pi.rows_layout == 2, pi.batched_rows_layout == 2

    def debug_resident(self):
        """Resident ("lane") layout report of the uploaded batch -> (built, lanes per problem, largest points per lane,
        j-rows in all); built = the batched solver keeps every problem on chip for its whole solve (clc_resident.hpp)."""
        pi = self.path_info()
        return bool(pi.batched_resident), pi.batched_lanes, pi.batched_points_per_lane, pi.batched_lane_rows

    def debug_resident_single(self):
        """Lane layout of the single-problem array -> (built, lanes, points per lane); built = clc_solve with the default
        flags runs the whole LM solve in ONE single-workgroup launch (problems of at most 512 x 22 points)."""
        pi = self.path_info()
        return bool(pi.single_resident), pi.single_lanes, pi.single_points_per_lane

    def debug_coop(self):
        """Cooperative whole-GPU solve of the single-problem array (csrc/clc_coop.hpp) -> (layout built, largest points per
        lane, solves run on it, launches that timed out, resting on this handle)."""
        pi = self.path_info()
        return bool(pi.coop_resident), pi.coop_points_per_lane, pi.coop_solves, pi.coop_timeouts, bool(pi.coop_resting)

    def debug_coop_control(self, drop_next: int = 0, reenable: bool = False):
        """Test hook: launch the next cooperative solve `drop_next` workgroups short (it must time out and fall back); clear the
        disabled state."""
        check(self._hook("clc_debug_coop_control")(self._h, C.c_int(drop_next), C.c_int(int(reenable))), "clc_debug_coop_control")

    def debug_coop_set_tag(self, tag: int):
        """Test hook: first pass tag of the next cooperative solve (32-bit; exercises the wrap)."""
        check(self._hook("clc_debug_coop_set_tag")(self._h, C.c_uint(tag)), "clc_debug_coop_set_tag")

    def debug_wave_split(self, grid: int):
        """Wave split table of the row layout for `grid` workgroups -> (split[grid * 8 + 1], first[n_rows])."""
        n_rows = self.debug_rows()[1]
        split = np.zeros(grid * 8 + 1, dtype=np.int32)
        first = np.zeros(max(n_rows, 1), dtype=np.int32)
        self._hook("clc_debug_wave_split").argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        check(self._hook("clc_debug_wave_split")(self._h, C.c_int(grid), split.ctypes.data, first.ctypes.data), "clc_debug_wave_split")
        return split, first[:n_rows]

    def time_steps(self, pose: np.ndarray, first: int, last: int) -> Tuple[float, int]:
        """Mean period [ms] of the step_kernel launches first..last of one default solve (HIP events on the handle's
        stream right before launch `first` and right after launch `last`) and the solve's number of passes."""
        ms = C.c_double()
        n = C.c_int()
        check(self._hook("clc_time_steps")(self._h, dptr(np.ascontiguousarray(pose, dtype=np.float64)), C.c_int(first), C.c_int(last),
                                     C.byref(ms), C.byref(n)), "clc_time_steps")
        return ms.value, n.value

    def time_batched_eval(self, poses: np.ndarray, reps: int = 20) -> float:
        """Mean duration [ms] of `reps` back-to-back batched_eval_kernel launches over all uploaded problems at `poses`."""
        ms = C.c_double()
        P = self.num_problems
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(P, 7))
        check(self._hook("clc_time_batched_eval")(self._h, dptr(poses), C.c_int(reps), C.byref(ms)), "clc_time_batched_eval")
        return ms.value

    def time_eval(self, pose: np.ndarray, reps: int = 20, with_loss: bool = True, loss_scale_factor: float = 0.05,
                  with_jacobian: bool = True) -> float:
        """Mean duration [ms] of `reps` back-to-back evaluation-kernel launches (HIP events on the
        handle's stream)."""
        ms = C.c_double()
        check(self._hook("clc_time_eval")(self._h, dptr(np.ascontiguousarray(pose, dtype=np.float64)), C.c_int(int(with_loss)),
                                    C.c_double(loss_scale_factor), C.c_int(int(with_jacobian)), C.c_int(reps), C.byref(ms)),
              "clc_time_eval")
        return ms.value
