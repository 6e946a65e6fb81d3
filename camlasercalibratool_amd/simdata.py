"""Seeded, ROS-free synthetic input generators for the point-to-plane extrinsic path.

`GenerateSimData` restates the reference's only ground-truth fixture,
main/calibr_simulation.cpp:10-108 (which is seeded from std::random_device, i.e. not
reproducible), with an explicit seed and the same distributions.  The other generators
produce the configurations named in BASELINE.json / SURVEY.md §8(d):

  C1  sim_default(seed)              50 poses x 180 rays at 1 deg (literal restatement)
  C2  sim_fixed_count(...)           P poses x exactly K rays in the valid interval (10^6 obs)
  C3  sim_batch(...)                 many independent problems, each its own ground-truth Tlc
  C5  sim_board_edges(...)           scan clipped to the 0.5 m board so the board-edge
                                     ("boundary") residuals of LaseCamCalCeres.cpp:258-294 hold

All data are pose-major CSR (`ObservationSet`), the flattened form of
std::vector<Oberserve> (include/LaseCamCalCeres.h:11-24).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

# Ground truth of the reference simulation, main/calibr_simulation.cpp:15-20.
GT_RLC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
GT_TLC = np.array([0.1, 0.2, 0.3])


# ----------------------------------------------------------------------------------
# containers
# ----------------------------------------------------------------------------------
@dataclass
class Oberserve:
    """Mirror of `struct Oberserve` (sic), include/LaseCamCalCeres.h:11-24.

    tagPose_Qca is stored (w, x, y, z) — the Eigen::Quaterniond constructor order used
    by the reference (`Eigen::Quaterniond(1,0,0,0)`, LaseCamCalCeres.h:16).
    """

    tagPose_Qca: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0, 0.0]))
    tagPose_tca: np.ndarray = field(default_factory=lambda: np.zeros(3))
    points: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    points_on_line: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))


@dataclass
class ObservationSet:
    """Pose-major CSR form of std::vector<Oberserve> (what the C-ABI flatten takes)."""

    tag_q: np.ndarray  # [P,4] (w,x,y,z)
    tag_t: np.ndarray  # [P,3]
    pts_off: np.ndarray  # [P+1] int64
    pts: np.ndarray  # [M,3]
    ptl_off: np.ndarray  # [P+1] int64  (points_on_line)
    ptl: np.ndarray  # [ML,3]

    @property
    def n_poses(self) -> int:
        return int(self.tag_q.shape[0])

    @staticmethod
    def from_list(obs: Sequence[Oberserve]) -> "ObservationSet":
        P = len(obs)
        tag_q = np.zeros((P, 4))
        tag_t = np.zeros((P, 3))
        pts_off = np.zeros(P + 1, dtype=np.int64)
        ptl_off = np.zeros(P + 1, dtype=np.int64)
        for i, o in enumerate(obs):
            tag_q[i] = o.tagPose_Qca
            tag_t[i] = o.tagPose_tca
            pts_off[i + 1] = pts_off[i] + len(o.points)
            ptl_off[i + 1] = ptl_off[i] + len(o.points_on_line)
        pts = np.concatenate([np.asarray(o.points, dtype=np.float64).reshape(-1, 3) for o in obs]) if P else np.zeros((0, 3))
        ptl = np.concatenate([np.asarray(o.points_on_line, dtype=np.float64).reshape(-1, 3) for o in obs]) if P else np.zeros((0, 3))
        return ObservationSet(tag_q, tag_t, pts_off, np.ascontiguousarray(pts), ptl_off, np.ascontiguousarray(ptl))

    def to_list(self) -> List[Oberserve]:
        out = []
        for i in range(self.n_poses):
            out.append(
                Oberserve(
                    self.tag_q[i].copy(),
                    self.tag_t[i].copy(),
                    self.pts[self.pts_off[i] : self.pts_off[i + 1]].copy(),
                    self.ptl[self.ptl_off[i] : self.ptl_off[i + 1]].copy(),
                )
            )
        return out


# ----------------------------------------------------------------------------------
# small SO(3) helpers (Eigen conventions)
# ----------------------------------------------------------------------------------
def rot_zyx(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Rz(a) * Ry(b) * Rx(c), batched — calibr_simulation.cpp:42-44."""
    a, b, c = np.atleast_1d(a), np.atleast_1d(b), np.atleast_1d(c)
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    R = np.empty(a.shape + (3, 3))
    R[..., 0, 0] = ca * cb
    R[..., 0, 1] = ca * sb * sc - sa * cc
    R[..., 0, 2] = ca * sb * cc + sa * sc
    R[..., 1, 0] = sa * cb
    R[..., 1, 1] = sa * sb * sc + ca * cc
    R[..., 1, 2] = sa * sb * cc - ca * sc
    R[..., 2, 0] = -sb
    R[..., 2, 1] = cb * sc
    R[..., 2, 2] = cb * cc
    return R


def rot_to_quat_wxyz(R: np.ndarray) -> np.ndarray:
    """Eigen::Quaterniond(Matrix3d), returned (w,x,y,z); batched."""
    R = np.asarray(R, dtype=np.float64)
    single = R.ndim == 2
    R = R.reshape(-1, 3, 3)
    q = np.empty((R.shape[0], 4))
    for n in range(R.shape[0]):
        m = R[n]
        t = m[0, 0] + m[1, 1] + m[2, 2]
        if t > 0.0:
            t = np.sqrt(t + 1.0)
            w = 0.5 * t
            t = 0.5 / t
            q[n] = (w, (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t)
        else:
            i = 0
            if m[1, 1] > m[0, 0]:
                i = 1
            if m[2, 2] > m[i, i]:
                i = 2
            j = (i + 1) % 3
            k = (j + 1) % 3
            t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
            v = [0.0, 0.0, 0.0]
            v[i] = 0.5 * t
            t = 0.5 / t
            w = (m[k, j] - m[j, k]) * t
            v[j] = (m[j, i] + m[i, j]) * t
            v[k] = (m[k, i] + m[i, k]) * t
            q[n] = (w, v[0], v[1], v[2])
    return q[0] if single else q


def quat_wxyz_to_rot(q: np.ndarray) -> np.ndarray:
    """Eigen::Quaterniond::toRotationMatrix() for (w,x,y,z); batched, no normalisation."""
    q = np.asarray(q, dtype=np.float64)
    single = q.ndim == 1
    q = q.reshape(-1, 4)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - (tyy + tzz)
    R[:, 0, 1] = txy - twz
    R[:, 0, 2] = txz + twy
    R[:, 1, 0] = txy + twz
    R[:, 1, 1] = 1 - (txx + tzz)
    R[:, 1, 2] = tyz - twx
    R[:, 2, 0] = txz - twy
    R[:, 2, 1] = tyz + twx
    R[:, 2, 2] = 1 - (txx + tyy)
    return R[0] if single else R


def tlc_to_tcl(Rlc: np.ndarray, tlc: np.ndarray) -> np.ndarray:
    """4x4 Tcl = inverse of Tlc (calibr_simulation.cpp:129)."""
    T = np.eye(4)
    T[:3, :3] = Rlc.T
    T[:3, 3] = -Rlc.T @ tlc
    return T


def pose7_from_T(T: np.ndarray) -> np.ndarray:
    """pose = [t, qx,qy,qz,qw] from a 4x4 (LaseCamCalCeres.cpp:215-219)."""
    q = rot_to_quat_wxyz(T[:3, :3])
    return np.array([T[0, 3], T[1, 3], T[2, 3], q[1], q[2], q[3], q[0]])


def T_from_pose7(p: np.ndarray) -> np.ndarray:
    """4x4 from pose (LaseCamCalCeres.cpp:311-314)."""
    T = np.eye(4)
    T[:3, :3] = quat_wxyz_to_rot(np.array([p[6], p[3], p[4], p[5]]))
    T[:3, 3] = p[:3]
    return T


# ----------------------------------------------------------------------------------
# generators
# ----------------------------------------------------------------------------------
def _draw_tag_poses(rng: np.random.Generator, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """Rca = Rz(U)Ry(U)Rx(U), U~U(-pi/6,pi/6); tca=(U(-3,3),U(-3,3),U(1,5)).
    Draw order per pose follows calibr_simulation.cpp:42-44,58 (yaw, pitch, roll, x, y, z)."""
    u = rng.random((n, 6))
    ang = (u[:, :3] * 2.0 - 1.0) * (np.pi / 6.0)
    Rca = rot_zyx(ang[:, 0], ang[:, 1], ang[:, 2])
    tca = np.stack([u[:, 3] * 6.0 - 3.0, u[:, 4] * 6.0 - 3.0, 1.0 + u[:, 5] * 4.0], axis=1)
    return Rca, tca


def _plane_in_laser(Rlc, tlc, Rca, tca):
    """Tag plane (z_tag = 0) in the laser frame: (Tla^-1)^T (0,0,1,0), :62-73."""
    Rla = Rlc @ Rca
    tla = (Rlc @ tca[..., None])[..., 0] + tlc
    n = Rla[..., :, 2]
    d = -np.sum(n * tla, axis=-1)
    return n, d


def points_from_tag_poses(Rca: np.ndarray, tca: np.ndarray, n_rays: int = 180, noise_sigma: float = 0.0,
                          rng: Optional[np.random.Generator] = None, Rlc: np.ndarray = GT_RLC,
                          tlc: np.ndarray = GT_TLC) -> ObservationSet:
    """The deterministic part of GenerateSimData (main/calibr_simulation.cpp:60-104): tag plane in the laser frame,
    n_rays rays at theta = -pi/2 + j*pi/180 (:81), depth = -d/(ray.n) (:83), keep depth >= 0 and |x|<5 and |y|<5
    (:85-94); points_on_line = points (:101-102).  A pose may legitimately contribute zero points."""
    n_poses = Rca.shape[0]
    n, d = _plane_in_laser(Rlc, tlc, Rca, tca)
    theta = -np.pi / 2 + np.arange(n_rays) * (np.pi / 180.0)
    ray = np.stack([np.cos(theta), np.sin(theta), np.zeros_like(theta)], axis=1)  # [K,3]
    with np.errstate(divide="ignore", invalid="ignore"):
        depth = -d[:, None] / (n @ ray.T)  # [P,K]
    if noise_sigma > 0.0:
        depth = depth + rng.normal(0.0, noise_sigma, size=depth.shape)
    p = depth[:, :, None] * ray[None, :, :]
    valid = (~np.isnan(depth)) & (depth >= 0) & (np.abs(p[:, :, 0]) < 5) & (np.abs(p[:, :, 1]) < 5)
    counts = valid.sum(axis=1)
    off = np.zeros(n_poses + 1, dtype=np.int64)
    off[1:] = np.cumsum(counts)
    pts = np.ascontiguousarray(p[valid])  # row-major boolean mask keeps pose order
    tag_q = rot_to_quat_wxyz(Rca)
    return ObservationSet(tag_q, np.array(tca, dtype=np.float64).copy(), off, pts, off.copy(), pts.copy())


def GenerateSimData(seed: int, n_poses: int = 50, n_rays: int = 180, noise_sigma: float = 0.0,
                    Rlc: np.ndarray = GT_RLC, tlc: np.ndarray = GT_TLC) -> ObservationSet:
    """C1 — restatement of GenerateSimData, main/calibr_simulation.cpp:10-108: tag poses drawn from the reference's
    distributions (with numpy's generator — the reference seeds its engine from std::random_device, so there is no
    stream to reproduce), then `points_from_tag_poses`, which tests/test_ref_pin.py checks against the reference's own
    function on the reference's own poses.  noise_sigma adds N(0,sigma) range noise (the reference is noise-free)."""
    rng = np.random.default_rng(seed)
    Rca, tca = _draw_tag_poses(rng, n_poses)
    return points_from_tag_poses(Rca, tca, n_rays, noise_sigma, rng, Rlc, tlc)


def _valid_interval(n, d):
    """theta-interval of rays from the origin that hit the line n.x*x+n.y*y+d=0 inside
    x>=0, |x|<5, |y|<5 with depth>=0.  Returns (lo, hi, ok)."""
    # clip the line against the box [0,5]x[-5,5]
    nx, ny = n[:, 0], n[:, 1]
    P = n.shape[0]
    lo = np.full(P, np.nan)
    hi = np.full(P, np.nan)
    # parametrise line: point p0 = -d*n_xy/|n_xy|^2, direction t = (-ny, nx)
    nn = nx * nx + ny * ny
    ok = nn > 1e-12
    p0x = np.where(ok, -d * nx / np.where(ok, nn, 1), 0.0)
    p0y = np.where(ok, -d * ny / np.where(ok, nn, 1), 0.0)
    tx, ty = -ny, nx
    smin = np.full(P, -np.inf)
    smax = np.full(P, np.inf)

    def clip(num, den, smin, smax, ok):
        # constraint: num + s*den >= 0
        with np.errstate(divide="ignore", invalid="ignore"):
            s0 = -num / den
        pos = den > 1e-15
        neg = den < -1e-15
        zer = ~(pos | neg)
        smin = np.where(pos, np.maximum(smin, s0), smin)
        smax = np.where(neg, np.minimum(smax, s0), smax)
        ok = ok & ~(zer & (num < 0))
        return smin, smax, ok

    eps = 1e-6
    smin, smax, ok = clip(p0x - eps, tx, smin, smax, ok)            # x >= eps
    smin, smax, ok = clip((5 - eps) - p0x, -tx, smin, smax, ok)     # x <= 5-eps
    smin, smax, ok = clip(p0y + (5 - eps), ty, smin, smax, ok)      # y >= -5+eps
    smin, smax, ok = clip((5 - eps) - p0y, -ty, smin, smax, ok)     # y <= 5-eps
    ok = ok & (smax > smin) & np.isfinite(smin) & np.isfinite(smax)
    ax, ay = p0x + smin * tx, p0y + smin * ty
    bx, by = p0x + smax * tx, p0y + smax * ty
    ta = np.arctan2(ay, ax)
    tb = np.arctan2(by, bx)
    lo = np.minimum(ta, tb)
    hi = np.maximum(ta, tb)
    ok = ok & (hi - lo > 1e-3)
    return lo, hi, ok


def sim_fixed_count(seed: int, n_poses: int, pts_per_pose: int, noise_sigma: float = 0.0,
                    Rlc: np.ndarray = GT_RLC, tlc: np.ndarray = GT_TLC) -> ObservationSet:
    """C2 — same pose distribution as the reference simulation, but every pose carries
    exactly `pts_per_pose` equally spaced rays inside its valid angular interval
    (SURVEY.md §8d), so N = n_poses * pts_per_pose exactly.  Poses whose board line does
    not cross the lidar's field are redrawn (deterministically, from the same stream)."""
    rng = np.random.default_rng(seed)
    Rs, ts, los, his = [], [], [], []
    have = 0
    while have < n_poses:
        want = max(64, int((n_poses - have) * 1.3) + 8)
        Rca, tca = _draw_tag_poses(rng, want)
        n, d = _plane_in_laser(Rlc, tlc, Rca, tca)
        lo, hi, ok = _valid_interval(n, d)
        Rs.append(Rca[ok]); ts.append(tca[ok]); los.append(lo[ok]); his.append(hi[ok])
        have += int(ok.sum())
    Rca = np.concatenate(Rs)[:n_poses]
    tca = np.concatenate(ts)[:n_poses]
    lo = np.concatenate(los)[:n_poses]
    hi = np.concatenate(his)[:n_poses]
    n, d = _plane_in_laser(Rlc, tlc, Rca, tca)
    K = pts_per_pose
    frac = (np.arange(K) + 0.5) / K
    theta = lo[:, None] + (hi - lo)[:, None] * frac[None, :]  # [P,K]
    cx, sy = np.cos(theta), np.sin(theta)
    depth = -d[:, None] / (cx * n[:, 0:1] + sy * n[:, 1:2])
    if noise_sigma > 0.0:
        depth = depth + rng.normal(0.0, noise_sigma, size=depth.shape)
    pts = np.stack([depth * cx, depth * sy, np.zeros_like(depth)], axis=2).reshape(-1, 3)
    off = np.arange(n_poses + 1, dtype=np.int64) * K
    return ObservationSet(rot_to_quat_wxyz(Rca), tca.copy(), off, np.ascontiguousarray(pts), off.copy(), pts.copy())


def perturbed_gt(rng: np.random.Generator, rot_rad: float = 0.2, trans_m: float = 0.3):
    """A ground-truth Tlc drawn around the simulation's GT."""
    a = (rng.random(3) * 2 - 1) * rot_rad
    dR = rot_zyx(a[0], a[1], a[2])[0]
    dt = (rng.random(3) * 2 - 1) * trans_m
    return dR @ GT_RLC, GT_TLC + dt


def sim_batch(seed: int, n_problems: int, n_poses: int, pts_per_pose: int,
              noise_sigma: float = 0.0) -> Tuple[List[ObservationSet], np.ndarray]:
    """C3/C4 — independent T_cl problems: different seeds AND different ground truths.
    Returns (problems, gt_Tcl[n_problems,4,4])."""
    master = np.random.default_rng(seed)
    probs, gts = [], np.empty((n_problems, 4, 4))
    for k in range(n_problems):
        Rlc, tlc = perturbed_gt(master)
        s = int(master.integers(0, 2**31 - 1))
        probs.append(sim_fixed_count(s, n_poses, pts_per_pose, noise_sigma, Rlc, tlc))
        gts[k] = tlc_to_tcl(Rlc, tlc)
    return probs, gts


def sim_degenerate(kind: str, seed: int = 3, n_poses: int = 40, n_rays: int = 60, noise_sigma: float = 0.0) -> ObservationSet:
    """Unobservable configurations (the closed form's "system unobservable" branch, src/LaseCamCalCeres.cpp:164-181, and
    the null-space report of the analysis pass, :371-379):
      "only_pitch"       all boards rotate about one camera axis and move along one (main/calibr_simulation.cpp:50-51):
                         the normal matrix is numerically (not exactly) singular;
      "parallel_boards"  every board faces the camera squarely (R_ca = I): plane normal (0, 0, 1) for all poses, six
                         columns of A are exactly zero, A^T A has exact zero pivots."""
    rng = np.random.default_rng(seed)
    P = n_poses
    if kind == "only_pitch":
        ang = (rng.random(P) * 2 - 1) * np.pi / 6
        Rca = rot_zyx(np.zeros(P), ang, np.zeros(P))
        tca = np.stack([np.zeros(P), np.zeros(P), rng.uniform(1, 5, P)], 1)
    elif kind == "parallel_boards":
        Rca = np.tile(np.eye(3), (P, 1, 1))
        tca = np.stack([rng.uniform(-1, 1, P), rng.uniform(-1, 1, P), rng.uniform(1, 5, P)], 1)
    else:
        raise ValueError(kind)
    n, d = _plane_in_laser(GT_RLC, GT_TLC, Rca, tca)
    theta = np.linspace(-0.6, 0.6, n_rays)
    pts, off = [], [0]
    for i in range(P):
        den = np.cos(theta) * n[i, 0] + np.sin(theta) * n[i, 1]
        with np.errstate(divide="ignore"):
            depth = -d[i] / den
        if noise_sigma > 0:
            depth = depth + rng.normal(0, noise_sigma, depth.shape)
        ok = np.isfinite(depth) & (depth > 0) & (depth < 8)
        pts.append(np.stack([depth[ok] * np.cos(theta[ok]), depth[ok] * np.sin(theta[ok]), np.zeros(int(ok.sum()))], 1))
        off.append(off[-1] + int(ok.sum()))
    off = np.array(off, dtype=np.int64)
    flat = np.ascontiguousarray(np.concatenate(pts))
    return ObservationSet(rot_to_quat_wxyz(Rca), tca, off, flat, off.copy(), flat.copy())


# ----------------------------------------------------------------------------------
# C3/C4 shards: problem k is a pure function of (seed, k), vectorised over problems
# ----------------------------------------------------------------------------------
def _rot_to_quat_wxyz_vec(R: np.ndarray) -> np.ndarray:
    """Vectorised `rot_to_quat_wxyz` (same branches, same operation order -> same bits)."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    N = R.shape[0]
    q = np.empty((N, 4))
    d = np.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]], axis=1)
    tr = d[:, 0] + d[:, 1] + d[:, 2]
    pos = tr > 0.0
    if pos.any():
        m = R[pos]
        t = np.sqrt(tr[pos] + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        q[pos] = np.stack([w, (m[:, 2, 1] - m[:, 1, 2]) * t, (m[:, 0, 2] - m[:, 2, 0]) * t, (m[:, 1, 0] - m[:, 0, 1]) * t], axis=1)
    if (~pos).any():
        idx = np.nonzero(~pos)[0]
        q[idx] = rot_to_quat_wxyz(R[idx]).reshape(-1, 4)
    return q


@dataclass
class ProblemShard:
    """Problems [lo, hi) of a seeded batch in pose-major form (every scan has exactly K points)."""

    lo: int
    hi: int
    tag_q: np.ndarray   # [P, n_poses, 4] (w,x,y,z)
    tag_t: np.ndarray   # [P, n_poses, 3]
    pts: np.ndarray     # [P, n_poses, K, 3]
    gt_Tcl: np.ndarray  # [P, 4, 4]

    @property
    def n_problems(self) -> int:
        return self.hi - self.lo

    def problem(self, k: int) -> ObservationSet:
        """Problem with LOCAL index k as an ObservationSet (what the reference's call surface takes)."""
        n_poses, K = self.pts.shape[1], self.pts.shape[2]
        off = np.arange(n_poses + 1, dtype=np.int64) * K
        flat = np.ascontiguousarray(self.pts[k].reshape(-1, 3))
        return ObservationSet(self.tag_q[k].copy(), self.tag_t[k].copy(), off, flat, off.copy(), flat.copy())

    def records(self) -> Tuple[np.ndarray, np.ndarray]:
        """The flat clc_observation records of the whole shard + CSR problem offsets: the vectorised equivalent of
        the reference's residual-block loop (src/LaseCamCalCeres.cpp:222-257; plane :227-231, scale :239-240) for
        (use_linefitting_data = anything, use_boundary_constraint = false) — bitwise what
        clc_flatten_observations produces problem by problem (tests/test_simdata_shard.py)."""
        P, n_poses, K = self.pts.shape[0], self.pts.shape[1], self.pts.shape[2]
        R = quat_wxyz_to_rot(self.tag_q.reshape(-1, 4))  # Eigen: Quaterniond -> toRotationMatrix, as flatten does
        n = R[:, :, 2]
        t = self.tag_t.reshape(-1, 3)
        d = -((n[:, 0] * t[:, 0] + n[:, 1] * t[:, 1]) + n[:, 2] * t[:, 2])
        rec = np.empty((P * n_poses, K, 8))
        rec[:, :, 0:3] = n[:, None, :]
        rec[:, :, 3] = d[:, None]
        rec[:, :, 4:7] = self.pts.reshape(P * n_poses, K, 3)
        rec[:, :, 7] = 1.0 / np.sqrt(float(K))
        off = np.arange(P + 1, dtype=np.int64) * (n_poses * K)
        return rec.reshape(-1, 8), off

    def start_poses(self, sigma: float = 0.05) -> np.ndarray:
        """Initial guesses: the ground truth moved by a seeded N(0, sigma) tangent step (first order in the
        rotation: q <- normalize(q (x) [1, dtheta/2]), PoseLocalParameterization::Plus), [P,7]."""
        x = np.empty((self.n_problems, 7))
        for k in range(self.n_problems):
            g = pose7_from_T(self.gt_Tcl[k])
            dl = np.random.default_rng([977, self.lo + k]).normal(size=6) * sigma
            b = dl[3:] * 0.5
            ax, ay, az, aw = g[3], g[4], g[5], g[6]
            w = aw - ax * b[0] - ay * b[1] - az * b[2]
            qx = aw * b[0] + ax + ay * b[2] - az * b[1]
            qy = aw * b[1] + ay + az * b[0] - ax * b[2]
            qz = aw * b[2] + az + ax * b[1] - ay * b[0]
            nrm = np.sqrt(qx * qx + qy * qy + qz * qz + w * w)
            x[k] = (g[0] + dl[0], g[1] + dl[1], g[2] + dl[2], qx / nrm, qy / nrm, qz / nrm, w / nrm)
        return x


def sim_shard(seed: int, lo: int, hi: int, n_poses: int, pts_per_pose: int, noise_sigma: float = 0.0,
              candidates: int = 64) -> ProblemShard:
    """C3/C4 — problems [lo, hi) of a batch of independent T_cl problems (own ground truth drawn around the
    simulation's, own board poses, own range noise; same distributions as `sim_batch` / main/calibr_simulation.cpp).
    Problem k depends on (seed, k) only, so every rank of a sharded job generates exactly its own shard and any
    single problem can be regenerated for a check.  Vectorised over the problems of the shard."""
    P, K = hi - lo, pts_per_pose
    u_gt = np.empty((P, 6)); u_pose = np.empty((P, candidates, 6)); noise = np.zeros((P, n_poses, K))
    rngs = []
    for i in range(P):
        rng = np.random.default_rng([seed, lo + i])
        u_gt[i] = rng.random(6)
        u_pose[i] = rng.random((candidates, 6))
        if noise_sigma > 0.0:
            noise[i] = rng.normal(0.0, noise_sigma, size=(n_poses, K))
        rngs.append(rng)
    # ground truths (perturbed_gt)
    a = (u_gt[:, :3] * 2 - 1) * 0.2
    Rlc = rot_zyx(a[:, 0], a[:, 1], a[:, 2]) @ GT_RLC
    tlc = GT_TLC + (u_gt[:, 3:] * 2 - 1) * 0.3
    # candidate board poses (_draw_tag_poses) and the first n_poses whose board line crosses the lidar's field
    ang = (u_pose[:, :, :3] * 2.0 - 1.0) * (np.pi / 6.0)
    Rca = rot_zyx(ang[..., 0], ang[..., 1], ang[..., 2])                      # [P,C,3,3]
    tca = np.stack([u_pose[..., 3] * 6.0 - 3.0, u_pose[..., 4] * 6.0 - 3.0, 1.0 + u_pose[..., 5] * 4.0], axis=-1)
    n, d = _plane_in_laser(Rlc[:, None], tlc[:, None], Rca, tca)             # [P,C,3], [P,C]
    lo_t, hi_t, ok = _valid_interval(n.reshape(-1, 3), d.reshape(-1))
    ok = ok.reshape(P, candidates)
    order = np.argsort(~ok, axis=1, kind="stable")[:, :n_poses]              # first valid candidates, in draw order
    enough = ok.sum(axis=1) >= n_poses
    rows = np.arange(P)[:, None]
    Rca, tca, n, d = Rca[rows, order], tca[rows, order], n[rows, order], d[rows, order]
    lo_t, hi_t = lo_t.reshape(P, candidates)[rows, order], hi_t.reshape(P, candidates)[rows, order]
    for i in np.nonzero(~enough)[0]:  # rare: keep drawing from the problem's own stream
        have = int(ok[i].sum())
        while have < n_poses:
            Rc, tc = _draw_tag_poses(rngs[i], candidates)
            nn, dd = _plane_in_laser(Rlc[i], tlc[i], Rc, tc)
            l2, h2, o2 = _valid_interval(nn, dd)
            for j in np.nonzero(o2)[0][: n_poses - have]:
                Rca[i, have], tca[i, have], n[i, have], d[i, have] = Rc[j], tc[j], nn[j], dd[j]
                lo_t[i, have], hi_t[i, have] = l2[j], h2[j]
                have += 1
    frac = (np.arange(K) + 0.5) / K
    theta = lo_t[..., None] + (hi_t - lo_t)[..., None] * frac                # [P,n_poses,K]
    cx, sy = np.cos(theta), np.sin(theta)
    depth = -d[..., None] / (cx * n[..., 0:1] + sy * n[..., 1:2])
    if noise_sigma > 0.0:
        depth = depth + noise
    pts = np.stack([depth * cx, depth * sy, np.zeros_like(depth)], axis=-1)
    gt = np.empty((P, 4, 4))
    for i in range(P):
        gt[i] = tlc_to_tcl(Rlc[i], tlc[i])
    tag_q = _rot_to_quat_wxyz_vec(Rca.reshape(-1, 3, 3)).reshape(P, n_poses, 4)
    return ProblemShard(lo, hi, tag_q, np.ascontiguousarray(tca), np.ascontiguousarray(pts), gt)


def sim_shard_records(seed: int, lo: int, hi: int, n_poses: int, pts_per_pose: int, noise_sigma: float = 0.0,
                      chunk: int = 256):
    """Records, offsets, start poses and ground truths of problems [lo, hi), generated in chunks of `chunk` problems
    so the temporaries stay small (a C4 shard is 8 192 problems x 10^4 records = 5.2 GB of records).
    -> (records [N,8], offsets [P+1], x0 [P,7], gt_Tcl [P,4,4])"""
    P = hi - lo
    per = n_poses * pts_per_pose
    rec = np.empty((P * per, 8))
    x0 = np.empty((P, 7)); gt = np.empty((P, 4, 4))
    for c0 in range(0, P, chunk):
        c1 = min(P, c0 + chunk)
        sh = sim_shard(seed, lo + c0, lo + c1, n_poses, pts_per_pose, noise_sigma)
        r, _ = sh.records()
        rec[c0 * per : c1 * per] = r
        x0[c0:c1] = sh.start_poses()
        gt[c0:c1] = sh.gt_Tcl
    off = np.arange(P + 1, dtype=np.int64) * per
    return rec, off, x0, gt


# Board geometry of the boundary terms, LaseCamCalCeres.cpp:262-268.
BOARD_ORIG = 0.0265 + 0.0165
BOARD_SIZE = 0.5


def sim_board_edges(seed: int, n_poses: int, pts_per_pose: int, noise_sigma: float = 0.0,
                    Rlc: np.ndarray = GT_RLC, tlc: np.ndarray = GT_TLC) -> ObservationSet:
    """C5 — scans clipped to the 0.5 m board so that `points.front()` lies on the tag-frame
    edge {y=-0.043, z=0} and `points.back()` on {x=-0.043, z=0}: the geometry the two
    board-edge residuals of LaseCamCalCeres.cpp:258-294 assume.

    Construction (in the laser frame, then mapped to the camera frame with the GT Tcl):
    pick A on the first edge and B on the second, choose the tag orientation so that the
    chord A-B lies in the lidar plane z=0, place A at a random (range, bearing) in front of
    the lidar, and sample rays from A to B inclusive."""
    rng = np.random.default_rng(seed)
    o = BOARD_ORIG
    P, K = n_poses, pts_per_pose
    tag_q = np.empty((P, 4)); tag_t = np.empty((P, 3)); pts = np.empty((P, K, 3))
    Rcl = Rlc.T
    tcl = -Rlc.T @ tlc
    i = 0
    while i < P:
        A = np.array([-o + rng.uniform(0.15, 0.45), -o, 0.0])        # on edge y = -o
        B = np.array([-o, -o + rng.uniform(0.15, 0.45), 0.0])        # on edge x = -o
        u = (B - A) / np.linalg.norm(B - A)                           # chord direction (tag)
        phi = rng.uniform(-np.pi, np.pi)
        v = np.array([np.cos(phi), np.sin(phi), 0.0])                 # its image in the lidar plane
        psi = rng.uniform(np.pi / 6, np.pi / 2) * rng.choice([-1.0, 1.0])
        e3 = np.array([0.0, 0.0, 1.0])
        w = np.cross(e3, v)
        nl = np.cos(psi) * w + np.sin(psi) * e3                       # tag normal in laser frame
        # R_la maps u->v, e3_tag->nl, (e3 x u)->(nl x v)
        Mt = np.stack([u, np.cross(e3, u), e3], axis=1)
        Ml = np.stack([v, np.cross(nl, v), nl], axis=1)
        Rla = Ml @ Mt.T
        rho = rng.uniform(0.8, 4.0)
        th = rng.uniform(-np.pi / 3, np.pi / 3)
        Al = np.array([rho * np.cos(th), rho * np.sin(th), 0.0])
        tla = Al - Rla @ A
        Bl = Rla @ B + tla
        thA, thB = np.arctan2(Al[1], Al[0]), np.arctan2(Bl[1], Bl[0])
        if not (Bl[0] > 0.2 and abs(thB) < np.pi / 2 - 0.05 and thB > thA + 1e-3):
            continue  # scan order must run A -> B with increasing bearing
        n = Rla[:, 2]
        d = -n @ tla
        theta = np.linspace(thA, thB, K)
        den = np.cos(theta) * n[0] + np.sin(theta) * n[1]
        if np.any(np.abs(den) < 1e-3):
            continue
        depth = -d / den
        if np.any(depth <= 0):
            continue
        if noise_sigma > 0.0:
            depth = depth + rng.normal(0.0, noise_sigma, size=depth.shape)
        pts[i, :, 0] = depth * np.cos(theta)
        pts[i, :, 1] = depth * np.sin(theta)
        pts[i, :, 2] = 0.0
        Rca = Rcl @ Rla
        tca = Rcl @ tla + tcl
        tag_q[i] = rot_to_quat_wxyz(Rca)
        tag_t[i] = tca
        i += 1
    off = np.arange(P + 1, dtype=np.int64) * K
    flat = np.ascontiguousarray(pts.reshape(-1, 3))
    return ObservationSet(tag_q, tag_t, off, flat, off.copy(), flat.copy())
