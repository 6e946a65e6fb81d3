// clc_stream.hpp — the streaming loops of the evaluation kernels, one per device layout of the observation array:
//   stream_tiles         64-byte AoSoA tiles (the records as handed over, re-tiled)          64 B / observation
//   stream_ctiles[_deep] compact tiles + group table (lossless, per-point arithmetic)        28 B / observation
//   stream_rows[_policy] row layout, per-scan moments (clc_rows.hpp) — the default           17 B / observation
// A wave owns a contiguous run of tiles / rows, keeps several loads in flight and accumulates into 28 (LM) or 45 (closed
// form) registers per lane; `get_pose` is called after the prologue loads are issued.
#pragma once
#include "clc_device.hpp"

namespace clc {

template <bool NT>
__device__ __forceinline__ void load_tile(const double* __restrict__ tiles, long long tile, int lane,
                                          double2 (&f)[8]) {
  const v2d* base = reinterpret_cast<const v2d*>(tiles + tile * TILE_DOUBLES) + lane;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v2d v;
    if (NT)
      v = __builtin_nontemporal_load(base + k * 64);
    else
      v = base[k * 64];
    f[k].x = v[0];
    f[k].y = v[1];
  }
}

template <bool WITH_LOSS, bool WITH_JAC>
__device__ __forceinline__ void accumulate_tile(const PoseU& P, double inv_lf2, const double2 (&f)[8],
                                                double (&acc)[NACC]) {
  accumulate_observation<WITH_LOSS, WITH_JAC>(P, inv_lf2, f[0].x, f[1].x, f[2].x, f[3].x, f[4].x,
                                              f[5].x, f[6].x, f[7].x, acc);
  accumulate_observation<WITH_LOSS, WITH_JAC>(P, inv_lf2, f[0].y, f[1].y, f[2].y, f[3].y, f[4].y,
                                              f[5].y, f[6].y, f[7].y, acc);
}

// The streaming loop shared by every evaluation kernel: wave `wave_global` of `n_waves`
// consumes tiles wave_global, wave_global + n_waves, ... of an array of n records (whole tiles
// + one ragged, zero-padded tile that is masked by index).
//
// `get_pose(PoseU&) -> bool` fetches the point to evaluate (and the solve's status: false = the
// solve has terminated, nothing to do).  It is called AFTER the wave's first tile loads have been
// issued: the pose was written a few microseconds earlier by the controller on another CU, so
// reading it is a trip to memory — which now overlaps the first tile's latency instead of
// preceding it (the kernel used to spend ~1.3 us of its ~8 waiting for status, pose and first
// tile one after the other).
template <bool WITH_LOSS, bool WITH_JAC, bool PREFETCH, bool NT, class PoseFn>
__device__ __forceinline__ bool stream_tiles(const double* __restrict__ tiles, const long long n,
                                             const WaveMap wm, const int lane, PoseFn get_pose,
                                             const double& inv_lf2, double (&acc)[NACC]) {
  const long long n_full = n / TILE;
  const int rem = (int)(n % TILE);
  const long long T = n_full + (rem != 0 ? 1 : 0);
  const long long t_begin = wm.begin(T), t_last = wm.end(T);     // this wave's run (may include the ragged tile)
  const long long t_end = t_last < n_full ? t_last : n_full;      // whole tiles only
  PoseU P;
  if (PREFETCH) {
    // two register buffers; the loads of the wave's next tile are in flight while one is consumed
    double2 fa[8], fb[8];
    long long tile = t_begin;
    if (tile < t_end) load_tile<NT>(tiles, tile, lane, fa);
    if (!get_pose(P)) return false;
    while (tile < t_end) {
      if (tile + 1 < t_end) load_tile<NT>(tiles, tile + 1, lane, fb);
      accumulate_tile<WITH_LOSS, WITH_JAC>(P, inv_lf2, fa, acc);
      if (tile + 1 >= t_end) break;
      if (tile + 2 < t_end) load_tile<NT>(tiles, tile + 2, lane, fa);
      accumulate_tile<WITH_LOSS, WITH_JAC>(P, inv_lf2, fb, acc);
      tile += 2;
    }
  } else {
    if (!get_pose(P)) return false;
    for (long long tile = t_begin; tile < t_end; ++tile) {
      double2 f[8];
      load_tile<NT>(tiles, tile, lane, f);
      accumulate_tile<WITH_LOSS, WITH_JAC>(P, inv_lf2, f, acc);
    }
  }
  if (rem != 0 && t_begin <= n_full && n_full < t_last) {  // ragged last tile: masked lanes
    double2 f[8];
    load_tile<false>(tiles, n_full, lane, f);
    if (2 * lane < rem)
      accumulate_observation<WITH_LOSS, WITH_JAC>(P, inv_lf2, f[0].x, f[1].x, f[2].x, f[3].x, f[4].x,
                                                  f[5].x, f[6].x, f[7].x, acc);
    if (2 * lane + 1 < rem)
      accumulate_observation<WITH_LOSS, WITH_JAC>(P, inv_lf2, f[0].y, f[1].y, f[2].y, f[3].y, f[4].y,
                                                  f[5].y, f[6].y, f[7].y, acc);
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// Compact layout.  Every record of one scan carries the same plane and
// scale (src/LaseCamCalCeres.cpp:231,240,245), so the 64-byte records compress LOSSLESSLY into
//   * a group table  groups[g] = {n.x, n.y, n.z, d, scale, 0}  (48 B, one entry per run of
//     records with bit-identical (n, d, scale)), and
//   * compact tiles of 128 points: x[128], y[128], z[128] (FP64) + gid[128] (u32) = 3 584 B,
// i.e. 28 bytes of HBM traffic per observation instead of 64.  The arithmetic per observation
// is unchanged (same operands, same order), so results are bitwise those of the 64-byte path.
// ---------------------------------------------------------------------------------------

struct CTile {  // one lane's share of a compact tile: 2 points + their group ids
  double2 p[3];
  v2u g;
};

struct Planes2 {  // gathered group entries of the lane's 2 points
  double nx[2], ny[2], nz[2], d[2], s[2];
};

template <bool NT>
__device__ __forceinline__ void load_ctile(const double* __restrict__ ctiles, long long tile, int lane, CTile& c) {
  const double* base = ctiles + tile * CTILE_DOUBLES;
  const v2d* pb = reinterpret_cast<const v2d*>(base) + lane;
  const v2u* gb = reinterpret_cast<const v2u*>(base + 3 * TILE) + lane;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v2d v;
    if (NT) v = __builtin_nontemporal_load(pb + k * 64);
    else v = pb[k * 64];
    c.p[k].x = v[0];
    c.p[k].y = v[1];
  }
  if (NT) c.g = __builtin_nontemporal_load(gb);
  else c.g = *gb;
}

// Group-table gather: consecutive points belong to the same scan, so the 64 lanes of a wave
// read one or two distinct 48-byte entries per instruction (broadcast out of L1/L2).
__device__ __forceinline__ void gather_planes(const double* __restrict__ groups, const v2u g, Planes2& q) {
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const double* gp = groups + (size_t)g[o] * GROUP_DOUBLES;
    const v2d a = *reinterpret_cast<const v2d*>(gp);
    const v2d b = *reinterpret_cast<const v2d*>(gp + 2);
    q.nx[o] = a[0]; q.ny[o] = a[1]; q.nz[o] = b[0]; q.d[o] = b[1];
    q.s[o] = gp[4];
  }
}

template <bool WITH_LOSS, bool WITH_JAC, int O>
__device__ __forceinline__ void accumulate_cpoint(const PoseU& P, double inv_lf2, const CTile& c, const Planes2& q,
                                                  double (&acc)[NACC]) {
  accumulate_observation<WITH_LOSS, WITH_JAC>(P, inv_lf2, q.nx[O], q.ny[O], q.nz[O], q.d[O],
                                              O == 0 ? c.p[0].x : c.p[0].y, O == 0 ? c.p[1].x : c.p[1].y,
                                              O == 0 ? c.p[2].x : c.p[2].y, q.s[O], acc);
}

// Streaming loop over compact tiles, software-pipelined in two ways: the point/gid loads of the
// wave's next tile are issued before the current tile is consumed, and the group gather of the
// next tile is issued between the two observations of the current one (its gids have landed by
// then, and the second observation's arithmetic hides the gather latency).
template <bool WITH_LOSS, bool WITH_JAC, bool NT, class PoseFn>
__device__ __forceinline__ bool stream_ctiles(const double* __restrict__ ctiles,
                                              const double* __restrict__ groups, const long long n,
                                              const WaveMap wm, const int lane, PoseFn get_pose,
                                              const double& inv_lf2, double (&acc)[NACC]) {
  const long long n_full = n / TILE;
  const int rem = (int)(n % TILE);
  const long long T = n_full + (rem != 0 ? 1 : 0);
  const long long t_begin = wm.begin(T), t_last = wm.end(T);
  const long long t_end = t_last < n_full ? t_last : n_full;
  CTile A, B;
  Planes2 PA, PB;
  long long tile = t_begin;
  bool have = tile < t_end;
  if (have) load_ctile<NT>(ctiles, tile, lane, A);
  PoseU P;
  if (!get_pose(P)) return false;  // after the first loads are in flight (see stream_tiles)
  if (have) gather_planes(groups, A.g, PA);
  while (have) {
    const bool has1 = tile + 1 < t_end;
    if (has1) load_ctile<NT>(ctiles, tile + 1, lane, B);
    accumulate_cpoint<WITH_LOSS, WITH_JAC, 0>(P, inv_lf2, A, PA, acc);
    if (has1) gather_planes(groups, B.g, PB);
    accumulate_cpoint<WITH_LOSS, WITH_JAC, 1>(P, inv_lf2, A, PA, acc);
    if (!has1) break;
    const bool has2 = tile + 2 < t_end;
    if (has2) load_ctile<NT>(ctiles, tile + 2, lane, A);
    accumulate_cpoint<WITH_LOSS, WITH_JAC, 0>(P, inv_lf2, B, PB, acc);
    if (has2) gather_planes(groups, A.g, PA);
    accumulate_cpoint<WITH_LOSS, WITH_JAC, 1>(P, inv_lf2, B, PB, acc);
    tile += 2;
    have = has2;
  }
  if (rem != 0 && t_begin <= n_full && n_full < t_last) {  // ragged last tile (zero padded, gid 0)
    load_ctile<false>(ctiles, n_full, lane, A);
    gather_planes(groups, A.g, PA);
    if (2 * lane < rem) accumulate_cpoint<WITH_LOSS, WITH_JAC, 0>(P, inv_lf2, A, PA, acc);
    if (2 * lane + 1 < rem) accumulate_cpoint<WITH_LOSS, WITH_JAC, 1>(P, inv_lf2, A, PA, acc);
  }
  return true;
}

// Deeper software pipeline for arrays beyond the Infinity Cache.  With one tile in flight per
// wave a CU keeps 8 x 3.5 KB = 28 KB outstanding, which by Little's law sustains only ~4.4 TB/s
// chip-wide (measured).  Here, while tile t is consumed, the wave has in flight: the points of
// tiles t+1 and t+2, the group ids of t+2 and t+3, and the plane gather of t+1.  Issue order per
// trip is oldest-needed-first — gather(t+1), points(t+2), gid(t+3) — so the in-order vmcnt wait
// for what tile t+1 needs never drains the younger loads.  Buffers rotate with period 3 (points,
// gids) and 2 (planes); the trip loop is unrolled x6 so every buffer index is a compile-time
// constant (runtime-indexed register arrays would go to scratch).
struct CPoints {
  double2 p[3];
};

template <bool NT>
__device__ __forceinline__ void load_cpoints(const double* __restrict__ ctiles, long long tile, int lane, CPoints& c) {
  const v2d* pb = reinterpret_cast<const v2d*>(ctiles + tile * CTILE_DOUBLES) + lane;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v2d v;
    if (NT) v = __builtin_nontemporal_load(pb + k * 64);
    else v = pb[k * 64];
    c.p[k].x = v[0];
    c.p[k].y = v[1];
  }
}

template <bool NT>
__device__ __forceinline__ v2u load_cgid(const double* __restrict__ ctiles, long long tile, int lane) {
  const v2u* gb = reinterpret_cast<const v2u*>(ctiles + tile * CTILE_DOUBLES + 3 * TILE) + lane;
  if (NT) return __builtin_nontemporal_load(gb);
  return *gb;
}

template <bool WITH_LOSS, bool WITH_JAC, int O>
__device__ __forceinline__ void accumulate_cpoint2(const PoseU& P, double inv_lf2, const CPoints& c, const Planes2& q,
                                                   double (&acc)[NACC]) {
  accumulate_observation<WITH_LOSS, WITH_JAC>(P, inv_lf2, q.nx[O], q.ny[O], q.nz[O], q.d[O],
                                              O == 0 ? c.p[0].x : c.p[0].y, O == 0 ? c.p[1].x : c.p[1].y,
                                              O == 0 ? c.p[2].x : c.p[2].y, q.s[O], acc);
}

template <bool WITH_LOSS, bool WITH_JAC, bool NT, class PoseFn>
__device__ __forceinline__ bool stream_ctiles_deep(const double* __restrict__ ctiles,
                                                   const double* __restrict__ groups, const long long n,
                                                   const WaveMap wm, const int lane, PoseFn get_pose,
                                                   const double& inv_lf2, double (&acc)[NACC]) {
  const long long n_full = n / TILE;
  const int rem = (int)(n % TILE);
  const long long T = n_full + (rem != 0 ? 1 : 0);
  const long long t_begin = wm.begin(T), t_last = wm.end(T);
  const long long t_end = t_last < n_full ? t_last : n_full;
  CPoints pt[3];
  v2u gid[3];
  Planes2 pl[2];
  PoseU P;
  if (t_begin < t_end) {
    // prologue: gids of the first three tiles, points of the first two, planes of the first
    gid[0] = load_cgid<NT>(ctiles, t_begin, lane);
    if (t_begin + 1 < t_end) gid[1] = load_cgid<NT>(ctiles, t_begin + 1, lane);
    if (t_begin + 2 < t_end) gid[2] = load_cgid<NT>(ctiles, t_begin + 2, lane);
    load_cpoints<NT>(ctiles, t_begin, lane, pt[0]);
    if (t_begin + 1 < t_end) load_cpoints<NT>(ctiles, t_begin + 1, lane, pt[1]);
  }
  if (!get_pose(P)) return false;  // after the prologue loads are in flight (see stream_tiles)
  if (t_begin < t_end) {
    gather_planes(groups, gid[0], pl[0]);
    for (long long base = t_begin; base < t_end; base += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const long long t = base + u;
        if (t >= t_end) break;
        // trip for tile t: points in pt[u%3], planes in pl[u%2]; gid[(u+1)%3] = gid(t+1) has landed
        if (t + 1 < t_end) gather_planes(groups, gid[(u + 1) % 3], pl[(u + 1) % 2]);
        if (t + 2 < t_end) load_cpoints<NT>(ctiles, t + 2, lane, pt[(u + 2) % 3]);
        if (t + 3 < t_end) gid[u % 3] = load_cgid<NT>(ctiles, t + 3, lane);  // gid(t) is dead: its planes are gathered
        accumulate_cpoint2<WITH_LOSS, WITH_JAC, 0>(P, inv_lf2, pt[u % 3], pl[u % 2], acc);
        accumulate_cpoint2<WITH_LOSS, WITH_JAC, 1>(P, inv_lf2, pt[u % 3], pl[u % 2], acc);
      }
    }
  }
  if (rem != 0 && t_begin <= n_full && n_full < t_last) {  // ragged last tile (zero padded, gid 0)
    CTile A;
    Planes2 PA;
    load_ctile<false>(ctiles, n_full, lane, A);
    gather_planes(groups, A.g, PA);
    if (2 * lane < rem) accumulate_cpoint<WITH_LOSS, WITH_JAC, 0>(P, inv_lf2, A, PA, acc);
    if (2 * lane + 1 < rem) accumulate_cpoint<WITH_LOSS, WITH_JAC, 1>(P, inv_lf2, A, PA, acc);
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// Row layout (clc_rows.hpp; SURVEY.md §8f row 3 taken to its end).  At upload the records are grouped into scans
// (runs of bit-identical (n, d, scale)), every scan is padded to whole ROWS of 64 points, and the device keeps
//   * xy[row][64] (x, y) interleaved — 1 KiB per row, one coalesced 16-byte load per lane — and
//   * desc[row] (64 B, wave-uniform: scalar loads): the scan's plane and scale, the valid count of the row, and
//     whether the row starts a scan,
// i.e. 17 B of traffic per observation for scans that fill their rows (28 B in the compact layout, 64 B algorithmic).
// Arrays with some p.z != 0 (never produced by the reference's own scan conversion, but Oberserve::points is Vector3d) keep a
// third 8-byte value per point: rows of ROW_DOUBLES_Z = 192 doubles, 64 z after the 64 (x, y) pairs, 25.4 B per observation.
// A wave owns a contiguous run of rows, keeps DEPTH row loads in flight, accumulates
// per-scan moments (one point per lane per row, ~26 FP64 instructions) and expands them into the 28 accumulators
// when the scan changes (rows_flush).  Same lane->row map for every launch: bitwise reproducible.
// ---------------------------------------------------------------------------------------
#ifndef CLC_ROWS_DEPTH
#define CLC_ROWS_DEPTH 8
#endif
constexpr int ROWS_DEPTH = CLC_ROWS_DEPTH;
// rows that carry z, in the kernels that also hold the LM controller (24 B per point in flight instead of 16, 14 moments instead
// of 9: with 8 rows in flight hipcc parks 52-68 bytes per lane in scratch there)
constexpr int ROWS_DEPTH_Z = 6;

template <bool NT, int STRIDE = ROW_DOUBLES>
__device__ __forceinline__ v2d load_row(const double* __restrict__ xy, long long row, int lane) {
  const v2d* p = reinterpret_cast<const v2d*>(xy + row * STRIDE) + lane;
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
// Rows of scans with points off the lidar plane (ROW_DOUBLES_Z per row): 64 z follow the 64 (x, y) pairs.
template <bool NT>
__device__ __forceinline__ double load_row_z(const double* __restrict__ xy, long long row, int lane) {
  const double* p = xy + row * ROW_DOUBLES_Z + ROW_DOUBLES + lane;
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// Descriptor of a row as ONE 8-byte vector load per lane (lane l holds double l & 7 of the 64-byte descriptor), so that it
// rides the same in-order, DEPTH-deep vmcnt pipeline as the points; the wave-uniform fields are then read with
// v_readlane.  (Scalar loads of the descriptors, one row ahead, left every wave waiting ~1 us of HBM latency per
// 128-byte line of descriptors: 4-5 TB/s beyond the Infinity Cache instead of what the row stream allows.)
__device__ __forceinline__ double load_desc_lane(const RowDesc* __restrict__ desc, int row, int lane) {
  return reinterpret_cast<const double*>(desc + row)[lane & 7];
}

// What a wave does with the rows it streams is a policy: begin_scan(pose, plane of the scan), point(x, y) for every
// valid point of a row, flush(acc) when the scan ends.  LmRows = the LM evaluation (clc_rows.hpp), Normal9Rows = the 9x9
// normal equation of the closed-form initialiser (K5).
template <bool WITH_LOSS>
struct LmRows {
  static constexpr int NA = NACC;
  static constexpr bool Z = false;
  const double& inv_lf2;  // set by get_pose (it may depend on options that arrive late)
  RowPlane q;
  RowMoments M;
  __device__ __forceinline__ explicit LmRows(const double& inv) : inv_lf2(inv) {}
  __device__ __forceinline__ void begin_scan(const PoseU& P, double nx, double ny, double nz, double d, double s) {
    rows_plane_setup(P.R, P.t, nx, ny, nz, d, s, q);
    rows_moments_reset<WITH_LOSS>(M);
  }
  __device__ __forceinline__ void point(double x, double y, bool renorm) { rows_point<WITH_LOSS>(q, inv_lf2, x, y, M, renorm); }
  __device__ __forceinline__ void flush(double (&acc)[NACC]) { rows_flush<WITH_LOSS>(q, M, acc); }
};

// The same for rows that carry z (p.z != 0 somewhere in the array): 14 moments per scan (clc_rows.hpp, rows3_*).
template <bool WITH_LOSS>
struct LmRows3 {
  static constexpr int NA = NACC;
  static constexpr bool Z = true;
  const double& inv_lf2;
  RowPlane q;
  RowMoments3 M;
  __device__ __forceinline__ explicit LmRows3(const double& inv) : inv_lf2(inv) {}
  __device__ __forceinline__ void begin_scan(const PoseU& P, double nx, double ny, double nz, double d, double s) {
    rows_plane_setup(P.R, P.t, nx, ny, nz, d, s, q);
    rows3_moments_reset<WITH_LOSS>(M);
  }
  __device__ __forceinline__ void point(double x, double y, double z, bool renorm) { rows3_point<WITH_LOSS>(q, inv_lf2, x, y, z, M, renorm); }
  __device__ __forceinline__ void flush(double (&acc)[NACC]) { rows3_flush<WITH_LOSS>(q, M, acc); }
};

constexpr int NACC9 = 45;

// Row A_k = kron([x, y, 1], n), b_k = -d (src/LaseCamCalCeres.cpp:144-158): A^T A = sum kron(b b^T, n n^T) and
// A^T b = -d kron(sum b, n) share the scan's n, so a lane only accumulates the 6 moments of b = (x, y, 1) per scan
// (5 FP64 instructions per point) and expands them once per scan.
// acc layout (as normal9_kernel): [bb(6: xx xy x yy y 1)] x [nn(6: 00 01 02 11 12 22)] then A^T b (9: b-major).
struct Normal9Rows {
  static constexpr int NA = NACC9;
  static constexpr bool Z = false;  // bar_p = (x, y, 1): the closed form never reads p.z (src/LaseCamCalCeres.cpp:147)
  double nx, ny, nz, md;
  double sxx, sxy, sx, syy, sy, s1;
  __device__ __forceinline__ void begin_scan(const PoseU&, double nx_, double ny_, double nz_, double d, double) {
    nx = nx_; ny = ny_; nz = nz_; md = -d;
    sxx = sxy = sx = syy = sy = s1 = 0.0;
  }
  __device__ __forceinline__ void point(double x, double y, bool) {
    sxx = fma(x, x, sxx);
    sxy = fma(x, y, sxy);
    syy = fma(y, y, syy);
    sx += x;
    sy += y;
    s1 += 1.0;
  }
  __device__ __forceinline__ void flush(double (&acc)[NACC9]) {
    const double nn[6] = {nx * nx, nx * ny, nx * nz, ny * ny, ny * nz, nz * nz};
    const double bb[6] = {sxx, sxy, sx, syy, sy, s1};
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[6 * i + j] = fma(bb[i], nn[j], acc[6 * i + j]);
    const double bv[3] = {sx, sy, s1};
    const double nv[3] = {nx * md, ny * md, nz * md};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[36 + 3 * i + j] = fma(bv[i], nv[j], acc[36 + 3 * i + j]);
  }
};

template <class Policy, bool NT, int DEPTH = ROWS_DEPTH, int STRIDE = ROW_DOUBLES, class PoseFn>
__device__ __forceinline__ bool stream_rows_policy(Policy& pol, const double* __restrict__ xy_all,
                                                   const RowDesc* __restrict__ desc_all, long long r_begin_in,
                                                   long long r_end_in, const int lane, PoseFn get_pose,
                                                   double (&acc)[Policy::NA]) {
  // wave-uniform run [r_begin, r_end): loop control on the scalar unit (32-bit row index relative to the run's first
  // row — 64-bit compares would go through the vector unit)
  const long long r_begin = uniform_ll(r_begin_in);
  const int n = __builtin_amdgcn_readfirstlane((int)(r_end_in - r_begin_in));
  static_assert(!Policy::Z || STRIDE == ROW_DOUBLES_Z, "z rows come with the 192-double stride");
  const double* __restrict__ xy = xy_all + r_begin * STRIDE;
  const RowDesc* __restrict__ desc = desc_all + r_begin;
  v2d buf[DEPTH];
  double dbuf[DEPTH];
  double zbuf[Policy::Z ? DEPTH : 1];
  // Prologue: DEPTH rows in flight, issued UNCONDITIONALLY from clamped row indices (the row arrays carry one padding
  // row, so even an empty run reads mapped memory).  As `if (u < n) load` each load sat in its own branch, and the
  // waits hipcc places at the joins made a wave stall on its first rows of points before it had issued the last ones —
  // and before the barrier in front of the step kernel's controller.
  const int n_last = n > 0 ? n - 1 : 0;
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    const int ru = u < n_last ? u : n_last;
    dbuf[u] = load_desc_lane(desc, ru, lane);
    buf[u] = load_row<NT, STRIDE>(xy, ru, lane);
    if constexpr (Policy::Z) zbuf[u] = load_row_z<NT>(xy, ru, lane);
  }
  PoseU P;
  if (!get_pose(P)) return false;  // after the prologue loads are in flight (see stream_tiles)
  for (int base = 0; base < n; base += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int r = base + u;
      if (r >= n) break;
      const double dv = dbuf[u];
      const v2d v = buf[u];
      double z = 0.0;
      if constexpr (Policy::Z) z = zbuf[u];
      if (r + DEPTH < n) {
        dbuf[u] = load_desc_lane(desc, r + DEPTH, lane);
        buf[u] = load_row<NT, STRIDE>(xy, r + DEPTH, lane);
        if constexpr (Policy::Z) zbuf[u] = load_row_z<NT>(xy, r + DEPTH, lane);
      }
      const int count = __builtin_amdgcn_readlane(__double2loint(dv), 5);  // RowDesc: double 5 = {count, first}
      const int first = __builtin_amdgcn_readlane(__double2hiint(dv), 5);
      if (first != 0 || r == 0) {  // wave-uniform: the scan changes (or the wave's run begins inside one)
        if (r != 0) pol.flush(acc);
        pol.begin_scan(P, readlane_d(dv, 0), readlane_d(dv, 1), readlane_d(dv, 2), readlane_d(dv, 3), readlane_d(dv, 4));
      }
      if (lane < count) {  // (the running product is renormalised on every second row)
        if constexpr (Policy::Z) pol.point(v[0], v[1], z, (u & 1) != 0);
        else pol.point(v[0], v[1], (u & 1) != 0);
      }
    }
  }
  if (n > 0) pol.flush(acc);
  return true;
}

template <bool WITH_LOSS, bool NT, int DEPTH = ROWS_DEPTH, bool Z = false, class PoseFn>
__device__ __forceinline__ bool stream_rows(const double* __restrict__ xy_all, const RowDesc* __restrict__ desc_all,
                                            long long r_begin_in, long long r_end_in, const int lane, PoseFn get_pose,
                                            const double& inv_lf2, double (&acc)[NACC]) {
  if constexpr (Z) {
    LmRows3<WITH_LOSS> pol(inv_lf2);
    return stream_rows_policy<LmRows3<WITH_LOSS>, NT, DEPTH, ROW_DOUBLES_Z>(pol, xy_all, desc_all, r_begin_in, r_end_in, lane, get_pose, acc);
  } else {
    LmRows<WITH_LOSS> pol(inv_lf2);
    return stream_rows_policy<LmRows<WITH_LOSS>, NT, DEPTH>(pol, xy_all, desc_all, r_begin_in, r_end_in, lane, get_pose, acc);
  }
}

}  // namespace clc
