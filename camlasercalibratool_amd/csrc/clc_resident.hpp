// clc_resident.hpp — K4r: batched independent problems kept ON CHIP across their LM iterations.
//
// The reference solves every problem with one ceres::Solve (src/LaseCamCalCeres.cpp:301-307): the observations are
// read once into the ceres::Problem and every LM iteration works on them in place.  The lockstep batched path
// (clc_kernels.hpp K4) streams every still-running problem from HBM once per LM iteration instead — 5.6 passes over a
// C4 shard's 1.43 GB.  Here ONE workgroup owns ONE problem for its whole solve and reads the problem's scan points from
// HBM exactly once, into registers (PR points per lane) and LDS (PL points per lane); every evaluation pass of the
// solve then runs on chip, and so do the reduction and the wavefront LM controller (lm_advance_wave) between passes.
//
// "Lane layout" (built at upload, res_build_kernel): a problem is dealt to the NL = 64 NW lanes of its workgroup so
// that every lane holds points of ONE scan only — scan s with c_s points takes L_s = ceil(c_s / ppl) lanes with
// floor/ceil(c_s / L_s) points each, ppl = the smallest points-per-lane for which the problem's scans fit NL lanes —
// and is stored j-major: xyl[(row0 + j) * NL + lane] = j-th (x, y) of the lane, zero-padded to ppl.  So
//   * a lane's plane (n, d, scale; m = R^T n, c0 = n.t + d per pass) lives in ITS registers: no wave-uniform scan
//     bookkeeping, no scan-change branches, no descriptors in the point loop;
//   * the per-scan moment expansion (rows_flush, ~130 instructions) runs ONCE per lane per pass instead of once per
//     scan segment per wave (2.5-3.5 times per 20 rows in the row kernel);
//   * the point loop has no masks at all: every lane processes exactly ppl points; the zero padding (x = y = 0)
//     touches only S0, T0 and the cost product — by the same amount per padded point, w(c0), w(c0) c0 and
//     1 + c0^2/lf^2 — which is taken out analytically once per pass.
// Sums are taken in a different order than on the row layout, so results agree with the other batched paths to
// rounding (reduction tolerance 1e-11), not bit for bit; iteration counts and terminations are the same
// (tests/test_gpu_resident.py).
//
// Occupancy: two waves per SIMD.  The instantiations (abi_*.hip: kResPR256 / kResPL256 / kResPR512 / kResPL512):
//   <NW=4, PR=23, PL=19>: 256-thread workgroups, TWO problems resident per CU (256 lanes x 19 points x 16 B = 77.8 KB of LDS +
//     92 VGPRs of points each): while one problem's wave 0 runs the serial controller (~5 500 cycles on one SIMD), the other
//     problem's waves stream.  Capacity 256 lanes x 42 points.  The batched default.
//   <NW=8, PR=4, PL=18>: one 512-thread workgroup per CU (512 lanes x 18 x 16 B = 147.5 KB of LDS), capacity 512 lanes x 22
//     points = 11 264: batches with more than 256 scans per problem (or flag 8192), and the single-problem solve of clc_solve.
// CTRL selects the controller between passes: 0 = lm_advance_wave on the LDS state (clc_controller.hpp), 1 = the register-state
// controller of the cooperative kernel (clc_lmuni.hpp; bit-identical decisions and results, tests/test_gpu_lmuni.py).
#pragma once
#include "clc_kernels.hpp"
#include "clc_lmuni.hpp"

namespace clc {

struct ResLane {   // per lane of a problem's workgroup
  int32_t gid;     // scan (group table index: {n.x, n.y, n.z, d, scale, 0})
  int32_t cnt;     // valid points of the lane, <= ppl; 0: idle lane
};

// scans [g0, g1) that problem p = records [r0, r1) touches.  A batched problem never shares a scan with its neighbours
// (mark_problem_starts_kernel); the chunks of ONE problem dealt to the workgroups of the cooperative solve (clc_coop.hpp) may
// begin and end inside a scan: res_scan_extent() clamps a scan to the chunk, and the two parts are simply two lanes' worth of
// points of the same plane.
__device__ __forceinline__ void res_problem_scans(const long long* __restrict__ rec_off, const unsigned int* __restrict__ gid,
                                                  long long p, long long& r0, long long& r1, long long& g0, long long& g1) {
  r0 = rec_off[p];
  r1 = rec_off[p + 1];
  if (r1 <= r0) { g0 = g1 = 0; return; }
  g0 = gid[r0];
  g1 = (long long)gid[r1 - 1] + 1;
}
__device__ __forceinline__ void res_scan_extent(const long long* __restrict__ starts, long long g, long long r0, long long r1,
                                                long long& lo, long long& hi) {
  lo = starts[g];
  hi = starts[g + 1];
  lo = lo < r0 ? r0 : lo;
  hi = hi > r1 ? r1 : hi;
}

// One thread per problem: ppl[p] = the smallest points-per-lane (<= max_ppl) with sum_s ceil(c_s / ppl) <= n_lanes;
// 0 for an empty problem.  *fail is set when some problem does not fit.
static __global__ void res_plan_kernel(const long long* __restrict__ rec_off, const unsigned int* __restrict__ gid,
                                const long long* __restrict__ starts, long long n_problems, long long n, long long n_groups,
                                int n_lanes, int max_ppl, unsigned int* __restrict__ ppl_out, unsigned int* __restrict__ fail) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_problems) return;
  long long r0, r1, g0, g1;
  res_problem_scans(rec_off, gid, p, r0, r1, g0, g1);
  const long long n_p = r1 - r0;
  unsigned int ppl = 0;
  if (n_p > 0) {
    bool found = false;
    if (g1 - g0 <= n_lanes) {
      long long lo = (n_p + n_lanes - 1) / n_lanes;
      if (lo < 1) lo = 1;
      for (long long c = lo; c <= max_ppl && !found; ++c) {
        long long lanes = 0;
        for (long long g = g0; g < g1 && lanes <= n_lanes; ++g) {
          long long a, b;
          res_scan_extent(starts, g, r0, r1, a, b);
          lanes += (b - a + c - 1) / c;
        }
        if (lanes <= n_lanes) { ppl = (unsigned int)c; found = true; }
      }
    }
    if (!found) { atomicOr(fail, 1u); ppl = 0; }
  }
  ppl_out[p] = ppl;
}

// One workgroup of NL threads per problem: lane descriptors + the j-major point rows.
template <int NL>
__global__ __launch_bounds__(NL) void res_build_kernel(const double* __restrict__ aos, const long long* __restrict__ rec_off,
                                                       const unsigned int* __restrict__ gid, const long long* __restrict__ starts,
                                                       long long n, long long n_groups, const unsigned int* __restrict__ res_row,
                                                       ResLane* __restrict__ desc, double* __restrict__ xyl, double* __restrict__ zl = nullptr) {
  __shared__ int lane_first[NL + 1];
  const long long p = blockIdx.x;
  const int t = threadIdx.x;
  const unsigned int row0 = res_row[p];
  const int ppl = (int)(res_row[p + 1] - row0);
  long long r0, r1, g0, g1;
  res_problem_scans(rec_off, gid, p, r0, r1, g0, g1);
  const int ns = ppl > 0 ? (int)(g1 - g0) : 0;  // <= NL (res_plan_kernel)
  int L = 0;
  if (t < ns) {
    long long a, b;
    res_scan_extent(starts, g0 + t, r0, r1, a, b);
    L = (int)((b - a + ppl - 1) / ppl);
  }
  lane_first[t + 1] = L;
  if (t == 0) lane_first[0] = 0;
  __syncthreads();
  for (int off = 1; off < NL; off <<= 1) {  // inclusive scan of lane_first[1..NL]
    const int a = t >= off ? lane_first[t + 1 - off] : 0;
    __syncthreads();
    lane_first[t + 1] += a;
    __syncthreads();
  }
  const int total = lane_first[ns];
  ResLane dl;
  dl.gid = 0;
  dl.cnt = 0;
  long long k0 = 0;
  if (t < total) {
    int lo = 0, hi = ns;  // lane_first[lo] <= t < lane_first[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (lane_first[mid] <= t) lo = mid; else hi = mid;
    }
    const int i = t - lane_first[lo], Ls = lane_first[lo + 1] - lane_first[lo];
    long long s0, s1;
    res_scan_extent(starts, g0 + lo, r0, r1, s0, s1);
    const long long c = s1 - s0;
    const long long q = c / Ls, r = c % Ls;
    dl.gid = (int32_t)(g0 + lo);
    dl.cnt = (int32_t)(q + (i < r ? 1 : 0));
    k0 = s0 + (long long)i * q + (i < r ? i : r);
  }
  desc[(size_t)p * NL + t] = dl;
  v2d* out = reinterpret_cast<v2d*>(xyl) + (size_t)row0 * NL + t;
  for (int j = 0; j < ppl; ++j) {
    v2d v;
    v[0] = 0.0;
    v[1] = 0.0;
    if (j < dl.cnt) {
      const v2d* rec = reinterpret_cast<const v2d*>(aos + 8 * (k0 + j));
      v = rec[2];  // p.x, p.y
    }
    out[(size_t)j * NL] = v;
  }
  if (zl != nullptr) {  // points off the lidar plane: their z in a j-major array of its own, 8 bytes per slot (clc_coop.hpp, WITH_Z)
    double* outz = zl + (size_t)row0 * NL + t;
    for (int j = 0; j < ppl; ++j) outz[(size_t)j * NL] = j < dl.cnt ? aos[8 * (k0 + j) + 6] : 0.0;
  }
}

#ifdef CLC_STAMPS
// Debug build only (scripts/stamps_resident.py): stamps of the first RES_STAMP_WGS problems' waves 0 and 1 —
// slot 0: wall clock (100 MHz) at kernel entry, 1: wall clock at exit, 2: shader clock at entry, 3: after the loads landed;
// then per pass p < RES_STAMP_PASSES at 4 + 6 p: shader clock at pass start, after the point loop, after the wave
// reduction, after barrier 1, (wave 0) after the totals, after the controller / barrier 2.
constexpr int RES_STAMP_WGS = 1024, RES_STAMP_PASSES = 8, RES_STAMP_SLOTS = 4 + 6 * RES_STAMP_PASSES;
static __device__ long long clc_res_stamp_buf[RES_STAMP_WGS][2][RES_STAMP_SLOTS];
static __device__ unsigned long long clc_res_stamp_ctrl[RES_STAMP_WGS][16];  // lm_advance_wave's packed phase deltas of a workgroup's last pass
#ifndef CLC_RES_STAMP_BASE
#define CLC_RES_STAMP_BASE 0
#endif
#define RES_STAMP(slot, val)                                                                              \
  do {                                                                                                    \
    if (lane == 0 && (wave == cw || wave == ((cw + 1) & 3)) && blockIdx.x >= CLC_RES_STAMP_BASE &&        \
        blockIdx.x < CLC_RES_STAMP_BASE + RES_STAMP_WGS && (slot) < RES_STAMP_SLOTS)                      \
      clc_res_stamp_buf[blockIdx.x - CLC_RES_STAMP_BASE][wave == cw ? 0 : 1][slot] = (val);               \
  } while (0)
#else
#define RES_STAMP(slot, val) do {} while (0)
#endif

template <bool NT>
__device__ __forceinline__ v2d res_load(const v2d* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// The whole LM solve of one problem per workgroup, the problem resident in registers + LDS.
// uni_ppl >= 0: every problem of the batch has this many points per lane (row0 = problem index x uni_ppl, no offset read); -1: offsets from
// res_row; <= -2: multi-start — every workgroup holds problem 0's layout (-2 - uni_ppl points per lane) and solves from its own start pose.
// trace (nullable; single-problem launches only): the iteration records of clc_solve.
// rec_base, rec_host, rec_seg_off, rec_goal: see batched_write_record (clc_kernels.hpp) — used when summaries == nullptr.
// host_done (nullable): set to 1 in host memory, system-scope release, once the outcome of EVERY problem of the launch is written
// (done_count: device counter of finished workgroups, zero between launches; nullable for a single-workgroup launch) — the host
// polls it instead of blocking on the stream.
// CTRL: 1 = the wave-uniform controller with its state in the registers of wave 0 (clc_lmuni.hpp); 0 = lm_advance_wave on the LDS
// state (the round-2/3 controller: what tests/test_gpu_lmuni.py compares the other against, bit for bit — same pass, same totals).
// WITH_Z (8-wave form only): the points carry z (p.z != 0 somewhere in the batch) — 24-byte slots, a third register / LDS array per lane
// (`zl`: the z rows, j-major like the (x, y) rows), 14 moments per lane (rows3_*, clc_rows.hpp), padded slots masked (as clc_coop.hpp).
template <bool WITH_LOSS, bool NT, int NW, int PR, int PL, int CTRL, bool WITH_Z = false>
__global__ __launch_bounds__(NW * 64, 2) void resident_solve_kernel(
    const double* __restrict__ xyl, const unsigned int* __restrict__ res_row, const ResLane* __restrict__ lane_desc,
    const double* __restrict__ groups, const int uni_ppl, const clc_options opt, clc_iteration* __restrict__ trace,
    const int trace_cap, double* __restrict__ poses, clc_summary* __restrict__ summaries, double* __restrict__ results,
    int32_t* __restrict__ host_done, unsigned int* __restrict__ done_count, const double rec_base = 0.0,
    double* __restrict__ rec_host = nullptr, const long long rec_seg_off = 0, const unsigned long long rec_goal = 0,
    const double* __restrict__ zl = nullptr) {
  static_assert(!WITH_Z || NW == 8, "points with z: the 8-wave form (one problem per CU)");
  constexpr bool RES_LEAN = NW == 4 || WITH_Z;  // the controller next to 92 (60) VGPRs of points: the small-footprint form (clc_controller.hpp)
  // How the pass gets its pose: CTRL 0 — x_eval from the LDS state, every wave turning the quaternion into a wave-uniform (SGPR)
  // rotation itself; CTRL 1 — as the cooperative kernel: rotation + translation + status published by the controller (six 16-byte
  // LDS reads).  Same quat_to_rot on the same quaternion either way: the passes of the two forms are bit-identical.
  constexpr bool POSE_PUB = CTRL != 0;
#ifndef CLC_RES_PAD_ANALYTIC
#define CLC_RES_PAD_ANALYTIC 1
#endif
  // Padded slots: every lane runs its ppl slots unmasked and the contribution of its zero padding — the same per padded slot — is taken
  // out analytically once per pass (a logarithm + a reciprocal per lane).  The alternative of clc_coop.hpp (CLC_RES_PAD_ANALYTIC = 0:
  // masked slots in the groups that some lane has padding in) was measured here and lost: the wave-uniform branch per group and the
  // second copy of the point code cost the 4-wave form 12 % (C4 shard kernel 0.907 against 0.808 ms, C3 0.141 / 0.124, C1 0.143 / 0.134).
  constexpr bool PAD_ANALYTIC = CLC_RES_PAD_ANALYTIC != 0 && !WITH_Z;  // (the analytic correction is written for (x, y) slots)
  using Moments = typename std::conditional<WITH_Z, RowMoments3, RowMoments>::type;
  constexpr int NL = NW * 64;
  constexpr int NP = PR + PL;  // points a lane can hold
  constexpr int CH = 6;        // LDS points are read in chunks of CH, one chunk ahead of the arithmetic
  constexpr int NCH = (PL + CH - 1) / CH;
  __shared__ v2d sh_pts[(PL > 0 ? PL : 1) * NL];
  __shared__ double sh_ptz[WITH_Z ? PL * NL : 1];  // the z of the LDS-held slots
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ __attribute__((aligned(16))) double sh_tot[64];  // CTRL 1: two buffers of 32, the totals of the passes alternate (clc_lmuni.hpp)
  __shared__ double sh_wsum[NW][NACC];
  __shared__ double sh_park[32 + (sizeof(LmScratch) + 7) / 8];
  __shared__ __attribute__((aligned(16))) double sh_pub[LM_PUB_WORDS];
  const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the iteration trace exists for the single-problem (8-wave) launches only: in the 4-wave form the controller shares the
  // register file with 92 VGPRs of points, and the trace record's code must not be there at all
  clc_iteration* const tr = (NW == 8 && !WITH_Z) ? trace : nullptr;
  const int tr_cap = (NW == 8 && !WITH_Z) ? trace_cap : 0;
  // The controller runs on wave 0.  (Picking wave 0 / wave 2 by the workgroup's LDS allocation base — s_getreg HW_REG_LDS_ALLOC — so
  // that the two workgroups of a CU never run their controllers on the same SIMD was measured: no gain, and the run-time wave
  // index cost a live SGPR in a kernel that already spills scalars — a -DCLC_STAMPS build of that form computed wrong steps.)
  constexpr int cw = 0;
  RES_STAMP(0, wall_clock64());
  RES_STAMP(2, clock64());
  LmState& st = *reinterpret_cast<LmState*>(sh_state);
  // start pose: host memory behind PCIe (~2 us) — requested first, consumed after the points have landed
  const double pose_w = poses[7 * (size_t)prob + (tid < 7 ? tid : 0)];
  unsigned int row0;
  int ppl;
  int lay = prob;  // which problem's lane layout this workgroup holds
  if (uni_ppl >= 0) {
    row0 = (unsigned int)prob * (unsigned int)uni_ppl;
    ppl = uni_ppl;
  } else if (uni_ppl <= -2) {
    // multi-start on shared observations (clc_solve_multistart): every workgroup holds problem 0's points — ONE copy in HBM, served by
    // L2 after the first round of workgroups — and solves from ITS OWN start pose; ppl = -2 - uni_ppl
    row0 = 0u;
    ppl = -2 - uni_ppl;
    lay = 0;
  } else {
    row0 = res_row[prob];
    ppl = __builtin_amdgcn_readfirstlane((int)(res_row[prob + 1] - row0));
  }
  // ---- the problem's points: HBM -> registers / LDS, once.  Unconditional loads from clamped row indices (the array
  // carries one padding row): a load inside `if (j < ppl)` sits in its own branch with its own wait at the join.  Slots
  // beyond ppl become zeros (the pass may run one slot past ppl).
  const v2d* __restrict__ src = reinterpret_cast<const v2d*>(xyl) + (size_t)row0 * NL + tid;
  const int j_last = ppl > 0 ? ppl - 1 : 0;
  // Order: the LDS-bound rows first, then the register rows — loads return in order, so the LDS part is written (and the
  // barrier in front of the first pass passed) while the register rows are still in flight; the first pass then consumes
  // them as they arrive.  (A lane reads back only its own LDS slots: no barrier is needed for sh_pts.)
  const double* __restrict__ srcz = WITH_Z ? zl + (size_t)row0 * NL + tid : nullptr;
  v2d lds_v[PL > 0 ? PL : 1];
  double lds_z[WITH_Z ? PL : 1];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int j = PR + i;
    lds_v[i] = res_load<NT>(src + (size_t)(j < j_last ? j : j_last) * NL);
    if (WITH_Z) lds_z[i] = srcz[(size_t)(j < j_last ? j : j_last) * NL];
  }
  v2d reg[PR];
  double regz[WITH_Z ? PR : 1];
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    reg[j] = res_load<NT>(src + (size_t)(j < j_last ? j : j_last) * NL);
    if (WITH_Z) regz[j] = srcz[(size_t)(j < j_last ? j : j_last) * NL];
  }
  const ResLane dl = lane_desc[(size_t)lay * NL + tid];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    v2d v = lds_v[i];
    double vz = WITH_Z ? lds_z[i] : 0.0;
    if (PR + i >= ppl) { v[0] = 0.0; v[1] = 0.0; vz = 0.0; }
    sh_pts[i * NL + tid] = v;
    if (WITH_Z) sh_ptz[i * NL + tid] = vz;
  }
  // The lane's plane is fetched again in every pass (48 bytes per lane out of L1/L2: a problem's group entries are ~1 KB)
  // rather than held in 10 VGPRs across the controller, which needs every register it can get.
  const double* __restrict__ gp = groups + (size_t)dl.gid * GROUP_DOUBLES;
  const int cnt = dl.cnt;
  if (tid < 7) sh_park[tid] = pose_w;
  const double inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
  // points processed per lane and pass: ppl rounded up to whole groups (GRP points per basic block: independent
  // dependency chains for a wave that has its SIMD to itself while the co-resident problem is in its controller)
#ifndef CLC_RES_GROUP
#define CLC_RES_GROUP 3
#endif
  // Points per basic block (the running product is renormalised on the last point of a group: three factors below 2^341 each
  // cannot overflow).  Three independent dependency chains per block in the 4-wave form — a wave often has its SIMD to itself
  // there, while the co-resident problem is in its controller: C4 shard 0.95 -> 0.92 ms against pairs, groups of six 0.91
  // (scripts/r03_resident.py); pairs in the 8-wave form, which never runs alone on a SIMD (and spills with triples).
  constexpr int GRP = NW == 4 ? CLC_RES_GROUP : 2;
  const int ppl_up = (ppl + GRP - 1) / GRP * GRP;
  const int ppl_eff = ppl_up < NP ? ppl_up : NP;
  // Padding (as clc_coop.hpp): in the blocks that some lane of the wave has padded slots in, a padded slot is evaluated with
  // r0 = 0 (cost factor exactly 1) and weight 0 (rows_point_masked): no moment moves.
  const int cnt_m = cnt > 0 ? cnt : ppl_eff;
  int cmin = cnt_m;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o2 = __shfl_xor(cmin, off, 64);
    cmin = o2 < cmin ? o2 : cmin;
  }
  cmin = __builtin_amdgcn_readfirstlane(cmin);
  LmU F;
  if (wave == cw) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (CTRL) {
      double x0[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) x0[i] = sh_park[i];
      lmu_init(F, st, opt, x0, lane);
      lmu_publish(sh_pub, x0, CLC_RUNNING, lane);
    } else if (tid == 0) {
      lm_init(st, opt, sh_park);
    }
  }
  __syncthreads();
  RES_STAMP(3, clock64());
#pragma unroll
  for (int j = 0; j < PR; ++j)
    if (j >= ppl) { reg[j][0] = 0.0; reg[j][1] = 0.0; if (WITH_Z) regz[j] = 0.0; }
  int pass_no = 0;  // (debug stamps)

  // one evaluation pass at the published pose: the wave's 28 totals -> sh_wsum[wave]
  auto pass = [&]() {
    RES_STAMP(4 + 6 * pass_no, clock64());
    int t = tid;
    asm volatile("" : "+v"(t));  // (opaque: the pass's LDS addresses are recomputed here, not hoisted out of the loop and spilled)
    // plane of the lane's scan (idle lanes: zeros, scale 0 — their moments are finite and expand to nothing)
    double nx, ny, nz, pd, ps;
    {
      glb_cdouble* g2 = global_opaque(gp);  // (an address the compiler cannot prove loop-invariant: the loads stay in the pass, as global loads)
      const v2d a = *reinterpret_cast<glb_cv2d*>(g2);
      const v2d b = *reinterpret_cast<glb_cv2d*>(g2 + 2);
      const double s5 = g2[4];
      const bool on = cnt > 0;
      nx = on ? a[0] : 0.0; ny = on ? a[1] : 0.0; nz = on ? b[0] : 0.0; pd = on ? b[1] : 0.0; ps = on ? s5 : 0.0;
    }
    v2d buf[2][CH];  // LDS points, chunk c in buf[c & 1]; chunk 0 is read before the register points are consumed
    double bufz[2][WITH_Z ? CH : 1];
    if (PL > 0) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (u < PL) {
          buf[0][u] = sh_pts[u * NL + t];
          if (WITH_Z) bufz[0][u] = sh_ptz[u * NL + t];
        }
    }
    RowPlane q;
    if (POSE_PUB) {
      const lds_cv2d* pb = lds_opaque(sh_pub);
      const v2d p0 = pb[0], p1 = pb[1], p2 = pb[2], p3 = pb[3], p4 = pb[4], p5 = pb[5];  // 6 broadcast 16-byte reads
      const double Rm[9] = {p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1], p4[0]};
      const double tv[3] = {p4[1], p5[0], p5[1]};
      rows_plane_setup(Rm, tv, nx, ny, nz, pd, ps, q);
    } else {
      PoseU P;
      double x[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) x[i] = st.x_eval[i];
      load_pose(x, P);
      rows_plane_setup(P.R, P.t, nx, ny, nz, pd, ps, q);
    }
#ifdef CLC_STAMPS
    if (wave != cw) { asm volatile("" :: "v"(q.mx), "v"(q.c0)); RES_STAMP(8 + 6 * pass_no, clock64()); }  // plane + pose have arrived
#endif
    Moments M;
    lane_moments_reset<WITH_LOSS>(M);
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += GRP) {
      if (j0 < ppl) {  // wave-uniform; the points of a group in one basic block
        if (PAD_ANALYTIC || j0 + GRP <= cmin) {  // wave-uniform: no lane of the wave has padding in this group
#pragma unroll
          for (int j = j0; j < j0 + GRP && j < NP; ++j) {
            if (j >= PR && (j - PR) % CH == 0 && (j - PR) / CH + 1 < NCH) {  // entering an LDS chunk: request the next one
              const int c1 = (j - PR) / CH + 1;
#pragma unroll
              for (int u = 0; u < CH; ++u)
                if (c1 * CH + u < PL) {
                  buf[c1 & 1][u] = sh_pts[(c1 * CH + u) * NL + t];
                  if (WITH_Z) bufz[c1 & 1][u] = sh_ptz[(c1 * CH + u) * NL + t];
                }
            }
            const v2d v = j < PR ? reg[j < PR ? j : 0] : buf[((j - PR) / CH) & 1][(j - PR) % CH];
            const double vz = !WITH_Z ? 0.0 : (j < PR ? regz[j < PR ? j : 0] : bufz[((j - PR) / CH) & 1][(j - PR) % CH]);
            lane_point<WITH_LOSS>(q, inv_lf2, v[0], v[1], vz, M, /*renorm=*/j == j0 + GRP - 1);  // (rows_flush normalises an incomplete last group)
          }
        } else {
#pragma unroll
          for (int j = j0; j < j0 + GRP && j < NP; ++j) {
            if (j >= PR && (j - PR) % CH == 0 && (j - PR) / CH + 1 < NCH) {
              const int c1 = (j - PR) / CH + 1;
#pragma unroll
              for (int u = 0; u < CH; ++u)
                if (c1 * CH + u < PL) {
                  buf[c1 & 1][u] = sh_pts[(c1 * CH + u) * NL + t];
                  if (WITH_Z) bufz[c1 & 1][u] = sh_ptz[(c1 * CH + u) * NL + t];
                }
            }
            const v2d v = j < PR ? reg[j < PR ? j : 0] : buf[((j - PR) / CH) & 1][(j - PR) % CH];
            const double vz = !WITH_Z ? 0.0 : (j < PR ? regz[j < PR ? j : 0] : bufz[((j - PR) / CH) & 1][(j - PR) % CH]);
            lane_point_masked<WITH_LOSS>(q, j < cnt_m, inv_lf2, v[0], v[1], vz, M, /*renorm=*/j == j0 + GRP - 1);
          }
        }
      }
    }
    double lp = 0.0;
    if (PAD_ANALYTIC) {  // the zero padding out again: npad points (0, 0) with r0 = c0 each
      const int npad = ppl_eff - cnt;
      const double np = (double)npad;
      const double c0 = q.c0;
      if (WITH_LOSS) {
        const double sum_p = fma(c0 * c0, inv_lf2, 1.0);
        const double w_p = rcp_ge1(sum_p);
        const double cs = npad > 0 ? np * w_p : 0.0;
        M.S0 -= cs;
        M.T0 = fma(-cs, c0, M.T0);
        int e;
        const double m = frexp_pos(sum_p, e);
        lp = npad > 0 ? np * log_mant_exp(m, e) : 0.0;
      } else {
        const double cs = npad > 0 ? np : 0.0;
        M.S0 -= cs;
        M.T0 = fma(-cs, c0, M.T0);
        M.prod = fma(-cs * c0, c0, M.prod);
      }
    } else {
      lane_pad_correction<WITH_LOSS>(M, (double)(ppl_eff - cnt_m));
    }
    RES_STAMP(5 + 6 * pass_no, clock64());
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    lane_flush<WITH_LOSS>(q, M, acc);
    if (PAD_ANALYTIC && WITH_LOSS) acc[27] = fma(-q.s2, lp, acc[27]);
    wave_reduce_butterfly(acc, sh_wsum[wave], lane);
    RES_STAMP(6 + 6 * pass_no, clock64());
  };
  // totals of the pass (fixed order) + the controller, on wave 0; the other waves leave from its barrier
  auto totals = [&]() {
    if (lane < NACC) {
      double s = sh_wsum[0][lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += sh_wsum[w][lane];
      sh_tot[(CTRL ? 32 * (1 - F.hx) : 0) + lane] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int cap = opt.max_num_iterations + 2;
  if (CTRL) {
    // one loop, the first pass told apart by a run-time flag (clc_lmuni.hpp: two inlined copies of the controller in one kernel cost the
    // register allocator ~100 doubles of spills)
    for (int k = 0; k <= cap && lm_pub_status(sh_pub) == CLC_RUNNING; ++k) {  // (status: published before the barrier)
      pass();
      __syncthreads();
      RES_STAMP(7 + 6 * pass_no, clock64());
      if (wave == cw) {
        totals();
        RES_STAMP(8 + 6 * pass_no, clock64());
        __builtin_amdgcn_s_setprio(3);  // the controller wins the issue arbitration against the co-resident problem's streaming wave
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        if (k > 0) lmu_pre(F, st, opt, sh_tot, tr, tr_cap, lane_c);
        lmu_post(k == 0, F, opt, sh_tot, sh_pub, tr, tr_cap, lane_c);
        __syncthreads();  // the other waves leave for their pass
        __builtin_amdgcn_s_setprio(0);
      } else {
        __syncthreads();
      }
      RES_STAMP(9 + 6 * pass_no, clock64());
      ++pass_no;
    }
  } else {
    // (the first pass is peeled: with both controller instantiations inside one loop the kernel spilled)
    pass();
    __syncthreads();
    RES_STAMP(7, clock64());
    if (wave == cw) {
      totals();
      RES_STAMP(8, clock64());
      __builtin_amdgcn_s_setprio(3);  // the serial controller wins the issue arbitration against the co-resident problem's streaming wave
      int lane_c = lane;
      asm volatile("" : "+v"(lane_c));  // (opaque: the controller's per-lane LDS addresses are recomputed here, not hoisted out of the pass loop and spilled)
      lm_advance_wave<true, RES_LEAN>(st, opt, tr, tr_cap, sh_tot, sh_park, lane_c);  // contains the barrier ...
      __builtin_amdgcn_s_setprio(0);
    } else {
      __syncthreads();  // ... the other waves meet here
    }
    RES_STAMP(9, clock64());
    pass_no = 1;
    for (int k = 0; k < cap && st.status == CLC_RUNNING; ++k) {  // (status: published before the barrier)
      pass();
      __syncthreads();
      RES_STAMP(7 + 6 * pass_no, clock64());
      if (wave == cw) {
        totals();
        RES_STAMP(8 + 6 * pass_no, clock64());
        __builtin_amdgcn_s_setprio(3);
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
#if defined(CLC_STAMPS) && !defined(CLC_RES_NO_CROW)
        // the controller's own phase stamps (CLC_CK in lm_advance_wave: packed 16-bit cycle deltas into words 15 and 6 of the row)
        unsigned long long* crow = (blockIdx.x >= CLC_RES_STAMP_BASE && blockIdx.x < CLC_RES_STAMP_BASE + RES_STAMP_WGS)
                                       ? reinterpret_cast<unsigned long long*>(&clc_res_stamp_ctrl[blockIdx.x - CLC_RES_STAMP_BASE][0]) : nullptr;
        lm_advance_wave<false, RES_LEAN>(st, opt, tr, tr_cap, sh_tot, sh_park, lane_c, crow);
#else
        lm_advance_wave<false, RES_LEAN>(st, opt, tr, tr_cap, sh_tot, sh_park, lane_c);
#endif
        __builtin_amdgcn_s_setprio(0);
      } else {
        __syncthreads();
      }
      RES_STAMP(9 + 6 * pass_no, clock64());
      ++pass_no;
    }
  }
  RES_STAMP(1, wall_clock64());
  if (wave == cw) {
    if (CTRL) {
      if (F.status == CLC_RUNNING) F.status = CLC_FAILURE;  // unreachable: the controller stops at the iteration cap
      lmu_finish(F, st, opt, sh_tot, tr, tr_cap, lane);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) {
      if (st.status == CLC_RUNNING) st.status = CLC_FAILURE;  // unreachable: the controller stops at the iteration cap
      // summaries == nullptr: the records-only form (clc_solve_batched_gather) — `results` / `rec_host` are the communicator's gather
      // buffer and its pinned host twin: the problem's result record, global index rec_base + prob, goes straight into this rank's
      // segment of both, the shard's totals into their leading record; no pose, no summary
      if (summaries != nullptr) batched_write_outcome(st, prob, poses, summaries, results);
      else batched_write_record(st, prob, results, rec_host, rec_seg_off, rec_base, rec_goal);
      if (host_done != nullptr) {
        bool last = true;
        if (done_count != nullptr) {
          // this workgroup's outcome is complete in host / device memory before it counts itself in (system-scope release);
          // the last one to arrive has therefore seen everybody's, resets the counter for the next launch and raises the flag
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          const unsigned int n = __hip_atomic_fetch_add(done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
          last = n == gridDim.x - 1;
          if (last) __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(host_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

}  // namespace clc
