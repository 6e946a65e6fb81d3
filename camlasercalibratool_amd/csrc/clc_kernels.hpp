// clc_kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the point-to-plane
// extrinsic path.  Included by clc_abi.hip only.
//
// Data layout in HBM ("tiled records"): the observation array handed over the C-ABI is an
// array of 64-byte records {n[3], d, p[3], scale} (clc_observation).  On upload it is
// re-tiled once into AoSoA tiles of TILE=128 records:
//     tile t, field f (0..7), slot j (0..127)  ->  tiles[t*1024 + f*128 + j]
// so that a wavefront reads one field of one tile with ONE fully coalesced 1-KiB
// `global_load_dwordx4` (lane l gets slots 2l, 2l+1 as a double2).  A wave consumes a tile
// per loop trip (8 such loads, 2 observations per lane) and owns a contiguous run of tiles.  Every observation is read exactly
// once per evaluation pass: 64 algorithmic bytes per residual+Jacobian evaluation.
//
// Kernels
//   retile_kernel, retile_batched_kernel      AoS records -> 64-byte tiles (once per upload)
//   group_flag/build_groups/build_ctiles      upload-time lossless re-encoding into the compact layout
//   eval_kernel            K1: per-observation residual + analytic 6-DoF Jacobian (a3), Cauchy
//                          corrector (a4), rank-1 accumulation of {H(21), g(6), cost} in 28 FP64
//                          registers per lane, wave butterfly reduction (permlane32/16 swap + DPP),
//                          LDS-staged per-wave partials, one 28-double partial per workgroup;
//                          template variants: loss / Jacobian / prefetch / nt loads / compact layout
//   lm_kernel              K2: fixed-order reduction of the block partials + the LM controller
//                          (clc_lm.hpp) for the single-problem solve; publishes to a pinned mailbox
//   eval_lm_kernel         K1+K2 in one launch (controller in the last-arriving workgroup), optional
//   reduce_kernel          fixed-order reduction of block partials for clc_eval
//   batched_*_kernel       K4: independent problems in lockstep (eval + controller per iteration)
//   normal9_kernel         K5: 9x9 normal equation of the closed-form initialiser
//   line_fit_kernel        K6: LineFittingCeres, one wavefront per scan, LM loop in-kernel
//   factor_kernel, plus_kernel                plug-in level parity with the Ceres callbacks
#pragma once
#include <hip/hip_runtime.h>

#include "clc_lm.hpp"
#include "clc_math.hpp"
#include "clc_rows.hpp"

namespace clc {

constexpr int TILE = 128;               // records per tile
constexpr int TILE_DOUBLES = TILE * 8;  // 1024 doubles = 8 KiB
constexpr int NACC = 28;                // 21 (H upper triangle) + 6 (g) + 1 (cost)
constexpr int BLOCK = 256;              // threads per workgroup (4 waves)

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double make_uniform(double v) {
  // value is wave-uniform: move it to SGPRs so it costs no VGPRs in the streaming loop
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

struct Pose7 {  // a pose passed by value as a kernel argument
  double v[7];
};

struct PoseU {  // wave-uniform pose: rotation matrix (row-major) + translation
  double R[9];
  double t[3];
};

__device__ __forceinline__ void load_pose(const double* __restrict__ pose, PoseU& P) {
  double x[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) x[i] = pose[i];
  double R[9];
  quat_to_rot(x + 3, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) P.R[i] = make_uniform(R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) P.t[i] = make_uniform(x[i]);
}

// Reciprocal to ~1 ulp: v_rcp_f64 seed + two Newton steps (no IEEE corner cases needed: the
// argument is a finite number >= 1).
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

// log(x) for finite x >= 1 (the Cauchy argument 1 + r^2/a^2): fdlibm-style reduction
// x = 2^e * m, m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f/(2+f),
// log(m) = f - s*(f - R(s^2)) with the 7-term minimax R, log(x) = e*ln2 + log(m).  < 1 ulp.
__device__ __forceinline__ double log_ge1(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * fast_rcp(2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  // log(x) = dk*ln2_hi - ((hfsq - (s*(hfsq+R) + dk*ln2_lo)) - f)
  return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// One observation (record {n,d,p,s}), pose (R,t):
//   r0 = n.(R p + t) + d, residual r = s r0                          (LaseCamCalCeres.cpp:47-48)
//   u  = [n, p x R^T n]  : Jacobian row J = s u (tangent space of Plus)            (:56-57)
//   Cauchy a = lf*s (:249): rho0 = a^2 log(1 + r^2/a^2), rho1 = 1/(1 + r^2/a^2).  The scale
//   cancels inside the loss argument: r^2/a^2 = r0^2/lf^2, so sum = 1 + r0^2/lf^2 needs no
//   per-observation division.
//   Corrector (rho'' <= 0 branch): J~ = sqrt(rho1) J, r~ = sqrt(rho1) r, hence
//     H += k u u^T, g += k r0 u  with k = rho1 s^2 ;  cost += s^2 log(sum)  (x lf^2/2 at the end)
//   Without loss: k = s^2, cost += s^2 r0^2 (x 1/2 at the end).
// acc[27] therefore holds sum s^2*log(sum) (or sum r^2); finalize_cost() applies the factor.
template <bool WITH_LOSS, bool WITH_JAC>
__device__ __forceinline__ void accumulate_observation(const PoseU& P, const double inv_lf2,
                                                       const double nx, const double ny,
                                                       const double nz, const double d,
                                                       const double px, const double py,
                                                       const double pz, const double s,
                                                       double (&acc)[NACC]) {
  // m = R^T n
  const double mx = fma(P.R[6], nz, fma(P.R[3], ny, P.R[0] * nx));
  const double my = fma(P.R[7], nz, fma(P.R[4], ny, P.R[1] * nx));
  const double mz = fma(P.R[8], nz, fma(P.R[5], ny, P.R[2] * nx));
  // n.(R p + t) + d = m.p + (n.t + d)
  const double c0 = fma(P.t[2], nz, fma(P.t[1], ny, fma(P.t[0], nx, d)));
  const double r0 = fma(mz, pz, fma(my, py, fma(mx, px, c0)));
  const double s2 = s * s;
  double k = s2;
  if (WITH_LOSS) {
    const double sum = fma(r0 * r0, inv_lf2, 1.0);
    acc[27] = fma(s2, log_ge1(sum), acc[27]);
    k = s2 * fmax(2.2250738585072014e-308, fast_rcp(sum));
  } else {
    acc[27] = fma(s2 * r0, r0, acc[27]);
  }
  if (WITH_JAC) {
    double u[6], ku[6];
    u[0] = nx;
    u[1] = ny;
    u[2] = nz;
    u[3] = fma(py, mz, -(pz * my));
    u[4] = fma(pz, mx, -(px * mz));
    u[5] = fma(px, my, -(py * mx));
#pragma unroll
    for (int a = 0; a < 6; ++a) ku[a] = k * u[a];
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) {
        acc[idx] = fma(ku[a], u[b], acc[idx]);
        ++idx;
      }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] = fma(ku[a], r0, acc[21 + a]);
  }
}

// cost = 1/2 sum rho: acc[27] holds sum s^2 log(sum) (loss) or sum r^2 (no loss).
__device__ __forceinline__ double finalize_cost(double acc27, bool with_loss, double lf) {
  return with_loss ? 0.5 * (lf * lf) * acc27 : 0.5 * acc27;
}

// ---------------------------------------------------------------------------------------
// wavefront reduction of 28 FP64 accumulators
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_read(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// x.lanes[32..63] <-> y.lanes[0..31]
__device__ __forceinline__ void swap_halves(double& x, double& y) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]);
  y = __hiloint2double((int)hi[1], (int)lo[1]);
}

// x.rows{1,3} <-> y.rows{0,2}   (rows of 16 lanes)
__device__ __forceinline__ void swap_rows(double& x, double& y) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]);
  y = __hiloint2double((int)hi[1], (int)lo[1]);
}

// Butterfly (reduce-scatter) wave reduction: 28 -> 14 registers with v_permlane32_swap,
// 14 -> 7 with v_permlane16_swap, then a 4-step DPP all-reduce inside each 16-lane row.
// 147 cross-lane/add instructions instead of 28*6*3 for 28 independent shuffles.
// Result: wave total of acc[i + 7*rho] is in register i of every lane of row rho; the
// lanes with (lane & 15) == 0 store it to out[i + 7*rho].
__device__ __forceinline__ void wave_reduce_butterfly(double (&acc)[NACC], double* out, int lane) {
  double r[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    double x = acc[i], y = acc[i + 14];
    swap_halves(x, y);
    r[i] = x + y;  // lanes 0-31: acc[i] over {l, l+32}; lanes 32-63: acc[i+14]
  }
  double u[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    double x = r[i], y = r[i + 7];
    swap_rows(x, y);
    u[i] = x + y;  // row rho holds partial sums of acc[i + 7*rho]
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    double v = u[i];
    v += dpp_read<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_read<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_read<0x141>(v);  // row_half_mirror
    v += dpp_read<0x140>(v);  // row_mirror
    u[i] = v;
  }
  if ((lane & 15) == 0) {
    const int rho = lane >> 4;
#pragma unroll
    for (int i = 0; i < 7; ++i) out[i + 7 * rho] = u[i];
  }
}

// Reference reduction: 28 independent xor-shuffles (kept for A/B and as the checker of the
// butterfly in tests).
__device__ __forceinline__ void wave_reduce_shuffle(double (&acc)[NACC], double* out, int lane) {
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) out[k] = v;
  }
}

// Block-level: per-wave totals staged in LDS, summed in wave order by the first 28 threads.
template <int NWAVES>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], int reduce_mode,
                                                   double* __restrict__ out28) {
  __shared__ double wsum[NWAVES][NACC];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if ((reduce_mode & 1) == 0)
    wave_reduce_butterfly(acc, wsum[wave], lane);
  else
    wave_reduce_shuffle(acc, wsum[wave], lane);
  __syncthreads();
  if (threadIdx.x < NACC) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s += wsum[w][threadIdx.x];
    out28[threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------
// retile: AoS records -> tiles.  One thread per record (one-time cost per upload).
// ---------------------------------------------------------------------------------------
__global__ void retile_kernel(const double* __restrict__ aos, double* __restrict__ tiles,
                              long long n, long long n_padded) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_padded) return;
  const long long t = k / TILE;
  const int j = (int)(k % TILE);
  double v[8];
  if (k < n) {
    const double2* src = reinterpret_cast<const double2*>(aos + 8 * k);
    const double2 a = src[0], b = src[1], c = src[2], d = src[3];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
  } else {
#pragma unroll
    for (int f = 0; f < 8; ++f) v[f] = 0.0;
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) tiles[t * TILE_DOUBLES + f * TILE + j] = v[f];
}

// Batched variant: one workgroup per problem; problem k's records [rec_off[k], rec_off[k+1])
// go to its own whole tiles starting at tile_off[k].
__global__ void retile_batched_kernel(const double* __restrict__ aos,
                                      const long long* __restrict__ rec_off,
                                      const long long* __restrict__ tile_off,
                                      double* __restrict__ tiles) {
  const int prob = blockIdx.x;
  const long long r0 = rec_off[prob];
  const long long n = rec_off[prob + 1] - r0;
  const long long n_padded = (tile_off[prob + 1] - tile_off[prob]) * TILE;
  double* tb = tiles + tile_off[prob] * TILE_DOUBLES;
  for (long long k = threadIdx.x; k < n_padded; k += blockDim.x) {
    double v[8];
    if (k < n) {
      const double2* src = reinterpret_cast<const double2*>(aos + 8 * (r0 + k));
      const double2 a = src[0], b = src[1], c = src[2], d = src[3];
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
    } else {
#pragma unroll
      for (int f = 0; f < 8; ++f) v[f] = 0.0;
    }
    const long long t = k / TILE;
    const int j = (int)(k % TILE);
#pragma unroll
    for (int f = 0; f < 8; ++f) tb[t * TILE_DOUBLES + f * TILE + j] = v[f];
  }
}

// ---------------------------------------------------------------------------------------
// K1 — fused evaluation + reduction.
// grid: any number of 256-thread blocks; waves stride over tiles.  Output: one 28-double
// partial per block (deterministic: fixed lane->record map, fixed reduction shape).
// status (nullable): device-side termination flag of the LM controller; a finished solve
// turns the remaining enqueued launches into no-ops.
// ---------------------------------------------------------------------------------------
// launch flags (clc_set_launch)
constexpr int FLAG_REDUCE_SHUFFLE = 1;  // reference wave reduction instead of the butterfly
constexpr int FLAG_PREFETCH = 2;        // software-pipelined tile loads (next tile in flight while computing)
constexpr int FLAG_NONTEMPORAL = 4;     // nt loads for the streamed tiles
constexpr int FLAG_FUSED_LM = 8;        // clc_solve: controller in the tail of the evaluation launch
constexpr int FLAG_DEEP = 64;            // compact layout: two tiles of points in flight per wave (HBM-resident arrays)
constexpr int FLAG_WG512 = 32;          // 512-thread workgroups with the 3:2 old/young wave tile weighting
constexpr int FLAG_STEP = 128;          // clc_solve: one step_kernel launch per LM iteration (compact or row layout)
constexpr int FLAG_ROWS = 256;          // row layout (clc_rows.hpp): 16 B/observation + 64 B/row, per-scan moments
constexpr int FLAG_EQUAL_WAVES = 512;        // row layout, 512-thread workgroups: equal shares per wave, cut at scan starts, instead of the 3:2 old/young weighting
constexpr int FLAG_BATCHED_WG256 = 1024;     // batched row kernel: 256-thread workgroups + block reduction instead of one wave per workgroup
constexpr int FLAG_BATCHED_LOCKSTEP = 2048;  // one-workgroup-per-problem batches: lockstep launches instead of batched_solve_kernel
constexpr int FLAG_NO_RESIDENT = 4096;       // batched solver: not the on-chip resident kernel (clc_resident.hpp) even where the problems fit
constexpr int FLAG_RESIDENT_WG512 = 8192;    // resident layout over 512 lanes per problem (one workgroup per CU) even where 256 lanes hold it

typedef double v2d __attribute__((ext_vector_type(2)));

// Static tile -> wave map, two levels, no division by a run-time weight total.  The T tiles of the array are dealt to the
// workgroups as evenly as integers allow (workgroup b gets q or q + 1 consecutive tiles, q = T / n_blocks, the first
// T % n_blocks workgroups the extra one); inside a workgroup the waves take consecutive sub-runs proportional to their
// weights.  In a 512-thread workgroup the four first-launched waves (one per SIMD) win the issue arbitration against
// the four younger ones sharing their SIMDs (measured: 8.0k vs 11.5k cycles for equal work, 99 % repeatable), so the
// older slots get W_OLD = 3 and the younger W_YOUNG = 2 units: both finish together instead of leaving the tail of the
// launch at half occupancy.  The map is a pure function of (array length, grid, block size), so the summation order —
// and the result, bit for bit — stays fixed.  (An earlier single-level form, floor(T * unit / total_units), cost two
// 64-bit or three 32-bit per-lane divisions per wave in front of its first load, ~250 instructions in each of 2 048
// waves; here the one division is wave-uniform and runs on the scalar unit.  No measurable change in solve time — in
// the step kernel the controller chain hides the prologue — but a tenth fewer VALU instructions per launch.)
struct WaveMap {
  unsigned int block, n_blocks;
  int cw0, cw1, cwt;  // this wave owns weight units [cw0, cw1) of the workgroup's cwt
  __device__ __forceinline__ long long bound(long long T, int cw) const {
    if (T < (1LL << 26)) {  // always, short of 8.6e9 observations: 32-bit arithmetic, one division by the grid size
      const unsigned int t = (unsigned int)T, q = t / n_blocks, r = t - q * n_blocks;
      const unsigned int wg0 = block * q + (block < r ? block : r);
      const unsigned int wgn = q + (block < r ? 1u : 0u);
      return (long long)(wg0 + (wgn * (unsigned int)cw) / (unsigned int)cwt);
    }
    const long long q = T / n_blocks, r = T - q * n_blocks;
    const long long wg0 = (long long)block * q + ((long long)block < r ? (long long)block : r);
    const long long wgn = q + ((long long)block < r ? 1 : 0);
    return wg0 + wgn * cw / cwt;
  }
  __device__ __forceinline__ long long begin(long long T) const { return bound(T, cw0); }
  __device__ __forceinline__ long long end(long long T) const { return bound(T, cw1); }
};

constexpr int W_OLD = 3, W_YOUNG = 2;

template <int BT, bool WEIGHTED = true>
__device__ __forceinline__ WaveMap make_wave_map(int block, int n_blocks, int wave) {
  WaveMap m;
  m.block = (unsigned int)block;
  m.n_blocks = (unsigned int)n_blocks;
  if (BT == 512 && WEIGHTED) {
    m.cwt = 4 * W_OLD + 4 * W_YOUNG;
    m.cw0 = wave < 4 ? wave * W_OLD : 4 * W_OLD + (wave - 4) * W_YOUNG;
    m.cw1 = m.cw0 + (wave < 4 ? W_OLD : W_YOUNG);
  } else {
    m.cwt = BT / 64;
    m.cw0 = wave;
    m.cw1 = wave + 1;
  }
  return m;
}

// ---------------------------------------------------------------------------------------
// Scan-aligned wave shares of the row layout (equal-shares mode, flag 512 — the default)
// ---------------------------------------------------------------------------------------
// A wave whose run of rows begins or ends inside a scan pays one more per-scan expansion (~130 instructions) than a
// wave that owns whole scans; with scans about as long as a wave's share — C2: 500-point scans = 8 rows, 7.6 rows per
// wave — that was two expansions per wave instead of one, 0.7 us of a 9.9 us launch (scripts/probes/align_exp.py).
// So the boundaries of the equal split are moved to the nearest scan start within half a share, once per (upload,
// grid), into a table of n_blocks * 8 + 1 row indices that lives behind the descriptor array (its own padding row
// included): `wave_split(desc, n_rows)`.  Half a share (rounded up) keeps the boundaries ordered; a boundary with no
// scan start that close stays where the arithmetic split puts it.  Kernels of this mode read their run from the table
// (wave_run), so the step kernel and the [evaluation, controller] launch pair still sum in the same order.
__host__ __device__ inline size_t wave_split_bytes(long long n_rows) { return 16 * ((size_t)n_rows + 1) + 64; }
__device__ __forceinline__ const int* wave_split(const RowDesc* __restrict__ desc, long long n_rows) {
  return reinterpret_cast<const int*>(desc + n_rows + 1);
}
struct WaveRun { long long begin, end; };
__device__ __forceinline__ WaveRun wave_run(const RowDesc* __restrict__ desc, long long n_rows, int block, int wave) {
  const int* __restrict__ sp = wave_split(desc, n_rows) + (__builtin_amdgcn_readfirstlane(block) * 8 + __builtin_amdgcn_readfirstlane(wave));
  WaveRun r;
  r.begin = sp[0];
  r.end = sp[1];
  return r;
}
__global__ void wave_split_kernel(const RowDesc* __restrict__ desc, const int n_rows, const int n_blocks, int* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, total = n_blocks * 8 + 1;
  if (t >= total) return;
  if (t == total - 1) { out[t] = n_rows; return; }
  const WaveMap m = make_wave_map<512, false>(t >> 3, n_blocks, t & 7);
  const int nominal = (int)m.begin(n_rows);
  const int window = ((n_rows / n_blocks) / 8 + 1) / 2;  // <= the smallest share: nearest-start maps of ordered points stay ordered
  int best = nominal;
  if (nominal > 0 && nominal < n_rows) {
    for (int j = 0; j <= window; ++j) {
      const int lo = nominal - j, hi = nominal + j;
      if (lo >= 1 && desc[lo].first != 0) { best = lo; break; }
      if (hi < n_rows && desc[hi].first != 0) { best = hi; break; }
    }
  }
  out[t] = best;
}

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
constexpr int CTILE_DOUBLES = 3 * TILE + TILE / 2;  // 448 doubles = 3 584 B
constexpr int GROUP_DOUBLES = 6;                    // 48 B, 16-B aligned
constexpr int FLAG_COMPACT = 16;                    // stream the compact layout when it is available

typedef unsigned int v2u __attribute__((ext_vector_type(2)));

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

// upload-time helpers of the compact layout -------------------------------------------------
// flag[k] = 1 when record k starts a new group: (n, d, scale) differ bitwise from record k-1.
__global__ void group_flag_kernel(const double* __restrict__ aos, long long n, unsigned char* __restrict__ flag) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  bool nw = (k == 0);
  if (!nw) {
    const unsigned long long* a = reinterpret_cast<const unsigned long long*>(aos + 8 * k);
    const unsigned long long* b = a - 8;
    nw = (a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3]) | (a[7] != b[7]);
  }
  flag[k] = nw ? 1 : 0;
}

__global__ void build_groups_kernel(const double* __restrict__ aos, const long long* __restrict__ starts,
                                    long long n_groups, double* __restrict__ groups) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const double* r = aos + 8 * starts[g];
  double* o = groups + g * GROUP_DOUBLES;
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[7]; o[5] = 0.0;
}

// Records [rec_off[b], rec_off[b+1]) of "problem" b -> compact tiles starting at tile_off[b]
// (single problem: one entry).  One workgroup per problem, grid-stride over y for long ones.
__global__ void build_ctiles_kernel(const double* __restrict__ aos, const unsigned int* __restrict__ gid,
                                    const long long* __restrict__ rec_off, const long long* __restrict__ tile_off,
                                    double* __restrict__ ctiles) {
  const int prob = blockIdx.x;
  const long long r0 = rec_off[prob];
  const long long n = rec_off[prob + 1] - r0;
  const long long n_padded = (tile_off[prob + 1] - tile_off[prob]) * TILE;
  double* tb = ctiles + tile_off[prob] * CTILE_DOUBLES;
  for (long long k = (long long)blockIdx.y * blockDim.x + threadIdx.x; k < n_padded;
       k += (long long)gridDim.y * blockDim.x) {
    double x = 0.0, y = 0.0, z = 0.0;
    unsigned int g = 0u;
    if (k < n) {
      const double* r = aos + 8 * (r0 + k);
      x = r[4]; y = r[5]; z = r[6];
      g = gid[r0 + k];
    }
    const long long t = k / TILE;
    const int j = (int)(k % TILE);
    double* base = tb + t * CTILE_DOUBLES;
    base[j] = x;
    base[TILE + j] = y;
    base[2 * TILE + j] = z;
    reinterpret_cast<unsigned int*>(base + 3 * TILE)[j] = g;
  }
}


// ---------------------------------------------------------------------------------------
// Row layout (clc_rows.hpp; SURVEY.md §8f row 3 taken to its end).  At upload the records are grouped into scans
// (runs of bit-identical (n, d, scale)), every scan is padded to whole ROWS of 64 points, and the device keeps
//   * xy[row][64] (x, y) interleaved — 1 KiB per row, one coalesced 16-byte load per lane — and
//   * desc[row] (64 B, wave-uniform: scalar loads): the scan's plane and scale, the valid count of the row, and
//     whether the row starts a scan,
// i.e. 17 B of traffic per observation for scans that fill their rows (28 B in the compact layout, 64 B algorithmic).
// Needs p.z == 0 for every record (always true for the reference's scan points); otherwise the upload keeps the
// compact / 64-byte layouts only.  A wave owns a contiguous run of rows, keeps DEPTH row loads in flight, accumulates
// per-scan moments (one point per lane per row, ~26 FP64 instructions) and expands them into the 28 accumulators
// when the scan changes (rows_flush).  Same lane->row map for every launch: bitwise reproducible.
// ---------------------------------------------------------------------------------------
#ifndef CLC_ROWS_DEPTH
#define CLC_ROWS_DEPTH 8
#endif
constexpr int ROWS_DEPTH = CLC_ROWS_DEPTH;

template <bool NT>
__device__ __forceinline__ v2d load_row(const double* __restrict__ xy, long long row, int lane) {
  const v2d* p = reinterpret_cast<const v2d*>(xy + row * ROW_DOUBLES) + lane;
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

__device__ __forceinline__ long long uniform_ll(long long v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xFFFFFFFFLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}

// Descriptor of a row as ONE 8-byte vector load per lane (lane l holds double l & 7 of the 64-byte descriptor), so that it
// rides the same in-order, DEPTH-deep vmcnt pipeline as the points; the wave-uniform fields are then read with
// v_readlane.  (Scalar loads of the descriptors, one row ahead, left every wave waiting ~1 us of HBM latency per
// 128-byte line of descriptors: 4-5 TB/s beyond the Infinity Cache instead of what the row stream allows.)
__device__ __forceinline__ double load_desc_lane(const RowDesc* __restrict__ desc, int row, int lane) {
  return reinterpret_cast<const double*>(desc + row)[lane & 7];
}

__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// What a wave does with the rows it streams is a policy: begin_scan(pose, plane of the scan), point(x, y) for every
// valid point of a row, flush(acc) when the scan ends.  LmRows = the LM evaluation (clc_rows.hpp), Normal9Rows = the 9x9
// normal equation of the closed-form initialiser (K5).
template <bool WITH_LOSS>
struct LmRows {
  static constexpr int NA = NACC;
  const double& inv_lf2;  // set by get_pose (it may depend on options that arrive late)
  RowPlane q;
  RowMoments M;
  __device__ __forceinline__ explicit LmRows(const double& inv) : inv_lf2(inv) {}
  __device__ __forceinline__ void begin_scan(const PoseU& P, double nx, double ny, double nz, double d, double s) {
    rows_plane_setup(P.R, P.t, nx, ny, nz, d, s, q);
    rows_moments_reset<WITH_LOSS>(M);
  }
  __device__ __forceinline__ void point(double x, double y) { rows_point<WITH_LOSS>(q, inv_lf2, x, y, M); }
  __device__ __forceinline__ void flush(double (&acc)[NACC]) { rows_flush<WITH_LOSS>(q, M, acc); }
};

constexpr int NACC9 = 45;

// Row A_k = kron([x, y, 1], n), b_k = -d (src/LaseCamCalCeres.cpp:144-158): A^T A = sum kron(b b^T, n n^T) and
// A^T b = -d kron(sum b, n) share the scan's n, so a lane only accumulates the 6 moments of b = (x, y, 1) per scan
// (5 FP64 instructions per point) and expands them once per scan.
// acc layout (as normal9_kernel): [bb(6: xx xy x yy y 1)] x [nn(6: 00 01 02 11 12 22)] then A^T b (9: b-major).
struct Normal9Rows {
  static constexpr int NA = NACC9;
  double nx, ny, nz, md;
  double sxx, sxy, sx, syy, sy, s1;
  __device__ __forceinline__ void begin_scan(const PoseU&, double nx_, double ny_, double nz_, double d, double) {
    nx = nx_; ny = ny_; nz = nz_; md = -d;
    sxx = sxy = sx = syy = sy = s1 = 0.0;
  }
  __device__ __forceinline__ void point(double x, double y) {
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

template <class Policy, bool NT, int DEPTH = ROWS_DEPTH, class PoseFn>
__device__ __forceinline__ bool stream_rows_policy(Policy& pol, const double* __restrict__ xy_all,
                                                   const RowDesc* __restrict__ desc_all, long long r_begin_in,
                                                   long long r_end_in, const int lane, PoseFn get_pose,
                                                   double (&acc)[Policy::NA]) {
  // wave-uniform run [r_begin, r_end): loop control on the scalar unit (32-bit row index relative to the run's first
  // row — 64-bit compares would go through the vector unit)
  const long long r_begin = uniform_ll(r_begin_in);
  const int n = __builtin_amdgcn_readfirstlane((int)(r_end_in - r_begin_in));
  const double* __restrict__ xy = xy_all + r_begin * ROW_DOUBLES;
  const RowDesc* __restrict__ desc = desc_all + r_begin;
  v2d buf[DEPTH];
  double dbuf[DEPTH];
  // Prologue: DEPTH rows in flight, issued UNCONDITIONALLY from clamped row indices (the row arrays carry one padding
  // row, so even an empty run reads mapped memory).  As `if (u < n) load` each load sat in its own branch, and the
  // waits hipcc places at the joins made a wave stall on its first rows of points before it had issued the last ones —
  // and before the barrier in front of the step kernel's controller.
  const int n_last = n > 0 ? n - 1 : 0;
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    const int ru = u < n_last ? u : n_last;
    dbuf[u] = load_desc_lane(desc, ru, lane);
    buf[u] = load_row<NT>(xy, ru, lane);
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
      if (r + DEPTH < n) {
        dbuf[u] = load_desc_lane(desc, r + DEPTH, lane);
        buf[u] = load_row<NT>(xy, r + DEPTH, lane);
      }
      const int count = __builtin_amdgcn_readlane(__double2loint(dv), 5);  // RowDesc: double 5 = {count, first}
      const int first = __builtin_amdgcn_readlane(__double2hiint(dv), 5);
      if (first != 0 || r == 0) {  // wave-uniform: the scan changes (or the wave's run begins inside one)
        if (r != 0) pol.flush(acc);
        pol.begin_scan(P, readlane_d(dv, 0), readlane_d(dv, 1), readlane_d(dv, 2), readlane_d(dv, 3), readlane_d(dv, 4));
      }
      if (lane < count) pol.point(v[0], v[1]);
    }
  }
  if (n > 0) pol.flush(acc);
  return true;
}

template <bool WITH_LOSS, bool NT, int DEPTH = ROWS_DEPTH, class PoseFn>
__device__ __forceinline__ bool stream_rows(const double* __restrict__ xy_all, const RowDesc* __restrict__ desc_all,
                                            long long r_begin_in, long long r_end_in, const int lane, PoseFn get_pose,
                                            const double& inv_lf2, double (&acc)[NACC]) {
  LmRows<WITH_LOSS> pol(inv_lf2);
  return stream_rows_policy<LmRows<WITH_LOSS>, NT, DEPTH>(pol, xy_all, desc_all, r_begin_in, r_end_in, lane, get_pose, acc);
}

// ---- upload-time kernels of the row layout (all O(N) work on the device) -------------------------------------
// Inclusive prefix sum of small unsigned values (flags, rows per scan), three passes: per-block totals, a one-block
// scan of the totals, per-block scan + offset.  out[i] = sum(in[0..i]) - minus_one.
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;

template <class TIn>
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_totals_kernel(const TIn* __restrict__ in, long long n,
                                                                        unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long sh[SCAN_THREADS];
  const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned long long s = 0;
  for (int j = 0; j < SCAN_ITEMS; ++j)
    if (base + j < n) s += (unsigned long long)in[base + j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = SCAN_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = sh[0];
}

// exclusive scan of the block totals in place, one workgroup; totals[n_blocks] receives the grand total
__global__ __launch_bounds__(SCAN_THREADS) void scan_totals_kernel(unsigned long long* __restrict__ totals, long long n_blocks) {
  __shared__ unsigned long long sh[SCAN_THREADS];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long long base = 0; base < n_blocks; base += SCAN_THREADS) {
    const long long i = base + threadIdx.x;
    const unsigned long long v = i < n_blocks ? totals[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
      const unsigned long long a = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < n_blocks) totals[i] = carry + sh[threadIdx.x] - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == SCAN_THREADS - 1) carry += sh[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[n_blocks] = carry;
}

template <class TIn>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const TIn* __restrict__ in, long long n,
                                                                 const unsigned long long* __restrict__ totals,
                                                                 unsigned int minus_one, unsigned int* __restrict__ out) {
  __shared__ unsigned long long sh[SCAN_THREADS];
  const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned int v[SCAN_ITEMS];
  unsigned long long s = 0;
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = base + j < n ? (unsigned int)in[base + j] : 0u;
    s += v[j];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < SCAN_THREADS; off <<= 1) {
    const unsigned long long a = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += a;
    __syncthreads();
  }
  unsigned long long run = totals[blockIdx.x] + sh[threadIdx.x] - s;  // exclusive prefix of this thread's first item
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    run += v[j];
    if (base + j < n) out[base + j] = (unsigned int)(run - minus_one);
  }
}

// flag[k] = 1 when record k starts a new scan ((n, d, scale) differ bitwise from record k-1); *any_z is set when some
// record has p.z != 0 (the row layout then does not apply).
__global__ void scan_flag_kernel(const double* __restrict__ aos, long long n, unsigned char* __restrict__ flag,
                                 unsigned int* __restrict__ any_z) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool z = false;
  if (k < n) {
    bool nw = (k == 0);
    const unsigned long long* a = reinterpret_cast<const unsigned long long*>(aos + 8 * k);
    if (!nw) {
      const unsigned long long* b = a - 8;
      nw = (a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3]) | (a[7] != b[7]);
    }
    flag[k] = nw ? 1 : 0;
    z = aos[8 * k + 6] != 0.0;
  }
  if (__any(z) && (threadIdx.x & 63) == 0) atomicOr(any_z, 1u);
}

// a problem never shares a scan with its predecessor
__global__ void mark_problem_starts_kernel(const long long* __restrict__ rec_off, long long n_problems, long long n,
                                           unsigned char* __restrict__ flag) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_problems && rec_off[p] < n) flag[rec_off[p]] = 1;
}

// starts[g] = first record of scan g; starts[G] = n; rows[g] = rows the scan occupies (filled by scan_rows_kernel)
__global__ void scan_starts_kernel(const unsigned char* __restrict__ flag, const unsigned int* __restrict__ gid, long long n,
                                   long long n_groups, long long* __restrict__ starts) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n && flag[k]) starts[gid[k]] = k;
  if (k == 0) starts[n_groups] = n;
}

__global__ void scan_rows_kernel(const long long* __restrict__ starts, long long n_groups, unsigned int* __restrict__ rows) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_groups) rows[g] = (unsigned int)((starts[g + 1] - starts[g] + ROW - 1) / ROW);
}

// row_begin[G+1]: exclusive prefix of rows[] (row_begin[0] = 0 written here).  One thread per row SLOT.
__global__ void build_rows_kernel(const double* __restrict__ aos, const long long* __restrict__ starts,
                                  const unsigned int* __restrict__ row_begin, long long n_groups, long long n_rows,
                                  double* __restrict__ xy, RowDesc* __restrict__ desc) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t >> 6;
  const int lane = (int)(t & 63);
  if (r >= n_rows) return;
  // scan of row r: the last g with row_begin[g] <= r  (wave-uniform search)
  long long lo = 0, hi = n_groups;  // invariant: row_begin[lo] <= r < row_begin[hi]
  while (hi - lo > 1) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)row_begin[mid] <= r) lo = mid; else hi = mid;
  }
  const long long g = lo;
  const long long first = starts[g] + (r - (long long)row_begin[g]) * ROW;
  const long long end = starts[g + 1];
  const long long k = first + lane;
  double x = 0.0, y = 0.0;
  if (k < end) { x = aos[8 * k + 4]; y = aos[8 * k + 5]; }
  v2d v; v[0] = x; v[1] = y;
  reinterpret_cast<v2d*>(xy + r * ROW_DOUBLES)[lane] = v;
  if (lane == 0) {
    const double* a = aos + 8 * starts[g];
    RowDesc d;
    d.nx = a[0]; d.ny = a[1]; d.nz = a[2]; d.d = a[3]; d.s = a[7];
    d.count = (int32_t)((end - first) < ROW ? (end - first) : ROW);
    d.first = (r == (long long)row_begin[g]) ? 1 : 0;
    d.pad_[0] = 0.0; d.pad_[1] = 0.0;
    desc[r] = d;
  }
}

// prob_row[p] = first row of problem p (problems start scans); prob_row[P] = n_rows
__global__ void problem_rows_kernel(const long long* __restrict__ rec_off, const unsigned int* __restrict__ gid,
                                    const unsigned int* __restrict__ row_begin, long long n_problems, long long n,
                                    long long n_rows, long long* __restrict__ prob_row) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_problems) return;
  prob_row[p] = (p < n_problems && rec_off[p] < n) ? (long long)row_begin[gid[rec_off[p]]] : n_rows;
}

// groups[g] of the compact layout, from the device-resident starts
__global__ void build_groups_dev_kernel(const double* __restrict__ aos, const long long* __restrict__ starts,
                                        long long n_groups, double* __restrict__ groups) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const double* r = aos + 8 * starts[g];
  double* o = groups + g * GROUP_DOUBLES;
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[7]; o[5] = 0.0;
}

// ---------------------------------------------------------------------------------------
// Residual-block construction on the device (src/LaseCamCalCeres.cpp:222-295) from the pose-major form of
// std::vector<Oberserve>: tag poses + CSR scan points stay resident (24 B per point crossed PCIe instead of the 64-byte
// records), and the records of any (use_linefitting_data, use_boundary_constraint) selection are produced here.
// Every operation is an individually rounded IEEE operation (__dmul_rn / __dadd_rn / __dsub_rn: no FMA contraction), in
// the order of clc::host::flatten, so the records are bitwise those of the host path (clc_flatten_observations).
// One workgroup per pose; rec_off[i] = first record of pose i (exclusive prefix of per-pose record counts, host-built:
// O(poses)).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3_rn(const double* a, const double* b, double* c) {
  c[0] = __dsub_rn(__dmul_rn(a[1], b[2]), __dmul_rn(a[2], b[1]));
  c[1] = __dsub_rn(__dmul_rn(a[2], b[0]), __dmul_rn(a[0], b[2]));
  c[2] = __dsub_rn(__dmul_rn(a[0], b[1]), __dmul_rn(a[1], b[0]));
}

__device__ __forceinline__ void pi_from_ppp_rn(const double* x1, const double* x2, const double* x3, double* pi) {
  const double a[3] = {__dsub_rn(x1[0], x3[0]), __dsub_rn(x1[1], x3[1]), __dsub_rn(x1[2], x3[2])};
  const double b[3] = {__dsub_rn(x2[0], x3[0]), __dsub_rn(x2[1], x3[1]), __dsub_rn(x2[2], x3[2])};
  double c12[3];
  cross3_rn(a, b, pi);
  cross3_rn(x1, x2, c12);
  pi[3] = -__dadd_rn(__dadd_rn(__dmul_rn(x3[0], c12[0]), __dmul_rn(x3[1], c12[1])), __dmul_rn(x3[2], c12[2]));
}

__global__ __launch_bounds__(BLOCK) void flatten_kernel(const int n_poses, const double* __restrict__ tag_q_wxyz,
                                                        const double* __restrict__ tag_t, const long long* __restrict__ pts_off,
                                                        const double* __restrict__ pts, const long long* __restrict__ ptl_off,
                                                        const double* __restrict__ ptl, const int linefit, const int boundary,
                                                        const long long* __restrict__ rec_off, double* __restrict__ rec) {
  const int i = blockIdx.x;
  if (i >= n_poses) return;
  // plane of the tag (z_tag = 0) in the camera frame: [R_ca e3 ; -(R_ca e3).t_ca]   (:227-231)
  const double x = tag_q_wxyz[4 * i + 1], y = tag_q_wxyz[4 * i + 2], z = tag_q_wxyz[4 * i + 3], w = tag_q_wxyz[4 * i];
  const double tx = __dmul_rn(2.0, x), ty = __dmul_rn(2.0, y), tz = __dmul_rn(2.0, z);
  const double twx = __dmul_rn(tx, w), twy = __dmul_rn(ty, w), twz = __dmul_rn(tz, w);
  const double txx = __dmul_rn(tx, x), txy = __dmul_rn(ty, x), txz = __dmul_rn(tz, x);
  const double tyy = __dmul_rn(ty, y), tyz = __dmul_rn(tz, y), tzz = __dmul_rn(tz, z);
  double R[9];
  R[0] = __dsub_rn(1.0, __dadd_rn(tyy, tzz)); R[1] = __dsub_rn(txy, twz);                R[2] = __dadd_rn(txz, twy);
  R[3] = __dadd_rn(txy, twz);                R[4] = __dsub_rn(1.0, __dadd_rn(txx, tzz)); R[5] = __dsub_rn(tyz, twx);
  R[6] = __dsub_rn(txz, twy);                R[7] = __dadd_rn(tyz, twx);                R[8] = __dsub_rn(1.0, __dadd_rn(txx, tyy));
  const double t[3] = {tag_t[3 * i], tag_t[3 * i + 1], tag_t[3 * i + 2]};
  const double n[3] = {R[2], R[5], R[8]};
  const double d = -__dadd_rn(__dadd_rn(__dmul_rn(n[0], t[0]), __dmul_rn(n[1], t[1])), __dmul_rn(n[2], t[2]));
  const long long* off = linefit ? ptl_off : pts_off;  // :233-237
  const double* P = linefit ? ptl : pts;
  const long long lo = off[i], cnt = off[i + 1] - lo;
  const double scale = __ddiv_rn(1.0, __dsqrt_rn((double)cnt));  // :239-240
  double* out = rec + 8 * rec_off[i];
  for (long long j = threadIdx.x; j < cnt; j += blockDim.x) {
    double* o = out + 8 * j;
    const double* p = P + 3 * (lo + j);
    o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = d;
    o[4] = p[0]; o[5] = p[1]; o[6] = p[2]; o[7] = scale;
  }
  if (boundary && linefit && threadIdx.x == 0) {  // :258-294 (the host checked that the scan is not empty, :278)
    const double orig = 0.0265 + 0.0165;  // :262
    const double pm[3][3] = {{0.0 - orig, 0.0 - orig, 0.0}, {0.5 - orig, 0.0 - orig, 0.0}, {0.0 - orig, 0.5 - orig, 0.0}};
    double pc[3][3];
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a)  // :270-272
        pc[k][a] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(R[3 * a], pm[k][0]), __dmul_rn(R[3 * a + 1], pm[k][1])),
                                       __dmul_rn(R[3 * a + 2], pm[k][2])), t[a]);
    const double zero[3] = {0.0, 0.0, 0.0};
    double pi1[4], pi2[4];
    pi_from_ppp_rn(pc[0], pc[1], zero, pi1);  // :275
    pi_from_ppp_rn(pc[0], pc[2], zero, pi2);  // :276
    const double* front = pts + 3 * pts_off[i];           // obi.points.at(0), :278
    const double* back = pts + 3 * (pts_off[i + 1] - 1);  // obi.points.at(size-1), :279
    double* a = out + 8 * cnt;
    a[0] = pi1[0]; a[1] = pi1[1]; a[2] = pi1[2]; a[3] = pi1[3]; a[4] = front[0]; a[5] = front[1]; a[6] = front[2]; a[7] = scale;
    double* b = a + 8;
    b[0] = pi2[0]; b[1] = pi2[1]; b[2] = pi2[2]; b[3] = pi2[3]; b[4] = back[0]; b[5] = back[1]; b[6] = back[2]; b[7] = scale;
  }
}

// K1 on the row layout: same contract as eval_kernel (one 28-double partial per workgroup).
template <bool WITH_LOSS, bool NT, int BT, bool WEIGHTED>
__global__ __launch_bounds__(BT) void eval_rows_kernel(const double* __restrict__ xy, const RowDesc* __restrict__ desc,
                                                       const long long n_rows, const double* __restrict__ pose,
                                                       const int32_t* __restrict__ status, const double lf,
                                                       const int reduce_mode, double* __restrict__ partials,
                                                       const Pose7 pose_arg, const int use_pose_arg) {
  auto get_pose = [&](PoseU& P) -> bool {
    if (use_pose_arg) {
      load_pose(pose_arg.v, P);
      return true;
    }
    const int32_t st = status != nullptr ? *status : (int32_t)CLC_RUNNING;
    load_pose(pose, P);
    return st == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  long long r0, r1;
  if (BT == 512 && !WEIGHTED) {  // equal shares, boundaries at scan starts (wave_split_kernel)
    const WaveRun run = wave_run(desc, n_rows, blockIdx.x, threadIdx.x >> 6);
    r0 = run.begin;
    r1 = run.end;
  } else {
    const WaveMap wm = make_wave_map<BT, WEIGHTED>(blockIdx.x, gridDim.x, threadIdx.x >> 6);
    r0 = wm.begin(n_rows);
    r1 = wm.end(n_rows);
  }
  if (!stream_rows<WITH_LOSS, NT>(xy, desc, r0, r1, lane, get_pose, inv_lf2, acc)) return;
  block_reduce_store<BT / 64>(acc, reduce_mode, partials + (size_t)blockIdx.x * NACC);
}

template <bool WITH_LOSS, bool WITH_JAC, bool PREFETCH, bool NT, bool COMPACT, int BT>
__global__ __launch_bounds__(BT) void eval_kernel(const double* __restrict__ tiles,
                                                  const double* __restrict__ groups,
                                                  const long long n,
                                                  const double* __restrict__ pose,
                                                  const int32_t* __restrict__ status,
                                                  const double lf, const int reduce_mode,
                                                  double* __restrict__ partials, const Pose7 pose_arg,
                                                  const int use_pose_arg) {
  // first launch of a solve: the LM state is not initialised yet (the first lm_kernel does that),
  // so the pose comes by value and the (stale) termination flag is ignored
  auto get_pose = [&](PoseU& P) -> bool {
    if (use_pose_arg) {
      load_pose(pose_arg.v, P);
      return true;
    }
    const int32_t st = status != nullptr ? *status : (int32_t)CLC_RUNNING;  // both scalar loads issued before the
    load_pose(pose, P);                                                     // first use: one wait, not two
    return st == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<BT>(blockIdx.x, gridDim.x, threadIdx.x >> 6);
  bool active;
  if (COMPACT && PREFETCH)
    active = stream_ctiles_deep<WITH_LOSS, WITH_JAC, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else if (COMPACT)
    active = stream_ctiles<WITH_LOSS, WITH_JAC, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else
    active = stream_tiles<WITH_LOSS, WITH_JAC, PREFETCH, NT>(tiles, n, wm, lane, get_pose, inv_lf2, acc);
  if (!active) return;  // uniform over the launch: the solve had already terminated
  block_reduce_store<BT / 64>(acc, reduce_mode, partials + (size_t)blockIdx.x * NACC);
}

// Profiling twin of the default evaluation kernel (loss, Jacobian, compact layout): identical work,
// plus per-workgroup stamps {wall start, wall end (100 MHz s_memrealtime, chip-global),
// shader cycles: prologue, streaming loop, reduction epilogue}.  Debug/analysis only.
template <int BT>
__global__ __launch_bounds__(BT) void eval_timeline_kernel(const double* __restrict__ ctiles,
                                                           const double* __restrict__ groups, const long long n,
                                                           const double* __restrict__ pose, const double lf,
                                                           double* __restrict__ partials,
                                                           long long* __restrict__ stamps) {
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  PoseU P;
  load_pose(pose, P);
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long wave_global = (long long)blockIdx.x * (BT / 64) + wave;
  const WaveMap wm = make_wave_map<BT>(blockIdx.x, gridDim.x, wave);
  const long long c1 = clock64();
  stream_ctiles<true, true, false>(ctiles, groups, n, wm, lane, [&](PoseU& Q) { Q = P; return true; }, inv_lf2, acc);
  const long long c2 = clock64();
  const long long w2 = wall_clock64();
  block_reduce_store<BT / 64>(acc, 0, partials + (size_t)blockIdx.x * NACC);
  const long long c3 = clock64();
  if (lane == 0) {  // one record per WAVE: {wall start, wall end of loop, cycles prologue, loop, epilogue}
    long long* s = stamps + 8 * (size_t)wave_global;
    s[0] = w0; s[1] = w2; s[2] = c1 - c0; s[3] = c2 - c1; s[4] = c3 - c2; s[5] = wall_clock64();
  }
}

// Fixed-order sum of the block partials: thread (c, rg) sums rows rg, rg+8, ... of column c
// (16 independent loads in flight per round — a dependent load chain here costs more than
// the whole evaluation kernel), then the 8 row groups are combined in order.
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int n_blocks,
                                                double (*red)[32]) {
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  constexpr int RG = BLOCK / 32, UNROLL = 16;
  double s = 0.0;
  if (c < NACC) {
    for (int b0 = rg; b0 < n_blocks; b0 += RG * UNROLL) {
      double v[UNROLL];
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) {
        const int b = b0 + RG * j;
        v[j] = (b < n_blocks) ? partials[(size_t)b * NACC + c] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) s += v[j];
    }
  }
  red[rg][c] = s;
  __syncthreads();
}

__global__ __launch_bounds__(BLOCK) void reduce_kernel(const double* __restrict__ partials,
                                                       int n_blocks, int with_loss, double lf,
                                                       double* __restrict__ out28) {
  __shared__ double red[BLOCK / 32][32];
  reduce_partials(partials, n_blocks, red);
  if (threadIdx.x < NACC) {
    double s = 0.0;
#pragma unroll
    for (int rg = 0; rg < BLOCK / 32; ++rg) s += red[rg][threadIdx.x];
    out28[threadIdx.x] = (threadIdx.x == 27) ? finalize_cost(s, with_loss != 0, lf) : s;
  }
}

// ---------------------------------------------------------------------------------------
// K2 — reduction of the block partials + LM controller (single problem).
// ---------------------------------------------------------------------------------------
constexpr int LM_STATE_WORDS = (int)((sizeof(LmState) + 7) / 8);

// Host-visible completion record in pinned (fine-grained) host memory.  lm_kernel publishes
// the number of evaluation passes consumed after every LM step and, at termination, the
// result — so the host can keep the launch queue primed without ever blocking on the stream.
struct HostMailbox {
  int32_t status;  // CLC_RUNNING until the controller terminates
  int32_t n_done;  // evaluation passes consumed so far
  clc_summary summary;
  double pose[7];
  long long prof[8];  // shader-clock stamps of the last lm_kernel launch (debug/profiling)
};

// Tail shared by lm_kernel (own launch) and eval_lm_kernel (last-arriving workgroup of the
// evaluation launch): fixed-order reduction of the block partials + LM controller + publish.
// COHERENT: read partials with agent-scope (sc1) loads — required when they were produced by
// other workgroups of the SAME launch.
template <bool COHERENT>
__device__ __forceinline__ double load_partial(const double* p) {
  if (COHERENT)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// The global loads of the controller, issued as early as possible and consumed later (lm_tail): the thread's
// word of the LM state, its share of the first 256 partial rows and, for the first lane of wave 1, the pass count.
// Summation tree (the same for every caller, so all solve paths agree bit for bit): 16 row groups, group g = rows
// g, g + 16, g + 32, ... summed in that order, then the 16 group sums combined in order.  With HT = 512 helper
// threads, thread (c, g) owns group g of column c (16 loads per round of 256 rows); with HT = 256 it owns groups g and
// g + 8 (two separate sums of 16).  The row buffer is mapped in whole rounds of 256 rows (ensure_partials), so the
// addresses need no clamp: one base pointer, constant strides; rows beyond the grid are masked when they are summed.
// (A "load or 0.0" select on the runtime row count made hipcc branch around every load, cdna_hip_programming.md §5
// trap (c).)
constexpr int LM_GROUPS = 16;

struct LmLoads {
  double v[32];  // HT = 256: [0,16) group g, [16,32) group g + 8;  HT = 512: [0,16) group g
  double my_word;
  long long passes_before;
};

template <bool COHERENT, bool FIRST, int HT>
__device__ __forceinline__ void lm_issue_loads(const double* __restrict__ partials, const LmState* __restrict__ state,
                                               LmLoads& L) {
  static_assert(HT == 256 || HT == 512, "helper threads");
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  L.passes_before = 0;
  L.my_word = 0.0;
  if (threadIdx.x < HT) {  // wave-uniform
    if (!FIRST && threadIdx.x == 64) L.passes_before = state->n_evals;
    const int cw = threadIdx.x < LM_STATE_WORDS ? threadIdx.x : LM_STATE_WORDS - 1;
    L.my_word = reinterpret_cast<const double*>(state)[cw];
    const int cc = c < NACC ? c : NACC - 1;
    const double* base = partials + (size_t)g * NACC + cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) L.v[j] = load_partial<COHERENT>(base + (size_t)(LM_GROUPS * j) * NACC);
    if (HT == 256) {
#pragma unroll
      for (int j = 0; j < 16; ++j) L.v[16 + j] = load_partial<COHERENT>(base + (size_t)(8 + LM_GROUPS * j) * NACC);
    }
  }
}

// Cycle stamps inside lm_advance_wave (debug build -DCLC_STAMPS, scripts/r02_stamps.py); nothing otherwise.
#ifdef CLC_STAMPS
#define CLC_CK() do { ck[nck++] = clock64(); } while (0)
#else
#define CLC_CK() do {} while (0)
#endif

// ---------------------------------------------------------------------------------------
// lm_advance on a wavefront (the step kernel's controller)
// ---------------------------------------------------------------------------------------
// The serial controller (clc_lm.hpp, one lane, state in LDS) is a chain of dependent FP64 instructions and LDS round
// trips: 7 300 cycles = 3.0 us of every step_kernel launch, in every workgroup (scripts/r02_stamps.py).  What the
// instructions cost when ONE wave runs them alone (scripts/probes/latency_probe.hip, cycles): dependent FMA 4-6, a
// value through v_readlane into the next FMA 24-31, IEEE division 72-98, IEEE sqrt 108-146, dependent LDS read 72-93,
// compare + select (or branch) 40-50.  So the controller here is written for a short critical path and few branches:
//   * the state is read from LDS once, up front, in one batch; everything scalar (pose, costs, radius, the triangular
//     solves' running values) is computed redundantly by all lanes ("uniform") from broadcast LDS reads;
//   * lane i < 6 owns row i of the Gauss-Newton matrix: scaling, damping, the factorisation's column updates and the
//     matrix-vector product of the model cost change are one instruction for all rows; the Cholesky factorisation is
//     right-looking (column j scaled, then subtracted from the columns to its right) with the diagonal in its own
//     register — per element the same subtractions in the same order as the left-looking serial loop — and the only
//     values that cross lanes are the pivots, the column entries, the forward substitution's z and the gradient
//     (v_readlane);
//   * the trust-region step is computed BEFORE the convergence tests that may make it unnecessary, so that the two
//     Plus operations of an iteration — Plus(x, -g) for the projected gradient norm and Plus(x, step) for the
//     candidate — run as one instruction stream in lanes 0 and 1; tolerance tests, acceptance and the radius update are
//     selects, not branches.  Nothing of the speculative step is committed unless the serial controller would have
//     computed it.  A step that turns out invalid (rare) is handed to the serial loop (lm_iterate);
//   * the candidate and the status are published first, the workgroup's waves meet at ONE barrier (inside this
//     function for the calling wave, in lm_tail_after_barrier for the others) and the rest of the state is written
//     back behind it, while the other waves already stream.
// Every expression keeps the operand order and the fused multiply-adds of clc_lm.hpp / clc_math.hpp: pose, summary and
// iteration trace of a solve are BIT-IDENTICAL to lm_advance<Se3Manifold>'s (the [evaluation, lm_kernel] launch pair
// still runs the serial controller: test_step_kernel_solve_matches_two_kernel_path, the randomized problem test and
// the invalid-step test compare the two bit for bit).  The LM state agrees as well while a solve runs; after a
// termination by parameter / function tolerance the fields that are not outputs (x, g, H, radius) hold the rejected
// pass instead of the last accepted one — nothing reads them any more.
// Called by all 64 lanes of one wave; `tot` (LDS): the 28 totals of this pass (H 0..20, g 21..26, cost sum 27), written
// by this same wave (LDS operations of one wave execute in program order; the caller fences).
// LEAN (the resident batched kernel, clc_resident.hpp): the same arithmetic with a small register footprint.  The calling
// wave keeps ~100 VGPRs of scan points alive across the controller there, and this function, written for a short critical
// path, holds ~180 VGPRs (x, x_eval, the column scales, the scaled matrix and the bookkeeping all stay in registers from
// the first batch of LDS reads to the write-back).  LEAN stores what is final as soon as it is known — the state's home
// is LDS anyway —, parks the scaled system in `park` and reads x, the scales, the gradient and the scaled system back
// right before the model cost change and the two Plus that need them: three more LDS round trips (~300 cycles of a
// controller that overlaps the co-resident problem's streaming there), ~60 VGPRs less.  No trace in this mode.
template <bool FIRST, bool LEAN = false>
__device__ __forceinline__ void lm_advance_wave(LmState& s, const clc_options& o, clc_iteration* __restrict__ trace,
                                                const int trace_cap, const double* tot, double* park, const int lane,
                                                unsigned long long* stamp_row = nullptr /* debug builds */) {
  // `park`: LDS nobody else touches; 32 doubles in, room for the serial controller's temporaries (LmScratch) on the
  // invalid-step path — held in registers they made hipcc spill the whole kernel's controller.
  // FIRST: the pass at the start point (state fresh from lm_init, phase 0); otherwise the pass at a candidate (phase 1 —
  // the only other phase a running solve can be in).  The caller has checked that the solve is still running.
  // Written without early exits and with selects instead of branches wherever both sides are cheap: a compare feeding a
  // branch or a select costs a single wave 40-50 cycles (latency_probe), and there were ~35 of them.
  constexpr int NP = 6, NA = 7;
  constexpr double DMAX = 1.7976931348623157e308;
#ifdef CLC_STAMPS
  long long ck[12];
  int nck = 0;
#endif
  CLC_CK();
  const unsigned i6 = lane < NP ? (unsigned)lane : NP - 1u;
  // packed upper triangle: index of (a, b), a <= b, is a * (2 NP - 1 - a) / 2 + b
  const unsigned rowbase = (i6 * (2u * NP - 1u - i6)) >> 1;
  const unsigned dgi = rowbase + i6;  // H[i][i]
  unsigned hidx[NP];
#pragma unroll
  for (unsigned b = 0; b < NP; ++b) hidx[b] = b < i6 ? ((b * (2u * NP - 1u - b)) >> 1) + i6 : rowbase + b;
  // ---- everything that comes from LDS, in one batch: this pass ...
  const double cost_acc = tot[27];
  double g = tot[21 + i6], Hd = tot[dgi], Hrow[NP];  // lane i: g[i], H[i][i], row i of H
#pragma unroll
  for (int b = 0; b < NP; ++b) Hrow[b] = tot[hidx[b]];
  // ... and the state
  const int iteration = s.iteration, n_invalid_in = s.n_invalid, reuse_in = s.reuse_diagonal;
  const int n_succ_in = s.num_successful, n_unsucc_in = s.num_unsuccessful, n_trace_in = s.n_trace;
  const long long n_evals = s.n_evals + 1;
  double x[NA], xe[NA], sc[NP];
#pragma unroll
  for (int i = 0; i < NA; ++i) { x[i] = s.x[i]; xe[i] = s.x_eval[i]; }
#pragma unroll
  for (int b = 0; b < NP; ++b) sc[b] = s.scale[b];
  double x_norm = s.x_norm, x_cost = s.x_cost, minimum_cost = s.minimum_cost, initial_cost = s.initial_cost;
  double min_iter_cost = s.min_iter_cost, radius = s.radius, dfac = s.decrease_factor, gmax = s.gmax;
  const double mcc = s.model_cost_change;
  double scale = s.scale[i6];
  const double diag = s.diag[i6];
  // every load above is issued before the first value is consumed: one LDS round trip, not three
  __builtin_amdgcn_sched_barrier(0);
  const double cost_e = finalize_cost(cost_acc, o.use_loss != 0, o.loss_scale_factor);
  const bool finite_eval = fabs(cost_e) <= DMAX;
  // ---- the pass just evaluated: early terminations (flags; no output of the solve changes then), acceptance ----
  int early = CLC_RUNNING;  // termination before the iteration is recorded
  bool success = true;
  int reuse = reuse_in;
  int it_iteration = 0;
  double it_cost, it_cost_change = 0.0, it_step_norm = 0.0, it_rel = 0.0;
  if (FIRST) {
    // ---- IterationZero ----
    early = finite_eval ? CLC_RUNNING : CLC_FAILURE;
    x_cost = cost_e;
    if (o.jacobi_scaling) {  // once per solve: computed by the lane that owns the column, broadcast through LDS
      scale = 1.0 / (1.0 + sqrt(Hd));
      if (lane < NP) s.scale[lane] = scale;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int b = 0; b < NP; ++b) sc[b] = s.scale[b];
    }
    initial_cost = x_cost;
    min_iter_cost = x_cost;
    it_cost = x_cost;
  } else {
    it_iteration = iteration;
    const double candidate_cost = finite_eval ? cost_e : DMAX;
    // ---- ParameterToleranceReached, FunctionToleranceReached ----
    double sn = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) sn += (x[i] - xe[i]) * (x[i] - xe[i]);
    it_step_norm = sqrt_pos(sn);
    it_cost_change = x_cost - candidate_cost;
    const bool par_tol = it_step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance);
    const bool fun_tol = fabs(it_cost_change) <= o.function_tolerance * x_cost;
    early = par_tol ? CLC_CONVERGENCE_PARAMETER : (fun_tol ? CLC_CONVERGENCE_FUNCTION : CLC_RUNNING);
    // ---- IsStepSuccessful; HandleSuccessfulStep / HandleUnsuccessfulStep as selects ----
    it_rel = it_cost_change * rcp_pos_safe(mcc);
    success = it_rel > o.min_relative_decrease;
    const double q = 2.0 * it_rel - 1.0;  // StepAccepted
    double den = 1.0 - q * q * q;
    den = den > (1.0 / 3.0) ? den : (1.0 / 3.0);
    double r_acc = radius * rcp_pos(den);
    r_acc = r_acc < o.max_trust_region_radius ? r_acc : o.max_trust_region_radius;
    const double r_rej = radius * rcp_pos(dfac);  // StepRejected: radius / decrease_factor, exact (a power of two)
    radius = success ? r_acc : r_rej;
    dfac = success ? 2.0 : dfac * 2.0;
    reuse = success ? 0 : 1;
    it_cost = candidate_cost;
    double xn2 = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) xn2 += xe[i] * xe[i];
    const double xe_norm = sqrt_pos(xn2);
    x_norm = success ? xe_norm : x_norm;
    x_cost = success ? candidate_cost : x_cost;
#pragma unroll
    for (int i = 0; i < NA; ++i) x[i] = success ? xe[i] : x[i];
    if (!success) {
      // a rejected step (the minority) is recomputed from the Gauss-Newton system at x: one more LDS round trip on this
      // branch instead of 8 more doubles loaded and selected on every pass (the kernel has no registers to spare)
      g = s.g[i6];
      Hd = s.H[dgi];
#pragma unroll
      for (int b = 0; b < NP; ++b) Hrow[b] = s.H[hidx[b]];
    }
  }
  const int it_succ = success ? 1 : 0;
  const int n_succ = n_succ_in + it_succ, n_unsucc = n_unsucc_in + (1 - it_succ);
  const bool xout_dirty = success && x_cost < minimum_cost;
  minimum_cost = xout_dirty ? x_cost : minimum_cost;
  min_iter_cost = it_cost < min_iter_cost ? it_cost : min_iter_cost;
  // What is final already and not part of a solve's outputs goes back to LDS now, unconditionally (after an early
  // termination nobody reads it): the stores overlap the solve below and free their registers — with everything held
  // until the write-back the kernel spilled.  The trace record's fields wait in `park`.
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) s.x[i] = x[i];
    s.x_norm = x_norm;
    s.x_cost = x_cost;
    s.decrease_factor = dfac;
    s.radius = radius;
    park[0] = it_cost;
    park[1] = it_cost_change;
    park[2] = it_step_norm;
    park[3] = it_rel;
  }
  if (lane < NP) {  // (a rejected step stores back what it loaded)
    s.g[lane] = g;
    s.scale[lane] = scale;
#pragma unroll
    for (int b = 0; b < NP; ++b)
      if (b >= lane) s.H[hidx[b]] = Hrow[b];
  }
  if (LEAN) {
    // the part of write_back() that is known by now (outputs of the solve: only if the pass did not terminate it)
    if (lane == 0 && early == CLC_RUNNING) {
      s.num_successful = n_succ;
      s.num_unsuccessful = n_unsucc;
      s.n_trace = n_trace_in + 1;
      s.n_evals = n_evals;
      s.initial_cost = initial_cost;
      s.minimum_cost = minimum_cost;
      s.min_iter_cost = min_iter_cost;
      if (xout_dirty) {
#pragma unroll
        for (int i = 0; i < NA; ++i) s.x_out[i] = x[i];
      }
    }
  }
  CLC_CK();
  // ---- lm_compute_step, ahead of the tests that may make it unnecessary (committed after them) ----
  double Hs[NP], A[NP];
#pragma unroll
  for (int b = 0; b < NP; ++b) {
    Hs[b] = Hrow[b] * (scale * sc[b]);  // entry b == lane is the diagonal, H[i][i] * (scale[i] * scale[i])
    A[b] = Hs[b];                       // working copy for the factorisation; its diagonal entry is not used (Ad)
  }
  const double gs = g * scale;
  const double Hds = Hd * (scale * scale);
  double dcl = Hds;
  dcl = dcl > o.min_lm_diagonal ? dcl : o.min_lm_diagonal;
  dcl = dcl < o.max_lm_diagonal ? dcl : o.max_lm_diagonal;
  const double diag_n = reuse ? diag : dcl;
  const double inv_radius = rcp_pos(radius);
  double Ad = Hds + diag_n * inv_radius;  // lane i: the damped diagonal entry, updated in place by the factorisation
  if (LEAN) {  // row i of the scaled system waits in LDS for the model cost change (park[8 + 8 i ...]: Hs[0..5], gs)
    if (lane < NP) {
#pragma unroll
      for (int b = 0; b < NP; ++b) park[8 + 8 * lane + b] = Hs[b];
      park[8 + 8 * lane + 6] = gs;
    }
  }
  CLC_CK();
  // Cholesky, right-looking: Lc[j] = column j of L (lane i: L[i][j], meaningful for i > j), inv[j] = 1 / L[j][j]
  // (a pivot <= 0 or NaN makes its reciprocal square root, and with it y[j], NaN: the finiteness test of y below is the
  // serial code's two tests in one)
  double inv[NP], Lc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const double d = readlane_d(Ad, j);
    inv[j] = rsqrt_pos(d);
    Lc[j] = A[j] * inv[j];
    Ad -= Lc[j] * Lc[j];
#pragma unroll
    for (int k = j + 1; k < NP; ++k) A[k] -= Lc[j] * readlane_d(Lc[j], k);
  }
  CLC_CK();
  // L z = gs: lane i carries row i's running value; z[k] is final after k subtractions
  double z[NP], run = gs;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    z[k] = readlane_d(run * inv[k], k);
    run -= Lc[k] * z[k];
  }
  // L^T y = z, uniform, subtractions in ascending k like the serial loop
  double y[NP];
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double acc = z[i];
#pragma unroll
    for (int k = i + 1; k < NP; ++k) acc -= readlane_d(Lc[i], k) * y[k];
    y[i] = acc * inv[i];
  }
  double fin = 0.0;  // 0 * y is (+-)0 for finite y and NaN otherwise
#pragma unroll
  for (int c = 0; c < NP; ++c) fin = fma(y[c], 0.0, fin);
  const bool ok = fin == 0.0;
  CLC_CK();
  double step_n[NP], sg = 0.0, shs = 0.0, row = 0.0;
#pragma unroll
  for (int a = 0; a < NP; ++a) step_n[a] = -y[a];
  double Hs2[NP], gs2 = gs;
#pragma unroll
  for (int b = 0; b < NP; ++b) Hs2[b] = Hs[b];
  if (LEAN) {  // back from LDS (same wave: program order), not hoisted above the factorisation
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int b = 0; b < NP; ++b) Hs2[b] = park[8 + 8 * i6 + b];
    gs2 = park[8 + 8 * i6 + 6];
  }
#pragma unroll
  for (int b = 0; b < NP; ++b) row += Hs2[b] * step_n[b];  // lane a: (Hs step)[a]
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    sg += step_n[a] * readlane_d(gs2, a);
    shs += step_n[a] * readlane_d(row, a);
  }
  const double mcc_n = -(sg + 0.5 * shs);
  const bool step_ok = ok && mcc_n > 0.0;
  CLC_CK();
  // ---- Plus: lane 0 the projected gradient (after a change of x / g), lane 1 the candidate ----
  double cand[NA];
  {
    double g3 = g, sc3[NP], x3[NA];
#pragma unroll
    for (int c = 0; c < NP; ++c) sc3[c] = sc[c];
#pragma unroll
    for (int i = 0; i < NA; ++i) x3[i] = x[i];
    if (LEAN) {  // x, g and the column scales as stored above (lane 0 / lanes < NP wrote them; every lane reads)
      __builtin_amdgcn_sched_barrier(0);
      g3 = s.g[i6];
#pragma unroll
      for (int c = 0; c < NP; ++c) sc3[c] = s.scale[c];
#pragma unroll
      for (int i = 0; i < NA; ++i) x3[i] = s.x[i];
    }
    double dlt[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const double ng = -readlane_d(g3, c);
      const double dc = step_n[c] * sc3[c];  // undo column scaling
      dlt[c] = lane == 1 ? dc : ng;
    }
    pose_plus_rcp(x3, dlt, cand);
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) m = fmax(m, fabs(x3[i] - cand[i]));
    const double gnew = readlane_d(m, 0);
    gmax = success ? gnew : gmax;
  }
  const double it_gmax = gmax;
  CLC_CK();
  // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
  const bool cap_hit = it_iteration >= o.max_num_iterations;
  const bool grad_tol = success && it_gmax <= o.gradient_tolerance;
  const bool rad_tol = radius <= o.min_trust_region_radius;
  const int status = cap_hit ? CLC_NO_CONVERGENCE : (grad_tol ? CLC_CONVERGENCE_GRADIENT : (rad_tol ? CLC_CONVERGENCE_RADIUS : CLC_RUNNING));
  // next iteration: the step computed above is the one the serial controller computes at this point
  const bool cont = status == CLC_RUNNING;
  const bool candidate_ready = cont && step_ok;
  const bool invalid_step = cont && !step_ok;  // rare: handed to the serial loop below, once the state is back in LDS
  const bool step_dirty = cont && ok;  // factorisation and solve went through: step and model cost change are stored
  CLC_CK();
  // ---- what the other waves wait for — the next point to evaluate and whether the solve goes on — first; they leave
  // for their rows at the barrier below while this wave writes the rest of the state back (the trace record, ~45 LDS
  // words, the bookkeeping's selects: ~0.7 us that used to sit in front of every wave's first row) ----
  auto write_back = [&]() {
    if (lane == 0) {
      if (trace != nullptr && n_trace_in < trace_cap) {
        clc_iteration it;
        it.iteration = it_iteration;
        it.step_is_valid = 1;
        it.step_is_successful = it_succ;
        it.pad_ = 0;
        it.cost = park[0];
        it.cost_change = park[1];
        it.gradient_max_norm = it_gmax;
        it.step_norm = park[2];
        it.relative_decrease = park[3];
        it.trust_region_radius = radius;
        trace[n_trace_in] = it;
      }
      s.phase = candidate_ready ? 1 : (FIRST ? 0 : 1);
      s.iteration = candidate_ready ? it_iteration + 1 : iteration;
      s.n_invalid = candidate_ready ? 0 : n_invalid_in;
      s.reuse_diagonal = cont ? 1 : reuse;
      s.gmax = gmax;
      if (!LEAN) {  // (LEAN: stored as soon as they were known)
        s.num_successful = n_succ;
        s.num_unsuccessful = n_unsucc;
        s.n_trace = n_trace_in + 1;
        s.n_evals = n_evals;
        s.initial_cost = initial_cost;
        s.minimum_cost = minimum_cost;
        s.min_iter_cost = min_iter_cost;
        if (xout_dirty) {
#pragma unroll
          for (int i = 0; i < NA; ++i) s.x_out[i] = x[i];
        }
      }
      if (step_dirty) {
#pragma unroll
        for (int a = 0; a < NP; ++a) s.step[a] = step_n[a];
        s.model_cost_change = mcc_n;
      }
    }
    if (lane < NP) s.diag[lane] = cont ? diag_n : diag;
  };
  const bool slow_path = early == CLC_RUNNING && invalid_step;
  if (lane == 0) {
    s.status = early != CLC_RUNNING ? early : status;
    if (early != CLC_RUNNING) s.n_evals = n_evals;  // terminated by a tolerance on the pass itself: nothing else changes
  }
  if (lane == 1 && early == CLC_RUNNING && candidate_ready) {
#pragma unroll
    for (int i = 0; i < NA; ++i) s.x_eval[i] = cand[i];
  }
  if (slow_path) {
    // rare: HandleInvalidStep and whatever follows it (shrunken radius, another step, ...) on the serial controller,
    // which works on the complete state in LDS and decides status and candidate — before anybody leaves
    write_back();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      clc_iteration it;
      it.iteration = it_iteration + 1; it.step_is_valid = 0; it.step_is_successful = 0; it.pad_ = 0;
      it.cost = 0.0; it.cost_change = 0.0; it.gradient_max_norm = 0.0; it.step_norm = 0.0;
      it.relative_decrease = 0.0; it.trust_region_radius = 0.0;
      LmScratch& w = *reinterpret_cast<LmScratch*>(park + 32);  // temporaries in LDS: this path must not cost registers
      lm_iterate(s, w, o, trace, trace_cap, it, true);
    }
  }
  CLC_CK();
  __syncthreads();  // pairs with the barrier the other waves of the workgroup execute in lm_tail_after_barrier
  if (early == CLC_RUNNING && !slow_path) write_back();
  CLC_CK();
#ifdef CLC_STAMPS
  if (stamp_row && lane == 0 && nck == 10) {
    unsigned long long packed0 = 0, packed1 = 0;
    for (int i = 0; i < 4; ++i) packed0 |= (unsigned long long)((ck[i + 1] - ck[i]) & 0xFFFF) << (16 * i);
    for (int i = 0; i < 4; ++i) packed1 |= (unsigned long long)((ck[i + 5] - ck[i + 4]) & 0xFFFF) << (16 * i);
    stamp_row[15] = packed0;
    stamp_row[6] = packed1;
  }
#endif
}

// `state` is where the LM state is read from; it is written back to `state_out` (nullptr: not at all — the
// step kernel's non-leading workgroups run the controller redundantly and keep the result in LDS only).
// CHECK_STATUS: the staged state is inspected before anything is consumed or published; if the solve had already
// terminated the function returns false right after the first barrier (state staged in LDS, nothing else done).
// `red` is [LM_GROUPS][32] doubles of LDS.
// Phase A of the tail: this thread's share of the row sums -> LDS, its word of the LM state -> LDS.  No barrier: a caller
// may issue further loads (the step kernel: its first rows of points) between this and lm_tail_finish.
template <bool COHERENT, int HT>
__device__ __forceinline__ void lm_tail_sums(const double* __restrict__ partials, int n_blocks, double (*red)[32],
                                             double* sh_state, const LmLoads& L, long long* stamps /* nullable: [2] */,
                                             const int n_stage_words = LM_STATE_WORDS) {
  static_assert(LM_STATE_WORDS <= 256, "one state word per thread");
  const bool helper = threadIdx.x < HT;
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  if (helper) {  // wave-uniform
    const int cc = c < NACC ? c : NACC - 1;
    if (stamps && threadIdx.x == 0) stamps[0] = clock64();
    if ((int)threadIdx.x < n_stage_words) sh_state[threadIdx.x] = L.my_word;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s0 += (c < NACC && g + LM_GROUPS * j < n_blocks) ? L.v[j] : 0.0;
    if (HT == 256) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s1 += (c < NACC && g + 8 + LM_GROUPS * j < n_blocks) ? L.v[16 + j] : 0.0;
    }
    if (stamps && threadIdx.x == 0) stamps[1] = clock64();
    for (int b0 = 256; b0 < n_blocks; b0 += 256) {  // grids beyond 256 workgroups: further rounds of 256 rows
      const double* bb = partials + (size_t)(b0 + g) * NACC + cc;
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = load_partial<COHERENT>(bb + (size_t)(LM_GROUPS * j) * NACC);
#pragma unroll
      for (int j = 0; j < 16; ++j) s0 += (c < NACC && b0 + g + LM_GROUPS * j < n_blocks) ? v[j] : 0.0;
      if (HT == 256) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = load_partial<COHERENT>(bb + (size_t)(8 + LM_GROUPS * j) * NACC);
#pragma unroll
        for (int j = 0; j < 16; ++j) s1 += (c < NACC && b0 + g + 8 + LM_GROUPS * j < n_blocks) ? v[j] : 0.0;
      }
    }
    red[g][c] = s0;
    if (HT == 256) red[g + 8][c] = s1;
  }
}

template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS, bool WAVE = false>
__device__ __forceinline__ bool lm_tail_after_barrier(const LmState* __restrict__ state, LmState* __restrict__ state_out,
                                                      const clc_options& opt, clc_iteration* __restrict__ trace, int trace_cap,
                                                      HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                                      const long long c0, const Pose7* init_pose, const LmLoads& L,
                                                      unsigned long long* stamp_row = nullptr);

// Phase B: barrier, ordered combination of the 16 row groups, LM controller, publication.
template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS = false>
__device__ __forceinline__ bool lm_tail_finish(const LmState* __restrict__ state, LmState* __restrict__ state_out,
                                               const clc_options& opt,
                                               clc_iteration* __restrict__ trace, int trace_cap,
                                               HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                               const long long c0, const Pose7* init_pose, const LmLoads& L) {
  __syncthreads();
  return lm_tail_after_barrier<COHERENT, FIRST, HT, CHECK_STATUS>(state, state_out, opt, trace, trace_cap, mailbox, red, sh_state,
                                                                  c0, init_pose, L);
}

// ... and what follows the barrier (the step kernel reads its options from LDS between the two).
template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS, bool WAVE>
__device__ __forceinline__ bool lm_tail_after_barrier(const LmState* __restrict__ /*state*/, LmState* __restrict__ state_out,
                                                      const clc_options& opt,
                                                      clc_iteration* __restrict__ trace, int trace_cap,
                                                      HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                                      const long long c0, const Pose7* init_pose, const LmLoads& L,
                                                      unsigned long long* stamp_row /* debug builds; nullptr otherwise */) {
  // Called by every thread of the workgroup (it contains a barrier).
  // Progress for the host's launch-ahead metering is published EARLY, by the first lane of wave 1
  // (not the controller's wave): the ~1.5 us a store to pinned host memory needs to be
  // acknowledged then overlaps the controller instead of delaying the end of the launch.
  const long long passes_before = L.passes_before;
  if (CHECK_STATUS && !FIRST && reinterpret_cast<const LmState*>(sh_state)->status != CLC_RUNNING) return false;
  // the 16 row groups are combined in order by 28 lanes in parallel (one column each): done by the controller's
  // lane alone this was hundreds of serial FP64 adds behind LDS reads, ~0.4 us of the launch
  if (threadIdx.x < 32) {
    double t = 0.0;
#pragma unroll
    for (int gg = 0; gg < LM_GROUPS; ++gg) t += red[gg][threadIdx.x];
    red[0][threadIdx.x] = t;  // row 0 now holds the totals
    if (opt.profile_events && threadIdx.x == 0 && mailbox != nullptr) mailbox->prof[6] = clock64();
  }
  // no workgroup barrier here: the 28 lanes above and the controller's lane below are the same wave, whose LDS
  // operations execute in program order; the other waves go straight to the barrier at the end
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (threadIdx.x == 64 && mailbox != nullptr)
    __hip_atomic_store(&mailbox->n_done, (int32_t)(passes_before + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (WAVE) {
    // The controller on all 64 lanes of wave 0 (lm_advance_wave).  It publishes the next point / the status, meets the
    // other waves at ONE workgroup barrier — from which they leave for their rows — and only then writes the rest of
    // the state back, publishes a termination to the host and (leading workgroup) copies the state to device memory.
    if (threadIdx.x < 64) {
      const long long c1 = clock64();
      if (stamp_row && threadIdx.x == 0) stamp_row[11] = wall_clock64();
      LmState& st = *reinterpret_cast<LmState*>(sh_state);
      if (FIRST) {  // first iteration of a solve: nothing to load
        if (threadIdx.x == 0) lm_init(st, opt, init_pose->v);
        // lane 0's stores must be visible to the other lanes' loads below: without the fences hipcc is free to hoist those
        // loads above the (for them never executed) stores — they then read the previous solve's terminated state
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      static_assert(sizeof(LmScratch) <= (LM_GROUPS - 2) * 32 * sizeof(double), "LmScratch fits rows 2.. of red");
      lm_advance_wave<FIRST>(st, opt, trace, trace_cap, &red[0][0], &red[1][0], (int)threadIdx.x, stamp_row);  // contains the barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (threadIdx.x == 0) {
        const long long c2 = clock64();
        if (stamp_row) { stamp_row[12] = wall_clock64(); stamp_row[13] = (unsigned long long)c1; stamp_row[14] = (unsigned long long)c2; }
        if (mailbox != nullptr) {
          if (opt.profile_events) { mailbox->prof[0] = c0; mailbox->prof[1] = c1; mailbox->prof[2] = c2; }
          if (st.status != CLC_RUNNING) {
            // termination: payload first, then system-scope release stores of the flags
            clc_summary sm;
            lm_fill_summary(st, sm);
            sm.solve_ms = 0.0;
            sm.eval_kernel_ms = 0.0;
            sm.eval_kernel_launches = 0;
            mailbox->summary = sm;
            for (int i = 0; i < 7; ++i) mailbox->pose[i] = st.x_out[i];
            __hip_atomic_store(&mailbox->n_done, (int32_t)st.n_evals, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mailbox->status, st.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          if (opt.profile_events) mailbox->prof[3] = clock64();
        }
      }
      if (state_out != nullptr) {
        for (int w = threadIdx.x; w < LM_STATE_WORDS; w += 64) reinterpret_cast<double*>(state_out)[w] = sh_state[w];
      }
    } else {
      __syncthreads();  // the barrier inside lm_advance_wave
    }
    return true;
  }
  if (threadIdx.x == 0) {
    const long long c1 = clock64();
    LmState& st = *reinterpret_cast<LmState*>(sh_state);
    if (stamp_row) stamp_row[11] = wall_clock64();
    double tot[NACC];
#pragma unroll
    for (int cc = 0; cc < NACC; ++cc) tot[cc] = red[0][cc];
    // The LM state is used in place in LDS: copied into registers and back it cost 256 VGPRs + 48 AGPRs
    // (occupancy 1 for the fused kernel); in place 148-162, at the same controller time.
    if (FIRST) lm_init(st, opt, init_pose->v);  // first iteration of a solve: nothing to load
    LmScratch scratch;
    lm_advance(st, scratch, opt, trace, trace_cap,
               finalize_cost(tot[27], opt.use_loss != 0, opt.loss_scale_factor), tot + 21, tot);
    const long long c2 = clock64();
    if (stamp_row) { stamp_row[12] = wall_clock64(); stamp_row[13] = (unsigned long long)c1; stamp_row[14] = (unsigned long long)c2; }
    if (mailbox != nullptr) {
      if (opt.profile_events) { mailbox->prof[0] = c0; mailbox->prof[1] = c1; mailbox->prof[2] = c2; }
      if (st.status != CLC_RUNNING) {
        // termination: payload first, then system-scope release stores of the flags
        clc_summary sm;
        lm_fill_summary(st, sm);
        sm.solve_ms = 0.0;
        sm.eval_kernel_ms = 0.0;
        sm.eval_kernel_launches = 0;
        mailbox->summary = sm;
        for (int i = 0; i < 7; ++i) mailbox->pose[i] = st.x_out[i];
        __hip_atomic_store(&mailbox->n_done, (int32_t)st.n_evals, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox->status, st.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (opt.profile_events) mailbox->prof[3] = clock64();
    }
  }
  __syncthreads();
  if (state_out != nullptr && threadIdx.x < LM_STATE_WORDS)
    reinterpret_cast<double*>(state_out)[threadIdx.x] = sh_state[threadIdx.x];
  return true;
}

// Both phases back to back (lm_kernel, eval_lm_kernel).
template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS = false>
__device__ __forceinline__ bool lm_tail(const double* __restrict__ partials, int n_blocks,
                                        const LmState* __restrict__ state, LmState* __restrict__ state_out,
                                        const clc_options& opt,
                                        clc_iteration* __restrict__ trace, int trace_cap,
                                        HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                        const long long c0, const Pose7* init_pose, LmLoads& L) {
  lm_tail_sums<COHERENT, HT>(partials, n_blocks, red, sh_state, L,
                             (opt.profile_events && mailbox != nullptr) ? &mailbox->prof[4] : nullptr);
  return lm_tail_finish<COHERENT, FIRST, HT, CHECK_STATUS>(state, state_out, opt, trace, trace_cap, mailbox, red, sh_state, c0,
                                                           init_pose, L);
}

template <bool FIRST>
__global__ __launch_bounds__(BLOCK) void lm_kernel(const double* __restrict__ partials,
                                                   int n_blocks, LmState* __restrict__ state,
                                                   const clc_options opt,
                                                   clc_iteration* __restrict__ trace,
                                                   int trace_cap, HostMailbox* mailbox, const Pose7 pose0) {
  __shared__ double red[LM_GROUPS][32];
  __shared__ double sh_state[LM_STATE_WORDS];
  const long long c0 = clock64();
  if (!FIRST && state->status != CLC_RUNNING) return;
  LmLoads L;
  lm_issue_loads<false, FIRST, BLOCK>(partials, state, L);
  lm_tail<false, FIRST, BLOCK>(partials, n_blocks, state, state, opt, trace, trace_cap, mailbox, red, sh_state, c0, &pose0, L);
}

// ---------------------------------------------------------------------------------------
// K1+K2 fused — evaluation launch whose LAST-ARRIVING workgroup runs the reduction + LM
// controller, so one LM iteration is ONE launch.  Inter-workgroup hand-off (placement
// independent, MI355X per-XCD L2s are not coherent):
//   producer: partial row stored write-through (agent-scope relaxed atomic stores = sc1),
//             every storing wave drains vmcnt(0), workgroup barrier, ONE lane takes a ticket
//             with a relaxed agent-scope fetch_add;
//   consumer: the workgroup that draws ticket == gridDim-1 reads all rows with agent-scope
//             (sc1) loads, which bypass its CU's L1 — no stale lines possible.
// The ticket counter is reset by the last workgroup (all others have already arrived) and is
// zeroed by lm_init_kernel before the first launch of a solve.
// ---------------------------------------------------------------------------------------
template <bool WITH_LOSS, bool NT, bool COMPACT, bool DEEP, int BT>
__global__ __launch_bounds__(BT) void eval_lm_kernel(const double* __restrict__ tiles,
                                                     const double* __restrict__ groups, const long long n,
                                                     LmState* __restrict__ state, const clc_options opt,
                                                     double* __restrict__ partials,
                                                     unsigned int* __restrict__ ticket_counter,
                                                     clc_iteration* __restrict__ trace, int trace_cap,
                                                     HostMailbox* mailbox) {
  __shared__ double red[LM_GROUPS][32];
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ double wsum[BT / 64][NACC];
  __shared__ int sh_last;
  auto get_pose = [&](PoseU& P) -> bool {
    const int32_t st = state->status;  // issued together with the pose loads: one wait
    load_pose(state->x_eval, P);
    return st == CLC_RUNNING;
  };
  const double lf = opt.loss_scale_factor;
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const WaveMap wm = make_wave_map<BT>(blockIdx.x, gridDim.x, wave);
  bool active;
  if (COMPACT && DEEP)
    active = stream_ctiles_deep<WITH_LOSS, true, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else if (COMPACT)
    active = stream_ctiles<WITH_LOSS, true, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else
    active = stream_tiles<WITH_LOSS, true, true, NT>(tiles, n, wm, lane, get_pose, inv_lf2, acc);
  if (!active) return;  // uniform over the launch
  wave_reduce_butterfly(acc, wsum[wave], lane);
  __syncthreads();
  // ---- publish this workgroup's partial row (write-through) and take a ticket ----
  if (threadIdx.x < NACC) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) s += wsum[w][threadIdx.x];
    __hip_atomic_store(partials + (size_t)blockIdx.x * NACC + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every (storing) wave: stores acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!sh_last) return;
  // ---- last-arriving workgroup: every other row is complete and visible at agent scope ----
  const long long c0 = clock64();
  if (threadIdx.x == 0) __hip_atomic_store(ticket_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  LmLoads L;
  lm_issue_loads<true, false, BT>(partials, state, L);
  lm_tail<true, false, BT>(partials, (int)gridDim.x, state, state, opt, trace, trace_cap, mailbox, red, sh_state, c0, nullptr, L);
}

// ---------------------------------------------------------------------------------------
// Step kernel — ONE launch per LM iteration, no second kernel, no inter-workgroup hand-off.
// Launch k evaluates at the point x_k and leaves one partial row per workgroup; launch k+1 starts
// by letting EVERY workgroup read those rows and run the controller redundantly (same inputs, same
// instruction stream: bit-identical x_{k+1} everywhere), then streams its tiles at x_{k+1}.  That
// removes the controller's own launch (launch boundary + kernel-argument fetch + its barrier
// skew, ~4 us of the ~16 us an iteration took at 10^6 observations) and overlaps the controller
// with the first tile loads, which are issued before it (the streaming loops call `get_pose` after
// their prologue loads).  Workgroup 0 alone writes the advanced state, the trace and the host
// mailbox.  LM state and partial rows are double-buffered by launch parity, because a workgroup of
// launch k+1 may still be reading what another one is already overwriting.
//   MODE 0: first launch of a solve (pose by value, nothing to consume)
//   MODE 1: second launch (controller initialises the LM state: lm_init + first rows)
//   MODE 2: steady state
// A launch that finds the solve already terminated (queued ahead by the host) copies the state forward, so that
// the launch queued behind it sees the termination too, and exits without touching the host mailbox.
// ---------------------------------------------------------------------------------------
// Per-solve constants of the step-kernel chain in DEVICE memory.  Kernel arguments beyond the first 16 dwords are fetched
// from the kernarg segment (host-coherent memory: a ~1 us round trip) before the first instruction that needs one of them
// can run; the steady-state launches therefore take ONLY preloadable arguments (8 pointers / two ints: the
// -amdgpu-kernarg-preload-count build preloads them into SGPRs while the wave is dispatched), the grid size comes as an
// explicit argument rather than from the implicit kernarg block, and everything else — the solver options, the start
// pose, where the trace and the host mailbox live — is read from this block, which launch 0 of the solve (MODE 0: it has
// them as ordinary by-value arguments) leaves behind in device memory.
struct SolveParams {
  clc_options opt;
  Pose7 pose0;
  clc_iteration* trace;
  HostMailbox* mailbox;
  int32_t trace_cap;
  int32_t pad_;
};

// The per-solve block and the two LM state buffers (double-buffered by launch parity) live in one allocation, so one
// pointer argument reaches all three: user SGPRs hold 16 dwords, two of them the kernarg segment pointer — 14 dwords of
// preloaded arguments is all a kernel gets.
struct SolveBlock {
  SolveParams prm;
  LmState st[2];
};

constexpr int PRM_WORDS = (int)(sizeof(SolveParams) / 8);
static_assert(sizeof(SolveParams) % 8 == 0 && offsetof(SolveBlock, st) == sizeof(SolveParams), "SolveBlock is params, then states");
static_assert(LM_STATE_WORDS + PRM_WORDS <= 256, "one block word per thread");

#ifdef CLC_STAMPS
// Debug build only (scripts/r02_stamps.py): wall-clock (100 MHz) stamps of every workgroup of the first 64 launches.
constexpr int STAMP_LAUNCHES = 64, STAMP_WGS = 512, STAMP_SLOTS = 16;
__device__ unsigned long long clc_stamp_buf[STAMP_LAUNCHES][STAMP_WGS][STAMP_SLOTS];
#define CLC_STAMP(slot, tid)                                                                                        \
  do {                                                                                                              \
    if (threadIdx.x == (tid) && blockIdx.x < STAMP_WGS && launch_index < STAMP_LAUNCHES)                            \
      clc_stamp_buf[launch_index][blockIdx.x][slot] = wall_clock64();                                               \
  } while (0)
#define CLC_STAMP_ROW (blockIdx.x < STAMP_WGS && launch_index < STAMP_LAUNCHES ? &clc_stamp_buf[launch_index][blockIdx.x][0] : nullptr)
#else
#define CLC_STAMP(slot, tid) do {} while (0)
#define CLC_STAMP_ROW nullptr
#endif

// LAYOUT 0: compact tiles (ctiles + group table, n = observations); 1: row layout (ctiles = xy rows, groups = row
// descriptors, n = rows; DEEP selects non-temporal loads, WEIGHTED the 3:2 old/young wave shares).
template <bool WITH_LOSS, bool DEEP, int MODE, int LAYOUT = 0, bool WEIGHTED = true>
__global__ __launch_bounds__(512) void step_kernel(const double* __restrict__ rows_in,
                                                   const double* __restrict__ ctiles,
                                                   const double* __restrict__ groups, const int n, const int grid_parity,
                                                   const int launch_index, double* __restrict__ rows_out,
                                                   SolveBlock* __restrict__ blk, const SolveParams init) {
  // Preloaded: rows_in, ctiles, groups, n, grid_parity, launch_index, rows_out, blk = 13 dwords.  `init` is read by MODE 0
  // only (the other instantiations never touch it, so they never wait for the kernarg segment).
  // grid_parity = number of workgroups | (launch index & 1) << 30: this launch reads LM state st[parity ^ 1], writes st[parity].
  // The front of a steady-state launch is ONE memory round trip: every thread issues, back to back, its 16 partial-row
  // loads and its word of {LM state, per-solve parameters} (vector loads; nothing waits on a scalar load), then the
  // waves sum their rows, issue their first rows of points, and meet at the barrier in front of the controller.
  __shared__ double red[LM_GROUPS][32];
  __shared__ double sh_state[LM_STATE_WORDS + PRM_WORDS];  // [0, LM_STATE_WORDS): LM state; then the SolveParams words
  const long long c0 = clock64();
  CLC_STAMP(0, 0);
  const bool leader = blockIdx.x == 0;
  const int grid = grid_parity & 0x3FFFFFFF;
  const int parity = (grid_parity >> 30) & 1;
  LmState* __restrict__ state_out = &blk->st[parity];
  LmLoads L;
  if (MODE != 0) {
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    // word t of the staging area: state word t (t < LM_STATE_WORDS) or parameter word t - LM_STATE_WORDS
    const int t = threadIdx.x < LM_STATE_WORDS + PRM_WORDS ? (int)threadIdx.x : LM_STATE_WORDS + PRM_WORDS - 1;
    const int w = t < LM_STATE_WORDS ? PRM_WORDS + LM_STATE_WORDS * (parity ^ 1) + t : t - LM_STATE_WORDS;
    L.my_word = reinterpret_cast<const double*>(blk)[w];
    L.passes_before = launch_index - 1;  // launch k consumes pass k - 1: no need to read the counter back
    const int cc = c < NACC ? c : NACC - 1;
    const double* base = rows_in + (size_t)g * NACC + cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) L.v[j] = base[(size_t)(LM_GROUPS * j) * NACC];
    // nothing below may be scheduled in between the loads above (hipcc had pulled the first row add — and its
    // s_waitcnt — in front of the last four loads: a second memory round trip on the critical path)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MODE == 0 && leader && threadIdx.x == 0) blk->prm = init;  // for launches 1, 2, ... of this solve
  const SolveParams* prm = reinterpret_cast<const SolveParams*>(sh_state + LM_STATE_WORDS);  // valid after the barrier
  double inv_lf2 = 0.0;  // set by get_pose: it depends on the options, which arrive with the state
  auto get_pose = [&](PoseU& P) -> bool {
    if (MODE == 0) {
      load_pose(init.pose0.v, P);
      inv_lf2 = make_uniform(1.0 / (init.opt.loss_scale_factor * init.opt.loss_scale_factor));
      return true;
    }
    CLC_STAMP(1, 0);
    __syncthreads();  // state, parameters and row-group sums of every wave are in LDS
    CLC_STAMP(2, 0);
    const clc_options& opt = prm->opt;
    inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
    const bool consumed = lm_tail_after_barrier<false, MODE == 1, 512, true, true>(
        nullptr, leader ? state_out : nullptr, opt, leader ? prm->trace : nullptr, leader ? prm->trace_cap : 0,
        leader ? prm->mailbox : nullptr, red, sh_state, c0, &prm->pose0, L, CLC_STAMP_ROW);
    if (!consumed) {
      // the solve had terminated before this launch: hand the state on (the launch queued behind this one reads
      // the other buffer) and leave; the host mailbox is NOT touched — it may already belong to the next solve
      if (leader && threadIdx.x < LM_STATE_WORDS) reinterpret_cast<double*>(state_out)[threadIdx.x] = sh_state[threadIdx.x];
      return false;
    }
    CLC_STAMP(3, 0);
    const LmState* st = reinterpret_cast<const LmState*>(sh_state);
    const bool running = st->status == CLC_RUNNING;
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = st->x_eval[i];
    load_pose(x, P);
    return running;
  };
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<512, WEIGHTED>(blockIdx.x, grid, threadIdx.x >> 6);
  // Order of the memory traffic of a launch: every wave's share of the previous launch's rows FIRST (57 KB per
  // workgroup through a 64 B/clk vector cache: the controller waits for the slowest wave's rows), and only when a wave
  // has summed its rows does it issue its first rows of points — those have the whole controller to arrive.
  if (MODE != 0) lm_tail_sums<false, 512>(rows_in, grid, red, sh_state, L, nullptr, LM_STATE_WORDS + PRM_WORDS);
  bool active;
  if (LAYOUT == 1) {
    const RowDesc* desc = reinterpret_cast<const RowDesc*>(groups);
    long long r0 = wm.begin(n), r1 = wm.end(n);
    if (!WEIGHTED) {  // equal shares, boundaries at scan starts (wave_split_kernel)
      const WaveRun run = wave_run(desc, n, blockIdx.x, threadIdx.x >> 6);
      r0 = run.begin;
      r1 = run.end;
    }
    active = stream_rows<WITH_LOSS, DEEP>(ctiles, desc, r0, r1, lane, get_pose, inv_lf2, acc);
  }
  else if (DEEP) active = stream_ctiles_deep<WITH_LOSS, true, false>(ctiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else active = stream_ctiles<WITH_LOSS, true, false>(ctiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  if (!active) return;  // the controller terminated the solve: nothing to evaluate
  CLC_STAMP(4, 0);
  CLC_STAMP(8, 448);
#ifdef CLC_STAMPS
  {  // block_reduce_store<8>, stamped
    __shared__ double wsum_dbg[8][NACC];
    wave_reduce_butterfly(acc, wsum_dbg[threadIdx.x >> 6], lane);
    CLC_STAMP(9, 0);
    __syncthreads();
    CLC_STAMP(10, 0);
    if (threadIdx.x < NACC) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += wsum_dbg[w][threadIdx.x];
      rows_out[(size_t)blockIdx.x * NACC + threadIdx.x] = s;
    }
  }
#else
  block_reduce_store<8>(acc, 0, rows_out + (size_t)blockIdx.x * NACC);
#endif
  CLC_STAMP(5, 0);
  CLC_STAMP(7, 448);
}

__global__ void lm_init_kernel(LmState* __restrict__ state, const clc_options opt, const Pose7 pose0,
                               unsigned int* __restrict__ ticket_counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *ticket_counter = 0u;
    LmState s;
    lm_init(s, opt, pose0.v);
    *state = s;
  }
}

// ---------------------------------------------------------------------------------------
// plug-in level kernels (element-wise parity with the reference's Ceres callbacks)
// ---------------------------------------------------------------------------------------
// PointInPlaneFactor::Evaluate per record, literal operation order of
// src/LaseCamCalCeres.cpp:43-66 (pt_c = R p + t; r = s (n.pt_c + d); J = s [n, n^T(-R [p]x), 0]).
__global__ void factor_kernel(const double* __restrict__ tiles, long long n,
                              const double* __restrict__ pose, double* __restrict__ residuals,
                              double* __restrict__ jac7) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* tb = tiles + (k / TILE) * TILE_DOUBLES + (k % TILE);
  const double nx = tb[0], ny = tb[TILE], nz = tb[2 * TILE], d = tb[3 * TILE];
  const double p[3] = {tb[4 * TILE], tb[5 * TILE], tb[6 * TILE]};
  const double s = tb[7 * TILE];
  double x[7], R[9];
  for (int i = 0; i < 7; ++i) x[i] = pose[i];
  quat_to_rot(x + 3, R);
  double ptc[3];
  for (int i = 0; i < 3; ++i)
    ptc[i] = ((R[3 * i] * p[0] + R[3 * i + 1] * p[1]) + R[3 * i + 2] * p[2]) + x[i];
  residuals[k] = s * (((nx * ptc[0] + ny * ptc[1]) + nz * ptc[2]) + d);
  if (jac7 != nullptr) {
    const double S[9] = {0.0, -p[2], p[1], p[2], 0.0, -p[0], -p[1], p[0], 0.0};
    double* j = jac7 + 7 * k;
    j[0] = s * nx;
    j[1] = s * ny;
    j[2] = s * nz;
    for (int c = 0; c < 3; ++c) {
      double M[3];
      for (int i = 0; i < 3; ++i)
        M[i] = ((-R[3 * i]) * S[c] + (-R[3 * i + 1]) * S[3 + c]) + (-R[3 * i + 2]) * S[6 + c];
      j[3 + c] = s * ((nx * M[0] + ny * M[1]) + nz * M[2]);
    }
    j[6] = 0.0;
  }
}

// PoseLocalParameterization::Plus, one thread per (x, delta) pair.
__global__ void plus_kernel(const double* __restrict__ x, const double* __restrict__ delta,
                            double* __restrict__ out, long long n) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double a[7], d[6], o[7];
  for (int i = 0; i < 7; ++i) a[i] = x[7 * k + i];
  for (int i = 0; i < 6; ++i) d[i] = delta[6 * k + i];
  pose_plus(a, d, o);
  for (int i = 0; i < 7; ++i) out[7 * k + i] = o[i];
}

// Debug/test entry: run only the wave reduction on caller-provided lane values
// in[64][28] -> out[28].
__global__ void wave_reduce_test_kernel(const double* __restrict__ in, double* __restrict__ out,
                                        int reduce_mode) {
  const int lane = threadIdx.x & 63;
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = in[lane * NACC + k];
  __shared__ double o[NACC];
  if (reduce_mode == 0)
    wave_reduce_butterfly(acc, o, lane);
  else
    wave_reduce_shuffle(acc, o, lane);
  __syncthreads();
  if (threadIdx.x < NACC) out[threadIdx.x] = o[threadIdx.x];
}

// ---------------------------------------------------------------------------------------
// K5 — 9x9 normal equation of the closed-form initialiser (LaseCamCalCeres.cpp:144-161).
// Row A_k = kron([x, y, 1], n), b_k = -d.  A^T A = sum kron(bb^T, nn^T): 6 x 6 unique
// products, A^T b: 9.  45 accumulators per lane; same streaming/reduction shape as K1.
// Output per block: 45 doubles: [bb(6: xx xy x yy y 1)][nn(6: 00 01 02 11 12 22)] then 9.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void accumulate_normal9(double nx, double ny, double nz, double d,
                                                   double x, double y, double (&acc)[NACC9]) {
  const double nn[6] = {nx * nx, nx * ny, nx * nz, ny * ny, ny * nz, nz * nz};
  const double bb[6] = {x * x, x * y, x, y * y, y, 1.0};
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[6 * i + j] = fma(bb[i], nn[j], acc[6 * i + j]);
  const double md = -d;
  const double bv[3] = {x, y, 1.0};
  const double nv[3] = {nx, ny, nz};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[36 + 3 * i + j] = fma(bv[i] * nv[j], md, acc[36 + 3 * i + j]);
}

__global__ __launch_bounds__(BLOCK) void normal9_kernel(const double* __restrict__ tiles,
                                                        const long long n,
                                                        double* __restrict__ partials) {
  double acc[NACC9];
#pragma unroll
  for (int i = 0; i < NACC9; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const long long wave_global = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * (BLOCK / 64);
  const long long n_tiles = (n + TILE - 1) / TILE;
  for (long long tile = wave_global; tile < n_tiles; tile += n_waves) {
    const double2* base = reinterpret_cast<const double2*>(tiles + tile * TILE_DOUBLES) + lane;
    double2 f[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] = base[k * 64];
    const long long k0 = tile * TILE + 2 * lane;
    if (k0 < n) accumulate_normal9(f[0].x, f[1].x, f[2].x, f[3].x, f[4].x, f[5].x, acc);
    if (k0 + 1 < n) accumulate_normal9(f[0].y, f[1].y, f[2].y, f[3].y, f[4].y, f[5].y, acc);
  }
  __shared__ double wsum[BLOCK / 64][NACC9];
#pragma unroll
  for (int k = 0; k < NACC9; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) wsum[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC9) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += wsum[w][threadIdx.x];
    partials[(size_t)blockIdx.x * NACC9 + threadIdx.x] = s;
  }
}

// Fixed-order sum of the 45-column block partials.  Thread (c, rg) sums rows rg, rg + 4, ... of column c with 16 independent
// loads in flight per round (a dependent load chain over 256 rows cost 16 us here — more than K5 itself), then the four
// row groups are combined in order.
__global__ __launch_bounds__(BLOCK) void reduce9_kernel(const double* __restrict__ partials,
                                                        int n_blocks, double* __restrict__ out) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  constexpr int UNROLL = 16;
  double s = 0.0;
  if (c < NACC9) {
    for (int b0 = rg; b0 < n_blocks; b0 += 4 * UNROLL) {
      double v[UNROLL];
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) {
        const int b = b0 + 4 * j;
        v[j] = (b < n_blocks) ? partials[(size_t)b * NACC9 + c] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) s += v[j];
    }
  }
  red[rg][c] = s;
  __syncthreads();
  if (threadIdx.x < NACC9)
    out[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// K5 on the row layout: 16 B/point instead of 48 of the 64-byte tiles, 5 FP64 instructions per point.
// Wave reduction of the 45 accumulators: 45 -> 23 registers with one permlane32 swap round, then xor-shuffles.
template <bool NT>
__global__ __launch_bounds__(BLOCK) void normal9_rows_kernel(const double* __restrict__ xy, const RowDesc* __restrict__ desc,
                                                             const long long n_rows, double* __restrict__ partials) {
  double acc[NACC9];
#pragma unroll
  for (int i = 0; i < NACC9; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<BLOCK>(blockIdx.x, gridDim.x, threadIdx.x >> 6);
  Normal9Rows pol;
  stream_rows_policy<Normal9Rows, NT, ROWS_DEPTH>(pol, xy, desc, wm.begin(n_rows), wm.end(n_rows), lane,
                                      [](PoseU&) { return true; }, acc);
  __shared__ double wsum[BLOCK / 64][NACC9 + 1];
  // halves: after the swap, lanes 0-31 hold acc[i] of {l, l+32} summed, lanes 32-63 acc[i+23]
  double r[23];
#pragma unroll
  for (int i = 0; i < 23; ++i) {
    double x = acc[i], y = (i + 23 < NACC9) ? acc[i + 23] : 0.0;
    swap_halves(x, y);
    r[i] = x + y;
  }
#pragma unroll
  for (int i = 0; i < 23; ++i) {
    double v = r[i];
    v += dpp_read<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_read<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_read<0x141>(v);  // row_half_mirror
    v += dpp_read<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
    v += __shfl_xor(v, 16, 64);  // the two rows of each half
    if (lane == 0) wsum[threadIdx.x >> 6][i] = v;
    if (lane == 32 && i + 23 < NACC9 + 1) wsum[threadIdx.x >> 6][i + 23] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC9) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += wsum[w][threadIdx.x];
    partials[(size_t)blockIdx.x * NACC9 + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------
// K4 — batched independent problems (lockstep): every LM iteration is one launch of
// batched_eval_kernel (blocks_per_problem workgroups stream each still-running problem at
// its own candidate pose) followed by batched_lm_kernel (one thread per problem runs the LM
// controller).  Problems that have terminated cost nothing in later launches.  Problem k
// owns tiles [tile_off[k], tile_off[k+1]) (padded to whole tiles) and n_obs[k] records; no
// communication between problems.
// ---------------------------------------------------------------------------------------
template <bool WITH_LOSS, bool COMPACT, bool NT, bool DEEP>
__global__ __launch_bounds__(BLOCK) void batched_eval_kernel(
    const double* __restrict__ tiles, const double* __restrict__ groups,
    const long long* __restrict__ tile_off, const long long* __restrict__ n_obs,
    const LmState* __restrict__ states, const int blocks_per_problem, const double lf,
    double* __restrict__ partials) {
  const int prob = blockIdx.x / blocks_per_problem;
  const int j = blockIdx.x - prob * blocks_per_problem;
  const LmState* st = states + prob;
  auto get_pose = [&](PoseU& P) -> bool {
    const int32_t s = st->status;  // issued together with the pose loads: one wait
    load_pose(st->x_eval, P);
    return s == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<BLOCK>(j, blocks_per_problem, threadIdx.x >> 6);
  bool active;
  if (COMPACT && DEEP)
    active = stream_ctiles_deep<WITH_LOSS, true, NT>(tiles + tile_off[prob] * CTILE_DOUBLES, groups, n_obs[prob], wm, lane,
                                                     get_pose, inv_lf2, acc);
  else if (COMPACT)
    active = stream_ctiles<WITH_LOSS, true, NT>(tiles + tile_off[prob] * CTILE_DOUBLES, groups, n_obs[prob], wm, lane,
                                                get_pose, inv_lf2, acc);
  else
    active = stream_tiles<WITH_LOSS, true, true, NT>(tiles + tile_off[prob] * TILE_DOUBLES, n_obs[prob], wm, lane,
                                                     get_pose, inv_lf2, acc);
  if (!active) return;  // uniform over the problem's workgroups: it has terminated
  block_reduce_store<BLOCK / 64>(acc, 0, partials + (size_t)blockIdx.x * NACC);
}

// K4 on the row layout: problem k owns rows [prob_row[k], prob_row[k+1]).
// K4 on the row layout: problem k owns rows [prob_row[k], prob_row[k+1]).
// BT = 64 (default): ONE WAVE per workgroup — no LDS, no barrier, the wave's 28 totals go straight to its own partial row
// (blocks_per_problem counts waves).  A problem of 10^4 observations is 160 rows: with 256-thread workgroups the dispatcher
// refills a CU only when the slowest of four waves has finished and every workgroup pays a barrier + LDS pass; with
// single-wave workgroups every wave slot is refilled the moment it frees (measured on one C4 shard: see DESIGN.md K4).
template <bool WITH_LOSS, bool NT, int BT>
__global__ __launch_bounds__(BT) void batched_rows_eval_kernel(
    const double* __restrict__ xy, const RowDesc* __restrict__ desc, const long long* __restrict__ prob_row,
    const LmState* __restrict__ states, const int blocks_per_problem, const double lf, double* __restrict__ partials) {
  const int prob = blockIdx.x / blocks_per_problem;
  const int j = blockIdx.x - prob * blocks_per_problem;
  const LmState* st = states + prob;
  auto get_pose = [&](PoseU& P) -> bool {
    const int32_t s = st->status;
    load_pose(st->x_eval, P);
    return s == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const long long r0 = prob_row[prob], r1 = prob_row[prob + 1];
  const WaveMap wm = make_wave_map<BT>(j, blocks_per_problem, threadIdx.x >> 6);
  if (!stream_rows<WITH_LOSS, NT>(xy, desc, r0 + wm.begin(r1 - r0), r0 + wm.end(r1 - r0), lane, get_pose, inv_lf2, acc)) return;
  if (BT == 64)
    wave_reduce_butterfly(acc, partials + (size_t)blockIdx.x * NACC, lane);
  else
    block_reduce_store<BT / 64>(acc, 0, partials + (size_t)blockIdx.x * NACC);
}

__global__ __launch_bounds__(64) void batched_init_kernel(LmState* __restrict__ states, const clc_options opt,
                                    const double* __restrict__ poses, int n_problems, unsigned int* __restrict__ active,
                                    unsigned int* __restrict__ ticket) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) { *active = 0u; *ticket = 0u; }  // the counters of batched_lm_kernel (two memset launches less per batch)
  if (p >= n_problems) return;
  double x[7];
  for (int i = 0; i < 7; ++i) x[i] = poses[7 * (size_t)p + i];
  LmState s;
  lm_init(s, opt, x);
  states[p] = s;
}

// One thread per problem: fixed-order sum of the problem's block partials, then the LM
// controller.  The workgroups count the problems that still need another evaluation; the
// last-arriving workgroup (ticket pattern, agent-scope atomics) publishes that count and the
// iteration index to the pinned host mailbox, so the host can run launch-ahead without
// blocking (status flips to non-zero when no problem is left running).
// Outcome of one problem: pose + summary to the (pinned, device-mapped) host arrays clc_solve_batched returns, and one
// clc_result_record in DEVICE memory, where clc_gather_results picks it up for the RCCL all-gather (global_index holds
// the local index here; the gather adds the shard's first global index).
__device__ __forceinline__ void batched_write_outcome(const LmState& s, int p, double* __restrict__ poses,
                                                      clc_summary* __restrict__ summaries, double* __restrict__ results) {
  for (int i = 0; i < 7; ++i) poses[7 * (size_t)p + i] = s.x_out[i];
  clc_summary sm;
  lm_fill_summary(s, sm);
  sm.solve_ms = 0.0;
  sm.eval_kernel_ms = 0.0;
  sm.eval_kernel_launches = 0;
  summaries[p] = sm;
  double* r = results + 12 * (size_t)p;
  for (int i = 0; i < 7; ++i) r[i] = s.x_out[i];
  r[7] = sm.final_cost;
  r[8] = sm.initial_cost;
  r[9] = (double)sm.num_iterations;
  r[10] = (double)(sm.termination == CLC_RUNNING ? CLC_FAILURE : sm.termination);
  r[11] = (double)p;
}

// A problem writes its outcome in the launch in which it terminates (no separate finish launch).
__global__ __launch_bounds__(64) void batched_lm_kernel(const double* __restrict__ partials, const int blocks_per_problem,
                                  LmState* __restrict__ states, const clc_options opt,
                                  const int n_problems, unsigned int* __restrict__ active,
                                  unsigned int* __restrict__ ticket, const int launch_index,
                                  HostMailbox* mailbox, double* __restrict__ poses, clc_summary* __restrict__ summaries,
                                  double* __restrict__ results) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool still_running = false;
  if (p < n_problems && states[p].status == CLC_RUNNING) {
    double tot[NACC];
    for (int c = 0; c < NACC; ++c) tot[c] = 0.0;
    for (int j = 0; j < blocks_per_problem; ++j) {
      const double* pp = partials + ((size_t)p * blocks_per_problem + j) * NACC;
      for (int c = 0; c < NACC; ++c) tot[c] += pp[c];
    }
    LmState s = states[p];
    LmScratch w;
    lm_advance(s, w, opt, nullptr, 0, finalize_cost(tot[27], opt.use_loss != 0, opt.loss_scale_factor), tot + 21, tot);
    states[p] = s;
    still_running = (s.status == CLC_RUNNING);
    if (!still_running) batched_write_outcome(s, p, poses, summaries, results);
  }
  if (still_running) __hip_atomic_fetch_add(active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {  // every other workgroup has added its count
      const unsigned int a = __hip_atomic_load(active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(active, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mailbox != nullptr) {
        __hip_atomic_store(&mailbox->n_done, (int32_t)(launch_index + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a == 0u) __hip_atomic_store(&mailbox->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// K4, one workgroup per problem (batches of about a thousand problems, C3): the WHOLE solve of a problem in one launch.
// The problem's four waves stream its rows at the candidate, reduce through LDS (block_reduce_store: the same order as
// the lockstep form's single partial row), wave 0 runs the wavefront controller on the state in LDS — the other waves
// start on the next pass at its barrier — until the controller stops.  No kernel boundary between LM iterations, no
// controller launches, no partial rows or LM states through memory, and no lockstep: a problem that needs three
// iterations leaves after three.  Same arithmetic as the lockstep form (bit-identical results).  The loop is bounded by
// the iteration cap whatever the controller does.
template <bool WITH_LOSS, bool NT>
__global__ __launch_bounds__(BLOCK, 2) void batched_solve_kernel(
    const double* __restrict__ xy, const RowDesc* __restrict__ desc, const long long* __restrict__ prob_row,
    const clc_options opt, double* __restrict__ poses, clc_summary* __restrict__ summaries, double* __restrict__ results) {
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ double sh_tot[32];
  __shared__ double sh_park[32 + (sizeof(LmScratch) + 7) / 8];
  const int prob = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  LmState& st = *reinterpret_cast<LmState*>(sh_state);
  if (threadIdx.x == 0) lm_init(st, opt, poses + 7 * (size_t)prob);
  __syncthreads();
  const double inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
  const long long r0 = prob_row[prob], n_rows = prob_row[prob + 1] - r0;
  const WaveMap wm = make_wave_map<BLOCK>(0, 1, wave);
  const long long rb = r0 + wm.begin(n_rows), re = r0 + wm.end(n_rows);
  auto get_pose = [&](PoseU& P) -> bool {
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = st.x_eval[i];
    load_pose(x, P);
    return true;
  };
  // one evaluation pass + the controller; FIRST: the pass at the start point
  auto pass_first = [&]() {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    stream_rows<WITH_LOSS, NT>(xy, desc, rb, re, lane, get_pose, inv_lf2, acc);
    block_reduce_store<BLOCK / 64>(acc, 0, sh_tot);
    __syncthreads();
    if (wave == 0) lm_advance_wave<true>(st, opt, nullptr, 0, sh_tot, sh_park, lane);  // contains the barrier ...
    else __syncthreads();                                                                // ... the other waves meet here
  };
  auto pass_next = [&]() {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    stream_rows<WITH_LOSS, NT>(xy, desc, rb, re, lane, get_pose, inv_lf2, acc);
    block_reduce_store<BLOCK / 64>(acc, 0, sh_tot);
    __syncthreads();
    if (wave == 0) lm_advance_wave<false>(st, opt, nullptr, 0, sh_tot, sh_park, lane);
    else __syncthreads();
  };
  pass_first();
  const int cap = opt.max_num_iterations + 2;
  for (int pass = 0; pass < cap && st.status == CLC_RUNNING; ++pass) pass_next();  // (status: published before the barrier)
  if (wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      if (st.status == CLC_RUNNING) st.status = CLC_FAILURE;  // unreachable: the controller stops at the iteration cap
      batched_write_outcome(st, prob, poses, summaries, results);
    }
  }
}

// Stragglers only: problems the host loop stopped launching for while they were still running (they are reported as
// failures by clc_solve_batched).  Normally every problem has written its outcome in batched_lm_kernel already.
__global__ void batched_finish_kernel(const LmState* __restrict__ states, int n_problems,
                                      double* __restrict__ poses, clc_summary* __restrict__ summaries,
                                      double* __restrict__ results) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_problems) return;
  if (states[p].status != CLC_RUNNING) return;
  batched_write_outcome(states[p], p, poses, summaries, results);
}

// Send buffer of the all-gather: cap records per rank, the first n_local real (global index = base + local index),
// the rest padding (global_index = -1, everything else 0).
__global__ void pack_results_kernel(const double* __restrict__ results, long long n_local, long long cap,
                                    double base_index, double* __restrict__ send) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cap) return;
  double* o = send + 12 * k;
  if (k < n_local) {
    const double* r = results + 12 * k;
    for (int i = 0; i < 11; ++i) o[i] = r[i];
    o[11] = base_index + (double)k;
  } else {
    for (int i = 0; i < 11; ++i) o[i] = 0.0;
    o[11] = -1.0;
  }
}

// ---------------------------------------------------------------------------------------
// K6 — LineFittingCeres batched over scans (src/LaseCamCalCeres.cpp:385-433).
// A 16-lane row per scan runs the whole <= 10-iteration LM loop in-kernel: the lanes stride over
// the scan's points (residual m0 x + m1 y + 1, Jacobian [x, y], Cauchy loss a, corrector as in K1),
// a DPP all-reduce leaves the identical {H00, H01, H11, g0, g1, cost} in every lane of the row, and every
// lane runs the 2-parameter controller redundantly in registers (SIMT: no broadcast needed).
// A scan is ~10^2 points = a few KiB re-read from L1/L2 per iteration: latency-bound,
// parallel over scans (16 scans per workgroup).
// ---------------------------------------------------------------------------------------
constexpr int LINE_LANES = 16;                    // lanes per scan: one DPP row
constexpr int LINE_SCANS_PER_WAVE = 64 / LINE_LANES;
constexpr int LINE_SCANS_PER_BLOCK = (BLOCK / 64) * LINE_SCANS_PER_WAVE;

// Four scans per wavefront, 16 lanes (one DPP row) each: the in-wave LM controller — ~300 instructions per iteration,
// what a scan of ~10^2 points costs most — then serves four scans per issue slot, and the reduction of
// {H00, H01, H11, g0, g1, cost} is four DPP steps inside the row instead of six cross-lane shuffles
// (one wave per scan: 490 us per 10^5 scans; this form: see DESIGN.md K6).  Rows whose scan has terminated (or does
// not exist) are masked off as a whole, so the row-local DPP reads only ever see active lanes.
template <bool WITH_LOSS>
__global__ __launch_bounds__(BLOCK) void line_fit_kernel(const double* __restrict__ xy,
                                                         const long long* __restrict__ off, const int n_scans,
                                                         const clc_options opt, double* __restrict__ lines,
                                                         clc_summary* __restrict__ summaries) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (LINE_LANES - 1);
  const int scan = (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * LINE_SCANS_PER_WAVE + (lane / LINE_LANES);
  if (scan >= n_scans) return;
  const double2* pts = reinterpret_cast<const double2*>(xy) + off[scan];
  const long long n = off[scan + 1] - off[scan];
  using M = Euclid2Manifold;
  LmStateT<M> st;
  LmScratchT<M> w;
  const double x0[2] = {lines[2 * (size_t)scan], lines[2 * (size_t)scan + 1]};
  lm_init(st, opt, x0);
  const double a = opt.loss_scale_factor;
  const double inv_b = 1.0 / (a * a);
  while (st.status == CLC_RUNNING) {
    const double m0 = st.x_eval[0], m1 = st.x_eval[1];
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // H00 H01 H11 g0 g1 cost
    for (long long k = sub; k < n; k += LINE_LANES) {
      const double2 p = pts[k];
      const double r = fma(m0, p.x, fma(m1, p.y, 1.0));  // :391
      double wt = 1.0;
      if (WITH_LOSS) {
        const double sum = fma(r * r, inv_b, 1.0);
        acc[5] += log_ge1(sum);
        wt = fmax(2.2250738585072014e-308, fast_rcp(sum));
      } else {
        acc[5] = fma(r, r, acc[5]);
      }
      const double wx = wt * p.x, wy = wt * p.y;
      acc[0] = fma(wx, p.x, acc[0]);
      acc[1] = fma(wx, p.y, acc[1]);
      acc[2] = fma(wy, p.y, acc[2]);
      acc[3] = fma(wx, r, acc[3]);
      acc[4] = fma(wy, r, acc[4]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // all-reduce inside the 16-lane row (commutative pairs: all lanes bitwise equal)
      double v = acc[i];
      v += dpp_read<0xB1>(v);   // quad_perm [1,0,3,2]
      v += dpp_read<0x4E>(v);   // quad_perm [2,3,0,1]
      v += dpp_read<0x141>(v);  // row_half_mirror
      v += dpp_read<0x140>(v);  // row_mirror
      acc[i] = v;
    }
    const double cost = WITH_LOSS ? 0.5 * (a * a) * acc[5] : 0.5 * acc[5];
    lm_advance(st, w, opt, nullptr, 0, cost, acc + 3, acc);
  }
  if (sub == 0) {
    lines[2 * (size_t)scan] = st.x_out[0];
    lines[2 * (size_t)scan + 1] = st.x_out[1];
    if (summaries != nullptr) {
      clc_summary sm;
      lm_fill_summary(st, sm);
      sm.solve_ms = 0.0;
      sm.eval_kernel_ms = 0.0;
      sm.eval_kernel_launches = 0;
      summaries[scan] = sm;
    }
  }
}

// ---------------------------------------------------------------------------------------
// TranScanToPoints batched over scans (src/utilities.cpp:181-215): ray i of scan s at
// theta = angle_min[s] + i * angle_increment[s] -> (r cos, r sin, 0), or (1000, 1000, 0) when the
// range is outside [range_min[s], 30).  One thread per ray; streaming, 4 B in / 24 B out.
// ---------------------------------------------------------------------------------------
__global__ void scan_to_points_kernel(const float* __restrict__ ranges, const long long* __restrict__ off,
                                      const int n_scans, const float* __restrict__ angle_min,
                                      const float* __restrict__ angle_inc, const float* __restrict__ range_min,
                                      double* __restrict__ points) {
  const int s = blockIdx.y;
  if (s >= n_scans) return;
  const long long lo = off[s], n = off[s + 1] - lo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float r = ranges[lo + i];
    const double th = angle_min[s] + (double)i * angle_inc[s];
    double x = (double)r * cos(th), y = (double)r * sin(th);
    if (!(r < 30.0 && r >= range_min[s])) { x = 1000.0; y = 1000.0; }
    double* p = points + 3 * (lo + i);
    p[0] = x; p[1] = y; p[2] = 0.0;
  }
}

// The same for device-resident scans of any count: one thread per ray, its scan found by binary search in the offsets.
__global__ void scan_to_points_flat_kernel(const float* __restrict__ ranges, const long long* __restrict__ off,
                                           const long long n_scans, const long long n_rays,
                                           const float* __restrict__ angle_min, const float* __restrict__ angle_inc,
                                           const float* __restrict__ range_min, double* __restrict__ points) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rays) return;
  long long lo = 0, hi = n_scans;  // off[lo] <= k < off[hi]
  while (hi - lo > 1) {
    const long long mid = (lo + hi) >> 1;
    if (off[mid] <= k) lo = mid; else hi = mid;
  }
  const long long s = lo, i = k - off[s];
  const float r = ranges[k];
  const double th = angle_min[s] + (double)i * angle_inc[s];
  double x = (double)r * cos(th), y = (double)r * sin(th);
  if (!(r < 30.0 && r >= range_min[s])) { x = 1000.0; y = 1000.0; }
  double* p = points + 3 * k;
  p[0] = x; p[1] = y; p[2] = 0.0;
}

}  // namespace clc
