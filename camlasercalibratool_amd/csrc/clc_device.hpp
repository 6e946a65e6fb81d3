// clc_device.hpp — device-side building blocks shared by every kernel of the point-to-plane path (gfx950 / CDNA4, wave64):
// wave-uniform pose, the per-observation rank-1 accumulation (PointInPlaneFactor::Evaluate + Cauchy corrector,
// src/LaseCamCalCeres.cpp:43-66,249), the wavefront / workgroup reductions of the 28 accumulators, the launch flags of
// clc_set_launch, and the static tile -> wave maps.  Included by the kernel headers, never by host-only code.
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

__device__ __forceinline__ long long uniform_ll(long long v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xFFFFFFFFLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}


__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}


// ---------------------------------------------------------------------------------------
// launch flags (clc_set_launch)
constexpr int FLAG_REDUCE_SHUFFLE = 1;  // reference wave reduction instead of the butterfly
constexpr int FLAG_PREFETCH = 2;        // software-pipelined tile loads (next tile in flight while computing)
constexpr int FLAG_NONTEMPORAL = 4;     // nt loads for the streamed tiles
constexpr int FLAG_COMPACT = 16;      // stream the compact layout (clc_stream.hpp) when it is available
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
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// Loads from LDS (or from device memory) that must stay INSIDE a loop, issued where they are written.  Passing the POINTER through an
// empty asm does that, but a generic pointer that went through an asm has lost its address space: the loads become FLAT loads — in
// round 4 the pose reads at the head of every on-chip pass and the 14 reads of the totals in the LM controller were flat loads of LDS
// (vmcnt AND lgkmcnt, a longer round trip than ds_read, and their wait also waits for every global access in flight).  Here the
// OFFSET is what the compiler cannot see through; the pointer keeps its address space: ds_read_b128 / global_load.
typedef __attribute__((address_space(3))) const v2d lds_cv2d;
typedef __attribute__((address_space(1))) const v2d glb_cv2d;
typedef __attribute__((address_space(1))) const double glb_cdouble;
__device__ __forceinline__ const lds_cv2d* lds_opaque(const void* shared_ptr) {
  unsigned int off = 0;
  asm volatile("" : "+v"(off));
  return reinterpret_cast<const lds_cv2d*>(((__attribute__((address_space(3))) const char*)shared_ptr) + off);
}
__device__ __forceinline__ glb_cdouble* global_opaque(const double* device_ptr) {
  unsigned int off = 0;
  asm volatile("" : "+v"(off));
  return reinterpret_cast<glb_cdouble*>(((__attribute__((address_space(1))) const char*)device_ptr) + off);
}

// compact layout (clc_stream.hpp): tiles of 128 points x[128], y[128], z[128] (FP64) + gid[128] (u32); one 48-byte group entry per scan
constexpr int CTILE_DOUBLES = 3 * TILE + TILE / 2;  // 448 doubles = 3 584 B
constexpr int GROUP_DOUBLES = 6;                    // {n.x, n.y, n.z, d, scale, 0}: 48 B, 16-B aligned

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
// (A smaller share for wave 0 — which in the step kernel reaches its rows 0.5-0.8 us late, behind the controller's write-back —
// was measured and rejected: with scans of a whole number of rows the boundaries snap to scan starts, so unequal nominal
// shares turn into one wave with two scans next to waves with one: 8.7 -> 9.7-10.0 us per step launch at C2 for any share
// between 5:6 and 0:6, no change at 2.5e5 observations or with 32-row scans; scripts/r03_wave0_share.py, profiles/r03_wave0_share.md.)
static __global__ void wave_split_kernel(const RowDesc* __restrict__ desc, const int n_rows, const int n_blocks, int* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, total = n_blocks * 8 + 1;
  if (t >= total) return;
  if (t == total - 1) { out[t] = n_rows; return; }
  const WaveMap m = make_wave_map<512, false>(t >> 3, n_blocks, t & 7);
  const int nominal = (int)m.begin(n_rows);
  // Boundaries move to the nearest scan start within half a share.  "Nearest start within a fixed window, else stay" is a
  // monotone map of the ordered nominal boundaries, so the table stays a partition (a share may come out empty).
  const int window = ((n_rows / n_blocks) / 8 + 1) / 2;
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

}  // namespace clc
