// clc_frontend.hpp — kernels beside the LM solve: plug-in level parity with the Ceres callbacks (factor_kernel,
// plus_kernel), the closed-form initialiser's 9x9 normal equation (K5, src/LaseCamCalCeres.cpp:144-161), the batched scan
// line fit (K6, :385-433) and TranScanToPoints (src/utilities.cpp:181-215).
#pragma once
#include "clc_controller.hpp"
#include "clc_stream.hpp"

namespace clc {

// ---------------------------------------------------------------------------------------
// plug-in level kernels (element-wise parity with the reference's Ceres callbacks)
// ---------------------------------------------------------------------------------------
// PointInPlaneFactor::Evaluate per record, literal operation order of
// src/LaseCamCalCeres.cpp:43-66 (pt_c = R p + t; r = s (n.pt_c + d); J = s [n, n^T(-R [p]x), 0]).
static __global__ void factor_kernel(const double* __restrict__ tiles, long long n,
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
static __global__ void plus_kernel(const double* __restrict__ x, const double* __restrict__ delta,
                            double* __restrict__ out, long long n) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double a[7], d[6], o[7];
  for (int i = 0; i < 7; ++i) a[i] = x[7 * k + i];
  for (int i = 0; i < 6; ++i) d[i] = delta[6 * k + i];
  pose_plus(a, d, o);
  for (int i = 0; i < 7; ++i) out[7 * k + i] = o[i];
}

// Debug/test entry: the DEVICE branches of the scalar helpers (clc_math.hpp, clc_rows.hpp: v_rcp / v_rsq seeds + Newton steps,
// frexp + the polynomial logarithm), element-wise on caller-provided values — the host unit shims compile their plain-expression
// branches, so only a kernel can pin these (tests/test_gpu_device_math.py).
static __global__ void math_probe_kernel(const int op, const double* __restrict__ in, double* __restrict__ out, const long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  double y = 0.0;
  if (op == 0) y = rsqrt_pos(x);
  else if (op == 1) y = rcp_pos(x);
  else if (op == 2) y = rcp_pos_safe(x);
  else if (op == 3) y = sqrt_pos(x);
  else if (op == 4) y = rcp_ge1(x);
  else if (op == 5) y = rcp_ge1_weight(x);
  else if (op == 6) {
    int e;
    const double m = frexp_pos(x, e);
    y = log_mant_exp(m, e);
  }
  out[i] = y;
}

// Debug/test entry: run only the wave reduction on caller-provided lane values
// in[64][28] -> out[28].
static __global__ void wave_reduce_test_kernel(const double* __restrict__ in, double* __restrict__ out,
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

static __global__ __launch_bounds__(BLOCK) void normal9_kernel(const double* __restrict__ tiles,
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
static __global__ __launch_bounds__(BLOCK) void reduce9_kernel(const double* __restrict__ partials,
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
template <bool NT, int STRIDE = ROW_DOUBLES>
__global__ __launch_bounds__(BLOCK) void normal9_rows_kernel(const double* __restrict__ xy, const RowDesc* __restrict__ desc,
                                                             const long long n_rows, double* __restrict__ partials) {
  double acc[NACC9];
#pragma unroll
  for (int i = 0; i < NACC9; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<BLOCK>(blockIdx.x, gridDim.x, threadIdx.x >> 6);
  Normal9Rows pol;
  stream_rows_policy<Normal9Rows, NT, ROWS_DEPTH, STRIDE>(pol, xy, desc, wm.begin(n_rows), wm.end(n_rows), lane,
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
static __global__ void scan_to_points_kernel(const float* __restrict__ ranges, const long long* __restrict__ off,
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
static __global__ void scan_to_points_flat_kernel(const float* __restrict__ ranges, const long long* __restrict__ off,
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
