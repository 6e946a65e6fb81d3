// clc_layouts.hpp — upload-time kernels: every O(N) step of turning the 64-byte residual-block records of the C-ABI
// (clc_observation: what a PointInPlaneFactor + its CauchyLoss hold, src/LaseCamCalCeres.cpp:19-21,249) into the layouts
// the evaluation kernels stream — 64-byte tiles, the compact 28-byte layout, the 17-byte row layout — and the
// residual-block construction itself from resident pose-major scans (:222-295).  One-off per upload.
#pragma once
#include "clc_device.hpp"

namespace clc {

// ---------------------------------------------------------------------------------------
// retile: AoS records -> tiles.  One thread per record (one-time cost per upload).
// ---------------------------------------------------------------------------------------
static __global__ void retile_kernel(const double* __restrict__ aos, double* __restrict__ tiles,
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
static __global__ void retile_batched_kernel(const double* __restrict__ aos,
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

// upload-time helpers of the compact layout -------------------------------------------------
// flag[k] = 1 when record k starts a new group: (n, d, scale) differ bitwise from record k-1.
static __global__ void group_flag_kernel(const double* __restrict__ aos, long long n, unsigned char* __restrict__ flag) {
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

static __global__ void build_groups_kernel(const double* __restrict__ aos, const long long* __restrict__ starts,
                                    long long n_groups, double* __restrict__ groups) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const double* r = aos + 8 * starts[g];
  double* o = groups + g * GROUP_DOUBLES;
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[7]; o[5] = 0.0;
}

// Records [rec_off[b], rec_off[b+1]) of "problem" b -> compact tiles starting at tile_off[b]
// (single problem: one entry).  One workgroup per problem, grid-stride over y for long ones.
static __global__ void build_ctiles_kernel(const double* __restrict__ aos, const unsigned int* __restrict__ gid,
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
static __global__ __launch_bounds__(SCAN_THREADS) void scan_totals_kernel(unsigned long long* __restrict__ totals, long long n_blocks) {
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
// record has p.z != 0 (the rows then carry z: ROW_DOUBLES_Z doubles per row; the on-chip resident layout does not apply).
static __global__ void scan_flag_kernel(const double* __restrict__ aos, long long n, unsigned char* __restrict__ flag,
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
static __global__ void mark_problem_starts_kernel(const long long* __restrict__ rec_off, long long n_problems, long long n,
                                           unsigned char* __restrict__ flag) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_problems && rec_off[p] < n) flag[rec_off[p]] = 1;
}

// starts[g] = first record of scan g; starts[G] = n; rows[g] = rows the scan occupies (filled by scan_rows_kernel)
static __global__ void scan_starts_kernel(const unsigned char* __restrict__ flag, const unsigned int* __restrict__ gid, long long n,
                                   long long n_groups, long long* __restrict__ starts) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n && flag[k]) starts[gid[k]] = k;
  if (k == 0) starts[n_groups] = n;
}

static __global__ void scan_rows_kernel(const long long* __restrict__ starts, long long n_groups, unsigned int* __restrict__ rows) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_groups) rows[g] = (unsigned int)((starts[g + 1] - starts[g] + ROW - 1) / ROW);
}

// row_begin[G+1]: exclusive prefix of rows[] (row_begin[0] = 0 written here).  One thread per row SLOT.
// stride = ROW_DOUBLES, or ROW_DOUBLES_Z: the 64 z of the row are stored after its (x, y) pairs.
static __global__ void build_rows_kernel(const double* __restrict__ aos, const long long* __restrict__ starts,
                                  const unsigned int* __restrict__ row_begin, long long n_groups, long long n_rows,
                                  const int stride, double* __restrict__ xy, RowDesc* __restrict__ desc) {
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
  double x = 0.0, y = 0.0, z = 0.0;
  if (k < end) { x = aos[8 * k + 4]; y = aos[8 * k + 5]; z = aos[8 * k + 6]; }
  v2d v; v[0] = x; v[1] = y;
  reinterpret_cast<v2d*>(xy + r * stride)[lane] = v;
  if (stride == ROW_DOUBLES_Z) xy[r * stride + ROW_DOUBLES + lane] = z;
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
static __global__ void problem_rows_kernel(const long long* __restrict__ rec_off, const unsigned int* __restrict__ gid,
                                    const unsigned int* __restrict__ row_begin, long long n_problems, long long n,
                                    long long n_rows, long long* __restrict__ prob_row) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_problems) return;
  prob_row[p] = (p < n_problems && rec_off[p] < n) ? (long long)row_begin[gid[rec_off[p]]] : n_rows;
}

// groups[g] of the compact layout, from the device-resident starts
static __global__ void build_groups_dev_kernel(const double* __restrict__ aos, const long long* __restrict__ starts,
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

static __global__ __launch_bounds__(BLOCK) void flatten_kernel(const int n_poses, const double* __restrict__ tag_q_wxyz,
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

}  // namespace clc
