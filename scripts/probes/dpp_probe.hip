// Probe: what the building blocks of the register-resident LM controller (csrc/clc_lmregs.hpp) cost ONE wave that has its SIMD to
// itself (the controller's situation), gfx950: dependent chains, cycles per step (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 128
#define TIC(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+v"(v)); } while (0)
#define TOC(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+v"(v)); } while (0)
template <int K> __device__ __forceinline__ double row_bc(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
template <int K> __device__ __forceinline__ double fma_bc(double a, double b, double c) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b), "n"(K));
  return c;
}
template <int K> __device__ __forceinline__ double bc32(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double seed) {
  __shared__ double lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = seed + threadIdx.x;
  __syncthreads();
  if (threadIdx.x < 64) {
    double a = seed, b = seed * 0.5, r = 0;
    long long t0, t1;
    // 0: dependent fma
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma(a, 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[0] = t1 - t0; r += a;
    // 1: dependent v_fmac_f64_dpp (bcast of the previous result)
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma_bc<3>(a, 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[1] = t1 - t0; r += a;
    // 2: v_mov_b64_dpp + fma
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma(row_bc<3>(a), 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[2] = t1 - t0; r += a;
    // 3: 2 x v_mov_b32_dpp (builtin) + fma
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma(bc32<3>(a), 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[3] = t1 - t0; r += a;
    // 4: readlane_d + fma
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma(readlane_d(a, 3), 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[4] = t1 - t0; r += a;
    // 5: v_rsq_f64 + add
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_amdgcn_rsq(a) + 2.0;
    TOC(a); if (threadIdx.x == 0) cyc[5] = t1 - t0; r += a;
    // 6: v_rcp_f64 + add
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_amdgcn_rcp(a) + 2.0;
    TOC(a); if (threadIdx.x == 0) cyc[6] = t1 - t0; r += a;
    // 7: independent fmac_dpp x 4 chains
    double c0 = seed, c1 = seed + 1, c2 = seed + 2, c3 = seed + 3;
    TIC(c0);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { c0 = fma_bc<1>(b, 1.0000001, c0); c1 = fma_bc<2>(b, 1.0000001, c1); c2 = fma_bc<3>(b, 1.0000001, c2); c3 = fma_bc<4>(b, 1.0000001, c3); }
    TOC(c0); if (threadIdx.x == 0) cyc[7] = t1 - t0; r += c0 + c1 + c2 + c3;
    // 8: select chain on doubles (compare with constant + 2 cndmask)
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = (a > 0.5 + i) ? a + 1.0 : a * 1.0000001;
    TOC(a); if (threadIdx.x == 0) cyc[8] = t1 - t0; r += a;
    // 9: LDS broadcast read, dependent address
    int idx = threadIdx.x & 7;
    TIC(idx);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) idx = ((int)lds[idx & 63]) & 7;
    TOC(idx); if (threadIdx.x == 0) cyc[9] = (t1 - t0) * 4; r += idx;
    // 10: IEEE division
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) a = 3.0 / a + 1.0;
    TOC(a); if (threadIdx.x == 0) cyc[10] = (t1 - t0) * 4; r += a;
    // 11: IEEE sqrt
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) a = sqrt(a) + 2.0;
    TOC(a); if (threadIdx.x == 0) cyc[11] = (t1 - t0) * 4; r += a;
    // 12: clock64 back to back (stamp overhead)
    TIC(a);
    long long q = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_barrier(0); q += clock64(); }
    TOC(a); if (threadIdx.x == 0) cyc[12] = (t1 - t0) * 8; r += (double)(q & 1);
    // 13: dependent mul
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = a * 1.0000001;
    TOC(a); if (threadIdx.x == 0) cyc[13] = t1 - t0; r += a;
    // 14: 8 independent fma chains
    double e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = seed + j;
    TIC(e[0]);
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = fma(e[j], 1.0000001, b);
    TOC(e[0]); if (threadIdx.x == 0) cyc[14] = t1 - t0;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += e[j];
    out[threadIdx.x] = r;
  }
  __syncthreads();
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 512 * 8); (void)hipMalloc(&c, 32 * 8);
  const char* names[15] = {"dependent fma", "dependent v_fmac_f64_dpp (+s_nop 1)", "v_mov_b64_dpp + fma", "2 x v_mov_b32_dpp + fma", "readlane_d + fma", "v_rsq_f64 + add",
                           "v_rcp_f64 + add", "4 independent v_fmac_f64_dpp chains", "cmp + select(2 cndmask) + add/mul", "dependent LDS read + cvt", "IEEE div + add", "IEEE sqrt + add",
                           "clock64 (x8: per 128)", "dependent mul", "8 independent fma chains"};
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, c, 1.25);
  long long h[32]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 15; ++i) printf("  %-40s %7.1f cycles per step\n", names[i], (double)h[i] / N);
  return 0;
}
