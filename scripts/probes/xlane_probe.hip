// Probe: cost of the cross-lane primitives of the 28-accumulator reduction for ONE wave alone on its SIMD (gfx950), cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 64
#define TIC(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+v"(v)); } while (0)
#define TOC(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+v"(v)); } while (0)
template <int CTRL> __device__ __forceinline__ double dpp_read(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void swap_halves(double& x, double& y) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]); y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap_rows(double& x, double& y) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]); y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double seed) {
  __shared__ double lds[64 * 32];
  if (threadIdx.x < 64) {
    double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x, r = 0;
    long long t0, t1;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) { double x = a, y = b; swap_halves(x, y); a = x + y; }
    TOC(a); if (threadIdx.x == 0) cyc[0] = t1 - t0; r += a;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) { double x = a, y = b; swap_rows(x, y); a = x + y; }
    TOC(a); if (threadIdx.x == 0) cyc[1] = t1 - t0; r += a;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a += dpp_read<0xB1>(a);
    TOC(a); if (threadIdx.x == 0) cyc[2] = t1 - t0; r += a;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a += dpp_read<0x140>(a);
    TOC(a); if (threadIdx.x == 0) cyc[3] = t1 - t0; r += a;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a += __shfl_xor(a, 32, 64);
    TOC(a); if (threadIdx.x == 0) cyc[4] = t1 - t0; r += a;
    // 5: independent: 14 swap_halves + add (the first butterfly level on 28 values), per swap
    double v[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) v[i] = seed + i + threadIdx.x;
    TIC(v[0]);
    double w[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) { double x = v[i], y = v[i + 14]; swap_halves(x, y); w[i] = x + y; }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) s += w[i];
    TOC(s); if (threadIdx.x == 0) cyc[5] = (t1 - t0) * N / 14; r += s;
    // 6: LDS transpose alternative: 28 ds_write_b64 + wait
    TIC(a);
#pragma unroll
    for (int i = 0; i < 28; ++i) lds[i * 64 + threadIdx.x] = v[i] + a;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    TOC(a); if (threadIdx.x == 0) cyc[6] = (t1 - t0) * N / 28;
    // 7: 28 ds_read_b64 (conflict-free) + sum
    TIC(a);
    double q = 0;
#pragma unroll
    for (int i = 0; i < 28; ++i) q += lds[i * 64 + (threadIdx.x ^ 1)];
    a += q;
    TOC(a); if (threadIdx.x == 0) cyc[7] = (t1 - t0) * N / 28; r += a;
    // 8: 28 broadcast ds_read_b64 (same address in all lanes) + sum
    TIC(a);
    q = 0;
#pragma unroll
    for (int i = 0; i < 28; ++i) q += lds[i * 64];
    a += q;
    TOC(a); if (threadIdx.x == 0) cyc[8] = (t1 - t0) * N / 28; r += a;
    // 9: readlane_d of 28 values (independent) -> sum
    TIC(a);
    q = 0;
#pragma unroll
    for (int i = 0; i < 28; ++i) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v[i]), 3), hi = __builtin_amdgcn_readlane(__double2hiint(v[i]), 3);
      q += __hiloint2double(hi, lo);
    }
    a += q;
    TOC(a); if (threadIdx.x == 0) cyc[9] = (t1 - t0) * N / 28; r += a;
    out[threadIdx.x] = r;
  }
  __syncthreads();
  // 10: workgroup barrier cost with all four waves arriving together
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[10] = (t1 - t0) * N / 16;
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 512 * 8); (void)hipMalloc(&c, 32 * 8);
  const char* names[11] = {"swap_halves (2 permlane32_swap) + add, dependent", "swap_rows (2 permlane16_swap) + add, dependent", "dpp quad_perm x2 + add, dependent",
                           "dpp row_mirror x2 + add, dependent", "__shfl_xor 32 (double) + add, dependent", "swap_halves + add, 14 independent (per swap)",
                           "ds_write_b64, 28 independent (per write)", "ds_read_b64 conflict-free, 28 + sum (per read)", "ds_read_b64 broadcast, 28 + sum (per read)",
                           "readlane_d, 28 independent + sum (per value)", "__syncthreads, 4 waves in step (per barrier)"};
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, c, 1.25);
  long long h[32]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 11; ++i) printf("  %-52s %7.1f cycles\n", names[i], (double)h[i] / N);
  return 0;
}
