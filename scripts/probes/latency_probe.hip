// Probe: dependent-issue latency of the instructions the LM controller is made of, for ONE wave of a 512-thread
// workgroup while the other seven wait at a barrier (the controller's situation), gfx950.  cycles = s_memtime ticks.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
#define TIC(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+v"(v)); } while (0)
#define TOC(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+v"(v)); } while (0)
__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, double seed, int only_lane0) {
  __shared__ double lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = seed + threadIdx.x;
  __syncthreads();
  if (threadIdx.x < 64 && (!only_lane0 || threadIdx.x == 0)) {
    double a = seed, b = seed * 0.5, r = 0;
    long long t0, t1;
    // 1: dependent fma chain
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma(a, 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[0] = t1 - t0; r += a;
    // 2: 4 independent fma chains
    double c0 = seed, c1 = seed + 1, c2 = seed + 2, c3 = seed + 3;
    TIC(c0);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { c0 = fma(c0, 1.0000001, b); c1 = fma(c1, 1.0000001, b); c2 = fma(c2, 1.0000001, b); c3 = fma(c3, 1.0000001, b); }
    TOC(c0); if (threadIdx.x == 0) cyc[1] = t1 - t0; r += c0 + c1 + c2 + c3;
    // 3: dependent mul chain
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = a * 1.0000001;
    TOC(a); if (threadIdx.x == 0) cyc[2] = t1 - t0; r += a;
    // 4: readlane -> fma chain (value through SGPRs each step)
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) a = fma(readlane_d(a, 3), 1.0000001, b);
    TOC(a); if (threadIdx.x == 0) cyc[3] = (t1 - t0) * 4; r += a;
    // 5: rsq chain
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) a = __builtin_amdgcn_rsq(a) + 2.0;
    TOC(a); if (threadIdx.x == 0) cyc[4] = (t1 - t0) * 2; r += a;   // two instructions per step
    // 6: IEEE division chain
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) a = 3.0 / a + 1.0;
    TOC(a); if (threadIdx.x == 0) cyc[5] = (t1 - t0) * 8; r += a;   // per division+add, scaled to N steps
    // 7: sqrt chain
    a = seed + 2.0;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) a = sqrt(a) + 2.0;
    TOC(a); if (threadIdx.x == 0) cyc[6] = (t1 - t0) * 8; r += a;
    // 8: dependent LDS read chain (address from the previous value)
    int idx = threadIdx.x & 7;
    TIC(idx);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) idx = ((int)lds[idx & 63]) & 7;
    TOC(idx); if (threadIdx.x == 0) cyc[7] = (t1 - t0) * 8; r += idx;
    // 9: v_cndmask chain on doubles
    a = seed;
    TIC(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = (a > 0.5 + i) ? a : a + 1.0;
    TOC(a); if (threadIdx.x == 0) cyc[8] = t1 - t0; r += a;
    out[threadIdx.x] = r;
  }
  __syncthreads();
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 512 * 8); (void)hipMalloc(&c, 16 * 8);
  const char* names[9] = {"dependent fma", "4 independent fma chains", "dependent mul", "readlane_d + fma", "rsq + add", "IEEE div + add", "sqrt + add", "dependent LDS read+cvt", "cmp + select + add"};
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, c, 1.25, mode);
    long long h[16]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s:\n", mode ? "lane 0 only" : "all 64 lanes");
    for (int i = 0; i < 9; ++i) printf("  %-28s %6.1f cycles per step (%d steps)\n", names[i], (double)h[i] / N, N);
  }
  return 0;
}
