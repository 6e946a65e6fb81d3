// Probe: ds_bpermute_b32 on gfx950 in the controller's setting: 512-thread workgroup, totals summed from LDS by the
// lower 32 lanes of wave 0, gathered by all 64 lanes of wave 0 inside `if (threadIdx.x < 64)`.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double lane_gather(double v, int src) {
  int addr = src << 2;
  asm volatile("" : "+v"(addr));
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(512) void k(double* out, int flag) {
  __shared__ double red[16][32];
  if (threadIdx.x < 512) red[threadIdx.x >> 5][threadIdx.x & 31] = 100.0 * (threadIdx.x >> 5) + (threadIdx.x & 31);
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x < 32) {
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
    if (flag == 7) red[0][threadIdx.x] = t;
  }
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 64 && flag) out[500] = 1.0;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int i6 = lane < 6 ? lane : 5;
    out[lane] = lane_gather(t, 21 + i6);
    out[64 + lane] = lane_gather(t, i6 * 6 - (i6 * (i6 - 1)) / 2);
    const int lo = __builtin_amdgcn_readlane(__double2loint(t), 22), hi = __builtin_amdgcn_readlane(__double2hiint(t), 22);
    out[128 + lane] = __hiloint2double(hi, lo);
  }
}
int main() {
  double* d; (void)hipMalloc(&d, 512 * 8);
  hipLaunchKernelGGL(k, dim3(2), dim3(512), 0, 0, d, 0);
  double h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int v = 0; v < 3; ++v) { printf("variant %d:", v); for (int i = 0; i < 10; ++i) printf(" %g", h[64 * v + i]); printf(" ... %g %g\n", h[64*v+31], h[64*v+63]); }
  return 0;
}
