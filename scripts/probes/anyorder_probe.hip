// Probe: does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950, i.e. does launch k+1 start while launch k runs?
// Kernel A spins ~30 us and stamps start/end (wall_clock64, 100 MHz); kernel B stamps its start.  Ordinary launch vs
// any-order launch of B behind A on the same stream.  build: hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin_kernel(unsigned long long* stamps, int ticks) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
  while (wall_clock64() - t0 < (unsigned long long)ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void stamp_kernel(unsigned long long* stamps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = wall_clock64();
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {1, 64, 256}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(d, 0, 64, s);
        hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, s, d, 3000);
        if (mode == 0)
          hipLaunchKernelGGL(stamp_kernel, dim3(grid), dim3(256), 0, s, d);
        else
          hipExtLaunchKernelGGL(stamp_kernel, dim3(grid), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
        hipError_t e = hipStreamSynchronize(s);
        unsigned long long h[3];
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("mode=%s grid=%3d rc=%d  A: %.2f us long;  B starts %.2f us after A starts (%+.2f us vs A's end)\n",
               mode ? "any-order" : "ordered  ", grid, (int)e, (h[1] - h[0]) / 100.0, (double)(long long)(h[2] - h[0]) / 100.0,
               (double)(long long)(h[2] - h[1]) / 100.0);
      }
    }
  return 0;
}
