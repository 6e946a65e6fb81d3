// Probe: does a hipGraph shorten the boundary between two dependent kernels on gfx950?  Kernel A spins ~20 us and stamps
// its end (wall_clock64, 100 MHz), kernel B stamps its start; chains of 8 (A,B alternating) as stream launches vs one graph launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin_kernel(unsigned long long* stamps, int slot, int ticks) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = t0;
  while (wall_clock64() - t0 < (unsigned long long)ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}
int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 1024);
  hipStream_t s;
  (void)hipStreamCreate(&s);
  const int N = 8;
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {1, 256}) {
      hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
      if (mode == 1) {
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(512), 0, s, d, k, 500);
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      }
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemsetAsync(d, 0, 1024, s);
        if (mode == 0) for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(512), 0, s, d, k, 500);
        else (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        unsigned long long h[2 * N];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double gap = 0;
        for (int k = 1; k < N; ++k) gap += (double)(long long)(h[2 * k] - h[2 * k - 1]) / 100.0;
        printf("%s grid=%3d: mean gap end(k-1) -> start(k) = %.2f us (kernel %.2f us)\n", mode ? "graph " : "stream", grid, gap / (N - 1), (h[1] - h[0]) / 100.0);
      }
    }
  return 0;
}
