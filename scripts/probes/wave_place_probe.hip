// Where the hardware puts the waves of two co-resident 256-thread workgroups of a CU (the shape of resident_solve_kernel<4 waves>:
// 77.8 KB of LDS each): HW_ID (wave slot, SIMD, CU, SE), XCC_ID and the LDS allocation base per wave.
// build: hipcc --offload-arch=gfx950 -O2 wave_place_probe.hip -o wave_place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned int* out, int spin) {
  __shared__ double big[77 * 1024 / 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  big[threadIdx.x] = 1.0;
  __syncthreads();
  const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  const unsigned int lds = __builtin_amdgcn_s_getreg((31 << 11) | 6);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);  // keep the slot so that pairs of workgroups are co-resident
  if (lane == 0) {
    unsigned int* o = out + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = hw; o[1] = xcc; o[2] = lds; o[3] = (unsigned int)big[0];
  }
}
int main() {
  const int wgs = 512;
  unsigned int* d;
  hipMalloc(&d, wgs * 16 * sizeof(unsigned int));
  hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, d, 20000);  // 200 us
  std::vector<unsigned int> h(wgs * 16);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned int, std::vector<int>> by_cu;
  for (int w = 0; w < wgs; ++w) {
    const unsigned int hw = h[w * 16], xcc = h[w * 16 + 1] & 15;
    const unsigned int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(w);
  }
  std::printf("workgroups %d, distinct (xcc, se, sh, cu) %zu\n", wgs, by_cu.size());
  int shown = 0, same = 0, pairs = 0;
  for (auto& kv : by_cu) {
    if (kv.second.size() == 2) {
      ++pairs;
      const int a = kv.second[0], b = kv.second[1];
      const unsigned int sa = (h[a * 16] >> 4) & 3, sb = (h[b * 16] >> 4) & 3;
      if (sa == sb) ++same;
    }
    if (shown < 6) {
      ++shown;
      std::printf("cu key %06x:", kv.first);
      for (int w : kv.second) {
        std::printf("  wg %d [lds base %u] simd of waves 0-3:", w, h[w * 16 + 2] & 0xff);
        for (int k = 0; k < 4; ++k) std::printf(" %u", (h[(w * 4 + k) * 4] >> 4) & 3);
      }
      std::printf("\n");
    }
  }
  std::printf("CUs with two workgroups: %d; wave 0 of both on the SAME simd: %d\n", pairs, same);
  // does wg id % 8 predict the xcc?
  int match = 0;
  for (int w = 0; w < wgs; ++w) match += (int)((h[w * 16 + 1] & 15) == (unsigned)(w % 8));
  std::printf("xcc == wg %% 8 for %d of %d workgroups\n", match, wgs);
  return 0;
}
