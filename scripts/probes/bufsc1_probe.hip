// Probe: does a 16-byte agent-scope (sc1) raw-buffer store of one workgroup become visible to sc1 raw-buffer loads of another?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void k(unsigned int* board, int bytes, unsigned int tag, long long* out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(board, 0, bytes, 0x00020000);
  if (blockIdx.x == 0) {
    __builtin_amdgcn_s_sleep(100);
    v4u w; w[0] = tag; w[1] = threadIdx.x; w[2] = tag; w[3] = 7;
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, 16u * threadIdx.x, 0, 16);
  } else {
    long long t0 = wall_clock64(), n = 0;
    for (;;) {
      v4u w = __builtin_amdgcn_raw_buffer_load_b128(rs, 16u * threadIdx.x, 0, 16);
      ++n;
      if (__all(w[0] == tag && w[2] == tag)) { if (threadIdx.x == 0) { out[0] = n; out[1] = wall_clock64() - t0; out[2] = w[1] + w[3]; } break; }
      if (wall_clock64() - t0 > 1000000) { if (threadIdx.x == 0) { out[0] = -n; out[1] = w[0]; out[2] = w[2]; } break; }
    }
  }
}
int main() {
  unsigned int* b; long long* o;
  (void)hipMalloc(&b, 1 << 20); (void)hipMemset(b, 0, 1 << 20); (void)hipMalloc(&o, 64);
  for (unsigned int tag = 1; tag < 4; ++tag) {
    hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, b, 1 << 20, tag, o);
    long long h[3]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("tag %u: polls %lld, ticks(10ns) or word %lld, %lld\n", tag, h[0], h[1], h[2]);
  }
  return 0;
}
