// clc_legacy.hpp — paths the default build no longer contains (-DCLC_LEGACY_PATHS brings them back for the bit-identity
// tests and for A/B runs): the ticket-fused evaluation + controller launch of round 1 (flag 8: superseded by the step
// kernel, which needs no inter-workgroup hand-off) and the per-wave timeline twin of the compact evaluation kernel.
#pragma once
#ifdef CLC_LEGACY_PATHS
#include "clc_controller.hpp"
#include "clc_stream.hpp"

namespace clc {

// Profiling twin of the default evaluation kernel (loss, Jacobian, compact layout): identical work,
// plus per-workgroup stamps {wall start, wall end (100 MHz s_memrealtime, chip-global),
// shader cycles: prologue, streaming loop, reduction epilogue}.  Debug/analysis only.
template <int BT>
__global__ __launch_bounds__(BT) void eval_timeline_kernel(const double* __restrict__ ctiles,
                                                           const double* __restrict__ groups, const long long n,
                                                           const double* __restrict__ pose, const double lf,
                                                           double* __restrict__ partials,
                                                           long long* __restrict__ stamps) {
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  PoseU P;
  load_pose(pose, P);
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long wave_global = (long long)blockIdx.x * (BT / 64) + wave;
  const WaveMap wm = make_wave_map<BT>(blockIdx.x, gridDim.x, wave);
  const long long c1 = clock64();
  stream_ctiles<true, true, false>(ctiles, groups, n, wm, lane, [&](PoseU& Q) { Q = P; return true; }, inv_lf2, acc);
  const long long c2 = clock64();
  const long long w2 = wall_clock64();
  block_reduce_store<BT / 64>(acc, 0, partials + (size_t)blockIdx.x * NACC);
  const long long c3 = clock64();
  if (lane == 0) {  // one record per WAVE: {wall start, wall end of loop, cycles prologue, loop, epilogue}
    long long* s = stamps + 8 * (size_t)wave_global;
    s[0] = w0; s[1] = w2; s[2] = c1 - c0; s[3] = c2 - c1; s[4] = c3 - c2; s[5] = wall_clock64();
  }
}

// ---------------------------------------------------------------------------------------
// K1+K2 fused — evaluation launch whose LAST-ARRIVING workgroup runs the reduction + LM
// controller, so one LM iteration is ONE launch.  Inter-workgroup hand-off (placement
// independent, MI355X per-XCD L2s are not coherent):
//   producer: partial row stored write-through (agent-scope relaxed atomic stores = sc1),
//             every storing wave drains vmcnt(0), workgroup barrier, ONE lane takes a ticket
//             with a relaxed agent-scope fetch_add;
//   consumer: the workgroup that draws ticket == gridDim-1 reads all rows with agent-scope
//             (sc1) loads, which bypass its CU's L1 — no stale lines possible.
// The ticket counter is reset by the last workgroup (all others have already arrived) and is
// zeroed by lm_init_kernel before the first launch of a solve.
// ---------------------------------------------------------------------------------------
template <bool WITH_LOSS, bool NT, bool COMPACT, bool DEEP, int BT>
__global__ __launch_bounds__(BT) void eval_lm_kernel(const double* __restrict__ tiles,
                                                     const double* __restrict__ groups, const long long n,
                                                     LmState* __restrict__ state, const clc_options opt,
                                                     double* __restrict__ partials,
                                                     unsigned int* __restrict__ ticket_counter,
                                                     clc_iteration* __restrict__ trace, int trace_cap,
                                                     HostMailbox* mailbox) {
  __shared__ double red[LM_GROUPS][32];
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ double wsum[BT / 64][NACC];
  __shared__ int sh_last;
  auto get_pose = [&](PoseU& P) -> bool {
    const int32_t st = state->status;  // issued together with the pose loads: one wait
    load_pose(state->x_eval, P);
    return st == CLC_RUNNING;
  };
  const double lf = opt.loss_scale_factor;
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const WaveMap wm = make_wave_map<BT>(blockIdx.x, gridDim.x, wave);
  bool active;
  if (COMPACT && DEEP)
    active = stream_ctiles_deep<WITH_LOSS, true, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else if (COMPACT)
    active = stream_ctiles<WITH_LOSS, true, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else
    active = stream_tiles<WITH_LOSS, true, true, NT>(tiles, n, wm, lane, get_pose, inv_lf2, acc);
  if (!active) return;  // uniform over the launch
  wave_reduce_butterfly(acc, wsum[wave], lane);
  __syncthreads();
  // ---- publish this workgroup's partial row (write-through) and take a ticket ----
  if (threadIdx.x < NACC) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) s += wsum[w][threadIdx.x];
    __hip_atomic_store(partials + (size_t)blockIdx.x * NACC + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every (storing) wave: stores acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!sh_last) return;
  // ---- last-arriving workgroup: every other row is complete and visible at agent scope ----
  const long long c0 = clock64();
  if (threadIdx.x == 0) __hip_atomic_store(ticket_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  LmLoads L;
  lm_issue_loads<true, false, BT>(partials, state, L);
  lm_tail<true, false, BT>(partials, (int)gridDim.x, state, state, opt, trace, trace_cap, mailbox, red, sh_state, c0, nullptr, L);
}

}  // namespace clc
#endif  // CLC_LEGACY_PATHS
