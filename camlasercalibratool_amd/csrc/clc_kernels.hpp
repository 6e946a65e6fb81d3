// clc_kernels.hpp — the evaluation and solve kernels of the point-to-plane extrinsic path (gfx950 / CDNA4, wave64).
// Included by the abi_*.hip translation units (through clc_abi_internal.hpp).  See DESIGN.md §3 for the measurements.
//
//   eval_rows_kernel / eval_kernel    K1: fused residual + analytic 6-DoF Jacobian (a3) + Cauchy corrector (a4) + reduction of
//                                     {H(21), g(6), cost} into one 28-double partial row per workgroup (clc_eval, the
//                                     [evaluation, controller] launch pair)
//   reduce_kernel, lm_kernel          K2: fixed-order reduction of the rows (+ the LM controller, clc_controller.hpp)
//   step_kernel                       K3: ONE launch per LM iteration — every workgroup runs the controller on the previous
//                                     launch's rows, then streams (the default clc_solve)
//   batched_*_kernel                  K4: independent problems — lockstep [evaluation, controller] launches, or the whole
//                                     solve of a problem in one launch (batched_solve_kernel; on chip: clc_resident.hpp)
// Building blocks: clc_device.hpp (accumulation, reductions, flags, wave maps), clc_stream.hpp (streaming loops per layout),
// clc_controller.hpp (row reduction + LM controller), clc_layouts.hpp (upload), clc_frontend.hpp (K5, K6, plug-in level).
#pragma once
#include "clc_controller.hpp"
#include "clc_device.hpp"
#include "clc_layouts.hpp"
#include "clc_stream.hpp"

namespace clc {

// K1 on the row layout: same contract as eval_kernel (one 28-double partial per workgroup).
// Z: the rows carry z (ROW_DOUBLES_Z stride, 14 moments per scan).
template <bool WITH_LOSS, bool NT, int BT, bool WEIGHTED, int DEPTH = ROWS_DEPTH, bool Z = false>
__global__ __launch_bounds__(BT) void eval_rows_kernel(const double* __restrict__ xy, const RowDesc* __restrict__ desc,
                                                       const long long n_rows, const double* __restrict__ pose,
                                                       const int32_t* __restrict__ status, const double lf,
                                                       const int reduce_mode, double* __restrict__ partials,
                                                       const Pose7 pose_arg, const int use_pose_arg) {
  auto get_pose = [&](PoseU& P) -> bool {
    if (use_pose_arg) {
      load_pose(pose_arg.v, P);
      return true;
    }
    const int32_t st = status != nullptr ? *status : (int32_t)CLC_RUNNING;
    load_pose(pose, P);
    return st == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  long long r0, r1;
  if (BT == 512 && !WEIGHTED) {  // equal shares, boundaries at scan starts (wave_split_kernel)
    const WaveRun run = wave_run(desc, n_rows, blockIdx.x, threadIdx.x >> 6);
    r0 = run.begin;
    r1 = run.end;
  } else {
    const WaveMap wm = make_wave_map<BT, WEIGHTED>(blockIdx.x, gridDim.x, threadIdx.x >> 6);
    r0 = wm.begin(n_rows);
    r1 = wm.end(n_rows);
  }
  if (!stream_rows<WITH_LOSS, NT, DEPTH, Z>(xy, desc, r0, r1, lane, get_pose, inv_lf2, acc)) return;
  block_reduce_store<BT / 64>(acc, reduce_mode, partials + (size_t)blockIdx.x * NACC);
}

template <bool WITH_LOSS, bool WITH_JAC, bool PREFETCH, bool NT, bool COMPACT, int BT>
__global__ __launch_bounds__(BT) void eval_kernel(const double* __restrict__ tiles,
                                                  const double* __restrict__ groups,
                                                  const long long n,
                                                  const double* __restrict__ pose,
                                                  const int32_t* __restrict__ status,
                                                  const double lf, const int reduce_mode,
                                                  double* __restrict__ partials, const Pose7 pose_arg,
                                                  const int use_pose_arg) {
  // first launch of a solve: the LM state is not initialised yet (the first lm_kernel does that),
  // so the pose comes by value and the (stale) termination flag is ignored
  auto get_pose = [&](PoseU& P) -> bool {
    if (use_pose_arg) {
      load_pose(pose_arg.v, P);
      return true;
    }
    const int32_t st = status != nullptr ? *status : (int32_t)CLC_RUNNING;  // both scalar loads issued before the
    load_pose(pose, P);                                                     // first use: one wait, not two
    return st == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<BT>(blockIdx.x, gridDim.x, threadIdx.x >> 6);
  bool active;
  if (COMPACT && PREFETCH)
    active = stream_ctiles_deep<WITH_LOSS, WITH_JAC, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else if (COMPACT)
    active = stream_ctiles<WITH_LOSS, WITH_JAC, NT>(tiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else
    active = stream_tiles<WITH_LOSS, WITH_JAC, PREFETCH, NT>(tiles, n, wm, lane, get_pose, inv_lf2, acc);
  if (!active) return;  // uniform over the launch: the solve had already terminated
  block_reduce_store<BT / 64>(acc, reduce_mode, partials + (size_t)blockIdx.x * NACC);
}

// Fixed-order sum of the block partials: thread (c, rg) sums rows rg, rg+8, ... of column c
// (16 independent loads in flight per round — a dependent load chain here costs more than
// the whole evaluation kernel), then the 8 row groups are combined in order.
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int n_blocks,
                                                double (*red)[32]) {
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  constexpr int RG = BLOCK / 32, UNROLL = 16;
  double s = 0.0;
  if (c < NACC) {
    for (int b0 = rg; b0 < n_blocks; b0 += RG * UNROLL) {
      double v[UNROLL];
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) {
        const int b = b0 + RG * j;
        v[j] = (b < n_blocks) ? partials[(size_t)b * NACC + c] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < UNROLL; ++j) s += v[j];
    }
  }
  red[rg][c] = s;
  __syncthreads();
}

static __global__ __launch_bounds__(BLOCK) void reduce_kernel(const double* __restrict__ partials,
                                                       int n_blocks, int with_loss, double lf,
                                                       double* __restrict__ out28) {
  __shared__ double red[BLOCK / 32][32];
  reduce_partials(partials, n_blocks, red);
  if (threadIdx.x < NACC) {
    double s = 0.0;
#pragma unroll
    for (int rg = 0; rg < BLOCK / 32; ++rg) s += red[rg][threadIdx.x];
    out28[threadIdx.x] = (threadIdx.x == 27) ? finalize_cost(s, with_loss != 0, lf) : s;
  }
}

template <bool FIRST>
__global__ __launch_bounds__(BLOCK) void lm_kernel(const double* __restrict__ partials,
                                                   int n_blocks, LmState* __restrict__ state,
                                                   const clc_options opt,
                                                   clc_iteration* __restrict__ trace,
                                                   int trace_cap, HostMailbox* mailbox, const Pose7 pose0) {
  __shared__ double red[LM_GROUPS][32];
  __shared__ double sh_state[LM_STATE_WORDS];
  const long long c0 = clock64();
  if (!FIRST && state->status != CLC_RUNNING) return;
  LmLoads L;
  lm_issue_loads<false, FIRST, BLOCK>(partials, state, L);
  lm_tail<false, FIRST, BLOCK>(partials, n_blocks, state, state, opt, trace, trace_cap, mailbox, red, sh_state, c0, &pose0, L);
}

// ---------------------------------------------------------------------------------------
// Step kernel — ONE launch per LM iteration, no second kernel, no inter-workgroup hand-off.
// Launch k evaluates at the point x_k and leaves one partial row per workgroup; launch k+1 starts
// by letting EVERY workgroup read those rows and run the controller redundantly (same inputs, same
// instruction stream: bit-identical x_{k+1} everywhere), then streams its tiles at x_{k+1}.  That
// removes the controller's own launch (launch boundary + kernel-argument fetch + its barrier
// skew, ~4 us of the ~16 us an iteration took at 10^6 observations) and overlaps the controller
// with the first tile loads, which are issued before it (the streaming loops call `get_pose` after
// their prologue loads).  Workgroup 0 alone writes the advanced state, the trace and the host
// mailbox.  LM state and partial rows are double-buffered by launch parity, because a workgroup of
// launch k+1 may still be reading what another one is already overwriting.
//   MODE 0: first launch of a solve (pose by value, nothing to consume)
//   MODE 1: second launch (controller initialises the LM state: lm_init + first rows)
//   MODE 2: steady state
// A launch that finds the solve already terminated (queued ahead by the host) copies the state forward, so that
// the launch queued behind it sees the termination too, and exits without touching the host mailbox.
// ---------------------------------------------------------------------------------------
// Per-solve constants of the step-kernel chain in DEVICE memory.  Kernel arguments beyond the first 16 dwords are fetched
// from the kernarg segment (host-coherent memory: a ~1 us round trip) before the first instruction that needs one of them
// can run; the steady-state launches therefore take ONLY preloadable arguments (8 pointers / two ints: the
// -amdgpu-kernarg-preload-count build preloads them into SGPRs while the wave is dispatched), the grid size comes as an
// explicit argument rather than from the implicit kernarg block, and everything else — the solver options, the start
// pose, where the trace and the host mailbox live — is read from this block, which launch 0 of the solve (MODE 0: it has
// them as ordinary by-value arguments) leaves behind in device memory.
struct SolveParams {
  clc_options opt;
  Pose7 pose0;
  clc_iteration* trace;
  HostMailbox* mailbox;
  int32_t trace_cap;
  int32_t pad_;
};

// The per-solve block and the two LM state buffers (double-buffered by launch parity) live in one allocation, so one
// pointer argument reaches all three: user SGPRs hold 16 dwords, two of them the kernarg segment pointer — 14 dwords of
// preloaded arguments is all a kernel gets.
struct SolveBlock {
  SolveParams prm;
  LmState st[2];
};

constexpr int PRM_WORDS = (int)(sizeof(SolveParams) / 8);
static_assert(sizeof(SolveParams) % 8 == 0 && offsetof(SolveBlock, st) == sizeof(SolveParams), "SolveBlock is params, then states");
static_assert(LM_STATE_WORDS + PRM_WORDS <= 256, "one block word per thread");

#ifdef CLC_STAMPS
// Debug build only (scripts/stamps_step.py): wall-clock (100 MHz) stamps of every workgroup of the first 64 launches.
constexpr int STAMP_LAUNCHES = 64, STAMP_WGS = 512, STAMP_SLOTS = 16;
static __device__ unsigned long long clc_stamp_buf[STAMP_LAUNCHES][STAMP_WGS][STAMP_SLOTS];
#define CLC_STAMP(slot, tid)                                                                                        \
  do {                                                                                                              \
    if (threadIdx.x == (tid) && blockIdx.x < STAMP_WGS && launch_index < STAMP_LAUNCHES)                            \
      clc_stamp_buf[launch_index][blockIdx.x][slot] = wall_clock64();                                               \
  } while (0)
#define CLC_STAMP_ROW (blockIdx.x < STAMP_WGS && launch_index < STAMP_LAUNCHES ? &clc_stamp_buf[launch_index][blockIdx.x][0] : nullptr)
#else
#define CLC_STAMP(slot, tid) do {} while (0)
#define CLC_STAMP_ROW nullptr
#endif

// LAYOUT 0: compact tiles (ctiles + group table, n = observations); 1: row layout (ctiles = xy rows, groups = row
// descriptors, n = rows; DEEP selects non-temporal loads, WEIGHTED the 3:2 old/young wave shares); 2: rows that carry z.
template <bool WITH_LOSS, bool DEEP, int MODE, int LAYOUT = 0, bool WEIGHTED = true>
__global__ __launch_bounds__(512) void step_kernel(const double* __restrict__ rows_in,
                                                   const double* __restrict__ ctiles,
                                                   const double* __restrict__ groups, const int n, const int grid_parity,
                                                   const int launch_index, double* __restrict__ rows_out,
                                                   SolveBlock* __restrict__ blk, const SolveParams init) {
  // Preloaded: rows_in, ctiles, groups, n, grid_parity, launch_index, rows_out, blk = 13 dwords.  `init` is read by MODE 0
  // only (the other instantiations never touch it, so they never wait for the kernarg segment).
  // grid_parity = number of workgroups | (launch index & 1) << 30: this launch reads LM state st[parity ^ 1], writes st[parity].
  // The front of a steady-state launch is ONE memory round trip: every thread issues, back to back, its 16 partial-row
  // loads and its word of {LM state, per-solve parameters} (vector loads; nothing waits on a scalar load), then the
  // waves sum their rows, issue their first rows of points, and meet at the barrier in front of the controller.
  __shared__ double red[LM_GROUPS][32];
  __shared__ double sh_state[LM_STATE_WORDS + PRM_WORDS];  // [0, LM_STATE_WORDS): LM state; then the SolveParams words
  const long long c0 = clock64();
  CLC_STAMP(0, 0);
  const bool leader = blockIdx.x == 0;
  const int grid = grid_parity & 0x3FFFFFFF;
  const int parity = (grid_parity >> 30) & 1;
  LmState* __restrict__ state_out = &blk->st[parity];
  LmLoads L;
  if (MODE != 0) {
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    // word t of the staging area: state word t (t < LM_STATE_WORDS) or parameter word t - LM_STATE_WORDS
    const int t = threadIdx.x < LM_STATE_WORDS + PRM_WORDS ? (int)threadIdx.x : LM_STATE_WORDS + PRM_WORDS - 1;
    const int w = t < LM_STATE_WORDS ? PRM_WORDS + LM_STATE_WORDS * (parity ^ 1) + t : t - LM_STATE_WORDS;
    L.my_word = reinterpret_cast<const double*>(blk)[w];
    L.passes_before = launch_index - 1;  // launch k consumes pass k - 1: no need to read the counter back
    const int cc = c < NACC ? c : NACC - 1;
    const double* base = rows_in + (size_t)g * NACC + cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) L.v[j] = base[(size_t)(LM_GROUPS * j) * NACC];
    // nothing below may be scheduled in between the loads above (hipcc had pulled the first row add — and its
    // s_waitcnt — in front of the last four loads: a second memory round trip on the critical path)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MODE == 0 && leader && threadIdx.x == 0) blk->prm = init;  // for launches 1, 2, ... of this solve
  const SolveParams* prm = reinterpret_cast<const SolveParams*>(sh_state + LM_STATE_WORDS);  // valid after the barrier
  double inv_lf2 = 0.0;  // set by get_pose: it depends on the options, which arrive with the state
  auto get_pose = [&](PoseU& P) -> bool {
    if (MODE == 0) {
      load_pose(init.pose0.v, P);
      inv_lf2 = make_uniform(1.0 / (init.opt.loss_scale_factor * init.opt.loss_scale_factor));
      return true;
    }
    CLC_STAMP(1, 0);
    __syncthreads();  // state, parameters and row-group sums of every wave are in LDS
    CLC_STAMP(2, 0);
    const clc_options& opt = prm->opt;
    inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
    const bool consumed = lm_tail_after_barrier<false, MODE == 1, 512, true, true>(
        nullptr, leader ? state_out : nullptr, opt, leader ? prm->trace : nullptr, leader ? prm->trace_cap : 0,
        leader ? prm->mailbox : nullptr, red, sh_state, c0, &prm->pose0, L, CLC_STAMP_ROW);
    if (!consumed) {
      // the solve had terminated before this launch: hand the state on (the launch queued behind this one reads
      // the other buffer) and leave; the host mailbox is NOT touched — it may already belong to the next solve
      if (leader && threadIdx.x < LM_STATE_WORDS) reinterpret_cast<double*>(state_out)[threadIdx.x] = sh_state[threadIdx.x];
      return false;
    }
    CLC_STAMP(3, 0);
    const LmState* st = reinterpret_cast<const LmState*>(sh_state);
    const bool running = st->status == CLC_RUNNING;
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = st->x_eval[i];
    load_pose(x, P);
    return running;
  };
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<512, WEIGHTED>(blockIdx.x, grid, threadIdx.x >> 6);
  // Order of the memory traffic of a launch: every wave's share of the previous launch's rows FIRST (57 KB per
  // workgroup through a 64 B/clk vector cache: the controller waits for the slowest wave's rows), and only when a wave
  // has summed its rows does it issue its first rows of points — those have the whole controller to arrive.
  if (MODE != 0) lm_tail_sums<false, 512>(rows_in, grid, red, sh_state, L, nullptr, LM_STATE_WORDS + PRM_WORDS);
  bool active;
  if (LAYOUT != 0) {
    const RowDesc* desc = reinterpret_cast<const RowDesc*>(groups);
    long long r0 = wm.begin(n), r1 = wm.end(n);
    if (!WEIGHTED) {  // equal shares, boundaries at scan starts (wave_split_kernel)
      const WaveRun run = wave_run(desc, n, blockIdx.x, threadIdx.x >> 6);
      r0 = run.begin;
      r1 = run.end;
    }
    active = stream_rows<WITH_LOSS, DEEP, LAYOUT == 2 ? ROWS_DEPTH_Z : ROWS_DEPTH, LAYOUT == 2>(ctiles, desc, r0, r1, lane, get_pose, inv_lf2, acc);
  }
  else if (DEEP) active = stream_ctiles_deep<WITH_LOSS, true, false>(ctiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  else active = stream_ctiles<WITH_LOSS, true, false>(ctiles, groups, n, wm, lane, get_pose, inv_lf2, acc);
  if (!active) return;  // the controller terminated the solve: nothing to evaluate
  CLC_STAMP(4, 0);
  CLC_STAMP(8, 448);
#ifdef CLC_STAMPS
  {  // block_reduce_store<8>, stamped
    __shared__ double wsum_dbg[8][NACC];
    wave_reduce_butterfly(acc, wsum_dbg[threadIdx.x >> 6], lane);
    CLC_STAMP(9, 0);
    __syncthreads();
    CLC_STAMP(10, 0);
    if (threadIdx.x < NACC) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += wsum_dbg[w][threadIdx.x];
      rows_out[(size_t)blockIdx.x * NACC + threadIdx.x] = s;
    }
  }
#else
  block_reduce_store<8>(acc, 0, rows_out + (size_t)blockIdx.x * NACC);
#endif
  CLC_STAMP(5, 0);
  CLC_STAMP(7, 448);
}

static __global__ void lm_init_kernel(LmState* __restrict__ state, const clc_options opt, const Pose7 pose0,
                               unsigned int* __restrict__ ticket_counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *ticket_counter = 0u;
    LmState s;
    lm_init(s, opt, pose0.v);
    *state = s;
  }
}

// ---------------------------------------------------------------------------------------
// K4 — batched independent problems (lockstep): every LM iteration is one launch of
// batched_eval_kernel (blocks_per_problem workgroups stream each still-running problem at
// its own candidate pose) followed by batched_lm_kernel (one thread per problem runs the LM
// controller).  Problems that have terminated cost nothing in later launches.  Problem k
// owns tiles [tile_off[k], tile_off[k+1]) (padded to whole tiles) and n_obs[k] records; no
// communication between problems.
// ---------------------------------------------------------------------------------------
template <bool WITH_LOSS, bool COMPACT, bool NT, bool DEEP>
__global__ __launch_bounds__(BLOCK) void batched_eval_kernel(
    const double* __restrict__ tiles, const double* __restrict__ groups,
    const long long* __restrict__ tile_off, const long long* __restrict__ n_obs,
    const LmState* __restrict__ states, const int blocks_per_problem, const double lf,
    double* __restrict__ partials) {
  const int prob = blockIdx.x / blocks_per_problem;
  const int j = blockIdx.x - prob * blocks_per_problem;
  const LmState* st = states + prob;
  auto get_pose = [&](PoseU& P) -> bool {
    const int32_t s = st->status;  // issued together with the pose loads: one wait
    load_pose(st->x_eval, P);
    return s == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const WaveMap wm = make_wave_map<BLOCK>(j, blocks_per_problem, threadIdx.x >> 6);
  bool active;
  if (COMPACT && DEEP)
    active = stream_ctiles_deep<WITH_LOSS, true, NT>(tiles + tile_off[prob] * CTILE_DOUBLES, groups, n_obs[prob], wm, lane,
                                                     get_pose, inv_lf2, acc);
  else if (COMPACT)
    active = stream_ctiles<WITH_LOSS, true, NT>(tiles + tile_off[prob] * CTILE_DOUBLES, groups, n_obs[prob], wm, lane,
                                                get_pose, inv_lf2, acc);
  else
    active = stream_tiles<WITH_LOSS, true, true, NT>(tiles + tile_off[prob] * TILE_DOUBLES, n_obs[prob], wm, lane,
                                                     get_pose, inv_lf2, acc);
  if (!active) return;  // uniform over the problem's workgroups: it has terminated
  block_reduce_store<BLOCK / 64>(acc, 0, partials + (size_t)blockIdx.x * NACC);
}

// K4 on the row layout: problem k owns rows [prob_row[k], prob_row[k+1]).
// K4 on the row layout: problem k owns rows [prob_row[k], prob_row[k+1]).
// BT = 64 (default): ONE WAVE per workgroup — no LDS, no barrier, the wave's 28 totals go straight to its own partial row
// (blocks_per_problem counts waves).  A problem of 10^4 observations is 160 rows: with 256-thread workgroups the dispatcher
// refills a CU only when the slowest of four waves has finished and every workgroup pays a barrier + LDS pass; with
// single-wave workgroups every wave slot is refilled the moment it frees (measured on one C4 shard: see DESIGN.md K4).
template <bool WITH_LOSS, bool NT, int BT, bool Z = false>
__global__ __launch_bounds__(BT) void batched_rows_eval_kernel(
    const double* __restrict__ xy, const RowDesc* __restrict__ desc, const long long* __restrict__ prob_row,
    const LmState* __restrict__ states, const int blocks_per_problem, const double lf, double* __restrict__ partials) {
  const int prob = blockIdx.x / blocks_per_problem;
  const int j = blockIdx.x - prob * blocks_per_problem;
  const LmState* st = states + prob;
  auto get_pose = [&](PoseU& P) -> bool {
    const int32_t s = st->status;
    load_pose(st->x_eval, P);
    return s == CLC_RUNNING;
  };
  const double inv_lf2 = make_uniform(1.0 / (lf * lf));
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const int lane = threadIdx.x & 63;
  const long long r0 = prob_row[prob], r1 = prob_row[prob + 1];
  const WaveMap wm = make_wave_map<BT>(j, blocks_per_problem, threadIdx.x >> 6);
  if (!stream_rows<WITH_LOSS, NT, ROWS_DEPTH, Z>(xy, desc, r0 + wm.begin(r1 - r0), r0 + wm.end(r1 - r0), lane, get_pose, inv_lf2, acc)) return;
  if (BT == 64)
    wave_reduce_butterfly(acc, partials + (size_t)blockIdx.x * NACC, lane);
  else
    block_reduce_store<BT / 64>(acc, 0, partials + (size_t)blockIdx.x * NACC);
}

static __global__ __launch_bounds__(64) void batched_init_kernel(LmState* __restrict__ states, const clc_options opt,
                                    const double* __restrict__ poses, int n_problems, unsigned int* __restrict__ active,
                                    unsigned int* __restrict__ ticket) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) { *active = 0u; *ticket = 0u; }  // the counters of batched_lm_kernel (two memset launches less per batch)
  if (p >= n_problems) return;
  double x[7];
  for (int i = 0; i < 7; ++i) x[i] = poses[7 * (size_t)p + i];
  LmState s;
  lm_init(s, opt, x);
  states[p] = s;
}

// One thread per problem: fixed-order sum of the problem's block partials, then the LM
// controller.  The workgroups count the problems that still need another evaluation; the
// last-arriving workgroup (ticket pattern, agent-scope atomics) publishes that count and the
// iteration index to the pinned host mailbox, so the host can run launch-ahead without
// blocking (status flips to non-zero when no problem is left running).
// Outcome of one problem: pose + summary to the (pinned, device-mapped) host arrays clc_solve_batched returns, and one
// clc_result_record in DEVICE memory, where clc_gather_results picks it up for the RCCL all-gather (global_index holds
// the local index here; the gather adds the shard's first global index).
__device__ __forceinline__ void batched_write_outcome(const LmState& s, int p, double* __restrict__ poses,
                                                      clc_summary* __restrict__ summaries, double* __restrict__ results) {
  for (int i = 0; i < 7; ++i) poses[7 * (size_t)p + i] = s.x_out[i];
  clc_summary sm;
  lm_fill_summary(s, sm);
  sm.solve_ms = 0.0;
  sm.eval_kernel_ms = 0.0;
  sm.eval_kernel_launches = 0;
  summaries[p] = sm;
  double* r = results + 12 * (size_t)p;
  for (int i = 0; i < 7; ++i) r[i] = s.x_out[i];
  r[7] = sm.final_cost;
  r[8] = sm.initial_cost;
  r[9] = (double)sm.num_iterations;
  r[10] = (double)(sm.termination == CLC_RUNNING ? CLC_FAILURE : sm.termination);
  r[11] = (double)p;
}

// The records-only outcome (clc_solve_batched_gather).  `base` / `host_base`: the communicator's gather buffer in device memory and
// its pinned host twin, both laid out [one record of running totals][the gathered array]; this rank's segment of the array starts
// seg_off doubles in.  The clc_result_record of problem p, GLOBAL index base_index + p, is written twice: into the device segment
// (where the in-place all-gather sends from) and into the host segment (this rank's share of the result needs no copy afterwards).
// Totals (64-bit, device-scope atomics, never reset: the host takes differences): [0] += evaluation passes, [1] += iterations,
// [2] += problems that did not converge, [3] += 1.  The workgroup that brings [3] to `goal` is the last of its launch: everybody else's
// atomics are complete by then (returning atomics, waited for before the count moves), and it copies the four totals to the host twin.
__device__ __forceinline__ void batched_write_record(const LmState& s, int p, double* __restrict__ base, double* __restrict__ host_base,
                                                     const long long seg_off, const double base_index, const unsigned long long goal) {
  clc_summary sm;
  lm_fill_summary(s, sm);
  const int term = sm.termination == CLC_RUNNING ? CLC_FAILURE : sm.termination;
  double v[12];
  for (int i = 0; i < 7; ++i) v[i] = s.x_out[i];
  v[7] = sm.final_cost;
  v[8] = sm.initial_cost;
  v[9] = (double)sm.num_iterations;
  v[10] = (double)term;
  v[11] = base_index + (double)p;
  double* r = base + 12 + seg_off + 12 * (size_t)p;
  double* rh = host_base + 12 + seg_off + 12 * (size_t)p;
  for (int i = 0; i < 12; ++i) { r[i] = v[i]; rh[i] = v[i]; }
  unsigned long long* stats = reinterpret_cast<unsigned long long*>(base);
  const unsigned long long a0 = __hip_atomic_fetch_add(stats + 0, (unsigned long long)sm.num_evaluations, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long a1 = __hip_atomic_fetch_add(stats + 1, (unsigned long long)(sm.num_iterations > 0 ? sm.num_iterations : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long a2 = __hip_atomic_fetch_add(stats + 2, (term == CLC_NO_CONVERGENCE || term == CLC_FAILURE) ? 1ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(a0), "v"(a1), "v"(a2) : "memory");  // the three have been performed before the count moves
  const unsigned long long arrived = __hip_atomic_fetch_add(stats + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  if (arrived == goal) {
    unsigned long long* hs = reinterpret_cast<unsigned long long*>(host_base);
    for (int i = 0; i < 3; ++i) hs[i] = __hip_atomic_load(stats + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hs[3] = arrived;
  }
}

// A problem writes its outcome in the launch in which it terminates (no separate finish launch).
static __global__ __launch_bounds__(64) void batched_lm_kernel(const double* __restrict__ partials, const int blocks_per_problem,
                                  LmState* __restrict__ states, const clc_options opt,
                                  const int n_problems, unsigned int* __restrict__ active,
                                  unsigned int* __restrict__ ticket, const int launch_index,
                                  HostMailbox* mailbox, double* __restrict__ poses, clc_summary* __restrict__ summaries,
                                  double* __restrict__ results) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool still_running = false;
  if (p < n_problems && states[p].status == CLC_RUNNING) {
    double tot[NACC];
    for (int c = 0; c < NACC; ++c) tot[c] = 0.0;
    for (int j = 0; j < blocks_per_problem; ++j) {
      const double* pp = partials + ((size_t)p * blocks_per_problem + j) * NACC;
      for (int c = 0; c < NACC; ++c) tot[c] += pp[c];
    }
    LmState s = states[p];
    LmScratch w;
    lm_advance(s, w, opt, nullptr, 0, finalize_cost(tot[27], opt.use_loss != 0, opt.loss_scale_factor), tot + 21, tot);
    states[p] = s;
    still_running = (s.status == CLC_RUNNING);
    if (!still_running) batched_write_outcome(s, p, poses, summaries, results);
  }
  if (still_running) __hip_atomic_fetch_add(active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {  // every other workgroup has added its count
      const unsigned int a = __hip_atomic_load(active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(active, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mailbox != nullptr) {
        __hip_atomic_store(&mailbox->n_done, (int32_t)(launch_index + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a == 0u) __hip_atomic_store(&mailbox->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// K4, one workgroup per problem (batches of about a thousand problems, C3): the WHOLE solve of a problem in one launch.
// The problem's four waves stream its rows at the candidate, reduce through LDS (block_reduce_store: the same order as
// the lockstep form's single partial row), wave 0 runs the wavefront controller on the state in LDS — the other waves
// start on the next pass at its barrier — until the controller stops.  No kernel boundary between LM iterations, no
// controller launches, no partial rows or LM states through memory, and no lockstep: a problem that needs three
// iterations leaves after three.  Same arithmetic as the lockstep form (bit-identical results).  The loop is bounded by
// the iteration cap whatever the controller does.
template <bool WITH_LOSS, bool NT>
__global__ __launch_bounds__(BLOCK, 2) void batched_solve_kernel(
    const double* __restrict__ xy, const RowDesc* __restrict__ desc, const long long* __restrict__ prob_row,
    const clc_options opt, double* __restrict__ poses, clc_summary* __restrict__ summaries, double* __restrict__ results) {
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ double sh_tot[32];
  __shared__ double sh_park[32 + (sizeof(LmScratch) + 7) / 8];
  const int prob = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  LmState& st = *reinterpret_cast<LmState*>(sh_state);
  if (threadIdx.x == 0) lm_init(st, opt, poses + 7 * (size_t)prob);
  __syncthreads();
  const double inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
  const long long r0 = prob_row[prob], n_rows = prob_row[prob + 1] - r0;
  const WaveMap wm = make_wave_map<BLOCK>(0, 1, wave);
  const long long rb = r0 + wm.begin(n_rows), re = r0 + wm.end(n_rows);
  auto get_pose = [&](PoseU& P) -> bool {
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = st.x_eval[i];
    load_pose(x, P);
    return true;
  };
  // one evaluation pass + the controller; FIRST: the pass at the start point
  auto pass_first = [&]() {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    stream_rows<WITH_LOSS, NT>(xy, desc, rb, re, lane, get_pose, inv_lf2, acc);
    block_reduce_store<BLOCK / 64>(acc, 0, sh_tot);
    __syncthreads();
    if (wave == 0) lm_advance_wave<true>(st, opt, nullptr, 0, sh_tot, sh_park, lane);  // contains the barrier ...
    else __syncthreads();                                                                // ... the other waves meet here
  };
  auto pass_next = [&]() {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    stream_rows<WITH_LOSS, NT>(xy, desc, rb, re, lane, get_pose, inv_lf2, acc);
    block_reduce_store<BLOCK / 64>(acc, 0, sh_tot);
    __syncthreads();
    if (wave == 0) lm_advance_wave<false>(st, opt, nullptr, 0, sh_tot, sh_park, lane);
    else __syncthreads();
  };
  pass_first();
  const int cap = opt.max_num_iterations + 2;
  for (int pass = 0; pass < cap && st.status == CLC_RUNNING; ++pass) pass_next();  // (status: published before the barrier)
  if (wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      if (st.status == CLC_RUNNING) st.status = CLC_FAILURE;  // unreachable: the controller stops at the iteration cap
      batched_write_outcome(st, prob, poses, summaries, results);
    }
  }
}

// Stragglers only: problems the host loop stopped launching for while they were still running (they are reported as
// failures by clc_solve_batched).  Normally every problem has written its outcome in batched_lm_kernel already.
static __global__ void batched_finish_kernel(const LmState* __restrict__ states, int n_problems,
                                      double* __restrict__ poses, clc_summary* __restrict__ summaries,
                                      double* __restrict__ results) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_problems) return;
  if (states[p].status != CLC_RUNNING) return;
  batched_write_outcome(states[p], p, poses, summaries, results);
}

// Send buffer of the all-gather: cap records per rank, the first n_local real (global index = base + local index),
// the rest padding (global_index = -1, everything else 0).
static __global__ void pack_results_kernel(const double* __restrict__ results, long long n_local, long long cap,
                                    double base_index, double* __restrict__ send) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cap) return;
  double* o = send + 12 * k;
  if (k < n_local) {
    const double* r = results + 12 * k;
    for (int i = 0; i < 11; ++i) o[i] = r[i];
    o[11] = base_index + (double)k;
  } else {
    for (int i = 0; i < 11; ++i) o[i] = 0.0;
    o[11] = -1.0;
  }
}

}  // namespace clc

#include "clc_frontend.hpp"
