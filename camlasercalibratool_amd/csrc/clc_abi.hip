// clc_abi.hip — implementation of the C-ABI declared in include/clc.h: HIP host code that
// owns device memory, the stream and the launch sequence of the kernels in
// clc_kernels.hpp.  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=8 clc_abi.hip -o libclc_hip.so
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <link.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/clc.h"
#include "clc_host.hpp"
#include "clc_kernels.hpp"
#include "clc_resident.hpp"
#include "clc_coop.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess)
    std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else
    std::snprintf(buf, sizeof(buf), "%s", what);
  g_last_error = buf;
  return code;
}

#define CLC_HIP(expr)                                                 \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) return fail(CLC_ERR_HIP, #expr, e_);        \
  } while (0)

// Temporaries of one call come from a per-handle pool of device blocks: hipMalloc / hipFree of tens of megabytes cost
// milliseconds each with the system runtime (and hipFree synchronises the device), which made a 0.7 ms
// clc_select_observations take 20 ms when called from a plain C++ program.  A block goes back to the pool on scope
// exit and is handed out again (best fit) to later calls; blocks beyond 1 GiB are really freed.  Every entry point
// synchronises its stream before it returns, so a recycled block is never still in use.
struct DevPool {
  struct Block { void* p; size_t cap; };
  std::vector<Block> free_blocks;
  static constexpr size_t kKeepLimit = (size_t)1 << 30;
  hipError_t acquire(size_t bytes, void** out, size_t* cap) {
    bytes = std::max<size_t>(bytes, 256);
    int best = -1;
    for (int i = 0; i < (int)free_blocks.size(); ++i)
      if (free_blocks[(size_t)i].cap >= bytes && (best < 0 || free_blocks[(size_t)i].cap < free_blocks[(size_t)best].cap)) best = i;
    if (best >= 0 && free_blocks[(size_t)best].cap <= 4 * bytes + ((size_t)1 << 20)) {
      *out = free_blocks[(size_t)best].p;
      *cap = free_blocks[(size_t)best].cap;
      free_blocks.erase(free_blocks.begin() + best);
      return hipSuccess;
    }
    *cap = bytes;
    return hipMalloc(out, bytes);
  }
  void release(void* p, size_t cap) {
    if (!p) return;
    if (cap > kKeepLimit || free_blocks.size() >= 64) { (void)hipFree(p); return; }
    free_blocks.push_back({p, cap});
  }
  void clear() {
    for (const Block& b : free_blocks) (void)hipFree(b.p);
    free_blocks.clear();
  }
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap_bytes = 0;
  DevPool* pool;
  explicit DevBuf(DevPool* pl) : pool(pl) {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) pool->release(p, cap_bytes); }
  hipError_t alloc(size_t count) {
    if (p) { pool->release(p, cap_bytes); p = nullptr; }
    return pool->acquire(std::max<size_t>(count, 1) * sizeof(T), reinterpret_cast<void**>(&p), &cap_bytes);
  }
};

bool all_finite(const double* p, int n) {
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(p[i])) return false;
  return true;
}

constexpr int kDefaultLookahead = 2;
// experiments: CLC_LAUNCH_AHEAD overrides the default launch-ahead depth (options with launch_ahead = 0)
int default_lookahead() {
  static const int v = [] {
    const char* e = std::getenv("CLC_LAUNCH_AHEAD");
    const int n = e ? std::atoi(e) : 0;
    return n > 0 ? n : kDefaultLookahead;
  }();
  return v;
}
constexpr size_t kInfinityCacheBytes = 256u << 20;  // MI355X memory-side cache (MI355X_MICROARCH.md)
constexpr int kDefaultLaunchFlags = 2 | 16 | 32 | 128 | 256 | 512;  // prefetch + compact layout + 512-thread weighted workgroups + step kernel (clc::FLAG_*), tuned on MI355X (scripts/tune_eval.py, scripts/step_check.py)
constexpr int kDefaultBlocksPerCU = 1;   // 4 waves per CU with 2 tiles in flight each     // single-problem solver: launch-ahead depth
constexpr int kSmallDoubles = 512;  // device + pinned scratch for small transfers

}  // namespace

// Lane layout of clc_resident.hpp: j-major point rows, lane descriptors, row offsets per problem.
struct ResLayout {
  double* d_xy = nullptr;
  size_t xy_cap = 0;
  double* d_desc = nullptr;  // clc::ResLane [P * lanes]
  size_t desc_cap = 0;
  double* d_row = nullptr;   // unsigned int [P + 1]
  size_t row_cap = 0;
  int lanes = 0;             // lanes per problem of the built layout (256 / 512)
  int max_ppl = 0;           // largest points-per-lane over the problems
  int uni_ppl = -1;          // >= 0: every problem has this many points per lane
  long long rows = 0;        // j-rows in all
  bool ok = false;
};

constexpr long long kCoopBackoff0 = 16;  // solves the cooperative path rests after its first abort (doubles with every further one)

struct clc_handle {
  DevPool pool;  // temporaries of the entry points (DevBuf)
  int device = 0;
  int num_cus = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // single problem
  double* d_tiles = nullptr;
  size_t tiles_cap_bytes = 0;
  size_t n_obs = 0;
  // compact copy of the same observations (28 B/obs), built at upload when they compress
  double* d_ctiles = nullptr;
  size_t ctiles_cap_bytes = 0;
  double* d_groups = nullptr;
  size_t groups_cap_bytes = 0;
  long long n_groups = 0;
  bool compact_ok = false;
  // row layout of the same observations (clc_rows.hpp): xy rows + row descriptors
  double* d_rxy = nullptr;
  size_t rxy_cap_bytes = 0;
  double* d_rdesc = nullptr;
  size_t rdesc_cap_bytes = 0;
  long long n_rows = 0;
  bool rows_ok = false;
  bool rows_z = false;  // the rows carry z (some record has p.z != 0): ROW_DOUBLES_Z doubles per row
  int split_grid = -1;  // grid the wave split table behind d_rdesc was built for (-1: none)
  // resident pose-major scans (clc_store_observations): device copies + the host-side CSR offsets
  double* d_sq = nullptr; size_t sq_cap = 0;     // tag_q (w,x,y,z) [P*4]
  double* d_st = nullptr; size_t st_cap = 0;     // tag_t [P*3]
  double* d_spts = nullptr; size_t spts_cap = 0; // points [M*3]
  double* d_sptl = nullptr; size_t sptl_cap = 0; // points_on_line [ML*3]
  double* d_soff = nullptr; size_t soff_cap = 0; // pts_off [P+1], ptl_off [P+1], rec_off [P+1] as long long
  std::vector<long long> s_pts_off, s_ptl_off;
  int store_poses = -1;                          // -1: nothing stored
  int64_t store_generation = 0;                  // bumped by every successful clc_store_observations (clc_store_generation)
  // launch geometry
  int grid_override = 0;
  int launch_flags = kDefaultLaunchFlags;
  int auto_disable = 0;     // clc_set_auto_paths: 1 no cooperative solve, 2 no single-workgroup resident solve, 4 cooperative kernel's controller in the single-workgroup kernel
  bool launch_auto = true;  // default flags: size-dependent choices (deep pipeline) are made per launch
  double* d_partials = nullptr;
  int partials_cap_blocks = 0;
  // LM state
  clc::SolveBlock* d_block = nullptr;  // {per-solve constants of the step-kernel chain, LM state x 2}: one allocation
  clc::LmState* d_state = nullptr;     // = &d_block->st[0]
  clc_iteration* d_trace = nullptr;
  int trace_cap = 0;
  // scratch
  double* d_small = nullptr;
  double* h_small = nullptr;  // pinned
  clc::HostMailbox* h_mailbox = nullptr;  // pinned, device-visible
  clc::HostMailbox* d_mailbox = nullptr;  // device address of the same memory
  std::vector<hipEvent_t> ev;
  // batched problems
  double* d_btiles = nullptr;
  size_t btiles_cap_bytes = 0;
  double* d_bctiles = nullptr;
  size_t bctiles_cap_bytes = 0;
  double* d_bgroups = nullptr;
  size_t bgroups_cap_bytes = 0;
  long long bn_groups = 0;
  bool bcompact_ok = false;
  double* d_brxy = nullptr;
  size_t brxy_cap_bytes = 0;
  double* d_brdesc = nullptr;
  size_t brdesc_cap_bytes = 0;
  long long bn_rows = 0;
  bool brows_ok = false;
  bool brows_z = false;
  long long* d_prob_row = nullptr;  // [P+1] first row of every problem
  // resident ("lane") layouts (clc_resident.hpp): of the batched problems, and of a single problem small enough for one workgroup
  ResLayout bres, sres;
  // cooperative whole-GPU solve of one problem (clc_coop.hpp): the problem's lane layout in 256 chunks, the exchange boards, the
  // next free pass tag; disabled on the handle after a launch that timed out (the step chain takes over)
  ResLayout cres;
  clc::CoopBoard* d_board = nullptr;
  unsigned int coop_tag = 1;
  int coop_checked = 0;  // 0: co-residency not checked yet, 1: 256 workgroups fit the device, -1: they do not
  // after a launch that aborted the path rests for `coop_backoff` eligible solves (16, doubling with every further abort up to 2^20:
  // a GPU shared with long-running kernels of somebody else settles on the step chain; a one-off collision costs the first-pass
  // census timeout, 0.2 ms, once)
  long long coop_eligible = 0, coop_retry_at = 0, coop_backoff = kCoopBackoff0;
  int coop_aborts = 0;
  long long coop_solves = 0;
  int coop_test_drop = 0;  // test hook: launch the next cooperative solve this many workgroups short (its exchange must time out)
  // single-problem resident solve: start pose in / result out through page-locked, device-mapped host memory
  double* h_spose = nullptr;          // [7] host view
  double* d_spose = nullptr;          // device view of the same allocation
  clc_summary* h_ssummary = nullptr;
  clc_summary* d_ssummary = nullptr;
  long long* d_tile_off = nullptr;
  long long* d_nobs = nullptr;
  // batched poses / summaries live in pinned, device-mapped host memory: the init kernel reads the start poses and the
  // finish kernel writes the results straight over PCIe (57 + 64 KB at C3) — three staged hipMemcpy calls through
  // pageable memory cost ~35 us each, a fifth of a C3 batch
  double* h_poses = nullptr;            // host view
  double* d_poses = nullptr;            // device view of the same allocation
  clc_summary* h_summaries = nullptr;
  clc_summary* d_summaries = nullptr;
  double* d_results = nullptr;      // clc_result_record per problem of the last clc_solve_batched (device; clc_gather_results)
  size_t results_valid = 0;         // number of valid records in d_results
  unsigned int* d_queue = nullptr;  // small device counter (active problems)
  unsigned int* d_ticket = nullptr; // arrival counter of the fused evaluation+controller launch
  double* d_partials_b = nullptr;   // second row buffer (inside the d_partials allocation) for the step kernel
  clc::LmState* d_state_b = nullptr;  // second LM state buffer for the step kernel
  clc::LmState* d_states = nullptr;
  double* d_bpartials = nullptr;
  size_t bpartials_cap_blocks = 0;
  long long batch_max_tiles = 0;
  long long batch_max_rows = 0;  // most rows of the row layout any one problem owns (exact, from prob_row)
  size_t batch_total_tiles = 0;
  size_t n_problems = 0;
  size_t problems_cap = 0;
};

namespace {

int eval_grid(const clc_handle* h, size_t n) {
  const long long tiles = (long long)((n + clc::TILE - 1) / clc::TILE);
  const bool big = (h->launch_flags & clc::FLAG_WG512) != 0;
  // Every CU takes a share (the tile map is proportional, a wave may own zero tiles): up to one
  // workgroup per CU keeps the partial-row reduction of lm_kernel short; with 256-thread workgroups,
  // arrays long enough to give every wave >= 16 tiles are streamed with 2 workgroups per CU.
  const int per_cu = (!big && tiles >= 16LL * (clc::BLOCK / 64) * 2 * h->num_cus) ? 2 * kDefaultBlocksPerCU : kDefaultBlocksPerCU;
  const long long cap = h->grid_override > 0 ? h->grid_override : (long long)per_cu * h->num_cus;
  const long long want = tiles < 1 ? 1 : tiles;  // never more workgroups than tiles
  return (int)(want < cap ? want : cap);
}

// Partial rows.  The capacity is a whole number of 256-row rounds: the controller reads rows in rounds of 256 from
// unclamped addresses and masks the ones beyond the grid afterwards (clc::lm_tail), so every round must be mapped.
int ensure_partials(clc_handle* h, int blocks) {
  if (blocks <= h->partials_cap_blocks) return CLC_OK;
  const int cap = (blocks + clc::BLOCK - 1) / clc::BLOCK * clc::BLOCK;
  if (h->d_partials) CLC_HIP(hipFree(h->d_partials));
  h->d_partials = nullptr;
  h->partials_cap_blocks = 0;
  // two buffers: the step kernel alternates between them by launch parity
  CLC_HIP(hipMalloc(&h->d_partials, 2 * sizeof(double) * (size_t)cap * clc::NACC9));
  // on the handle's stream: a null-stream memset is not ordered against kernels on a non-blocking stream
  CLC_HIP(hipMemsetAsync(h->d_partials, 0, 2 * sizeof(double) * (size_t)cap * clc::NACC9, h->stream));
  h->partials_cap_blocks = cap;
  h->d_partials_b = h->d_partials + (size_t)cap * clc::NACC9;
  return CLC_OK;
}

int ensure_trace(clc_handle* h, int cap) {
  if (cap <= h->trace_cap) return CLC_OK;
  if (h->d_trace) CLC_HIP(hipFree(h->d_trace));
  h->d_trace = nullptr;
  CLC_HIP(hipMalloc(&h->d_trace, sizeof(clc_iteration) * (size_t)cap));
  h->trace_cap = cap;
  return CLC_OK;
}

int ensure_events(clc_handle* h, size_t n) {
  while (h->ev.size() < n) {
    hipEvent_t e;
    CLC_HIP(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  return CLC_OK;
}

// Row layout in use for the single-problem array?
// With the library's default flags, arrays below ~2x10^5 observations keep the per-point compact layout: a launch is pure
// fixed cost there and the row kernel's 16 prologue loads + per-scan expansion make it 0.4-0.5 us longer per LM iteration
// (8.5 vs 9.0 us at 5.5x10^3 observations, 8.8 vs 9.2 at 10^5; 13.2 vs 11.1 at 10^6 — scripts/r02_ab.py).
bool use_rows(const clc_handle* h) {
  if ((h->launch_flags & clc::FLAG_ROWS) == 0 || !h->rows_ok) return false;
  return !h->launch_auto || !h->compact_ok || h->n_obs >= 200000;
}
bool use_brows(const clc_handle* h) { return (h->launch_flags & clc::FLAG_ROWS) != 0 && h->brows_ok; }
// Rows streamed from HBM rather than the Infinity Cache (> 1.5x its size) are loaded non-temporally.
bool rows_nontemporal(const clc_handle* h, long long n_rows, bool z = false) {
  const size_t bytes = (size_t)n_rows * ((z ? clc::ROW_DOUBLES_Z : clc::ROW_DOUBLES) * sizeof(double) + sizeof(clc::RowDesc));
  return (h->launch_flags & clc::FLAG_NONTEMPORAL) != 0 || (h->launch_auto && bytes > kInfinityCacheBytes + kInfinityCacheBytes / 2);
}

// The wave split table of the row layout's equal-shares mode (clc_kernels.hpp wave_split_kernel): rebuilt, on the
// handle's stream in front of the launches that read it, when the grid changed since the last upload.
void ensure_wave_split(clc_handle* h, int grid) {
  if (h->split_grid == grid) return;
  const clc::RowDesc* desc = reinterpret_cast<const clc::RowDesc*>(h->d_rdesc);
  int* table = reinterpret_cast<int*>(reinterpret_cast<char*>(h->d_rdesc) + ((size_t)h->n_rows + 1) * sizeof(clc::RowDesc));
  const int total = grid * 8 + 1;
  hipLaunchKernelGGL(clc::wave_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, desc, (int)h->n_rows,
                     grid, table);
  h->split_grid = grid;
}

template <bool WITH_LOSS, bool WITH_JAC>
void launch_eval_v(clc_handle* h, int grid, const double* d_pose, const int32_t* d_status, double lf,
                   const clc::Pose7& pose_arg, int use_pose_arg) {
  const int fl = h->launch_flags;
#define CLC_LAUNCH(PF, NT, CP, BT)                                                                          \
  hipLaunchKernelGGL((clc::eval_kernel<WITH_LOSS, WITH_JAC, PF, NT, CP, BT>), dim3(grid), dim3(BT), 0,       \
                     h->stream, (CP) ? h->d_ctiles : h->d_tiles, h->d_groups, (long long)h->n_obs, d_pose,   \
                     d_status, lf, fl, h->d_partials, pose_arg, use_pose_arg)
  const bool pf = (fl & clc::FLAG_PREFETCH) != 0, nt = (fl & clc::FLAG_NONTEMPORAL) != 0;
  const bool cp = (fl & clc::FLAG_COMPACT) != 0 && h->compact_ok;
  const bool big = (fl & clc::FLAG_WG512) != 0;
  if (use_rows(h)) {  // row layout: the Jacobian comes with the moments, a cost-only pass would save nothing
    const bool rnt = rows_nontemporal(h, h->n_rows, h->rows_z);
    if (h->rows_z) {  // rows that carry z: 3:2 wave shares, 8 rows in flight
#define CLC_LAUNCH_RZ(NT, BT)                                                                                              \
  hipLaunchKernelGGL((clc::eval_rows_kernel<WITH_LOSS, NT, BT, true, clc::ROWS_DEPTH, true>), dim3(grid), dim3(BT), 0, h->stream, h->d_rxy, \
                     reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, d_pose, d_status, lf, fl, h->d_partials, pose_arg, use_pose_arg)
      if (big) { if (rnt) CLC_LAUNCH_RZ(true, 512); else CLC_LAUNCH_RZ(false, 512); }
      else { if (rnt) CLC_LAUNCH_RZ(true, 256); else CLC_LAUNCH_RZ(false, 256); }
#undef CLC_LAUNCH_RZ
      return;
    }
    // Equal, scan-aligned shares (flag 512) pay where a wave's share is a scan or two; the evaluation kernel ALONE with
    // tens of rows per wave and more is 3-7 % faster with the 3:2 old/young shares (scripts/r02_ab.py: 6.2 vs 6.8 us at
    // 1e6 observations, but 15.4 vs 14.7 at 4e6 and 45.1 vs 42.1 at 1.6e7) — the step kernel is not (its wave 0 starts
    // late anyway): it keeps the equal shares at every size.
    const bool eq = (fl & clc::FLAG_EQUAL_WAVES) != 0 && !(h->launch_auto && h->n_rows > 16LL * 8 * grid);
    // rows in flight per wave: 8 while the array is served by the Infinity Cache, 12 (206 VGPRs, still 2 waves/SIMD) when it
    // streams from HBM with non-temporal loads — throughput there tracks the bytes in flight per CU (profiles/r03_occupancy.md:
    // 4 rows 0.40 of peak, 8 rows 0.81, 12 rows 0.82-0.83, 16 rows 0.81; 3 waves/SIMD cannot hold more than 6 rows each: 0.80)
#define CLC_LAUNCH_R(NT, BT, WG)                                                                              \
  hipLaunchKernelGGL((clc::eval_rows_kernel<WITH_LOSS, NT, BT, WG, (NT) ? 12 : clc::ROWS_DEPTH>), dim3(grid), dim3(BT), 0, h->stream, h->d_rxy, \
                     reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, d_pose, d_status, lf, fl,       \
                     h->d_partials, pose_arg, use_pose_arg)
#ifdef CLC_EVAL_VARIANTS
    // Occupancy experiment of profiles/r03_occupancy.md (scripts/r03_occupancy.py; -DCLC_EVAL_VARIANTS build only):
    // CLC_EVAL_VARIANT = <threads>x<rows in flight per wave>: 768x4, 768x6 (3 waves/SIMD), 512x4, 512x12, 512x16 (2 waves/SIMD)
    static const int variant = [] {
      const char* e = std::getenv("CLC_EVAL_VARIANT");
      const char* names[] = {"768x4", "512x4", "512x12", "512x16", "768x6"};
      for (int i = 0; e && i < 5; ++i)
        if (std::strcmp(e, names[i]) == 0) return i + 1;
      return 0;
    }();
#define CLC_LAUNCH_VAR(BT, DEPTH)                                                                                                  \
  do {                                                                                                                             \
    if (rnt) hipLaunchKernelGGL((clc::eval_rows_kernel<WITH_LOSS, true, BT, true, DEPTH>), dim3(grid), dim3(BT), 0, h->stream, h->d_rxy,  \
                                reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, d_pose, d_status, lf, fl, h->d_partials, pose_arg, use_pose_arg); \
    else hipLaunchKernelGGL((clc::eval_rows_kernel<WITH_LOSS, false, BT, true, DEPTH>), dim3(grid), dim3(BT), 0, h->stream, h->d_rxy,     \
                            reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, d_pose, d_status, lf, fl, h->d_partials, pose_arg, use_pose_arg);    \
    return;                                                                                                                        \
  } while (0)
    if (variant == 1) CLC_LAUNCH_VAR(768, 4);
    if (variant == 2) CLC_LAUNCH_VAR(512, 4);
    if (variant == 3) CLC_LAUNCH_VAR(512, 12);
    if (variant == 4) CLC_LAUNCH_VAR(512, 16);
    if (variant == 5) CLC_LAUNCH_VAR(768, 6);
#undef CLC_LAUNCH_VAR
#endif
    if (big) {
      if (eq) ensure_wave_split(h, grid);
      if (eq) { if (rnt) CLC_LAUNCH_R(true, 512, false); else CLC_LAUNCH_R(false, 512, false); }
      else { if (rnt) CLC_LAUNCH_R(true, 512, true); else CLC_LAUNCH_R(false, 512, true); }
    } else {
      if (rnt) CLC_LAUNCH_R(true, 256, true); else CLC_LAUNCH_R(false, 256, true);
    }
#undef CLC_LAUNCH_R
    return;
  }
  // Compact layout: the deep pipeline (two tiles of points in flight per wave) pays only when the array streams
  // from HBM, i.e. no longer fits the 256 MiB Infinity Cache (scripts/size_sweep.py: +10 % at 9e8 B, -8 % at 1e8 B).
  // Well beyond the cache (> 1.5x) the streamed tiles are also loaded non-temporally (+5-8 % at 4.5e8-9e8 B; plain
  // loads win while the array is cache-resident, and at 2.9e8 B — C3 — there is nothing in it).
  const bool beyond_cache = h->launch_auto && (size_t)h->n_obs * 28 > kInfinityCacheBytes;
  const bool deep = (fl & clc::FLAG_DEEP) != 0 || beyond_cache;
  if (cp) {
    const bool pf = deep;
    const bool nt = (fl & clc::FLAG_NONTEMPORAL) != 0 ||
                    (h->launch_auto && (size_t)h->n_obs * 28 > kInfinityCacheBytes + kInfinityCacheBytes / 2);
    if (big && pf) { if (nt) CLC_LAUNCH(true, true, true, 512); else CLC_LAUNCH(true, false, true, 512); }
    else if (big) { if (nt) CLC_LAUNCH(false, true, true, 512); else CLC_LAUNCH(false, false, true, 512); }
    else if (pf) { if (nt) CLC_LAUNCH(true, true, true, 256); else CLC_LAUNCH(true, false, true, 256); }
    else { if (nt) CLC_LAUNCH(false, true, true, 256); else CLC_LAUNCH(false, false, true, 256); }
  } else if (big) {
    if (nt) CLC_LAUNCH(true, true, false, 512); else CLC_LAUNCH(true, false, false, 512);
  }
  else if (pf && nt) CLC_LAUNCH(true, true, false, 256);
  else if (pf) CLC_LAUNCH(true, false, false, 256);
  else if (nt) CLC_LAUNCH(false, true, false, 256);
  else CLC_LAUNCH(false, false, false, 256);
#undef CLC_LAUNCH
}

template <bool WITH_JAC>
void launch_eval(clc_handle* h, int grid, bool with_loss, const double* d_pose,
                 const int32_t* d_status, double lf, const clc::Pose7* pose_arg = nullptr) {
  const clc::Pose7 zero = {};
  const clc::Pose7& pa = pose_arg ? *pose_arg : zero;
  if (with_loss) launch_eval_v<true, WITH_JAC>(h, grid, d_pose, d_status, lf, pa, pose_arg ? 1 : 0);
  else launch_eval_v<false, WITH_JAC>(h, grid, d_pose, d_status, lf, pa, pose_arg ? 1 : 0);
}

int retile_into(clc_handle* h, const double* d_aos, size_t n, double** d_tiles, size_t* cap_bytes) {
  const size_t n_padded = ((n + clc::TILE - 1) / clc::TILE) * clc::TILE;
  const size_t bytes = std::max<size_t>(n_padded, clc::TILE) * 8 * sizeof(double);
  if (bytes > *cap_bytes) {
    if (*d_tiles) CLC_HIP(hipFree(*d_tiles));
    *d_tiles = nullptr;
    *cap_bytes = 0;
    CLC_HIP(hipMalloc(d_tiles, bytes));
    *cap_bytes = bytes;
  }
  if (n_padded > 0) {
    const int threads = 256;
    const long long blocks = ((long long)n_padded + threads - 1) / threads;
    hipLaunchKernelGGL(clc::retile_kernel, dim3((unsigned)blocks), dim3(threads), 0, h->stream, d_aos,
                       *d_tiles, (long long)n, (long long)n_padded);
    CLC_HIP(hipGetLastError());
  }
  return CLC_OK;
}

// Upload-time re-encoding of staged AoS records, entirely on the device (O(1) host work, a few words copied back):
//   * scans = runs of records with bit-identical (n, d, scale) (and never across two problems): flags, prefix sum ->
//     scan index per record, scan starts;
//   * compact layout (clc_kernels.hpp "Compact layout"): group table + 28-byte tiles, bitwise lossless;
//   * row layout (clc_rows.hpp): every scan padded to whole rows of 64 points, (x, y) rows + one descriptor per row —
//     only when every record has p.z == 0 and the padding at most doubles the array.
// Nothing is kept when the records do not group at least 4:1 (hand-made arrays without scan structure).
int ensure_bytes(double** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap && *p) return CLC_OK;
  if (*p) CLC_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  CLC_HIP(hipMalloc(p, bytes));
  *cap = bytes;
  return CLC_OK;
}

template <class TIn>
int device_scan(clc_handle* h, const TIn* d_in, long long n, unsigned int minus_one, unsigned int* d_out,
                unsigned long long* d_totals /* [blocks + 1] */) {
  const long long blocks = (n + clc::SCAN_CHUNK - 1) / clc::SCAN_CHUNK;
  hipLaunchKernelGGL((clc::scan_block_totals_kernel<TIn>), dim3((unsigned)blocks), dim3(clc::SCAN_THREADS), 0, h->stream,
                     d_in, n, d_totals);
  hipLaunchKernelGGL(clc::scan_totals_kernel, dim3(1), dim3(clc::SCAN_THREADS), 0, h->stream, d_totals, blocks);
  hipLaunchKernelGGL((clc::scan_apply_kernel<TIn>), dim3((unsigned)blocks), dim3(clc::SCAN_THREADS), 0, h->stream, d_in, n,
                     d_totals, minus_one, d_out);
  CLC_HIP(hipGetLastError());
  return CLC_OK;
}

struct LayoutTargets {
  double** d_ct; size_t* ct_cap; double** d_gr; size_t* gr_cap; long long* n_groups; bool* compact_ok;
  double** d_rxy; size_t* rxy_cap; double** d_rdesc; size_t* rdesc_cap; long long* n_rows; bool* rows_ok;
  long long** d_prob_row;  // nullptr for the single-problem array
  ResLayout* res = nullptr;  // also build the on-chip resident ("lane") layout (clc_resident.hpp) into this
  bool* rows_z = nullptr;    // out: the rows carry z
  ResLayout* coop = nullptr; // single problem only: its lane layout in COOP_WGS chunks (clc_coop.hpp)
};

// Resident layout limits: what the instantiations of resident_solve_kernel hold per lane (registers + LDS).
// 256-lane form: 256-thread workgroups, two problems per CU; 512-lane form: one 512-thread workgroup per CU (problems with more
// than 256 scans, or flag 8192).  Both hold 512 x 22 = 256 x 44 - 512 points at most.
constexpr int kResPR256 = 23, kResPL256 = 19, kResPR512 = 4, kResPL512 = 18;
// controller of the batched launches (clc_resident.hpp CTRL): 4-wave form / 8-wave form
constexpr int kResCtrl4 = 0, kResCtrl8 = 0;

// The lane layout of the batched problems (clc_resident.hpp) from the staged records and their scan structure: plan
// (points per lane of every problem, on the device), offsets (O(P) on the host), lane descriptors + j-major point rows.
// Leaves L.ok false — and the streaming layouts in charge — when some problem does not fit a workgroup.
int build_resident(clc_handle* h, ResLayout& L, int first_try, const double* d_aos, long long n, size_t P, size_t G,
                   const long long* d_rec_off, const unsigned int* d_gid, const long long* d_starts, int max_ppl_override = 0) {
  L.ok = false;
  L.lanes = 0;
  L.max_ppl = 0;
  L.rows = 0;
  if ((h->launch_flags & clc::FLAG_NO_RESIDENT) != 0) return CLC_OK;
  const int threads = 256;
  DevBuf<unsigned int> bppl(&h->pool), bfail(&h->pool);
  CLC_HIP(bppl.alloc(P));
  CLC_HIP(bfail.alloc(1));
  std::vector<unsigned int> ppl(P);
  int lanes = 0;
  for (int nl = first_try; nl <= (max_ppl_override > 0 ? first_try : 512) && lanes == 0; nl *= 2) {
    unsigned int failed = 0;
    CLC_HIP(hipMemsetAsync(bfail.p, 0, sizeof(unsigned int), h->stream));
    hipLaunchKernelGGL(clc::res_plan_kernel, dim3((unsigned)((P + threads - 1) / threads)), dim3(threads), 0, h->stream, d_rec_off,
                       d_gid, d_starts, (long long)P, n, (long long)G, nl, max_ppl_override > 0 ? max_ppl_override : (nl == 256 ? kResPR256 + kResPL256 : kResPR512 + kResPL512), bppl.p, bfail.p);
    CLC_HIP(hipGetLastError());
    CLC_HIP(hipMemcpyAsync(&failed, bfail.p, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipMemcpyAsync(ppl.data(), bppl.p, sizeof(unsigned int) * P, hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    if (!failed) lanes = nl;
  }
  if (lanes == 0) return CLC_OK;  // some problem is too long (or has too many scans) for one workgroup
  std::vector<unsigned int> row(P + 1, 0u);
  unsigned long long total = 0;
  unsigned int max_ppl = 0;
  bool uniform = true;
  for (size_t k = 0; k < P; ++k) {
    row[k] = (unsigned int)total;
    total += ppl[k];
    max_ppl = std::max(max_ppl, ppl[k]);
    uniform = uniform && ppl[k] == ppl[0];
  }
  if (total >= 0xFFFFFFF0ull) return CLC_OK;
  row[P] = (unsigned int)total;
  int rc = ensure_bytes(&L.d_row, &L.row_cap, (P + 1) * sizeof(unsigned int));
  if (rc != CLC_OK) return rc;
  rc = ensure_bytes(&L.d_desc, &L.desc_cap, P * (size_t)lanes * sizeof(clc::ResLane));
  if (rc != CLC_OK) return rc;
  // one padding row: the kernel's loads run unconditionally from clamped row indices (an empty last problem reads it)
  rc = ensure_bytes(&L.d_xy, &L.xy_cap, ((size_t)total + 1) * (size_t)lanes * 2 * sizeof(double));
  if (rc != CLC_OK) return rc;
  CLC_HIP(hipMemcpyAsync(L.d_row, row.data(), (P + 1) * sizeof(unsigned int), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemsetAsync(L.d_xy + (size_t)total * (size_t)lanes * 2, 0, (size_t)lanes * 2 * sizeof(double), h->stream));
  const unsigned int* d_row = reinterpret_cast<const unsigned int*>(L.d_row);
  clc::ResLane* d_desc = reinterpret_cast<clc::ResLane*>(L.d_desc);
  if (lanes == 256)
    hipLaunchKernelGGL((clc::res_build_kernel<256>), dim3((unsigned)P), dim3(256), 0, h->stream, d_aos, d_rec_off, d_gid, d_starts, n,
                       (long long)G, d_row, d_desc, L.d_xy);
  else
    hipLaunchKernelGGL((clc::res_build_kernel<512>), dim3((unsigned)P), dim3(512), 0, h->stream, d_aos, d_rec_off, d_gid, d_starts, n,
                       (long long)G, d_row, d_desc, L.d_xy);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));  // `row` is a host temporary
  L.lanes = lanes;
  L.max_ppl = (int)max_ppl;
  L.uni_ppl = uniform && P > 0 ? (int)ppl[0] : -1;
  L.rows = (long long)total;
  L.ok = true;
  return CLC_OK;
}

int build_layouts(clc_handle* h, const double* d_aos, size_t n_total, const std::vector<long long>& rec_off,
                  const std::vector<long long>& tile_off, const LayoutTargets& T) {
  *T.compact_ok = false;
  *T.rows_ok = false;
  *T.n_groups = 0;
  *T.n_rows = 0;
  const size_t P = rec_off.size() - 1;
  if (n_total == 0 || P == 0) return CLC_OK;
  if (n_total >= 0xFFFFFFF0ull) return CLC_OK;  // scan indices are 32-bit; such arrays keep the 64-byte tiles
  const long long n = (long long)n_total;
  const int threads = 256;
  const long long scan_blocks = (n + clc::SCAN_CHUNK - 1) / clc::SCAN_CHUNK;
  DevBuf<unsigned char> bflag(&h->pool);
  DevBuf<unsigned int> bgid(&h->pool), bzflag(&h->pool);
  DevBuf<unsigned long long> btotals(&h->pool);
  DevBuf<long long> broff(&h->pool), btoff(&h->pool);
  CLC_HIP(bflag.alloc(n_total));
  CLC_HIP(bgid.alloc(n_total));
  CLC_HIP(bzflag.alloc(1));
  CLC_HIP(btotals.alloc((size_t)scan_blocks + 1));
  CLC_HIP(broff.alloc(P + 1));
  CLC_HIP(btoff.alloc(P + 1));
  CLC_HIP(hipMemcpyAsync(broff.p, rec_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(btoff.p, tile_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemsetAsync(bzflag.p, 0, sizeof(unsigned int), h->stream));
  hipLaunchKernelGGL(clc::scan_flag_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, h->stream, d_aos,
                     n, bflag.p, bzflag.p);
  hipLaunchKernelGGL(clc::mark_problem_starts_kernel, dim3((unsigned)((P + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, broff.p, (long long)P, n, bflag.p);
  CLC_HIP(hipGetLastError());
  int rc = device_scan<unsigned char>(h, bflag.p, n, 1u, bgid.p, btotals.p);
  if (rc != CLC_OK) return rc;
  unsigned int last_gid = 0, any_z = 0;
  CLC_HIP(hipMemcpyAsync(&last_gid, bgid.p + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipMemcpyAsync(&any_z, bzflag.p, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  const size_t G = (size_t)last_gid + 1;
  // fewer than 4 points per scan on average: the streaming layouts do not pay (the step chain keeps the 64-byte tiles) — the on-chip
  // layouts, where a lane carries its own plane anyway, are still built
  const bool sparse = G * 4 > n_total;
  if (sparse && T.res == nullptr && T.coop == nullptr) return CLC_OK;
  DevBuf<long long> bstarts(&h->pool);
  CLC_HIP(bstarts.alloc(G + 1));
  hipLaunchKernelGGL(clc::scan_starts_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, h->stream,
                     bflag.p, bgid.p, n, (long long)G, bstarts.p);
  CLC_HIP(hipGetLastError());
  // ---- compact layout ----
  const size_t total_tiles = (size_t)tile_off[P];
  rc = ensure_bytes(T.d_ct, T.ct_cap, std::max<size_t>(total_tiles, 1) * clc::CTILE_DOUBLES * sizeof(double));
  if (rc != CLC_OK) return rc;
  rc = ensure_bytes(T.d_gr, T.gr_cap, G * clc::GROUP_DOUBLES * sizeof(double));
  if (rc != CLC_OK) return rc;
  hipLaunchKernelGGL(clc::build_groups_dev_kernel, dim3((unsigned)((G + threads - 1) / threads)), dim3(threads), 0, h->stream,
                     d_aos, bstarts.p, (long long)G, *T.d_gr);
  {
    long long max_padded = 0;
    for (size_t k = 0; k < P; ++k) max_padded = std::max(max_padded, (tile_off[k + 1] - tile_off[k]) * clc::TILE);
    const unsigned ydim = (unsigned)std::min<long long>(4096, std::max<long long>(1, (max_padded + threads - 1) / threads));
    hipLaunchKernelGGL(clc::build_ctiles_kernel, dim3((unsigned)P, ydim), dim3(threads), 0, h->stream, d_aos, bgid.p, broff.p,
                       btoff.p, *T.d_ct);
  }
  CLC_HIP(hipGetLastError());
  // ---- row layout ----
  bool rows_ok = false;
  long long R = 0;
  DevBuf<unsigned int> brows(&h->pool), brbeg(&h->pool);
  DevBuf<unsigned long long> btot2(&h->pool);
  const size_t row_doubles = any_z ? clc::ROW_DOUBLES_Z : clc::ROW_DOUBLES;
  if (T.rows_z) *T.rows_z = any_z != 0;
  {
    const long long gblocks = ((long long)G + clc::SCAN_CHUNK - 1) / clc::SCAN_CHUNK;
    CLC_HIP(brows.alloc(G));
    CLC_HIP(brbeg.alloc(G + 1));
    CLC_HIP(btot2.alloc((size_t)gblocks + 1));
    hipLaunchKernelGGL(clc::scan_rows_kernel, dim3((unsigned)((G + threads - 1) / threads)), dim3(threads), 0, h->stream,
                       bstarts.p, (long long)G, brows.p);
    CLC_HIP(hipMemsetAsync(brbeg.p, 0, sizeof(unsigned int), h->stream));
    rc = device_scan<unsigned int>(h, brows.p, (long long)G, 0u, brbeg.p + 1, btot2.p);
    if (rc != CLC_OK) return rc;
    unsigned int total_rows = 0;
    CLC_HIP(hipMemcpyAsync(&total_rows, brbeg.p + G, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    R = (long long)total_rows;
    rows_ok = !sparse && R > 0 && (size_t)R * clc::ROW <= 3 * n_total + 64 * P;  // rows at least a third full on average
  }
  if (rows_ok) {
    // one padding row each: the streaming loop's prologue loads run unconditionally from clamped row indices
    rc = ensure_bytes(T.d_rxy, T.rxy_cap, ((size_t)R + 1) * row_doubles * sizeof(double));
    if (rc != CLC_OK) return rc;
    rc = ensure_bytes(T.d_rdesc, T.rdesc_cap, ((size_t)R + 1) * sizeof(clc::RowDesc) + clc::wave_split_bytes(R));  // + the wave split table
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipMemsetAsync(*T.d_rxy + (size_t)R * row_doubles, 0, row_doubles * sizeof(double), h->stream));
    CLC_HIP(hipMemsetAsync(reinterpret_cast<char*>(*T.d_rdesc) + (size_t)R * sizeof(clc::RowDesc), 0, sizeof(clc::RowDesc), h->stream));
    const long long slots = R * clc::ROW;
    hipLaunchKernelGGL(clc::build_rows_kernel, dim3((unsigned)((slots + threads - 1) / threads)), dim3(threads), 0, h->stream,
                       d_aos, bstarts.p, brbeg.p, (long long)G, R, (int)row_doubles, *T.d_rxy, reinterpret_cast<clc::RowDesc*>(*T.d_rdesc));
    if (T.d_prob_row)
      hipLaunchKernelGGL(clc::problem_rows_kernel, dim3((unsigned)((P + 1 + threads - 1) / threads)), dim3(threads), 0,
                         h->stream, broff.p, bgid.p, brbeg.p, (long long)P, n, R, *T.d_prob_row);
    CLC_HIP(hipGetLastError());
  }
  if (T.res != nullptr) {
    T.res->ok = false;
    if (!any_z) {
      // batches: 256 lanes (two problems per CU) unless flag 8192; a single problem: 512 lanes (it has its CU to itself)
      const int first_try = (T.d_prob_row == nullptr || (h->launch_flags & clc::FLAG_RESIDENT_WG512) != 0) ? 512 : 256;
      rc = build_resident(h, *T.res, first_try, d_aos, n, P, G, broff.p, bgid.p, bstarts.p);
      if (rc != CLC_OK) return rc;
    }
  }
  if (T.coop != nullptr) {
    T.coop->ok = false;
    if (!any_z && P == 1 && n > 0 && h->num_cus >= clc::COOP_WGS && !(T.res != nullptr && T.res->ok)) {
      // the one problem in COOP_WGS chunks of equal record counts (a chunk may begin and end inside a scan: res_scan_extent)
      std::vector<long long> chunk(clc::COOP_WGS + 1);
      for (int c = 0; c <= clc::COOP_WGS; ++c) chunk[c] = (long long)((__int128)n * c / clc::COOP_WGS);
      DevBuf<long long> bchunk(&h->pool);
      CLC_HIP(bchunk.alloc(chunk.size()));
      CLC_HIP(hipMemcpyAsync(bchunk.p, chunk.data(), chunk.size() * sizeof(long long), hipMemcpyHostToDevice, h->stream));
      CLC_HIP(hipStreamSynchronize(h->stream));
      rc = build_resident(h, *T.coop, clc::COOP_NL, d_aos, n, (size_t)clc::COOP_WGS, G, bchunk.p, bgid.p, bstarts.p, clc::COOP_PR + clc::COOP_PL);
      if (rc != CLC_OK) return rc;
    }
  }
  CLC_HIP(hipStreamSynchronize(h->stream));  // the temporaries above are freed on return
  *T.n_groups = (long long)G;
  *T.compact_ok = !sparse;
  *T.n_rows = R;
  *T.rows_ok = rows_ok;
  return CLC_OK;
}

// Launch geometry of the batched solver (shared by clc_solve_batched and the timing hook).
struct BatchedLaunch {
  int bpp = 1;            // workgroups per problem
  size_t n_blocks = 0;
  int lm_threads = 64;
  unsigned lm_blocks = 0;
  bool compact = false, deep = false, nt = false;
  bool rows = false, rows_nt = false, rows_wave = false;
  bool one_wave = false;  // rows_wave with exactly one wave per problem
  bool whole_solve = false;  // batched_solve_kernel: one workgroup per problem, the whole solve in one launch
  bool resident = false;     // resident_solve_kernel: the same with the problem read from HBM once and kept on chip
  bool res_nt = false;
};

int batched_launch_setup(clc_handle* h, const clc_options& /*opt*/, BatchedLaunch* bl) {
  const size_t P = h->n_problems;
  bl->rows = use_brows(h);
  // one wave per workgroup once the batch is many times wider than the chip (C4 shard: 8 192 problems, -5...7 % per
  // batch); for batches of about a thousand problems the 256-thread form is 3-4 % ahead (scripts/r02_shard_step_timing.py)
  bl->rows_wave = bl->rows && (h->launch_flags & clc::FLAG_BATCHED_WG256) == 0 && (!h->launch_auto || P >= 8 * (size_t)h->num_cus);
  // enough workgroups to fill the chip: >= 2 per CU in total, never more than one per 4 tiles
  const size_t target_blocks = h->grid_override > 0 ? (size_t)h->grid_override : 4 * (size_t)h->num_cus;
  int bpp = (int)((target_blocks + P - 1) / P);
  const long long max_tiles = h->batch_max_tiles;
  const int bpp_cap = (int)std::max<long long>(1, max_tiles / 4);
  bpp = std::max(1, std::min(bpp, bpp_cap));
  // batched_lm_kernel sums a problem's partial rows in ONE thread: with hundreds of rows per problem (a handful of long
  // problems) that sum took longer than the evaluation (393 us per pass at 4 problems x 9.6e4 observations, 256 rows each)
  bpp = std::min(bpp, 16);
  // single-wave workgroups: as many waves as the 256-thread form would have — except for batches at least four times
  // wider than the chip's resident waves (C4 shard), where ONE wave per problem is faster still (224-235 vs 239-245 us
  // per launch, 1.52 vs 1.60 ms per batch): no partial rows to combine, scans never cut
  bl->one_wave = bl->rows_wave && bpp == 1 && P >= 32 * (size_t)h->num_cus;
  if (bl->rows_wave && !bl->one_wave) bpp *= clc::BLOCK / 64;
  const size_t n_blocks = P * (size_t)bpp;
  if (n_blocks > h->bpartials_cap_blocks) {
    if (h->d_bpartials) CLC_HIP(hipFree(h->d_bpartials));
    h->d_bpartials = nullptr; h->bpartials_cap_blocks = 0;
    CLC_HIP(hipMalloc(&h->d_bpartials, sizeof(double) * n_blocks * clc::NACC));
    h->bpartials_cap_blocks = n_blocks;
  }
  bl->bpp = bpp;
  bl->n_blocks = n_blocks;
  bl->lm_threads = 64;
  bl->lm_blocks = (unsigned)((P + bl->lm_threads - 1) / bl->lm_threads);
  bl->compact = (h->launch_flags & clc::FLAG_COMPACT) != 0 && h->bcompact_ok;
  const bool bbeyond = h->launch_auto && h->batch_total_tiles * clc::CTILE_DOUBLES * sizeof(double) > kInfinityCacheBytes;
  bl->nt = (h->launch_flags & clc::FLAG_NONTEMPORAL) != 0 ||
           (bl->compact && h->launch_auto &&
            h->batch_total_tiles * clc::CTILE_DOUBLES * sizeof(double) > kInfinityCacheBytes + kInfinityCacheBytes / 2);
  bl->deep = (h->launch_flags & clc::FLAG_DEEP) != 0 || bbeyond;
  bl->rows_nt = bl->rows && rows_nontemporal(h, h->bn_rows, h->brows_z);
  // One workgroup per problem running the problem's WHOLE solve in one launch (batched_solve_kernel) beats the lockstep
  // launches wherever a pass over the batch is not bandwidth-bound anyway — per evaluation pass, 10^4-observation
  // problems: 17 vs 69 us at 24 problems, 28 vs 52 at 512, 49 vs 65 at 1 024 (C3), 96 vs 115 at 2 048, a tie at 4 096
  // (0.7 GB), 390 vs 370 at 8 192 (1.4 GB); 10^5-observation problems (1 500 rows each): 78 vs 54 us at 4 problems,
  // 91 vs 70 at 24, a tie at 256 (scripts/probes/c3_exp.py).  So: unless the rows exceed 1 GiB (a C4 shard: lockstep, one
  // wave per problem) or a single problem is so long (> 1 024 rows, ~6.5e4 observations) that four waves are too few.
  {
    const size_t row_bytes = (size_t)h->bn_rows * ((h->brows_z ? clc::ROW_DOUBLES_Z : clc::ROW_DOUBLES) * sizeof(double) + sizeof(clc::RowDesc));
    bl->whole_solve = bl->rows && !h->brows_z && (h->launch_flags & clc::FLAG_BATCHED_LOCKSTEP) == 0 && row_bytes <= (1ull << 30) && h->batch_max_rows <= 1024;
  }
  // Problems that fit a workgroup's registers + LDS are read from HBM once and solved on chip (clc_resident.hpp).
  bl->resident = h->bres.ok && (h->launch_flags & (clc::FLAG_NO_RESIDENT | clc::FLAG_BATCHED_LOCKSTEP)) == 0;
  {
    const size_t res_bytes = (size_t)h->bres.rows * (size_t)h->bres.lanes * 2 * sizeof(double);
    bl->res_nt = (h->launch_flags & clc::FLAG_NONTEMPORAL) != 0 || (h->launch_auto && res_bytes > kInfinityCacheBytes + kInfinityCacheBytes / 2);
  }
  return CLC_OK;
}

void launch_batched_eval(clc_handle* h, const clc_options& opt, const BatchedLaunch& bl) {
  const size_t n_blocks = bl.n_blocks;
  const int bpp = bl.bpp;
  if (bl.rows) {
#define CLC_LAUNCH_BR(LOSS, NT, BT)                                                                            \
  hipLaunchKernelGGL((clc::batched_rows_eval_kernel<LOSS, NT, BT>), dim3((unsigned)n_blocks), dim3(BT), 0, h->stream,   \
                     h->d_brxy, reinterpret_cast<const clc::RowDesc*>(h->d_brdesc), h->d_prob_row, h->d_states, bpp,     \
                     opt.loss_scale_factor, h->d_bpartials)
    if (h->brows_z) {
#define CLC_LAUNCH_BRZ(LOSS, NT, BT)                                                                           \
  hipLaunchKernelGGL((clc::batched_rows_eval_kernel<LOSS, NT, BT, true>), dim3((unsigned)n_blocks), dim3(BT), 0, h->stream, \
                     h->d_brxy, reinterpret_cast<const clc::RowDesc*>(h->d_brdesc), h->d_prob_row, h->d_states, bpp,     \
                     opt.loss_scale_factor, h->d_bpartials)
      if (bl.rows_wave) {
        if (opt.use_loss) { if (bl.rows_nt) CLC_LAUNCH_BRZ(true, true, 64); else CLC_LAUNCH_BRZ(true, false, 64); }
        else { if (bl.rows_nt) CLC_LAUNCH_BRZ(false, true, 64); else CLC_LAUNCH_BRZ(false, false, 64); }
      } else {
        if (opt.use_loss) { if (bl.rows_nt) CLC_LAUNCH_BRZ(true, true, 256); else CLC_LAUNCH_BRZ(true, false, 256); }
        else { if (bl.rows_nt) CLC_LAUNCH_BRZ(false, true, 256); else CLC_LAUNCH_BRZ(false, false, 256); }
      }
#undef CLC_LAUNCH_BRZ
      return;
    }
    if (bl.rows_wave) {
      if (opt.use_loss) { if (bl.rows_nt) CLC_LAUNCH_BR(true, true, 64); else CLC_LAUNCH_BR(true, false, 64); }
      else { if (bl.rows_nt) CLC_LAUNCH_BR(false, true, 64); else CLC_LAUNCH_BR(false, false, 64); }
    } else {
      if (opt.use_loss) { if (bl.rows_nt) CLC_LAUNCH_BR(true, true, 256); else CLC_LAUNCH_BR(true, false, 256); }
      else { if (bl.rows_nt) CLC_LAUNCH_BR(false, true, 256); else CLC_LAUNCH_BR(false, false, 256); }
    }
#undef CLC_LAUNCH_BR
    return;
  }
  const bool bcompact = bl.compact, bdeep = bl.deep, bnt = bl.nt;
#define CLC_LAUNCH_B(LOSS, CP, NT)                                                                          \
  hipLaunchKernelGGL((clc::batched_eval_kernel<LOSS, CP, NT, false>), dim3((unsigned)n_blocks), dim3(clc::BLOCK), 0, \
                     h->stream, (CP) ? h->d_bctiles : h->d_btiles, h->d_bgroups, h->d_tile_off, h->d_nobs,    \
                     h->d_states, bpp, opt.loss_scale_factor, h->d_bpartials)
#define CLC_LAUNCH_BD(LOSS, NT)                                                                             \
  hipLaunchKernelGGL((clc::batched_eval_kernel<LOSS, true, NT, true>), dim3((unsigned)n_blocks), dim3(clc::BLOCK), 0, \
                     h->stream, h->d_bctiles, h->d_bgroups, h->d_tile_off, h->d_nobs, h->d_states, bpp,           \
                     opt.loss_scale_factor, h->d_bpartials)
  if (bcompact && bdeep) {
    if (opt.use_loss) { if (bnt) CLC_LAUNCH_BD(true, true); else CLC_LAUNCH_BD(true, false); }
    else { if (bnt) CLC_LAUNCH_BD(false, true); else CLC_LAUNCH_BD(false, false); }
  } else if (bcompact) {
    if (opt.use_loss) { if (bnt) CLC_LAUNCH_B(true, true, true); else CLC_LAUNCH_B(true, true, false); }
    else { if (bnt) CLC_LAUNCH_B(false, true, true); else CLC_LAUNCH_B(false, true, false); }
  } else {
    if (opt.use_loss) { if (bnt) CLC_LAUNCH_B(true, false, true); else CLC_LAUNCH_B(true, false, false); }
    else { if (bnt) CLC_LAUNCH_B(false, false, true); else CLC_LAUNCH_B(false, false, false); }
  }
#undef CLC_LAUNCH_B
#undef CLC_LAUNCH_BD
}

}  // namespace

extern "C" {

int clc_version(void) { return CLC_VERSION; }

const char* clc_last_error(void) { return g_last_error.c_str(); }

void clc_options_default(clc_options* o) {
  if (!o) return;
  o->max_num_iterations = 100;  // src/LaseCamCalCeres.cpp:304
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->use_loss = 1;              // #define LOSSFUNCTION, :212
  o->loss_scale_factor = 0.05;  // CauchyLoss(0.05 * scale), :249
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->launch_ahead = 0;
  o->profile_events = 0;
}

int clc_create(clc_handle** out, int device) {
  if (!out) return fail(CLC_ERR_INVALID_ARG, "clc_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(CLC_ERR_NO_DEVICE, "clc_create: no HIP device available (this library has no CPU fallback)", e);
  if (device < 0 || device >= count) return fail(CLC_ERR_INVALID_ARG, "clc_create: bad device index");
  CLC_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  CLC_HIP(hipGetDeviceProperties(&prop, device));
  clc_handle* h = new clc_handle();
  h->device = device;
  h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = std::getenv("CLC_AUTO_PATHS_DISABLE")) {
    const int m = std::atoi(e);
    if (m >= 0 && m <= 7) h->auto_disable = m;
  }
  CLC_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  CLC_HIP(hipMalloc(&h->d_block, sizeof(clc::SolveBlock)));
  h->d_state = &h->d_block->st[0];
  h->d_state_b = &h->d_block->st[1];
  CLC_HIP(hipMalloc(&h->d_small, sizeof(double) * kSmallDoubles));
  CLC_HIP(hipHostMalloc(&h->h_small, sizeof(double) * kSmallDoubles, hipHostMallocDefault));
  CLC_HIP(hipMalloc(&h->d_queue, sizeof(unsigned int)));
  CLC_HIP(hipMalloc(&h->d_ticket, sizeof(unsigned int)));
  CLC_HIP(hipMemsetAsync(h->d_ticket, 0, sizeof(unsigned int), h->own_stream));
  CLC_HIP(hipMemsetAsync(h->d_queue, 0, sizeof(unsigned int), h->own_stream));  // (counters are zero between launches)
  CLC_HIP(hipStreamSynchronize(h->own_stream));  // the caller may switch streams (clc_set_stream) before the first launch
  CLC_HIP(hipHostMalloc(&h->h_mailbox, sizeof(clc::HostMailbox), hipHostMallocCoherent | hipHostMallocMapped));
  CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_mailbox), h->h_mailbox, 0));
  std::memset(h->h_mailbox, 0, sizeof(clc::HostMailbox));
  // start pose / outcome of the single-workgroup resident solve (one allocation: 8 doubles of pose, then the summary)
  CLC_HIP(hipHostMalloc(&h->h_spose, 8 * sizeof(double) + sizeof(clc_summary), hipHostMallocCoherent | hipHostMallocMapped));
  CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_spose), h->h_spose, 0));
  h->h_ssummary = reinterpret_cast<clc_summary*>(h->h_spose + 8);
  h->d_ssummary = reinterpret_cast<clc_summary*>(h->d_spose + 8);
  *out = h;
  return CLC_OK;
}

void clc_destroy(clc_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  void* ptrs[] = {h->d_tiles, h->d_partials, h->d_block, h->d_trace, h->d_small, h->d_btiles,
                  h->d_tile_off, h->d_nobs, h->d_queue, h->d_states,
                  h->d_bpartials, h->d_ticket, h->d_ctiles, h->d_groups, h->d_bctiles, h->d_bgroups, h->d_results,
                  h->d_rxy, h->d_rdesc, h->d_brxy, h->d_brdesc, h->bres.d_xy, h->bres.d_desc, h->bres.d_row, h->sres.d_xy, h->sres.d_desc, h->sres.d_row, h->cres.d_xy, h->cres.d_desc, h->cres.d_row, h->d_board, h->d_prob_row, h->d_sq, h->d_st, h->d_spts, h->d_sptl, h->d_soff};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->pool.clear();
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->h_mailbox) (void)hipHostFree(h->h_mailbox);
  if (h->h_spose) (void)hipHostFree(h->h_spose);
  if (h->h_poses) (void)hipHostFree(h->h_poses);
  if (h->h_summaries) (void)hipHostFree(h->h_summaries);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

int clc_set_stream(clc_handle* h, void* hip_stream) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_set_stream: NULL handle");
  hipStream_t next = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
  if (next == h->stream) return CLC_OK;
  // A finished solve may still have up to launch_ahead + 1 no-op launches queued on the old stream; they forward the
  // terminated LM state into the buffers the next solve initialises.  Drain them before anything runs elsewhere.
  CLC_HIP(hipSetDevice(h->device));
  CLC_HIP(hipStreamSynchronize(h->stream));
  h->stream = next;
  return CLC_OK;
}

int clc_set_auto_paths(clc_handle* h, int disable_mask) {
  if (!h || disable_mask < 0 || disable_mask > 7) return fail(CLC_ERR_INVALID_ARG, "clc_set_auto_paths: bad argument");
  h->auto_disable = disable_mask;
  return CLC_OK;
}

int clc_set_launch(clc_handle* h, int grid_blocks, int flags) {
  if (!h || grid_blocks < 0 || flags < -1 || flags > 16383)
    return fail(CLC_ERR_INVALID_ARG, "clc_set_launch: bad argument");
  h->grid_override = grid_blocks;
  h->launch_flags = flags < 0 ? kDefaultLaunchFlags : flags;
  h->launch_auto = flags < 0;
  return CLC_OK;
}

int clc_device_info(clc_handle* h, char* name, int name_cap, int* num_cus) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_device_info: NULL handle");
  hipDeviceProp_t prop;
  CLC_HIP(hipGetDeviceProperties(&prop, h->device));
  if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (num_cus) *num_cus = h->num_cus;
  return CLC_OK;
}

int clc_flatten_observations(int n_poses, const double* tag_q_wxyz, const double* tag_t,
                             const int64_t* pts_off, const double* pts, const int64_t* ptl_off,
                             const double* ptl, int use_linefitting_data,
                             int use_boundary_constraint, clc_observation* records,
                             int64_t* n_records) {
  if (n_poses < 0 || !n_records || (n_poses > 0 && (!tag_q_wxyz || !tag_t || !pts_off || !ptl_off)))
    return fail(CLC_ERR_INVALID_ARG, "clc_flatten_observations: bad argument");
  int rc = clc::host::flatten(n_poses, tag_q_wxyz, tag_t, pts_off, pts, ptl_off, ptl,
                              use_linefitting_data != 0, use_boundary_constraint != 0, records, n_records);
  if (rc == CLC_ERR_EMPTY_SCAN)
    return fail(rc, "clc_flatten_observations: boundary constraint on an empty scan (reference: std::out_of_range at LaseCamCalCeres.cpp:278)");
  return rc;
}

int clc_upload_device(clc_handle* h, const clc_observation* records_dev, size_t n) {
  if (!h || (n > 0 && !records_dev)) return fail(CLC_ERR_INVALID_ARG, "clc_upload_device: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  h->compact_ok = false;
  h->rows_ok = false;
  h->sres.ok = false;
  h->cres.ok = false;
  h->split_grid = -1;
  int rc = retile_into(h, reinterpret_cast<const double*>(records_dev), n, &h->d_tiles, &h->tiles_cap_bytes);
  if (rc != CLC_OK) return rc;
  CLC_HIP(hipStreamSynchronize(h->stream));
  h->n_obs = n;
  const std::vector<long long> rec_off = {0, (long long)n};
  const std::vector<long long> tile_off = {0, (long long)((n + clc::TILE - 1) / clc::TILE)};
  const LayoutTargets T = {&h->d_ctiles, &h->ctiles_cap_bytes, &h->d_groups, &h->groups_cap_bytes, &h->n_groups, &h->compact_ok,
                           &h->d_rxy, &h->rxy_cap_bytes, &h->d_rdesc, &h->rdesc_cap_bytes, &h->n_rows, &h->rows_ok, nullptr,
                           // a problem one workgroup can hold (<= 512 lanes x 22 points) also gets the lane layout: clc_solve then runs
                           // its whole LM loop in ONE single-workgroup launch from registers + LDS (solve_resident_single)
                           n <= (size_t)512 * (kResPR512 + kResPL512) ? &h->sres : nullptr, &h->rows_z,
                           // what one workgroup cannot hold (more points; or more than 512 scans, or scans whose lengths leave
                           // too many half-filled lanes: a lane holds points of ONE scan), up to 65 536 lanes x 40 points, is
                           // dealt to 256 workgroups: the cooperative solve (clc_coop.hpp)
                           n <= (size_t)clc::COOP_WGS * clc::COOP_NL * (clc::COOP_PR + clc::COOP_PL) ? &h->cres : nullptr};
  return build_layouts(h, reinterpret_cast<const double*>(records_dev), n, rec_off, tile_off, T);
}

int clc_upload(clc_handle* h, const clc_observation* records, size_t n) {
  if (!h || (n > 0 && !records)) return fail(CLC_ERR_INVALID_ARG, "clc_upload: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> aos(&h->pool);
  if (n > 0) {
    CLC_HIP(aos.alloc(n * 8));
    CLC_HIP(hipMemcpy(aos.p, records, n * sizeof(clc_observation), hipMemcpyHostToDevice));
  }
  return clc_upload_device(h, reinterpret_cast<const clc_observation*>(aos.p), n);
}

// ---- resident scans + device-side problem assembly ---------------------------------------------------------------
namespace {

int grow(double** p, size_t* cap, size_t bytes) { return ensure_bytes(p, cap, std::max<size_t>(bytes, 8)); }

// per-pose record offsets of a selection (host, O(poses)); CLC_ERR_EMPTY_SCAN mirrors the reference's .at(0) throw
int selection_offsets(const clc_handle* h, bool linefit, bool boundary, std::vector<long long>* rec_off) {
  const int P = h->store_poses;
  rec_off->assign((size_t)P + 1, 0);
  const std::vector<long long>& off = linefit ? h->s_ptl_off : h->s_pts_off;
  for (int i = 0; i < P; ++i) {
    long long c = off[(size_t)i + 1] - off[(size_t)i];
    if (boundary && linefit) {
      if (h->s_pts_off[(size_t)i + 1] - h->s_pts_off[(size_t)i] <= 0) return CLC_ERR_EMPTY_SCAN;
      c += 2;
    }
    (*rec_off)[(size_t)i + 1] = (*rec_off)[(size_t)i] + c;
  }
  return CLC_OK;
}

// builds the records of the selection on the device into *aos (allocated here)
int flatten_on_device(clc_handle* h, bool linefit, bool boundary, DevBuf<double>* aos, long long* n_out) {
  if (h->store_poses < 0) return fail(CLC_ERR_NO_DATA, "clc_select_observations: no scans stored (clc_store_observations)");
  std::vector<long long> rec_off;
  const int rc = selection_offsets(h, linefit, boundary, &rec_off);
  if (rc == CLC_ERR_EMPTY_SCAN)
    return fail(rc, "clc_select_observations: boundary constraint on an empty scan (reference: std::out_of_range at LaseCamCalCeres.cpp:278)");
  const int P = h->store_poses;
  const long long N = rec_off[(size_t)P];
  *n_out = N;
  CLC_HIP(aos->alloc((size_t)std::max<long long>(N, 1) * 8));
  if (N == 0 || P == 0) return CLC_OK;
  long long* d_off = reinterpret_cast<long long*>(h->d_soff);
  CLC_HIP(hipMemcpyAsync(d_off + 2 * ((size_t)P + 1), rec_off.data(), sizeof(long long) * ((size_t)P + 1), hipMemcpyHostToDevice,
                         h->stream));
  hipLaunchKernelGGL(clc::flatten_kernel, dim3((unsigned)P), dim3(clc::BLOCK), 0, h->stream, P, h->d_sq, h->d_st, d_off,
                     h->d_spts, d_off + ((size_t)P + 1), h->d_sptl, linefit ? 1 : 0, boundary ? 1 : 0,
                     d_off + 2 * ((size_t)P + 1), aos->p);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));  // rec_off (host vector) must outlive the copy
  return CLC_OK;
}

}  // namespace

void* clc_pinned_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 8), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void clc_pinned_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int clc_store_observations(clc_handle* h, int n_poses, const double* tag_q_wxyz, const double* tag_t,
                           const int64_t* pts_off, const double* pts, const int64_t* ptl_off, const double* ptl) {
  if (!h || n_poses < 0 || (n_poses > 0 && (!tag_q_wxyz || !tag_t || !pts_off || !ptl_off)))
    return fail(CLC_ERR_INVALID_ARG, "clc_store_observations: bad argument");
  const size_t P = (size_t)n_poses;
  h->store_poses = -1;
  h->s_pts_off.assign(P + 1, 0);
  h->s_ptl_off.assign(P + 1, 0);
  for (size_t i = 0; i <= P && P > 0; ++i) {
    h->s_pts_off[i] = pts_off[i] - pts_off[0];
    h->s_ptl_off[i] = ptl_off[i] - ptl_off[0];
    if (i > 0 && (h->s_pts_off[i] < h->s_pts_off[i - 1] || h->s_ptl_off[i] < h->s_ptl_off[i - 1]))
      return fail(CLC_ERR_INVALID_ARG, "clc_store_observations: offsets not monotone");
  }
  const size_t M = (size_t)h->s_pts_off[P], ML = (size_t)h->s_ptl_off[P];
  if ((M > 0 && !pts) || (ML > 0 && !ptl)) return fail(CLC_ERR_INVALID_ARG, "clc_store_observations: NULL points");
  CLC_HIP(hipSetDevice(h->device));
  int rc = grow(&h->d_sq, &h->sq_cap, P * 4 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_st, &h->st_cap, P * 3 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_spts, &h->spts_cap, M * 3 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_sptl, &h->sptl_cap, ML * 3 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_soff, &h->soff_cap, 3 * (P + 1) * sizeof(long long));
  if (rc != CLC_OK) return rc;
  if (P > 0) {
    CLC_HIP(hipMemcpyAsync(h->d_sq, tag_q_wxyz, P * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CLC_HIP(hipMemcpyAsync(h->d_st, tag_t, P * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    long long* d_off = reinterpret_cast<long long*>(h->d_soff);
    CLC_HIP(hipMemcpyAsync(d_off, h->s_pts_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    CLC_HIP(hipMemcpyAsync(d_off + (P + 1), h->s_ptl_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  }
  if (M > 0) CLC_HIP(hipMemcpyAsync(h->d_spts, pts + 3 * pts_off[0], M * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (ML > 0) CLC_HIP(hipMemcpyAsync(h->d_sptl, ptl + 3 * ptl_off[0], ML * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));  // the caller's arrays may go away
  h->store_poses = n_poses;
  h->store_generation++;
  return CLC_OK;
}

int64_t clc_store_generation(const clc_handle* h) { return h ? h->store_generation : -1; }

int clc_select_observations(clc_handle* h, int use_linefitting_data, int use_boundary_constraint, int64_t* n_records) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_select_observations: NULL handle");
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> aos(&h->pool);
  long long N = 0;
  int rc = flatten_on_device(h, use_linefitting_data != 0, use_boundary_constraint != 0, &aos, &N);
  if (rc != CLC_OK) return rc;
  if (n_records) *n_records = (int64_t)N;
  return clc_upload_device(h, reinterpret_cast<const clc_observation*>(aos.p), (size_t)N);
}

// test hook: the device-built records of a selection, copied back (records_out[N*8], N from clc_select_observations)
int clc_debug_flatten_device(clc_handle* h, int use_linefitting_data, int use_boundary_constraint, double* records_out,
                             int64_t cap_records, int64_t* n_records) {
  if (!h || !n_records) return fail(CLC_ERR_INVALID_ARG, "clc_debug_flatten_device: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> aos(&h->pool);
  long long N = 0;
  int rc = flatten_on_device(h, use_linefitting_data != 0, use_boundary_constraint != 0, &aos, &N);
  if (rc != CLC_OK) return rc;
  *n_records = N;
  if (records_out && N > 0) {
    if (cap_records < N) return fail(CLC_ERR_INVALID_ARG, "clc_debug_flatten_device: buffer too small");
    CLC_HIP(hipMemcpy(records_out, aos.p, (size_t)N * 8 * sizeof(double), hipMemcpyDeviceToHost));
  }
  return CLC_OK;
}

size_t clc_num_observations(const clc_handle* h) { return h ? h->n_obs : 0; }

int clc_factor_evaluate(clc_handle* h, const double pose[7], double* residuals, double* jacobians) {
  if (!h || !pose || !residuals) return fail(CLC_ERR_INVALID_ARG, "clc_factor_evaluate: bad argument");
  if (!h->d_tiles) return fail(CLC_ERR_NO_DATA, "clc_factor_evaluate: no observations uploaded");
  CLC_HIP(hipSetDevice(h->device));
  const size_t n = h->n_obs;
  if (n == 0) return CLC_OK;
  DevBuf<double> br(&h->pool), bj(&h->pool);
  CLC_HIP(br.alloc(n));
  if (jacobians) CLC_HIP(bj.alloc(n * 7));
  double *d_r = br.p, *d_j = bj.p;
  std::memcpy(h->h_small, pose, 7 * sizeof(double));
  CLC_HIP(hipMemcpyAsync(h->d_small, h->h_small, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int threads = 256;
  hipLaunchKernelGGL(clc::factor_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, h->d_tiles, (long long)n, h->d_small, d_r, d_j);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(residuals, d_r, n * sizeof(double), hipMemcpyDeviceToHost));
  if (jacobians) CLC_HIP(hipMemcpy(jacobians, d_j, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
  return CLC_OK;
}

int clc_pose_plus(clc_handle* h, const double* x, const double* delta, double* out, size_t n) {
  if (!h || (n > 0 && (!x || !delta || !out))) return fail(CLC_ERR_INVALID_ARG, "clc_pose_plus: bad argument");
  if (n == 0) return CLC_OK;
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> buf(&h->pool);
  CLC_HIP(buf.alloc(n * 20));
  double *d_x = buf.p, *d_d = buf.p + 7 * n, *d_o = buf.p + 13 * n;
  CLC_HIP(hipMemcpy(d_x, x, n * 7 * sizeof(double), hipMemcpyHostToDevice));
  CLC_HIP(hipMemcpy(d_d, delta, n * 6 * sizeof(double), hipMemcpyHostToDevice));
  const int threads = 256;
  hipLaunchKernelGGL(clc::plus_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, d_x, d_d, d_o, (long long)n);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(out, d_o, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
  return CLC_OK;
}

int clc_pose_plus_jacobian(const double* /*x*/, double jacobian[42]) {
  if (!jacobian) return fail(CLC_ERR_INVALID_ARG, "clc_pose_plus_jacobian: NULL output");
  for (int i = 0; i < 42; ++i) jacobian[i] = 0.0;
  for (int i = 0; i < 6; ++i) jacobian[6 * i + i] = 1.0;  // [I6; 0], pose_local_parameterization.cpp:36-37
  return CLC_OK;
}

int clc_eval(clc_handle* h, const double pose[7], int with_loss, double loss_scale_factor,
             double* cost, double g[6], double H[21]) {
  if (!h || !pose || !cost) return fail(CLC_ERR_INVALID_ARG, "clc_eval: bad argument");
  if (!h->d_tiles) return fail(CLC_ERR_NO_DATA, "clc_eval: no observations uploaded");
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_eval: non-finite pose");
  if (with_loss && !(loss_scale_factor > 0.0)) return fail(CLC_ERR_INVALID_ARG, "clc_eval: loss_scale_factor must be > 0");
  CLC_HIP(hipSetDevice(h->device));
  const int grid = eval_grid(h, h->n_obs);
  int rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  std::memcpy(h->h_small, pose, 7 * sizeof(double));
  CLC_HIP(hipMemcpyAsync(h->d_small, h->h_small, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const bool want_jac = (g != nullptr) || (H != nullptr);
  if (want_jac)
    launch_eval<true>(h, grid, with_loss != 0, h->d_small, nullptr, loss_scale_factor);
  else
    launch_eval<false>(h, grid, with_loss != 0, h->d_small, nullptr, loss_scale_factor);
  CLC_HIP(hipGetLastError());
  hipLaunchKernelGGL(clc::reduce_kernel, dim3(1), dim3(clc::BLOCK), 0, h->stream, h->d_partials, grid,
                     with_loss, loss_scale_factor, h->d_small + 16);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipMemcpyAsync(h->h_small + 16, h->d_small + 16, clc::NACC * sizeof(double), hipMemcpyDeviceToHost,
                         h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  const double* r = h->h_small + 16;
  *cost = r[27];
  if (g) for (int i = 0; i < 6; ++i) g[i] = want_jac ? r[21 + i] : 0.0;
  if (H) for (int i = 0; i < 21; ++i) H[i] = r[i];
  return CLC_OK;
}

namespace {

// clc_solve as a chain of step_kernel launches (clc_kernels.hpp "Step kernel"): launch 0 evaluates at the initial
// pose, launch k >= 1 consumes the rows of launch k-1 in every workgroup and evaluates at the next point.  The
// host only keeps `lookahead` launches queued beyond the last pass the device reported consumed.
// win_first/win_last/win_ms (profiling hook clc_time_steps): HIP events are recorded on the stream right before launch
// `win_first` and right after launch `win_last`; *win_ms receives the elapsed time between them.
int solve_stepped(clc_handle* h, const clc_options& opt, int grid, double pose[7], clc_summary* summary,
                  clc_iteration* trace, int trace_cap, std::chrono::steady_clock::time_point t0,
                  int win_first = -1, int win_last = -1, float* win_ms = nullptr) {
  const bool want_trace = trace != nullptr && trace_cap > 0;
  if (want_trace) {
    const int rc = ensure_trace(h, opt.max_num_iterations + 8);
    if (rc != CLC_OK) return rc;
  }
  const int lookahead = opt.launch_ahead > 0 ? opt.launch_ahead : default_lookahead();
  const int max_launches = opt.max_num_iterations + 2;  // (max_iterations + 1) evaluations + the final controller pass
  clc::HostMailbox* mb = h->h_mailbox;
  mb->n_done = 0;
  mb->status = CLC_RUNNING;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  clc::Pose7 p0;
  for (int i = 0; i < 7; ++i) p0.v[i] = pose[i];
  clc::SolveParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.opt = opt;
  prm.pose0 = p0;
  prm.trace = want_trace ? h->d_trace : nullptr;
  prm.mailbox = h->d_mailbox;
  prm.trace_cap = want_trace ? h->trace_cap : 0;
  const bool deep = (h->launch_flags & clc::FLAG_DEEP) != 0 ||
                    (h->launch_auto && (size_t)h->n_obs * 28 > kInfinityCacheBytes);
  const bool rows = use_rows(h);
  const bool rows_z = rows && h->rows_z;
  const bool rows_nt = rows && rows_nontemporal(h, h->n_rows, rows_z);
  const bool rows_eq = (h->launch_flags & clc::FLAG_EQUAL_WAVES) != 0 && !rows_z;
  if (rows && rows_eq) ensure_wave_split(h, grid);
  double* rows_buf[2] = {h->d_partials, h->d_partials_b};
  int launched = 0, status = CLC_RUNNING, last_done = 0;
  long long spins = 0;
  auto t_last_progress = std::chrono::steady_clock::now();
  for (;;) {
    status = __atomic_load_n(&mb->status, __ATOMIC_ACQUIRE);
    if (status != CLC_RUNNING) break;
    // passes consumed = launches whose rows are used up.  Clamped to what this solve has launched: the early progress
    // store of the PREVIOUS solve's last launches is relaxed and may land after the reset above.
    const int done = std::min(__atomic_load_n(&mb->n_done, __ATOMIC_ACQUIRE), launched);
    if (launched < max_launches && launched - done <= lookahead) {
      const int k = launched;
      if (win_ms && k == win_first) CLC_HIP(hipEventRecord(h->ev[0], h->stream));
      // launch k reads state[(k-1)&1] / rows[(k-1)&1] and writes state[k&1] / rows[k&1]
      const double* r_in = rows_buf[(k + 1) & 1];
      double* r_out = rows_buf[k & 1];
#define CLC_LAUNCH_STEP(LOSS, DEEP, MODE)                                                                     \
  hipLaunchKernelGGL((clc::step_kernel<LOSS, DEEP, MODE>), dim3(grid), dim3(512), 0, h->stream, r_in,            \
                     h->d_ctiles, h->d_groups, (int)h->n_obs, grid | ((k & 1) << 30), k, r_out, h->d_block, prm)
#define CLC_LAUNCH_STEP_R(LOSS, NT, MODE, WG)                                                                 \
  hipLaunchKernelGGL((clc::step_kernel<LOSS, NT, MODE, 1, WG>), dim3(grid), dim3(512), 0, h->stream, r_in,       \
                     h->d_rxy, h->d_rdesc, (int)h->n_rows, grid | ((k & 1) << 30), k, r_out, h->d_block, prm)
#define CLC_LAUNCH_STEP_M(LOSS, DEEP)                                                                         \
  do { if (k == 0) CLC_LAUNCH_STEP(LOSS, DEEP, 0); else if (k == 1) CLC_LAUNCH_STEP(LOSS, DEEP, 1);             \
       else CLC_LAUNCH_STEP(LOSS, DEEP, 2); } while (0)
#define CLC_LAUNCH_STEP_RM(LOSS, NT, WG)                                                                      \
  do { if (k == 0) CLC_LAUNCH_STEP_R(LOSS, NT, 0, WG); else if (k == 1) CLC_LAUNCH_STEP_R(LOSS, NT, 1, WG);     \
       else CLC_LAUNCH_STEP_R(LOSS, NT, 2, WG); } while (0)
      if (rows_z) {  // rows that carry z (LAYOUT 2): 3:2 wave shares
#define CLC_LAUNCH_STEP_Z(LOSS, NT)                                                                           \
  do { if (k == 0) hipLaunchKernelGGL((clc::step_kernel<LOSS, NT, 0, 2, true>), dim3(grid), dim3(512), 0, h->stream, r_in, h->d_rxy, h->d_rdesc, (int)h->n_rows, grid | ((k & 1) << 30), k, r_out, h->d_block, prm); \
       else if (k == 1) hipLaunchKernelGGL((clc::step_kernel<LOSS, NT, 1, 2, true>), dim3(grid), dim3(512), 0, h->stream, r_in, h->d_rxy, h->d_rdesc, (int)h->n_rows, grid | ((k & 1) << 30), k, r_out, h->d_block, prm); \
       else hipLaunchKernelGGL((clc::step_kernel<LOSS, NT, 2, 2, true>), dim3(grid), dim3(512), 0, h->stream, r_in, h->d_rxy, h->d_rdesc, (int)h->n_rows, grid | ((k & 1) << 30), k, r_out, h->d_block, prm); } while (0)
        if (opt.use_loss) { if (rows_nt) CLC_LAUNCH_STEP_Z(true, true); else CLC_LAUNCH_STEP_Z(true, false); }
        else { if (rows_nt) CLC_LAUNCH_STEP_Z(false, true); else CLC_LAUNCH_STEP_Z(false, false); }
#undef CLC_LAUNCH_STEP_Z
      }
      else if (rows) {
        if (opt.use_loss) {
          if (rows_eq) { if (rows_nt) CLC_LAUNCH_STEP_RM(true, true, false); else CLC_LAUNCH_STEP_RM(true, false, false); }
          else { if (rows_nt) CLC_LAUNCH_STEP_RM(true, true, true); else CLC_LAUNCH_STEP_RM(true, false, true); }
        } else {
          if (rows_eq) { if (rows_nt) CLC_LAUNCH_STEP_RM(false, true, false); else CLC_LAUNCH_STEP_RM(false, false, false); }
          else { if (rows_nt) CLC_LAUNCH_STEP_RM(false, true, true); else CLC_LAUNCH_STEP_RM(false, false, true); }
        }
      }
      else if (opt.use_loss) { if (deep) CLC_LAUNCH_STEP_M(true, true); else CLC_LAUNCH_STEP_M(true, false); }
      else { if (deep) CLC_LAUNCH_STEP_M(false, true); else CLC_LAUNCH_STEP_M(false, false); }
#undef CLC_LAUNCH_STEP_M
#undef CLC_LAUNCH_STEP_RM
#undef CLC_LAUNCH_STEP_R
#undef CLC_LAUNCH_STEP
      if (win_ms && k == win_last) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
      ++launched;
      continue;
    }
    if (done != last_done) { last_done = done; t_last_progress = std::chrono::steady_clock::now(); spins = 0; }
    if ((++spins & 0xFFFF) == 0) {
      hipError_t e = hipStreamQuery(h->stream);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(CLC_ERR_HIP, "clc_solve: stream error", e);
      if (e == hipSuccess) {
        status = __atomic_load_n(&mb->status, __ATOMIC_ACQUIRE);
        if (status != CLC_RUNNING) break;
        if (launched >= max_launches) return fail(CLC_ERR_HIP, "clc_solve: controller did not terminate");
      }
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last_progress).count();
      if (waited > 30.0) return fail(CLC_ERR_HIP, "clc_solve: no progress from the device for 30 s");
    }
  }
  CLC_HIP(hipGetLastError());
  std::atomic_thread_fence(std::memory_order_acquire);
  *summary = mb->summary;
  for (int i = 0; i < 7; ++i) pose[i] = mb->pose[i];
  summary->eval_kernel_ms = 0.0;
  summary->eval_kernel_launches = 0;
  if (want_trace) {
    CLC_HIP(hipStreamSynchronize(h->stream));
    const int n = std::min(std::min(summary->num_iterations + 1, trace_cap), h->trace_cap);
    if (n > 0) CLC_HIP(hipMemcpy(trace, h->d_trace, sizeof(clc_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  summary->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (win_ms) {
    *win_ms = -1.f;
    if (launched > win_last && win_first >= 0) {
      CLC_HIP(hipStreamSynchronize(h->stream));
      CLC_HIP(hipEventElapsedTime(win_ms, h->ev[0], h->ev[1]));
    }
  }
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve: non-finite result");
  return CLC_OK;
}

// A problem that fits ONE workgroup (<= 512 lanes x 22 points; the lane layout was built at upload): the whole LM solve in a
// single launch of resident_solve_kernel<8 waves> — points read from HBM once into registers + LDS, every pass, reduction and
// controller step on chip, no kernel boundary and no partial rows between LM iterations.  This is the reference's own problem
// size (main/calibr_simulation.cpp: 50 poses x ~114 points; main/calibr_offline.cpp: O(10^2) poses): 4.3 us per LM
// iteration instead of the 7.1 us of the 256-workgroup step chain, which at this size is all launch boundary, row
// exchange and controller.  One CU works, 255 idle — the problem has 5.7e3 points.
int solve_resident_single(clc_handle* h, const clc_options& opt, double pose[7], clc_summary* summary, clc_iteration* trace,
                          int trace_cap, std::chrono::steady_clock::time_point t0) {
  const bool want_trace = trace != nullptr && trace_cap > 0;
  if (want_trace) {
    const int rc = ensure_trace(h, opt.max_num_iterations + 8);
    if (rc != CLC_OK) return rc;
  }
  for (int i = 0; i < 7; ++i) h->h_spose[i] = pose[i];
  int32_t* h_done = reinterpret_cast<int32_t*>(h->h_spose + 7);  // completion flag behind the pose (same pinned allocation)
  int32_t* d_done = reinterpret_cast<int32_t*>(h->d_spose + 7);
  __atomic_store_n(h_done, 0, __ATOMIC_RELAXED);
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const unsigned int* d_row = reinterpret_cast<const unsigned int*>(h->sres.d_row);
  const clc::ResLane* d_desc = reinterpret_cast<const clc::ResLane*>(h->sres.d_desc);
  clc_iteration* d_trace = want_trace ? h->d_trace : nullptr;
  const int d_cap = want_trace ? h->trace_cap : 0;
#define CLC_LAUNCH_SINGLE(LOSS, CTRL)                                                                                                   \
  hipLaunchKernelGGL((clc::resident_solve_kernel<LOSS, false, 8, kResPR512, kResPL512, CTRL>), dim3(1), dim3(512), 0, h->stream, h->sres.d_xy, \
                     d_row, d_desc, h->d_groups, h->sres.uni_ppl, opt, d_trace, d_cap, h->d_spose, h->d_ssummary, h->d_small, d_done, nullptr)
  const bool uni_ctrl = (h->auto_disable & 4) != 0;  // the cooperative kernel's controller here: the bit-identity test of the two
  if (opt.use_loss) { if (uni_ctrl) CLC_LAUNCH_SINGLE(true, 1); else CLC_LAUNCH_SINGLE(true, 0); }
  else { if (uni_ctrl) CLC_LAUNCH_SINGLE(false, 1); else CLC_LAUNCH_SINGLE(false, 0); }
#undef CLC_LAUNCH_SINGLE
  CLC_HIP(hipGetLastError());
  // The kernel sets the flag (system-scope release) after the outcome is written: polling it avoids the wake-up latency of a
  // blocking stream synchronisation (~15 us of a ~120 us solve).  Bounded: a wedged queue falls through to the synchronisation,
  // which reports the error.
  {
    long long spins = 0;
    const auto t_spin = std::chrono::steady_clock::now();
    while (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == 0) {
      if ((++spins & 0xFFFF) == 0) {
        if (hipStreamQuery(h->stream) != hipErrorNotReady) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 30.0) break;
      }
    }
    if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == 0 || want_trace) CLC_HIP(hipStreamSynchronize(h->stream));
  }
  *summary = *h->h_ssummary;
  for (int i = 0; i < 7; ++i) pose[i] = h->h_spose[i];
  if (want_trace) {
    const int n = std::min(std::min(summary->num_iterations + 1, trace_cap), h->trace_cap);
    if (n > 0) CLC_HIP(hipMemcpy(trace, h->d_trace, sizeof(clc_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  summary->eval_kernel_ms = 0.0;
  summary->eval_kernel_launches = 0;
  summary->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve: non-finite result");
  return CLC_OK;
}

// clc_solve as ONE launch of 256 co-resident workgroups that keep the problem on chip (clc_coop.hpp).  Returns kCoopFallback when
// the path cannot be used (device too small, or the launch timed out in its exchange): the caller runs the step chain instead.
constexpr int kCoopFallback = -1000;
int solve_coop(clc_handle* h, const clc_options& opt, double pose[7], clc_summary* summary, clc_iteration* trace, int trace_cap,
               std::chrono::steady_clock::time_point t0) {
  if (h->coop_checked == 0) {
    int a = 0, b = 0;
    const hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, clc::coop_solve_kernel<true, false>, clc::COOP_THREADS, 0);
    const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, clc::coop_solve_kernel<false, false>, clc::COOP_THREADS, 0);
    h->coop_checked = (e1 == hipSuccess && e2 == hipSuccess && a >= 1 && b >= 1 && h->num_cus >= clc::COOP_WGS) ? 1 : -1;
    (void)hipGetLastError();
  }
  if (h->coop_checked < 0) return kCoopFallback;
  if (!h->d_board) {
    CLC_HIP(hipMalloc(&h->d_board, sizeof(clc::CoopBoard)));
    // (ordered on the handle's stream AND waited for: the caller may switch streams, clc_set_stream, before the next solve)
    CLC_HIP(hipMemsetAsync(h->d_board, 0, sizeof(clc::CoopBoard), h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    h->coop_tag = 1;
    {  // (tuning hook: first-poll offsets, clc_coop.hpp)
      unsigned long long d[2] = {0, 0};
      if (const char* e = std::getenv("CLC_COOP_D1")) d[0] = (unsigned long long)std::atoll(e);
      if (const char* e = std::getenv("CLC_COOP_D2")) d[1] = (unsigned long long)std::atoll(e);
      if (d[0] || d[1]) CLC_HIP(hipMemcpy(&h->d_board->ctl[1], d, sizeof(d), hipMemcpyHostToDevice));
    }
  }
  const unsigned int passes = (unsigned int)opt.max_num_iterations + 4u;
  if (h->coop_tag > 0xFFFFFFFFu - passes - 8u) {  // the 32-bit pass tags are used up: start over on clean boards
    CLC_HIP(hipMemsetAsync(h->d_board, 0, sizeof(clc::CoopBoard), h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    h->coop_tag = 1;
  }
  const bool want_trace = trace != nullptr && trace_cap > 0;
  if (want_trace) {
    const int rc = ensure_trace(h, opt.max_num_iterations + 8);
    if (rc != CLC_OK) return rc;
  }
  const bool timed = opt.profile_events == 2;  // HIP event pair around the one launch -> clc_summary.eval_kernel_ms
  if (timed) {
    const int rc = ensure_events(h, 2);
    if (rc != CLC_OK) return rc;
  }
  int32_t* h_done = reinterpret_cast<int32_t*>(h->h_spose + 7);  // completion flag behind the pose (same pinned allocation)
  int32_t* d_done = reinterpret_cast<int32_t*>(h->d_spose + 7);
  __atomic_store_n(h_done, 0, __ATOMIC_RELAXED);
  std::atomic_thread_fence(std::memory_order_seq_cst);
  clc::Pose7 p0;
  for (int i = 0; i < 7; ++i) p0.v[i] = pose[i];
  const unsigned int* d_row = reinterpret_cast<const unsigned int*>(h->cres.d_row);
  const clc::ResLane* d_desc = reinterpret_cast<const clc::ResLane*>(h->cres.d_desc);
  clc_iteration* d_trace = want_trace ? h->d_trace : nullptr;
  const int d_cap = want_trace ? h->trace_cap : 0;
  const unsigned int tag0 = h->coop_tag;
  h->coop_tag += passes;
  const unsigned int wgs = (unsigned int)(clc::COOP_WGS - h->coop_test_drop);
  h->coop_test_drop = 0;
  if (timed) CLC_HIP(hipEventRecord(h->ev[0], h->stream));
  if (opt.use_loss)
    hipLaunchKernelGGL((clc::coop_solve_kernel<true, false>), dim3(wgs), dim3(clc::COOP_THREADS), 0, h->stream, h->cres.d_xy, d_row, d_desc,
                       h->d_groups, h->cres.uni_ppl, opt, p0, d_trace, d_cap, h->d_board, tag0, h->d_spose, h->d_ssummary, h->d_small, d_done);
  else
    hipLaunchKernelGGL((clc::coop_solve_kernel<false, false>), dim3(wgs), dim3(clc::COOP_THREADS), 0, h->stream, h->cres.d_xy, d_row, d_desc,
                       h->d_groups, h->cres.uni_ppl, opt, p0, d_trace, d_cap, h->d_board, tag0, h->d_spose, h->d_ssummary, h->d_small, d_done);
  CLC_HIP(hipGetLastError());
  if (timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
  {  // the kernel raises the flag (system-scope release) after the outcome is written; bounded like solve_resident_single
    long long spins = 0;
    const auto t_spin = std::chrono::steady_clock::now();
    while (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == 0) {
      if ((++spins & 0xFFFF) == 0) {
        if (hipStreamQuery(h->stream) != hipErrorNotReady) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spin).count() > 30.0) break;
      }
    }
    if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) != clc::COOP_DONE_OK || want_trace || timed) CLC_HIP(hipStreamSynchronize(h->stream));
  }
  if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) != clc::COOP_DONE_OK) {
    // an exchange timed out (a workgroup was not resident in time): nothing was written; the path rests (see coop_backoff)
    h->coop_retry_at = h->coop_eligible + h->coop_backoff;
    h->coop_backoff = std::min<long long>(h->coop_backoff * 2, 1LL << 20);
    ++h->coop_aborts;
    return kCoopFallback;
  }
  ++h->coop_solves;
  *summary = *h->h_ssummary;
  for (int i = 0; i < 7; ++i) pose[i] = h->h_spose[i];
  if (want_trace) {
    const int n = std::min(std::min(summary->num_iterations + 1, trace_cap), h->trace_cap);
    if (n > 0) CLC_HIP(hipMemcpy(trace, h->d_trace, sizeof(clc_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  summary->eval_kernel_ms = 0.0;
  summary->eval_kernel_launches = 0;
  if (timed) {
    float ms = 0.f;
    CLC_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    summary->eval_kernel_ms = (double)ms;
    summary->eval_kernel_launches = 1;
  }
  summary->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve: non-finite result");
  return CLC_OK;
}

}  // namespace

// Profiling hook (not part of include/clc.h): one default clc_solve through the step kernel with HIP events on the
// handle's stream right before launch `first` and right after launch `last`; *avg_ms = elapsed / (last - first + 1),
// i.e. the mean period of those back-to-back step_kernel launches.  Launch 0 evaluates the start pose, launch k >= 1
// consumes pass k-1; choose 2 <= first <= last <= passes - 1 to cover steady-state launches that all streamed.
extern "C" int clc_time_steps(clc_handle* h, const double pose0[7], int first, int last, double* avg_ms, int* passes) {
  if (!h || !pose0 || !avg_ms || first < 0 || last < first) return fail(CLC_ERR_INVALID_ARG, "clc_time_steps: bad argument");
  if (!h->d_tiles || !(h->compact_ok || h->rows_ok)) return fail(CLC_ERR_NO_DATA, "clc_time_steps: no (compact / row) observations uploaded");
  CLC_HIP(hipSetDevice(h->device));
  int rc = ensure_events(h, 2);
  if (rc != CLC_OK) return rc;
  const int grid = eval_grid(h, h->n_obs);
  rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  clc_options opt;
  clc_options_default(&opt);
  double pose[7];
  for (int i = 0; i < 7; ++i) pose[i] = pose0[i];
  clc_summary sm;
  float ms = -1.f;
  rc = solve_stepped(h, opt, grid, pose, &sm, nullptr, 0, std::chrono::steady_clock::now(), first, last, &ms);
  if (rc != CLC_OK) return rc;
  if (passes) *passes = (int)sm.num_evaluations;
  if (ms < 0.f) return fail(CLC_ERR_INVALID_ARG, "clc_time_steps: the solve ended before launch `last`");
  *avg_ms = (double)ms / (double)(last - first + 1);
  return CLC_OK;
}

int clc_solve(clc_handle* h, const clc_options* opt_in, double pose[7], clc_summary* summary,
              clc_iteration* trace, int trace_cap) {
  if (!h || !pose || !summary || trace_cap < 0) return fail(CLC_ERR_INVALID_ARG, "clc_solve: bad argument");
  if (!h->d_tiles) return fail(CLC_ERR_NO_DATA, "clc_solve: no observations uploaded");
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve: non-finite initial pose");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_options_default(&opt);
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, "clc_solve: max_num_iterations < 0");
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, "clc_solve: loss_scale_factor must be > 0");
  CLC_HIP(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();

  // a problem one workgroup holds: the whole solve in one single-workgroup launch (default flags only: the explicit flag
  // sets select the step chain / launch pair the bit-identity tests compare; profile_events = 1 asks for per-pass events)
  if (h->sres.ok && h->launch_auto && (h->auto_disable & 2) == 0 && h->grid_override == 0 && opt.profile_events != 1) return solve_resident_single(h, opt, pose, summary, trace, trace_cap, t0);
  // a problem the 256 CUs hold together: the whole solve in one launch of 256 co-resident workgroups (same conditions)
  if (h->cres.ok && h->launch_auto && (h->auto_disable & 1) == 0 && h->grid_override == 0 && opt.profile_events != 1 && ++h->coop_eligible > h->coop_retry_at) {
    const int rc = solve_coop(h, opt, pose, summary, trace, trace_cap, t0);
    if (rc != kCoopFallback) return rc;
  }
  const int grid = eval_grid(h, h->n_obs);
  int rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  if ((h->launch_flags & clc::FLAG_STEP) != 0 && (((h->launch_flags & clc::FLAG_COMPACT) != 0 && h->compact_ok) || use_rows(h)) &&
      (h->launch_flags & clc::FLAG_WG512) != 0 && h->n_obs < 0x7FFFFFFFull &&
      opt.profile_events != 1)  // 1: HIP events around K1, two-kernel path
    return solve_stepped(h, opt, grid, pose, summary, trace, trace_cap, t0);
  const int max_evals = opt.max_num_iterations + 1;
  const bool want_trace = trace != nullptr && trace_cap > 0;
  if (want_trace) {
    rc = ensure_trace(h, opt.max_num_iterations + 8);
    if (rc != CLC_OK) return rc;
  }
  if (opt.profile_events) {
    rc = ensure_events(h, 2 * (size_t)max_evals);
    if (rc != CLC_OK) return rc;
  }
  // Launch-ahead depth: the host keeps this many LM iterations queued beyond the last one the
  // device has reported done (pinned mailbox), so the stream never drains and the host never
  // blocks; at most `lookahead` already-queued iterations turn into no-ops after termination.
  const int lookahead = opt.launch_ahead > 0 ? opt.launch_ahead : default_lookahead();
  // Controller in the tail of the evaluation launch (one launch per LM iteration) vs its own launch: fused saves a
  // launch boundary (~1 us per iteration) when the evaluation is short, and loses a little when many workgroups
  // queue for the ticket (scripts/size_sweep.py: 0.178 vs 0.193 ms at 5 500 obs, 0.222 vs 0.213 at 1e5, equal at 1e6).
#ifdef CLC_LEGACY_PATHS
  const bool fused = !use_rows(h) && ((h->launch_flags & clc::FLAG_FUSED_LM) != 0 || (h->launch_auto && grid < h->num_cus));
#else
  const bool fused = false;  // (eval_lm_kernel lives in clc_legacy.hpp; the default build runs the launch pair here)
#endif

  clc::HostMailbox* mb = h->h_mailbox;
  mb->n_done = 0;
  mb->status = CLC_RUNNING;
  std::atomic_thread_fence(std::memory_order_seq_cst);

  clc::Pose7 p0;
  for (int i = 0; i < 7; ++i) p0.v[i] = pose[i];
  if (fused) {  // the fused kernel reads the LM state at entry: initialise it with its own launch
    hipLaunchKernelGGL(clc::lm_init_kernel, dim3(1), dim3(64), 0, h->stream, h->d_state, opt, p0, h->d_ticket);
    CLC_HIP(hipGetLastError());
  }

  const double* d_x_eval = reinterpret_cast<const double*>(
      reinterpret_cast<const char*>(h->d_state) + offsetof(clc::LmState, x_eval));
  const int32_t* d_status = reinterpret_cast<const int32_t*>(
      reinterpret_cast<const char*>(h->d_state) + offsetof(clc::LmState, status));
  clc_iteration* d_trace = want_trace ? h->d_trace : nullptr;
  const int d_trace_cap = want_trace ? h->trace_cap : 0;

  int launched = 0;
  int status = CLC_RUNNING;
  long long spins = 0;
  auto t_last_progress = std::chrono::steady_clock::now();
  int last_done = 0;
  for (;;) {
    status = __atomic_load_n(&mb->status, __ATOMIC_ACQUIRE);
    if (status != CLC_RUNNING) break;
    const int done = std::min(__atomic_load_n(&mb->n_done, __ATOMIC_ACQUIRE), launched);  // see solve_stepped
    if (launched < max_evals && launched - done < lookahead) {
      if (opt.profile_events) CLC_HIP(hipEventRecord(h->ev[2 * launched], h->stream));
#ifdef CLC_LEGACY_PATHS
      if (fused) {
        const bool nt = (h->launch_flags & clc::FLAG_NONTEMPORAL) != 0;
        const bool cp = (h->launch_flags & clc::FLAG_COMPACT) != 0 && h->compact_ok;
        const bool big = (h->launch_flags & clc::FLAG_WG512) != 0;
        const bool deep = (h->launch_flags & clc::FLAG_DEEP) != 0 ||
                          (h->launch_auto && (size_t)h->n_obs * 28 > kInfinityCacheBytes);
#define CLC_LAUNCH_FUSED(LOSS, NT, CP, DEEP, BT)                                                          \
  hipLaunchKernelGGL((clc::eval_lm_kernel<LOSS, NT, CP, DEEP, BT>), dim3(grid), dim3(BT), 0, h->stream,   \
                     (CP) ? h->d_ctiles : h->d_tiles, h->d_groups, (long long)h->n_obs, h->d_state, opt,  \
                     h->d_partials, h->d_ticket, d_trace, d_trace_cap, h->d_mailbox)
#define CLC_LAUNCH_FUSED_L(NT, CP, DEEP, BT)                                                               \
  do { if (opt.use_loss) CLC_LAUNCH_FUSED(true, NT, CP, DEEP, BT); else CLC_LAUNCH_FUSED(false, NT, CP, DEEP, BT); } while (0)
        if (cp && big && deep) CLC_LAUNCH_FUSED_L(false, true, true, 512);
        else if (cp && big) CLC_LAUNCH_FUSED_L(false, true, false, 512);
        else if (cp && deep) CLC_LAUNCH_FUSED_L(false, true, true, 256);
        else if (cp) CLC_LAUNCH_FUSED_L(false, true, false, 256);
        else if (nt) CLC_LAUNCH_FUSED_L(true, false, false, 256);
        else CLC_LAUNCH_FUSED_L(false, false, false, 256);
#undef CLC_LAUNCH_FUSED_L
#undef CLC_LAUNCH_FUSED
        if (opt.profile_events) CLC_HIP(hipEventRecord(h->ev[2 * launched + 1], h->stream));
      } else
#endif
      {
        // iteration 0 carries the initial pose by value and initialises the LM state in lm_kernel
        const bool first = launched == 0;
        launch_eval<true>(h, grid, opt.use_loss != 0, d_x_eval, d_status, opt.loss_scale_factor, first ? &p0 : nullptr);
        if (opt.profile_events) CLC_HIP(hipEventRecord(h->ev[2 * launched + 1], h->stream));
        if (first)
          hipLaunchKernelGGL(clc::lm_kernel<true>, dim3(1), dim3(clc::BLOCK), 0, h->stream, h->d_partials, grid,
                             h->d_state, opt, d_trace, d_trace_cap, h->d_mailbox, p0);
        else
          hipLaunchKernelGGL(clc::lm_kernel<false>, dim3(1), dim3(clc::BLOCK), 0, h->stream, h->d_partials, grid,
                             h->d_state, opt, d_trace, d_trace_cap, h->d_mailbox, p0);
      }
      ++launched;
      continue;
    }
    // nothing to launch: wait for the device (bounded: a wedged queue must not hang the caller)
    if (done != last_done) { last_done = done; t_last_progress = std::chrono::steady_clock::now(); spins = 0; }
    if ((++spins & 0xFFFF) == 0) {
      hipError_t e = hipStreamQuery(h->stream);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(CLC_ERR_HIP, "clc_solve: stream error", e);
      if (e == hipSuccess) {  // queue drained: the mailbox must be final now
        status = __atomic_load_n(&mb->status, __ATOMIC_ACQUIRE);
        if (status != CLC_RUNNING) break;
        if (launched >= max_evals && __atomic_load_n(&mb->n_done, __ATOMIC_ACQUIRE) >= launched)
          return fail(CLC_ERR_HIP, "clc_solve: controller did not terminate");
      }
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last_progress).count();
      if (waited > 30.0) return fail(CLC_ERR_HIP, "clc_solve: no progress from the device for 30 s");
    }
  }
  CLC_HIP(hipGetLastError());
  std::atomic_thread_fence(std::memory_order_acquire);
  *summary = mb->summary;
  for (int i = 0; i < 7; ++i) pose[i] = mb->pose[i];
  summary->eval_kernel_ms = 0.0;
  summary->eval_kernel_launches = 0;
  if (want_trace || opt.profile_events) CLC_HIP(hipStreamSynchronize(h->stream));
  if (want_trace) {
    const int n = std::min(std::min(summary->num_iterations + 1, trace_cap), h->trace_cap);
    if (n > 0) CLC_HIP(hipMemcpy(trace, h->d_trace, sizeof(clc_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  if (opt.profile_events) {
    const int n_real = (int)std::min<int64_t>(summary->num_evaluations, launched);
    double tot = 0.0;
    for (int i = 0; i < n_real; ++i) {
      float ms = 0.f;
      CLC_HIP(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
      tot += ms;
    }
    summary->eval_kernel_ms = tot;
    summary->eval_kernel_launches = n_real;
  }
  summary->solve_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve: non-finite result");
  return CLC_OK;
}

int clc_information(clc_handle* h, const double pose[7], double H[36], double b[6], double* chi2,
                    double sv[6], double V[36], int* n_null) {
  if (!h || !pose || !H || !b || !chi2 || !sv || !n_null)
    return fail(CLC_ERR_INVALID_ARG, "clc_information: bad argument");
  double cost, g[6], H21[21];
  int rc = clc_eval(h, pose, /*with_loss=*/0, 0.0, &cost, g, H21);  // :323-362: no loss
  if (rc != CLC_OK) return rc;
  int idx = 0;
  for (int a = 0; a < 6; ++a)
    for (int c = a; c < 6; ++c) {
      H[6 * a + c] = H21[idx];
      H[6 * c + a] = H21[idx];
      ++idx;
    }
  for (int a = 0; a < 6; ++a) b[a] = -g[a];  // b -= J^T r, :357
  *chi2 = 2.0 * cost;                        // chi += r*r, :359
  double Vtmp[36];
  clc::host::jacobi_eig_sym(H, 6, sv, V ? V : Vtmp);  // JacobiSVD(H), :366
  int n = 0;
  for (int i = 0; i < 6; ++i)
    if (sv[i] < 1e-8) ++n;  // :371
  *n_null = n;
  return CLC_OK;
}

int clc_closed_form(clc_handle* h, double Tlc[16], int* unobservable, double sv9[9]) {
  if (!h || !Tlc || !unobservable) return fail(CLC_ERR_INVALID_ARG, "clc_closed_form: bad argument");
  if (!h->d_tiles || h->n_obs == 0) return fail(CLC_ERR_NO_DATA, "clc_closed_form: no observations uploaded");
  CLC_HIP(hipSetDevice(h->device));
  const int grid = eval_grid(h, h->n_obs);
  int rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  if (use_rows(h)) {
    const clc::RowDesc* rdesc = reinterpret_cast<const clc::RowDesc*>(h->d_rdesc);
    if (h->rows_z) {  // bar_p = (x, y, 1): z is not read, only the row stride differs
      if (rows_nontemporal(h, h->n_rows, true))
        hipLaunchKernelGGL((clc::normal9_rows_kernel<true, clc::ROW_DOUBLES_Z>), dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy, rdesc, h->n_rows, h->d_partials);
      else
        hipLaunchKernelGGL((clc::normal9_rows_kernel<false, clc::ROW_DOUBLES_Z>), dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy, rdesc, h->n_rows, h->d_partials);
    }
    else if (rows_nontemporal(h, h->n_rows))
      hipLaunchKernelGGL(clc::normal9_rows_kernel<true>, dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy,
                         reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, h->d_partials);
    else
      hipLaunchKernelGGL(clc::normal9_rows_kernel<false>, dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy,
                         reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, h->d_partials);
  } else {
    hipLaunchKernelGGL(clc::normal9_kernel, dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_tiles,
                       (long long)h->n_obs, h->d_partials);
  }
  CLC_HIP(hipGetLastError());
  hipLaunchKernelGGL(clc::reduce9_kernel, dim3(1), dim3(clc::BLOCK), 0, h->stream, h->d_partials, grid,
                     h->d_small + 128);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipMemcpyAsync(h->h_small + 128, h->d_small + 128, clc::NACC9 * sizeof(double),
                         hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  const double* r = h->h_small + 128;
  auto tri3 = [](int a, int b) { if (a > b) std::swap(a, b); return a * 3 - (a * (a - 1)) / 2 + (b - a); };
  double AtA[81], Atb[9];
  for (int ci = 0; ci < 3; ++ci)
    for (int ri = 0; ri < 3; ++ri) {
      for (int cj = 0; cj < 3; ++cj)
        for (int rj = 0; rj < 3; ++rj) AtA[9 * (3 * ci + ri) + (3 * cj + rj)] = r[6 * tri3(ci, cj) + tri3(ri, rj)];
      Atb[3 * ci + ri] = r[36 + 3 * ci + ri];
    }
  rc = clc::host::closed_form_from_normal(AtA, Atb, Tlc, unobservable, sv9);
  if (rc != CLC_OK) return fail(rc, "clc_closed_form: non-finite solution of the 9x9 normal equation");
  return CLC_OK;
}

// ---- batched ---------------------------------------------------------------------------
namespace {
// records: host pointer (on_device = false: staged through a temporary device buffer) or device pointer
int upload_batched_impl(clc_handle* h, const clc_observation* records, bool on_device, const int64_t* offsets,
                        size_t n_problems) {
  if (!h || !offsets || (n_problems > 0 && !records))
    return fail(CLC_ERR_INVALID_ARG, "clc_upload_batched: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  const size_t P = n_problems;
  std::vector<long long> tile_off(P + 1, 0), nobs(P, 0);
  long long max_tiles = 0;
  for (size_t k = 0; k < P; ++k) {
    const int64_t n = offsets[k + 1] - offsets[k];
    if (n < 0) return fail(CLC_ERR_INVALID_ARG, "clc_upload_batched: offsets not monotone");
    nobs[k] = n;
    tile_off[k + 1] = tile_off[k] + (n + clc::TILE - 1) / clc::TILE;
    max_tiles = std::max<long long>(max_tiles, tile_off[k + 1] - tile_off[k]);
  }
  h->batch_max_tiles = max_tiles;
  const size_t total_tiles = (size_t)tile_off[P];
  h->batch_total_tiles = total_tiles;
  const size_t bytes = std::max<size_t>(total_tiles, 1) * clc::TILE_DOUBLES * sizeof(double);
  if (bytes > h->btiles_cap_bytes) {
    if (h->d_btiles) CLC_HIP(hipFree(h->d_btiles));
    h->d_btiles = nullptr; h->btiles_cap_bytes = 0;
    CLC_HIP(hipMalloc(&h->d_btiles, bytes));
    h->btiles_cap_bytes = bytes;
  }
  if (P > h->problems_cap) {
    void* olds[] = {h->d_tile_off, h->d_nobs, h->d_states, h->d_results, h->d_prob_row};
    for (void* p : olds) if (p) CLC_HIP(hipFree(p));
    if (h->h_poses) CLC_HIP(hipHostFree(h->h_poses));
    if (h->h_summaries) CLC_HIP(hipHostFree(h->h_summaries));
    h->h_poses = nullptr; h->h_summaries = nullptr;
    h->d_tile_off = nullptr; h->d_nobs = nullptr; h->d_poses = nullptr; h->d_summaries = nullptr;
    h->d_states = nullptr; h->d_results = nullptr; h->results_valid = 0; h->d_prob_row = nullptr;
    h->problems_cap = 0;
    CLC_HIP(hipMalloc(&h->d_tile_off, sizeof(long long) * (P + 1)));
    CLC_HIP(hipMalloc(&h->d_nobs, sizeof(long long) * P));
    CLC_HIP(hipHostMalloc(&h->h_poses, sizeof(double) * 7 * P, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_poses), h->h_poses, 0));
    CLC_HIP(hipHostMalloc(&h->h_summaries, sizeof(clc_summary) * P, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_summaries), h->h_summaries, 0));
    CLC_HIP(hipMalloc(&h->d_states, sizeof(clc::LmState) * P));
    CLC_HIP(hipMalloc(&h->d_results, sizeof(clc_result_record) * P));
    CLC_HIP(hipMalloc(&h->d_prob_row, sizeof(long long) * (P + 1)));
    h->problems_cap = P;
  }
  if (P == 0) { h->n_problems = 0; return CLC_OK; }
  CLC_HIP(hipMemcpy(h->d_tile_off, tile_off.data(), sizeof(long long) * (P + 1), hipMemcpyHostToDevice));
  CLC_HIP(hipMemcpy(h->d_nobs, nobs.data(), sizeof(long long) * P, hipMemcpyHostToDevice));
  // stage the AoS records, then re-tile every problem into its own whole tiles
  const size_t n_total = (size_t)(offsets[P] - offsets[0]);
  DevBuf<double> baos(&h->pool);
  DevBuf<long long> boff(&h->pool);
  double* d_aos = nullptr;
  if (n_total > 0 && on_device) {
    d_aos = const_cast<double*>(reinterpret_cast<const double*>(records + offsets[0]));
  } else if (n_total > 0) {
    CLC_HIP(baos.alloc(n_total * 8));
    CLC_HIP(hipMemcpy(baos.p, records + offsets[0], n_total * sizeof(clc_observation), hipMemcpyHostToDevice));
    d_aos = baos.p;
  }
  std::vector<long long> rel(P + 1);
  for (size_t k = 0; k <= P; ++k) rel[k] = offsets[k] - offsets[0];
  CLC_HIP(boff.alloc(P + 1));
  long long* d_off = boff.p;
  CLC_HIP(hipMemcpy(d_off, rel.data(), sizeof(long long) * (P + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(clc::retile_batched_kernel, dim3((unsigned)P), dim3(256), 0, h->stream, d_aos, d_off,
                     h->d_tile_off, h->d_btiles);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  int crc = CLC_OK;
  h->bcompact_ok = false;
  h->brows_ok = false;
  h->bres.ok = false;
  h->results_valid = 0;
  if (e == hipSuccess && d_aos) {
    const LayoutTargets T = {&h->d_bctiles, &h->bctiles_cap_bytes, &h->d_bgroups, &h->bgroups_cap_bytes, &h->bn_groups,
                             &h->bcompact_ok, &h->d_brxy, &h->brxy_cap_bytes, &h->d_brdesc, &h->brdesc_cap_bytes, &h->bn_rows,
                             &h->brows_ok, &h->d_prob_row, &h->bres, &h->brows_z};
    crc = build_layouts(h, d_aos, n_total, rel, tile_off, T);
  }
  if (e != hipSuccess) return fail(CLC_ERR_HIP, "clc_upload_batched: retile", e);
  if (crc != CLC_OK) return crc;
  h->batch_max_rows = 0;
  if (h->brows_ok) {  // O(P) words back: the whole-solve kernel is chosen on the real longest problem, not an estimate
    std::vector<long long> pr(P + 1);
    CLC_HIP(hipMemcpy(pr.data(), h->d_prob_row, sizeof(long long) * (P + 1), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < P; ++k) h->batch_max_rows = std::max(h->batch_max_rows, pr[k + 1] - pr[k]);
  }
  h->n_problems = P;
  return CLC_OK;
}
}  // namespace

int clc_upload_batched(clc_handle* h, const clc_observation* records, const int64_t* offsets, size_t n_problems) {
  return upload_batched_impl(h, records, false, offsets, n_problems);
}

int clc_upload_batched_device(clc_handle* h, const clc_observation* records_dev, const int64_t* offsets, size_t n_problems) {
  return upload_batched_impl(h, records_dev, true, offsets, n_problems);
}

size_t clc_num_problems(const clc_handle* h) { return h ? h->n_problems : 0; }

int clc_batched_host_buffers(clc_handle* h, double** poses, clc_summary** summaries) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_batched_host_buffers: NULL handle");
  if (h->n_problems == 0 || !h->h_poses) return fail(CLC_ERR_NO_DATA, "clc_batched_host_buffers: no problems uploaded");
  if (poses) *poses = h->h_poses;
  if (summaries) *summaries = h->h_summaries;
  return CLC_OK;
}

int clc_solve_batched(clc_handle* h, const clc_options* opt_in, double* poses, clc_summary* summaries) {
  if (!h || !poses || !summaries) return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched: bad argument");
  if (!h->d_btiles || h->n_problems == 0) return fail(CLC_ERR_NO_DATA, "clc_solve_batched: no problems uploaded");
  // the handle's own pinned arrays (clc_batched_host_buffers): solved in place, no staging copies
  const bool in_place = poses == h->h_poses && summaries == h->h_summaries;
  if ((poses == h->h_poses) != (summaries == h->h_summaries))
    return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched: pass both of the handle's host buffers or neither");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_options_default(&opt);
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched: max_num_iterations < 0");
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched: loss_scale_factor must be > 0");
  const size_t P = h->n_problems;
  for (size_t i = 0; i < 7 * P; ++i)
    if (!std::isfinite(poses[i])) return fail(CLC_ERR_NONFINITE, "clc_solve_batched: non-finite initial pose");
  CLC_HIP(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  BatchedLaunch bl;
  {
    const int rc = batched_launch_setup(h, opt, &bl);
    if (rc != CLC_OK) return rc;
  }
  const int bpp = bl.bpp;
  // (the previous batch ended with a stream synchronisation: nothing still reads or writes the staging buffers)
  if (!in_place) std::memcpy(h->h_poses, poses, sizeof(double) * 7 * P);
  if (bl.resident) {
    // one workgroup per problem, the problem read from HBM once and kept in registers + LDS for its whole solve
    const unsigned int* d_row = reinterpret_cast<const unsigned int*>(h->bres.d_row);
    const clc::ResLane* d_desc = reinterpret_cast<const clc::ResLane*>(h->bres.d_desc);
#define CLC_LAUNCH_RES(LOSS, NT, NW, PR, PL)                                                                                  \
  hipLaunchKernelGGL((clc::resident_solve_kernel<LOSS, NT, NW, PR, PL, kResCtrl##NW>), dim3((unsigned)P), dim3(NW * 64), 0, h->stream, \
                     h->bres.d_xy, d_row, d_desc, h->d_bgroups, h->bres.uni_ppl, opt, nullptr, 0, h->d_poses, h->d_summaries, h->d_results, nullptr, nullptr)
#define CLC_LAUNCH_RES_V(NW, PR, PL)                                                                                          \
  do {                                                                                                                        \
    if (opt.use_loss) { if (bl.res_nt) CLC_LAUNCH_RES(true, true, NW, PR, PL); else CLC_LAUNCH_RES(true, false, NW, PR, PL); } \
    else { if (bl.res_nt) CLC_LAUNCH_RES(false, true, NW, PR, PL); else CLC_LAUNCH_RES(false, false, NW, PR, PL); }            \
  } while (0)
    // (A completion flag raised by the last workgroup to finish, polled by the host instead of this blocking synchronisation, was
    // measured: every workgroup then needs a system-scope release before it counts itself in, which on this part writes back L2 —
    // C4 shard 0.93 -> 1.28 ms, C3 0.150 -> 0.166.  The single-workgroup solve keeps its flag: one release per solve.)
    const bool timed = opt.profile_events == 1;  // HIP event pair around the one launch -> clc_summary.eval_kernel_ms of every problem
    if (timed) {
      const int rc = ensure_events(h, 2);
      if (rc != CLC_OK) return rc;
      CLC_HIP(hipEventRecord(h->ev[0], h->stream));
    }
    if (h->bres.lanes == 256) CLC_LAUNCH_RES_V(4, kResPR256, kResPL256);
    else CLC_LAUNCH_RES_V(8, kResPR512, kResPL512);
#undef CLC_LAUNCH_RES_V
#undef CLC_LAUNCH_RES
    CLC_HIP(hipGetLastError());
    if (timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));  // kernel completion makes the outcomes written over PCIe visible
    float kernel_ms = 0.0f;
    if (timed) CLC_HIP(hipEventElapsedTime(&kernel_ms, h->ev[0], h->ev[1]));
    h->results_valid = P;
    if (!in_place) {
      std::memcpy(poses, h->h_poses, sizeof(double) * 7 * P);
      std::memcpy(summaries, h->h_summaries, sizeof(clc_summary) * P);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t k = 0; k < P; ++k) {
      summaries[k].solve_ms = ms;
      if (timed) { summaries[k].eval_kernel_ms = (double)kernel_ms; summaries[k].eval_kernel_launches = 1; }
    }
    return CLC_OK;
  }
  if (bl.whole_solve) {
    // one 256-thread workgroup per problem: the whole solve of every problem in ONE launch (batched_solve_kernel)
    const clc::RowDesc* bdesc = reinterpret_cast<const clc::RowDesc*>(h->d_brdesc);
#define CLC_LAUNCH_SOLVE(LOSS, NT)                                                                                      \
  hipLaunchKernelGGL((clc::batched_solve_kernel<LOSS, NT>), dim3((unsigned)P), dim3(clc::BLOCK), 0, h->stream, h->d_brxy, \
                     bdesc, h->d_prob_row, opt, h->d_poses, h->d_summaries, h->d_results)
    if (opt.use_loss) { if (bl.rows_nt) CLC_LAUNCH_SOLVE(true, true); else CLC_LAUNCH_SOLVE(true, false); }
    else { if (bl.rows_nt) CLC_LAUNCH_SOLVE(false, true); else CLC_LAUNCH_SOLVE(false, false); }
#undef CLC_LAUNCH_SOLVE
    CLC_HIP(hipGetLastError());
    CLC_HIP(hipStreamSynchronize(h->stream));  // kernel completion makes the outcomes written over PCIe visible
    h->results_valid = P;
    if (!in_place) {
      std::memcpy(poses, h->h_poses, sizeof(double) * 7 * P);
      std::memcpy(summaries, h->h_summaries, sizeof(clc_summary) * P);
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t k = 0; k < P; ++k) summaries[k].solve_ms = ms;
    return CLC_OK;
  }
  const int lm_threads = bl.lm_threads;
  const unsigned lm_blocks = bl.lm_blocks;
  hipLaunchKernelGGL(clc::batched_init_kernel, dim3(lm_blocks), dim3(lm_threads), 0, h->stream, h->d_states,
                     opt, h->d_poses, (int)P, h->d_queue, h->d_ticket);
  CLC_HIP(hipGetLastError());
  const int lookahead = opt.launch_ahead > 0 ? opt.launch_ahead : default_lookahead();
  const int max_evals = opt.max_num_iterations + 1;
  clc::HostMailbox* mb = h->h_mailbox;
  mb->n_done = 0;
  mb->status = CLC_RUNNING;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  int launched = 0;
  long long spins = 0;
  int last_done = 0;
  auto t_last_progress = std::chrono::steady_clock::now();
  for (;;) {
    if (__atomic_load_n(&mb->status, __ATOMIC_ACQUIRE) != CLC_RUNNING) break;
    const int done = __atomic_load_n(&mb->n_done, __ATOMIC_ACQUIRE);
    if (launched < max_evals && launched - done < lookahead) {
      launch_batched_eval(h, opt, bl);
      hipLaunchKernelGGL(clc::batched_lm_kernel, dim3(lm_blocks), dim3(lm_threads), 0, h->stream,
                         h->d_bpartials, bpp, h->d_states, opt, (int)P, h->d_queue, h->d_ticket, launched,
                         h->d_mailbox, h->d_poses, h->d_summaries, h->d_results);
      ++launched;
      continue;
    }
    if (launched >= max_evals && done >= launched) break;  // iteration cap reached for the stragglers
    if (done != last_done) { last_done = done; t_last_progress = std::chrono::steady_clock::now(); spins = 0; }
    if ((++spins & 0xFFFF) == 0) {
      hipError_t e = hipStreamQuery(h->stream);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(CLC_ERR_HIP, "clc_solve_batched: stream error", e);
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last_progress).count();
      if (waited > 60.0) return fail(CLC_ERR_HIP, "clc_solve_batched: no progress from the device for 60 s");
    }
  }
  CLC_HIP(hipGetLastError());
  if (__atomic_load_n(&mb->status, __ATOMIC_ACQUIRE) == CLC_RUNNING) {  // iteration cap of this loop: some problem still runs
    hipLaunchKernelGGL(clc::batched_finish_kernel, dim3(lm_blocks), dim3(lm_threads), 0, h->stream, h->d_states,
                       (int)P, h->d_poses, h->d_summaries, h->d_results);
    CLC_HIP(hipGetLastError());
  }
  CLC_HIP(hipStreamSynchronize(h->stream));  // kernel completion makes the outcomes written over PCIe visible
  h->results_valid = P;
  if (!in_place) {
    std::memcpy(poses, h->h_poses, sizeof(double) * 7 * P);
    std::memcpy(summaries, h->h_summaries, sizeof(clc_summary) * P);
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (size_t k = 0; k < P; ++k) {
    summaries[k].solve_ms = ms;
    if (summaries[k].termination == CLC_RUNNING) summaries[k].termination = CLC_FAILURE;
  }
  return CLC_OK;
}

// ---- line fitting ---------------------------------------------------------------------------
void clc_line_options_default(clc_options* o) {
  clc_options_default(o);
  if (!o) return;
  o->max_num_iterations = 10;   // src/LaseCamCalCeres.cpp:425
  o->loss_scale_factor = 0.05;  // CauchyLoss(0.05), :416 (no per-residual scale here)
}

int clc_line_fit_batched(clc_handle* h, const clc_options* opt_in, const double* xy, const int64_t* offsets,
                         size_t n_scans, double* lines, clc_summary* summaries) {
  if (!h || !offsets || !lines || (n_scans > 0 && offsets[n_scans] > offsets[0] && !xy))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: bad argument");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_line_options_default(&opt);
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: max_num_iterations < 0");
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: loss_scale_factor must be > 0");
  if (n_scans == 0) return CLC_OK;
  if (n_scans > 0x7FFFFFF0ull) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: too many scans");
  for (size_t k = 0; k < n_scans; ++k)
    if (offsets[k + 1] < offsets[k]) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: offsets not monotone");
  for (size_t i = 0; i < 2 * n_scans; ++i)
    if (!std::isfinite(lines[i])) return fail(CLC_ERR_NONFINITE, "clc_line_fit_batched: non-finite initial line");
  CLC_HIP(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  const size_t n_pts = (size_t)(offsets[n_scans] - offsets[0]);
  std::vector<long long> rel(n_scans + 1);
  for (size_t k = 0; k <= n_scans; ++k) rel[k] = offsets[k] - offsets[0];
  DevBuf<double> bxy(&h->pool), blines(&h->pool);
  DevBuf<long long> boff(&h->pool);
  DevBuf<clc_summary> bsum(&h->pool);
  CLC_HIP(bxy.alloc(n_pts * 2));
  CLC_HIP(boff.alloc(n_scans + 1));
  CLC_HIP(blines.alloc(n_scans * 2));
  if (summaries) CLC_HIP(bsum.alloc(n_scans));
  double *d_xy = bxy.p, *d_lines = blines.p;
  long long* d_off = boff.p;
  clc_summary* d_sum = bsum.p;
  hipError_t e = hipSuccess;
  if (n_pts > 0) e = hipMemcpyAsync(d_xy, xy + 2 * offsets[0], n_pts * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_off, rel.data(), (n_scans + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_lines, lines, n_scans * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    const unsigned blocks = (unsigned)((n_scans + clc::LINE_SCANS_PER_BLOCK - 1) / clc::LINE_SCANS_PER_BLOCK);
    if (opt.use_loss)
      hipLaunchKernelGGL((clc::line_fit_kernel<true>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, d_xy, d_off,
                         (int)n_scans, opt, d_lines, d_sum);
    else
      hipLaunchKernelGGL((clc::line_fit_kernel<false>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, d_xy, d_off,
                         (int)n_scans, opt, d_lines, d_sum);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(lines, d_lines, n_scans * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && summaries)
    e = hipMemcpyAsync(summaries, d_sum, n_scans * sizeof(clc_summary), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail(CLC_ERR_HIP, "clc_line_fit_batched", e);
  if (summaries) {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t k = 0; k < n_scans; ++k) summaries[k].solve_ms = ms;
  }
  return CLC_OK;
}

int clc_scan_to_points(clc_handle* h, const float* ranges, const int64_t* offsets, size_t n_scans,
                       const float* angle_min, const float* angle_increment, const float* range_min,
                       double* points) {
  if (!h || !offsets || (n_scans > 0 && (!angle_min || !angle_increment || !range_min)))
    return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: bad argument");
  if (n_scans == 0) return CLC_OK;
  if (n_scans > 65535) return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: at most 65535 scans per call");
  for (size_t k = 0; k < n_scans; ++k)
    if (offsets[k + 1] < offsets[k]) return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: offsets not monotone");
  const size_t n = (size_t)(offsets[n_scans] - offsets[0]);
  if (n == 0) return CLC_OK;
  if (!ranges || !points) return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  std::vector<long long> rel(n_scans + 1);
  long long longest = 0;
  for (size_t k = 0; k <= n_scans; ++k) rel[k] = offsets[k] - offsets[0];
  for (size_t k = 0; k < n_scans; ++k) longest = std::max(longest, rel[k + 1] - rel[k]);
  DevBuf<float> br(&h->pool), bam(&h->pool), bai(&h->pool), brm(&h->pool);
  DevBuf<long long> boff(&h->pool);
  DevBuf<double> bp(&h->pool);
  CLC_HIP(br.alloc(n)); CLC_HIP(bam.alloc(n_scans)); CLC_HIP(bai.alloc(n_scans)); CLC_HIP(brm.alloc(n_scans));
  CLC_HIP(boff.alloc(n_scans + 1)); CLC_HIP(bp.alloc(3 * n));
  CLC_HIP(hipMemcpyAsync(br.p, ranges + offsets[0], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(bam.p, angle_min, n_scans * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(bai.p, angle_increment, n_scans * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(brm.p, range_min, n_scans * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(boff.p, rel.data(), (n_scans + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  const int threads = 256;
  const unsigned gx = (unsigned)std::min<long long>(64, std::max<long long>(1, (longest + threads - 1) / threads));
  hipLaunchKernelGGL(clc::scan_to_points_kernel, dim3(gx, (unsigned)n_scans), dim3(threads), 0, h->stream, br.p, boff.p,
                     (int)n_scans, bam.p, bai.p, brm.p, bp.p);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipMemcpyAsync(points + 3 * offsets[0], bp.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  return CLC_OK;
}

int clc_line_fit_batched_device(clc_handle* h, const clc_options* opt_in, const double* xy_dev, const int64_t* offsets_dev,
                                size_t n_scans, double* lines_dev, clc_summary* summaries_dev) {
  if (!h || (n_scans > 0 && (!offsets_dev || !lines_dev || !xy_dev)))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: bad argument");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_line_options_default(&opt);
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: max_num_iterations < 0");
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: loss_scale_factor must be > 0");
  if (n_scans == 0) return CLC_OK;
  if (n_scans > 0x7FFFFFF0ull) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: too many scans");
  CLC_HIP(hipSetDevice(h->device));
  static_assert(sizeof(long long) == sizeof(int64_t), "offset type");
  const unsigned blocks = (unsigned)((n_scans + clc::LINE_SCANS_PER_BLOCK - 1) / clc::LINE_SCANS_PER_BLOCK);
  const long long* d_off = reinterpret_cast<const long long*>(offsets_dev);
  if (opt.use_loss)
    hipLaunchKernelGGL((clc::line_fit_kernel<true>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, xy_dev, d_off,
                       (int)n_scans, opt, lines_dev, summaries_dev);
  else
    hipLaunchKernelGGL((clc::line_fit_kernel<false>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, xy_dev, d_off,
                       (int)n_scans, opt, lines_dev, summaries_dev);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  return CLC_OK;
}

int clc_scan_to_points_device(clc_handle* h, const float* ranges_dev, const int64_t* offsets_dev, size_t n_scans,
                              size_t n_rays, const float* angle_min_dev, const float* angle_increment_dev,
                              const float* range_min_dev, double* points_dev) {
  if (!h || (n_scans > 0 && (!offsets_dev || !angle_min_dev || !angle_increment_dev || !range_min_dev)) ||
      (n_rays > 0 && (!ranges_dev || !points_dev || n_scans == 0)))
    return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points_device: bad argument");
  if (n_rays == 0) return CLC_OK;
  CLC_HIP(hipSetDevice(h->device));
  const int threads = 256;
  hipLaunchKernelGGL(clc::scan_to_points_flat_kernel, dim3((unsigned)((n_rays + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, ranges_dev, reinterpret_cast<const long long*>(offsets_dev), (long long)n_scans,
                     (long long)n_rays, angle_min_dev, angle_increment_dev, range_min_dev, points_dev);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  return CLC_OK;
}

// ---- multi-GPU: RCCL gather of the sharded batch's result records ------------------------------
}  // extern "C"

#include <rccl/rccl.h>  // types and prototypes only: the functions are bound at run time (no link dependency)

namespace {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;        // optional: what RCCL itself says the communicator spans
  decltype(&ncclCommUserRank) CommUserRank = nullptr;  // optional
  std::string origin;
  std::string error;
};

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
  const char* name = info->dlpi_name;
  if (name && std::strstr(name, "librccl")) {
    *static_cast<std::string*>(data) = name;
    return 1;
  }
  return 0;
}

// One RCCL per process: reuse the copy that is already mapped (PyTorch bundles its own), else load the system one.
RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> candidates;
    if (const char* env = std::getenv("CLC_RCCL_LIBRARY")) candidates.push_back(env);
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) candidates.push_back(loaded);
    candidates.push_back("librccl.so.1");
    candidates.push_back("librccl.so");
    candidates.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& c : candidates) {
      api.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.lib) { api.origin = c; break; }
      const char* why = dlerror();
      api.error += c + ": " + (why ? why : "?") + "; ";
    }
    if (!api.lib) return;
#define CLC_BIND(field, sym)                                                   \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym));      \
  if (!api.field) { api.error += std::string("missing symbol ") + sym + "; "; }
    CLC_BIND(GetUniqueId, "ncclGetUniqueId")
    CLC_BIND(CommInitRank, "ncclCommInitRank")
    CLC_BIND(CommDestroy, "ncclCommDestroy")
    CLC_BIND(AllGather, "ncclAllGather")
    CLC_BIND(GetErrorString, "ncclGetErrorString")
#undef CLC_BIND
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.lib, "ncclCommUserRank"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
      api.lib = nullptr;
    }
  });
  return api;
}

int rccl_fail(const char* what, ncclResult_t r) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
  g_last_error = buf;
  return CLC_ERR_COMM;
}

}  // namespace

struct clc_comm {
  clc_handle* h = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  double* d_send = nullptr;
  double* d_recv = nullptr;
  double* h_recv = nullptr;  // pinned
  size_t cap = 0;            // records per rank the buffers hold
};

extern "C" {

int clc_comm_unique_id(char id[CLC_COMM_ID_BYTES]) {
  if (!id) return fail(CLC_ERR_INVALID_ARG, "clc_comm_unique_id: NULL id");
  static_assert(sizeof(ncclUniqueId) == CLC_COMM_ID_BYTES, "ncclUniqueId size");
  RcclApi& api = rccl();
  if (!api.lib) return fail(CLC_ERR_COMM, ("clc_comm_unique_id: RCCL not available: " + api.error).c_str());
  ncclUniqueId u;
  ncclResult_t r = api.GetUniqueId(&u);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  std::memcpy(id, u.internal, CLC_COMM_ID_BYTES);
  return CLC_OK;
}

int clc_comm_create(clc_comm** out, clc_handle* h, const char id[CLC_COMM_ID_BYTES], int rank, int world) {
  if (!out || !h || !id || world < 1 || rank < 0 || rank >= world)
    return fail(CLC_ERR_INVALID_ARG, "clc_comm_create: bad argument");
  *out = nullptr;
  RcclApi& api = rccl();
  if (!api.lib) return fail(CLC_ERR_COMM, ("clc_comm_create: RCCL not available: " + api.error).c_str());
  CLC_HIP(hipSetDevice(h->device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, CLC_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  ncclResult_t r = api.CommInitRank(&comm, world, u, rank);
  if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
  clc_comm* c = new clc_comm();
  c->h = h;
  c->comm = comm;
  c->rank = rank;
  c->world = world;
  // what RCCL itself reports for the communicator (clc_comm_world / clc_comm_rank return these): a SCALE record can then
  // show that RCCL saw N ranks, not just that N was passed in
  int n = 0;
  if (api.CommCount && api.CommCount(comm, &n) == ncclSuccess && n > 0) c->world = n;
  if (api.CommUserRank && api.CommUserRank(comm, &n) == ncclSuccess) c->rank = n;
  *out = c;
  return CLC_OK;
}

void clc_comm_destroy(clc_comm* c) {
  if (!c) return;
  if (c->h) {
    (void)hipSetDevice(c->h->device);
    (void)hipStreamSynchronize(c->h->stream);
  }
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  delete c;
}

int clc_comm_rank(const clc_comm* c) { return c ? c->rank : -1; }
int clc_comm_world(const clc_comm* c) { return c ? c->world : 0; }
// which RCCL was bound ("" before the first comm call) — diagnostics / tests
const char* clc_comm_library(void) { return rccl().origin.c_str(); }

int clc_gather_results(clc_comm* c, int64_t first_global_index, size_t cap_per_rank, clc_result_record* all_records) {
  if (!c || cap_per_rank == 0 || first_global_index < 0)
    return fail(CLC_ERR_INVALID_ARG, "clc_gather_results: bad argument");
  clc_handle* h = c->h;
  size_t n_local = h->results_valid;
  // This is a collective: a rank that returned before the all-gather would leave every other rank blocked in it.  A rank
  // with a LOCAL problem therefore still enters the collective — with an all-padding send buffer — and reports its error
  // afterwards.  (Arguments every rank passes alike — a NULL communicator, cap_per_rank == 0 — are rejected above on all
  // ranks together; a failed device allocation below cannot be papered over: the communicator is then unusable.)
  int local_rc = CLC_OK;
  const char* local_msg = nullptr;
  if (n_local > cap_per_rank) { local_rc = CLC_ERR_INVALID_ARG; local_msg = "clc_gather_results: cap_per_rank < local problems (this rank contributed padding only)"; n_local = 0; }
  else if (n_local > 0 && !h->d_results) { local_rc = CLC_ERR_NO_DATA; local_msg = "clc_gather_results: no solved batch on the handle (this rank contributed padding only)"; n_local = 0; }
  CLC_HIP(hipSetDevice(h->device));
  if (cap_per_rank > c->cap) {
    if (c->d_send) CLC_HIP(hipFree(c->d_send));
    if (c->d_recv) CLC_HIP(hipFree(c->d_recv));
    if (c->h_recv) CLC_HIP(hipHostFree(c->h_recv));
    c->d_send = c->d_recv = c->h_recv = nullptr;
    c->cap = 0;
    CLC_HIP(hipMalloc(&c->d_send, sizeof(clc_result_record) * cap_per_rank));
    CLC_HIP(hipMalloc(&c->d_recv, sizeof(clc_result_record) * cap_per_rank * (size_t)c->world));
    CLC_HIP(hipHostMalloc(&c->h_recv, sizeof(clc_result_record) * cap_per_rank * (size_t)c->world, hipHostMallocDefault));
    c->cap = cap_per_rank;
  }
  const int threads = 256;
  hipLaunchKernelGGL(clc::pack_results_kernel, dim3((unsigned)((cap_per_rank + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, h->d_results, (long long)n_local, (long long)cap_per_rank, (double)first_global_index,
                     c->d_send);
  CLC_HIP(hipGetLastError());
  const size_t count = cap_per_rank * (sizeof(clc_result_record) / sizeof(double));
  ncclResult_t r = rccl().AllGather(c->d_send, c->d_recv, count, ncclDouble, c->comm, h->stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  const size_t bytes = sizeof(clc_result_record) * cap_per_rank * (size_t)c->world;
  CLC_HIP(hipMemcpyAsync(c->h_recv, c->d_recv, bytes, hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  if (all_records) std::memcpy(all_records, c->h_recv, bytes);
  if (local_rc != CLC_OK) return fail(local_rc, local_msg);
  return CLC_OK;
}

const clc_result_record* clc_comm_records(const clc_comm* c) {
  return c ? reinterpret_cast<const clc_result_record*>(c->h_recv) : nullptr;
}

// ---- test hooks --------------------------------------------------------------------------
// Runs only the wavefront reduction on in[64*28] -> out[28] (reduce_mode 0 butterfly, 1 shuffle).
int clc_debug_wave_reduce(clc_handle* h, const double* in, double* out, int reduce_mode) {
  if (!h || !in || !out) return fail(CLC_ERR_INVALID_ARG, "clc_debug_wave_reduce: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> buf(&h->pool);
  CLC_HIP(buf.alloc(64 * clc::NACC + clc::NACC));
  double* d = buf.p;
  CLC_HIP(hipMemcpy(d, in, sizeof(double) * 64 * clc::NACC, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(clc::wave_reduce_test_kernel, dim3(1), dim3(64), 0, h->stream, d, d + 64 * clc::NACC,
                     reduce_mode);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(out, d + 64 * clc::NACC, sizeof(double) * clc::NACC, hipMemcpyDeviceToHost));
  return CLC_OK;
}

// Per-workgroup timeline of one compact-layout evaluation launch: stamps[grid*8] =
// {wall_start, wall_end (10 ns ticks), cycles prologue, loop, epilogue, 0, 0, 0}; returns grid.
int clc_debug_eval_timeline(clc_handle* h, const double pose[7], double lf, long long* stamps, int cap_waves,
                            int block_threads) {
  if (!h || !pose || !stamps || (block_threads != 256 && block_threads != 512))
    return fail(CLC_ERR_INVALID_ARG, "clc_debug_eval_timeline: bad argument");
#ifndef CLC_LEGACY_PATHS
  return fail(CLC_ERR_INVALID_ARG, "clc_debug_eval_timeline: not in this build (-DCLC_LEGACY_PATHS)");
#else
  if (!h->compact_ok) return fail(CLC_ERR_NO_DATA, "clc_debug_eval_timeline: needs the compact layout");
  CLC_HIP(hipSetDevice(h->device));
  const int grid = eval_grid(h, h->n_obs);
  const int n_waves = grid * (block_threads / 64);
  if (n_waves > cap_waves) return fail(CLC_ERR_INVALID_ARG, "clc_debug_eval_timeline: stamps buffer too small");
  int rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  DevBuf<long long> bs(&h->pool);
  CLC_HIP(bs.alloc((size_t)n_waves * 8));
  std::memcpy(h->h_small, pose, 7 * sizeof(double));
  CLC_HIP(hipMemcpyAsync(h->d_small, h->h_small, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  for (int rep = 0; rep < 4; ++rep) {  // the last launch is the one reported (warm)
    if (block_threads == 256)
      hipLaunchKernelGGL((clc::eval_timeline_kernel<256>), dim3(grid), dim3(256), 0, h->stream, h->d_ctiles, h->d_groups,
                         (long long)h->n_obs, h->d_small, lf, h->d_partials, bs.p);
    else
      hipLaunchKernelGGL((clc::eval_timeline_kernel<512>), dim3(grid), dim3(512), 0, h->stream, h->d_ctiles, h->d_groups,
                         (long long)h->n_obs, h->d_small, lf, h->d_partials, bs.p);
  }
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(stamps, bs.p, sizeof(long long) * (size_t)n_waves * 8, hipMemcpyDeviceToHost));
  return n_waves;
#endif
}

// What this build of the library contains beyond the default: bit 0 = the legacy paths of clc_legacy.hpp
// (-DCLC_LEGACY_PATHS: flag 8 of clc_set_launch, clc_debug_eval_timeline), bit 1 = debug stamps (-DCLC_STAMPS).
int clc_debug_build_features(void) {
  int f = 0;
#ifdef CLC_LEGACY_PATHS
  f |= 1;
#endif
#ifdef CLC_STAMPS
  f |= 2;
#endif
  return f;
}

// Layout report: compact[0/1] + group counts for the single-problem array and the batch.
// Row-layout report: rows[0/1] + row counts for the single-problem array and the batch.
int clc_debug_rows(clc_handle* h, int* rows, long long* n_rows, int* brows, long long* bn_rows) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_rows: NULL handle");
  if (rows) *rows = h->rows_ok ? (h->rows_z ? 2 : 1) : 0;  // 2: the rows carry z
  if (n_rows) *n_rows = h->n_rows;
  if (brows) *brows = h->brows_ok ? (h->brows_z ? 2 : 1) : 0;
  if (bn_rows) *bn_rows = h->bn_rows;
  return CLC_OK;
}

// Test hook (not part of include/clc.h): the wave split table of the row layout for `grid` workgroups — grid * 8 + 1 row
// indices — and, per row, whether it starts a scan (first[n_rows], may be NULL).
int clc_debug_wave_split(clc_handle* h, int grid, int* split, int* first) {
  if (!h || grid < 1 || !split) return fail(CLC_ERR_INVALID_ARG, "clc_debug_wave_split: bad arguments");
  if (!h->rows_ok) return fail(CLC_ERR_NO_DATA, "clc_debug_wave_split: no row layout");
  CLC_HIP(hipSetDevice(h->device));
  h->split_grid = -1;
  ensure_wave_split(h, grid);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  const char* base = reinterpret_cast<const char*>(h->d_rdesc);
  CLC_HIP(hipMemcpy(split, base + ((size_t)h->n_rows + 1) * sizeof(clc::RowDesc), sizeof(int) * ((size_t)grid * 8 + 1), hipMemcpyDeviceToHost));
  if (first) {
    std::vector<clc::RowDesc> d((size_t)h->n_rows);
    CLC_HIP(hipMemcpy(d.data(), base, sizeof(clc::RowDesc) * (size_t)h->n_rows, hipMemcpyDeviceToHost));
    for (long long r = 0; r < h->n_rows; ++r) first[r] = d[(size_t)r].first;
  }
  return CLC_OK;
}

// Resident-layout report of the batch: built[0/1], lanes per problem, largest points-per-lane, j-rows in all.
int clc_debug_resident(clc_handle* h, int* ok, int* lanes, int* max_ppl, long long* rows) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_resident: NULL handle");
  if (ok) *ok = h->bres.ok ? 1 : 0;
  if (lanes) *lanes = h->bres.lanes;
  if (max_ppl) *max_ppl = h->bres.max_ppl;
  if (rows) *rows = h->bres.rows;
  return CLC_OK;
}

// The same for the single-problem array (built for problems one workgroup can hold; clc_solve then runs in one launch).
int clc_debug_resident_single(clc_handle* h, int* ok, int* lanes, int* max_ppl) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_resident_single: NULL handle");
  if (ok) *ok = h->sres.ok ? 1 : 0;
  if (lanes) *lanes = h->sres.lanes;
  if (max_ppl) *max_ppl = h->sres.max_ppl;
  return CLC_OK;
}

// The cooperative whole-GPU solve (clc_coop.hpp): layout built, largest points per lane, solves run on it, launches that timed out,
// disabled on this handle.
extern "C" int clc_debug_coop(clc_handle* h, int* ok, int* max_ppl, long long* solves, int* aborts, int* disabled) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_coop: NULL handle");
  if (ok) *ok = h->cres.ok ? 1 : 0;
  if (max_ppl) *max_ppl = h->cres.max_ppl;
  if (solves) *solves = h->coop_solves;
  if (aborts) *aborts = h->coop_aborts;
  if (disabled) *disabled = (h->coop_eligible < h->coop_retry_at || h->coop_checked < 0) ? 1 : 0;  // resting after a time-out / device too small
  return CLC_OK;
}

// Test hook for the safety net of the cooperative solve: drop_next > 0 launches the NEXT cooperative solve that many workgroups short
// (the exchange of the others must time out, nothing is written, clc_solve falls back to the step chain and disables the path);
// reenable != 0 clears the disabled state again.
extern "C" int clc_debug_coop_control(clc_handle* h, int drop_next, int reenable) {
  if (!h || drop_next < 0 || drop_next >= clc::COOP_WGS) return fail(CLC_ERR_INVALID_ARG, "clc_debug_coop_control: bad argument");
  h->coop_test_drop = drop_next;
  if (reenable) {
    h->coop_retry_at = 0;
    h->coop_backoff = kCoopBackoff0;
  }
  return CLC_OK;
}

// Test hook: the next cooperative solve starts its pass tags here (to exercise the wrap of the 32-bit tags).
extern "C" int clc_debug_coop_set_tag(clc_handle* h, unsigned int tag) {
  if (!h || tag == 0) return fail(CLC_ERR_INVALID_ARG, "clc_debug_coop_set_tag: bad argument");
  h->coop_tag = tag;
  return CLC_OK;
}

int clc_debug_layout(clc_handle* h, int* compact, long long* n_groups, int* bcompact, long long* bn_groups) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_layout: NULL handle");
  if (compact) *compact = h->compact_ok ? 1 : 0;
  if (n_groups) *n_groups = h->n_groups;
  if (bcompact) *bcompact = h->bcompact_ok ? 1 : 0;
  if (bn_groups) *bn_groups = h->bn_groups;
  return CLC_OK;
}

// Shader-clock stamps of the last lm_kernel launch: [0] kernel entry, [1] after state load +
// partial reduction, [2] after the LM controller, [3] after publishing to the host mailbox;
// [4] row loads issued, [5] rows landed and summed per thread, [6] row groups combined (all need opt.profile_events).
int clc_debug_lm_profile(clc_handle* h, long long out[8]) {
  if (!h || !out) return fail(CLC_ERR_INVALID_ARG, "clc_debug_lm_profile: bad argument");
  CLC_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 8; ++i) out[i] = h->h_mailbox->prof[i];
  return CLC_OK;
}

// Times `reps` back-to-back launches of batched_eval_kernel over ALL uploaded problems at the poses given
// (poses[P*7]; every problem active, as in the first LM iteration of a batch) with HIP events on the handle's stream.
int clc_time_batched_eval(clc_handle* h, const double* poses, int reps, double* avg_ms) {
  if (!h || !poses || !avg_ms || reps < 1) return fail(CLC_ERR_INVALID_ARG, "clc_time_batched_eval: bad argument");
  if (!h->d_btiles || h->n_problems == 0) return fail(CLC_ERR_NO_DATA, "clc_time_batched_eval: no problems uploaded");
  CLC_HIP(hipSetDevice(h->device));
  int rc = ensure_events(h, 2);
  if (rc != CLC_OK) return rc;
  clc_options opt;
  clc_options_default(&opt);
  BatchedLaunch bl;
  rc = batched_launch_setup(h, opt, &bl);
  if (rc != CLC_OK) return rc;
  const size_t P = h->n_problems;
  std::memcpy(h->h_poses, poses, sizeof(double) * 7 * P);
  hipLaunchKernelGGL(clc::batched_init_kernel, dim3(bl.lm_blocks), dim3(bl.lm_threads), 0, h->stream, h->d_states, opt,
                     h->d_poses, (int)P, h->d_queue, h->d_ticket);
  for (int w = 0; w < 2; ++w) launch_batched_eval(h, opt, bl);
  CLC_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; ++r) launch_batched_eval(h, opt, bl);
  CLC_HIP(hipEventRecord(h->ev[1], h->stream));
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  CLC_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *avg_ms = (double)ms / reps;
  return CLC_OK;
}

// Times `reps` back-to-back launches of the evaluation kernel (K1 only) with HIP events on
// the handle's stream; *avg_ms = mean kernel-to-kernel period.
int clc_time_eval(clc_handle* h, const double pose[7], int with_loss, double lf, int with_jac, int reps,
                  double* avg_ms) {
  if (!h || !pose || !avg_ms || reps < 1) return fail(CLC_ERR_INVALID_ARG, "clc_time_eval: bad argument");
  if (!h->d_tiles) return fail(CLC_ERR_NO_DATA, "clc_time_eval: no observations uploaded");
  CLC_HIP(hipSetDevice(h->device));
  const int grid = eval_grid(h, h->n_obs);
  int rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  rc = ensure_events(h, 2);
  if (rc != CLC_OK) return rc;
  std::memcpy(h->h_small, pose, 7 * sizeof(double));
  CLC_HIP(hipMemcpyAsync(h->d_small, h->h_small, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  for (int w = 0; w < 3; ++w) {
    if (with_jac) launch_eval<true>(h, grid, with_loss != 0, h->d_small, nullptr, lf);
    else launch_eval<false>(h, grid, with_loss != 0, h->d_small, nullptr, lf);
  }
  CLC_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; ++r) {
    if (with_jac) launch_eval<true>(h, grid, with_loss != 0, h->d_small, nullptr, lf);
    else launch_eval<false>(h, grid, with_loss != 0, h->d_small, nullptr, lf);
  }
  CLC_HIP(hipEventRecord(h->ev[1], h->stream));
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  CLC_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *avg_ms = (double)ms / reps;
  return CLC_OK;
}

}  // extern "C"

#ifdef CLC_STAMPS
// Debug build only: copy the step-kernel stamp buffer out and clear it.
extern "C" int clc_debug_res_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_res_stamp_buf)) bytes = sizeof(clc::clc_res_stamp_buf);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_res_stamp_buf), bytes) != hipSuccess) return CLC_ERR_HIP;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(clc::clc_res_stamp_buf)) != hipSuccess) return CLC_ERR_HIP;
  return hipMemset(p, 0, sizeof(clc::clc_res_stamp_buf)) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
extern "C" int clc_debug_coop_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_coop_stamp_buf)) bytes = sizeof(clc::clc_coop_stamp_buf);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_coop_stamp_buf), bytes) != hipSuccess) return CLC_ERR_HIP;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(clc::clc_coop_stamp_buf)) != hipSuccess) return CLC_ERR_HIP;
  return hipMemset(p, 0, sizeof(clc::clc_coop_stamp_buf)) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
extern "C" int clc_debug_lmregs_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_lmu_ck)) bytes = sizeof(clc::clc_lmu_ck);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_lmu_ck), bytes) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
extern "C" int clc_debug_res_ctrl_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_res_stamp_ctrl)) bytes = sizeof(clc::clc_res_stamp_ctrl);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_res_stamp_ctrl), bytes) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
extern "C" int clc_debug_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_stamp_buf)) bytes = sizeof(clc::clc_stamp_buf);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_stamp_buf), bytes) != hipSuccess) return CLC_ERR_HIP;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(clc::clc_stamp_buf)) != hipSuccess) return CLC_ERR_HIP;
  return hipMemset(p, 0, sizeof(clc::clc_stamp_buf)) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
#endif
