// clc_abi_internal.hpp — what the translation units of the C-ABI (abi_*.hip) share: the handle, the per-call device
// pool, error reporting and the helpers that cross unit boundaries.  Not installed; include/clc.h is the interface.
//   abi_core.hip      handle lifecycle, options, launch configuration, small shared helpers
//   abi_layouts.hip   upload paths: re-encoding of the records into the compact / row / lane layouts, stored scans
//   abi_solve.hip     clc_eval, clc_solve and its launch sequences (step chain, single-workgroup resident, cooperative)
//   abi_frontend.hip  factor evaluation, manifold plus, information matrix, closed form, line fitting, scan conversion
//   abi_batched.hip   clc_solve_batched
//   abi_comm.hip      RCCL gather of the sharded batch's result records
//   abi_debug.hip     clc_debug_* / clc_time_* (test and profiling hooks; only with -DCLC_TEST_HOOKS)
// Built for gfx950 only (camlasercalibratool_amd/_build.py): every unit with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=8 -c, linked with hipcc -shared.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/clc.h"
#pragma GCC visibility pop
#include "clc_host.hpp"
#include "clc_kernels.hpp"
#include "clc_resident.hpp"
#include "clc_coop.hpp"

namespace clc_abi {

extern thread_local std::string g_last_error;
int fail(int code, const char* what, hipError_t e = hipSuccess);  // abi_core.hip

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

inline bool all_finite(const double* p, int n) {
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(p[i])) return false;
  return true;
}

constexpr int kDefaultLookahead = 2;
int default_lookahead();  // abi_core.hip
constexpr size_t kInfinityCacheBytes = 256u << 20;  // MI355X memory-side cache (MI355X_MICROARCH.md)
constexpr int kDefaultLaunchFlags = 2 | 16 | 32 | 128 | 256 | 512;  // prefetch + compact layout + 512-thread weighted workgroups + step kernel (clc::FLAG_*), tuned on MI355X (scripts/tune_eval.py, scripts/step_check.py)
constexpr int kDefaultBlocksPerCU = 1;   // 4 waves per CU with 2 tiles in flight each     // single-problem solver: launch-ahead depth
constexpr int kSmallDoubles = 512;  // device + pinned scratch for small transfers


// Lane layout of clc_resident.hpp: j-major point rows, lane descriptors, row offsets per problem.
struct ResLayout {
  double* d_xy = nullptr;
  size_t xy_cap = 0;
  double* d_desc = nullptr;  // clc::ResLane [P * lanes]
  size_t desc_cap = 0;
  double* d_row = nullptr;   // unsigned int [P + 1]
  size_t row_cap = 0;
  double* d_z = nullptr;     // with_z: the slots' z, j-major like d_xy, 8 bytes per slot
  size_t z_cap = 0;
  bool with_z = false;       // some record has p.z != 0: 24-byte slots (cooperative layout; batched layout: the 512-lane z form)
  int wgs = 0;               // cooperative layout only: the workgroups the problem is dealt to (COOP_WGS, or COOP_SMALL_WGS: one-hop form)
  int lanes = 0;             // lanes per problem of the built layout (256 / 512)
  int max_ppl = 0;           // largest points-per-lane over the problems
  int uni_ppl = -1;          // >= 0: every problem has this many points per lane
  long long rows = 0;        // j-rows in all
  bool ok = false;
};

constexpr long long kCoopBackoff0 = 16;  // solves the cooperative path rests after its first abort (doubles with every further one)

}  // namespace clc_abi

struct clc_handle {
  clc_abi::DevPool pool;  // temporaries of the entry points (DevBuf)
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
  // host-side knowledge of reference-size stored scans (<= 16 384 points, <= 4 096 poses): what lets clc_select_observations plan the
  // layouts of the selection on the host and enqueue their construction without a single read-back (abi_layouts.hip, small_fast_upload)
  bool store_small = false;
  std::vector<double> s_tag_q, s_tag_t;          // tag poses (w, x, y, z) / t of the stored scans
  bool s_any_z_pts = false, s_any_z_ptl = false, s_any_z_ends = false;  // some p.z != 0 in points / points_on_line / the first + last point of a scan
  // staging of the small-problem upload path: ONE pinned block + its device twin per handle; `ev_stage` marks the last copy out of it
  char* h_stage = nullptr; char* d_stage = nullptr; size_t stage_cap = 0;
  hipEvent_t ev_stage = nullptr; bool stage_busy = false;
  double* d_small_aos = nullptr; size_t small_aos_cap = 0;  // the records of a small problem (never a pool block: kernels may still read it after the call returned)
  bool fast_small = true;                        // (hooks build: clc_debug_fast_small switches the path off for the A/B tests)
  long long fast_small_uploads = 0;
  bool store_lines_equal_points = false;         // points_on_line is bit for bit points (reference-size inputs only: see clc_store_observations)
  long long selection_key = -1;                  // the (stored scans, selection) the observation array was built from by clc_select_observations; -1: none
  long long selection_cfg = -1;                  // ... under these upload-time settings (launch flags, auto paths)
  // launch geometry
  int grid_override = 0;
  int launch_flags = clc_abi::kDefaultLaunchFlags;
  int auto_disable = 0;     // clc_set_auto_paths: 1 no cooperative solve, 2 no single-workgroup resident solve, 8 (at upload) no one-hop form
  bool small_on_coop = false;    // clc_set_small_on_coop (at upload): problems one workgroup holds also get the cooperative layout
  bool single_uni_ctrl = false;  // hooks build (clc_debug_single_controller): the single-workgroup kernel runs the cooperative kernel's controller
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
  clc_abi::ResLayout bres, sres;
  // cooperative whole-GPU solve of one problem (clc_coop.hpp): the problem's lane layout in 256 chunks, the exchange boards, the
  // next free pass tag; disabled on the handle after a launch that timed out (the step chain takes over)
  clc_abi::ResLayout cres;
  clc::CoopBoard* d_board = nullptr;
  unsigned int coop_tag = 1;
  int coop_checked = 0;  // 0: co-residency not checked yet, 1: 256 workgroups fit the device, -1: they do not
  // after a launch that aborted the path rests for `coop_backoff` eligible solves (16, doubling with every further abort up to 2^20:
  // a GPU shared with long-running kernels of somebody else settles on the step chain; a one-off collision costs the first-pass
  // census timeout, 0.2 ms, once)
  long long coop_eligible = 0, coop_retry_at = 0, coop_backoff = clc_abi::kCoopBackoff0;
  int coop_aborts = 0;
  int coop_gate_waits_expired = 0;  // solves that took the step chain because another handle's cooperative launch held the device for > 5 ms
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
  // clc_solve_multistart: start poses / summaries (pinned, device-mapped) and result records of the starts
  double* h_ms_poses = nullptr; double* d_ms_poses = nullptr;
  clc_summary* h_ms_summaries = nullptr; clc_summary* d_ms_summaries = nullptr;
  double* d_ms_results = nullptr;
  size_t ms_cap = 0;
};

namespace clc_abi {

// ---- abi_core.hip ----
int eval_grid(const clc_handle* h, size_t n);
int ensure_partials(clc_handle* h, int blocks);
int ensure_trace(clc_handle* h, int cap);
int ensure_events(clc_handle* h, size_t n);
int ensure_bytes(double** p, size_t* cap, size_t bytes);
bool use_rows(const clc_handle* h);
bool use_brows(const clc_handle* h);
bool rows_nontemporal(const clc_handle* h, long long n_rows, bool z = false);
void ensure_wave_split(clc_handle* h, int grid);

// Every unit with kernels: load its code object on the current device now (hipFuncGetAttributes on a kernel of the unit), so that
// the first call after clc_create does not pay the lazy module load (~2-8 ms per unit).
void warm_layouts();
void warm_solve();
void warm_frontend();
void warm_batched();
inline void warm_kernel(const void* f) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, f) != hipSuccess) (void)hipGetLastError();  // (never fatal: the launch itself reports a real problem)
}

// ---- abi_layouts.hip ----
// Resident layout limits: what the instantiations of resident_solve_kernel hold per lane (registers + LDS).
// 256-lane form: 256-thread workgroups, two problems per CU; 512-lane form: one 512-thread workgroup per CU (problems with more
// than 256 scans, or flag 8192).  Both hold 512 x 22 = 256 x 44 - 512 points at most.
constexpr int kResPR256 = 23, kResPL256 = 19, kResPR512 = 4, kResPL512 = 18;
// batches whose points carry z (p.z != 0 in some record): the 512-lane form with 24-byte slots — 10 points per lane in registers
// (60 VGPRs) + 12 in LDS (512 x 12 x 24 B = 147.5 KB): 512 x 22 points, the same capacity as the (x, y) form
constexpr int kResPRz = 10, kResPLz = 12;
// controller of the batched launches (clc_resident.hpp CTRL): 4-wave form / 8-wave form
constexpr int kResCtrl4 = 0, kResCtrl8 = 0;

struct LayoutTargets {
  double** d_ct; size_t* ct_cap; double** d_gr; size_t* gr_cap; long long* n_groups; bool* compact_ok;
  double** d_rxy; size_t* rxy_cap; double** d_rdesc; size_t* rdesc_cap; long long* n_rows; bool* rows_ok;
  long long** d_prob_row;  // nullptr for the single-problem array
  ResLayout* res = nullptr;  // also build the on-chip resident ("lane") layout (clc_resident.hpp) into this
  bool* rows_z = nullptr;    // out: the rows carry z
  ResLayout* coop = nullptr; // single problem only: its lane layout in COOP_WGS chunks (clc_coop.hpp)
};
int build_layouts(clc_handle* h, const double* d_aos, size_t n_total, const std::vector<long long>& rec_off,
                  const std::vector<long long>& tile_off, const LayoutTargets& T);
// builds the records of the stored scans' selection on the device into *aos (allocated here)
int flatten_on_device(clc_handle* h, bool linefit, bool boundary, DevBuf<double>* aos, long long* n_out);

// ---- abi_solve.hip ----
// one launch of the evaluation kernel the handle's flags and layouts select (K1), partial rows into h->d_partials
void launch_eval(clc_handle* h, int grid, bool with_jac, bool with_loss, const double* d_pose, const int32_t* d_status, double lf,
                 const clc::Pose7* pose_arg = nullptr);
int solve_stepped(clc_handle* h, const clc_options& opt, int grid, double pose[7], clc_summary* summary, clc_iteration* trace,
                  int trace_cap, std::chrono::steady_clock::time_point t0, int ev_first = -1, int ev_last = -1, float* ev_ms = nullptr);

// ---- abi_batched.hip ----
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
int batched_launch_setup(clc_handle* h, const clc_options& opt, BatchedLaunch* bl);
void launch_batched_eval(clc_handle* h, const clc_options& opt, const BatchedLaunch& bl);
// ONE launch of resident_solve_kernel over the handle's batch (bl.resident) on the handle's stream; start poses from the handle's pinned
// buffer.  d_summaries != nullptr: outcomes into d_poses / d_summaries / d_results (clc_solve_batched).  d_summaries == nullptr: the
// records-only form (clc_solve_batched_gather) — d_results / rec_host = the communicator's gather buffer and its pinned host twin
// ([totals record][gathered array]), this rank's segment seg_off doubles into the array, global index rec_base + k, `goal` = the
// totals' arrival count at the end of this launch (batched_write_record, clc_kernels.hpp).
// multistart != nullptr (clc_solve_multistart): `n_starts` workgroups, all on problem 0's layout, start poses / outcomes in the given buffers.
struct MultiStartLaunch {
  size_t n_starts = 0;
  double* d_poses = nullptr;
};
void launch_resident_batch(clc_handle* h, const clc_options& opt, const BatchedLaunch& bl, clc_summary* d_summaries, double* d_results,
                           double rec_base, double* rec_host, long long seg_off, unsigned long long goal, const MultiStartLaunch* multistart = nullptr);
// the checks clc_solve_batched makes on its options and start poses (shared with clc_solve_batched_gather); CLC_OK or the error set
int batched_check_inputs(const char* who, const clc_options& opt, const double* poses, size_t n_problems);

}  // namespace clc_abi
