// abi_core.hip — handle lifecycle, options, launch configuration and the small helpers every unit uses.
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

namespace clc_abi {

thread_local std::string g_last_error;

int fail(int code, const char* what, hipError_t e) {
  char buf[512];
  if (e != hipSuccess)
    std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else
    std::snprintf(buf, sizeof(buf), "%s", what);
  g_last_error = buf;
  return code;
}

// experiments: CLC_LAUNCH_AHEAD overrides the default launch-ahead depth (options with launch_ahead = 0)
int default_lookahead() {
  static const int v = [] {
    const char* e = std::getenv("CLC_LAUNCH_AHEAD");
    const int n = e ? std::atoi(e) : 0;
    return n > 0 ? n : kDefaultLookahead;
  }();
  return v;
}

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
bool rows_nontemporal(const clc_handle* h, long long n_rows, bool z) {
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

int ensure_bytes(double** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap && *p) return CLC_OK;
  if (*p) CLC_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  CLC_HIP(hipMalloc(p, bytes));
  *cap = bytes;
  return CLC_OK;
}

}  // namespace clc_abi

namespace {
__global__ void first_launch_kernel(unsigned int* p) {
  if (p == nullptr) return;
  p[0] = 0u;
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

static int create_init(clc_handle* h);

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
  // (everything below can fail half-way — a busy or full device: the handle and what it already owns are released then)
  const int rc = create_init(h);
  if (rc != CLC_OK) { clc_destroy(h); return rc; }
  *out = h;
  return CLC_OK;
}

// The allocations, the stream and the warm-up of a new handle; on any error the caller (clc_create) destroys the handle.
static int create_init(clc_handle* h) {
  if (const char* e = std::getenv("CLC_AUTO_PATHS_DISABLE")) {
    const int m = std::atoi(e);
    if (m >= 0 && m <= 11 && (m & 4) == 0) h->auto_disable = m;
  }
  if (const char* e = std::getenv("CLC_SMALL_ON_COOP")) h->small_on_coop = std::atoi(e) != 0;
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
  // The code objects of the units that hold kernels are loaded here, not by the first call that launches one of their kernels: the
  // reference's programs call the path once per process, and that one call then took 20 ms instead of 1 (CLC_LAZY_MODULES=1: leave
  // it to the first launches — a process that creates handles it may never use).
  if (!std::getenv("CLC_LAZY_MODULES")) {
    warm_layouts();
    warm_solve();
    warm_frontend();
    // (the batched unit's code object — the twelve resident instantiations, a quarter of the library — is loaded by the first
    // clc_upload_batched instead: the reference's one calibration per process never launches a batched kernel)
    // ... and what else only the FIRST launch / copy / allocation of a process pays: the stream's hardware queue (first launch), the
    // runtime's staging buffers for pageable copies (first hipMemcpy each way), the buffers every solve uses
    hipLaunchKernelGGL(first_launch_kernel, dim3(1), dim3(64), 0, h->stream, h->d_queue);
    double tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CLC_HIP(hipMemcpyAsync(h->d_small, tmp, sizeof(tmp), hipMemcpyHostToDevice, h->stream));
    CLC_HIP(hipMemcpyAsync(tmp, h->d_small, sizeof(tmp), hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    CLC_HIP(hipMemcpy(h->d_small, tmp, sizeof(tmp), hipMemcpyHostToDevice));
    CLC_HIP(hipMemcpy(tmp, h->d_small, sizeof(tmp), hipMemcpyDeviceToHost));
    {  // (copies of hundreds of kilobytes from pageable memory take another path than tiny ones: 7 ms the first time)
      std::vector<double> big((size_t)1 << 17, 0.0);
      DevBuf<double> dbig(&h->pool);
      CLC_HIP(dbig.alloc(big.size()));
      CLC_HIP(hipMemcpy(dbig.p, big.data(), big.size() * sizeof(double), hipMemcpyHostToDevice));
      CLC_HIP(hipMemcpy(big.data(), dbig.p, big.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    int rc = ensure_partials(h, h->num_cus);
    if (rc == CLC_OK) rc = ensure_trace(h, 128);
    if (rc == CLC_OK) rc = ensure_events(h, 2);
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipStreamSynchronize(h->stream));
  }
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
                  h->d_rxy, h->d_rdesc, h->d_brxy, h->d_brdesc, h->bres.d_xy, h->bres.d_desc, h->bres.d_row, h->sres.d_xy, h->sres.d_desc, h->sres.d_row, h->cres.d_xy, h->cres.d_desc, h->cres.d_row, h->cres.d_z, h->sres.d_z, h->bres.d_z, h->d_board, h->d_prob_row, h->d_sq, h->d_st, h->d_spts, h->d_sptl, h->d_soff};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->pool.clear();
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->h_mailbox) (void)hipHostFree(h->h_mailbox);
  if (h->h_spose) (void)hipHostFree(h->h_spose);
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  if (h->d_stage) (void)hipFree(h->d_stage);
  if (h->ev_stage) (void)hipEventDestroy(h->ev_stage);
  if (h->d_small_aos) (void)hipFree(h->d_small_aos);
  if (h->h_ms_poses) (void)hipHostFree(h->h_ms_poses);
  if (h->h_ms_summaries) (void)hipHostFree(h->h_ms_summaries);
  if (h->d_ms_results) (void)hipFree(h->d_ms_results);
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
  if (!h || disable_mask < 0 || (disable_mask & ~(1 | 2 | 8)) != 0) return fail(CLC_ERR_INVALID_ARG, "clc_set_auto_paths: the mask is a sum of 1, 2, 8");
  h->auto_disable = disable_mask;
  return CLC_OK;
}

int clc_set_small_on_coop(clc_handle* h, int enable) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_set_small_on_coop: NULL handle");
  h->small_on_coop = enable != 0;
  return CLC_OK;
}

int clc_get_path_info(const clc_handle* h, clc_path_info* out) {
  if (!h || !out) return fail(CLC_ERR_INVALID_ARG, "clc_get_path_info: bad argument");
  std::memset(out, 0, sizeof(*out));
  out->single_resident = h->sres.ok ? 1 : 0;
  out->single_lanes = h->sres.ok ? h->sres.lanes : 0;
  out->single_points_per_lane = h->sres.ok ? h->sres.max_ppl : 0;
  out->coop_resident = h->cres.ok ? 1 : 0;
  out->coop_points_per_lane = h->cres.ok ? h->cres.max_ppl : 0;
  out->coop_points_carry_z = h->cres.ok && h->cres.with_z ? 1 : 0;
  out->coop_workgroups = h->cres.ok ? h->cres.wgs : 0;
  out->coop_resting = h->coop_eligible < h->coop_retry_at ? 1 : 0;
  out->coop_timeouts = h->coop_aborts;
  out->batched_resident = h->bres.ok ? 1 : 0;
  out->batched_lanes = h->bres.ok ? h->bres.lanes : 0;
  out->batched_points_per_lane = h->bres.ok ? h->bres.max_ppl : 0;
  out->batched_points_carry_z = h->bres.ok && h->bres.with_z ? 1 : 0;
  out->rows_layout = h->rows_ok ? (h->rows_z ? 2 : 1) : 0;
  out->batched_rows_layout = h->brows_ok ? (h->brows_z ? 2 : 1) : 0;
  out->coop_solves = h->coop_solves;
  out->batched_lane_rows = h->bres.ok ? h->bres.rows : 0;
  out->n_rows = h->n_rows;
  out->batched_n_rows = h->bn_rows;
  out->coop_gate_waits_expired = h->coop_gate_waits_expired;
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


}  // extern "C"
