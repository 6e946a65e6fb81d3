// abi_comm.hip — multi-GPU: RCCL all-gather of the sharded batch's result records.
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

#include <dlfcn.h>
#include <link.h>

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

int rccl_fail(const char* what, ncclResult_t r);
// The all-gather of a communicator.  A communicator WITHOUT an RCCL handle exists only in the -DCLC_TEST_HOOKS build
// (clc_debug_comm_create_layout: "rank r of n" on the one visible GPU, which RCCL refuses to give two ranks): its all-gather moves this
// rank's segment into place and leaves the other ranks' segments as they are — what the buffer arithmetic of the gather calls
// (segment offsets, padding, which parts are copied to the host) can be tested against at rank > 0.
ncclResult_t comm_all_gather(ncclComm_t comm, int rank, const double* send, double* recv, size_t count, hipStream_t stream) {
  if (comm) return rccl().AllGather(send, recv, count, ncclDouble, comm, stream);
  double* mine = recv + (size_t)rank * count;
  if (send != mine && hipMemcpyAsync(mine, send, count * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
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
  double* d_recv = nullptr;  // = d_base + 12: the gathered records
  double* h_recv = nullptr;  // = h_base + 12 (pinned)
  double* d_base = nullptr;  // the allocations: ONE record in front of the gathered array holds the running totals of
  double* h_base = nullptr;  // clc_solve_batched_gather.  h_base is mapped: the kernel writes this rank's records and the totals there itself
  double* hd_base = nullptr; // device address of h_base
  size_t cap = 0;            // records per rank the buffers hold
  // clc_solve_batched_gather: the kernel writes this rank's records into ITS segment of d_recv (the in-place form of the all-gather);
  // the record in front of d_recv / h_recv: 4 running totals (clc_batch_stats' counters, never reset)
  unsigned long long stats_seen[4] = {0, 0, 0, 0};
  long long pad_from = -1, pad_base = -1;  // own segment's padding records are in place for this many local problems
};

static __global__ void pad_records_kernel(double* __restrict__ seg, long long n_local, long long cap) {
  const long long k = n_local + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cap) return;
  double* o = seg + 12 * k;
  for (int i = 0; i < 11; ++i) o[i] = 0.0;
  o[11] = -1.0;
}

static int comm_ensure_buffers(clc_comm* c, size_t cap_per_rank) {
  if (cap_per_rank <= c->cap) return CLC_OK;
  if (c->d_send) CLC_HIP(hipFree(c->d_send));
  if (c->d_base) CLC_HIP(hipFree(c->d_base));
  if (c->h_base) CLC_HIP(hipHostFree(c->h_base));
  c->d_send = c->d_recv = c->h_recv = c->d_base = c->h_base = c->hd_base = nullptr;
  c->cap = 0;
  c->pad_from = c->pad_base = -1;
  const size_t n_rec = cap_per_rank * (size_t)c->world + 1;  // the totals' record + the gathered array
  CLC_HIP(hipMalloc(&c->d_send, sizeof(clc_result_record) * cap_per_rank));
  CLC_HIP(hipMalloc(&c->d_base, sizeof(clc_result_record) * n_rec));
  CLC_HIP(hipMemset(c->d_base, 0, sizeof(clc_result_record) * n_rec));
  CLC_HIP(hipHostMalloc(&c->h_base, sizeof(clc_result_record) * n_rec, hipHostMallocMapped));
  CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->hd_base), c->h_base, 0));
  std::memset(c->h_base, 0, sizeof(clc_result_record));
  c->d_recv = c->d_base + 12;
  c->h_recv = c->h_base + 12;
  std::memset(c->stats_seen, 0, sizeof(c->stats_seen));
  c->cap = cap_per_rank;
  return CLC_OK;
}

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
  if (c->d_base) (void)hipFree(c->d_base);
  if (c->h_base) (void)hipHostFree(c->h_base);
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
  {
    const int rc = comm_ensure_buffers(c, cap_per_rank);
    if (rc != CLC_OK) return rc;
  }
  c->pad_from = c->pad_base = -1;  // (this all-gather rewrites the own segment of d_recv: the fused form pads it again)
  const int threads = 256;
  hipLaunchKernelGGL(clc::pack_results_kernel, dim3((unsigned)((cap_per_rank + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, h->d_results, (long long)n_local, (long long)cap_per_rank, (double)first_global_index,
                     c->d_send);
  CLC_HIP(hipGetLastError());
  const size_t count = cap_per_rank * (sizeof(clc_result_record) / sizeof(double));
  ncclResult_t r = comm_all_gather(c->comm, c->rank, c->d_send, c->d_recv, count, h->stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  const size_t bytes = sizeof(clc_result_record) * cap_per_rank * (size_t)c->world;
  CLC_HIP(hipMemcpyAsync(c->h_recv, c->d_recv, bytes, hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  if (all_records) std::memcpy(all_records, c->h_recv, bytes);
  if (local_rc != CLC_OK) return fail(local_rc, local_msg);
  return CLC_OK;
}

int clc_solve_batched_gather(clc_comm* c, const clc_options* opt_in, const double* poses0, int64_t first_global_index,
                             size_t cap_per_rank, clc_result_record* all_records, clc_batch_stats* stats) {
  if (!c || cap_per_rank == 0 || first_global_index < 0) return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather: bad argument");
  clc_handle* h = c->h;
  const size_t P = h->n_problems;
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_options_default(&opt);
  // This is a collective (see clc_gather_results): a rank with a LOCAL problem still enters the all-gather, with padding only, and
  // reports afterwards.
  int local_rc = CLC_OK;
  std::string local_msg;
  auto local_fail = [&](int rc, const char* msg) { if (local_rc == CLC_OK) { local_rc = rc; local_msg = msg; } };
  if (P > cap_per_rank) local_fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather: cap_per_rank < local problems (this rank contributed padding only)");
  if (P > 0 && !poses0) local_fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather: NULL start poses (this rank contributed padding only)");
  if (P > 0 && local_rc == CLC_OK && (!h->d_btiles || !h->h_poses)) local_fail(CLC_ERR_NO_DATA, "clc_solve_batched_gather: no problems uploaded");
  if (P > 0 && local_rc == CLC_OK) {
    // options as clc_solve_batched checks them; the start poses go into the handle's pinned buffer (what the kernel reads) and are
    // checked for non-finite values in the same pass over them.  (The previous step ended with a stream synchronisation: nothing
    // still reads that buffer.)
    int rc = batched_check_inputs("clc_solve_batched_gather", opt, poses0, 0);
    if (rc == CLC_OK) {
      unsigned long long bad = 0;
      double* dst = h->h_poses;
      const bool copy = poses0 != dst;
      for (size_t i = 0; i < 7 * P; ++i) {
        const double v = poses0[i];
        unsigned long long b;
        std::memcpy(&b, &v, sizeof(b));
        bad |= (unsigned long long)(((b >> 52) & 0x7FFull) == 0x7FFull);
        if (copy) dst[i] = v;
      }
      if (bad) rc = fail(CLC_ERR_NONFINITE, "clc_solve_batched_gather: non-finite initial pose");
    }
    if (rc != CLC_OK) local_fail(rc, clc_last_error());
  }
  CLC_HIP(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  BatchedLaunch bl;
  if (P > 0 && local_rc == CLC_OK) {
    const int rc = batched_launch_setup(h, opt, &bl);
    if (rc != CLC_OK) local_fail(rc, clc_last_error());
  }
  if (P > 0 && local_rc == CLC_OK && !bl.resident) {
    // The batch does not run as the one-launch on-chip solve (a problem too large for a workgroup, points with z, explicit flags): the
    // two-call form — same records, poses and summaries cross PCIe as well.
    int rc = clc_solve_batched(h, &opt, h->h_poses, h->h_summaries);  // (the start poses are in the pinned buffer already)
    if (rc != CLC_OK) { local_fail(rc, clc_last_error()); h->results_valid = 0; }
    const int rc2 = clc_gather_results(c, first_global_index, cap_per_rank, all_records);
    if (stats && rc == CLC_OK) {
      std::memset(stats, 0, sizeof(*stats));
      for (size_t k = 0; k < P; ++k) {
        stats->evaluations += h->h_summaries[k].num_evaluations;
        stats->iterations += h->h_summaries[k].num_iterations;
        stats->not_converged += h->h_summaries[k].termination == CLC_NO_CONVERGENCE || h->h_summaries[k].termination == CLC_FAILURE;
      }
      stats->problems = (int64_t)P;
      stats->fused = 0;
      stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (local_rc != CLC_OK) return fail(local_rc, local_msg.c_str());
    return rc2;
  }
  {
    const int rc = comm_ensure_buffers(c, cap_per_rank);
    if (rc != CLC_OK) return rc;
  }
  const size_t n_local = local_rc == CLC_OK ? P : 0;
  const long long seg_off = 12ll * (long long)c->rank * (long long)cap_per_rank;  // this rank's segment of the gathered array (doubles)
  double* seg = c->d_recv + seg_off;
  if (c->pad_from != (long long)n_local || c->pad_base != (long long)cap_per_rank) {  // (first call, or the shape changed)
    const long long n_pad = (long long)cap_per_rank - (long long)n_local;
    if (n_pad > 0) {  // the padding records of this rank's segment: device (what the all-gather sends) and host
      hipLaunchKernelGGL(pad_records_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, h->stream, seg, (long long)n_local, (long long)cap_per_rank);
      CLC_HIP(hipGetLastError());
      for (size_t k = n_local; k < cap_per_rank; ++k) {
        double* o = c->h_recv + seg_off + 12 * k;
        for (int i = 0; i < 11; ++i) o[i] = 0.0;
        o[11] = -1.0;
      }
    }
    c->pad_from = (long long)n_local;
    c->pad_base = (long long)cap_per_rank;
  }
  const bool timed = opt.profile_events == 1;
  if (timed) {
    const int rc = ensure_events(h, 2);
    if (rc != CLC_OK) return rc;
  }
  if (n_local > 0) {
    if (timed) CLC_HIP(hipEventRecord(h->ev[0], h->stream));
    // the last workgroup to finish (totals' arrival count == goal) copies the totals to the host twin
    launch_resident_batch(h, opt, bl, nullptr, c->d_base, (double)first_global_index, c->hd_base, seg_off, c->stats_seen[3] + (unsigned long long)P);
    CLC_HIP(hipGetLastError());
    if (timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
  }
  // The all-gather in place (send buffer = this rank's segment of the receive buffer).  This rank's own records and the totals are in
  // host memory already — the kernel wrote them there —, so only the OTHER ranks' segments are copied down (world size 1: nothing);
  // one synchronisation.
  const size_t count = cap_per_rank * (sizeof(clc_result_record) / sizeof(double));
  ncclResult_t r = comm_all_gather(c->comm, c->rank, seg, c->d_recv, count, h->stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  const size_t seg_bytes = sizeof(clc_result_record) * cap_per_rank;
  const size_t bytes = seg_bytes * (size_t)c->world;
  if (c->rank > 0) CLC_HIP(hipMemcpyAsync(c->h_recv, c->d_recv, seg_bytes * (size_t)c->rank, hipMemcpyDeviceToHost, h->stream));
  if (c->rank + 1 < c->world)
    CLC_HIP(hipMemcpyAsync(c->h_recv + seg_off + 12 * cap_per_rank, c->d_recv + seg_off + 12 * cap_per_rank,
                           seg_bytes * (size_t)(c->world - 1 - c->rank), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));  // (kernel completion makes what it wrote over PCIe visible)
  h->results_valid = 0;  // (the handle's own result buffer was not written: a later clc_gather_results has nothing to send)
  if (all_records) std::memcpy(all_records, c->h_recv, bytes);
  unsigned long long now[4];
  std::memcpy(now, c->h_base, sizeof(now));
  if (n_local == 0) std::memcpy(now, c->stats_seen, sizeof(now));  // (no launch: the totals stand)
  if (n_local > 0 && now[3] != c->stats_seen[3] + (unsigned long long)P) {
    // the last workgroup publishes the totals when the arrival count reaches stats_seen[3] + P: anything else means an earlier call
    // left the count and the host's copy of it apart (it returned on an error between its launch and this bookkeeping) — resynchronise
    // from the device and report this call's totals as unknown rather than as a difference of unrelated numbers
    unsigned long long dev[4] = {0, 0, 0, 0};
    CLC_HIP(hipMemcpy(dev, c->d_base, sizeof(dev), hipMemcpyDeviceToHost));
    std::memcpy(c->stats_seen, dev, sizeof(dev));
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->problems = -1; stats->evaluations = -1; stats->iterations = -1; stats->not_converged = -1; stats->fused = 1; }
    if (local_rc != CLC_OK) return fail(local_rc, local_msg.c_str());
    return CLC_OK;
  }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->evaluations = (int64_t)(now[0] - c->stats_seen[0]);
    stats->iterations = (int64_t)(now[1] - c->stats_seen[1]);
    stats->not_converged = (int64_t)(now[2] - c->stats_seen[2]);
    stats->problems = (int64_t)(now[3] - c->stats_seen[3]);
    stats->fused = 1;
    if (timed && n_local > 0) {
      float ms = 0.f;
      CLC_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
      stats->kernel_ms = (double)ms;
    }
    stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  for (int i = 0; i < 4; ++i) c->stats_seen[i] = now[i];
  if (local_rc != CLC_OK) return fail(local_rc, local_msg.c_str());
  return CLC_OK;
}

#ifdef CLC_TEST_HOOKS
// Test hook (hooks build only): a communicator laid out as rank `rank` of `world` WITHOUT RCCL behind it (see comm_all_gather).
#pragma GCC visibility push(default)
int clc_debug_comm_create_layout(clc_comm** out, clc_handle* h, int rank, int world);
#pragma GCC visibility pop
int clc_debug_comm_create_layout(clc_comm** out, clc_handle* h, int rank, int world) {
  if (!out || !h || world < 1 || rank < 0 || rank >= world) return fail(CLC_ERR_INVALID_ARG, "clc_debug_comm_create_layout: bad argument");
  clc_comm* c = new clc_comm();
  c->h = h;
  c->rank = rank;
  c->world = world;
  *out = c;
  return CLC_OK;
}
#endif

const clc_result_record* clc_comm_records(const clc_comm* c) {
  return c ? reinterpret_cast<const clc_result_record*>(c->h_recv) : nullptr;
}

}  // extern "C"
