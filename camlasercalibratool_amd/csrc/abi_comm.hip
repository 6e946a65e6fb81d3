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
  decltype(&ncclGather) Gather = nullptr;              // optional (RCCL extension): the rooted form, clc_comm_set_root
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
    api.Gather = reinterpret_cast<decltype(api.Gather)>(dlsym(api.lib, "ncclGather"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.lib, "ncclCommUserRank"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
      api.lib = nullptr;
    }
  });
  return api;
}

int rccl_fail(const char* what, ncclResult_t r);
// The collective of a communicator: every rank contributes `count` doubles from `send` (= its own segment of `recv`: in place).
// root < 0: ncclAllGather — every rank receives every segment.  root >= 0: ncclGather (RCCL extension, rccl.h) — only `root` receives;
// with a librccl that lacks the symbol the all-gather runs instead (the records then arrive everywhere, the host copy stays at the root).
// A communicator WITHOUT an RCCL handle exists only in the -DCLC_TEST_HOOKS build (clc_debug_comm_create_layout: "rank r of n" on the one
// visible GPU, which RCCL refuses to give two ranks): its collective moves this rank's segment into place and leaves the other ranks'
// segments as they are — what the buffer arithmetic of the gather calls (segment offsets, padding, which parts are copied to the host
// at which rank) is tested against at rank > 0.
ncclResult_t comm_collect(ncclComm_t comm, int rank, int root, const double* send, double* recv, size_t count, hipStream_t stream, int* used_gather) {
  if (used_gather) *used_gather = 0;
  if (comm) {
    if (root >= 0 && rccl().Gather) {
      if (used_gather) *used_gather = 1;
      return rccl().Gather(send, recv, count, ncclDouble, root, comm, stream);
    }
    return rccl().AllGather(send, recv, count, ncclDouble, comm, stream);
  }
#ifdef CLC_TEST_HOOKS
  double* mine = recv + (size_t)rank * count;
  if (send != mine && hipMemcpyAsync(mine, send, count * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
#else
  (void)rank; (void)send; (void)recv; (void)count; (void)stream;
  return ncclInvalidArgument;  // (unreachable: the product library creates communicators through ncclCommInitRank only)
#endif
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
  int root = -1;             // clc_comm_set_root: >= 0 = only this rank receives the other ranks' records (and copies them to its host)
  double* d_base = nullptr;  // device: ONE record of running totals (clc_solve_batched_gather) in front of the gathered array
  double* d_recv = nullptr;  // = d_base + 12: the gathered records
  // Host side: TWO pinned, device-mapped twins of d_base.  The kernel of a step writes this rank's records and the totals into the
  // twin of that step itself; the other ranks' segments are copied down into the same twin.  The plain calls stay on twin `cur`; the
  // pipelined form alternates, so that step k's kernel never writes into the records of step k-1 the caller is still reading.
  double* h_base[2] = {nullptr, nullptr};
  double* hd_base[2] = {nullptr, nullptr};  // device addresses of h_base[]
  int cur = 0;               // twin of the last COMPLETED step (what clc_comm_records returns)
  size_t cap = 0;            // records per rank the buffers hold
  // the record in front of d_recv / h_recv: 4 running totals (clc_batch_stats' counters, never reset)
  unsigned long long stats_seen[4] = {0, 0, 0, 0};
  long long pad_from = -1, pad_base = -1;  // own segment's padding records (device + both twins) are in place for this many local problems
  // pipelined steps (clc_solve_batched_gather_pipelined): the copy of step k-1's records overlaps step k's kernel
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copy = nullptr;
  struct Flight {
    bool active = false;
    int twin = 0;
    size_t cap = 0, n_local = 0, P = 0;
    int local_rc = CLC_OK;
    std::string local_msg;
    bool fused = true, timed = false;
    std::chrono::steady_clock::time_point t0;
  } fl;
  // bookkeeping (clc_comm_get_info)
  long long collectives = 0, rooted_collectives = 0, host_copies = 0, host_copy_bytes = 0, pipelined_steps = 0;
  double* h_recv_of(int twin) const { return h_base[twin] + 12; }
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
  clc_handle* h = c->h;
  if (c->d_base) CLC_HIP(hipFree(c->d_base));
  for (int t = 0; t < 2; ++t) {
    if (c->h_base[t]) CLC_HIP(hipHostFree(c->h_base[t]));
    c->h_base[t] = c->hd_base[t] = nullptr;
  }
  c->d_recv = c->d_base = nullptr;
  c->cap = 0;
  c->cur = 0;
  c->pad_from = c->pad_base = -1;
  const size_t n_rec = cap_per_rank * (size_t)c->world + 1;  // the totals' record + the gathered array
  CLC_HIP(hipMalloc(&c->d_base, sizeof(clc_result_record) * n_rec));
  // (stream-ordered in front of the pad kernel and the solve launch: the handle's stream is non-blocking, a null-stream memset is not
  // ordered against it by the API)
  CLC_HIP(hipMemsetAsync(c->d_base, 0, sizeof(clc_result_record) * n_rec, h->stream));
  for (int t = 0; t < 2; ++t) {
    CLC_HIP(hipHostMalloc(&c->h_base[t], sizeof(clc_result_record) * n_rec, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->hd_base[t]), c->h_base[t], 0));
    std::memset(c->h_base[t], 0, sizeof(clc_result_record) * n_rec);
  }
  c->d_recv = c->d_base + 12;
  std::memset(c->stats_seen, 0, sizeof(c->stats_seen));
  c->cap = cap_per_rank;
  return CLC_OK;
}

// Which segments of the gathered array this rank brings to its host after the collective: all the OTHER ranks' (no root, or this rank
// is the root), or none.  (Its own segment is on the host already: the kernel wrote it / the two-call form copies it.)
static bool comm_copies_others(const clc_comm* c) { return c->root < 0 || c->rank == c->root; }

// Device -> host copies of the other ranks' segments into twin `tw`, on `stream`.
static int comm_copy_others(clc_comm* c, int tw, size_t cap_per_rank, hipStream_t stream) {
  const size_t seg_bytes = sizeof(clc_result_record) * cap_per_rank;
  const long long seg_off = 12ll * (long long)c->rank * (long long)cap_per_rank;
  double* hr = c->h_recv_of(tw);
  if (c->rank > 0) {
    CLC_HIP(hipMemcpyAsync(hr, c->d_recv, seg_bytes * (size_t)c->rank, hipMemcpyDeviceToHost, stream));
    ++c->host_copies; c->host_copy_bytes += (long long)(seg_bytes * (size_t)c->rank);
  }
  if (c->rank + 1 < c->world) {
    CLC_HIP(hipMemcpyAsync(hr + seg_off + 12 * cap_per_rank, c->d_recv + seg_off + 12 * cap_per_rank,
                           seg_bytes * (size_t)(c->world - 1 - c->rank), hipMemcpyDeviceToHost, stream));
    ++c->host_copies; c->host_copy_bytes += (long long)(seg_bytes * (size_t)(c->world - 1 - c->rank));
  }
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
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  if (c->d_base) (void)hipFree(c->d_base);
  for (int t = 0; t < 2; ++t)
    if (c->h_base[t]) (void)hipHostFree(c->h_base[t]);
  delete c;
}

int clc_comm_rank(const clc_comm* c) { return c ? c->rank : -1; }
int clc_comm_world(const clc_comm* c) { return c ? c->world : 0; }
// which RCCL was bound ("" before the first comm call) — diagnostics / tests
const char* clc_comm_library(void) { return rccl().origin.c_str(); }

int clc_comm_set_root(clc_comm* c, int root) {
  if (!c || root < -1 || root >= c->world) return fail(CLC_ERR_INVALID_ARG, "clc_comm_set_root: root must be -1 (every rank) or a rank of the communicator");
  if (c->fl.active) return fail(CLC_ERR_INVALID_ARG, "clc_comm_set_root: a pipelined step is in flight (clc_gather_flush first)");
  c->root = root;
  return CLC_OK;
}

int clc_comm_get_info(const clc_comm* c, clc_comm_info* out) {
  if (!c || !out) return fail(CLC_ERR_INVALID_ARG, "clc_comm_get_info: bad argument");
  std::memset(out, 0, sizeof(*out));
  out->struct_size = (int32_t)sizeof(*out);
  out->rank = c->rank;
  out->world = c->world;
  out->root = c->root;
  out->rooted_collective_available = (c->comm && rccl().Gather) ? 1 : 0;
  out->copies_other_ranks_to_host = comm_copies_others(c) ? 1 : 0;
  out->step_in_flight = c->fl.active ? 1 : 0;
  out->collectives = c->collectives;
  out->rooted_collectives = c->rooted_collectives;
  out->host_copies = c->host_copies;
  out->host_copy_bytes = c->host_copy_bytes;
  out->pipelined_steps = c->pipelined_steps;
  return CLC_OK;
}

int clc_gather_results(clc_comm* c, int64_t first_global_index, size_t cap_per_rank, clc_result_record* all_records) {
  if (!c || cap_per_rank == 0 || first_global_index < 0)
    return fail(CLC_ERR_INVALID_ARG, "clc_gather_results: bad argument");
  if (c->fl.active) return fail(CLC_ERR_INVALID_ARG, "clc_gather_results: a pipelined step is in flight (clc_gather_flush first)");
  clc_handle* h = c->h;
  size_t n_local = h->results_valid;
  // This is a collective: a rank that returned before it would leave every other rank blocked in it.  A rank
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
  c->pad_from = c->pad_base = -1;  // (this collective rewrites the own segment of d_recv: the fused form pads it again)
  const int threads = 256;
  const long long seg_off = 12ll * (long long)c->rank * (long long)cap_per_rank;
  // packed straight into this rank's segment of the receive buffer: the collective runs in place
  hipLaunchKernelGGL(clc::pack_results_kernel, dim3((unsigned)((cap_per_rank + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, h->d_results, (long long)n_local, (long long)cap_per_rank, (double)first_global_index,
                     c->d_recv + seg_off);
  if (hipGetLastError() != hipSuccess && local_rc == CLC_OK) { local_rc = CLC_ERR_HIP; local_msg = "clc_gather_results: pack kernel launch failed (this rank's segment is undefined)"; }
  const size_t count = cap_per_rank * (sizeof(clc_result_record) / sizeof(double));
  int rooted = 0;
  ncclResult_t r = comm_collect(c->comm, c->rank, c->root, c->d_recv + seg_off, c->d_recv, count, h->stream, &rooted);
  if (r != ncclSuccess) return rccl_fail(rooted ? "ncclGather" : "ncclAllGather", r);
  ++c->collectives; c->rooted_collectives += rooted;
  const size_t seg_bytes = sizeof(clc_result_record) * cap_per_rank;
  const size_t bytes = seg_bytes * (size_t)c->world;
  double* hr = c->h_recv_of(c->cur);
  // the own segment always (this form's kernel did not write it to the host), the other ranks' at the root / without a root
  CLC_HIP(hipMemcpyAsync(hr + seg_off, c->d_recv + seg_off, seg_bytes, hipMemcpyDeviceToHost, h->stream));
  ++c->host_copies; c->host_copy_bytes += (long long)seg_bytes;
  if (comm_copies_others(c)) {
    const int rc = comm_copy_others(c, c->cur, cap_per_rank, h->stream);
    if (rc != CLC_OK) return rc;
  }
  CLC_HIP(hipStreamSynchronize(h->stream));
  if (all_records) std::memcpy(all_records, hr, bytes);
  if (local_rc != CLC_OK) return fail(local_rc, local_msg);
  return CLC_OK;
}

}  // extern "C"

namespace {

// Enqueue ONE step on the solver's stream: checks, start poses into the pinned buffer, padding, the resident kernel whose epilogue
// writes the records (device segment + host twin `tw`), the collective in place.  Nothing is waited for.  `wait_copy`: the previous
// step's device -> host copies still read the other ranks' segments of d_recv — the collective waits for them (event).
// Returns CLC_OK with c->fl filled in (fl.fused = false: the batch does not run as the one-launch solve — the caller takes the two calls),
// or an error that made entering the collective impossible.
int step_enqueue(clc_comm* c, const clc_options& opt, const double* poses0, int64_t first_global_index, size_t cap_per_rank, int tw, bool wait_copy,
                 const char* who) {
  clc_handle* h = c->h;
  const size_t P = h->n_problems;
  clc_comm::Flight& fl = c->fl;
  fl = clc_comm::Flight();
  fl.twin = tw;
  fl.cap = cap_per_rank;
  fl.P = P;
  fl.t0 = std::chrono::steady_clock::now();
  // This is a collective (see clc_gather_results): a rank with a LOCAL problem still enters it, with padding only, and reports afterwards.
  auto local_fail = [&](int rc, const std::string& msg) { if (fl.local_rc == CLC_OK) { fl.local_rc = rc; fl.local_msg = msg; } };
  const std::string w(who);
  if (P > cap_per_rank) local_fail(CLC_ERR_INVALID_ARG, w + ": cap_per_rank < local problems (this rank contributed padding only)");
  if (P > 0 && !poses0) local_fail(CLC_ERR_INVALID_ARG, w + ": NULL start poses (this rank contributed padding only)");
  if (P > 0 && fl.local_rc == CLC_OK && (!h->d_btiles || !h->h_poses)) local_fail(CLC_ERR_NO_DATA, w + ": no problems uploaded");
  if (P > 0 && fl.local_rc == CLC_OK) {
    // options as clc_solve_batched checks them; the start poses go into the handle's pinned buffer (what the kernel reads) and are
    // checked for non-finite values in the same pass over them.  (The previous step's kernel has finished: nothing still reads that buffer.)
    int rc = batched_check_inputs(who, opt, poses0, 0);
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
      if (bad) rc = fail(CLC_ERR_NONFINITE, (w + ": non-finite initial pose").c_str());
    }
    if (rc != CLC_OK) local_fail(rc, clc_last_error());
  }
  CLC_HIP(hipSetDevice(h->device));
  BatchedLaunch bl;
  if (P > 0 && fl.local_rc == CLC_OK) {
    const int rc = batched_launch_setup(h, opt, &bl);
    if (rc != CLC_OK) local_fail(rc, clc_last_error());
  }
  if (P > 0 && fl.local_rc == CLC_OK && !bl.resident) { fl.fused = false; return CLC_OK; }
  {
    const int rc = comm_ensure_buffers(c, cap_per_rank);
    if (rc != CLC_OK) return rc;  // (no buffers: this rank cannot take part — see clc_gather_results)
  }
  const size_t n_local = fl.local_rc == CLC_OK ? P : 0;
  fl.n_local = n_local;
  const long long seg_off = 12ll * (long long)c->rank * (long long)cap_per_rank;  // this rank's segment of the gathered array (doubles)
  double* seg = c->d_recv + seg_off;
  if (c->pad_from != (long long)n_local || c->pad_base != (long long)cap_per_rank) {  // (first call, or the shape changed)
    const long long n_pad = (long long)cap_per_rank - (long long)n_local;
    if (n_pad > 0) {  // the padding records of this rank's segment: device (what the collective sends) and both host twins
      hipLaunchKernelGGL(pad_records_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, h->stream, seg, (long long)n_local, (long long)cap_per_rank);
      if (hipGetLastError() != hipSuccess) local_fail(CLC_ERR_HIP, w + ": padding kernel launch failed");
      for (int t = 0; t < 2; ++t)
        for (size_t k = n_local; k < cap_per_rank; ++k) {
          double* o = c->h_recv_of(t) + seg_off + 12 * k;
          for (int i = 0; i < 11; ++i) o[i] = 0.0;
          o[11] = -1.0;
        }
    }
    c->pad_from = (long long)n_local;
    c->pad_base = (long long)cap_per_rank;
  }
  fl.timed = opt.profile_events == 1 && ensure_events(h, 2) == CLC_OK;  // (no events: the step runs untimed rather than leave the collective)
  if (fl.local_rc != CLC_OK) fl.n_local = 0;
  if (fl.n_local > 0) {
    if (fl.timed) CLC_HIP(hipEventRecord(h->ev[0], h->stream));
    // the last workgroup to finish (totals' arrival count == goal) copies the totals to the host twin
    launch_resident_batch(h, opt, bl, nullptr, c->d_base, (double)first_global_index, c->hd_base[tw], seg_off, c->stats_seen[3] + (unsigned long long)P);
    if (hipGetLastError() != hipSuccess) { local_fail(CLC_ERR_HIP, w + ": kernel launch failed (this rank's records are undefined)"); fl.n_local = 0; }
    else if (fl.timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
  }
  // The collective in place (send buffer = this rank's segment of the receive buffer).
  if (wait_copy && c->ev_copy) CLC_HIP(hipStreamWaitEvent(h->stream, c->ev_copy, 0));
  const size_t count = cap_per_rank * (sizeof(clc_result_record) / sizeof(double));
  int rooted = 0;
  ncclResult_t r = comm_collect(c->comm, c->rank, c->root, seg, c->d_recv, count, h->stream, &rooted);
  if (r != ncclSuccess) {
    // the kernel may be running: bring the host's view of the running totals back in step with the device before reporting
    (void)hipStreamSynchronize(h->stream);
    unsigned long long dev[4] = {0, 0, 0, 0};
    if (hipMemcpy(dev, c->d_base, sizeof(dev), hipMemcpyDeviceToHost) == hipSuccess) std::memcpy(c->stats_seen, dev, sizeof(dev));
    return rccl_fail(rooted ? "ncclGather" : "ncclAllGather", r);
  }
  ++c->collectives; c->rooted_collectives += rooted;
  fl.active = true;
  return CLC_OK;
}

// The bookkeeping of a step whose kernel and collective have completed: totals of the local shard from twin fl.twin.
void step_stats(clc_comm* c, clc_batch_stats* stats) {
  clc_handle* h = c->h;
  clc_comm::Flight& fl = c->fl;
  unsigned long long now[4];
  std::memcpy(now, c->h_base[fl.twin], sizeof(now));
  if (fl.n_local == 0) std::memcpy(now, c->stats_seen, sizeof(now));  // (no launch: the totals stand)
  if (fl.n_local > 0 && now[3] != c->stats_seen[3] + (unsigned long long)fl.P) {
    // the last workgroup publishes the totals when the arrival count reaches stats_seen[3] + P: anything else means an earlier call
    // left the count and the host's copy of it apart — resynchronise from the device and report this call's totals as unknown rather
    // than as a difference of unrelated numbers
    unsigned long long dev[4] = {0, 0, 0, 0};
    if (hipMemcpy(dev, c->d_base, sizeof(dev), hipMemcpyDeviceToHost) == hipSuccess) std::memcpy(c->stats_seen, dev, sizeof(dev));
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->problems = -1; stats->evaluations = -1; stats->iterations = -1; stats->not_converged = -1; stats->fused = 1; }
    return;
  }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->evaluations = (int64_t)(now[0] - c->stats_seen[0]);
    stats->iterations = (int64_t)(now[1] - c->stats_seen[1]);
    stats->not_converged = (int64_t)(now[2] - c->stats_seen[2]);
    stats->problems = (int64_t)(now[3] - c->stats_seen[3]);
    stats->fused = 1;
    if (fl.timed && fl.n_local > 0) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) stats->kernel_ms = (double)ms;
      else (void)hipGetLastError();
    }
    stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - fl.t0).count();
  }
  for (int i = 0; i < 4; ++i) c->stats_seen[i] = now[i];
}

// The two-call form inside the one call (a batch that does not run as the one-launch on-chip solve): same records; poses and summaries
// cross PCIe as well.  The handle's pinned pose buffer is an INPUT of the gather calls: its start poses are put back afterwards, so that
// a caller who filled it once (clc_batched_host_buffers) and repeats the call starts from the same poses on either path.
int step_two_calls(clc_comm* c, const clc_options& opt, int64_t first_global_index, size_t cap_per_rank, clc_result_record* all_records, clc_batch_stats* stats) {
  clc_handle* h = c->h;
  const size_t P = h->n_problems;
  const auto t0 = c->fl.t0;
  std::vector<double> start(h->h_poses, h->h_poses + 7 * P);
  int rc = clc_solve_batched(h, &opt, h->h_poses, h->h_summaries);  // (the start poses are in the pinned buffer already)
  std::string msg = rc != CLC_OK ? clc_last_error() : "";
  if (rc != CLC_OK) h->results_valid = 0;
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
  std::memcpy(h->h_poses, start.data(), sizeof(double) * 7 * P);
  if (rc != CLC_OK) return fail(rc, msg.c_str());
  return rc2;
}

}  // namespace

extern "C" {

int clc_solve_batched_gather(clc_comm* c, const clc_options* opt_in, const double* poses0, int64_t first_global_index,
                             size_t cap_per_rank, clc_result_record* all_records, clc_batch_stats* stats) {
  if (!c || cap_per_rank == 0 || first_global_index < 0) return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather: bad argument");
  if (c->fl.active) return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather: a pipelined step is in flight (clc_gather_flush first)");
  clc_handle* h = c->h;
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_options_default(&opt);
  int rc = step_enqueue(c, opt, poses0, first_global_index, cap_per_rank, c->cur, false, "clc_solve_batched_gather");
  if (rc != CLC_OK) { c->fl.active = false; return rc; }
  if (!c->fl.fused) return step_two_calls(c, opt, first_global_index, cap_per_rank, all_records, stats);
  // This rank's own records and the totals are in host memory already — the kernel wrote them there —, so only the OTHER ranks' segments
  // are copied down, and only where they are wanted (no root: everywhere; a root: there.  World size 1: nothing); one synchronisation.
  if (comm_copies_others(c)) {
    rc = comm_copy_others(c, c->fl.twin, cap_per_rank, h->stream);
    if (rc != CLC_OK) { (void)hipStreamSynchronize(h->stream); c->fl.active = false; return rc; }
  }
  CLC_HIP(hipStreamSynchronize(h->stream));  // (kernel completion makes what it wrote over PCIe visible)
  h->results_valid = 0;  // (the handle's own result buffer was not written: a later clc_gather_results has nothing to send)
  if (all_records) std::memcpy(all_records, c->h_recv_of(c->cur), sizeof(clc_result_record) * cap_per_rank * (size_t)c->world);
  step_stats(c, stats);
  c->fl.active = false;
  if (c->fl.local_rc != CLC_OK) return fail(c->fl.local_rc, c->fl.local_msg.c_str());
  return CLC_OK;
}

int clc_solve_batched_gather_pipelined(clc_comm* c, const clc_options* opt_in, const double* poses0, int64_t first_global_index,
                                       size_t cap_per_rank, const clc_result_record** prev_records, clc_batch_stats* prev_stats) {
  if (!c || cap_per_rank == 0 || first_global_index < 0 || !prev_records)
    return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather_pipelined: bad argument");
  clc_handle* h = c->h;
  if (c->fl.active && cap_per_rank > c->cap)  // (growing the buffers would free the twin whose records this call hands back)
    return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather_pipelined: cap_per_rank grew while a step is in flight (clc_gather_flush first)");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_options_default(&opt);
  *prev_records = nullptr;
  if (prev_stats) std::memset(prev_stats, 0, sizeof(*prev_stats));
  CLC_HIP(hipSetDevice(h->device));
  if (!c->copy_stream) {
    CLC_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    CLC_HIP(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
  }
  // 1. the previous step: its kernel and collective have to be over before this step's start poses overwrite the pinned buffer its
  //    kernel reads and before this step's totals are defined; its device -> host copies go to the COPY stream — they run while this
  //    step's kernel does (3.) and this step's collective waits for them (event) before it overwrites the segments they read.
  const bool had_prev = c->fl.active;
  clc_comm::Flight prev = c->fl;
  int prev_rc = CLC_OK;
  bool copying = false;
  if (had_prev) {
    int rc = CLC_OK;
    if (hipStreamSynchronize(h->stream) != hipSuccess) {  // kernel + collective done; the own records and the totals are visible in the twin
      c->fl.active = false;
      return fail(CLC_ERR_HIP, "clc_solve_batched_gather_pipelined: the step in flight failed", hipGetLastError());
    }
    step_stats(c, prev_stats);
    if (comm_copies_others(c) && c->world > 1) {
      rc = comm_copy_others(c, prev.twin, prev.cap, c->copy_stream);
      if (rc != CLC_OK) { c->fl.active = false; return rc; }
      CLC_HIP(hipEventRecord(c->ev_copy, c->copy_stream));
      copying = true;
    }
    c->cur = prev.twin;
    if (prev.local_rc != CLC_OK) prev_rc = fail(prev.local_rc, prev.local_msg.c_str());
  }
  // 2. this step, into the other twin
  const int tw = had_prev ? 1 - prev.twin : c->cur;
  int rc = step_enqueue(c, opt, poses0, first_global_index, cap_per_rank, tw, copying, "clc_solve_batched_gather_pipelined");
  if (rc == CLC_OK && !c->fl.fused) {
    // not the one-launch solve (a problem too large for a workgroup, explicit flags): there is no kernel epilogue to overlap with — refused;
    // the previous step's records are still handed back
    c->fl.active = false;
    if (copying) CLC_HIP(hipStreamSynchronize(c->copy_stream));
    if (had_prev) *prev_records = reinterpret_cast<const clc_result_record*>(c->h_recv_of(prev.twin));
    return fail(CLC_ERR_INVALID_ARG, "clc_solve_batched_gather_pipelined: the batch does not run as the one-launch on-chip solve (use clc_solve_batched_gather)");
  }
  // 3. while this step's kernel runs: wait for the previous step's copies
  if (copying) CLC_HIP(hipStreamSynchronize(c->copy_stream));
  if (rc != CLC_OK) { c->fl.active = false; return rc; }
  ++c->pipelined_steps;
  h->results_valid = 0;
  if (had_prev) *prev_records = reinterpret_cast<const clc_result_record*>(c->h_recv_of(prev.twin));
  return prev_rc;
}

int clc_gather_flush(clc_comm* c, const clc_result_record** records, clc_batch_stats* stats) {
  if (!c || !records) return fail(CLC_ERR_INVALID_ARG, "clc_gather_flush: bad argument");
  *records = nullptr;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!c->fl.active) return CLC_OK;  // nothing in flight
  clc_handle* h = c->h;
  CLC_HIP(hipSetDevice(h->device));
  clc_comm::Flight& fl = c->fl;
  if (comm_copies_others(c)) {
    const int rc = comm_copy_others(c, fl.twin, fl.cap, h->stream);
    if (rc != CLC_OK) { (void)hipStreamSynchronize(h->stream); fl.active = false; return rc; }
  }
  CLC_HIP(hipStreamSynchronize(h->stream));
  step_stats(c, stats);
  c->cur = fl.twin;
  fl.active = false;
  *records = reinterpret_cast<const clc_result_record*>(c->h_recv_of(c->cur));
  if (fl.local_rc != CLC_OK) return fail(fl.local_rc, fl.local_msg.c_str());
  return CLC_OK;
}

#ifdef CLC_TEST_HOOKS
// Test hook (hooks build only): a communicator laid out as rank `rank` of `world` WITHOUT RCCL behind it (see comm_collect).
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
  return (c && c->h_base[c->cur]) ? reinterpret_cast<const clc_result_record*>(c->h_recv_of(c->cur)) : nullptr;
}

}  // extern "C"
