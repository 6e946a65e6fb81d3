// abi_debug.hip — test and profiling hooks (clc_debug_*, clc_time_*): NOT part of include/clc.h and not in the product library — this unit is empty without -DCLC_TEST_HOOKS.
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

#ifdef CLC_TEST_HOOKS

#pragma GCC visibility push(default)  // (the library is built with -fvisibility=hidden; include/clc.h does not declare these)

extern "C" {

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

// Profiling hook (not part of include/clc.h): one default clc_solve through the step kernel with HIP events on the
// handle's stream right before launch `first` and right after launch `last`; *avg_ms = elapsed / (last - first + 1),
// i.e. the mean period of those back-to-back step_kernel launches.  Launch 0 evaluates the start pose, launch k >= 1
// consumes pass k-1; choose 2 <= first <= last <= passes - 1 to cover steady-state launches that all streamed.
int clc_time_steps(clc_handle* h, const double pose0[7], int first, int last, double* avg_ms, int* passes) {
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

// op 0 rsqrt_pos, 1 rcp_pos, 2 rcp_pos_safe, 3 sqrt_pos, 4 rcp_ge1, 5 rcp_ge1_weight, 6 log via frexp_pos + log_mant_exp: out[i] = f(in[i])
// computed on the device.
int clc_debug_math(clc_handle* h, int op, const double* in, double* out, long long n) {
  if (!h || !in || !out || n < 0 || op < 0 || op > 6) return fail(CLC_ERR_INVALID_ARG, "clc_debug_math: bad argument");
  if (n == 0) return CLC_OK;
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> bi(&h->pool), bo(&h->pool);
  CLC_HIP(bi.alloc((size_t)n));
  CLC_HIP(bo.alloc((size_t)n));
  CLC_HIP(hipMemcpy(bi.p, in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(clc::math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, op, bi.p, bo.p, n);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(out, bo.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return CLC_OK;
}

// What this build of the library contains beyond the default: bit 1 = debug stamps (-DCLC_STAMPS).
int clc_debug_build_features(void) {
  int f = 0;
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

// Test hook: the single-workgroup solve runs the LM controller of the cooperative kernel (wave-uniform arithmetic, state in registers)
// instead of its own (state in LDS) — same arithmetic, bit-identical results, slower there; how the cooperative kernel's controller is
// compared with the serial one on identical totals (tests/test_gpu_lmuni.py).
extern "C" int clc_debug_single_controller(clc_handle* h, int cooperative_kernels) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_single_controller: NULL handle");
  h->single_uni_ctrl = cooperative_kernels != 0;
  return CLC_OK;
}

// Test hook: the host-planned upload of small problems (abi_layouts.hip, small_fast_upload) on / off — the A/B of the two pipelines;
// returns through *count how many uploads have taken it on this handle.
extern "C" int clc_debug_fast_small(clc_handle* h, int enable, long long* count) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_debug_fast_small: NULL handle");
  if (enable >= 0) h->fast_small = enable != 0;
  if (count) *count = h->fast_small_uploads;
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
    launch_eval(h, grid, with_jac != 0, with_loss != 0, h->d_small, nullptr, lf);
  }
  CLC_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; ++r) {
    launch_eval(h, grid, with_jac != 0, with_loss != 0, h->d_small, nullptr, lf);
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

// (the stamp buffers of the -DCLC_STAMPS builds are per translation unit: their read-out hooks sit next to the kernels, in
// abi_solve.hip and abi_batched.hip)

#pragma GCC visibility pop

#endif  // CLC_TEST_HOOKS
