// abi_solve.hip — clc_eval and clc_solve: the launch sequences of one problem (step-kernel chain, single-workgroup resident kernel, cooperative kernel).
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

namespace {

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

}  // namespace

namespace clc_abi {
void warm_solve() {
  warm_kernel(reinterpret_cast<const void*>(&clc::resident_solve_kernel<true, false, 8, kResPR512, kResPL512, 0>));
  warm_kernel(reinterpret_cast<const void*>(&clc::coop_solve_kernel<true, false>));
}

void launch_eval(clc_handle* h, int grid, bool with_jac, bool with_loss, const double* d_pose, const int32_t* d_status, double lf,
                 const clc::Pose7* pose_arg) {
  if (with_jac) ::launch_eval<true>(h, grid, with_loss, d_pose, d_status, lf, pose_arg);
  else ::launch_eval<false>(h, grid, with_loss, d_pose, d_status, lf, pose_arg);
}
}  // namespace clc_abi

extern "C" {

size_t clc_num_observations(const clc_handle* h) { return h ? h->n_obs : 0; }

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

}  // extern "C"

namespace {

}  // namespace

namespace clc_abi {
// clc_solve as a chain of step_kernel launches (clc_kernels.hpp "Step kernel"): launch 0 evaluates at the initial
// pose, launch k >= 1 consumes the rows of launch k-1 in every workgroup and evaluates at the next point.  The
// host only keeps `lookahead` launches queued beyond the last pass the device reported consumed.
// win_first/win_last/win_ms (profiling hook clc_time_steps): HIP events are recorded on the stream right before launch
// `win_first` and right after launch `win_last`; *win_ms receives the elapsed time between them.
int solve_stepped(clc_handle* h, const clc_options& opt, int grid, double pose[7], clc_summary* summary,
                  clc_iteration* trace, int trace_cap, std::chrono::steady_clock::time_point t0,
                  int win_first, int win_last, float* win_ms) {
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
}  // namespace clc_abi

namespace {


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
  const bool uni_ctrl = h->single_uni_ctrl;  // the cooperative kernel's controller here: the bit-identity test of the two (hooks build)
  const bool timed = opt.profile_events == 2;  // an event pair around the one launch -> eval_kernel_ms, eval_kernel_launches = 1
  if (timed) {
    const int rc = ensure_events(h, 2);
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipEventRecord(h->ev[0], h->stream));
  }
  if (opt.use_loss) { if (uni_ctrl) CLC_LAUNCH_SINGLE(true, 1); else CLC_LAUNCH_SINGLE(true, 0); }
  else { if (uni_ctrl) CLC_LAUNCH_SINGLE(false, 1); else CLC_LAUNCH_SINGLE(false, 0); }
#undef CLC_LAUNCH_SINGLE
  CLC_HIP(hipGetLastError());
  if (timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
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
    if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == 0 || want_trace || timed) CLC_HIP(hipStreamSynchronize(h->stream));
  }
  float kernel_ms = 0.0f;
  if (timed) CLC_HIP(hipEventElapsedTime(&kernel_ms, h->ev[0], h->ev[1]));
  *summary = *h->h_ssummary;
  for (int i = 0; i < 7; ++i) pose[i] = h->h_spose[i];
  if (want_trace) {
    const int n = std::min(std::min(summary->num_iterations + 1, trace_cap), h->trace_cap);
    if (n > 0) CLC_HIP(hipMemcpy(trace, h->d_trace, sizeof(clc_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  summary->eval_kernel_ms = timed ? (double)kernel_ms : 0.0;
  summary->eval_kernel_launches = timed ? 1 : 0;
  summary->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!all_finite(pose, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve: non-finite result");
  return CLC_OK;
}

// clc_solve as ONE launch of 256 co-resident workgroups that keep the problem on chip (clc_coop.hpp).  Returns kCoopFallback when
// the path cannot be used (device too small, or the launch timed out in its exchange): the caller runs the step chain instead.
constexpr int kCoopFallback = -1000;

// One cooperative launch at a time per device in this process.  The kernel needs a CU per workgroup; a second cooperative launch from
// another handle (another stream, another thread) interleaves its workgroups with the first one's, neither grid becomes co-resident,
// and BOTH time out at their census — round 4: two threads alternated aborts.  A solve is ~0.1 ms, so the second caller waits for the
// gate (bounded: 5 ms) rather than fall back to the slower step chain; if the wait runs out it takes the step chain without counting
// an abort.  (Other PROCESSES on the same GPU cannot be seen from here: for them the census, the abort word and the back-off remain.)
struct CoopGate {
  static constexpr int kMaxDevices = 64;
  static std::atomic<int>& slot(int device) {
    static std::atomic<int> busy[kMaxDevices];
    return busy[device >= 0 && device < kMaxDevices ? device : 0];
  }
  std::atomic<int>* held = nullptr;
  bool acquire(int device) {
    std::atomic<int>& a = slot(device);
    const auto t0 = std::chrono::steady_clock::now();
    long long spins = 0;
    for (;;) {
      int expected = 0;
      if (a.compare_exchange_weak(expected, 1, std::memory_order_acquire)) { held = &a; return true; }
      // the holder's solve is ~0.1 ms: spin briefly, then give the core away (with more solver threads than cores the spinners would
      // otherwise take the CPU the gate's holder needs to finish)
      if (++spins < 64) continue;
      std::this_thread::yield();
      if ((spins & 0xF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5e-3) return false;
    }
  }
  ~CoopGate() { if (held) held->store(0, std::memory_order_release); }
};

int solve_coop(clc_handle* h, const clc_options& opt, double pose[7], clc_summary* summary, clc_iteration* trace, int trace_cap,
               std::chrono::steady_clock::time_point t0) {
  if (h->coop_checked == 0) {
    // every instantiation that can be launched below must fit a CU (one workgroup each: co-residency is what makes the polling safe)
    const void* forms[] = {
        (const void*)clc::coop_solve_kernel<true, false, false, false>, (const void*)clc::coop_solve_kernel<false, false, false, false>,
        (const void*)clc::coop_solve_kernel<true, false, true, false>,  (const void*)clc::coop_solve_kernel<false, false, true, false>,
        (const void*)clc::coop_solve_kernel<true, false, false, true>,  (const void*)clc::coop_solve_kernel<false, false, false, true>,
        (const void*)clc::coop_solve_kernel<true, false, true, true>,   (const void*)clc::coop_solve_kernel<false, false, true, true>};
    bool fits = true;
    for (const void* f : forms) {
      int n = 0;
      fits = fits && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, clc::COOP_THREADS, 0) == hipSuccess && n >= 1;
    }
    h->coop_checked = (fits && h->num_cus >= clc::COOP_SMALL_WGS) ? 1 : -1;
    (void)hipGetLastError();
  }
  if (h->coop_checked < 0) return kCoopFallback;
  if (h->num_cus < (h->cres.wgs > 0 ? h->cres.wgs : clc::COOP_WGS)) return kCoopFallback;  // one workgroup per CU, or not at all
  CoopGate gate;  // (released when this function returns: the kernel has finished, or was never launched)
  if (!gate.acquire(h->device)) { ++h->coop_gate_waits_expired; return kCoopFallback; }
  if (!h->d_board) {
    CLC_HIP(hipMalloc(&h->d_board, sizeof(clc::CoopBoard)));
    // (ordered on the handle's stream AND waited for: the caller may switch streams, clc_set_stream, before the next solve)
    CLC_HIP(hipMemsetAsync(h->d_board, 0, sizeof(clc::CoopBoard), h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    h->coop_tag = 1;
#ifdef CLC_TEST_HOOKS
    {  // (tuning hook, hooks build only: first-poll offsets, clc_coop.hpp)
      unsigned long long d[2] = {0, 0};
      if (const char* e = std::getenv("CLC_COOP_D1")) d[0] = (unsigned long long)std::atoll(e);
      if (const char* e = std::getenv("CLC_COOP_D2")) d[1] = (unsigned long long)std::atoll(e);
      if (d[0] || d[1]) CLC_HIP(hipMemcpy(&h->d_board->ctl[1], d, sizeof(d), hipMemcpyHostToDevice));
    }
#endif
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
  const int n_wgs = h->cres.wgs > 0 ? h->cres.wgs : clc::COOP_WGS;  // COOP_WGS, or COOP_SMALL_WGS: the one-hop form
  const unsigned int wgs = (unsigned int)std::max(1, n_wgs - h->coop_test_drop);
  h->coop_test_drop = 0;
  if (timed) CLC_HIP(hipEventRecord(h->ev[0], h->stream));
#define CLC_LAUNCH_COOP(LOSS, Z, ONE)                                                                                                         \
  hipLaunchKernelGGL((clc::coop_solve_kernel<LOSS, false, Z, ONE>), dim3(wgs), dim3(clc::COOP_THREADS), 0, h->stream, h->cres.d_xy, h->cres.d_z, d_row, \
                     d_desc, h->d_groups, h->cres.uni_ppl, opt, p0, d_trace, d_cap, h->d_board, tag0, h->d_spose, h->d_ssummary, h->d_small, d_done, \
                     n_wgs)
  const bool one = n_wgs == clc::COOP_SMALL_WGS;  // the one-hop form on 32 workgroups
  if (h->cres.with_z) {  // 24-byte slots: p.z != 0
    if (one) { if (opt.use_loss) CLC_LAUNCH_COOP(true, true, true); else CLC_LAUNCH_COOP(false, true, true); }
    else { if (opt.use_loss) CLC_LAUNCH_COOP(true, true, false); else CLC_LAUNCH_COOP(false, true, false); }
  } else if (one) { if (opt.use_loss) CLC_LAUNCH_COOP(true, false, true); else CLC_LAUNCH_COOP(false, false, true); }
  else { if (opt.use_loss) CLC_LAUNCH_COOP(true, false, false); else CLC_LAUNCH_COOP(false, false, false); }
#undef CLC_LAUNCH_COOP
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

extern "C" {

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
  // (clc_set_small_on_coop at upload: such a problem ALSO has the cooperative layout and runs on 32 workgroups first — 4.6 instead
  // of 5.3-5.9 us per pass —, with this kernel as the fall-back when the cooperative launch times out or rests)
  const bool single_ok = h->sres.ok && h->launch_auto && (h->auto_disable & 2) == 0 && h->grid_override == 0 && opt.profile_events != 1;
  if (single_ok && !(h->cres.ok && h->small_on_coop)) return solve_resident_single(h, opt, pose, summary, trace, trace_cap, t0);
  // a problem the 256 CUs hold together: the whole solve in one launch of 256 (or 32) co-resident workgroups (same conditions)
  if (h->cres.ok && h->launch_auto && (h->auto_disable & 1) == 0 && h->grid_override == 0 && opt.profile_events != 1 && ++h->coop_eligible > h->coop_retry_at) {
    const int rc = solve_coop(h, opt, pose, summary, trace, trace_cap, t0);
    if (rc != kCoopFallback) return rc;
  }
  if (single_ok) return solve_resident_single(h, opt, pose, summary, trace, trace_cap, t0);
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
  clc::HostMailbox* mb = h->h_mailbox;
  mb->n_done = 0;
  mb->status = CLC_RUNNING;
  std::atomic_thread_fence(std::memory_order_seq_cst);

  clc::Pose7 p0;
  for (int i = 0; i < 7; ++i) p0.v[i] = pose[i];
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

}  // extern "C"

#if defined(CLC_STAMPS) && defined(CLC_TEST_HOOKS)
// Debug build only (scripts/*_stamps.py): copy the stamp buffers of THIS unit's kernels out (and clear them).
#pragma GCC visibility push(default)
extern "C" int clc_debug_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_stamp_buf)) bytes = sizeof(clc::clc_stamp_buf);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_stamp_buf), bytes) != hipSuccess) return CLC_ERR_HIP;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(clc::clc_stamp_buf)) != hipSuccess) return CLC_ERR_HIP;
  return hipMemset(p, 0, sizeof(clc::clc_stamp_buf)) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
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
#pragma GCC visibility pop
#endif
