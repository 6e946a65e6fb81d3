// abi_batched.hip — clc_solve_batched: many independent problems per launch (resident kernel, whole-solve kernel, lockstep launches).
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

namespace clc_abi {

void warm_batched() {
  warm_kernel(reinterpret_cast<const void*>(&clc::resident_solve_kernel<true, true, 4, kResPR256, kResPL256, kResCtrl4>));
  warm_kernel(reinterpret_cast<const void*>(&clc::batched_init_kernel));
}

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
    const size_t res_bytes = (size_t)h->bres.rows * (size_t)h->bres.lanes * (h->bres.with_z ? 3 : 2) * sizeof(double);
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

void launch_resident_batch(clc_handle* h, const clc_options& opt, const BatchedLaunch& bl, clc_summary* d_summaries, double* d_results,
                           double rec_base, double* rec_host, long long seg_off, unsigned long long goal, const MultiStartLaunch* multistart) {
  // one workgroup per problem, the problem read from HBM once and kept in registers + LDS for its whole solve
  // (multi-start: one workgroup per START, every one of them on problem 0's points)
  const size_t P = multistart ? multistart->n_starts : h->n_problems;
  double* const d_poses = multistart ? multistart->d_poses : h->d_poses;
  const int uni_ppl = multistart ? -2 - h->bres.max_ppl : h->bres.uni_ppl;
  const unsigned int* d_row = reinterpret_cast<const unsigned int*>(h->bres.d_row);
  const clc::ResLane* d_desc = reinterpret_cast<const clc::ResLane*>(h->bres.d_desc);
#define CLC_LAUNCH_RES(LOSS, NT, NW, PR, PL)                                                                                  \
  hipLaunchKernelGGL((clc::resident_solve_kernel<LOSS, NT, NW, PR, PL, kResCtrl##NW>), dim3((unsigned)P), dim3(NW * 64), 0, h->stream, \
                     h->bres.d_xy, d_row, d_desc, h->d_bgroups, uni_ppl, opt, nullptr, 0, d_poses, d_summaries, d_results, nullptr, nullptr, \
                     rec_base, rec_host, seg_off, goal)
#define CLC_LAUNCH_RES_V(NW, PR, PL)                                                                                          \
  do {                                                                                                                        \
    if (opt.use_loss) { if (bl.res_nt) CLC_LAUNCH_RES(true, true, NW, PR, PL); else CLC_LAUNCH_RES(true, false, NW, PR, PL); } \
    else { if (bl.res_nt) CLC_LAUNCH_RES(false, true, NW, PR, PL); else CLC_LAUNCH_RES(false, false, NW, PR, PL); }            \
  } while (0)
#define CLC_LAUNCH_RESZ(LOSS, NT)                                                                                              \
  hipLaunchKernelGGL((clc::resident_solve_kernel<LOSS, NT, 8, kResPRz, kResPLz, kResCtrl8, true>), dim3((unsigned)P), dim3(512), 0, h->stream, \
                     h->bres.d_xy, d_row, d_desc, h->d_bgroups, uni_ppl, opt, nullptr, 0, d_poses, d_summaries, d_results, nullptr, nullptr, \
                     rec_base, rec_host, seg_off, goal, h->bres.d_z)
  if (h->bres.with_z) {  // 24-byte slots (p.z != 0 in some record of the batch)
    if (opt.use_loss) { if (bl.res_nt) CLC_LAUNCH_RESZ(true, true); else CLC_LAUNCH_RESZ(true, false); }
    else { if (bl.res_nt) CLC_LAUNCH_RESZ(false, true); else CLC_LAUNCH_RESZ(false, false); }
  } else if (h->bres.lanes == 256) CLC_LAUNCH_RES_V(4, kResPR256, kResPL256);
  else CLC_LAUNCH_RES_V(8, kResPR512, kResPL512);
#undef CLC_LAUNCH_RESZ
#undef CLC_LAUNCH_RES_V
#undef CLC_LAUNCH_RES
}

int batched_check_inputs(const char* who, const clc_options& opt, const double* poses, size_t P) {
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, (std::string(who) + ": max_num_iterations < 0").c_str());
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, (std::string(who) + ": loss_scale_factor must be > 0").c_str());
  // all 7 P start-pose doubles finite: exponent field not all ones — an integer OR-reduction the compiler vectorises (57 344 values at C4)
  unsigned long long bad = 0;
  for (size_t i = 0; i < 7 * P; ++i) {
    unsigned long long b;
    std::memcpy(&b, &poses[i], sizeof(b));
    bad |= (unsigned long long)(((b >> 52) & 0x7FFull) == 0x7FFull);
  }
  if (bad) return fail(CLC_ERR_NONFINITE, (std::string(who) + ": non-finite initial pose").c_str());
  return CLC_OK;
}

}  // namespace clc_abi

extern "C" {

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
  const size_t P = h->n_problems;
  {
    const int rc = batched_check_inputs("clc_solve_batched", opt, poses, P);
    if (rc != CLC_OK) return rc;
  }
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
    // (A completion flag raised by the last workgroup to finish, polled by the host instead of this blocking synchronisation, was
    // measured: every workgroup then needs a system-scope release before it counts itself in, which on this part writes back L2 —
    // C4 shard 0.93 -> 1.28 ms, C3 0.150 -> 0.166.  The single-workgroup solve keeps its flag: one release per solve.)
    const bool timed = opt.profile_events == 1;  // HIP event pair around the one launch -> clc_summary.eval_kernel_ms of every problem
    if (timed) {
      const int rc = ensure_events(h, 2);
      if (rc != CLC_OK) return rc;
      CLC_HIP(hipEventRecord(h->ev[0], h->stream));
    }
    launch_resident_batch(h, opt, bl, h->d_summaries, h->d_results, 0.0, nullptr, 0, 0);
    CLC_HIP(hipGetLastError());
    if (timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
    // (kernel completion makes the outcomes written over PCIe visible; polling the stream with hipStreamQuery instead of this
    // blocking call was measured: no difference — the 70-80 us between the kernel's end event and the return are not the wake-up)
    CLC_HIP(hipStreamSynchronize(h->stream));
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


// Multi-hypothesis calibration on SHARED observations (BASELINE.json north_star: "batched/multi-hypothesis calibration"): n_starts
// independent LM solves — one ceres::Solve each, src/LaseCamCalCeres.cpp:299-309 — from n_starts start poses on the ONE problem the handle
// holds as a batch of one (clc_upload_batched* with n_problems = 1).  Where the problem fits a workgroup (the on-chip lane layout: at most
// 256 x 44 points and 256 scans, or 512 x 22 / 512 scans) ONE launch runs every start: a workgroup per start, every workgroup loading the
// SAME lane layout (one copy in HBM — 0.17 MB for 1e4 observations where 1 024 uploaded copies are 178 MB — read by the first round of
// workgroups, served from L2 to the rest) and solving from its own pose; otherwise the starts run one after the other on the batch's
// streaming path.  Same kernel, same arithmetic as clc_solve_batched of n_starts uploaded copies: bit-identical results.
int clc_solve_multistart(clc_handle* h, const clc_options* opt_in, size_t n_starts, double* poses, clc_summary* summaries) {
  if (!h || !poses || !summaries || n_starts == 0) return fail(CLC_ERR_INVALID_ARG, "clc_solve_multistart: bad argument");
  if (!h->d_btiles || h->n_problems != 1)
    return fail(CLC_ERR_NO_DATA, "clc_solve_multistart: the shared observations must be uploaded as a batch of ONE problem (clc_upload_batched, n_problems = 1)");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_options_default(&opt);
  {
    const int rc = batched_check_inputs("clc_solve_multistart", opt, poses, n_starts);
    if (rc != CLC_OK) return rc;
  }
  CLC_HIP(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  BatchedLaunch bl;
  {
    const int rc = batched_launch_setup(h, opt, &bl);
    if (rc != CLC_OK) return rc;
  }
  if (!bl.resident) {  // the problem does not fit a workgroup (or explicit flags): one start after the other, still ONE copy of the data
    for (size_t k = 0; k < n_starts; ++k) {
      const int rc = clc_solve_batched(h, &opt, poses + 7 * k, summaries + k);
      if (rc != CLC_OK) return rc;
    }
    h->results_valid = 0;  // (the handle's result buffer holds the last start only: nothing for clc_gather_results)
    return CLC_OK;
  }
  if (n_starts > h->ms_cap) {
    if (h->h_ms_poses) CLC_HIP(hipHostFree(h->h_ms_poses));
    if (h->h_ms_summaries) CLC_HIP(hipHostFree(h->h_ms_summaries));
    if (h->d_ms_results) CLC_HIP(hipFree(h->d_ms_results));
    h->h_ms_poses = h->d_ms_poses = h->d_ms_results = nullptr;
    h->h_ms_summaries = h->d_ms_summaries = nullptr;
    h->ms_cap = 0;
    CLC_HIP(hipHostMalloc(&h->h_ms_poses, sizeof(double) * 7 * n_starts, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_ms_poses), h->h_ms_poses, 0));
    CLC_HIP(hipHostMalloc(&h->h_ms_summaries, sizeof(clc_summary) * n_starts, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_ms_summaries), h->h_ms_summaries, 0));
    CLC_HIP(hipMalloc(&h->d_ms_results, sizeof(clc_result_record) * n_starts));
    h->ms_cap = n_starts;
  }
  // (the previous call ended with a stream synchronisation: nothing still reads or writes the staging buffers)
  std::memcpy(h->h_ms_poses, poses, sizeof(double) * 7 * n_starts);
  const bool timed = opt.profile_events == 1;
  if (timed) {
    const int rc = ensure_events(h, 2);
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipEventRecord(h->ev[0], h->stream));
  }
  MultiStartLaunch ms;
  ms.n_starts = n_starts;
  ms.d_poses = h->d_ms_poses;
  launch_resident_batch(h, opt, bl, h->d_ms_summaries, h->d_ms_results, 0.0, nullptr, 0, 0, &ms);
  CLC_HIP(hipGetLastError());
  if (timed) CLC_HIP(hipEventRecord(h->ev[1], h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));  // (kernel completion makes the outcomes written over PCIe visible)
  float kernel_ms = 0.0f;
  if (timed) CLC_HIP(hipEventElapsedTime(&kernel_ms, h->ev[0], h->ev[1]));
  std::memcpy(poses, h->h_ms_poses, sizeof(double) * 7 * n_starts);
  std::memcpy(summaries, h->h_ms_summaries, sizeof(clc_summary) * n_starts);
  const double ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (size_t k = 0; k < n_starts; ++k) {
    summaries[k].solve_ms = ms_wall;
    if (timed) { summaries[k].eval_kernel_ms = (double)kernel_ms; summaries[k].eval_kernel_launches = 1; }
  }
  for (size_t k = 0; k < n_starts; ++k)
    if (!all_finite(poses + 7 * k, 7)) return fail(CLC_ERR_NONFINITE, "clc_solve_multistart: non-finite result");
  return CLC_OK;
}

}  // extern "C"

#if defined(CLC_STAMPS) && defined(CLC_TEST_HOOKS)
// Debug build only (scripts/*_stamps.py): copy the stamp buffers of THIS unit's kernels out (and clear them).
#pragma GCC visibility push(default)
extern "C" int clc_debug_res_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_res_stamp_buf)) bytes = sizeof(clc::clc_res_stamp_buf);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_res_stamp_buf), bytes) != hipSuccess) return CLC_ERR_HIP;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(clc::clc_res_stamp_buf)) != hipSuccess) return CLC_ERR_HIP;
  return hipMemset(p, 0, sizeof(clc::clc_res_stamp_buf)) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
extern "C" int clc_debug_res_ctrl_stamps(void* dst, size_t bytes) {
  if (bytes > sizeof(clc::clc_res_stamp_ctrl)) bytes = sizeof(clc::clc_res_stamp_ctrl);
  if (hipDeviceSynchronize() != hipSuccess) return CLC_ERR_HIP;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(clc::clc_res_stamp_ctrl), bytes) == hipSuccess ? CLC_OK : CLC_ERR_HIP;
}
#pragma GCC visibility pop
#endif
