// clc_controller.hpp — everything between two evaluation passes of the single-problem and batched solves on the device:
// the fixed-order reduction of the per-workgroup partial rows and the Levenberg-Marquardt controller (clc_lm.hpp =
// Ceres' TrustRegionMinimizer / LevenbergMarquardtStrategy for src/LaseCamCalCeres.cpp:299-307), serial (one lane) and on
// a wavefront (lm_advance_wave; LEAN for the resident batched kernel), plus the host mailbox they publish to.
#pragma once
#include "clc_device.hpp"

namespace clc {

// ---------------------------------------------------------------------------------------
// K2 — reduction of the block partials + LM controller (single problem).
// ---------------------------------------------------------------------------------------
constexpr int LM_STATE_WORDS = (int)((sizeof(LmState) + 7) / 8);

// Host-visible completion record in pinned (fine-grained) host memory.  lm_kernel publishes
// the number of evaluation passes consumed after every LM step and, at termination, the
// result — so the host can keep the launch queue primed without ever blocking on the stream.
struct HostMailbox {
  int32_t status;  // CLC_RUNNING until the controller terminates
  int32_t n_done;  // evaluation passes consumed so far
  clc_summary summary;
  double pose[7];
  long long prof[8];  // shader-clock stamps of the last lm_kernel launch (debug/profiling)
};

// Tail shared by lm_kernel (own launch) and eval_lm_kernel (last-arriving workgroup of the
// evaluation launch): fixed-order reduction of the block partials + LM controller + publish.
// COHERENT: read partials with agent-scope (sc1) loads — required when they were produced by
// other workgroups of the SAME launch.
template <bool COHERENT>
__device__ __forceinline__ double load_partial(const double* p) {
  if (COHERENT)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// The global loads of the controller, issued as early as possible and consumed later (lm_tail): the thread's
// word of the LM state, its share of the first 256 partial rows and, for the first lane of wave 1, the pass count.
// Summation tree (the same for every caller, so all solve paths agree bit for bit): 16 row groups, group g = rows
// g, g + 16, g + 32, ... summed in that order, then the 16 group sums combined in order.  With HT = 512 helper
// threads, thread (c, g) owns group g of column c (16 loads per round of 256 rows); with HT = 256 it owns groups g and
// g + 8 (two separate sums of 16).  The row buffer is mapped in whole rounds of 256 rows (ensure_partials), so the
// addresses need no clamp: one base pointer, constant strides; rows beyond the grid are masked when they are summed.
// (A "load or 0.0" select on the runtime row count made hipcc branch around every load, cdna_hip_programming.md §5
// trap (c).)
constexpr int LM_GROUPS = 16;

struct LmLoads {
  double v[32];  // HT = 256: [0,16) group g, [16,32) group g + 8;  HT = 512: [0,16) group g
  double my_word;
  long long passes_before;
};

template <bool COHERENT, bool FIRST, int HT>
__device__ __forceinline__ void lm_issue_loads(const double* __restrict__ partials, const LmState* __restrict__ state,
                                               LmLoads& L) {
  static_assert(HT == 256 || HT == 512, "helper threads");
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  L.passes_before = 0;
  L.my_word = 0.0;
  if (threadIdx.x < HT) {  // wave-uniform
    if (!FIRST && threadIdx.x == 64) L.passes_before = state->n_evals;
    const int cw = threadIdx.x < LM_STATE_WORDS ? threadIdx.x : LM_STATE_WORDS - 1;
    L.my_word = reinterpret_cast<const double*>(state)[cw];
    const int cc = c < NACC ? c : NACC - 1;
    const double* base = partials + (size_t)g * NACC + cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) L.v[j] = load_partial<COHERENT>(base + (size_t)(LM_GROUPS * j) * NACC);
    if (HT == 256) {
#pragma unroll
      for (int j = 0; j < 16; ++j) L.v[16 + j] = load_partial<COHERENT>(base + (size_t)(8 + LM_GROUPS * j) * NACC);
    }
  }
}

// What the controller publishes for the other waves of the workgroup (LDS): the pose to evaluate next as rotation (row-major) +
// translation, and whether the solve goes on.
constexpr int LM_PUB_WORDS = 16;  // R[9], t[3], status (int in word 12)

__device__ __forceinline__ void lm_pub_pose(double* pub, const double* x) {  // called by one lane
  double R[9];
  quat_to_rot(x + 3, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) pub[i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) pub[9 + i] = x[i];
}
__device__ __forceinline__ int lm_pub_status(const double* pub) { return reinterpret_cast<const int*>(pub + 12)[0]; }

// Cycle stamps inside lm_advance_wave (debug build -DCLC_STAMPS, scripts/stamps_step.py); nothing otherwise.
#ifdef CLC_STAMPS
#define CLC_CK() do { ck[nck++] = clock64(); } while (0)
#else
#define CLC_CK() do {} while (0)
#endif

// ---------------------------------------------------------------------------------------
// lm_advance on a wavefront (the step kernel's controller)
// ---------------------------------------------------------------------------------------
// The serial controller (clc_lm.hpp, one lane, state in LDS) is a chain of dependent FP64 instructions and LDS round
// trips: 7 300 cycles = 3.0 us of every step_kernel launch, in every workgroup (scripts/stamps_step.py).  What the
// instructions cost when ONE wave runs them alone (scripts/probes/latency_probe.hip, cycles): dependent FMA 4-6, a
// value through v_readlane into the next FMA 24-31, IEEE division 72-98, IEEE sqrt 108-146, dependent LDS read 72-93,
// compare + select (or branch) 40-50.  So the controller here is written for a short critical path and few branches:
//   * the state is read from LDS once, up front, in one batch; everything scalar (pose, costs, radius, the triangular
//     solves' running values) is computed redundantly by all lanes ("uniform") from broadcast LDS reads;
//   * lane i < 6 owns row i of the Gauss-Newton matrix: scaling, damping, the factorisation's column updates and the
//     matrix-vector product of the model cost change are one instruction for all rows; the Cholesky factorisation is
//     right-looking (column j scaled, then subtracted from the columns to its right) with the diagonal in its own
//     register — per element the same subtractions in the same order as the left-looking serial loop — and the only
//     values that cross lanes are the pivots, the column entries, the forward substitution's z and the gradient
//     (v_readlane);
//   * the trust-region step is computed BEFORE the convergence tests that may make it unnecessary, so that the two
//     Plus operations of an iteration — Plus(x, -g) for the projected gradient norm and Plus(x, step) for the
//     candidate — run as one instruction stream in lanes 0 and 1; tolerance tests, acceptance and the radius update are
//     selects, not branches.  Nothing of the speculative step is committed unless the serial controller would have
//     computed it.  A step that turns out invalid (rare) is handed to the serial loop (lm_iterate);
//   * the candidate and the status are published first, the workgroup's waves meet at ONE barrier (inside this
//     function for the calling wave, in lm_tail_after_barrier for the others) and the rest of the state is written
//     back behind it, while the other waves already stream.
// Every expression keeps the operand order and the fused multiply-adds of clc_lm.hpp / clc_math.hpp: pose, summary and
// iteration trace of a solve are BIT-IDENTICAL to lm_advance<Se3Manifold>'s (the [evaluation, lm_kernel] launch pair
// still runs the serial controller: test_step_kernel_solve_matches_two_kernel_path, the randomized problem test and
// the invalid-step test compare the two bit for bit).  The LM state agrees as well while a solve runs; after a
// termination by parameter / function tolerance the fields that are not outputs (x, g, H, radius) hold the rejected
// pass instead of the last accepted one — nothing reads them any more.
// Called by all 64 lanes of one wave; `tot` (LDS): the 28 totals of this pass (H 0..20, g 21..26, cost sum 27), written
// by this same wave (LDS operations of one wave execute in program order; the caller fences).
// LEAN (the resident batched kernel, clc_resident.hpp): the same arithmetic with a small register footprint.  The calling
// wave keeps ~100 VGPRs of scan points alive across the controller there, and this function, written for a short critical
// path, holds ~180 VGPRs (x, x_eval, the column scales, the scaled matrix and the bookkeeping all stay in registers from
// the first batch of LDS reads to the write-back).  LEAN stores what is final as soon as it is known — the state's home
// is LDS anyway —, parks the scaled system in `park` and reads x, the scales, the gradient and the scaled system back
// right before the model cost change and the two Plus that need them: three more LDS round trips (~300 cycles of a
// controller that overlaps the co-resident problem's streaming there), ~60 VGPRs less.
template <bool FIRST, bool LEAN = false>
__device__ __forceinline__ void lm_advance_wave(LmState& s, const clc_options& o, clc_iteration* __restrict__ trace,
                                                const int trace_cap, const double* tot, double* park, const int lane,
                                                unsigned long long* stamp_row = nullptr /* debug builds */) {
  // `park`: LDS nobody else touches; 32 doubles in, room for the serial controller's temporaries (LmScratch) on the
  // invalid-step path — held in registers they made hipcc spill the whole kernel's controller.
  // FIRST: the pass at the start point (state fresh from lm_init, phase 0); otherwise the pass at a candidate (phase 1 —
  // the only other phase a running solve can be in).  The caller has checked that the solve is still running.
  // Written without early exits and with selects instead of branches wherever both sides are cheap: a compare feeding a
  // branch or a select costs a single wave 40-50 cycles (latency_probe), and there were ~35 of them.
  constexpr int NP = 6, NA = 7;
  constexpr double DMAX = 1.7976931348623157e308;
#ifdef CLC_STAMPS
  long long ck[12];
  int nck = 0;
#endif
  CLC_CK();
  const unsigned i6 = lane < NP ? (unsigned)lane : NP - 1u;
  // packed upper triangle: index of (a, b), a <= b, is a * (2 NP - 1 - a) / 2 + b
  const unsigned rowbase = (i6 * (2u * NP - 1u - i6)) >> 1;
  const unsigned dgi = rowbase + i6;  // H[i][i]
  unsigned hidx[NP];
#pragma unroll
  for (unsigned b = 0; b < NP; ++b) hidx[b] = b < i6 ? ((b * (2u * NP - 1u - b)) >> 1) + i6 : rowbase + b;
  // ---- everything that comes from LDS, in one batch: this pass ...
  const double cost_acc = tot[27];
  double g = tot[21 + i6], Hd = tot[dgi], Hrow[NP];  // lane i: g[i], H[i][i], row i of H
#pragma unroll
  for (int b = 0; b < NP; ++b) Hrow[b] = tot[hidx[b]];
  // ... and the state
  const int iteration = s.iteration, n_invalid_in = s.n_invalid, reuse_in = s.reuse_diagonal;
  const int n_succ_in = s.num_successful, n_unsucc_in = s.num_unsuccessful, n_trace_in = s.n_trace;
  const long long n_evals = s.n_evals + 1;
  double x[NA], xe[NA], sc[NP];
#pragma unroll
  for (int i = 0; i < NA; ++i) { x[i] = s.x[i]; xe[i] = s.x_eval[i]; }
#pragma unroll
  for (int b = 0; b < NP; ++b) sc[b] = s.scale[b];
  double x_norm = s.x_norm, x_cost = s.x_cost, minimum_cost = s.minimum_cost, initial_cost = s.initial_cost;
  double min_iter_cost = s.min_iter_cost, radius = s.radius, dfac = s.decrease_factor, gmax = s.gmax;
  const double mcc = s.model_cost_change;
  double scale = s.scale[i6];
  const double diag = s.diag[i6];
  // every load above is issued before the first value is consumed: one LDS round trip, not three
  __builtin_amdgcn_sched_barrier(0);
  const double cost_e = finalize_cost(cost_acc, o.use_loss != 0, o.loss_scale_factor);
  const bool finite_eval = fabs(cost_e) <= DMAX;
  // ---- the pass just evaluated: early terminations (flags; no output of the solve changes then), acceptance ----
  int early = CLC_RUNNING;  // termination before the iteration is recorded
  bool success = true;
  int reuse = reuse_in;
  int it_iteration = 0;
  double it_cost, it_cost_change = 0.0, it_step_norm = 0.0, it_rel = 0.0;
  if (FIRST) {
    // ---- IterationZero ----
    early = finite_eval ? CLC_RUNNING : CLC_FAILURE;
    x_cost = cost_e;
    if (o.jacobi_scaling) {  // once per solve: computed by the lane that owns the column, broadcast through LDS
      scale = 1.0 / (1.0 + sqrt(Hd));
      if (lane < NP) s.scale[lane] = scale;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int b = 0; b < NP; ++b) sc[b] = s.scale[b];
    }
    initial_cost = x_cost;
    min_iter_cost = x_cost;
    it_cost = x_cost;
  } else {
    it_iteration = iteration;
    const double candidate_cost = finite_eval ? cost_e : DMAX;
    // ---- ParameterToleranceReached, FunctionToleranceReached ----
    double sn = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) sn += (x[i] - xe[i]) * (x[i] - xe[i]);
    it_step_norm = sqrt_pos(sn);
    it_cost_change = x_cost - candidate_cost;
    const bool par_tol = it_step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance);
    const bool fun_tol = fabs(it_cost_change) <= o.function_tolerance * x_cost;
    early = par_tol ? CLC_CONVERGENCE_PARAMETER : (fun_tol ? CLC_CONVERGENCE_FUNCTION : CLC_RUNNING);
    // ---- IsStepSuccessful; HandleSuccessfulStep / HandleUnsuccessfulStep as selects ----
    it_rel = it_cost_change * rcp_pos_safe(mcc);
    success = it_rel > o.min_relative_decrease;
    const double q = 2.0 * it_rel - 1.0;  // StepAccepted
    double den = 1.0 - q * q * q;
    den = den > (1.0 / 3.0) ? den : (1.0 / 3.0);
    double r_acc = radius * rcp_pos(den);
    r_acc = r_acc < o.max_trust_region_radius ? r_acc : o.max_trust_region_radius;
    const double r_rej = radius * rcp_pos(dfac);  // StepRejected: radius / decrease_factor, exact (a power of two)
    radius = success ? r_acc : r_rej;
    dfac = success ? 2.0 : dfac * 2.0;
    reuse = success ? 0 : 1;
    it_cost = candidate_cost;
    double xn2 = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) xn2 += xe[i] * xe[i];
    const double xe_norm = sqrt_pos(xn2);
    x_norm = success ? xe_norm : x_norm;
    x_cost = success ? candidate_cost : x_cost;
#pragma unroll
    for (int i = 0; i < NA; ++i) x[i] = success ? xe[i] : x[i];
    if (!success) {
      // a rejected step (the minority) is recomputed from the Gauss-Newton system at x: one more LDS round trip on this
      // branch instead of 8 more doubles loaded and selected on every pass (the kernel has no registers to spare)
      g = s.g[i6];
      Hd = s.H[dgi];
#pragma unroll
      for (int b = 0; b < NP; ++b) Hrow[b] = s.H[hidx[b]];
    }
  }
  const int it_succ = success ? 1 : 0;
  const int n_succ = n_succ_in + it_succ, n_unsucc = n_unsucc_in + (1 - it_succ);
  const bool xout_dirty = success && x_cost < minimum_cost;
  minimum_cost = xout_dirty ? x_cost : minimum_cost;
  min_iter_cost = it_cost < min_iter_cost ? it_cost : min_iter_cost;
  // What is final already and not part of a solve's outputs goes back to LDS now, unconditionally (after an early
  // termination nobody reads it): the stores overlap the solve below and free their registers — with everything held
  // until the write-back the kernel spilled.  The trace record's fields wait in `park`.
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) s.x[i] = x[i];
    s.x_norm = x_norm;
    s.x_cost = x_cost;
    s.decrease_factor = dfac;
    s.radius = radius;
    park[0] = it_cost;
    park[1] = it_cost_change;
    park[2] = it_step_norm;
    park[3] = it_rel;
  }
  if (lane < NP) {  // (a rejected step stores back what it loaded)
    s.g[lane] = g;
    s.scale[lane] = scale;
#pragma unroll
    for (int b = 0; b < NP; ++b)
      if (b >= lane) s.H[hidx[b]] = Hrow[b];
  }
  if (LEAN) {
    // the part of write_back() that is known by now (outputs of the solve: only if the pass did not terminate it)
    if (lane == 0 && early == CLC_RUNNING) {
      s.num_successful = n_succ;
      s.num_unsuccessful = n_unsucc;
      s.n_trace = n_trace_in + 1;
      s.n_evals = n_evals;
      s.initial_cost = initial_cost;
      s.minimum_cost = minimum_cost;
      s.min_iter_cost = min_iter_cost;
      if (xout_dirty) {
#pragma unroll
        for (int i = 0; i < NA; ++i) s.x_out[i] = x[i];
      }
    }
  }
  CLC_CK();
  // ---- lm_compute_step, ahead of the tests that may make it unnecessary (committed after them) ----
  double Hs[NP], A[NP];
#pragma unroll
  for (int b = 0; b < NP; ++b) {
    Hs[b] = Hrow[b] * (scale * sc[b]);  // entry b == lane is the diagonal, H[i][i] * (scale[i] * scale[i])
    A[b] = Hs[b];                       // working copy for the factorisation; its diagonal entry is not used (Ad)
  }
  const double gs = g * scale;
  const double Hds = Hd * (scale * scale);
  double dcl = Hds;
  dcl = dcl > o.min_lm_diagonal ? dcl : o.min_lm_diagonal;
  dcl = dcl < o.max_lm_diagonal ? dcl : o.max_lm_diagonal;
  const double diag_n = reuse ? diag : dcl;
  const double inv_radius = rcp_pos(radius);
  double Ad = Hds + diag_n * inv_radius;  // lane i: the damped diagonal entry, updated in place by the factorisation
  if (LEAN) {  // row i of the scaled system waits in LDS for the model cost change (park[8 + 8 i ...]: Hs[0..5], gs)
    if (lane < NP) {
#pragma unroll
      for (int b = 0; b < NP; ++b) park[8 + 8 * lane + b] = Hs[b];
      park[8 + 8 * lane + 6] = gs;
    }
  }
  CLC_CK();
  // Cholesky, right-looking: Lc[j] = column j of L (lane i: L[i][j], meaningful for i > j), inv[j] = 1 / L[j][j]
  // (a pivot <= 0 or NaN makes its reciprocal square root, and with it y[j], NaN: the finiteness test of y below is the
  // serial code's two tests in one)
  double inv[NP], Lc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const double d = readlane_d(Ad, j);
    inv[j] = rsqrt_pos(d);
    Lc[j] = A[j] * inv[j];
    Ad -= Lc[j] * Lc[j];
#pragma unroll
    for (int k = j + 1; k < NP; ++k) A[k] -= Lc[j] * readlane_d(Lc[j], k);
  }
  CLC_CK();
  // L z = gs: lane i carries row i's running value; z[k] is final after k subtractions
  double z[NP], run = gs;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    z[k] = readlane_d(run * inv[k], k);
    run -= Lc[k] * z[k];
  }
  // L^T y = z, uniform, subtractions in ascending k like the serial loop
  double y[NP];
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double acc = z[i];
#pragma unroll
    for (int k = i + 1; k < NP; ++k) acc -= readlane_d(Lc[i], k) * y[k];
    y[i] = acc * inv[i];
  }
  double fin = 0.0;  // 0 * y is (+-)0 for finite y and NaN otherwise
#pragma unroll
  for (int c = 0; c < NP; ++c) fin = fma(y[c], 0.0, fin);
  const bool ok = fin == 0.0;
  CLC_CK();
  double step_n[NP], sg = 0.0, shs = 0.0, row = 0.0;
#pragma unroll
  for (int a = 0; a < NP; ++a) step_n[a] = -y[a];
  double Hs2[NP], gs2 = gs;
#pragma unroll
  for (int b = 0; b < NP; ++b) Hs2[b] = Hs[b];
  if (LEAN) {  // back from LDS (same wave: program order), not hoisted above the factorisation
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int b = 0; b < NP; ++b) Hs2[b] = park[8 + 8 * i6 + b];
    gs2 = park[8 + 8 * i6 + 6];
  }
#pragma unroll
  for (int b = 0; b < NP; ++b) row += Hs2[b] * step_n[b];  // lane a: (Hs step)[a]
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    sg += step_n[a] * readlane_d(gs2, a);
    shs += step_n[a] * readlane_d(row, a);
  }
  const double mcc_n = -(sg + 0.5 * shs);
  const bool step_ok = ok && mcc_n > 0.0;
  CLC_CK();
  // ---- Plus: lane 0 the projected gradient (after a change of x / g), lane 1 the candidate ----
  double cand[NA];
  {
    double g3 = g, sc3[NP], x3[NA];
#pragma unroll
    for (int c = 0; c < NP; ++c) sc3[c] = sc[c];
#pragma unroll
    for (int i = 0; i < NA; ++i) x3[i] = x[i];
    if (LEAN) {  // x, g and the column scales as stored above (lane 0 / lanes < NP wrote them; every lane reads)
      __builtin_amdgcn_sched_barrier(0);
      g3 = s.g[i6];
#pragma unroll
      for (int c = 0; c < NP; ++c) sc3[c] = s.scale[c];
#pragma unroll
      for (int i = 0; i < NA; ++i) x3[i] = s.x[i];
    }
    double dlt[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const double ng = -readlane_d(g3, c);
      const double dc = step_n[c] * sc3[c];  // undo column scaling
      dlt[c] = lane == 1 ? dc : ng;
    }
    pose_plus_rcp(x3, dlt, cand);
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < NA; ++i) m = fmax(m, fabs(x3[i] - cand[i]));
    const double gnew = readlane_d(m, 0);
    gmax = success ? gnew : gmax;
  }
  const double it_gmax = gmax;
  CLC_CK();
  // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
  const bool cap_hit = it_iteration >= o.max_num_iterations;
  const bool grad_tol = success && it_gmax <= o.gradient_tolerance;
  const bool rad_tol = radius <= o.min_trust_region_radius;
  const int status = cap_hit ? CLC_NO_CONVERGENCE : (grad_tol ? CLC_CONVERGENCE_GRADIENT : (rad_tol ? CLC_CONVERGENCE_RADIUS : CLC_RUNNING));
  // next iteration: the step computed above is the one the serial controller computes at this point
  const bool cont = status == CLC_RUNNING;
  const bool candidate_ready = cont && step_ok;
  const bool invalid_step = cont && !step_ok;  // rare: handed to the serial loop below, once the state is back in LDS
  const bool step_dirty = cont && ok;  // factorisation and solve went through: step and model cost change are stored
  CLC_CK();
  // ---- what the other waves wait for — the next point to evaluate and whether the solve goes on — first; they leave
  // for their rows at the barrier below while this wave writes the rest of the state back (the trace record, ~45 LDS
  // words, the bookkeeping's selects: ~0.7 us that used to sit in front of every wave's first row) ----
  auto write_back = [&]() {
    if (lane == 0) {
      if (trace != nullptr && n_trace_in < trace_cap) {
        clc_iteration it;
        it.iteration = it_iteration;
        it.step_is_valid = 1;
        it.step_is_successful = it_succ;
        it.pad_ = 0;
        it.cost = park[0];
        it.cost_change = park[1];
        it.gradient_max_norm = it_gmax;
        it.step_norm = park[2];
        it.relative_decrease = park[3];
        it.trust_region_radius = radius;
        trace[n_trace_in] = it;
      }
      s.phase = candidate_ready ? 1 : (FIRST ? 0 : 1);
      s.iteration = candidate_ready ? it_iteration + 1 : iteration;
      s.n_invalid = candidate_ready ? 0 : n_invalid_in;
      s.reuse_diagonal = cont ? 1 : reuse;
      s.gmax = gmax;
      if (!LEAN) {  // (LEAN: stored as soon as they were known)
        s.num_successful = n_succ;
        s.num_unsuccessful = n_unsucc;
        s.n_trace = n_trace_in + 1;
        s.n_evals = n_evals;
        s.initial_cost = initial_cost;
        s.minimum_cost = minimum_cost;
        s.min_iter_cost = min_iter_cost;
        if (xout_dirty) {
#pragma unroll
          for (int i = 0; i < NA; ++i) s.x_out[i] = x[i];
        }
      }
      if (step_dirty) {
#pragma unroll
        for (int a = 0; a < NP; ++a) s.step[a] = step_n[a];
        s.model_cost_change = mcc_n;
      }
    }
    if (lane < NP) s.diag[lane] = cont ? diag_n : diag;
  };
  const bool slow_path = early == CLC_RUNNING && invalid_step;
  if (lane == 0) {
    s.status = early != CLC_RUNNING ? early : status;
    if (early != CLC_RUNNING) s.n_evals = n_evals;  // terminated by a tolerance on the pass itself: nothing else changes
  }
  if (lane == 1 && early == CLC_RUNNING && candidate_ready) {
#pragma unroll
    for (int i = 0; i < NA; ++i) s.x_eval[i] = cand[i];
  }
  if (slow_path) {
    // rare: HandleInvalidStep and whatever follows it (shrunken radius, another step, ...) on the serial controller,
    // which works on the complete state in LDS and decides status and candidate — before anybody leaves
    write_back();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      clc_iteration it;
      it.iteration = it_iteration + 1; it.step_is_valid = 0; it.step_is_successful = 0; it.pad_ = 0;
      it.cost = 0.0; it.cost_change = 0.0; it.gradient_max_norm = 0.0; it.step_norm = 0.0;
      it.relative_decrease = 0.0; it.trust_region_radius = 0.0;
      LmScratch& w = *reinterpret_cast<LmScratch*>(park + 32);  // temporaries in LDS: this path must not cost registers
      lm_iterate(s, w, o, trace, trace_cap, it, true);
    }
  }
  CLC_CK();
  __syncthreads();  // pairs with the barrier the other waves of the workgroup execute in lm_tail_after_barrier
  if (early == CLC_RUNNING && !slow_path) write_back();
  CLC_CK();
#ifdef CLC_STAMPS
  if (stamp_row && lane == 0 && nck == 10) {
    unsigned long long packed0 = 0, packed1 = 0;
    for (int i = 0; i < 4; ++i) packed0 |= (unsigned long long)((ck[i + 1] - ck[i]) & 0xFFFF) << (16 * i);
    for (int i = 0; i < 4; ++i) packed1 |= (unsigned long long)((ck[i + 5] - ck[i + 4]) & 0xFFFF) << (16 * i);
    stamp_row[15] = packed0;
    stamp_row[6] = packed1;
  }
#endif
}

// `state` is where the LM state is read from; it is written back to `state_out` (nullptr: not at all — the
// step kernel's non-leading workgroups run the controller redundantly and keep the result in LDS only).
// CHECK_STATUS: the staged state is inspected before anything is consumed or published; if the solve had already
// terminated the function returns false right after the first barrier (state staged in LDS, nothing else done).
// `red` is [LM_GROUPS][32] doubles of LDS.
// Phase A of the tail: this thread's share of the row sums -> LDS, its word of the LM state -> LDS.  No barrier: a caller
// may issue further loads (the step kernel: its first rows of points) between this and lm_tail_finish.
template <bool COHERENT, int HT>
__device__ __forceinline__ void lm_tail_sums(const double* __restrict__ partials, int n_blocks, double (*red)[32],
                                             double* sh_state, const LmLoads& L, long long* stamps /* nullable: [2] */,
                                             const int n_stage_words = LM_STATE_WORDS) {
  static_assert(LM_STATE_WORDS <= 256, "one state word per thread");
  const bool helper = threadIdx.x < HT;
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  if (helper) {  // wave-uniform
    const int cc = c < NACC ? c : NACC - 1;
    if (stamps && threadIdx.x == 0) stamps[0] = clock64();
    if ((int)threadIdx.x < n_stage_words) sh_state[threadIdx.x] = L.my_word;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s0 += (c < NACC && g + LM_GROUPS * j < n_blocks) ? L.v[j] : 0.0;
    if (HT == 256) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s1 += (c < NACC && g + 8 + LM_GROUPS * j < n_blocks) ? L.v[16 + j] : 0.0;
    }
    if (stamps && threadIdx.x == 0) stamps[1] = clock64();
    for (int b0 = 256; b0 < n_blocks; b0 += 256) {  // grids beyond 256 workgroups: further rounds of 256 rows
      const double* bb = partials + (size_t)(b0 + g) * NACC + cc;
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = load_partial<COHERENT>(bb + (size_t)(LM_GROUPS * j) * NACC);
#pragma unroll
      for (int j = 0; j < 16; ++j) s0 += (c < NACC && b0 + g + LM_GROUPS * j < n_blocks) ? v[j] : 0.0;
      if (HT == 256) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = load_partial<COHERENT>(bb + (size_t)(8 + LM_GROUPS * j) * NACC);
#pragma unroll
        for (int j = 0; j < 16; ++j) s1 += (c < NACC && b0 + g + 8 + LM_GROUPS * j < n_blocks) ? v[j] : 0.0;
      }
    }
    red[g][c] = s0;
    if (HT == 256) red[g + 8][c] = s1;
  }
}

template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS, bool WAVE = false>
__device__ __forceinline__ bool lm_tail_after_barrier(const LmState* __restrict__ state, LmState* __restrict__ state_out,
                                                      const clc_options& opt, clc_iteration* __restrict__ trace, int trace_cap,
                                                      HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                                      const long long c0, const Pose7* init_pose, const LmLoads& L,
                                                      unsigned long long* stamp_row = nullptr);

// Phase B: barrier, ordered combination of the 16 row groups, LM controller, publication.
template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS = false>
__device__ __forceinline__ bool lm_tail_finish(const LmState* __restrict__ state, LmState* __restrict__ state_out,
                                               const clc_options& opt,
                                               clc_iteration* __restrict__ trace, int trace_cap,
                                               HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                               const long long c0, const Pose7* init_pose, const LmLoads& L) {
  __syncthreads();
  return lm_tail_after_barrier<COHERENT, FIRST, HT, CHECK_STATUS>(state, state_out, opt, trace, trace_cap, mailbox, red, sh_state,
                                                                  c0, init_pose, L);
}

// ... and what follows the barrier (the step kernel reads its options from LDS between the two).
template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS, bool WAVE>
__device__ __forceinline__ bool lm_tail_after_barrier(const LmState* __restrict__ /*state*/, LmState* __restrict__ state_out,
                                                      const clc_options& opt,
                                                      clc_iteration* __restrict__ trace, int trace_cap,
                                                      HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                                      const long long c0, const Pose7* init_pose, const LmLoads& L,
                                                      unsigned long long* stamp_row /* debug builds; nullptr otherwise */) {
  // Called by every thread of the workgroup (it contains a barrier).
  // Progress for the host's launch-ahead metering is published EARLY, by the first lane of wave 1
  // (not the controller's wave): the ~1.5 us a store to pinned host memory needs to be
  // acknowledged then overlaps the controller instead of delaying the end of the launch.
  const long long passes_before = L.passes_before;
  if (CHECK_STATUS && !FIRST && reinterpret_cast<const LmState*>(sh_state)->status != CLC_RUNNING) return false;
  // the 16 row groups are combined in order by 28 lanes in parallel (one column each): done by the controller's
  // lane alone this was hundreds of serial FP64 adds behind LDS reads, ~0.4 us of the launch
  if (threadIdx.x < 32) {
    double t = 0.0;
#pragma unroll
    for (int gg = 0; gg < LM_GROUPS; ++gg) t += red[gg][threadIdx.x];
    red[0][threadIdx.x] = t;  // row 0 now holds the totals
    if (opt.profile_events && threadIdx.x == 0 && mailbox != nullptr) mailbox->prof[6] = clock64();
  }
  // no workgroup barrier here: the 28 lanes above and the controller's lane below are the same wave, whose LDS
  // operations execute in program order; the other waves go straight to the barrier at the end
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (threadIdx.x == 64 && mailbox != nullptr)
    __hip_atomic_store(&mailbox->n_done, (int32_t)(passes_before + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (WAVE) {
    // The controller on all 64 lanes of wave 0 (lm_advance_wave).  It publishes the next point / the status, meets the
    // other waves at ONE workgroup barrier — from which they leave for their rows — and only then writes the rest of
    // the state back, publishes a termination to the host and (leading workgroup) copies the state to device memory.
    if (threadIdx.x < 64) {
      const long long c1 = clock64();
      if (stamp_row && threadIdx.x == 0) stamp_row[11] = wall_clock64();
      LmState& st = *reinterpret_cast<LmState*>(sh_state);
      if (FIRST) {  // first iteration of a solve: nothing to load
        if (threadIdx.x == 0) lm_init(st, opt, init_pose->v);
        // lane 0's stores must be visible to the other lanes' loads below: without the fences hipcc is free to hoist those
        // loads above the (for them never executed) stores — they then read the previous solve's terminated state
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      static_assert(sizeof(LmScratch) <= (LM_GROUPS - 2) * 32 * sizeof(double), "LmScratch fits rows 2.. of red");
      lm_advance_wave<FIRST>(st, opt, trace, trace_cap, &red[0][0], &red[1][0], (int)threadIdx.x, stamp_row);  // contains the barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (threadIdx.x == 0) {
        const long long c2 = clock64();
        if (stamp_row) { stamp_row[12] = wall_clock64(); stamp_row[13] = (unsigned long long)c1; stamp_row[14] = (unsigned long long)c2; }
        if (mailbox != nullptr) {
          if (opt.profile_events) { mailbox->prof[0] = c0; mailbox->prof[1] = c1; mailbox->prof[2] = c2; }
          if (st.status != CLC_RUNNING) {
            // termination: payload first, then system-scope release stores of the flags
            clc_summary sm;
            lm_fill_summary(st, sm);
            sm.solve_ms = 0.0;
            sm.eval_kernel_ms = 0.0;
            sm.eval_kernel_launches = 0;
            mailbox->summary = sm;
            for (int i = 0; i < 7; ++i) mailbox->pose[i] = st.x_out[i];
            __hip_atomic_store(&mailbox->n_done, (int32_t)st.n_evals, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mailbox->status, st.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          if (opt.profile_events) mailbox->prof[3] = clock64();
        }
      }
      if (state_out != nullptr) {
        for (int w = threadIdx.x; w < LM_STATE_WORDS; w += 64) reinterpret_cast<double*>(state_out)[w] = sh_state[w];
      }
    } else {
      __syncthreads();  // the barrier inside lm_advance_wave
    }
    return true;
  }
  if (threadIdx.x == 0) {
    const long long c1 = clock64();
    LmState& st = *reinterpret_cast<LmState*>(sh_state);
    if (stamp_row) stamp_row[11] = wall_clock64();
    double tot[NACC];
#pragma unroll
    for (int cc = 0; cc < NACC; ++cc) tot[cc] = red[0][cc];
    // The LM state is used in place in LDS: copied into registers and back it cost 256 VGPRs + 48 AGPRs
    // (occupancy 1 for the fused kernel); in place 148-162, at the same controller time.
    if (FIRST) lm_init(st, opt, init_pose->v);  // first iteration of a solve: nothing to load
    LmScratch scratch;
    lm_advance(st, scratch, opt, trace, trace_cap,
               finalize_cost(tot[27], opt.use_loss != 0, opt.loss_scale_factor), tot + 21, tot);
    const long long c2 = clock64();
    if (stamp_row) { stamp_row[12] = wall_clock64(); stamp_row[13] = (unsigned long long)c1; stamp_row[14] = (unsigned long long)c2; }
    if (mailbox != nullptr) {
      if (opt.profile_events) { mailbox->prof[0] = c0; mailbox->prof[1] = c1; mailbox->prof[2] = c2; }
      if (st.status != CLC_RUNNING) {
        // termination: payload first, then system-scope release stores of the flags
        clc_summary sm;
        lm_fill_summary(st, sm);
        sm.solve_ms = 0.0;
        sm.eval_kernel_ms = 0.0;
        sm.eval_kernel_launches = 0;
        mailbox->summary = sm;
        for (int i = 0; i < 7; ++i) mailbox->pose[i] = st.x_out[i];
        __hip_atomic_store(&mailbox->n_done, (int32_t)st.n_evals, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox->status, st.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (opt.profile_events) mailbox->prof[3] = clock64();
    }
  }
  __syncthreads();
  if (state_out != nullptr && threadIdx.x < LM_STATE_WORDS)
    reinterpret_cast<double*>(state_out)[threadIdx.x] = sh_state[threadIdx.x];
  return true;
}

// Both phases back to back (lm_kernel, eval_lm_kernel).
template <bool COHERENT, bool FIRST, int HT, bool CHECK_STATUS = false>
__device__ __forceinline__ bool lm_tail(const double* __restrict__ partials, int n_blocks,
                                        const LmState* __restrict__ state, LmState* __restrict__ state_out,
                                        const clc_options& opt,
                                        clc_iteration* __restrict__ trace, int trace_cap,
                                        HostMailbox* mailbox, double (*red)[32], double* sh_state,
                                        const long long c0, const Pose7* init_pose, LmLoads& L) {
  lm_tail_sums<COHERENT, HT>(partials, n_blocks, red, sh_state, L,
                             (opt.profile_events && mailbox != nullptr) ? &mailbox->prof[4] : nullptr);
  return lm_tail_finish<COHERENT, FIRST, HT, CHECK_STATUS>(state, state_out, opt, trace, trace_cap, mailbox, red, sh_state, c0,
                                                           init_pose, L);
}

}  // namespace clc
