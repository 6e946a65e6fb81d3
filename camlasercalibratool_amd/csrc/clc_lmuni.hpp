// clc_lmuni.hpp — the Levenberg-Marquardt controller of the on-chip solves as WAVE-UNIFORM arithmetic on a state that lives in the
// registers of ONE wave for the whole solve.
//
// ceres::Solve (src/LaseCamCalCeres.cpp:301-307) keeps its trust-region state in the minimiser object for the whole solve.  Between two
// evaluation passes of the on-chip kernels the controller is the critical path of the whole CU — the other waves wait for the next
// pose — so what counts is the latency of ONE wave running alone.  Measured on MI355X (scripts/probes/dpp_probe.hip,
// xlane_probe.hip; cycles per dependent step of a lone wave): FP64 FMA 4.1, compare + select 44, v_rsq / v_rcp_f64 20,
// v_fmac_f64_dpp row_newbcast 13-16, v_mov_b64_dpp + FMA 20, readlane + FMA 24, an LDS read or write ~20 per instruction (72 when
// the next address depends on it), IEEE division 73, IEEE sqrt 97.  So:
//   * a value that crosses lanes or LDS costs as much as four to five FMAs.  lm_advance_wave (clc_controller.hpp, round 2: lane i owns
//     row i of the 6 x 6 system, 61 v_readlane broadcasts, state in LDS) and its round-4 lane-packed DPP successor (clc_lmregs.hpp)
//     both spend 4 800-5 500 cycles per LM iteration.  Here every lane carries the COMPLETE state in plain registers and runs the
//     serial algorithm of clc_lm.hpp redundantly: ~330 FP64 instructions, no cross-lane traffic in the linear algebra at all;
//   * the wave that runs it holds no scan points (the cooperative kernel dedicates a fifth wave to exchange + controller; in the
//     single-workgroup kernel the lanes of wave 0 hold 4 points each), so ~75 doubles of state cost nothing elsewhere;
//   * only what the NEXT POSE needs is on the chain (lmu_post): acceptance of the pass, the radius, the scaled damped system, its
//     Cholesky solve, Plus, the rotation — ~400 instructions.  Everything else of the iteration runs in lmu_pre(), which the
//     cooperative kernel executes while the row exchange of the following pass is in flight: Ceres' `parameters_`, the
//     projected-gradient norm at the new x, the iteration record, the gradient-tolerance test, the model cost change and the
//     validity of the step under evaluation (finite, positive model cost change), step norm / parameter tolerance, the candidate's
//     norm, 1 / model_cost_change, the radius a rejection would leave.  A gradient-tolerance stop or an invalid step found there
//     means the pass under way is discarded by the next lmu_post (not counted as an evaluation): one wasted pass at the end of a
//     solve that stops on its gradient, one per invalid step (rare) — the price of ~2 500 cycles less in every other pass;
//   * the 28 totals arrive through LDS (one write, 14 broadcast 16-byte reads), the candidate is published as ROTATION +
//     TRANSLATION + status (7 16-byte LDS writes; the pass waves start with 7 reads instead of quaternion -> rotation + 24
//     v_readfirstlane each); no other value crosses lanes.
// The Gauss-Newton system at x (needed again only when a step is rejected) is not copied at all: the totals of the passes alternate
// between two LDS buffers and the state remembers which one belongs to x.
//
// Arithmetic: operand order and fused multiply-adds of clc_lm.hpp / clc_math.hpp (chol_solve's left-looking loops, unrolled), so for
// the same totals the decisions, the trace and the outputs are BIT-IDENTICAL to the serial controller's and to lm_advance_wave's —
// tests/test_gpu_lmuni.py compares whole solves of the single-workgroup kernel under the controllers, field by field.
// A step that turns out invalid shrinks the radius like a rejection, is recorded as an iteration of its own and is computed again from
// the same system — Ceres' HandleInvalidStep, on the same registers; of the LDS LmState only x_out and, at the end, the outputs are used.
#pragma once
#include "clc_controller.hpp"

namespace clc {

#ifdef CLC_STAMPS
// Debug build only (scripts/stamps_coop.py): shader-clock stamps inside lmu_post of workgroup 8 (non-first passes; the last one wins).
static __device__ long long clc_lmu_ck[16];
#define LMU_CK(i) do { if (blockIdx.x == 8 && lane == 0 && !first) clc_lmu_ck[i] = clock64(); } while (0)
#else
#define LMU_CK(i) do {} while (0)
#endif

struct LmU {  // every member wave-uniform
  double x[7], xe[7];  // accepted iterate, point the next pass evaluates  (Ceres' `parameters_`, the lowest-cost accepted iterate, lives in the LDS state: lmu_pre / lmu_finish write it)
  double x_norm, x_cost, minimum_cost, initial_cost, min_iter_cost, radius, dfac, mcc, gmax;
  double scale[6], diag[6];
  double y[6];         // the solution of the damped system that produced xe (step = -y): lmu_pre derives the model cost change from it
  int status, iteration, n_invalid, reuse, n_succ, n_unsucc, n_trace, n_evals;
  int hx;  // which of the two totals buffers holds g, H at x
  bool xout_pending;  // x is the new lowest-cost iterate and has not been written to the LDS state yet
  bool last_success;  // the iteration lmu_post finalised last moved x (its projected-gradient norm is still to be taken)
  int deferred;       // what lmu_pre found out about the step under evaluation: 0 nothing, LMU_STOP_GRADIENT, LMU_INVALID_STEP
  // the record of the iteration lmu_post finalised last; its gradient norm and the trace entry are lmu_pre's / lmu_finish's
  bool rec_pending;
  int rec_slot, rec_iteration, rec_succ;
  double rec_cost, rec_cost_change, rec_step_norm, rec_rel, rec_radius;
  // lmu_pre(): what the next lmu_post() needs from the state alone
  double step_norm, xe_norm, inv_mcc, r_rej;
  bool par_tol;
};
constexpr int LMU_STOP_GRADIENT = 1, LMU_INVALID_STEP = 2;

__device__ __forceinline__ void lmu_init(LmU& S, LmState& st, const clc_options& o, const double* x0 /* wave-uniform */, const int lane) {
  if (lane == 0) {  // (Ceres' `parameters_` starts as the start point: a solve that fails in iteration zero returns it)
#pragma unroll
    for (int i = 0; i < 7; ++i) st.x_out[i] = x0[i];
  }
  S.status = CLC_RUNNING;
  S.iteration = 0;
  S.n_invalid = 0;
  S.reuse = 0;
  S.n_succ = 0;
  S.n_unsucc = 0;
  S.n_trace = 0;
  S.n_evals = 0;
  S.hx = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) S.x[i] = S.xe[i] = x0[i];
  S.x_norm = norm_n<7>(x0);
  S.x_cost = 0.0;
  S.minimum_cost = 1.7976931348623157e308;
  S.initial_cost = 0.0;
  S.min_iter_cost = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) { S.scale[i] = 1.0; S.diag[i] = 0.0; S.y[i] = 0.0; }
  S.radius = o.initial_trust_region_radius;
  S.dfac = 2.0;
  S.mcc = 0.0;
  S.gmax = 0.0;
  S.step_norm = S.xe_norm = S.inv_mcc = S.r_rej = 0.0;
  S.par_tol = false;
  S.xout_pending = false;
  S.last_success = false;
  S.deferred = 0;
  S.rec_pending = false;
  S.rec_slot = S.rec_iteration = S.rec_succ = 0;
  S.rec_cost = S.rec_cost_change = S.rec_step_norm = S.rec_rel = S.rec_radius = 0.0;
}

// x -> Ceres' `parameters_` in the LDS state when the last accepted step lowered the minimum (off the chain: lmu_pre, lmu_finish).
__device__ __forceinline__ void lmu_flush_xout(LmU& S, LmState& st, const int lane) {
  if (S.xout_pending && lane == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) st.x_out[i] = S.x[i];
  }
  S.xout_pending = false;
}

// 28 wave-uniform totals from an LDS buffer: 14 broadcast 16-byte reads (~20 cycles each for a wave that runs alone).
__device__ __forceinline__ void lmu_read_totals(const double* buf, double* T) {
  const lds_cv2d* tp = lds_opaque(buf);  // (a per-lane address: the reads stay ds_read_b128 issued HERE, in one batch)
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const v2d v = tp[i];
    T[2 * i] = v[0];
    T[2 * i + 1] = v[1];
  }
}

// ||x - Plus(x, -g)||_inf in the ambient space (Ceres' projected-gradient norm; gradient_max_norm of clc_lm.hpp).
__device__ __forceinline__ double lmu_gradient_norm(const double* x, const double* g) {
  double ng[6], proj[7];
#pragma unroll
  for (int c = 0; c < 6; ++c) ng[c] = -g[c];
  pose_plus_rcp(x, ng, proj);
  double m = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) m = fmax(m, fabs(x[i] - proj[i]));
  return m;
}

// Hs = diag(scale) H diag(scale) (packed upper triangle), gs = diag(scale) g: lm_compute_step's first lines.
__device__ __forceinline__ void lmu_scale_system(const double* T, const double* scale, double* Hs, double* gs) {
  int idx = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = a; b < 6; ++b) {
      Hs[idx] = T[idx] * (scale[a] * scale[b]);
      ++idx;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a) gs[a] = T[21 + a] * scale[a];
}

// The iteration record lmu_post left, completed by the gradient norm: the trace entry (device memory; the leading workgroup of a
// traced solve), by one lane.
__device__ __forceinline__ void lmu_write_record(LmU& S, clc_iteration* __restrict__ trace, const int trace_cap, const int lane) {
  if (S.rec_pending && trace != nullptr && S.rec_slot < trace_cap && lane == 0) {
    clc_iteration it;
    it.iteration = S.rec_iteration;
    it.step_is_valid = 1;
    it.step_is_successful = S.rec_succ;
    it.pad_ = 0;
    it.cost = S.rec_cost;
    it.cost_change = S.rec_cost_change;
    it.gradient_max_norm = S.gmax;
    it.step_norm = S.rec_step_norm;
    it.relative_decrease = S.rec_rel;
    it.trust_region_radius = S.rec_radius;
    trace[S.rec_slot] = it;
  }
  S.rec_pending = false;
}

// The part of FinalizeIterationAndCheckIfMinimizerCanContinue that lmu_post leaves behind: the projected-gradient norm at the new x
// (only after a successful step: otherwise x and g stand), the iteration record, and the gradient-tolerance test.  `Tx`: the totals at x.
// true: the gradient tolerance is met — the solve ends with THIS iteration.
__device__ __forceinline__ bool lmu_close_iteration(LmU& S, const clc_options& o, const double* Tx, clc_iteration* __restrict__ trace,
                                                    const int trace_cap, const int lane) {
  const double gnew = lmu_gradient_norm(S.x, Tx + 21);
  S.gmax = S.last_success ? gnew : S.gmax;
  lmu_write_record(S, trace, trace_cap, lane);
  const bool stop = S.last_success && S.gmax <= o.gradient_tolerance;
  S.last_success = false;
  return stop;
}

// Between two passes, from the state alone (phase 1: x_eval is a candidate) — the cooperative kernel runs this while the row exchange
// is in flight: Ceres' `parameters_`; the iteration lmu_post finalised is closed (gradient norm, record, gradient tolerance); the step
// that produced x_eval is checked (finite, positive model cost change: lm_compute_step's last lines) — a stop or an invalid step
// found here makes the next lmu_post discard the pass under way —; ParameterToleranceReached, the candidate's norm, the reciprocal of
// the model cost change and the radius a rejection would leave.  Same expressions as lm_advance.
__device__ __forceinline__ void lmu_pre(LmU& S, LmState& st, const clc_options& o, const double* tot2, clc_iteration* __restrict__ trace,
                                        const int trace_cap, const int lane) {
  constexpr int NP = 6;
  lmu_flush_xout(S, st, lane);
  double T[28];
  lmu_read_totals(tot2 + 32 * S.hx, T);  // g, H at x
  const bool stop = lmu_close_iteration(S, o, T, trace, trace_cap, lane);
  // model_cost_change = -(J step)^T (r + J step / 2) of the step under evaluation
  double Hs[21], gs[NP];
  lmu_scale_system(T, S.scale, Hs, gs);
  double fin = 0.0;  // 0 * y is (+-)0 for finite y and NaN otherwise (a pivot <= 0 or NaN makes its reciprocal square root, and y, NaN)
#pragma unroll
  for (int c = 0; c < NP; ++c) fin = fma(S.y[c], 0.0, fin);
  const bool ok = fin == 0.0;
  double step_n[NP], sg = 0.0, shs = 0.0;
#pragma unroll
  for (int a = 0; a < NP; ++a) step_n[a] = -S.y[a];
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    sg += step_n[a] * gs[a];
    double row = 0.0;
#pragma unroll
    for (int b = 0; b < NP; ++b) row += Hs[a <= b ? tri<NP>(a, b) : tri<NP>(b, a)] * step_n[b];
    shs += step_n[a] * row;
  }
  const double mcc_n = -(sg + 0.5 * shs);
  const bool step_ok = ok && mcc_n > 0.0;
  S.mcc = mcc_n;
  S.n_invalid = step_ok ? 0 : S.n_invalid;  // (lm_iterate: a valid step clears the count of consecutive invalid ones)
  S.deferred = stop ? LMU_STOP_GRADIENT : (step_ok ? 0 : LMU_INVALID_STEP);
  double sn = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) sn += (S.x[i] - S.xe[i]) * (S.x[i] - S.xe[i]);
  S.step_norm = sqrt_pos(sn);
  S.par_tol = S.step_norm <= o.parameter_tolerance * (S.x_norm + o.parameter_tolerance);
  double xn2 = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) xn2 += S.xe[i] * S.xe[i];
  S.xe_norm = sqrt_pos(xn2);
  S.inv_mcc = rcp_pos_safe(S.mcc);
  S.r_rej = S.radius * rcp_pos(S.dfac);  // StepRejected: radius / decrease_factor, exact (a power of two)
  // (pinned here: without it the backend sinks these pure computations into lmu_post, behind the exchange they are meant to overlap)
  int pt = S.par_tol ? 1 : 0;
  asm volatile("" : "+v"(S.step_norm), "+v"(S.xe_norm), "+v"(S.inv_mcc), "+v"(S.r_rej), "+v"(pt), "+v"(S.deferred), "+v"(S.gmax));
  S.par_tol = pt != 0;
}

// The pose the pass evaluates next, as rotation + translation + status, for the other waves (LDS, 16-byte writes by one lane).
__device__ __forceinline__ void lmu_publish(double* pub, const double* xe, const int status, const int lane) {
  double R[9];
  quat_to_rot(xe + 3, R);
  if (lane == 0) {
    v2d* p = reinterpret_cast<v2d*>(pub);
    v2d a;
    a[0] = R[0]; a[1] = R[1]; p[0] = a;
    a[0] = R[2]; a[1] = R[3]; p[1] = a;
    a[0] = R[4]; a[1] = R[5]; p[2] = a;
    a[0] = R[6]; a[1] = R[7]; p[3] = a;
    a[0] = R[8]; a[1] = xe[0]; p[4] = a;
    a[0] = xe[1]; a[1] = xe[2]; p[5] = a;
    reinterpret_cast<int*>(pub + 12)[0] = status;
  }
}
__device__ __forceinline__ void lmu_publish_status(double* pub, const int status, const int lane) {
  if (lane == 0) reinterpret_cast<int*>(pub + 12)[0] = status;
}

// Consume the 28 totals of the pass at x_eval — `tot2` (LDS): two buffers of 32 doubles, the pass's totals in buffer 1 - S.hx, written
// by this wave (H 0..20 packed upper triangle, g 21..26, cost sum 27) — and advance to the next evaluation request or to termination.
// first (wave-uniform, a run-time flag: two instantiations inside one pass loop cost the register allocator ~100 doubles of spills): the
// pass at the start point (IterationZero).  At return S.status / S.xe are final and `pub` holds the next pose + status.
// What is on the chain here is only what the next pose needs: acceptance of the pass, the radius, the scaled damped system, its
// Cholesky solve, Plus, the rotation.  The rest of the iteration is lmu_pre's.
__device__ __forceinline__ void lmu_post(const bool first, LmU& S, const clc_options& o, const double* tot2, double* pub,
                                         clc_iteration* __restrict__ trace, const int trace_cap, const int lane) {
  constexpr int NP = 6, NA = 7;
  constexpr double DMAX = 1.7976931348623157e308;
  LMU_CK(7);  // (slot 0 at the end: the last pass of a solve leaves at its tolerance test)
  double T[28];
  lmu_read_totals(tot2 + 32 * (1 - S.hx), T);
  if (__builtin_expect(S.deferred != 0, 0)) {  // (wave-uniform, rare) the pass under way was evaluated for nothing
    if (S.deferred == LMU_STOP_GRADIENT) {  // the previous iteration had met the gradient tolerance: the solve ended there
      S.status = CLC_CONVERGENCE_GRADIENT;
      lmu_publish_status(pub, S.status, lane);
      return;
    }
    // ---- HandleInvalidStep (StepIsInvalid == StepRejected(0)) for iteration S.iteration: its step was not finite, or its model cost
    // change not positive.  Recorded as an iteration of its own; the step is computed again from the same system. ----
    S.deferred = 0;
    S.n_invalid += 1;
    if (S.n_invalid >= o.max_num_consecutive_invalid_steps) {
      S.status = CLC_FAILURE;
      lmu_publish_status(pub, S.status, lane);
      return;
    }
    S.radius = S.radius * rcp_pos(S.dfac);  // radius / decrease_factor, exact (a power of two)
    S.dfac = S.dfac * 2.0;
    S.reuse = 1;
    S.n_unsucc += 1;
    if (trace != nullptr && S.n_trace < trace_cap && lane == 0) {
      clc_iteration it;
      it.iteration = S.iteration; it.step_is_valid = 0; it.step_is_successful = 0; it.pad_ = 0;
      it.cost = S.x_cost; it.cost_change = 0.0; it.gradient_max_norm = S.gmax; it.step_norm = 0.0;
      it.relative_decrease = 0.0; it.trust_region_radius = S.radius;
      trace[S.n_trace] = it;
    }
    S.n_trace += 1;
    S.min_iter_cost = S.x_cost < S.min_iter_cost ? S.x_cost : S.min_iter_cost;
    const int st2 = S.iteration >= o.max_num_iterations ? CLC_NO_CONVERGENCE
                                                         : (S.radius <= o.min_trust_region_radius ? CLC_CONVERGENCE_RADIUS : CLC_RUNNING);
    if (st2 != CLC_RUNNING) {
      S.status = st2;
      lmu_publish_status(pub, st2, lane);
      return;
    }
    lmu_read_totals(tot2 + 32 * S.hx, T);  // g, H at x
  } else {
    S.n_evals += 1;
    const double cost_e = finalize_cost(T[27], o.use_loss != 0, o.loss_scale_factor);
    const bool finite_eval = fabs(cost_e) <= DMAX;
    const int it_iteration = first ? 0 : S.iteration;
    // ---- the pass itself: IterationZero needs a finite cost; a candidate may end the solve through ParameterToleranceReached
    // (lmu_pre) or FunctionToleranceReached — then nothing else changes (Ceres maps a failed candidate evaluation to cost = DBL_MAX) ----
    const double candidate_cost = finite_eval ? cost_e : DMAX;
    const double it_step_norm = first ? 0.0 : S.step_norm;
    const double it_cost_change = first ? 0.0 : S.x_cost - candidate_cost;
    const bool fun_tol = fabs(it_cost_change) <= o.function_tolerance * S.x_cost;
    const int early_c = S.par_tol ? CLC_CONVERGENCE_PARAMETER : (fun_tol ? CLC_CONVERGENCE_FUNCTION : CLC_RUNNING);
    const int early = first ? (finite_eval ? CLC_RUNNING : CLC_FAILURE) : early_c;
    // ---- IsStepSuccessful; HandleSuccessfulStep / HandleUnsuccessfulStep as selects (IterationZero: a "successful step" onto the
    // start point that leaves radius and decrease factor alone) ----
    const double it_rel = first ? 0.0 : it_cost_change * S.inv_mcc;
    const bool success = first || it_rel > o.min_relative_decrease;
    const double q = 2.0 * it_rel - 1.0;  // StepAccepted
    double den = 1.0 - q * q * q;
    den = den > (1.0 / 3.0) ? den : (1.0 / 3.0);
    double r_acc = S.radius * rcp_pos(den);
    r_acc = r_acc < o.max_trust_region_radius ? r_acc : o.max_trust_region_radius;
    const double r_new = success ? r_acc : S.r_rej;
    const double radius_n = first ? S.radius : r_new;
    // ---- the tests of FinalizeIterationAndCheckIfMinimizerCanContinue that need no gradient (the gradient tolerance is lmu_pre's) ----
    const int stat_c = it_iteration >= o.max_num_iterations ? CLC_NO_CONVERGENCE
                                                             : (radius_n <= o.min_trust_region_radius ? CLC_CONVERGENCE_RADIUS : CLC_RUNNING);
    if (early != CLC_RUNNING) {  // (wave-uniform) the solve ends on the pass itself: nothing else changes
      S.status = early;
      lmu_publish_status(pub, early, lane);
      return;
    }
    S.radius = radius_n;
    {
      const double d_new = success ? 2.0 : S.dfac * 2.0;
      S.dfac = first ? S.dfac : d_new;
    }
    S.reuse = success ? 0 : 1;
    S.x_norm = (success && !first) ? S.xe_norm : S.x_norm;  // (IterationZero keeps lm_init's norm of the start point)
    S.x_cost = success ? candidate_cost : S.x_cost;
#pragma unroll
    for (int i = 0; i < NA; ++i) S.x[i] = success ? S.xe[i] : S.x[i];
    S.initial_cost = first ? S.x_cost : S.initial_cost;
    S.min_iter_cost = first ? S.x_cost : S.min_iter_cost;
    const int it_succ = success ? 1 : 0;
    S.n_succ += it_succ;
    S.n_unsucc += 1 - it_succ;
    const bool xout_dirty = success && S.x_cost < S.minimum_cost;
    S.minimum_cost = xout_dirty ? S.x_cost : S.minimum_cost;
    S.xout_pending = S.xout_pending || xout_dirty;  // (written by the next lmu_pre / lmu_finish: x does not move before)
    S.min_iter_cost = candidate_cost < S.min_iter_cost ? candidate_cost : S.min_iter_cost;
    S.last_success = success;
    S.rec_pending = true;
    S.rec_slot = S.n_trace;
    S.rec_iteration = it_iteration;
    S.rec_succ = it_succ;
    S.rec_cost = candidate_cost;
    S.rec_cost_change = it_cost_change;
    S.rec_step_norm = it_step_norm;
    S.rec_rel = it_rel;
    S.rec_radius = radius_n;
    S.n_trace += 1;
    S.iteration = it_iteration;
    // g, H at x: the totals just read after a successful step (they are the Gauss-Newton system at the candidate — the buffers swap
    // roles); after a rejected step (the minority) the other buffer, read into the same registers
    if (success) {  // (wave-uniform)
      S.hx = 1 - S.hx;
    } else {
      lmu_read_totals(tot2 + 32 * S.hx, T);
    }
    if (first && o.jacobi_scaling) {  // (wave-uniform) once per solve
#pragma unroll
      for (int c = 0; c < NP; ++c) S.scale[c] = 1.0 / (1.0 + sqrt(T[tri<NP>(c, c)]));
    }
    if (stat_c != CLC_RUNNING) {  // (wave-uniform) the iteration cap or the radius tolerance: the gradient tolerance has precedence over the latter
      const bool g_stop = lmu_close_iteration(S, o, T, trace, trace_cap, lane);
      S.status = (stat_c == CLC_CONVERGENCE_RADIUS && g_stop) ? CLC_CONVERGENCE_GRADIENT : stat_c;
      lmu_publish_status(pub, S.status, lane);
      return;
    }
  }
  LMU_CK(1);
  // ---- lm_compute_step (clc_lm.hpp) for iteration S.iteration + 1 ----
  double Hs[21], gs[NP];
  lmu_scale_system(T, S.scale, Hs, gs);
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    double d = Hs[tri<NP>(c, c)];
    d = d > o.min_lm_diagonal ? d : o.min_lm_diagonal;
    d = d < o.max_lm_diagonal ? d : o.max_lm_diagonal;
    S.diag[c] = S.reuse ? S.diag[c] : d;
  }
  S.reuse = 1;
  LMU_CK(2);
  const double inv_radius = rcp_pos(S.radius);
  // chol_solve<6> (clc_math.hpp), unrolled on the packed upper triangle: A(i, j) = A(j, i) = Hs[tri(j, i)], j <= i; L holds the
  // factor's off-diagonal entries, inv[j] = 1 / L[j][j].  (A pivot <= 0 or NaN makes its reciprocal square root NaN, and with it y:
  // lmu_pre's finiteness test of y is the serial code's two tests in one.)
  double L[NP][NP], inv[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    double d = Hs[tri<NP>(j, j)];
    d += S.diag[j] * inv_radius;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    inv[j] = rsqrt_pos(d);
#pragma unroll
    for (int i = j + 1; i < NP; ++i) {
      double sv = Hs[tri<NP>(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) sv -= L[i][k] * L[j][k];
      L[i][j] = sv * inv[j];
    }
  }
  LMU_CK(3);
  double z[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double sv = gs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) sv -= L[i][k] * z[k];
    z[i] = sv * inv[i];
  }
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double sv = z[i];
#pragma unroll
    for (int k = i + 1; k < NP; ++k) sv -= L[k][i] * S.y[k];
    S.y[i] = sv * inv[i];
  }
  LMU_CK(4);
  // ---- Plus: the candidate (whether the step is valid is lmu_pre's to find out) ----
  {
    double dlt[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) dlt[c] = -S.y[c] * S.scale[c];  // undo column scaling
    pose_plus_rcp(S.x, dlt, S.xe);
  }
  S.iteration += 1;
  S.status = CLC_RUNNING;
  LMU_CK(5);
  // ---- what the other waves wait for: the next pose to evaluate ----
  lmu_publish(pub, S.xe, CLC_RUNNING, lane);
  LMU_CK(6);
#ifdef CLC_STAMPS
  if (blockIdx.x == 8 && lane == 0 && !first) clc_lmu_ck[0] = clc_lmu_ck[7];
#endif
}

// End of the solve: the last iteration's record, and the outputs into the LDS state (batched_write_outcome / lm_fill_summary read
// them there).  `tot2`: the totals buffers (the record's gradient norm may still be due).
__device__ __forceinline__ void lmu_finish(LmU& S, LmState& st, const clc_options& o, const double* tot2, clc_iteration* __restrict__ trace,
                                           const int trace_cap, const int lane) {
  if (S.rec_pending) {  // (wave-uniform) the solve ended before an lmu_pre closed the last iteration (tolerances on the pass after it)
    double T[28];
    lmu_read_totals(tot2 + 32 * S.hx, T);
    (void)lmu_close_iteration(S, o, T, trace, trace_cap, lane);
  }
  lmu_flush_xout(S, st, lane);
  if (lane == 0) {
    st.status = S.status;
    st.n_trace = S.n_trace;
    st.num_successful = S.n_succ;
    st.num_unsuccessful = S.n_unsucc;
    st.n_evals = S.n_evals;
    st.initial_cost = S.initial_cost;
    st.min_iter_cost = S.min_iter_cost;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace clc
