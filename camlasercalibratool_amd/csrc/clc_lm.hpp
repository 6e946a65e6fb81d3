// clc_lm.hpp — the Levenberg-Marquardt trust-region controller that replaces ceres::Solve
// for the problems this path builds: the 6-DoF pose problem of CamLaserCalibration
// (src/LaseCamCalCeres.cpp:299-307) and the 2-parameter line fit of LineFittingCeres (:423-428).
//
// It is a re-entrant state machine, not a loop: the evaluation kernels stream the
// observations and reduce {cost, g = J~^T r~, H = J~^T J~} at the point `x_eval`;
// `lm_advance` consumes that result, runs everything Ceres does between two evaluations (step
// acceptance, radius update, termination tests, Jacobi scaling, LM damping, the damped
// normal-equation solve, Plus) and leaves the next point to evaluate in `x_eval`.
// One thread executes it: in lm_kernel / the tail of eval_lm_kernel (single problem), once per
// problem in batched_lm_kernel, redundantly per wave in line_fit_kernel.  All state lives in
// device memory so a whole solve is enqueued without host round trips.
//
// The parameter space is a policy:  M::NP local / M::NA ambient parameters and M::plus(), the
// LocalParameterization's Plus (Se3Manifold = PoseLocalParameterization,
// src/pose_local_parameterization.cpp:15-31; Euclid2Manifold = plain addition).
//
// Semantics follow Ceres 1.13-2.1 TrustRegionMinimizer / LevenbergMarquardtStrategy with the
// options of SURVEY.md Appendix A.  Differences by design (results equal to rounding, verified
// against the oracle): the linear step solves the damped normal equations by Cholesky instead of
// Householder QR of [J;D]; the candidate is evaluated with its Jacobian in the same pass, so an
// accepted step needs no second pass.
// Written for device code; also compiled for the host by the unit shim in tests/.
#pragma once
#include "clc_math.hpp"
#include "../../include/clc.h"

namespace clc {

struct Se3Manifold {  // pose = [t(3), q(x,y,z,w)], tangent = [dt, dtheta]
  static constexpr int NP = 6, NA = 7;
  CLC_HD static void plus(const double* x, const double* d, double* out) { pose_plus_rcp(x, d, out); }
};

struct Euclid2Manifold {  // line m0 x + m1 y + 1 = 0
  static constexpr int NP = 2, NA = 2;
  CLC_HD static void plus(const double* x, const double* d, double* out) {
    out[0] = x[0] + d[0];
    out[1] = x[1] + d[1];
  }
};

template <class M>
struct LmStateT {
  static constexpr int NP = M::NP, NA = M::NA, NH = M::NP * (M::NP + 1) / 2;
  int32_t status;  // CLC_RUNNING or a termination code
  int32_t phase;   // 0: x_eval is the initial point, 1: x_eval is a candidate
  int32_t iteration;
  int32_t n_invalid;
  int32_t reuse_diagonal;
  int32_t num_successful;
  int32_t num_unsuccessful;
  int32_t n_trace;
  int64_t n_evals;
  double x[NA];       // current accepted iterate
  double x_eval[NA];  // point the next evaluation pass must use
  double x_out[NA];   // Ceres' `parameters_`: lowest-cost accepted iterate
  double x_norm, x_cost, minimum_cost, initial_cost, min_iter_cost;
  double g[NP], H[NH];  // gradient / Gauss-Newton matrix at x (unscaled, robustified)
  double scale[NP];     // Jacobi column scaling, fixed at iteration 0
  double diag[NP];      // clamp(diag(J_s^T J_s)) used for the LM damping
  double radius, decrease_factor;
  double step[NP];      // trust-region step in the scaled space
  double model_cost_change;
  double gmax;
};

// Temporaries of the controller.
template <class M>
struct LmScratchT {
  double Hs[M::NP * M::NP], A[M::NP * M::NP], L[M::NP * M::NP];
  double gs[M::NP], y[M::NP], z[M::NP], ng[M::NP], delta[M::NP];
  double proj[M::NA];
};

using LmState = LmStateT<Se3Manifold>;
using LmScratch = LmScratchT<Se3Manifold>;

template <class M>
CLC_HD void lm_init(LmStateT<M>& s, const clc_options& o, const double* x0) {
  constexpr int NP = M::NP, NA = M::NA, NH = LmStateT<M>::NH;
  s.status = CLC_RUNNING;
  s.phase = 0;
  s.iteration = 0;
  s.n_invalid = 0;
  s.reuse_diagonal = 0;
  s.num_successful = 0;
  s.num_unsuccessful = 0;
  s.n_trace = 0;
  s.n_evals = 0;
  CLC_ROLLED for (int i = 0; i < NA; ++i) s.x[i] = s.x_eval[i] = s.x_out[i] = x0[i];
  s.x_norm = norm_n<NA>(x0);
  s.x_cost = 0.0;
  s.minimum_cost = 1.7976931348623157e308;
  s.initial_cost = 0.0;
  s.min_iter_cost = 0.0;
  CLC_ROLLED for (int i = 0; i < NP; ++i) { s.g[i] = 0.0; s.scale[i] = 1.0; s.diag[i] = 0.0; s.step[i] = 0.0; }
  CLC_ROLLED for (int i = 0; i < NH; ++i) s.H[i] = 0.0;
  s.radius = o.initial_trust_region_radius;
  s.decrease_factor = 2.0;
  s.model_cost_change = 0.0;
  s.gmax = 0.0;
}

// ||x - Plus(x, -g)||_inf in the ambient space (Ceres' projected-gradient norm).
template <class M>
CLC_HD double gradient_max_norm(const double* x, const double* g, LmScratchT<M>& w) {
  CLC_ROLLED for (int i = 0; i < M::NP; ++i) w.ng[i] = -g[i];
  M::plus(x, w.ng, w.proj);
  double m = 0.0;
  CLC_ROLLED for (int i = 0; i < M::NA; ++i) {
    m = fmax(m, fabs(x[i] - w.proj[i]));
  }
  return m;
}

// LevenbergMarquardtStrategy::ComputeStep + the model-cost test of ComputeTrustRegionStep.
template <class M>
CLC_HD bool lm_compute_step(LmStateT<M>& s, const clc_options& o, LmScratchT<M>& w) {
  constexpr int NP = M::NP;
  int idx = 0;
  CLC_ROLLED for (int a = 0; a < NP; ++a)
    CLC_ROLLED for (int b = a; b < NP; ++b) {
      const double v = s.H[idx++] * (s.scale[a] * s.scale[b]);
      w.Hs[NP * a + b] = v;
      w.Hs[NP * b + a] = v;
    }
  CLC_ROLLED for (int a = 0; a < NP; ++a) w.gs[a] = s.g[a] * s.scale[a];
  if (!s.reuse_diagonal) {
    CLC_ROLLED for (int c = 0; c < NP; ++c) {
      double d = w.Hs[NP * c + c];
      d = d > o.min_lm_diagonal ? d : o.min_lm_diagonal;
      d = d < o.max_lm_diagonal ? d : o.max_lm_diagonal;
      s.diag[c] = d;
    }
  }
  CLC_ROLLED for (int i = 0; i < NP * NP; ++i) w.A[i] = w.Hs[i];
  // Ceres appends lm_diagonal = sqrt(diag / radius) as rows of [J; D]; in the normal
  // equations that is + D^2 = diag / radius on the diagonal.
  const double inv_radius = rcp_pos(s.radius);
  CLC_ROLLED for (int c = 0; c < NP; ++c) w.A[NP * c + c] += s.diag[c] * inv_radius;
  bool ok = chol_solve<NP>(w.A, w.gs, w.y, w.L, w.z);
  s.reuse_diagonal = 1;
  if (ok) {
    CLC_ROLLED for (int c = 0; c < NP; ++c)
      if (!(fabs(w.y[c]) <= 1.7976931348623157e308)) ok = false;  // NaN/Inf check
  }
  if (!ok) return false;
  double sg = 0.0, shs = 0.0;
  CLC_ROLLED for (int a = 0; a < NP; ++a) s.step[a] = -w.y[a];
  CLC_ROLLED for (int a = 0; a < NP; ++a) {
    sg += s.step[a] * w.gs[a];
    double row = 0.0;
    CLC_ROLLED for (int b = 0; b < NP; ++b) row += w.Hs[NP * a + b] * s.step[b];
    shs += s.step[a] * row;
  }
  // model_cost_change = -(J step)^T (r + J step / 2)
  s.model_cost_change = -(sg + 0.5 * shs);
  return s.model_cost_change > 0.0;
}

template <class M>
CLC_HD void lm_record(LmStateT<M>& s, clc_iteration* trace, int trace_cap, const clc_iteration& it) {
  if (trace && s.n_trace < trace_cap) trace[s.n_trace] = it;
  s.n_trace++;
  s.min_iter_cost = it.cost < s.min_iter_cost ? it.cost : s.min_iter_cost;
}

// The loop of TrustRegionMinimizer::Minimize between two evaluations, entered either at the top (`it` = the iteration
// just evaluated: finalize, record, test, compute the next step) or — `at_invalid_step` — right after a step computation
// that failed for iteration it.iteration (`it` otherwise zeroed; used by the wavefront controller of the step kernel,
// which runs the first round itself and hands the rare invalid step over to this serial code).
template <class M>
CLC_HD void lm_iterate(LmStateT<M>& s, LmScratchT<M>& w, const clc_options& o, clc_iteration* trace, int trace_cap,
                       clc_iteration& it, bool at_invalid_step) {
  constexpr int NP = M::NP, NA = M::NA;
  for (;;) {
    if (!at_invalid_step) {
      // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
      if (it.step_is_successful) {
        s.num_successful++;
        if (s.x_cost < s.minimum_cost) {
          s.minimum_cost = s.x_cost;
          CLC_ROLLED for (int i = 0; i < NA; ++i) s.x_out[i] = s.x[i];
        }
      } else {
        s.num_unsuccessful++;
      }
      it.trust_region_radius = s.radius;
      lm_record(s, trace, trace_cap, it);
      if (it.iteration >= o.max_num_iterations) { s.status = CLC_NO_CONVERGENCE; return; }
      if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) {
        s.status = CLC_CONVERGENCE_GRADIENT;
        return;
      }
      if (it.trust_region_radius <= o.min_trust_region_radius) {
        s.status = CLC_CONVERGENCE_RADIUS;
        return;
      }
      // ---- next iteration: ComputeTrustRegionStep ----
      const int next = it.iteration + 1;
      it.iteration = next; it.step_is_valid = 0; it.step_is_successful = 0;
      it.cost = 0.0; it.cost_change = 0.0; it.gradient_max_norm = 0.0; it.step_norm = 0.0;
      it.relative_decrease = 0.0; it.trust_region_radius = 0.0;
      if (lm_compute_step(s, o, w)) {
        s.n_invalid = 0;
        CLC_ROLLED for (int c = 0; c < NP; ++c) w.delta[c] = s.step[c] * s.scale[c];  // undo column scaling
        M::plus(s.x, w.delta, s.x_eval);  // candidate
        s.phase = 1;
        s.iteration = next;
        return;  // request an evaluation at x_eval
      }
    }
    at_invalid_step = false;
    // ---- HandleInvalidStep ----
    if (++s.n_invalid >= o.max_num_consecutive_invalid_steps) { s.status = CLC_FAILURE; return; }
    s.radius = s.radius / s.decrease_factor;  // StepIsInvalid == StepRejected(0)
    s.decrease_factor *= 2.0;
    s.reuse_diagonal = 1;
    it.cost = s.x_cost;
    it.gradient_max_norm = s.gmax;
  }
}

// Consume one evaluation {cost, g[NP], H[NH]} taken at s.x_eval and advance to the next
// evaluation request or to termination.
template <class M>
CLC_HD void lm_advance(LmStateT<M>& s, LmScratchT<M>& w, const clc_options& o, clc_iteration* trace,
                       int trace_cap, double cost_e, const double* g_e, const double* H_e) {
  constexpr int NP = M::NP, NA = M::NA, NH = LmStateT<M>::NH;
  if (s.status != CLC_RUNNING) return;
  s.n_evals++;
  clc_iteration it;
  it.iteration = 0; it.step_is_valid = 0; it.step_is_successful = 0; it.pad_ = 0;
  it.cost = 0.0; it.cost_change = 0.0; it.gradient_max_norm = 0.0; it.step_norm = 0.0;
  it.relative_decrease = 0.0; it.trust_region_radius = 0.0;

  const bool finite_eval = fabs(cost_e) <= 1.7976931348623157e308;  // false for NaN/Inf
  if (s.phase == 0) {
    // ---- IterationZero ----
    if (!finite_eval) { s.status = CLC_FAILURE; return; }
    s.x_cost = cost_e;
    CLC_ROLLED for (int i = 0; i < NP; ++i) s.g[i] = g_e[i];
    CLC_ROLLED for (int i = 0; i < NH; ++i) s.H[i] = H_e[i];
    if (o.jacobi_scaling)
      CLC_ROLLED for (int c = 0; c < NP; ++c) s.scale[c] = 1.0 / (1.0 + sqrt(s.H[tri<NP>(c, c)]));
    s.gmax = gradient_max_norm<M>(s.x, s.g, w);
    s.initial_cost = s.x_cost;
    s.min_iter_cost = s.x_cost;
    it.iteration = 0;
    it.cost = s.x_cost;
    it.gradient_max_norm = s.gmax;
    it.step_is_valid = 1;
    it.step_is_successful = 1;
  } else {
    it.iteration = s.iteration;
    it.step_is_valid = 1;
    // Ceres maps a failed candidate evaluation to cost = DBL_MAX (step gets rejected).
    const double candidate_cost = finite_eval ? cost_e : 1.7976931348623157e308;
    // ---- ParameterToleranceReached ----
    double sn = 0.0;
    CLC_ROLLED for (int i = 0; i < NA; ++i) sn += (s.x[i] - s.x_eval[i]) * (s.x[i] - s.x_eval[i]);
    it.step_norm = sqrt_pos(sn);
    if (it.step_norm <= o.parameter_tolerance * (s.x_norm + o.parameter_tolerance)) {
      s.status = CLC_CONVERGENCE_PARAMETER;
      return;
    }
    // ---- FunctionToleranceReached ----
    it.cost_change = s.x_cost - candidate_cost;
    if (fabs(it.cost_change) <= o.function_tolerance * s.x_cost) {
      s.status = CLC_CONVERGENCE_FUNCTION;
      return;
    }
    // ---- IsStepSuccessful (monotonic step evaluator) ----
    it.relative_decrease = it.cost_change * rcp_pos_safe(s.model_cost_change);  // model_cost_change > 0 (lm_compute_step), possibly tiny
    if (it.relative_decrease > o.min_relative_decrease) {
      // ---- HandleSuccessfulStep: the fused pass already produced g,H at the candidate ----
      CLC_ROLLED for (int i = 0; i < NA; ++i) s.x[i] = s.x_eval[i];
      s.x_norm = norm_n<NA>(s.x);
      s.x_cost = candidate_cost;
      CLC_ROLLED for (int i = 0; i < NP; ++i) s.g[i] = g_e[i];
      CLC_ROLLED for (int i = 0; i < NH; ++i) s.H[i] = H_e[i];
      s.gmax = gradient_max_norm<M>(s.x, s.g, w);
      it.cost = s.x_cost;
      it.gradient_max_norm = s.gmax;
      it.step_is_successful = 1;
      const double q = 2.0 * it.relative_decrease - 1.0;  // StepAccepted
      double den = 1.0 - q * q * q;
      den = den > (1.0 / 3.0) ? den : (1.0 / 3.0);
      s.radius = s.radius * rcp_pos(den);  // den >= 1/3
      s.radius = s.radius < o.max_trust_region_radius ? s.radius : o.max_trust_region_radius;
      s.decrease_factor = 2.0;
      s.reuse_diagonal = 0;
    } else {
      // ---- HandleUnsuccessfulStep ----
      it.step_is_successful = 0;
      s.radius = s.radius / s.decrease_factor;  // StepRejected
      s.decrease_factor *= 2.0;
      s.reuse_diagonal = 1;
      it.cost = candidate_cost;
      it.gradient_max_norm = s.gmax;
    }
  }

  lm_iterate(s, w, o, trace, trace_cap, it, false);
}

template <class M>
CLC_HD void lm_fill_summary(const LmStateT<M>& s, clc_summary& out) {
  out.termination = s.status;
  out.num_iterations = s.n_trace - 1;
  out.num_successful_steps = s.num_successful;
  out.num_unsuccessful_steps = s.num_unsuccessful;
  out.num_evaluations = s.n_evals;
  out.initial_cost = s.initial_cost;
  out.final_cost = s.initial_cost < s.min_iter_cost ? s.initial_cost : s.min_iter_cost;
}

}  // namespace clc
