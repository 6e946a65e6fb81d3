// mini_ceres.hpp — stand-in for the part of the Ceres Solver modelling API that the reference uses.
//
// TEST INFRASTRUCTURE ONLY.  Ceres is not installed in this image (find_package(Ceres REQUIRED),
// CMakeLists.txt:37, version unpinned) and there is no network.  This header lets the reference's
// own translation units compile and RUN: its cost function (PointInPlaneFactor::Evaluate), its
// local parameterisation (PoseLocalParameterization), its loss objects and its problem-assembly
// loops are the reference's code, executed as written.  What is NOT the reference is the minimiser:
// ceres::Solve() below hands the assembled problem to the oracle's restatement of Ceres'
// trust-region Levenberg-Marquardt loop (oracle/clc_oracle.cpp, lm_minimize) — so a run through this
// header pins the oracle's restatement of the reference-OWNED arithmetic (factor, Jacobian, Plus,
// scales, loss scale, boundary planes, closed form), not Ceres' iteration semantics.
//
// Semantics restated from the public Ceres API documentation (cost_function.h, loss_function.h,
// local_parameterization.h, problem.h, solver.h of Ceres 1.13-2.1):
//   * residual blocks are evaluated in insertion order; a block's loss rho(s), s = ||r||^2, enters as
//     cost += rho[0]/2 and through the corrector (Triggs): with rho'' <= 0 (or s = 0) the residual
//     and Jacobian are scaled by sqrt(rho'); otherwise by the alpha-form;
//   * the Jacobian handed to the linear solver is J_global(num_res x global) * ComputeJacobian
//     (global x local);
//   * CauchyLoss(a): rho(s) = b log(1 + s/b), b = a^2.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

// the oracle's minimiser behind C callbacks (oracle/clc_oracle.cpp)
#include "../oracle_types.h"
extern "C" {
typedef double (*oracle_eval_cb)(void* ctx, const double* x, double* res, double* J, long long ldj, double* g);
typedef void (*oracle_plus_cb)(void* ctx, const double* x, const double* delta, double* out);
void oracle_options_default(oracle_options* o);
int oracle_minimize_cb(int np, int na, long long n_res, oracle_eval_cb eval, oracle_plus_cb plus, void* ctx,
                       const oracle_options* opt, double* x, oracle_summary* summary, oracle_iteration* trace,
                       int trace_cap);
}

namespace ceres {

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32_t> sizes_;
  int num_residuals_;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
  }
};

// forward-mode dual number with N partials (enough of ceres::Jet for the reference's line functor)
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  explicit Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet operator+(const Jet& o) const { Jet r; r.a = a + o.a; for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i]; return r; }
  Jet operator-(const Jet& o) const { Jet r; r.a = a - o.a; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
  Jet operator*(const Jet& o) const { Jet r; r.a = a * o.a; for (int i = 0; i < N; ++i) r.v[i] = a * o.v[i] + v[i] * o.a; return r; }
};

template <typename Functor, int kNumResiduals, int N0>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, N0> {
 public:
  explicit AutoDiffCostFunction(Functor* f) : f_(f) {}
  virtual ~AutoDiffCostFunction() { delete f_; }
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    if (!jacobians || !jacobians[0]) return (*f_)(parameters[0], residuals);
    Jet<N0> x[N0], r[kNumResiduals];
    for (int i = 0; i < N0; ++i) { x[i].a = parameters[0][i]; x[i].v[i] = 1.0; }
    if (!(*f_)(x, r)) return false;
    for (int k = 0; k < kNumResiduals; ++k) {
      residuals[k] = r[k].a;
      for (int i = 0; i < N0; ++i) jacobians[0][k * N0 + i] = r[k].v[i];
    }
    return true;
  }

 private:
  Functor* f_;
};

class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};

class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / b_) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * c_;
    const double inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }

 private:
  const double b_, c_;
};

class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};

class Problem {
 public:
  struct ResidualBlock {
    CostFunction* cost;
    LossFunction* loss;
    double* x;
  };
  Problem() : x_(nullptr), size_(0), param_(nullptr) {}
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0) {
    blocks_.push_back(ResidualBlock{cost, loss, x0});
    if (!x_) { x_ = x0; size_ = cost->parameter_block_sizes().at(0); }
  }
  void AddParameterBlock(double* values, int size, LocalParameterization* p = nullptr) {
    x_ = values; size_ = size; param_ = p;
  }
  void SetParameterization(double* values, LocalParameterization* p) { x_ = values; param_ = p; }
  int NumResidualBlocks() const { return (int)blocks_.size(); }

  // one parameter block (all the reference ever builds)
  std::vector<ResidualBlock> blocks_;
  double* x_;
  int size_;
  LocalParameterization* param_;
};

// What the last ceres::Solve did, for the test harness (oracle/ref_shim/ref_capi.cpp).
struct SolveRecord {
  int termination = 0, num_iterations = 0, num_successful_steps = 0, num_unsuccessful_steps = 0;
  double initial_cost = 0.0, final_cost = 0.0;
  long long num_residual_blocks = 0;
};
inline SolveRecord& last_solve_record() {
  static SolveRecord r;
  return r;
}

#define CERES_VERSION_STRING "stand-in (oracle/ref_shim/mini_ceres.hpp -> the oracle's LM restatement)"
struct IterationSummary {  // ceres::IterationSummary subset
  int iteration = 0;
  bool step_is_valid = false, step_is_successful = false;
  double cost = 0.0, cost_change = 0.0, gradient_max_norm = 0.0, step_norm = 0.0, relative_decrease = 0.0,
         trust_region_radius = 0.0;
};

class Solver {
 public:
  struct Options {
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    int max_num_iterations = 50;
    bool minimizer_progress_to_stdout = false;
    int num_threads = 1;
  };
  struct Summary {
    SolveRecord rec;
    // the part of ceres::Solver::Summary tests/golden/dump_ceres_trace.cpp reads
    std::vector<IterationSummary> iterations;
    int termination_type = 0;
    double initial_cost = 0.0, final_cost = 0.0;
    std::string BriefReport() const {
      std::ostringstream os;
      os << "stand-in Ceres (oracle LM): iterations " << rec.num_iterations << ", initial cost " << rec.initial_cost
         << ", final cost " << rec.final_cost << ", termination " << rec.termination;
      return os.str();
    }
    std::string FullReport() const { return BriefReport(); }
  };
};

namespace shim {
inline double evaluate(void* ctx, const double* x, double* res, double* J, long long ldj, double* g) {
  const Problem& pb = *static_cast<const Problem*>(ctx);
  const int na = pb.size_;
  const int np = pb.param_ ? pb.param_->LocalSize() : na;
  std::vector<double> P;  // global x local, row-major
  if (pb.param_) {
    P.assign((size_t)na * np, 0.0);
    pb.param_->ComputeJacobian(x, P.data());
  }
  std::vector<double> jg((size_t)na), jl((size_t)np), gg((size_t)np, 0.0);
  const bool need_j = J || g;
  double cost = 0.0;
  long long k = 0;
  for (const Problem::ResidualBlock& b : pb.blocks_) {
    double r = 0.0;
    double* jac[1] = {jg.data()};
    const double* params[1] = {x};
    b.cost->Evaluate(params, &r, need_j ? jac : nullptr);  // one residual per block in this problem family
    if (need_j) {
      if (pb.param_) {
        for (int c = 0; c < np; ++c) {
          double s = 0.0;
          for (int a = 0; a < na; ++a) s += jg[(size_t)a] * P[(size_t)a * np + c];
          jl[(size_t)c] = s;
        }
      } else {
        for (int c = 0; c < np; ++c) jl[(size_t)c] = jg[(size_t)c];
      }
    }
    const double sq = r * r;
    if (!b.loss) {
      cost += 0.5 * sq;
    } else {
      double rho[3];
      b.loss->Evaluate(sq, rho);
      cost += 0.5 * rho[0];
      const double sqrt_rho1 = std::sqrt(rho[1]);
      if (sq == 0.0 || rho[2] <= 0.0) {
        if (need_j) for (int c = 0; c < np; ++c) jl[(size_t)c] *= sqrt_rho1;
        r *= sqrt_rho1;
      } else {
        const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(D);
        // J <- sqrt(rho') (J - alpha/s r r^T J), r <- sqrt(rho')/(1-alpha) r   (single residual: r r^T J = s J)
        if (need_j) for (int c = 0; c < np; ++c) jl[(size_t)c] = sqrt_rho1 * (jl[(size_t)c] - alpha * jl[(size_t)c]);
        r *= sqrt_rho1 / (1.0 - alpha);
      }
    }
    if (res) res[k] = r;
    if (J) for (int c = 0; c < np; ++c) J[(size_t)c * ldj + k] = jl[(size_t)c];
    if (need_j) for (int c = 0; c < np; ++c) gg[(size_t)c] += jl[(size_t)c] * r;
    ++k;
  }
  if (g) for (int c = 0; c < np; ++c) g[c] = gg[(size_t)c];
  return cost;
}

inline void plus(void* ctx, const double* x, const double* d, double* out) {
  const Problem& pb = *static_cast<const Problem*>(ctx);
  if (pb.param_) pb.param_->Plus(x, d, out);
  else for (int i = 0; i < pb.size_; ++i) out[i] = x[i] + d[i];
}
}  // namespace shim

inline void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  oracle_options o;
  oracle_options_default(&o);
  o.max_num_iterations = options.max_num_iterations;
  o.use_loss = 1;  // the loss objects are applied here, block by block
  oracle_summary s = {};
  const int na = problem->size_;
  const int np = problem->param_ ? problem->param_->LocalSize() : na;
  std::vector<oracle_iteration> trace((size_t)options.max_num_iterations + 8);
  oracle_minimize_cb(np, na, (long long)problem->blocks_.size(), shim::evaluate, shim::plus, problem,
                     &o, problem->x_, &s, trace.data(), (int)trace.size());
  SolveRecord rec;
  rec.termination = s.termination;
  rec.num_iterations = s.num_iterations;
  rec.num_successful_steps = s.num_successful_steps;
  rec.num_unsuccessful_steps = s.num_unsuccessful_steps;
  rec.initial_cost = s.initial_cost;
  rec.final_cost = s.final_cost;
  rec.num_residual_blocks = (long long)problem->blocks_.size();
  last_solve_record() = rec;
  if (summary) {
    summary->rec = rec;
    summary->termination_type = s.termination;
    summary->initial_cost = s.initial_cost;
    summary->final_cost = s.final_cost;
    summary->iterations.clear();
    for (int i = 0; i <= s.num_iterations && i < (int)trace.size(); ++i) {
      IterationSummary it;
      it.iteration = trace[(size_t)i].iteration;
      it.step_is_valid = trace[(size_t)i].step_is_valid != 0;
      it.step_is_successful = trace[(size_t)i].step_is_successful != 0;
      it.cost = trace[(size_t)i].cost;
      it.cost_change = trace[(size_t)i].cost_change;
      it.gradient_max_norm = trace[(size_t)i].gradient_max_norm;
      it.step_norm = trace[(size_t)i].step_norm;
      it.relative_decrease = trace[(size_t)i].relative_decrease;
      it.trust_region_radius = trace[(size_t)i].trust_region_radius;
      summary->iterations.push_back(it);
    }
  }
}

}  // namespace ceres
