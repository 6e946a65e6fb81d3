// oracle_types.h — plain-C structs shared by the oracle (clc_oracle.cpp) and the stand-in Ceres of
// oracle/ref_shim.  TEST INFRASTRUCTURE ONLY (see clc_oracle.cpp).
#pragma once
#include <stdint.h>

// Solver options: defaults = Ceres defaults + src/LaseCamCalCeres.cpp:302-304.
struct oracle_options {
  int32_t max_num_iterations;                 // 100   (LaseCamCalCeres.cpp:304)
  int32_t max_num_consecutive_invalid_steps;  // 5
  int32_t jacobi_scaling;                     // 1
  int32_t use_loss;                           // 1     (#define LOSSFUNCTION, :212)
  double loss_scale_factor;                   // 0.05  (CauchyLoss(0.05*scale), :249)
  double initial_trust_region_radius;         // 1e4
  double max_trust_region_radius;             // 1e16
  double min_trust_region_radius;             // 1e-32
  double min_relative_decrease;               // 1e-3
  double min_lm_diagonal;                     // 1e-6
  double max_lm_diagonal;                     // 1e32
  double function_tolerance;                  // 1e-6
  double gradient_tolerance;                  // 1e-10
  double parameter_tolerance;                 // 1e-8
};

// Termination codes (mirrors ceres::TerminationType + which test fired).
enum {
  ORACLE_CONVERGENCE_GRADIENT = 1,
  ORACLE_CONVERGENCE_PARAMETER = 2,
  ORACLE_CONVERGENCE_FUNCTION = 3,
  ORACLE_CONVERGENCE_RADIUS = 4,
  ORACLE_NO_CONVERGENCE = 5,
  ORACLE_FAILURE = 6,
};

struct oracle_iteration {  // ceres::IterationSummary subset
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t pad_;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
};

struct oracle_summary {
  int32_t termination;
  int32_t num_iterations;  // = iterations.size() - 1 (iteration 0 is the initial evaluation)
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int64_t num_residual_evaluations;  // evaluation passes over the data (cost, +jacobian)
  int64_t num_jacobian_evaluations;
  double initial_cost;
  double final_cost;
};

