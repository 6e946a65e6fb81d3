// tests/golden/dump_ceres_trace.cpp — hand-off tool, NOT built or run here (Ceres is absent from this image).
//
// Runs REAL ceres::Solve on a records file written by `python tests/golden/make_ceres_trace.py --write-inputs DIR`
// with exactly the problem the reference builds (src/LaseCamCalCeres.cpp:217-307: one residual block per record with
// its own CauchyLoss(0.05 * scale), one 7-parameter block with the SE(3) local parameterisation, DENSE_QR, 100
// iterations) and prints summary.iterations as JSON, to be diffed against tests/golden/ceres_trace_expected.json with
// tests/golden/compare_ceres_trace.py.  The residual / Jacobian / Plus below are written from the formulas the
// reference implements (PointInPlaneFactor::Evaluate :43-66; PoseLocalParameterization::Plus,
// src/pose_local_parameterization.cpp:15-31), not copied from it.
//
//   g++ -O2 -std=c++14 dump_ceres_trace.cpp -o dump_ceres_trace $(pkg-config --cflags --libs ceres eigen3)   # Ceres <= 2.1
//   ./dump_ceres_trace c1_seed1_sigma0.01.records.bin > c1_seed1_sigma0.01.ceres.json
#include <ceres/ceres.h>

#include <cstdio>
#include <vector>

struct PointInPlane : ceres::SizedCostFunction<1, 7> {  // record {n[3], d, p[3], scale}
  double n[3], d, p[3], s;
  bool Evaluate(double const* const* x, double* r, double** J) const override {
    const double* t = x[0];
    const double qx = t[3], qy = t[4], qz = t[5], qw = t[6];
    const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                         2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                         2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    double pc[3], m[3];
    for (int i = 0; i < 3; ++i) pc[i] = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2] + t[i];
    for (int i = 0; i < 3; ++i) m[i] = R[i] * n[0] + R[3 + i] * n[1] + R[6 + i] * n[2];  // R^T n
    r[0] = s * (n[0] * pc[0] + n[1] * pc[1] + n[2] * pc[2] + d);
    if (J && J[0]) {
      double* j = J[0];
      j[0] = s * n[0]; j[1] = s * n[1]; j[2] = s * n[2];
      j[3] = s * (p[1] * m[2] - p[2] * m[1]);  // n^T (-R [p]x) = (p x R^T n)^T
      j[4] = s * (p[2] * m[0] - p[0] * m[2]);
      j[5] = s * (p[0] * m[1] - p[1] * m[0]);
      j[6] = 0.0;
    }
    return true;
  }
};

struct PosePlus : ceres::LocalParameterization {  // p + dp ; q (x) [1, dtheta/2], normalised
  bool Plus(const double* x, const double* dl, double* o) const override {
    for (int i = 0; i < 3; ++i) o[i] = x[i] + dl[i];
    const double bx = dl[3] / 2, by = dl[4] / 2, bz = dl[5] / 2, ax = x[3], ay = x[4], az = x[5], aw = x[6];
    double q[4] = {aw * bx + ax + ay * bz - az * by, aw * by + ay + az * bx - ax * bz, aw * bz + az + ax * by - ay * bx,
                   aw - ax * bx - ay * by - az * bz};
    const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) o[3 + i] = q[i] / nrm;
    return true;
  }
  bool ComputeJacobian(const double*, double* J) const override {
    for (int i = 0; i < 42; ++i) J[i] = 0.0;
    for (int i = 0; i < 6; ++i) J[6 * i + i] = 1.0;
    return true;
  }
  int GlobalSize() const override { return 7; }
  int LocalSize() const override { return 6; }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  double nd; std::fread(&nd, 8, 1, f);
  const long N = (long)nd;
  double pose[7]; std::fread(pose, 8, 7, f);
  std::vector<double> rec(8 * N); std::fread(rec.data(), 8, 8 * N, f); std::fclose(f);
  ceres::Problem problem;
  problem.AddParameterBlock(pose, 7, new PosePlus);
  for (long k = 0; k < N; ++k) {
    PointInPlane* c = new PointInPlane;
    const double* o = &rec[8 * k];
    c->n[0] = o[0]; c->n[1] = o[1]; c->n[2] = o[2]; c->d = o[3]; c->p[0] = o[4]; c->p[1] = o[5]; c->p[2] = o[6]; c->s = o[7];
    problem.AddResidualBlock(c, new ceres::CauchyLoss(0.05 * o[7]), pose);
  }
  ceres::Solver::Options opt;  // src/LaseCamCalCeres.cpp:302-304; everything else Ceres defaults
  opt.linear_solver_type = ceres::DENSE_QR;
  opt.max_num_iterations = 100;
  ceres::Solver::Summary sum;
  ceres::Solve(opt, &problem, &sum);
  std::printf("{\"ceres_version\": \"%s\", \"termination\": %d, \"num_iterations\": %d, \"initial_cost\": %.17g, \"final_cost\": %.17g,\n \"pose\": [",
              CERES_VERSION_STRING, (int)sum.termination_type, (int)sum.iterations.size() - 1, sum.initial_cost, sum.final_cost);
  for (int i = 0; i < 7; ++i) std::printf("%.17g%s", pose[i], i < 6 ? ", " : "],\n \"iterations\": [\n");
  for (size_t i = 0; i < sum.iterations.size(); ++i) {
    const ceres::IterationSummary& it = sum.iterations[i];
    std::printf("  {\"iteration\": %d, \"cost\": %.17g, \"cost_change\": %.17g, \"gradient_max_norm\": %.17g, \"step_norm\": %.17g, "
                "\"relative_decrease\": %.17g, \"trust_region_radius\": %.17g, \"step_is_valid\": %d, \"step_is_successful\": %d}%s\n",
                it.iteration, it.cost, it.cost_change, it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius,
                (int)it.step_is_valid, (int)it.step_is_successful, i + 1 < sum.iterations.size() ? "," : "");
  }
  std::printf(" ]}\n");
  return 0;
}
