// =====================================================================================
// oracle/clc_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT.
//
// PARITY — what is pinned and what is not.
//  * PINNED against the reference's own code: everything the reference itself computes —
//    PointInPlaneFactor::Evaluate, PoseLocalParameterization, pi_from_ppp, the assembly
//    loops of CamLaserCalibration (scales, loss scales, boundary planes), the closed form,
//    LineFittingCeres' problem and TranScanToPoints.  The reference's translation units are
//    compiled from /root/reference against stand-in Eigen/Ceres/sensor_msgs headers
//    (oracle/ref_shim, `make ref` -> oracle/_ref/libref.so) and run; their outputs are
//    committed as tests/golden/ref_vectors.json and compared live in tests/test_ref_pin.py.
//  * UNPINNED: Ceres' minimiser.  The LM / DENSE_QR / loss-corrector arithmetic lives in
//    un-vendored, unpinned Ceres Solver (<= 2.1 by API; CMakeLists.txt:37) + Eigen3
//    (CMakeLists.txt:35); neither exists in this environment and the reference ships no tests
//    or golden vectors.  The trust-region loop below restates the published Ceres 1.13-2.1
//    algorithm; it is anchored only by the simulation ground truth
//    (main/calibr_simulation.cpp:15-20), finite differences, and an independent
//    scipy.optimize minimisation of the same objective (tests/test_oracle_*.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
// The product (camlasercalibratool_amd/csrc) never includes, links or calls it.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference root).  Plain C++17, no dependencies; double precision throughout.
// =====================================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_types.h"

extern "C" {

void oracle_options_default(oracle_options* o) {
  o->max_num_iterations = 100;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->use_loss = 1;
  o->loss_scale_factor = 0.05;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
}

// -------------------------------------------------------------------------------------
// Eigen conversions (Eigen is absent).
// -------------------------------------------------------------------------------------

// Eigen::Quaterniond::toRotationMatrix() — used at LaseCamCalCeres.cpp:47,57,229,313.
// q given as (x,y,z,w) = storage order of the 7-vector (LaseCamCalCeres.cpp:219).
// Row-major R[9].  Does NOT normalise (Eigen does not either).
void oracle_quat_to_rot(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Eigen::Quaterniond(Matrix3d) — LaseCamCalCeres.cpp:215, calibr_simulation.cpp:19,60,62.
void oracle_rot_to_quat(const double m[9], double q[4]) {
  auto M = [&](int r, int c) { return m[3 * r + c]; };
  double t = M(0, 0) + M(1, 1) + M(2, 2);
  double qq[4];  // x,y,z,w
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    qq[3] = 0.5 * t;
    t = 0.5 / t;
    qq[0] = (M(2, 1) - M(1, 2)) * t;
    qq[1] = (M(0, 2) - M(2, 0)) * t;
    qq[2] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    qq[i] = 0.5 * t;
    t = 0.5 / t;
    qq[3] = (M(k, j) - M(j, k)) * t;
    qq[j] = (M(j, i) + M(i, j)) * t;
    qq[k] = (M(k, i) + M(i, k)) * t;
  }
  q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
}

// -------------------------------------------------------------------------------------
// PointInPlaneFactor::Evaluate — src/LaseCamCalCeres.cpp:43-66 (+ skewSymmetric :35-42).
// pose = parameters[0] = [tx,ty,tz,qx,qy,qz,qw].  jac7 (nullable) = row-major 1x7.
// Operation order follows the Eigen expressions (3-term dot = (a0b0 + a1b1) + a2b2).
// -------------------------------------------------------------------------------------
void oracle_factor_evaluate(const double plane[4], const double point[3], double scale,
                            const double pose[7], double* residual, double* jac7) {
  double R[9];
  oracle_quat_to_rot(pose + 3, R);  // qcl.toRotationMatrix(), :47
  double ptc[3];
  for (int i = 0; i < 3; ++i)       // pt_c = R * point_ + tcl, :47
    ptc[i] = ((R[3 * i] * point[0] + R[3 * i + 1] * point[1]) + R[3 * i + 2] * point[2]) + pose[i];
  // residuals[0] = scale_ * (planar_.head(3)^T * pt_c + planar_[3]), :48
  residual[0] = scale * (((plane[0] * ptc[0] + plane[1] * ptc[1]) + plane[2] * ptc[2]) + plane[3]);
  if (jac7) {
    // skewSymmetric(point_), :35-42
    const double S[9] = {0.0, -point[2], point[1], point[2], 0.0, -point[0], -point[1], point[0], 0.0};
    double M[9];  // -R * skew(p), :57
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        M[3 * i + j] = ((-R[3 * i]) * S[j] + (-R[3 * i + 1]) * S[3 + j]) + (-R[3 * i + 2]) * S[6 + j];
    double jaco[6];
    for (int j = 0; j < 3; ++j) jaco[j] = plane[j];  // :56
    for (int j = 0; j < 3; ++j)                       // :57
      jaco[3 + j] = (plane[0] * M[j] + plane[1] * M[3 + j]) + plane[2] * M[6 + j];
    for (int j = 0; j < 6; ++j) jac7[j] = scale * jaco[j];  // :59
    jac7[6] = 0.0;                                          // :60
  }
}

// -------------------------------------------------------------------------------------
// PoseLocalParameterization — src/pose_local_parameterization.cpp:3-40.
// -------------------------------------------------------------------------------------
void oracle_pose_plus(const double x[7], const double delta[6], double out[7]) {
  // p = _p + dp, :28
  out[0] = x[0] + delta[0]; out[1] = x[1] + delta[1]; out[2] = x[2] + delta[2];
  // dq = deltaQ(delta+3) = [w=1, v=theta/2] (NOT normalised), :3-13
  const double bx = delta[3] / 2.0, by = delta[4] / 2.0, bz = delta[5] / 2.0, bw = 1.0;
  const double ax = x[3], ay = x[4], az = x[5], aw = x[6];
  // q = (_q * dq).normalized(), :29   (Hamilton product, Appendix B)
  const double w = aw * bw - ax * bx - ay * by - az * bz;
  const double qx = aw * bx + ax * bw + ay * bz - az * by;
  const double qy = aw * by + ay * bw + az * bx - ax * bz;
  const double qz = aw * bz + az * bw + ax * by - ay * bx;
  const double n2 = qx * qx + qy * qy + qz * qz + w * w;
  const double n = std::sqrt(n2);
  out[3] = qx / n; out[4] = qy / n; out[5] = qz / n; out[6] = w / n;
}

// ComputeJacobian: 7x6 row-major [I6; 0] — pose_local_parameterization.cpp:33-40.
void oracle_pose_plus_jacobian(const double* /*x*/, double jac[42]) {
  for (int i = 0; i < 42; ++i) jac[i] = 0.0;
  for (int i = 0; i < 6; ++i) jac[6 * i + i] = 1.0;
}

// ceres::CauchyLoss(a)::Evaluate(s, rho) — Ceres loss_function.cc.
void oracle_cauchy(double a, double s, double rho[3]) {
  const double b = a * a;
  const double c = 1.0 / b;
  const double sum = 1.0 + s * c;
  const double inv = 1.0 / sum;
  rho[0] = b * std::log(sum);
  rho[1] = std::max(std::numeric_limits<double>::min(), inv);
  rho[2] = -c * (inv * inv);
}

// -------------------------------------------------------------------------------------
// pi_from_ppp — src/utilities.cpp:267-272 (un-normalised plane through 3 points).
// -------------------------------------------------------------------------------------
static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
void oracle_pi_from_ppp(const double x1[3], const double x2[3], const double x3[3], double pi[4]) {
  double a[3] = {x1[0] - x3[0], x1[1] - x3[1], x1[2] - x3[2]};
  double b[3] = {x2[0] - x3[0], x2[1] - x3[1], x2[2] - x3[2]};
  double n[3], c12[3];
  cross3(a, b, n);
  cross3(x1, x2, c12);
  pi[0] = n[0]; pi[1] = n[1]; pi[2] = n[2];
  pi[3] = -((x3[0] * c12[0] + x3[1] * c12[1]) + x3[2] * c12[2]);
}

// plane-in-camera-frame: (Tctag^-1)^T (0,0,1,0) — LaseCamCalCeres.cpp:227-231.
// Closed form of that 4x4 inverse-transpose: n = R_ca e3, d = -n . t_ca.
// tag_q is (w,x,y,z) = Eigen constructor order used by Oberserve (LaseCamCalCeres.h:16).
static void plane_in_camera(const double tag_q_wxyz[4], const double tag_t[3], double plane[4]) {
  const double q[4] = {tag_q_wxyz[1], tag_q_wxyz[2], tag_q_wxyz[3], tag_q_wxyz[0]};
  double R[9];
  oracle_quat_to_rot(q, R);
  plane[0] = R[2]; plane[1] = R[5]; plane[2] = R[8];
  plane[3] = -((plane[0] * tag_t[0] + plane[1] * tag_t[1]) + plane[2] * tag_t[2]);
}

// -------------------------------------------------------------------------------------
// Problem assembly loop — src/LaseCamCalCeres.cpp:222-295.
// Inputs are the flattened std::vector<Oberserve> (LaseCamCalCeres.h:11-24):
//   tag_q[n_poses*4] (w,x,y,z), tag_t[n_poses*3],
//   points CSR: pts_off[n_poses+1], pts[...*3]; points_on_line CSR likewise.
// Output: records[N*8] = {n(3), d, p(3), scale}; returns N, or -1 if boundary mode hits
// an empty `points` (the reference throws std::out_of_range at :278).
// If records == nullptr only counts.
// -------------------------------------------------------------------------------------
long long oracle_flatten(int n_poses, const double* tag_q, const double* tag_t,
                         const long long* pts_off, const double* pts,
                         const long long* ptl_off, const double* ptl,
                         int use_linefitting_data, int use_boundary_constraint,
                         double* records) {
  long long N = 0;
  for (int i = 0; i < n_poses; ++i) {
    double plane[4];
    plane_in_camera(tag_q + 4 * i, tag_t + 3 * i, plane);
    const long long* off = use_linefitting_data ? ptl_off : pts_off;  // :233-237
    const double* P = use_linefitting_data ? ptl : pts;
    const long long cnt = off[i + 1] - off[i];
    const double scale = 1.0 / std::sqrt((double)cnt);  // :239-240
    for (long long j = off[i]; j < off[i + 1]; ++j) {
      if (records) {
        double* r = records + 8 * N;
        r[0] = plane[0]; r[1] = plane[1]; r[2] = plane[2]; r[3] = plane[3];
        r[4] = P[3 * j]; r[5] = P[3 * j + 1]; r[6] = P[3 * j + 2];
        r[7] = scale;
      }
      ++N;
    }
    if (use_boundary_constraint && use_linefitting_data) {  // :258-294
      const long long np = pts_off[i + 1] - pts_off[i];
      if (np <= 0) return -1;  // obi.points.at(0) throws, :278
      const double orig[3] = {0.0265 + 0.0165, 0.0265 + 0.0165, 0.0};  // :262
      double pm[3][3] = {{0, 0, 0}, {0.5, 0, 0}, {0., 0.5, 0}};       // :263-265
      const double q[4] = {tag_q[4 * i + 1], tag_q[4 * i + 2], tag_q[4 * i + 3], tag_q[4 * i]};
      double R[9];
      oracle_quat_to_rot(q, R);
      double pc[3][3];
      for (int k = 0; k < 3; ++k) {
        for (int a = 0; a < 3; ++a) pm[k][a] -= orig[a];  // :266-268
        for (int a = 0; a < 3; ++a)                        // :270-272
          pc[k][a] = ((R[3 * a] * pm[k][0] + R[3 * a + 1] * pm[k][1]) + R[3 * a + 2] * pm[k][2]) + tag_t[3 * i + a];
      }
      const double zero[3] = {0, 0, 0};
      double pi1[4], pi2[4];
      oracle_pi_from_ppp(pc[0], pc[1], zero, pi1);  // :275
      oracle_pi_from_ppp(pc[0], pc[2], zero, pi2);  // :276
      const double* pt1 = pts + 3 * pts_off[i];             // obi.points.at(0), :278
      const double* pt2 = pts + 3 * (pts_off[i + 1] - 1);   // obi.points.at(size-1), :279
      if (records) {
        double* r = records + 8 * N;
        r[0] = pi1[0]; r[1] = pi1[1]; r[2] = pi1[2]; r[3] = pi1[3];
        r[4] = pt1[0]; r[5] = pt1[1]; r[6] = pt1[2]; r[7] = scale;
        r += 8;
        r[0] = pi2[0]; r[1] = pi2[1]; r[2] = pi2[2]; r[3] = pi2[3];
        r[4] = pt2[0]; r[5] = pt2[1]; r[6] = pt2[2]; r[7] = scale;
      }
      N += 2;
    }
  }
  return N;
}

// -------------------------------------------------------------------------------------
// One Ceres evaluation pass over all residual blocks (ResidualBlock::Evaluate +
// ProgramEvaluator): user Evaluate -> x[I6;0] -> Cauchy -> corrector -> cost, gradient.
// obs[N*8] = {n,d,p,scale}.  Outputs (nullable): residuals[N] (robustified),
// J[N*6] column-major with leading dimension ldj (robustified, local), g[6] = J^T r.
// Returns cost = sum_k 0.5*rho0_k summed in residual order.
// -------------------------------------------------------------------------------------
double oracle_evaluate(const double* obs, long long N, const double pose[7], int with_loss,
                       double loss_factor, double* residuals, double* J, long long ldj,
                       double* g) {
  double cost = 0.0;
  double gg[6] = {0, 0, 0, 0, 0, 0};
  const bool need_j = (J != nullptr) || (g != nullptr);
  for (long long k = 0; k < N; ++k) {
    const double* o = obs + 8 * k;
    double r, j7[7];
    oracle_factor_evaluate(o, o + 4, o[7], pose, &r, need_j ? j7 : nullptr);
    double jl[6] = {0, 0, 0, 0, 0, 0};
    if (need_j) {
      // jacobians = global(1x7) * [I6;0](7x6): exact selection of the first 6 entries.
      for (int c = 0; c < 6; ++c) jl[c] = j7[c];
    }
    const double sq = r * r;
    if (!with_loss) {
      cost += 0.5 * sq;
    } else {
      double rho[3];
      oracle_cauchy(loss_factor * o[7], sq, rho);  // CauchyLoss(0.05*scale), :249
      cost += 0.5 * rho[0];
      // Corrector: rho[2] <= 0 always for Cauchy -> residual_scaling = sqrt(rho1), alpha=0.
      const double sr = std::sqrt(rho[1]);
      if (need_j)
        for (int c = 0; c < 6; ++c) jl[c] *= sr;  // CorrectJacobian first
      r *= sr;                                    // then CorrectResiduals
    }
    if (residuals) residuals[k] = r;
    if (J)
      for (int c = 0; c < 6; ++c) J[c * ldj + k] = jl[c];
    if (need_j)
      for (int c = 0; c < 6; ++c) gg[c] += jl[c] * r;
  }
  if (g)
    for (int c = 0; c < 6; ++c) g[c] = gg[c];
  return cost;
}

// Normal-equation evaluation (what the GPU path accumulates): cost, g[6], H[21]
// (upper triangle, row-major: 00 01 .. 05 11 12 ..).  threads>1 uses OpenMP with a
// fixed static partition (deterministic for a fixed thread count).
double oracle_evaluate_ne(const double* obs, long long N, const double pose[7], int with_loss,
                          double loss_factor, double* g, double* H, int threads) {
  if (threads < 1) threads = 1;
  std::vector<double> part((size_t)threads * 28, 0.0);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
  {
    int t = 0, nt = 1;
#ifdef _OPENMP
    t = omp_get_thread_num();
    nt = omp_get_num_threads();
#endif
    const long long lo = N * t / nt, hi = N * (t + 1) / nt;
    double acc[28];
    for (int i = 0; i < 28; ++i) acc[i] = 0.0;
    for (long long k = lo; k < hi; ++k) {
      const double* o = obs + 8 * k;
      double r, j7[7];
      oracle_factor_evaluate(o, o + 4, o[7], pose, &r, j7);
      const double sq = r * r;
      if (!with_loss) {
        acc[27] += 0.5 * sq;
      } else {
        double rho[3];
        oracle_cauchy(loss_factor * o[7], sq, rho);
        acc[27] += 0.5 * rho[0];
        const double sr = std::sqrt(rho[1]);
        for (int c = 0; c < 6; ++c) j7[c] *= sr;
        r *= sr;
      }
      int idx = 0;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) acc[idx++] += j7[a] * j7[b];
      for (int a = 0; a < 6; ++a) acc[21 + a] += j7[a] * r;
    }
    for (int i = 0; i < 28; ++i) part[(size_t)t * 28 + i] = acc[i];
  }
  double tot[28];
  for (int i = 0; i < 28; ++i) tot[i] = 0.0;
  for (int t = 0; t < threads; ++t)
    for (int i = 0; i < 28; ++i) tot[i] += part[(size_t)t * 28 + i];
  if (H) for (int i = 0; i < 21; ++i) H[i] = tot[i];
  if (g) for (int i = 0; i < 6; ++i) g[i] = tot[21 + i];
  return tot[27];
}

// Per-observation dump for element-wise diffs: raw (un-robustified) residual[N] and
// 1x7 global Jacobian rows jac[N*7] exactly as PointInPlaneFactor::Evaluate writes them.
void oracle_factor_evaluate_batch(const double* obs, long long N, const double pose[7],
                                  double* residuals, double* jac7) {
  for (long long k = 0; k < N; ++k) {
    const double* o = obs + 8 * k;
    oracle_factor_evaluate(o, o + 4, o[7], pose, residuals + k, jac7 ? jac7 + 7 * k : nullptr);
  }
}

// -------------------------------------------------------------------------------------
// Dense linear algebra used by the LM loop.
// -------------------------------------------------------------------------------------

// Householder QR least squares min ||A y - b||, A (m x n) column-major ld=m, overwritten.
// Mirrors Eigen::HouseholderQR (unblocked path; make_householder + apply on the left)
// as called by Ceres DenseQRSolver: x = A.householderQr().solve(rhs).
static void householder_qr_solve(double* A, long long m, int n, double* b, double* y) {
  for (int k = 0; k < n; ++k) {
    double* col = A + (size_t)k * m;
    double tail_sq = 0.0;
    for (long long i = k + 1; i < m; ++i) tail_sq += col[i] * col[i];
    const double c0 = col[k];
    double tau, beta;
    if (tail_sq <= DBL_MIN) {
      tau = 0.0; beta = c0;
      for (long long i = k + 1; i < m; ++i) col[i] = 0.0;
    } else {
      beta = std::sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.0) beta = -beta;
      const double denom = c0 - beta;
      for (long long i = k + 1; i < m; ++i) col[i] /= denom;  // essential part
      tau = (beta - c0) / beta;
    }
    col[k] = beta;
    if (tau != 0.0) {
      // apply H = I - tau v v^T (v = [1; essential]) to remaining columns and rhs
      for (int j = k + 1; j <= n; ++j) {
        double* t = (j < n) ? (A + (size_t)j * m) : b;
        double s = t[k];
        for (long long i = k + 1; i < m; ++i) s += col[i] * t[i];
        s *= tau;
        t[k] -= s;
        for (long long i = k + 1; i < m; ++i) t[i] -= s * col[i];
      }
    }
  }
  for (int i = n - 1; i >= 0; --i) {  // back substitution with R (upper n x n)
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A[(size_t)j * m + i] * y[j];
    y[i] = s / A[(size_t)i * m + i];
  }
}

// n x n SPD solve by Cholesky (normal-equation variant), n <= 9. Returns false if not PD.
static bool cholesky_solve(const double* Ain, const double* b, double* y, int n) {
  double L[81], z[9];
  for (int i = 0; i < n * n; ++i) L[i] = 0.0;
  for (int j = 0; j < n; ++j) {
    double d = Ain[n * j + j];
    for (int k = 0; k < j; ++k) d -= L[n * j + k] * L[n * j + k];
    if (!(d > 0.0)) return false;
    const double ljj = std::sqrt(d);
    L[n * j + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double s = Ain[n * i + j];
      for (int k = 0; k < j; ++k) s -= L[n * i + k] * L[n * j + k];
      L[n * i + j] = s / ljj;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[n * i + k] * z[k];
    z[i] = s / L[n * i + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < n; ++k) s -= L[n * k + i] * y[k];
    y[i] = s / L[n * i + i];
  }
  return true;
}

}  // extern "C" (helpers below use C++ types)

#include <functional>

// A least-squares problem as Ceres sees it: `np` local / `na` ambient parameters, N residuals.
//   eval_dense(x, res, J, ldj, g): one evaluation pass -> cost; robustified residuals res[N]
//       (nullable), robustified local Jacobian J (N x np, column-major, leading dim ldj; nullable),
//       gradient g[np] = J^T res (nullable).
//   eval_ne(x, g, H): the same pass reduced to the normal equation (H packed upper triangle).
//   plus(x, delta, out): the LocalParameterization's Plus.
struct LmProblem {
  int np, na;
  long long n_res;
  std::function<double(const double*, double*, double*, long long, double*)> eval_dense;
  std::function<double(const double*, double*, double*)> eval_ne;
  std::function<void(const double*, const double*, double*)> plus;
};

// -------------------------------------------------------------------------------------
// ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT / DENSE_QR as configured at
// src/LaseCamCalCeres.cpp:299-307 (pose problem) and :423-428 (line fit).  Control flow
// restates Ceres 1.13-2.1 TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy +
// TrustRegionStepEvaluator (monotonic) — SURVEY.md Appendix A, with three details taken from
// the Ceres source rather than the appendix's summary:
//   * gradient_max_norm = ||x - Plus(x, -g)||_inf in the ambient space;
//   * an invalid step calls StepRejected(0): radius /= decrease_factor; decrease_factor *= 2;
//   * final_cost = min over recorded iteration costs (Solver::SetSummaryFinalCost).
// linear_solver: 0 = DENSE_QR on the dense N x np Jacobian (what the reference runs),
//                1 = np x np normal equations + Cholesky (what the GPU path runs).
// -------------------------------------------------------------------------------------
static int lm_minimize(const LmProblem& pb, const oracle_options* opt, double* xio, oracle_summary* summary,
                       oracle_iteration* trace, int trace_cap, int linear_solver) {
  const int np = pb.np, na = pb.na;
  const long long N = pb.n_res;
  const bool use_qr = (linear_solver == 0);
  std::vector<double> J, Jaug, res, rhs;
  const long long m = N + np;
  if (use_qr) {
    J.resize((size_t)N * np);
    Jaug.resize((size_t)m * np);
    res.resize((size_t)N);
    rhs.resize((size_t)m);
  }
  const int nh = np * (np + 1) / 2;
  std::vector<double> x(na), x_cand(na), g(np), H(nh), scale(np, 1.0), diag(np), step(np), delta(np);
  for (int i = 0; i < na; ++i) x[i] = xio[i];
  auto norm_a = [&](const double* v) { double s = 0.0; for (int i = 0; i < na; ++i) s += v[i] * v[i]; return std::sqrt(s); };
  double x_cost = 0.0, candidate_cost = 0.0, model_cost_change = 0.0;
  double x_norm = norm_a(x.data());
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int n_invalid = 0;
  double minimum_cost = std::numeric_limits<double>::max();
  int n_trace = 0;
  summary->num_successful_steps = 0;
  summary->num_unsuccessful_steps = 0;
  summary->num_residual_evaluations = 0;
  summary->num_jacobian_evaluations = 0;
  summary->termination = ORACLE_NO_CONVERGENCE;
  double min_iter_cost = 0.0;
  oracle_iteration it;
  std::memset(&it, 0, sizeof(it));
  double last_gmax = 0.0;

  // EvaluateGradientAndJacobian(): evaluate at x, (iteration 0) compute Jacobi scaling,
  // scale columns, projected gradient norms.
  auto evaluate_gradient_and_jacobian = [&](bool first) {
    if (use_qr) x_cost = pb.eval_dense(x.data(), res.data(), J.data(), N, g.data());
    else x_cost = pb.eval_ne(x.data(), g.data(), H.data());
    summary->num_residual_evaluations++;
    summary->num_jacobian_evaluations++;
    if (opt->jacobi_scaling) {
      if (first) {
        for (int c = 0; c < np; ++c) {
          double sq = 0.0;
          if (use_qr) {
            const double* col = J.data() + (size_t)c * N;
            for (long long k = 0; k < N; ++k) sq += col[k] * col[k];
          } else {
            int idx = 0;
            for (int a = 0; a < c; ++a) idx += np - a;
            sq = H[idx];
          }
          scale[c] = 1.0 / (1.0 + std::sqrt(sq));
        }
      }
      if (use_qr) {
        for (int c = 0; c < np; ++c) {
          double* col = J.data() + (size_t)c * N;
          const double s = scale[c];
          for (long long k = 0; k < N; ++k) col[k] *= s;
        }
      }
    }
    std::vector<double> neg_g(np), proj(na);
    for (int c = 0; c < np; ++c) neg_g[c] = -g[c];
    pb.plus(x.data(), neg_g.data(), proj.data());
    double gmax = 0.0;
    for (int i = 0; i < na; ++i) gmax = std::max(gmax, std::fabs(x[i] - proj[i]));
    it.gradient_max_norm = gmax;
    last_gmax = gmax;
    it.cost = x_cost;
  };

  // scaled normal-equation pieces for the NE variant
  auto scaled_H = [&](double* Hs, double* gs) {
    int idx = 0;
    for (int a = 0; a < np; ++a)
      for (int b = a; b < np; ++b) {
        const double v = H[idx++] * scale[a] * scale[b];
        Hs[np * a + b] = v;
        Hs[np * b + a] = v;
      }
    for (int a = 0; a < np; ++a) gs[a] = g[a] * scale[a];
  };

  // ---- IterationZero ----
  it.iteration = 0;
  evaluate_gradient_and_jacobian(true);
  summary->initial_cost = x_cost;
  it.step_is_valid = 1;
  it.step_is_successful = 1;
  min_iter_cost = x_cost;

  int status = 0;
  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
    if (it.step_is_successful) {
      summary->num_successful_steps++;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
        for (int i = 0; i < na; ++i) xio[i] = x[i];
      }
    } else {
      summary->num_unsuccessful_steps++;
    }
    it.trust_region_radius = radius;
    if (trace && n_trace < trace_cap) trace[n_trace] = it;
    n_trace++;
    min_iter_cost = std::min(min_iter_cost, it.cost);
    if (it.iteration >= opt->max_num_iterations) { status = ORACLE_NO_CONVERGENCE; break; }
    if (it.step_is_successful && it.gradient_max_norm <= opt->gradient_tolerance) {
      status = ORACLE_CONVERGENCE_GRADIENT; break;
    }
    if (it.trust_region_radius <= opt->min_trust_region_radius) {
      status = ORACLE_CONVERGENCE_RADIUS; break;
    }

    const int next_iter = it.iteration + 1;
    std::memset(&it, 0, sizeof(it));
    it.iteration = next_iter;

    // ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep) ----
    bool solved = true;
    if (use_qr) {
      if (!reuse_diagonal) {
        for (int c = 0; c < np; ++c) {
          const double* col = J.data() + (size_t)c * N;
          double sq = 0.0;
          for (long long k = 0; k < N; ++k) sq += col[k] * col[k];
          diag[c] = std::min(std::max(sq, opt->min_lm_diagonal), opt->max_lm_diagonal);
        }
      }
      std::vector<double> lm_diag(np), y(np);
      for (int c = 0; c < np; ++c) lm_diag[c] = std::sqrt(diag[c] / radius);
      for (int c = 0; c < np; ++c) {  // [J; diag(D)]
        std::memcpy(Jaug.data() + (size_t)c * m, J.data() + (size_t)c * N, sizeof(double) * (size_t)N);
        for (int r = 0; r < np; ++r) Jaug[(size_t)c * m + N + r] = (r == c) ? lm_diag[c] : 0.0;
      }
      if (N > 0) std::memcpy(rhs.data(), res.data(), sizeof(double) * (size_t)N);
      for (int r = 0; r < np; ++r) rhs[(size_t)N + r] = 0.0;
      householder_qr_solve(Jaug.data(), m, np, rhs.data(), y.data());
      for (int c = 0; c < np; ++c) {
        if (!std::isfinite(y[c])) solved = false;
        step[c] = -y[c];
      }
      reuse_diagonal = true;
      if (solved) {
        // model_cost_change = -(J step)^T (r + J step / 2)
        double mcc = 0.0;
        for (long long k = 0; k < N; ++k) {
          double mr = 0.0;
          for (int c = 0; c < np; ++c) mr += J[(size_t)c * N + k] * step[c];
          mcc += mr * (res[k] + mr / 2.0);
        }
        model_cost_change = -mcc;
      }
    } else {
      double Hs[81], gs[9], A[81], y[9];
      scaled_H(Hs, gs);
      if (!reuse_diagonal)
        for (int c = 0; c < np; ++c)
          diag[c] = std::min(std::max(Hs[np * c + c], opt->min_lm_diagonal), opt->max_lm_diagonal);
      for (int i = 0; i < np * np; ++i) A[i] = Hs[i];
      for (int c = 0; c < np; ++c) {
        const double d = std::sqrt(diag[c] / radius);
        A[np * c + c] += d * d;
      }
      solved = cholesky_solve(A, gs, y, np);
      for (int c = 0; c < np; ++c) {
        if (solved && !std::isfinite(y[c])) solved = false;
        step[c] = -y[c];
      }
      reuse_diagonal = true;
      if (solved) {
        double sg = 0.0, shs = 0.0;
        for (int a = 0; a < np; ++a) {
          sg += step[a] * gs[a];
          double row = 0.0;
          for (int b = 0; b < np; ++b) row += Hs[np * a + b] * step[b];
          shs += step[a] * row;
        }
        model_cost_change = -(sg + 0.5 * shs);
      }
    }
    it.step_is_valid = (solved && model_cost_change > 0.0) ? 1 : 0;

    if (!it.step_is_valid) {
      // ---- HandleInvalidStep ----
      if (++n_invalid >= opt->max_num_consecutive_invalid_steps) { status = ORACLE_FAILURE; break; }
      radius = radius / decrease_factor;  // StepIsInvalid() == StepRejected(0.0)
      decrease_factor *= 2.0;
      reuse_diagonal = true;
      it.cost = x_cost;
      it.cost_change = 0.0;
      it.gradient_max_norm = last_gmax;
      it.step_norm = 0.0;
      it.relative_decrease = 0.0;
      it.step_is_successful = 0;
      continue;
    }
    n_invalid = 0;
    for (int c = 0; c < np; ++c) delta[c] = step[c] * scale[c];  // undo column scaling

    // ---- ComputeCandidatePointAndEvaluateCost ----
    pb.plus(x.data(), delta.data(), x_cand.data());
    if (use_qr) candidate_cost = pb.eval_dense(x_cand.data(), nullptr, nullptr, 0, nullptr);
    else candidate_cost = pb.eval_ne(x_cand.data(), nullptr, nullptr);
    summary->num_residual_evaluations++;

    // ---- ParameterToleranceReached ----
    {
      double s = 0.0;
      for (int i = 0; i < na; ++i) s += (x[i] - x_cand[i]) * (x[i] - x_cand[i]);
      it.step_norm = std::sqrt(s);
      const double tol = opt->parameter_tolerance * (x_norm + opt->parameter_tolerance);
      if (it.step_norm <= tol) { status = ORACLE_CONVERGENCE_PARAMETER; break; }
    }
    // ---- FunctionToleranceReached ----
    it.cost_change = x_cost - candidate_cost;
    if (std::fabs(it.cost_change) <= opt->function_tolerance * x_cost) {
      status = ORACLE_CONVERGENCE_FUNCTION; break;
    }
    // ---- IsStepSuccessful (monotonic TrustRegionStepEvaluator) ----
    it.relative_decrease = it.cost_change / model_cost_change;
    if (it.relative_decrease > opt->min_relative_decrease) {
      // ---- HandleSuccessfulStep ----
      for (int i = 0; i < na; ++i) x[i] = x_cand[i];
      x_norm = norm_a(x.data());
      evaluate_gradient_and_jacobian(false);
      it.step_is_successful = 1;
      const double q = 2.0 * it.relative_decrease - 1.0;  // StepAccepted
      radius = radius / std::max(1.0 / 3.0, 1.0 - q * q * q);
      radius = std::min(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      // ---- HandleUnsuccessfulStep ----
      it.step_is_successful = 0;
      radius = radius / decrease_factor;  // StepRejected
      decrease_factor *= 2.0;
      reuse_diagonal = true;
      it.cost = candidate_cost;
      it.gradient_max_norm = last_gmax;
    }
  }
  summary->termination = status;
  summary->num_iterations = n_trace - 1;
  summary->final_cost = std::min(summary->initial_cost, min_iter_cost);
  return status;
}

extern "C" {

// ceres::Solve on the pose problem of CamLaserCalibration (src/LaseCamCalCeres.cpp:299-307).
// threads: only used by linear_solver=1 (OpenMP evaluation).  pose is in/out (:219, :311-314).
int oracle_solve(const double* obs, long long N, const oracle_options* opt, double pose[7],
                 oracle_summary* summary, oracle_iteration* trace, int trace_cap,
                 int linear_solver, int threads) {
  LmProblem pb;
  pb.np = 6; pb.na = 7; pb.n_res = N;
  pb.eval_dense = [&](const double* x, double* res, double* J, long long ldj, double* g) {
    return oracle_evaluate(obs, N, x, opt->use_loss, opt->loss_scale_factor, res, J, ldj, g);
  };
  pb.eval_ne = [&](const double* x, double* g, double* H) {
    return oracle_evaluate_ne(obs, N, x, opt->use_loss, opt->loss_scale_factor, g, H, threads);
  };
  pb.plus = [](const double* x, const double* d, double* out) { oracle_pose_plus(x, d, out); };
  return lm_minimize(pb, opt, pose, summary, trace, trace_cap, linear_solver);
}

// -------------------------------------------------------------------------------------
// The same minimiser behind C callbacks: lets a caller that owns its residual blocks (the stand-in
// ceres::Solve of oracle/ref_shim, which drives the reference's own cost functions) run exactly
// the LM restatement above.  DENSE_QR path only.
//   eval(ctx, x, res, J, ldj, g) / plus(ctx, x, delta, out): as in LmProblem.
// -------------------------------------------------------------------------------------
extern "C" {
typedef double (*oracle_eval_cb)(void* ctx, const double* x, double* res, double* J, long long ldj, double* g);
typedef void (*oracle_plus_cb)(void* ctx, const double* x, const double* delta, double* out);
int oracle_minimize_cb(int np, int na, long long n_res, oracle_eval_cb eval, oracle_plus_cb plus, void* ctx,
                       const oracle_options* opt, double* x, oracle_summary* summary, oracle_iteration* trace,
                       int trace_cap) {
  LmProblem pb;
  pb.np = np; pb.na = na; pb.n_res = n_res;
  pb.eval_dense = [=](const double* xx, double* res, double* J, long long ldj, double* g) {
    return eval(ctx, xx, res, J, ldj, g);
  };
  pb.eval_ne = [](const double*, double*, double*) { return 0.0; };  // unused with DENSE_QR
  pb.plus = [=](const double* xx, const double* d, double* out) { plus(ctx, xx, d, out); };
  return lm_minimize(pb, opt, x, summary, trace, trace_cap, /*linear_solver=*/0);
}
}

// -------------------------------------------------------------------------------------
// LineFittingCeres — src/LaseCamCalCeres.cpp:385-433.  Per scan: residual_i = m0 x_i + m1 y_i + 1
// (LineFittingResidfual :390-391; autodiff Jacobian = [x_i, y_i]), CauchyLoss(0.05) per residual
// (:416), DENSE_QR, max_num_iterations = 10 (:424-425), other Ceres defaults; `line` in/out
// (:403, :430-431).  xy[n*2] are the x,y of the scan points (z is not used, :412).
// loss_a = 0.05; opt supplies the Ceres defaults with max_num_iterations = 10.
// -------------------------------------------------------------------------------------
double oracle_line_evaluate(const double* xy, long long n, const double line[2], int with_loss, double loss_a,
                            double* res, double* J, long long ldj, double* g, double* H3) {
  double cost = 0.0, gg[2] = {0, 0}, hh[3] = {0, 0, 0};
  const bool need_j = J || g || H3;
  for (long long k = 0; k < n; ++k) {
    const double x = xy[2 * k], y = xy[2 * k + 1];
    double r = line[0] * x + line[1] * y + 1.0;  // :391
    double j0 = x, j1 = y;
    const double sq = r * r;
    if (!with_loss) {
      cost += 0.5 * sq;
    } else {
      double rho[3];
      oracle_cauchy(loss_a, sq, rho);
      cost += 0.5 * rho[0];
      const double sr = std::sqrt(rho[1]);
      if (need_j) { j0 *= sr; j1 *= sr; }
      r *= sr;
    }
    if (res) res[k] = r;
    if (J) { J[k] = j0; J[ldj + k] = j1; }
    if (need_j) {
      gg[0] += j0 * r; gg[1] += j1 * r;
      hh[0] += j0 * j0; hh[1] += j0 * j1; hh[2] += j1 * j1;
    }
  }
  if (g) { g[0] = gg[0]; g[1] = gg[1]; }
  if (H3) { H3[0] = hh[0]; H3[1] = hh[1]; H3[2] = hh[2]; }
  return cost;
}

int oracle_line_fit(const double* xy, long long n, const oracle_options* opt, double loss_a, double line[2],
                    oracle_summary* summary, oracle_iteration* trace, int trace_cap, int linear_solver) {
  LmProblem pb;
  pb.np = 2; pb.na = 2; pb.n_res = n;
  pb.eval_dense = [&](const double* x, double* res, double* J, long long ldj, double* g) {
    return oracle_line_evaluate(xy, n, x, opt->use_loss, loss_a, res, J, ldj, g, nullptr);
  };
  pb.eval_ne = [&](const double* x, double* g, double* H) {
    return oracle_line_evaluate(xy, n, x, opt->use_loss, loss_a, nullptr, nullptr, 0, g, H);
  };
  pb.plus = [](const double* x, const double* d, double* out) { out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; };
  return lm_minimize(pb, opt, line, summary, trace, trace_cap, linear_solver);
}

// -------------------------------------------------------------------------------------
// Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9): eigenvalues w[n]
// (descending) and eigenvectors V (row-major, columns). For SPD matrices these are the
// singular values / right singular vectors Eigen::JacobiSVD returns at
// LaseCamCalCeres.cpp:162,366.
// -------------------------------------------------------------------------------------
static void jacobi_eig_sym(const double* Ain, int n, double* w, double* V) {
  double A[81];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off == 0.0) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int order[9];
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order, order + n, [&](int a, int b) { return A[a * n + a] > A[b * n + b]; });
  double Vt[81];
  for (int i = 0; i < n * n; ++i) Vt[i] = V[i];
  for (int c = 0; c < n; ++c) {
    w[c] = A[order[c] * n + order[c]];
    for (int r = 0; r < n; ++r) V[r * n + c] = Vt[r * n + order[c]];
  }
}

// -------------------------------------------------------------------------------------
// Post-solve information analysis — src/LaseCamCalCeres.cpp:316-381.
// Second pass WITHOUT loss and WITHOUT boundary terms (caller passes the point records
// only): H = sum J^T J (6x6 row-major), b = -sum J^T r, chi = sum r^2; singular values
// of H (descending) and V; n_null = #sv < 1e-8.
// -------------------------------------------------------------------------------------
void oracle_information(const double* obs, long long N, const double pose[7], double H[36],
                        double b[6], double* chi2, double sv[6], double V[36], int* n_null) {
  for (int i = 0; i < 36; ++i) H[i] = 0.0;
  for (int i = 0; i < 6; ++i) b[i] = 0.0;
  double chi = 0.0;
  for (long long k = 0; k < N; ++k) {
    const double* o = obs + 8 * k;
    double r, j7[7];
    oracle_factor_evaluate(o, o + 4, o[7], pose, &r, j7);
    for (int a = 0; a < 6; ++a)
      for (int c = 0; c < 6; ++c) H[6 * a + c] += j7[a] * j7[c];  // :356
    for (int a = 0; a < 6; ++a) b[a] -= j7[a] * r;                 // :357
    chi += r * r;                                                  // :359
  }
  *chi2 = chi;
  jacobi_eig_sym(H, 6, sv, V);
  int n = 0;
  for (int i = 0; i < 6; ++i)
    if (sv[i] < 1e-8) n++;  // :371
  *n_null = n;
}

// -------------------------------------------------------------------------------------
// U V^T of the singular value decomposition M = U S V^T of a 3x3 matrix
// (LaseCamCalCeres.cpp:195-196 — "nearest rotation", no determinant check).  V, S^2 from the
// eigen-decomposition of M^T M; U column by column as M v / sigma; columns with sigma = 0 (rank
// deficient M) are filled in so that U is orthogonal, as a full SVD guarantees.
// -------------------------------------------------------------------------------------
static void polar_orthogonal3(const double M[9], double Q[9]) {
  double MtM[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += M[3 * k + i] * M[3 * k + j];
      MtM[3 * i + j] = s;
    }
  double w[3], V[9], U[9];
  jacobi_eig_sym(MtM, 3, w, V);  // descending
  const double cutoff = 1e-14 * std::sqrt(w[0] > 0.0 ? w[0] : 0.0);
  int rank = 0;
  for (int c = 0; c < 3; ++c) {
    double u[3];
    for (int r = 0; r < 3; ++r) u[r] = (M[3 * r] * V[c] + M[3 * r + 1] * V[3 + c]) + M[3 * r + 2] * V[6 + c];
    const double len = std::sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
    if (!(len > cutoff) || len == 0.0) break;  // sorted: the rest vanish too
    for (int r = 0; r < 3; ++r) U[3 * r + c] = u[r] / len;
    ++rank;
  }
  if (rank == 0) { U[0] = 1.0; U[3] = 0.0; U[6] = 0.0; rank = 1; }
  if (rank == 1) {  // a unit vector orthogonal to column 0: cross with the axis it is least aligned with
    const double a[3] = {U[0], U[3], U[6]};
    int m = 0;
    if (std::fabs(a[1]) < std::fabs(a[m])) m = 1;
    if (std::fabs(a[2]) < std::fabs(a[m])) m = 2;
    double e[3] = {0.0, 0.0, 0.0}, c[3];
    e[m] = 1.0;
    cross3(a, e, c);
    const double len = std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
    for (int r = 0; r < 3; ++r) U[3 * r + 1] = c[r] / len;
    rank = 2;
  }
  if (rank == 2) {
    const double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    double c[3];
    cross3(a, b, c);
    for (int r = 0; r < 3; ++r) U[3 * r + 2] = c[r];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Q[3 * i + j] = (U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1]) + U[3 * i + 2] * V[3 * j + 2];
}

// AtA.ldlt().solve at :181.  Eigen::LDLT is a Cholesky factorisation with symmetric pivoting on the
// largest remaining diagonal entry, P A P^T = L D L^T, whose solve uses the pseudo-inverse of D
// (a zero pivot yields a zero component) — defined for semi-definite matrices as well, which is
// what the reference relies on when it carries on after the "system unobservable" notice.
static void ldlt_pivoted_solve(const double* A_in, const double* rhs, double* x, int n) {
  std::vector<double> A(A_in, A_in + n * n), D(n), y(n);
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) p[i] = i;
  for (int k = 0; k < n; ++k) {
    int best = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[i * n + i]) > std::fabs(A[best * n + best])) best = i;
    if (best != k) {
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[best * n + j]);
      for (int i = 0; i < n; ++i) std::swap(A[i * n + k], A[i * n + best]);
      std::swap(p[k], p[best]);
    }
    double dk = A[k * n + k];
    for (int j = 0; j < k; ++j) dk -= A[k * n + j] * A[k * n + j] * D[j];
    D[k] = dk;
    for (int i = k + 1; i < n; ++i) {
      double v = A[i * n + k];
      for (int j = 0; j < k; ++j) v -= A[i * n + j] * A[k * n + j] * D[j];
      A[i * n + k] = dk != 0.0 ? v / dk : 0.0;
    }
  }
  for (int i = 0; i < n; ++i) y[i] = rhs[p[i]];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) y[i] = std::fabs(D[i]) > 2.2250738585072014e-308 ? y[i] / D[i] : 0.0;
  for (int i = n - 1; i >= 0; --i)
    for (int j = i + 1; j < n; ++j) y[i] -= A[j * n + i] * y[j];
  for (int i = 0; i < n; ++i) x[p[i]] = y[i];
}

// -------------------------------------------------------------------------------------
// CamLaserCalClosedSolution — src/LaseCamCalCeres.cpp:112-203.
// Input: records of the points_on_line observations ({n,d,p,scale}; only n,d,p.x,p.y
// are used, :147).  Output Tlc[16] row-major 4x4, unobservable flag (:164-171),
// sv9[9] singular values of AtA.  Returns 0 (like the reference it always produces a Tlc).
// -------------------------------------------------------------------------------------
int oracle_closed_form(const double* obs, long long N, double Tlc[16], int* unobservable,
                       double sv9[9]) {
  double AtA[81], Atb[9];
  for (int i = 0; i < 81; ++i) AtA[i] = 0.0;
  for (int i = 0; i < 9; ++i) Atb[i] = 0.0;
  for (long long k = 0; k < N; ++k) {
    const double* o = obs + 8 * k;
    const double bar[3] = {o[4], o[5], 1.0};  // :147
    double Ai[9];
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) Ai[3 * c + r] = o[r] * bar[c];  // :150-152
    const double bi = -o[3];                                       // :155
    for (int a = 0; a < 9; ++a) {
      for (int c = 0; c < 9; ++c) AtA[9 * a + c] += Ai[a] * Ai[c];  // :161
      Atb[a] += Ai[a] * bi;
    }
  }
  double V[81];
  jacobi_eig_sym(AtA, 9, sv9, V);  // :162
  *unobservable = 0;
  for (int i = 0; i < 9; ++i)
    if (sv9[i] < 1e-10) *unobservable = 1;  // :167
  double Hh[9];
  ldlt_pivoted_solve(AtA, Atb, Hh, 9);  // :181
  const double* h1 = Hh; const double* h2 = Hh + 3; const double* h3 = Hh + 6;
  double h12[3];
  cross3(h1, h2, h12);
  // Rcl columns = h1, h2, h1 x h2 (:187-190); Rlc = Rcl^T (:191)
  double Rlc[9] = {h1[0], h1[1], h1[2], h2[0], h2[1], h2[2], h12[0], h12[1], h12[2]};
  double tlc[3];
  for (int i = 0; i < 3; ++i)  // tlc = -Rlc * h3, BEFORE orthogonalisation (:192)
    tlc[i] = -((Rlc[3 * i] * h3[0] + Rlc[3 * i + 1] * h3[1]) + Rlc[3 * i + 2] * h3[2]);
  double Q[9];
  polar_orthogonal3(Rlc, Q);  // U V^T (:195-196)
  for (int i = 0; i < 16; ++i) Tlc[i] = 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Tlc[4 * i + j] = Q[3 * i + j];
    Tlc[4 * i + 3] = tlc[i];
  }
  Tlc[15] = 1.0;
  return 0;
}

// -------------------------------------------------------------------------------------
// TranScanToPoints — src/utilities.cpp:181-215.  ranges are float32 as in
// sensor_msgs/LaserScan; angle_min / angle_increment / range_min are the message's float32
// fields (promoted to double in the expressions, :192-193,:204).  points[n*3].
// -------------------------------------------------------------------------------------
void oracle_scan_to_points(const float* ranges, long long n, float angle_min, float angle_increment,
                           float range_min, double* points) {
  for (long long i = 0; i < n; ++i) {
    const double c = std::cos(angle_min + (double)i * angle_increment);  // :192
    const double s = std::sin(angle_min + (double)i * angle_increment);  // :193
    const double x = (double)ranges[i] * c, y = (double)ranges[i] * s;   // :189-190,:196
    const double range_cutoff = 30.0;                                    // :201
    const float range = ranges[i];
    if (range < range_cutoff && range >= range_min) {                    // :203
      points[3 * i] = x; points[3 * i + 1] = y; points[3 * i + 2] = 0.0;
    } else {
      points[3 * i] = 1000.0; points[3 * i + 1] = 1000.0; points[3 * i + 2] = 0.0;  // :208
    }
  }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
