// clc_math.hpp — scalar FP64 building blocks of the point-to-plane path, written once for
// device code (hipcc) and for the host-side unit shim (g++, tests only).
//
//   quat_to_rot      Eigen::Quaterniond::toRotationMatrix  (src/LaseCamCalCeres.cpp:47,57,313)
//   pose_plus        PoseLocalParameterization::Plus       (src/pose_local_parameterization.cpp:15-31)
//   chol6_solve      the 6x6 damped normal-equation solve that replaces Ceres' DENSE_QR of [J;D]
//   jacobi_eig_sym   symmetric eigen-solver standing in for Eigen::JacobiSVD on SPD input
//                    (src/LaseCamCalCeres.cpp:162,366)
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CLC_HD __host__ __device__ __forceinline__
#define CLC_ROLLED
#else
#define CLC_HD inline
#define CLC_ROLLED
#endif

namespace clc {

// 1/sqrt(x) for x > 0.  Device: v_rsq_f64 seed (~2^-23) + two Newton steps (error ~1 ulp), 9 instructions instead of
// the ~27 of an IEEE sqrt followed by an IEEE division — the LM controller is a single lane's FP64 issue stream and
// pays for every one of them (eight reciprocal square roots per step: six Cholesky pivots, two Plus).
// Host (unit shim, tests only): the plain expression.
CLC_HD double rsqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = fma(y, fma(-h * y, y, 0.5), y);
  y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

// 1/x for x > 0 (<= 1 ulp): v_rcp_f64 seed + two Newton steps, 5 instructions (~40 cycles of dependent latency)
// instead of the ~10-instruction, 70-100-cycle IEEE division sequence (scripts/probes/latency_probe.hip) — the
// controller's divisions (1/radius, cost_change/model_cost_change, radius/den) sit on its critical path.
CLC_HD double rcp_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
#else
  return 1.0 / x;
#endif
}

// 1/x for x > 0 that may be tiny (the controller's model cost change): rcp_pos, and the IEEE quotient where it breaks
// down — for a subnormal x the v_rcp_f64 seed overflows to +inf and the Newton steps turn it into NaN, where Ceres'
// cost_change / model_cost_change is a plain division (+inf for a positive cost change: the step is accepted).
CLC_HD double rcp_pos_safe(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = rcp_pos(x);
  if (!(fabs(y) <= 1.7976931348623157e308)) y = 1.0 / x;
  return y;
#else
  return 1.0 / x;
#endif
}

// sqrt(x) for x >= 0 as x * rsqrt(x) (<= 2 ulp; 0 for 0): used for norms that only feed tolerance tests and the trace.
CLC_HD double sqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = x * rsqrt_pos(x);  // NaN for x == 0 (0 * inf): selected away, no branch
  return x > 0.0 ? r : 0.0;
#else
  return sqrt(x);
#endif
}

// q = (x,y,z,w) as stored in the 7-vector (src/LaseCamCalCeres.cpp:219); R row-major.
// No normalisation, like Eigen.
CLC_HD void quat_to_rot(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// p+ = p + dp ; q+ = normalize(q (x) [1, dtheta/2])  — right-multiplicative, dq not normalised.
CLC_HD void pose_plus(const double* x, const double* delta, double* out) {
  out[0] = x[0] + delta[0];
  out[1] = x[1] + delta[1];
  out[2] = x[2] + delta[2];
  const double bx = delta[3] / 2.0, by = delta[4] / 2.0, bz = delta[5] / 2.0;
  const double ax = x[3], ay = x[4], az = x[5], aw = x[6];
  const double w = aw - ax * bx - ay * by - az * bz;
  const double qx = aw * bx + ax + ay * bz - az * by;
  const double qy = aw * by + ay + az * bx - ax * bz;
  const double qz = aw * bz + az + ax * by - ay * bx;
  const double n = sqrt(qx * qx + qy * qy + qz * qz + w * w);
  out[3] = qx / n;
  out[4] = qy / n;
  out[5] = qz / n;
  out[6] = w / n;
}

// Same update with one reciprocal instead of four divisions (<= 1 ulp from pose_plus): used
// by the LM controller, where the serial division latency is what the GPU pays for.
CLC_HD void pose_plus_rcp(const double* x, const double* delta, double* out) {
  out[0] = x[0] + delta[0];
  out[1] = x[1] + delta[1];
  out[2] = x[2] + delta[2];
  const double bx = delta[3] * 0.5, by = delta[4] * 0.5, bz = delta[5] * 0.5;
  const double ax = x[3], ay = x[4], az = x[5], aw = x[6];
  const double w = aw - ax * bx - ay * by - az * bz;
  const double qx = aw * bx + ax + ay * bz - az * by;
  const double qy = aw * by + ay + az * bx - ax * bz;
  const double qz = aw * bz + az + ax * by - ay * bx;
  const double inv = rsqrt_pos(qx * qx + qy * qy + qz * qz + w * w);
  out[3] = qx * inv;
  out[4] = qy * inv;
  out[5] = qz * inv;
  out[6] = w * inv;
}

template <int N>
CLC_HD double norm_n(const double* x) {
  double s = 0.0;
  CLC_ROLLED for (int i = 0; i < N; ++i) s += x[i] * x[i];
  return sqrt_pos(s);
}
CLC_HD double norm7(const double* x) { return norm_n<7>(x); }

// index of (a,b), a<=b, in the packed upper triangle of an N x N matrix (row-major: 00 01 .. 11 ..)
template <int N>
CLC_HD int tri(int a, int b) { return a * N - (a * (a - 1)) / 2 + (b - a); }
CLC_HD int tri6(int a, int b) { return tri<6>(a, b); }

// Solve A y = b for symmetric positive definite N x N A (row-major). false if a pivot <= 0.
// One reciprocal square root per column (the serial controller is division-latency bound on
// the GPU); L holds the factor with the INVERSE diagonal on its diagonal.  L (N*N), z (N) scratch.
template <int N>
CLC_HD bool chol_solve(const double* A, const double* b, double* y, double* L, double* z) {
  CLC_ROLLED for (int j = 0; j < N; ++j) {
    double d = A[N * j + j];
    CLC_ROLLED for (int k = 0; k < j; ++k) d -= L[N * j + k] * L[N * j + k];
    if (!(d > 0.0)) return false;
    const double inv = rsqrt_pos(d);
    L[N * j + j] = inv;
    CLC_ROLLED for (int i = j + 1; i < N; ++i) {
      double s = A[N * i + j];
      CLC_ROLLED for (int k = 0; k < j; ++k) s -= L[N * i + k] * L[N * j + k];
      L[N * i + j] = s * inv;
    }
  }
  CLC_ROLLED for (int i = 0; i < N; ++i) {
    double s = b[i];
    CLC_ROLLED for (int k = 0; k < i; ++k) s -= L[N * i + k] * z[k];
    z[i] = s * L[N * i + i];
  }
  CLC_ROLLED for (int i = N - 1; i >= 0; --i) {
    double s = z[i];
    CLC_ROLLED for (int k = i + 1; k < N; ++k) s -= L[N * k + i] * y[k];
    y[i] = s * L[N * i + i];
  }
  return true;
}

}  // namespace clc
