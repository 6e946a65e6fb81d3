// clc_rows.hpp — per-scan moment form of the point-to-plane evaluation ("row layout", the default streaming path).
//
// Within one scan every residual block carries the same plane (n, d) and the same scale s
// (src/LaseCamCalCeres.cpp:231,239-240,245: one plane and one 1/sqrt(count) per Oberserve, one PointInPlaneFactor per
// point), and every scan point lies in the lidar plane z = 0 (src/utilities.cpp:203; main/calibr_simulation.cpp:79-96).
// With m = R^T n and c0 = n.t + d (constant over the scan) the quantities of PointInPlaneFactor::Evaluate (:43-66) are
//     r0 = n.(R p + t) + d = mx x + my y + c0                        (residual before the scale, :47-48)
//     u  = [n, p x m] = [nx, ny, nz, y mz, -x mz, x my - y mx]        (Jacobian row J = s u, :56-57)
// i.e. u is LINEAR in (1, x, y).  So instead of 28 rank-1 accumulations per point (21 + 6 + 1), a lane accumulates the
// weighted moments of its points of the scan,
//     S0 = sum w,  Sx = sum w x,  Sy = sum w y,  Sxx, Sxy, Syy,  T0 = sum w r0,  Tx = sum w r0 x,  Ty = sum w r0 y,
// with the Cauchy weight w = rho'(r^2) = 1 / (1 + r0^2 / lf^2) (the scale cancels inside the loss argument, see
// clc_kernels.hpp), and expands them ONCE per scan segment into the 6x6 normal equation:
//     H += s^2 A M A^T,   g += s^2 A [T0, Tx, Ty]^T,   u = A [1, x, y]^T.
// The robust cost 1/2 sum rho = lf^2/2 sum s^2 log(1 + r0^2/lf^2) needs no logarithm per point either: the lane keeps
// the running PRODUCT of (1 + r0^2/lf^2) as (mantissa, exponent) — one multiply and a frexp per point — and takes one
// logarithm per segment.
// Per point: ~21 FP64 instructions instead of ~100; per streamed byte: 16 B (x, y) instead of 28 (x, y, z, group id).
//
// Written for device code and for the host-side unit shim (tests/shim/rows_shim.cpp, g++): same source, the device-only
// intrinsics have plain-C fall-backs.
#pragma once
#include "clc_math.hpp"

namespace clc {

constexpr int ROW = 64;          // points per row = lanes per wavefront: one point per lane per row
constexpr int ROW_DOUBLES = 128; // (x, y) interleaved: lane l reads doubles 2l, 2l+1 with one 16-byte load
constexpr int ROW_DOUBLES_Z = 192;  // rows that carry z: the 64 z of the row follow its 64 (x, y) pairs
constexpr int NACC_ROWS = 28;    // same accumulator layout as the per-point path: H(21) g(6) cost(1)

// One descriptor per row (64 B, read with scalar loads — it is wave-uniform): the plane and scale of the row's scan
// (:227-231, :239-240), the number of valid points in the row (the last row of a scan is padded) and whether the row
// starts a new scan.
struct RowDesc {
  double nx, ny, nz, d, s;
  int32_t count;   // 1..64 valid points (lanes 0..count-1)
  int32_t first;   // 1: first row of its scan (moments restart here)
  double pad_[2];
};
static_assert(sizeof(RowDesc) == 64, "RowDesc is 64 bytes");

// x = m * 2^e with m in [0.5, 1) for finite x > 0.
CLC_HD double frexp_pos(double x, int& e) {
#if defined(__HIP_DEVICE_COMPILE__)
  e = __builtin_amdgcn_frexp_exp(x);
  return __builtin_amdgcn_frexp_mant(x);
#else
  return frexp(x, &e);
#endif
}

// 1/x to ~1 ulp for finite x >= 1 (device: v_rcp_f64 seed + two Newton steps).
CLC_HD double rcp_ge1(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
#else
  return 1.0 / x;
#endif
}

// The Cauchy weight 1/x, x = 1 + r0^2/lf^2 >= 1: v_rcp_f64 seed (relative error <= 2^-25, scripts/probes/rcp_probe.hip) + ONE
// Newton step: <= 10 ulp (1e-15 relative), two FP64 instructions less per point than rcp_ge1.  The weights only enter the
// normal equation (H, g: reduction tolerance 1e-11); the cost comes from the exact running product, not from them.
CLC_HD double rcp_ge1_weight(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
#else
  return 1.0 / x;
#endif
}

// log(m * 2^e) for m in [0.5, 1): fdlibm-style, m -> [sqrt(1/2), sqrt(2)), f = m - 1, s = f/(2+f),
// log(m) = f - s (f - R(s^2)), 7-term minimax R.  < 1 ulp.
CLC_HD double log_mant_exp(double m, int e) {
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * rcp_ge1(2.0 + f);  // 2 + f in [1.7, 2.42)
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// What a lane knows about the scan it is in: m = R^T n restricted to what the point loop needs, and the plane itself
// for the expansion.
struct RowPlane {
  double nx, ny, nz, s2;  // plane normal, scale^2
  double mx, my, mz, c0;  // m = R^T n, c0 = n.t + d
};

// R row-major, t: the pose (wave-uniform).
CLC_HD void rows_plane_setup(const double* R, const double* t, double nx, double ny, double nz, double d, double s,
                             RowPlane& q) {
  q.nx = nx; q.ny = ny; q.nz = nz;
  q.s2 = s * s;
  q.mx = fma(R[6], nz, fma(R[3], ny, R[0] * nx));
  q.my = fma(R[7], nz, fma(R[4], ny, R[1] * nx));
  q.mz = fma(R[8], nz, fma(R[5], ny, R[2] * nx));
  q.c0 = fma(t[2], nz, fma(t[1], ny, fma(t[0], nx, d)));
}

struct RowMoments {
  double S0, Sx, Sy, Sxx, Sxy, Syy, T0, Tx, Ty;
  double prod;  // WITH_LOSS: mantissa of prod (1 + r0^2/lf^2) in [0.5, 1);  without loss: sum r0^2
  int expo;     // WITH_LOSS: its binary exponent
};

template <bool WITH_LOSS>
CLC_HD void rows_moments_reset(RowMoments& M) {
  M.S0 = M.Sx = M.Sy = M.Sxx = M.Sxy = M.Syy = M.T0 = M.Tx = M.Ty = 0.0;
  M.prod = WITH_LOSS ? 0.5 : 0.0;  // 0.5 * 2^1 = 1
  M.expo = WITH_LOSS ? 1 : 0;
}

// One scan point (x, y, 0).
// renorm (WITH_LOSS): bring the running product back to (mantissa in [0.5, 1), exponent) after this point.  Callers do so
// on every SECOND point of a lane: a mantissa below 1 times two factors cannot overflow while each factor 1 + r0^2/lf^2 is
// below 2^511 (|r0| < 1e76 lf) — beyond that the cost is +inf and the solve fails like any non-finite evaluation — and
// rows_flush normalises whatever is left.  One multiply + 1 (instead of 3) instructions per point for the robust cost.
// The weight is rho'(s) = 1 / (1 + r0^2/lf^2) without Ceres' max(DBL_MIN, .) clamp, which only acts beyond 4.5e307.
template <bool WITH_LOSS>
CLC_HD void rows_point(const RowPlane& q, const double inv_lf2, const double x, const double y, RowMoments& M,
                       const bool renorm = true) {
  const double r0 = fma(q.my, y, fma(q.mx, x, q.c0));
  double w = 1.0;
  if (WITH_LOSS) {
    const double sum = fma(r0 * r0, inv_lf2, 1.0);
    w = rcp_ge1_weight(sum);
    const double pr = M.prod * sum;
    if (renorm) {
      int e;
      M.prod = frexp_pos(pr, e);
      M.expo += e;
    } else {
      M.prod = pr;
    }
  } else {
    M.prod = fma(r0, r0, M.prod);
  }
  const double wx = w * x, wy = w * y, wr = w * r0;
  M.S0 += w;
  M.Sx += wx;
  M.Sy += wy;
  M.Sxx = fma(wx, x, M.Sxx);
  M.Sxy = fma(wx, y, M.Sxy);
  M.Syy = fma(wy, y, M.Syy);
  M.T0 += wr;
  M.Tx = fma(wr, x, M.Tx);
  M.Ty = fma(wr, y, M.Ty);
}

// The same for a slot of the on-chip lane layouts (clc_coop.hpp, clc_resident.hpp) that may be padding: valid = false (a padded slot,
// stored as (0, 0)) must leave every moment and the cost product alone — r0 is forced to 0 (cost factor exactly 1) and the weight
// to 0.  (Forcing r0 alone and subtracting the count of padded slots from S0 afterwards was tried first: a padded slot then weighs
// 1, and from a start 1e4 m off, where the Cauchy weights of the real points are ~1e-11, S0 lost seven digits to the cancellation.
// Without the loss every weight IS 1, the subtraction is exact, and a variable weight would only cost the kernel its constant
// folding: there the caller calls rows_pad_correction.)
template <bool WITH_LOSS>
CLC_HD void rows_point_masked(const RowPlane& q, const bool valid, const double inv_lf2, const double x, const double y, RowMoments& M,
                              const bool renorm = true) {
  const double r1 = fma(q.my, y, fma(q.mx, x, q.c0));
  const double r0 = valid ? r1 : 0.0;
  double w = 1.0;
  if (WITH_LOSS) {
    const double sum = fma(r0 * r0, inv_lf2, 1.0);
    w = rcp_ge1_weight(sum);
    const double pr = M.prod * sum;
    if (renorm) {
      int e;
      M.prod = frexp_pos(pr, e);
      M.expo += e;
    } else {
      M.prod = pr;
    }
  } else {
    M.prod = fma(r0, r0, M.prod);
  }
  if (WITH_LOSS) w = valid ? w : 0.0;  // (without loss the weight is the constant 1: the caller takes the count of padded slots out of S0 — exact)
  const double wx = w * x, wy = w * y, wr = w * r0;
  M.S0 += w;
  M.Sx += wx;
  M.Sy += wy;
  M.Sxx = fma(wx, x, M.Sxx);
  M.Sxy = fma(wx, y, M.Sxy);
  M.Syy = fma(wy, y, M.Syy);
  M.T0 += wr;
  M.Tx = fma(wr, x, M.Tx);
  M.Ty = fma(wr, y, M.Ty);
}

template <bool WITH_LOSS>
CLC_HD void rows_pad_correction(RowMoments& M, const double n_padded_slots) {
  if (!WITH_LOSS) M.S0 -= n_padded_slots;
}

// Expand a lane's moments of one scan segment into its 28 accumulators
// (acc[0..20] H upper triangle row-major, acc[21..26] g, acc[27] sum s^2 log(sum) or sum s^2 r0^2; finalize_cost()
// applies lf^2/2 resp. 1/2 afterwards, as for the per-point path).
template <bool WITH_LOSS>
CLC_HD void rows_flush(const RowPlane& q, const RowMoments& M, double* acc) {
  const double s2 = q.s2;
  const double S0 = s2 * M.S0, Sx = s2 * M.Sx, Sy = s2 * M.Sy;
  const double Sxx = s2 * M.Sxx, Sxy = s2 * M.Sxy, Syy = s2 * M.Syy;
  const double T0 = s2 * M.T0, Tx = s2 * M.Tx, Ty = s2 * M.Ty;
  const double nx = q.nx, ny = q.ny, nz = q.nz, mx = q.mx, my = q.my, mz = q.mz;
  // translation block: n n^T S0
  const double ax = nx * S0, ay = ny * S0, az = nz * S0;
  acc[0] = fma(ax, nx, acc[0]);
  acc[1] = fma(ax, ny, acc[1]);
  acc[2] = fma(ax, nz, acc[2]);
  acc[6] = fma(ay, ny, acc[6]);
  acc[7] = fma(ay, nz, acc[7]);
  acc[11] = fma(az, nz, acc[11]);
  // cross block: n (sum w u_theta)^T,  sum w u_theta = (mz Sy, -mz Sx, my Sx - mx Sy)
  const double v3 = mz * Sy, v4 = -(mz * Sx), v5 = fma(my, Sx, -(mx * Sy));
  acc[3] = fma(nx, v3, acc[3]);
  acc[4] = fma(nx, v4, acc[4]);
  acc[5] = fma(nx, v5, acc[5]);
  acc[8] = fma(ny, v3, acc[8]);
  acc[9] = fma(ny, v4, acc[9]);
  acc[10] = fma(ny, v5, acc[10]);
  acc[12] = fma(nz, v3, acc[12]);
  acc[13] = fma(nz, v4, acc[13]);
  acc[14] = fma(nz, v5, acc[14]);
  // rotation block: sum w u_theta u_theta^T
  const double mz2 = mz * mz;
  const double A = fma(my, Sxx, -(mx * Sxy));  // sum w x (x my - y mx)
  const double B = fma(my, Sxy, -(mx * Syy));  // sum w y (x my - y mx)
  acc[15] = fma(mz2, Syy, acc[15]);
  acc[16] = fma(-mz2, Sxy, acc[16]);
  acc[17] = fma(mz, B, acc[17]);
  acc[18] = fma(mz2, Sxx, acc[18]);
  acc[19] = fma(-mz, A, acc[19]);
  acc[20] = fma(my, A, fma(-mx, B, acc[20]));
  // gradient: sum w r0 u
  acc[21] = fma(nx, T0, acc[21]);
  acc[22] = fma(ny, T0, acc[22]);
  acc[23] = fma(nz, T0, acc[23]);
  acc[24] = fma(mz, Ty, acc[24]);
  acc[25] = fma(-mz, Tx, acc[25]);
  acc[26] = fma(my, Tx, fma(-mx, Ty, acc[26]));
  // cost
  if (WITH_LOSS) {
    int e;
    const double m = frexp_pos(M.prod, e);  // (the last point of the segment may have left the product un-normalised)
    acc[27] = fma(s2, log_mant_exp(m, M.expo + e), acc[27]);
  } else
    acc[27] = fma(s2, M.prod, acc[27]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Scan points OFF the lidar plane (p.z != 0): Oberserve::points is std::vector<Eigen::Vector3d> (include/LaseCamCalCeres.h:22),
// and although the reference's own producers always write z = 0 (src/utilities.cpp:198-215), a caller of the Vector3d
// interface need not.  Same idea, one dimension more: r0 = m.p + c0 and u = [n, p x m] are linear in (1, x, y, z), so a
// lane accumulates the 14 weighted moments S0, S_a, S_ab (a <= b in {x, y, z}), T0, T_a and expands them once per scan
// segment: 28.5 FP64 instructions and 24 B per point (x, y as for z = 0 plus one more 8-byte row of z), against 21.5 / 16.
// ---------------------------------------------------------------------------------------------------------------------
struct RowMoments3 {
  double S0, Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz, Szz, T0, Tx, Ty, Tz;
  double prod;  // as RowMoments
  int expo;
};

template <bool WITH_LOSS>
CLC_HD void rows3_moments_reset(RowMoments3& M) {
  M.S0 = M.Sx = M.Sy = M.Sz = M.Sxx = M.Sxy = M.Sxz = M.Syy = M.Syz = M.Szz = M.T0 = M.Tx = M.Ty = M.Tz = 0.0;
  M.prod = WITH_LOSS ? 0.5 : 0.0;
  M.expo = WITH_LOSS ? 1 : 0;
}

// One scan point (x, y, z); renorm as for rows_point.
template <bool WITH_LOSS>
CLC_HD void rows3_point(const RowPlane& q, const double inv_lf2, const double x, const double y, const double z, RowMoments3& M,
                        const bool renorm = true) {
  const double r0 = fma(q.mz, z, fma(q.my, y, fma(q.mx, x, q.c0)));
  double w = 1.0;
  if (WITH_LOSS) {
    const double sum = fma(r0 * r0, inv_lf2, 1.0);
    w = rcp_ge1_weight(sum);
    const double pr = M.prod * sum;
    if (renorm) {
      int e;
      M.prod = frexp_pos(pr, e);
      M.expo += e;
    } else {
      M.prod = pr;
    }
  } else {
    M.prod = fma(r0, r0, M.prod);
  }
  const double wx = w * x, wy = w * y, wz = w * z, wr = w * r0;
  M.S0 += w;
  M.Sx += wx;
  M.Sy += wy;
  M.Sz += wz;
  M.Sxx = fma(wx, x, M.Sxx);
  M.Sxy = fma(wx, y, M.Sxy);
  M.Sxz = fma(wx, z, M.Sxz);
  M.Syy = fma(wy, y, M.Syy);
  M.Syz = fma(wy, z, M.Syz);
  M.Szz = fma(wz, z, M.Szz);
  M.T0 += wr;
  M.Tx = fma(wr, x, M.Tx);
  M.Ty = fma(wr, y, M.Ty);
  M.Tz = fma(wr, z, M.Tz);
}

// Expansion into the 28 accumulators (layout as rows_flush).  With u_theta = p x m = (y mz - z my, z mx - x mz, x my - y mx):
//   sum w u_theta            = (mz Sy - my Sz,  mx Sz - mz Sx,  my Sx - mx Sy)
//   sum w u_theta u_theta^T  = quadratic forms of the second moments (below), sum w r0 u_theta likewise from T_a.
template <bool WITH_LOSS>
CLC_HD void rows3_flush(const RowPlane& q, const RowMoments3& M, double* acc) {
  const double s2 = q.s2;
  const double S0 = s2 * M.S0, Sx = s2 * M.Sx, Sy = s2 * M.Sy, Sz = s2 * M.Sz;
  const double Sxx = s2 * M.Sxx, Sxy = s2 * M.Sxy, Sxz = s2 * M.Sxz, Syy = s2 * M.Syy, Syz = s2 * M.Syz, Szz = s2 * M.Szz;
  const double T0 = s2 * M.T0, Tx = s2 * M.Tx, Ty = s2 * M.Ty, Tz = s2 * M.Tz;
  const double nx = q.nx, ny = q.ny, nz = q.nz, mx = q.mx, my = q.my, mz = q.mz;
  // translation block: n n^T S0
  const double ax = nx * S0, ay = ny * S0, az = nz * S0;
  acc[0] = fma(ax, nx, acc[0]);
  acc[1] = fma(ax, ny, acc[1]);
  acc[2] = fma(ax, nz, acc[2]);
  acc[6] = fma(ay, ny, acc[6]);
  acc[7] = fma(ay, nz, acc[7]);
  acc[11] = fma(az, nz, acc[11]);
  // cross block: n (sum w u_theta)^T
  const double v3 = fma(mz, Sy, -(my * Sz)), v4 = fma(mx, Sz, -(mz * Sx)), v5 = fma(my, Sx, -(mx * Sy));
  acc[3] = fma(nx, v3, acc[3]);
  acc[4] = fma(nx, v4, acc[4]);
  acc[5] = fma(nx, v5, acc[5]);
  acc[8] = fma(ny, v3, acc[8]);
  acc[9] = fma(ny, v4, acc[9]);
  acc[10] = fma(ny, v5, acc[10]);
  acc[12] = fma(nz, v3, acc[12]);
  acc[13] = fma(nz, v4, acc[13]);
  acc[14] = fma(nz, v5, acc[14]);
  // rotation block: sum w u_theta u_theta^T via the vectors  P_a = sum w a (p x m)  for a in {x, y, z}:
  //   P_x = (mz Sxy - my Sxz, mx Sxz - mz Sxx, my Sxx - mx Sxy), P_y, P_z alike;  then
  //   sum w (p x m)(p x m)^T = row a of the cross-product matrix [.]x m applied to P:  (33) = mz P_y[0] - my P_z[0], ...
  const double Px0 = fma(mz, Sxy, -(my * Sxz)), Px1 = fma(mx, Sxz, -(mz * Sxx)), Px2 = fma(my, Sxx, -(mx * Sxy));
  const double Py0 = fma(mz, Syy, -(my * Syz)), Py1 = fma(mx, Syz, -(mz * Sxy)), Py2 = fma(my, Sxy, -(mx * Syy));
  const double Pz0 = fma(mz, Syz, -(my * Szz)), Pz1 = fma(mx, Szz, -(mz * Sxz)), Pz2 = fma(my, Sxz, -(mx * Syz));
  acc[15] = fma(mz, Py0, fma(-my, Pz0, acc[15]));  // (a a)
  acc[16] = fma(mz, Py1, fma(-my, Pz1, acc[16]));  // (a b)
  acc[17] = fma(mz, Py2, fma(-my, Pz2, acc[17]));  // (a c)
  acc[18] = fma(mx, Pz1, fma(-mz, Px1, acc[18]));  // (b b)
  acc[19] = fma(mx, Pz2, fma(-mz, Px2, acc[19]));  // (b c)
  acc[20] = fma(my, Px2, fma(-mx, Py2, acc[20]));  // (c c)
  // gradient: sum w r0 u
  acc[21] = fma(nx, T0, acc[21]);
  acc[22] = fma(ny, T0, acc[22]);
  acc[23] = fma(nz, T0, acc[23]);
  acc[24] = fma(mz, Ty, fma(-my, Tz, acc[24]));
  acc[25] = fma(mx, Tz, fma(-mz, Tx, acc[25]));
  acc[26] = fma(my, Tx, fma(-mx, Ty, acc[26]));
  // cost
  if (WITH_LOSS) {
    int e;
    const double m = frexp_pos(M.prod, e);
    acc[27] = fma(s2, log_mant_exp(m, M.expo + e), acc[27]);
  } else
    acc[27] = fma(s2, M.prod, acc[27]);
}

// The same for a slot of the on-chip lane layout that may be padding (see rows_point_masked).
template <bool WITH_LOSS>
CLC_HD void rows3_point_masked(const RowPlane& q, const bool valid, const double inv_lf2, const double x, const double y, const double z,
                               RowMoments3& M, const bool renorm = true) {
  const double r1 = fma(q.mz, z, fma(q.my, y, fma(q.mx, x, q.c0)));
  const double r0 = valid ? r1 : 0.0;
  double w = 1.0;
  if (WITH_LOSS) {
    const double sum = fma(r0 * r0, inv_lf2, 1.0);
    w = rcp_ge1_weight(sum);
    const double pr = M.prod * sum;
    if (renorm) {
      int e;
      M.prod = frexp_pos(pr, e);
      M.expo += e;
    } else {
      M.prod = pr;
    }
  } else {
    M.prod = fma(r0, r0, M.prod);
  }
  if (WITH_LOSS) w = valid ? w : 0.0;
  const double wx = w * x, wy = w * y, wz = w * z, wr = w * r0;
  M.S0 += w;
  M.Sx += wx;
  M.Sy += wy;
  M.Sz += wz;
  M.Sxx = fma(wx, x, M.Sxx);
  M.Sxy = fma(wx, y, M.Sxy);
  M.Sxz = fma(wx, z, M.Sxz);
  M.Syy = fma(wy, y, M.Syy);
  M.Syz = fma(wy, z, M.Syz);
  M.Szz = fma(wz, z, M.Szz);
  M.T0 += wr;
  M.Tx = fma(wr, x, M.Tx);
  M.Ty = fma(wr, y, M.Ty);
  M.Tz = fma(wr, z, M.Tz);
}

// One spelling for both point forms (overloaded on the moment type): what a kernel templated on "the points carry z" calls.
template <bool L> CLC_HD void lane_moments_reset(RowMoments& M) { rows_moments_reset<L>(M); }
template <bool L> CLC_HD void lane_moments_reset(RowMoments3& M) { rows3_moments_reset<L>(M); }
template <bool L> CLC_HD void lane_point(const RowPlane& q, double inv_lf2, double x, double y, double, RowMoments& M, bool renorm) { rows_point<L>(q, inv_lf2, x, y, M, renorm); }
template <bool L> CLC_HD void lane_point(const RowPlane& q, double inv_lf2, double x, double y, double z, RowMoments3& M, bool renorm) { rows3_point<L>(q, inv_lf2, x, y, z, M, renorm); }
template <bool L> CLC_HD void lane_point_masked(const RowPlane& q, bool valid, double inv_lf2, double x, double y, double, RowMoments& M, bool renorm) {
  rows_point_masked<L>(q, valid, inv_lf2, x, y, M, renorm);
}
template <bool L> CLC_HD void lane_point_masked(const RowPlane& q, bool valid, double inv_lf2, double x, double y, double z, RowMoments3& M, bool renorm) {
  rows3_point_masked<L>(q, valid, inv_lf2, x, y, z, M, renorm);
}
template <bool L> CLC_HD void lane_pad_correction(RowMoments& M, double n_padded_slots) { rows_pad_correction<L>(M, n_padded_slots); }
template <bool L> CLC_HD void lane_pad_correction(RowMoments3& M, double n_padded_slots) { if (!L) M.S0 -= n_padded_slots; }
template <bool L> CLC_HD void lane_flush(const RowPlane& q, const RowMoments& M, double* acc) { rows_flush<L>(q, M, acc); }
template <bool L> CLC_HD void lane_flush(const RowPlane& q, const RowMoments3& M, double* acc) { rows3_flush<L>(q, M, acc); }

}  // namespace clc
