// clc_host.hpp — host-side (CPU) pieces of the path that are O(poses) or O(1), not
// O(observations): problem assembly and the tiny dense factorizations that bracket the
// device reductions.  Everything O(N) runs in clc_kernels.hpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "../../include/clc.h"
#include "clc_math.hpp"

namespace clc {
namespace host {

inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// pi_from_ppp, src/utilities.cpp:267-272: [(x1-x3) x (x2-x3); -x3.(x1 x x2)], un-normalised.
inline void pi_from_ppp(const double* x1, const double* x2, const double* x3, double* pi) {
  const double a[3] = {x1[0] - x3[0], x1[1] - x3[1], x1[2] - x3[2]};
  const double b[3] = {x2[0] - x3[0], x2[1] - x3[1], x2[2] - x3[2]};
  double c12[3];
  cross3(a, b, pi);
  cross3(x1, x2, c12);
  pi[3] = -((x3[0] * c12[0] + x3[1] * c12[1]) + x3[2] * c12[2]);
}

// The planes of one board pose as the assembly loop forms them (src/LaseCamCalCeres.cpp:227-231, :258-276): the tag plane (z_tag = 0) in
// the camera frame, (Tctag^-1)^T (0,0,1,0) = [R_ca e3 ; -(R_ca e3).t_ca], and — for the board-edge terms — the two planes through the
// camera centre and a board edge (pi_from_ppp).  One definition for the host flatten below, the device flatten's host twin
// (clc_flatten_observations) and the host-side scan plan of small problems (abi_layouts.hip).
struct PosePlanes {
  double n[3], d;      // tag plane
  double pi1[4], pi2[4];  // edge planes (valid when asked for)
};
inline void pose_planes(const double* tag_q_wxyz, const double* tag_t, int i, bool edges, PosePlanes& out) {
  const double q[4] = {tag_q_wxyz[4 * i + 1], tag_q_wxyz[4 * i + 2], tag_q_wxyz[4 * i + 3], tag_q_wxyz[4 * i]};
  double R[9];
  quat_to_rot(q, R);
  const double* t = tag_t + 3 * i;
  out.n[0] = R[2]; out.n[1] = R[5]; out.n[2] = R[8];
  out.d = -((out.n[0] * t[0] + out.n[1] * t[1]) + out.n[2] * t[2]);
  if (edges) {  // :258-276
    const double orig = 0.0265 + 0.0165;                              // :262
    const double pm[3][3] = {{0.0 - orig, 0.0 - orig, 0.0},            // :263-268
                             {0.5 - orig, 0.0 - orig, 0.0},
                             {0.0 - orig, 0.5 - orig, 0.0}};
    double pc[3][3];
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a)  // :270-272
        pc[k][a] = ((R[3 * a] * pm[k][0] + R[3 * a + 1] * pm[k][1]) + R[3 * a + 2] * pm[k][2]) + t[a];
    const double zero[3] = {0.0, 0.0, 0.0};
    pi_from_ppp(pc[0], pc[1], zero, out.pi1);  // :275
    pi_from_ppp(pc[0], pc[2], zero, out.pi2);  // :276
  }
}

// Residual-block construction of CamLaserCalibration, src/LaseCamCalCeres.cpp:222-295.
// Returns CLC_OK / CLC_ERR_EMPTY_SCAN; *n_out = number of records.
inline int flatten(int n_poses, const double* tag_q, const double* tag_t, const int64_t* pts_off,
                   const double* pts, const int64_t* ptl_off, const double* ptl, bool linefit,
                   bool boundary, clc_observation* rec, int64_t* n_out) {
  int64_t N = 0;
  for (int i = 0; i < n_poses; ++i) {
    PosePlanes pp;
    pose_planes(tag_q, tag_t, i, boundary && linefit, pp);
    const double* n = pp.n;
    const double d = pp.d;
    const int64_t* off = linefit ? ptl_off : pts_off;  // :233-237
    const double* P = linefit ? ptl : pts;
    const int64_t cnt = off[i + 1] - off[i];
    const double scale = 1.0 / std::sqrt((double)cnt);  // :239-240
    for (int64_t j = off[i]; j < off[i + 1]; ++j, ++N) {
      if (!rec) continue;
      clc_observation& o = rec[N];
      o.n[0] = n[0]; o.n[1] = n[1]; o.n[2] = n[2]; o.d = d;
      o.p[0] = P[3 * j]; o.p[1] = P[3 * j + 1]; o.p[2] = P[3 * j + 2];
      o.scale = scale;
    }
    if (boundary && linefit) {  // :258-294
      if (pts_off[i + 1] - pts_off[i] <= 0) return CLC_ERR_EMPTY_SCAN;  // .at(0) throws, :278
      const double* pi1 = pp.pi1;
      const double* pi2 = pp.pi2;
      if (rec) {
        const double* front = pts + 3 * pts_off[i];            // obi.points.at(0), :278
        const double* back = pts + 3 * (pts_off[i + 1] - 1);   // obi.points.at(size-1), :279
        clc_observation& a = rec[N];
        a.n[0] = pi1[0]; a.n[1] = pi1[1]; a.n[2] = pi1[2]; a.d = pi1[3];
        a.p[0] = front[0]; a.p[1] = front[1]; a.p[2] = front[2]; a.scale = scale;
        clc_observation& b = rec[N + 1];
        b.n[0] = pi2[0]; b.n[1] = pi2[1]; b.n[2] = pi2[2]; b.d = pi2[3];
        b.p[0] = back[0]; b.p[1] = back[1]; b.p[2] = back[2]; b.scale = scale;
      }
      N += 2;
    }
  }
  *n_out = N;
  return CLC_OK;
}

// Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9), eigenvalues
// descending, eigenvectors in the columns of V (row-major).  For the symmetric PSD
// matrices of this path these are the singular values / vectors the reference prints from
// Eigen::JacobiSVD (src/LaseCamCalCeres.cpp:162-171, :365-379).
inline void jacobi_eig_sym(const double* Ain, int n, double* w, double* V) {
  double A[81];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
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
        for (int k = 0; k < n; ++k) {  // A <- A G
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {  // A <- G^T A
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {  // V <- V G
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

// AtA.ldlt().solve(rhs) at src/LaseCamCalCeres.cpp:181 (n <= 9).  Eigen's LDLT is the "robust Cholesky with pivoting"
// of its documentation: P A P^T = L D L^T, at every step the largest remaining diagonal entry is brought to the pivot
// position by a symmetric permutation; the solve applies P, L^-1, the PSEUDO-inverse of D (an exactly zero pivot
// contributes zero instead of a division by zero), L^-T, P^T.  It therefore returns a finite vector for a semi-definite
// (unobservable) normal matrix too — which is what the reference goes on to use after printing its notice (:173-181).
// A plain Cholesky, as round 1 used here, fails on such input.
// Pivot rule: like Eigen's unblocked kernel (LDLT.h, ldlt_inplace<Lower>::unblocked) the factorisation is LEFT-looking —
// step k updates only entry (k, k) and the column below it (A21 -= A20 (D A10^T)), never the trailing block — so "the
// largest remaining diagonal entry" is searched on diagonal entries that have not been updated yet, i.e. on the
// original diagonal of the remaining rows.  That is what this loop does too (a right-looking factorisation would
// pivot on the Schur complement's diagonal and pick different pivots on semi-definite input).  Restated from the
// published source, not run against an Eigen build (none in this image): on unobservable input the returned Tlc is
// pinned against this restatement only.
inline void ldlt_solve_n(const double* Ain, const double* b, double* x, int n) {
  double A[81], D[9], y[9];
  int perm[9];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double big = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[i * n + i]) > big) { big = std::fabs(A[i * n + i]); piv = i; }
    if (piv != k) {  // symmetric row/column swap (the full matrix is kept, both triangles)
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[piv * n + j]);
      for (int i = 0; i < n; ++i) std::swap(A[i * n + k], A[i * n + piv]);
      std::swap(perm[k], perm[piv]);
    }
    // column k of L and D[k] (rows/columns < k already hold L below the diagonal)
    double d = A[k * n + k];
    for (int j = 0; j < k; ++j) d -= A[k * n + j] * A[k * n + j] * D[j];
    D[k] = d;
    for (int i = k + 1; i < n; ++i) {
      double v = A[i * n + k];
      for (int j = 0; j < k; ++j) v -= A[i * n + j] * A[k * n + j] * D[j];
      A[i * n + k] = (d != 0.0) ? v / d : 0.0;
    }
  }
  for (int i = 0; i < n; ++i) y[i] = b[perm[i]];            // P b
  for (int i = 0; i < n; ++i)                                 // L^-1
    for (int j = 0; j < i; ++j) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) y[i] = (std::fabs(D[i]) > 2.2250738585072014e-308) ? y[i] / D[i] : 0.0;  // D^+
  for (int i = n - 1; i >= 0; --i)                            // L^-T
    for (int j = i + 1; j < n; ++j) y[i] -= A[j * n + i] * y[j];
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];              // P^T
}

// Nearest orthogonal matrix U V^T of a 3x3 M from its singular value decomposition (src/LaseCamCalCeres.cpp:195-196:
// JacobiSVD of Rlc, U * V^T; like the reference, no determinant check).  V and the singular values come from the
// symmetric eigen-decomposition of M^T M, U = M V / sigma column by column; a column whose singular value vanishes
// (rank-deficient M: h1 parallel to h2, or zero) is completed to an orthonormal basis — an SVD always has a full
// orthogonal U, so U V^T stays defined where M (M^T M)^(-1/2), the closed formula round 1 used, divides by zero.
inline void nearest_orthogonal3(const double* M, double* Q) {
  double MtM[9], w[3], V[9], U[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += M[3 * k + i] * M[3 * k + j];
      MtM[3 * i + j] = s;
    }
  jacobi_eig_sym(MtM, 3, w, V);  // descending
  const double smax = std::sqrt(std::max(w[0], 0.0));
  int have[3] = {0, 0, 0};
  for (int c = 0; c < 3; ++c) {
    double u[3] = {0, 0, 0};
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) u[r] += M[3 * r + k] * V[3 * k + c];
    const double nrm = std::sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
    if (nrm > 1e-14 * smax && nrm > 0.0) {
      for (int r = 0; r < 3; ++r) U[3 * r + c] = u[r] / nrm;
      have[c] = 1;
    }
  }
  // complete the missing columns (singular values are sorted: the missing ones come last)
  if (!have[0]) { U[0] = 1; U[3] = 0; U[6] = 0; have[0] = 1; }
  if (!have[1]) {  // any unit vector orthogonal to column 0
    const double a[3] = {U[0], U[3], U[6]};
    int m = std::fabs(a[0]) <= std::fabs(a[1]) ? (std::fabs(a[0]) <= std::fabs(a[2]) ? 0 : 2) : (std::fabs(a[1]) <= std::fabs(a[2]) ? 1 : 2);
    double e[3] = {0, 0, 0}, c1[3];
    e[m] = 1.0;
    cross3(a, e, c1);
    const double nrm = std::sqrt((c1[0] * c1[0] + c1[1] * c1[1]) + c1[2] * c1[2]);
    for (int r = 0; r < 3; ++r) U[3 * r + 1] = c1[r] / nrm;
    have[1] = 1;
  }
  if (!have[2]) {
    const double a[3] = {U[0], U[3], U[6]}, b2[3] = {U[1], U[4], U[7]};
    double c2[3];
    cross3(a, b2, c2);
    for (int r = 0; r < 3; ++r) U[3 * r + 2] = c2[r];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += U[3 * i + k] * V[3 * j + k];
      Q[3 * i + j] = s;
    }
}

// Back end of CamLaserCalClosedSolution (src/LaseCamCalCeres.cpp:162-200) given the reduced
// 9x9 normal equation.  Like the reference it always produces a Tlc — for an unobservable system (flagged through
// *unobservable, :164-178) whatever the pivoted LDLT and the SVD yield; CLC_ERR_NONFINITE only if that is not finite.
inline int closed_form_from_normal(const double* AtA, const double* Atb, double* Tlc,
                                   int* unobservable, double* sv9) {
  double w[9], V[81];
  jacobi_eig_sym(AtA, 9, w, V);  // :162
  int un = 0;
  for (int i = 0; i < 9; ++i) {
    if (w[i] < 1e-10) un = 1;  // :167
    if (sv9) sv9[i] = w[i];
  }
  *unobservable = un;
  double H[9];
  ldlt_solve_n(AtA, Atb, H, 9);  // :181
  const double *h1 = H, *h2 = H + 3, *h3 = H + 6;
  double h12[3];
  cross3(h1, h2, h12);
  // Rcl = [h1 h2 h1xh2] (:187-190), Rlc = Rcl^T (:191)
  const double Rlc[9] = {h1[0], h1[1], h1[2], h2[0], h2[1], h2[2], h12[0], h12[1], h12[2]};
  double tlc[3], Q[9];
  for (int i = 0; i < 3; ++i)  // tlc = -Rlc h3, before the orthogonal projection (:192)
    tlc[i] = -((Rlc[3 * i] * h3[0] + Rlc[3 * i + 1] * h3[1]) + Rlc[3 * i + 2] * h3[2]);
  nearest_orthogonal3(Rlc, Q);  // :195-196
  for (int i = 0; i < 16; ++i) Tlc[i] = 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Tlc[4 * i + j] = Q[3 * i + j];
    Tlc[4 * i + 3] = tlc[i];
  }
  Tlc[15] = 1.0;
  for (int i = 0; i < 16; ++i)
    if (!std::isfinite(Tlc[i])) return CLC_ERR_NONFINITE;
  return CLC_OK;
}

}  // namespace host
}  // namespace clc
