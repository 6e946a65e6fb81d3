// tests/shim/rows_shim.cpp — TEST ONLY.  Compiles the product's per-scan moment math
// (camlasercalibratool_amd/csrc/clc_rows.hpp: rows_plane_setup / rows_point / rows_flush, the code the row-layout
// kernels run per lane) for the host with g++ and drives it the way a wavefront does — 64 lane states, one point per
// lane per row, moments flushed where the scan changes — so the expansion formulas and the product/exponent form of
// the Cauchy cost can be checked against the oracle here, where there is no GPU.  The row building below mirrors
// build_rows_kernel (group = run of records with bit-identical (n, d, scale); every group padded to whole rows).
#include <cstring>
#include <vector>

#include "../../camlasercalibratool_amd/csrc/clc_rows.hpp"

extern "C" {

// records: n x 8 doubles {n(3), d, p(3), scale}.  out28: H(21) g(6) raw cost accumulator (before finalize).
// rows_per_wave: the emulated wave restarts (flushes, re-derives the plane) every this many rows, like a wave whose run
// begins in the middle of a scan.  Returns the number of rows, or -1 if some p.z != 0.
long shim_rows_eval(const double* rec, long n, const double* pose, int with_loss, double lf, long rows_per_wave,
                    double* out28) {
  using namespace clc;
  double R[9];
  quat_to_rot(pose + 3, R);
  const double* t = pose;
  const double inv_lf2 = 1.0 / (lf * lf);
  // rows
  std::vector<RowDesc> desc;
  std::vector<double> xy;
  long k = 0;
  while (k < n) {
    long e = k + 1;
    const double* a = rec + 8 * k;
    while (e < n && std::memcmp(rec + 8 * e, a, 4 * sizeof(double)) == 0 && std::memcmp(rec + 8 * e + 7, a + 7, sizeof(double)) == 0) ++e;
    for (long r0 = k; r0 < e; r0 += ROW) {
      RowDesc d;
      d.nx = a[0]; d.ny = a[1]; d.nz = a[2]; d.d = a[3]; d.s = a[7];
      d.count = (int32_t)((e - r0) < ROW ? (e - r0) : ROW);
      d.first = r0 == k;
      d.pad_[0] = d.pad_[1] = 0.0;
      desc.push_back(d);
      for (int l = 0; l < ROW; ++l) {
        const bool v = r0 + l < e;
        if (v && rec[8 * (r0 + l) + 6] != 0.0) return -1;
        xy.push_back(v ? rec[8 * (r0 + l) + 4] : 0.0);
        xy.push_back(v ? rec[8 * (r0 + l) + 5] : 0.0);
      }
    }
    k = e;
  }
  const long n_rows = (long)desc.size();
  std::vector<double> acc(64 * 28, 0.0);
  std::vector<RowMoments> M(64);
  RowPlane q;
  bool have = false;
  for (long r = 0; r < n_rows; ++r) {
    const RowDesc& d = desc[r];
    if (d.first || (rows_per_wave > 0 && r % rows_per_wave == 0)) {
      if (have)
        for (int l = 0; l < 64; ++l) { if (with_loss) rows_flush<true>(q, M[l], &acc[28 * l]); else rows_flush<false>(q, M[l], &acc[28 * l]); }
      rows_plane_setup(R, t, d.nx, d.ny, d.nz, d.d, d.s, q);
      for (int l = 0; l < 64; ++l) { if (with_loss) rows_moments_reset<true>(M[l]); else rows_moments_reset<false>(M[l]); }
      have = true;
    }
    for (int l = 0; l < d.count; ++l) {
      const double x = xy[(r * 64 + l) * 2], y = xy[(r * 64 + l) * 2 + 1];
      if (with_loss) rows_point<true>(q, inv_lf2, x, y, M[l]); else rows_point<false>(q, inv_lf2, x, y, M[l]);
    }
  }
  if (have)
    for (int l = 0; l < 64; ++l) { if (with_loss) rows_flush<true>(q, M[l], &acc[28 * l]); else rows_flush<false>(q, M[l], &acc[28 * l]); }
  for (int c = 0; c < 28; ++c) {
    double s = 0.0;
    for (int l = 0; l < 64; ++l) s += acc[28 * l + c];
    out28[c] = s;
  }
  return n_rows;
}

// The same drive for scan points off the lidar plane (p.z != 0): rows3_point / rows3_flush, with a third 8-byte row of z.
long shim_rows3_eval(const double* rec, long n, const double* pose, int with_loss, double lf, long rows_per_wave,
                     double* out28) {
  using namespace clc;
  double R[9];
  quat_to_rot(pose + 3, R);
  const double* t = pose;
  const double inv_lf2 = 1.0 / (lf * lf);
  std::vector<RowDesc> desc;
  std::vector<double> xyz;
  long k = 0;
  while (k < n) {
    long e = k + 1;
    const double* a = rec + 8 * k;
    while (e < n && std::memcmp(rec + 8 * e, a, 4 * sizeof(double)) == 0 && std::memcmp(rec + 8 * e + 7, a + 7, sizeof(double)) == 0) ++e;
    for (long r0 = k; r0 < e; r0 += ROW) {
      RowDesc d;
      d.nx = a[0]; d.ny = a[1]; d.nz = a[2]; d.d = a[3]; d.s = a[7];
      d.count = (int32_t)((e - r0) < ROW ? (e - r0) : ROW);
      d.first = r0 == k;
      d.pad_[0] = d.pad_[1] = 0.0;
      desc.push_back(d);
      for (int l = 0; l < ROW; ++l) {
        const bool v = r0 + l < e;
        for (int c = 0; c < 3; ++c) xyz.push_back(v ? rec[8 * (r0 + l) + 4 + c] : 0.0);
      }
    }
    k = e;
  }
  const long n_rows = (long)desc.size();
  std::vector<double> acc(64 * 28, 0.0);
  std::vector<RowMoments3> M(64);
  RowPlane q;
  bool have = false;
  auto flush_all = [&]() {
    for (int l = 0; l < 64; ++l) { if (with_loss) rows3_flush<true>(q, M[l], &acc[28 * l]); else rows3_flush<false>(q, M[l], &acc[28 * l]); }
  };
  for (long r = 0; r < n_rows; ++r) {
    const RowDesc& d = desc[r];
    if (d.first || (rows_per_wave > 0 && r % rows_per_wave == 0)) {
      if (have) flush_all();
      rows_plane_setup(R, t, d.nx, d.ny, d.nz, d.d, d.s, q);
      for (int l = 0; l < 64; ++l) { if (with_loss) rows3_moments_reset<true>(M[l]); else rows3_moments_reset<false>(M[l]); }
      have = true;
    }
    for (int l = 0; l < d.count; ++l) {
      const double* p = &xyz[(r * 64 + l) * 3];
      const bool renorm = (r & 1) != 0;  // as the kernels: product renormalised every second row
      if (with_loss) rows3_point<true>(q, inv_lf2, p[0], p[1], p[2], M[l], renorm); else rows3_point<false>(q, inv_lf2, p[0], p[1], p[2], M[l], renorm);
    }
  }
  if (have) flush_all();
  for (int c = 0; c < 28; ++c) {
    double s = 0.0;
    for (int l = 0; l < 64; ++l) s += acc[28 * l + c];
    out28[c] = s;
  }
  return n_rows;
}

double shim_log_mant_exp(double m, int e) { return clc::log_mant_exp(m, e); }

}  // extern "C"
