// abi_frontend.hip — the calls either side of the solve: factor evaluation, manifold plus, information matrix and closed form, line fitting, scan conversion.
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

namespace clc_abi {
void warm_frontend() {
  warm_kernel(reinterpret_cast<const void*>(&clc::factor_kernel));
  warm_kernel(reinterpret_cast<const void*>(&clc::normal9_kernel));
  warm_kernel(reinterpret_cast<const void*>(&clc::line_fit_kernel<true>));
}
}  // namespace clc_abi

extern "C" {

int clc_factor_evaluate(clc_handle* h, const double pose[7], double* residuals, double* jacobians) {
  if (!h || !pose || !residuals) return fail(CLC_ERR_INVALID_ARG, "clc_factor_evaluate: bad argument");
  if (!h->d_tiles) return fail(CLC_ERR_NO_DATA, "clc_factor_evaluate: no observations uploaded");
  CLC_HIP(hipSetDevice(h->device));
  const size_t n = h->n_obs;
  if (n == 0) return CLC_OK;
  DevBuf<double> br(&h->pool), bj(&h->pool);
  CLC_HIP(br.alloc(n));
  if (jacobians) CLC_HIP(bj.alloc(n * 7));
  double *d_r = br.p, *d_j = bj.p;
  std::memcpy(h->h_small, pose, 7 * sizeof(double));
  CLC_HIP(hipMemcpyAsync(h->d_small, h->h_small, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int threads = 256;
  hipLaunchKernelGGL(clc::factor_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, h->d_tiles, (long long)n, h->d_small, d_r, d_j);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(residuals, d_r, n * sizeof(double), hipMemcpyDeviceToHost));
  if (jacobians) CLC_HIP(hipMemcpy(jacobians, d_j, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
  return CLC_OK;
}

int clc_pose_plus(clc_handle* h, const double* x, const double* delta, double* out, size_t n) {
  if (!h || (n > 0 && (!x || !delta || !out))) return fail(CLC_ERR_INVALID_ARG, "clc_pose_plus: bad argument");
  if (n == 0) return CLC_OK;
  CLC_HIP(hipSetDevice(h->device));
  DevBuf<double> buf(&h->pool);
  CLC_HIP(buf.alloc(n * 20));
  double *d_x = buf.p, *d_d = buf.p + 7 * n, *d_o = buf.p + 13 * n;
  CLC_HIP(hipMemcpy(d_x, x, n * 7 * sizeof(double), hipMemcpyHostToDevice));
  CLC_HIP(hipMemcpy(d_d, delta, n * 6 * sizeof(double), hipMemcpyHostToDevice));
  const int threads = 256;
  hipLaunchKernelGGL(clc::plus_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, d_x, d_d, d_o, (long long)n);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  CLC_HIP(hipMemcpy(out, d_o, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
  return CLC_OK;
}

int clc_pose_plus_jacobian(const double* /*x*/, double jacobian[42]) {
  if (!jacobian) return fail(CLC_ERR_INVALID_ARG, "clc_pose_plus_jacobian: NULL output");
  for (int i = 0; i < 42; ++i) jacobian[i] = 0.0;
  for (int i = 0; i < 6; ++i) jacobian[6 * i + i] = 1.0;  // [I6; 0], pose_local_parameterization.cpp:36-37
  return CLC_OK;
}

int clc_information(clc_handle* h, const double pose[7], double H[36], double b[6], double* chi2,
                    double sv[6], double V[36], int* n_null) {
  if (!h || !pose || !H || !b || !chi2 || !sv || !n_null)
    return fail(CLC_ERR_INVALID_ARG, "clc_information: bad argument");
  double cost, g[6], H21[21];
  int rc = clc_eval(h, pose, /*with_loss=*/0, 0.0, &cost, g, H21);  // :323-362: no loss
  if (rc != CLC_OK) return rc;
  int idx = 0;
  for (int a = 0; a < 6; ++a)
    for (int c = a; c < 6; ++c) {
      H[6 * a + c] = H21[idx];
      H[6 * c + a] = H21[idx];
      ++idx;
    }
  for (int a = 0; a < 6; ++a) b[a] = -g[a];  // b -= J^T r, :357
  *chi2 = 2.0 * cost;                        // chi += r*r, :359
  double Vtmp[36];
  clc::host::jacobi_eig_sym(H, 6, sv, V ? V : Vtmp);  // JacobiSVD(H), :366
  int n = 0;
  for (int i = 0; i < 6; ++i)
    if (sv[i] < 1e-8) ++n;  // :371
  *n_null = n;
  return CLC_OK;
}

int clc_closed_form(clc_handle* h, double Tlc[16], int* unobservable, double sv9[9]) {
  if (!h || !Tlc || !unobservable) return fail(CLC_ERR_INVALID_ARG, "clc_closed_form: bad argument");
  if (!h->d_tiles || h->n_obs == 0) return fail(CLC_ERR_NO_DATA, "clc_closed_form: no observations uploaded");
  CLC_HIP(hipSetDevice(h->device));
  const int grid = eval_grid(h, h->n_obs);
  int rc = ensure_partials(h, grid);
  if (rc != CLC_OK) return rc;
  if (use_rows(h)) {
    const clc::RowDesc* rdesc = reinterpret_cast<const clc::RowDesc*>(h->d_rdesc);
    if (h->rows_z) {  // bar_p = (x, y, 1): z is not read, only the row stride differs
      if (rows_nontemporal(h, h->n_rows, true))
        hipLaunchKernelGGL((clc::normal9_rows_kernel<true, clc::ROW_DOUBLES_Z>), dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy, rdesc, h->n_rows, h->d_partials);
      else
        hipLaunchKernelGGL((clc::normal9_rows_kernel<false, clc::ROW_DOUBLES_Z>), dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy, rdesc, h->n_rows, h->d_partials);
    }
    else if (rows_nontemporal(h, h->n_rows))
      hipLaunchKernelGGL(clc::normal9_rows_kernel<true>, dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy,
                         reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, h->d_partials);
    else
      hipLaunchKernelGGL(clc::normal9_rows_kernel<false>, dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_rxy,
                         reinterpret_cast<const clc::RowDesc*>(h->d_rdesc), h->n_rows, h->d_partials);
  } else {
    hipLaunchKernelGGL(clc::normal9_kernel, dim3(grid), dim3(clc::BLOCK), 0, h->stream, h->d_tiles,
                       (long long)h->n_obs, h->d_partials);
  }
  CLC_HIP(hipGetLastError());
  hipLaunchKernelGGL(clc::reduce9_kernel, dim3(1), dim3(clc::BLOCK), 0, h->stream, h->d_partials, grid,
                     h->d_small + 128);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipMemcpyAsync(h->h_small + 128, h->d_small + 128, clc::NACC9 * sizeof(double),
                         hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  const double* r = h->h_small + 128;
  auto tri3 = [](int a, int b) { if (a > b) std::swap(a, b); return a * 3 - (a * (a - 1)) / 2 + (b - a); };
  double AtA[81], Atb[9];
  for (int ci = 0; ci < 3; ++ci)
    for (int ri = 0; ri < 3; ++ri) {
      for (int cj = 0; cj < 3; ++cj)
        for (int rj = 0; rj < 3; ++rj) AtA[9 * (3 * ci + ri) + (3 * cj + rj)] = r[6 * tri3(ci, cj) + tri3(ri, rj)];
      Atb[3 * ci + ri] = r[36 + 3 * ci + ri];
    }
  rc = clc::host::closed_form_from_normal(AtA, Atb, Tlc, unobservable, sv9);
  if (rc != CLC_OK) return fail(rc, "clc_closed_form: non-finite solution of the 9x9 normal equation");
  return CLC_OK;
}

// ---- line fitting ---------------------------------------------------------------------------
void clc_line_options_default(clc_options* o) {
  clc_options_default(o);
  if (!o) return;
  o->max_num_iterations = 10;   // src/LaseCamCalCeres.cpp:425
  o->loss_scale_factor = 0.05;  // CauchyLoss(0.05), :416 (no per-residual scale here)
}

int clc_line_fit_batched(clc_handle* h, const clc_options* opt_in, const double* xy, const int64_t* offsets,
                         size_t n_scans, double* lines, clc_summary* summaries) {
  if (!h || !offsets || !lines || (n_scans > 0 && offsets[n_scans] > offsets[0] && !xy))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: bad argument");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_line_options_default(&opt);
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: max_num_iterations < 0");
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: loss_scale_factor must be > 0");
  if (n_scans == 0) return CLC_OK;
  if (n_scans > 0x7FFFFFF0ull) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: too many scans");
  for (size_t k = 0; k < n_scans; ++k)
    if (offsets[k + 1] < offsets[k]) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched: offsets not monotone");
  for (size_t i = 0; i < 2 * n_scans; ++i)
    if (!std::isfinite(lines[i])) return fail(CLC_ERR_NONFINITE, "clc_line_fit_batched: non-finite initial line");
  CLC_HIP(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  const size_t n_pts = (size_t)(offsets[n_scans] - offsets[0]);
  std::vector<long long> rel(n_scans + 1);
  for (size_t k = 0; k <= n_scans; ++k) rel[k] = offsets[k] - offsets[0];
  DevBuf<double> bxy(&h->pool), blines(&h->pool);
  DevBuf<long long> boff(&h->pool);
  DevBuf<clc_summary> bsum(&h->pool);
  CLC_HIP(bxy.alloc(n_pts * 2));
  CLC_HIP(boff.alloc(n_scans + 1));
  CLC_HIP(blines.alloc(n_scans * 2));
  if (summaries) CLC_HIP(bsum.alloc(n_scans));
  double *d_xy = bxy.p, *d_lines = blines.p;
  long long* d_off = boff.p;
  clc_summary* d_sum = bsum.p;
  hipError_t e = hipSuccess;
  if (n_pts > 0) e = hipMemcpyAsync(d_xy, xy + 2 * offsets[0], n_pts * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_off, rel.data(), (n_scans + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_lines, lines, n_scans * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    const unsigned blocks = (unsigned)((n_scans + clc::LINE_SCANS_PER_BLOCK - 1) / clc::LINE_SCANS_PER_BLOCK);
    if (opt.use_loss)
      hipLaunchKernelGGL((clc::line_fit_kernel<true>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, d_xy, d_off,
                         (int)n_scans, opt, d_lines, d_sum);
    else
      hipLaunchKernelGGL((clc::line_fit_kernel<false>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, d_xy, d_off,
                         (int)n_scans, opt, d_lines, d_sum);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(lines, d_lines, n_scans * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && summaries)
    e = hipMemcpyAsync(summaries, d_sum, n_scans * sizeof(clc_summary), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail(CLC_ERR_HIP, "clc_line_fit_batched", e);
  if (summaries) {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t k = 0; k < n_scans; ++k) summaries[k].solve_ms = ms;
  }
  return CLC_OK;
}

int clc_scan_to_points(clc_handle* h, const float* ranges, const int64_t* offsets, size_t n_scans,
                       const float* angle_min, const float* angle_increment, const float* range_min,
                       double* points) {
  if (!h || !offsets || (n_scans > 0 && (!angle_min || !angle_increment || !range_min)))
    return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: bad argument");
  if (n_scans == 0) return CLC_OK;
  if (n_scans > 65535) return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: at most 65535 scans per call");
  for (size_t k = 0; k < n_scans; ++k)
    if (offsets[k + 1] < offsets[k]) return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: offsets not monotone");
  const size_t n = (size_t)(offsets[n_scans] - offsets[0]);
  if (n == 0) return CLC_OK;
  if (!ranges || !points) return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  std::vector<long long> rel(n_scans + 1);
  long long longest = 0;
  for (size_t k = 0; k <= n_scans; ++k) rel[k] = offsets[k] - offsets[0];
  for (size_t k = 0; k < n_scans; ++k) longest = std::max(longest, rel[k + 1] - rel[k]);
  DevBuf<float> br(&h->pool), bam(&h->pool), bai(&h->pool), brm(&h->pool);
  DevBuf<long long> boff(&h->pool);
  DevBuf<double> bp(&h->pool);
  CLC_HIP(br.alloc(n)); CLC_HIP(bam.alloc(n_scans)); CLC_HIP(bai.alloc(n_scans)); CLC_HIP(brm.alloc(n_scans));
  CLC_HIP(boff.alloc(n_scans + 1)); CLC_HIP(bp.alloc(3 * n));
  CLC_HIP(hipMemcpyAsync(br.p, ranges + offsets[0], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(bam.p, angle_min, n_scans * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(bai.p, angle_increment, n_scans * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(brm.p, range_min, n_scans * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(boff.p, rel.data(), (n_scans + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  const int threads = 256;
  const unsigned gx = (unsigned)std::min<long long>(64, std::max<long long>(1, (longest + threads - 1) / threads));
  hipLaunchKernelGGL(clc::scan_to_points_kernel, dim3(gx, (unsigned)n_scans), dim3(threads), 0, h->stream, br.p, boff.p,
                     (int)n_scans, bam.p, bai.p, brm.p, bp.p);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipMemcpyAsync(points + 3 * offsets[0], bp.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  return CLC_OK;
}

int clc_line_fit_batched_device(clc_handle* h, const clc_options* opt_in, const double* xy_dev, const int64_t* offsets_dev,
                                size_t n_scans, double* lines_dev, clc_summary* summaries_dev) {
  if (!h || (n_scans > 0 && (!offsets_dev || !lines_dev || !xy_dev)))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: bad argument");
  clc_options opt;
  if (opt_in) opt = *opt_in; else clc_line_options_default(&opt);
  if (opt.max_num_iterations < 0) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: max_num_iterations < 0");
  if (opt.use_loss && !(opt.loss_scale_factor > 0.0))
    return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: loss_scale_factor must be > 0");
  if (n_scans == 0) return CLC_OK;
  if (n_scans > 0x7FFFFFF0ull) return fail(CLC_ERR_INVALID_ARG, "clc_line_fit_batched_device: too many scans");
  CLC_HIP(hipSetDevice(h->device));
  static_assert(sizeof(long long) == sizeof(int64_t), "offset type");
  const unsigned blocks = (unsigned)((n_scans + clc::LINE_SCANS_PER_BLOCK - 1) / clc::LINE_SCANS_PER_BLOCK);
  const long long* d_off = reinterpret_cast<const long long*>(offsets_dev);
  if (opt.use_loss)
    hipLaunchKernelGGL((clc::line_fit_kernel<true>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, xy_dev, d_off,
                       (int)n_scans, opt, lines_dev, summaries_dev);
  else
    hipLaunchKernelGGL((clc::line_fit_kernel<false>), dim3(blocks), dim3(clc::BLOCK), 0, h->stream, xy_dev, d_off,
                       (int)n_scans, opt, lines_dev, summaries_dev);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  return CLC_OK;
}

int clc_scan_to_points_device(clc_handle* h, const float* ranges_dev, const int64_t* offsets_dev, size_t n_scans,
                              size_t n_rays, const float* angle_min_dev, const float* angle_increment_dev,
                              const float* range_min_dev, double* points_dev) {
  if (!h || (n_scans > 0 && (!offsets_dev || !angle_min_dev || !angle_increment_dev || !range_min_dev)) ||
      (n_rays > 0 && (!ranges_dev || !points_dev || n_scans == 0)))
    return fail(CLC_ERR_INVALID_ARG, "clc_scan_to_points_device: bad argument");
  if (n_rays == 0) return CLC_OK;
  CLC_HIP(hipSetDevice(h->device));
  const int threads = 256;
  hipLaunchKernelGGL(clc::scan_to_points_flat_kernel, dim3((unsigned)((n_rays + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, ranges_dev, reinterpret_cast<const long long*>(offsets_dev), (long long)n_scans,
                     (long long)n_rays, angle_min_dev, angle_increment_dev, range_min_dev, points_dev);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));
  return CLC_OK;
}


}  // extern "C"
