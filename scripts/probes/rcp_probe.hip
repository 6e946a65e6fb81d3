// Accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of 0, 1, 2 Newton steps on gfx950 (max relative error in ulps of 2^-52
// over 2^22 random arguments in [1, 2^40)): decides how many steps the per-point Cauchy weight needs.
// hipcc --offload-arch=gfx950 -O2 rcp_probe.hip -o rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
__global__ void k(const double* x, int n, double* err) {  // err[4 * i + s]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i];
  double r = __builtin_amdgcn_rcp(a);
  err[6 * i + 0] = r;
  r = fma(r, fma(-a, r, 1.0), r);
  err[6 * i + 1] = r;
  r = fma(r, fma(-a, r, 1.0), r);
  err[6 * i + 2] = r;
  double y = __builtin_amdgcn_rsq(a);
  err[6 * i + 3] = y;
  const double h = 0.5 * a;
  y = fma(y, fma(-h * y, y, 0.5), y);
  err[6 * i + 4] = y;
  y = fma(y, fma(-h * y, y, 0.5), y);
  err[6 * i + 5] = y;
}
int main() {
  const int n = 1 << 22;
  double* hx = (double*)malloc(sizeof(double) * n);
  double* he = (double*)malloc(sizeof(double) * 6 * n);
  srand48(7);
  for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + drand48(), (int)(drand48() * 40));
  double *dx, *de;
  hipMalloc(&dx, sizeof(double) * n);
  hipMalloc(&de, sizeof(double) * 6 * n);
  hipMemcpy(dx, hx, sizeof(double) * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, n, de);
  hipMemcpy(he, de, sizeof(double) * 6 * n, hipMemcpyDeviceToHost);
  double mx[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double a = hx[i];
    const long double rr = 1.0L / a, rs = 1.0L / sqrtl(a);
    for (int s = 0; s < 6; ++s) {
      const long double ref = s < 3 ? rr : rs;
      const double e = (double)(fabsl((long double)he[6 * i + s] - ref) / ref) * 4503599627370496.0;
      if (e > mx[s]) mx[s] = e;
    }
  }
  printf("v_rcp_f64 max rel err [ulp]: seed %.3g, 1 step %.3g, 2 steps %.3g\n", mx[0], mx[1], mx[2]);
  printf("v_rsq_f64 max rel err [ulp]: seed %.3g, 1 step %.3g, 2 steps %.3g\n", mx[3], mx[4], mx[5]);
  return 0;
}
