// accuracy of v_rsq_f64 and of rsq + 1 or 2 Newton steps against 1/sqrt(x) in double
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* o0, double* o1, double* o2, double* o3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double y = __builtin_amdgcn_rsq(a);
  o0[i] = y;
  double h = 0.5 * a;
  double e = fma(-(h * y), y, 0.5);
  double y1 = fma(y, e, y);
  o1[i] = y1;
  e = fma(-(h * y1), y1, 0.5);
  o2[i] = fma(y1, e, y1);
  // one third-order (Halley-type) step: y (1 + e (1 + 1.5 e)), e = 0.5 - h y^2  -> error ~ e^3
  e = fma(-(h * y), y, 0.5);
  o3[i] = fma(y * e, fma(1.5, e, 1.0), y);
}
int main() {
  const int n = 1 << 20;
  double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n], *h3 = new double[n];
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX - 0.5) * 60.0) * (1.0 + (double)rand() / RAND_MAX);
  double *dx, *d0, *d1, *d2, *d3;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h3, d3, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  for (int i = 0; i < n; ++i) {
    double r = 1.0 / sqrt(hx[i]);
    m0 = fmax(m0, fabs(h0[i] - r) / r); m1 = fmax(m1, fabs(h1[i] - r) / r); m2 = fmax(m2, fabs(h2[i] - r) / r); m3 = fmax(m3, fabs(h3[i] - r) / r);
  }
  printf("max rel err: rsq %.3e | +1 Newton %.3e | +2 Newton %.3e | +1 third-order %.3e  (eps = 2.2e-16)\n", m0, m1, m2, m3);
  return 0;
}
