// Cycles and agreement of the three 16x16 diagonal-block routines of tile_f64.hpp (Cholesky + inverse of the factor on one wave):
// diag16 (MFMA-blocked), diag16_valu (column form, v_readlane multipliers), diag16_col (column form, LDS-broadcast multipliers).
// Dependent chain of 64 calls each.  Build: hipcc -O3 --offload-arch=gfx950 -I hdpgpc_amd/csrc tools/probe_diag16.hip -o tools/probe_diag16
#include "tile_f64.hpp"
#include <stdio.h>
using namespace hgp;

template <int WHICH>
__device__ __forceinline__ d4 call(const d4& X, double* scr, int lane, PivotAcc& pa, double* Lout) {
  if (WHICH == 0) return diag16(X, scr, lane, pa, 0, Lout, 16, 16);
  if (WHICH == 1) return diag16_valu(X, scr, lane, pa, 0, Lout, 16, 16);
  if (WHICH == 3) return diag16_acc(X, scr, lane, pa, 0, Lout, 16, 16);
  return diag16_col(X, scr, lane, pa, 0, Lout, 16, 16);
}

template <int WHICH>
__global__ __launch_bounds__(64) void kd(const double* in, double* out, long long* cyc, int reps) {
  __shared__ double scr[DIAG_SCR];
  int lane = threadIdx.x;
  d4 X0;
  for (int r = 0; r < 4; ++r) X0[r] = in[(lane / 16 + 4 * r) * 16 + (lane & 15)];
  PivotAcc pa;
  pa.init();
  d4 w = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; ++i) {
    d4 X = X0;
    for (int r = 0; r < 4; ++r) X[r] += 1e-300 * w[r];
    w = call<WHICH>(X, scr, lane, pa, nullptr);
  }
  long long t1 = __builtin_readcyclecounter();
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = w[r];
  if (lane == 0) {
    out[256] = pa.logdet() / reps;
    out[257] = pa.info;
    cyc[0] = t1 - t0;
  }
  // one more call with the factor written out (checks the Lout path)
  PivotAcc pb;
  pb.init();
  call<WHICH>(X0, scr, lane, pb, out + 258);
}

int main() {
  double h[256], *in, *out;
  long long *cyc, hc;
  unsigned s = 12345u;
  double B[16][16];
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      s = s * 1664525u + 1013904223u;
      B[i][j] = ((s >> 8) & 0xffff) / 65536.0 - 0.5;
    }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double a = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < 16; ++k) a += B[i][k] * B[j][k];
      h[i * 16 + j] = 300.0 * a;
    }
  hipMalloc(&in, sizeof(h));
  hipMalloc(&out, 8 * 1024);
  hipMalloc(&cyc, 8);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  // host reference: L (lower) and W = L^{-1} in long double
  long double L[16][16] = {}, W[16][16] = {};
  long double logdet = 0;
  for (int j = 0; j < 16; ++j) {
    long double d = h[j * 16 + j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    L[j][j] = sqrtl(d);
    logdet += logl(d);
    for (int i = j + 1; i < 16; ++i) {
      long double a = h[i * 16 + j];
      for (int k = 0; k < j; ++k) a -= L[i][k] * L[j][k];
      L[i][j] = a / L[j][j];
    }
  }
  for (int j = 0; j < 16; ++j)
    for (int i = j; i < 16; ++i) {
      long double a = (i == j) ? 1.0L : 0.0L;
      for (int k = j; k < i; ++k) a -= L[i][k] * W[k][j];
      W[i][j] = a / L[i][i];
    }
  const char* names[4] = {"diag16 (MFMA-blocked)", "diag16_valu (readlane)", "diag16_col (LDS bcast)", "diag16_acc (acc layout)"};
  for (int which = 0; which < 4; ++which) {
    for (int it = 0; it < 2; ++it) {
      if (which == 0) hipLaunchKernelGGL(kd<0>, dim3(1), dim3(64), 0, 0, in, out, cyc, 64);
      if (which == 1) hipLaunchKernelGGL(kd<1>, dim3(1), dim3(64), 0, 0, in, out, cyc, 64);
      if (which == 2) hipLaunchKernelGGL(kd<2>, dim3(1), dim3(64), 0, 0, in, out, cyc, 64);
      if (which == 3) hipLaunchKernelGGL(kd<3>, dim3(1), dim3(64), 0, 0, in, out, cyc, 64);
      hipDeviceSynchronize();
    }
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    double o[1024];
    hipMemcpy(o, out, 8 * 1024, hipMemcpyDeviceToHost);
    // w[s] of lane (g, c) = W[c][4s+g]
    double ew = 0, el = 0, wmax = 0, lmax = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int sidx = 0; sidx < 4; ++sidx) {
        int g = lane / 16, c = lane % 16, j = 4 * sidx + g;
        double ref = (double)W[c][j];
        ew = fmax(ew, fabs(o[lane * 4 + sidx] - ref));
        wmax = fmax(wmax, fabs(ref));
      }
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        el = fmax(el, fabs(o[258 + i * 16 + j] - (double)L[i][j]));
        lmax = fmax(lmax, fabs((double)L[i][j]));
      }
    printf("%-26s %7.0f cycles per call   |W err| %.2e (max %.2e)  |L err| %.2e (max %.2e)  logdet %.15g (ref %.15Lg) info %g\n", names[which],
           hc / 64.0, ew, wmax, el, lmax, o[256], logdet, o[257]);
  }
  return 0;
}
