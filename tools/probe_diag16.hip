// Cycles of one diag16 (16x16 Cholesky + inverse on one wave), dependent chain of 64 calls.
// Build: hipcc -O3 --offload-arch=gfx950 -I hdpgpc_amd/csrc tools/probe_diag16.hip -o /tmp/probe_diag16
#include "tile_f64.hpp"
#include <stdio.h>
using namespace hgp;
__global__ __launch_bounds__(64) void kd(const double* in, double* out, long long* cyc, int reps) {
  __shared__ double scr[DIAG_SCR];
  int lane = threadIdx.x;
  d4 X0;
  for (int r = 0; r < 4; ++r) X0[r] = in[(lane / 16 + 4 * r) * 16 + (lane & 15)];
  PivotAcc pa;
  pa.init();
  d4 w = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {
    d4 X = X0;
    for (int r = 0; r < 4; ++r) X[r] += 1e-300 * w[r];
    w = diag16(X, scr, lane, pa, 0, nullptr, 0, 16);
  }
  long long t1 = clock64();
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = w[r];
  if (lane == 0) { out[256] = pa.logdet() + pa.info; cyc[0] = t1 - t0; }
}
int main() {
  double h[256], *in, *out; long long* cyc, hc;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1 + abs(i - j));
  hipMalloc(&in, sizeof(h)); hipMalloc(&out, 8 * 512); hipMalloc(&cyc, 8);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(kd, dim3(1), dim3(64), 0, 0, in, out, cyc, 64);
    hipDeviceSynchronize();
  }
  hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  double o[257]; hipMemcpy(o, out, 8 * 257, hipMemcpyDeviceToHost);
  printf("diag16: %.0f clock64 ticks per call (x 2400/100 = %.0f shader cycles if clock64 is the 100 MHz counter), logdet/64=%g w0=%g\n", hc / 64.0, hc / 64.0 * 24, o[256] / 64, o[0]);
  return 0;
}
