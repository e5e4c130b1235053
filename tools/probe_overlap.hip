// Does a latency-bound VALU wave (diag16 chain) hide under another wave's fp64 MFMA stream on the SAME SIMD?
// Workgroup of 8 waves on one CU: waves 0-3 land on SIMD 0-3, waves 4-7 on SIMD 0-3 again.
// mode 0: waves 0-3 run MFMA, waves 4-7 idle;  mode 1: waves 0-3 idle, waves 4-7 run diag16;  mode 2: both.
// Build: hipcc -O3 --offload-arch=gfx950 -I hdpgpc_amd/csrc tools/probe_overlap.hip -o tools/probe_overlap
#include "tile_f64.hpp"
#include <stdio.h>
using namespace hgp;
__global__ __launch_bounds__(512) void kd(const double* in, double* out, long long* cyc, int mode, int nm, int nd) {
  __shared__ double scr_all[8 * DIAG_SCR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* scr = scr_all + wave * DIAG_SCR;
  d4 X0;
  for (int r = 0; r < 4; ++r) X0[r] = in[(lane / 16 + 4 * r) * 16 + (lane & 15)];
  PivotAcc pa;
  pa.init();
  d4 w = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    if (mode != 1) {
      d4 c0 = w, c1 = w, c2 = w, c3 = w;
      double a = X0[0], b = X0[1];
      for (int i = 0; i < nm; ++i) {
        c0 = mfma(a, b, c0); c1 = mfma(a, b, c1); c2 = mfma(a, b, c2); c3 = mfma(a, b, c3);
      }
      w = c0 + c1 + c2 + c3;
    }
  } else {
    if (mode != 0) {
      for (int i = 0; i < nd; ++i) {
        d4 X = X0;
        for (int r = 0; r < 4; ++r) X[r] += 1e-300 * w[r];
        w = diag16(X, scr, lane, pa, 0, nullptr, 0, 16);
      }
    }
  }
  long long t1 = clock64();
  for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = w[r];
  if (lane == 0) { out[2048 + wave] = pa.logdet() + pa.info; cyc[wave] = t1 - t0; }
}
int main() {
  double h[256], *in, *out; long long* cyc, hc[8];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1 + abs(i - j));
  (void)hipMalloc(&in, sizeof(h)); (void)hipMalloc(&out, 8 * 4096); (void)hipMalloc(&cyc, 64);
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int nm = 1024, nd = 64;
  for (int mode = 0; mode < 3; ++mode) {
    for (int it = 0; it < 2; ++it) {
      hipLaunchKernelGGL(kd, dim3(1), dim3(512), 0, 0, in, out, cyc, mode, nm, nd);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("mode %d: mfma waves %lld %lld %lld %lld cycles (%d mfma: %.1f clk each) | diag waves %lld %lld %lld %lld (%d diag16: %.0f clk each)\n", mode,
           hc[0], hc[1], hc[2], hc[3], 4 * nm, hc[0] / (4.0 * nm), hc[4], hc[5], hc[6], hc[7], nd, hc[4] / (double)nd);
  }
  return 0;
}
