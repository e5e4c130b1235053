// What can a second wave on the same SIMD do while the first streams fp64 MFMA?  Waves 0-3: MFMA stream (nm x 4);
// waves 4-7: kind 0 = dependent v_fma_f64 chain, 1 = 8 independent v_fma_f64 chains, 2 = v_fma_f32 x8, 3 = v_add_u32 x8,
// 4 = ds_read_b64 stream, 5 = v_readlane chain.  Reported: cycles of each group alone and together.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_overlap2.hip -o tools/probe_overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void kd(double* out, long long* cyc, int mode, int kind, int nm, int nv) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 1e-3;
  double res = 0;
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    if (mode != 1) {
      d4 z = {0, 0, 0, 0}, c0 = z, c1 = z, c2 = z, c3 = z;
      double a = lane * 1e-3, b = 1.0 - lane * 1e-3;
      for (int i = 0; i < nm; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
      d4 s = c0 + c1 + c2 + c3; res = s[0] + s[1] + s[2] + s[3];
    }
  } else if (mode != 0) {
    if (kind == 0) { double x = lane * 1e-9 + 1.0; for (int i = 0; i < nv * 8; ++i) x = fma(x, 0.999999, 1e-9); res = x; }
    if (kind == 1) { double a = lane * 1e-9 + 1.0, c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
      for (int i = 0; i < nv; ++i) for (int k = 0; k < 8; ++k) c[k] = fma(a, c[k], 1e-9);
      for (int k = 0; k < 8; ++k) res += c[k]; }
    if (kind == 2) { float a = lane * 1e-6f + 1.0f, c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
      for (int i = 0; i < nv; ++i) for (int k = 0; k < 8; ++k) c[k] = fmaf(a, c[k], 1e-6f);
      for (int k = 0; k < 8; ++k) res += c[k]; }
    if (kind == 3) { unsigned c[8] = {0, 1, 2, 3, 4, 5, 6, 7}; unsigned a = lane;
      for (int i = 0; i < nv; ++i) for (int k = 0; k < 8; ++k) { c[k] = c[k] * 3u + a; asm volatile("" : "+v"(c[k])); }
      for (int k = 0; k < 8; ++k) res += c[k]; }
    if (kind == 4) { double s = 0; int idx = lane;
      for (int i = 0; i < nv; ++i) for (int k = 0; k < 8; ++k) { s += lds[(idx + 64 * k) & 4095]; idx += 7; }
      res = s; }
    if (kind == 5) { int v = lane;
      for (int i = 0; i < nv * 8; ++i) { v = __builtin_amdgcn_readlane(v, 5) + lane; asm volatile("" : "+v"(v)); }
      res = v; }
  }
  long long t1 = clock64();
  out[threadIdx.x] = res;
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  double* out; long long* cyc, hc[8];
  (void)hipMalloc(&out, 8 * 4096); (void)hipMalloc(&cyc, 64);
  const int nm = 512, nv = 1024;
  const char* names[] = {"dep v_fma_f64 chain", "8 indep v_fma_f64", "8 indep v_fma_f32", "8 indep int mad", "ds_read_b64 stream", "v_readlane chain"};
  for (int kind = 0; kind < 6; ++kind) {
    long long r[3][2];
    for (int mode = 0; mode < 3; ++mode) {
      for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(kd, dim3(1), dim3(512), 0, 0, out, cyc, mode, kind, nm, nv); (void)hipDeviceSynchronize(); }
      (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
      r[mode][0] = hc[0]; r[mode][1] = hc[4];
    }
    printf("%-22s: mfma alone %7lld | other alone %7lld | together: mfma %7lld other %7lld\n", names[kind], r[0][0], r[1][1], r[2][0], r[2][1]);
  }
  return 0;
}
