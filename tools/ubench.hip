// Micro-probes for gfx950 fp64: cycles per v_mfma_f64_16x16x4_f64 (independent / dependent chains),
// v_fma_f64, exp, rsqrt+div.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 2048
__global__ void k_mfma(double* out, long long* cyc, int nacc) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
  d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = clock64();
  if (nacc == 4) {
    for (int i = 0; i < N; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < 4 * N; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  }
  long long t1 = clock64();
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma(double* out, long long* cyc) {
  double a = threadIdx.x * 1e-9 + 1.0, b = 1e-9;
  double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) {
    c0 = fma(a, c0, b); c1 = fma(a, c1, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b);
    c4 = fma(a, c4, b); c5 = fma(a, c5, b); c6 = fma(a, c6, b); c7 = fma(a, c7, b);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_exp(double* out, long long* cyc) {
  double x = threadIdx.x * 1e-3, s = 0;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) { s += exp(-0.5 * x * x); x += 1e-4; }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rsq(double* out, long long* cyc) {
  double x = threadIdx.x * 1e-3 + 1.0;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) { x = 1.0 / sqrt(x) + 1.5; }   // dependent chain: latency of sqrt + div
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* cyc; long long h[4096];
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, sizeof(h));
  struct { const char* name; int blocks, threads, mode; } cases[] = {
    {"mfma 4acc 1wave/SIMD all CUs", 256, 256, 4}, {"mfma dep  1wave/SIMD all CUs", 256, 256, 1},
    {"mfma 4acc 2wave/SIMD all CUs", 256, 512, 4}, {"mfma 4acc 1 wave alone", 1, 64, 4}, {"mfma dep 1 wave alone", 1, 64, 1}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma, dim3(c.blocks), dim3(c.threads), 0, 0, out, cyc, c.mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, sizeof(long long) * c.blocks, hipMemcpyDeviceToHost);
      double waves = (double)c.blocks * c.threads / 64;
      if (rep) printf("%-32s: %8.1f clk/mfma (wave view), %.3f ms, %.2f TFLOP/s\n", c.name, (double)h[0] / (4.0 * N), ms,
                      waves * 4.0 * N * 2048.0 / (ms * 1e-3) / 1e12);
    }
  }
  for (int t = 256; t <= 1024; t *= 2) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(256), dim3(t), 0, 0, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
      if (rep) printf("v_fma_f64 x8 chains, %4d thr/CU   : %8.2f clk/fma (wave view), %.2f TFLOP/s\n", t, (double)h[0] / (8.0 * N),
                      256.0 * t * 8.0 * N * 2 / (ms * 1e-3) / 1e12);
    }
  }
  hipLaunchKernelGGL(k_exp, dim3(256), dim3(256), 0, 0, out, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  printf("exp(f64), 1 wave/SIMD            : %8.1f clk/exp (wave view)\n", (double)h[0] / N);
  hipLaunchKernelGGL(k_exp, dim3(256), dim3(1024), 0, 0, out, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  printf("exp(f64), 4 waves/SIMD           : %8.1f clk/exp (wave view)\n", (double)h[0] / N);
  hipLaunchKernelGGL(k_rsq, dim3(256), dim3(256), 0, 0, out, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  printf("1/sqrt(x) dependent chain        : %8.1f clk per (sqrt+div+add)\n", (double)h[0] / N);
  return 0;
}
