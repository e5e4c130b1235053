// Can v_mfma_f64_16x16x4_f64 co-execute with (a) v_fma_f64, (b) v_readlane/integer VALU of the same wave?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 4096
template <int MODE>   // 0: mfma only, 1: valu-f64 only, 2: interleaved mfma + f64 fma, 3: readlane/int only, 4: mfma + readlane/int
__global__ void k(double* out, long long* cyc) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-3;
  d4 c0 = {0, 0, 0, 0};
  double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
  int i0 = threadIdx.x, i1 = 7;
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) {
    if (MODE == 0 || MODE == 2 || MODE == 4) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 3; ++j) { f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); f3 = fma(f3, 1.0000001, 1e-9); }
    }
    if (MODE == 3 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < 6; ++j) { i1 += __builtin_amdgcn_readlane(i0, j); i0 = i0 * 3 + i1; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + f0 + f1 + f2 + f3 + i0 + i1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, double* out, long long* cyc) {
  long long h;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc); hipDeviceSynchronize(); }
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.1f clk per iteration\n", name, (double)h / N);
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8192);
  run<0>("1 mfma_f64", out, cyc);
  run<1>("12 v_fma_f64 (4 chains)", out, cyc);
  run<2>("1 mfma_f64 + 12 v_fma_f64 interleaved", out, cyc);
  run<3>("6 x (readlane + int mul/add)", out, cyc);
  run<4>("1 mfma_f64 + 6 x (readlane + int)", out, cyc);
  return 0;
}
