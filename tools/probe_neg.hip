// Does v_mfma_f64_16x16x4_f64 treat blgp as neg:[a,b,c] on gfx950?  Prints D for a=2,b=3,c=10 with blgp=0..7.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int BLGP> __device__ double run(double a, double b, d4 c) { d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, BLGP); return r[0]; }
__global__ void k(double* out) {
  double a = 2.0, b = 3.0; d4 c = {10.0, 10.0, 10.0, 10.0};
  double r0 = run<0>(a,b,c), r1 = run<1>(a,b,c), r2 = run<2>(a,b,c), r3 = run<3>(a,b,c), r4 = run<4>(a,b,c), r5 = run<5>(a,b,c), r6 = run<6>(a,b,c), r7 = run<7>(a,b,c);
  if (threadIdx.x == 0) { out[0]=r0; out[1]=r1; out[2]=r2; out[3]=r3; out[4]=r4; out[5]=r5; out[6]=r6; out[7]=r7; }
}
int main() { double* d; double h[8]; hipMalloc(&d, 64); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("blgp=%d: D=%g  (a*b*4=24, c=10)\n", i, h[i]); return 0; }
