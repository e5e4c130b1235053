// Dependent-issue latencies on gfx950, one wave alone on a SIMD (what bounds the pivot chain of diag16, tile_f64.hpp).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_lat.hip -o tools/probe_lat ; prints shader cycles (s_memtime) per op.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(REP4(__VA_ARGS__))
#define TIMED(name, idx, setup, ...)                                                                 \
  {                                                                                                  \
    setup;                                                                                           \
    unsigned long long t0 = __builtin_readcyclecounter();                                            \
    for (int i = 0; i < reps; ++i) { REP16(__VA_ARGS__) }                                                   \
    unsigned long long t1 = __builtin_readcyclecounter();                                            \
    if (threadIdx.x == 0) cyc[idx] = t1 - t0;                                                        \
  }

__global__ __launch_bounds__(64) void k(double* io, unsigned long long* cyc, int reps) {
  __shared__ double lds[128];
  const int lane = threadIdx.x;
  double a = io[lane], b = io[64 + lane], c = io[128 + lane];
  float fa = (float)a, fb = (float)b;
  lds[lane] = a;
  lds[64 + lane] = b;
  __syncthreads();
  // 0: dependent v_fma_f64
  TIMED("fma64", 0, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
  // 1: dependent v_mul_f64
  TIMED("mul64", 1, , asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
  // 2: dependent v_add_f64
  TIMED("add64", 2, , asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));)
  // 3: dependent v_rsq_f64
  TIMED("rsq64", 3, , asm volatile("v_rsq_f64 %0, %0" : "+v"(a));)
  // 4: dependent v_rcp_f64
  TIMED("rcp64", 4, , asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
  // 5: dependent v_fma_f32
  TIMED("fma32", 5, , asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa) : "v"(fb));)
  // 6: dependent v_rsq_f32
  TIMED("rsq32", 6, , asm volatile("v_rsq_f32 %0, %0" : "+v"(fa));)
  // 7: readlane x2 -> fma (VALU -> SGPR -> VALU round trip)
  TIMED("readlane+fma", 7, ,
        {
          int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3);
          double s = __hiloint2double(hi, lo);
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "s"(s), "v"(b));
        })
  // 8: DPP row_newbcast mov (b64 as two b32) -> fma
  TIMED("dpp_bcast+fma", 8, ,
        {
          int lo = __builtin_amdgcn_mov_dpp(__double2loint(a), 0x153, 0xf, 0xf, false);
          int hi = __builtin_amdgcn_mov_dpp(__double2hiint(a), 0x153, 0xf, 0xf, false);
          double s = __hiloint2double(hi, lo);
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(s), "v"(b));
        })
  // 9: ds_bpermute x2 -> fma
  TIMED("bpermute+fma", 9, ,
        {
          int lo = __builtin_amdgcn_ds_bpermute(12, __double2loint(a)), hi = __builtin_amdgcn_ds_bpermute(12, __double2hiint(a));
          double s = __hiloint2double(hi, lo);
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(s), "v"(b));
        })
  // 10: LDS write + read round trip -> fma
  TIMED("lds_rt+fma", 10, ,
        {
          lds[lane] = a;
          __builtin_amdgcn_wave_barrier();
          double s = lds[lane ^ 17];
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(s), "v"(b));
        })
  // 11: independent fma64 x4 chains (issue rate)
  {
    double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    TIMED("fma64x4", 11, ,
          asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                       : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    a += a1 + a2 + a3;
  }
  // 12: dependent MFMA f64 16x16x4 (D -> C)
  {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {a, b, c, a};
    TIMED("mfma_dep", 12, , acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);)
    a += acc[0] + acc[1] + acc[2] + acc[3];
    // 13: MFMA whose B operand is the previous result (acc -> operand)
    TIMED("mfma_dep_op", 13, , acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, acc[0], acc, 0, 0, 0);)
    a += acc[0] + acc[1] + acc[2] + acc[3];
    // 14: 4x4x4 MFMA dependent
    double s4 = a;
    TIMED("mfma4x4_dep", 14, , s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, s4, 0, 0, 0);)
    a += s4;
    // 15: MFMA then dependent VALU fma on its result then MFMA on that (MFMA -> VALU -> MFMA)
    TIMED("mfma+fma", 15, ,
          {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[0]) : "v"(b), "v"(c));
          })
    a += acc[0];
  }
  // 16: v_readlane only chain (s -> v_mov -> readlane)
  {
    int x = lane;
    TIMED("readlane_b32", 16, , x = __builtin_amdgcn_readlane(x, 5) + lane;)
    a += x;
  }
  // 17: permlane32_swap chain
  {
    unsigned x = lane;
    TIMED("permlane32", 17, , { auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); x = r[0] + r[1]; })
    a += x;
  }
  // 18: f32 chain of 3 different ops (mul, fma, add)
  TIMED("v_sqrt_f64", 18, , asm volatile("v_sqrt_f64 %0, %0" : "+v"(a));)
  // 19: two interleaved independent dependent chains of fma64 (ILP 2)
  {
    double a1 = a + 1;
    TIMED("fma64x2", 19, , asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(a1) : "v"(b), "v"(c));)
    a += a1;
  }
  io[192 + lane] = a + fa;
}

int main() {
  double h[256];
  for (int i = 0; i < 256; ++i) h[i] = 1.0 + 1e-3 * (i % 64);
  double* io;
  unsigned long long *cyc, hc[32];
  hipMalloc(&io, sizeof(h));
  hipMalloc(&cyc, sizeof(hc));
  hipMemcpy(io, h, sizeof(h), hipMemcpyHostToDevice);
  const int reps = 256;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, io, cyc, reps);
    hipDeviceSynchronize();
  }
  hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  const char* names[] = {"dep v_fma_f64", "dep v_mul_f64", "dep v_add_f64", "dep v_rsq_f64", "dep v_rcp_f64", "dep v_fma_f32", "dep v_rsq_f32",
                         "readlane x2 -> fma64", "dpp row_newbcast x2 -> fma64", "ds_bpermute x2 -> fma64", "LDS write/read -> fma64",
                         "4 independent fma64 (per group of 4)", "dep mfma 16x16x4 (acc)", "dep mfma (result as B operand)", "dep mfma 4x4x4",
                         "mfma + dependent fma64 (per pair)", "dep readlane_b32 + add", "dep permlane32_swap + add", "dep v_sqrt_f64",
                         "2 interleaved fma64 chains (per pair)"};
  for (int i = 0; i < 20; ++i) printf("%-44s %8.1f cycles\n", names[i], (double)hc[i] / (16.0 * reps));
  return 0;
}
