// Wave-level fp64 tile algebra for gfx950 (MI355X): one 64-lane wavefront owns a whole symmetric
// matrix of up to 128x128 as 16x16 tiles held in the MFMA accumulator layout, factors it with
// v_mfma_f64_16x16x4_f64 and solves against right-hand sides without leaving registers.
//
// Tile layout ("acc layout", the C/D map of v_mfma_f64_16x16x4_f64): lane l = 16*g + c holds
//   v[r] = X[g + 4r][c],  r = 0..3.
// MFMA operand maps (one double per lane per k-step of 4):  A-op  a = A[c][g],  B-op  b = B[g][c].
// Two identities make the whole factorisation register-resident:
//   (1) an acc tile X is directly the B operand of k-step s:  b_s = X[4s+g][c] = v[s];
//   (2) the same register is the A operand of X^T:            a_s = X^T[c][4s+g] = v[s].
// Hence  mfma(X.v[s], Y.v[s]) summed over s = X^T Y, which is exactly the trailing update of an
// UPPER-form Cholesky  A = U^T U  (U_KJ tiles, K <= J):   A_IJ -= U_KI^T U_KJ.
//
// Only the 16x16 diagonal blocks need cross-lane work (diag16 below); every other flop is MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// Diagnostic build only (make stamps): per-phase cycle sums with s_memtime; no stamp executes in the product build.
#ifdef HGP_STAMPS
#define HGP_STAMP_DECL unsigned long long hgp_t_, hgp_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define HGP_T0() hgp_t_ = __builtin_readcyclecounter()
#define HGP_ACC(i) do { unsigned long long n_ = __builtin_readcyclecounter(); hgp_acc_[i] += n_ - hgp_t_; hgp_t_ = n_; } while (0)
#else
#define HGP_STAMP_DECL
#define HGP_T0()
#define HGP_ACC(i)
#endif

namespace hgp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int DIAG_LD = 18;              // row stride (doubles) of the per-wave 16x16 LDS staging tile
constexpr int DIAG_SCR = 16 * DIAG_LD;   // doubles of LDS scratch each wave needs
constexpr double F64_EPS = 2.220446049250313e-16;

__host__ __device__ constexpr int tix(int I, int J, int NB) { return I * NB - (I * (I - 1)) / 2 + (J - I); }
#ifdef HGP_RACE_STRESS
// Diagnostic build (`make racestress`): after every workgroup barrier each wave sleeps a pseudo-random, wave-dependent 0-25 k
// cycles - more than any latency that could hide a missing barrier.  The GPU test-suite run against this build
// (HGP_LIB=build/probe/libhgp_race_stress.so) is the race check of every cooperative kernel.
__device__ __forceinline__ void hgp_sync_stress() {
  __syncthreads();
  const unsigned w = threadIdx.x >> 6;
  const unsigned h = ((w + 1u) * 2654435761u) ^ (unsigned)(__builtin_readcyclecounter() >> 9);
  const unsigned n = __builtin_amdgcn_readfirstlane(((h >> 3) & 3u) * 4u);
  for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
}
#define __syncthreads() ::hgp::hgp_sync_stress()
#endif

__host__ __device__ constexpr int ntiles(int NB) { return NB * (NB + 1) / 2; }

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// D = C - A B: for the f64 MFMA the blgp field is neg:[a,b,c] (checked on gfx950 with tools/probe_neg.hip),
// so the trailing updates need no VALU negation of the panel tiles.
__device__ __forceinline__ d4 mfma_sub(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
}

// Opaque copy of a per-lane value.  Everything below is fully unrolled over tiles; without this the
// compiler hoists hundreds of loop-invariant lane predicates and LDS addresses to the kernel entry and
// spills them.  Recomputing them next to their use costs a few VALU ops per tile.
__device__ __forceinline__ int launder(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ double launder_f64(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, B>), ..., f(integral_constant<int, E - 1>)
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// All-reduce over the 64 lanes without the LDS crossbar: inside each row of 16 lanes by DPP rotations (row_ror 8, 4, 2,
// 1: after the four steps every lane holds its row's result), across the four rows with the gfx950 permlane swaps.
// About 25 VALU instructions against 12 ds_bpermute round trips for the shuffle butterfly (k_hmm_messages: 0.77 -> 0.51 us per step).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_allreduce(double v) {
  auto op = [](double a, double b) { return MAX ? fmax(a, b) : a + b; };
  v = op(v, dpp_f64<0x128>(v));   // row_ror:8
  v = op(v, dpp_f64<0x124>(v));   // row_ror:4
  v = op(v, dpp_f64<0x122>(v));   // row_ror:2
  v = op(v, dpp_f64<0x121>(v));   // row_ror:1
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = op(__hiloint2double((int)b[0], (int)a[0]), __hiloint2double((int)b[1], (int)a[1]));
  lo = (unsigned)__double2loint(v);
  hi = (unsigned)__double2hiint(v);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return op(__hiloint2double((int)b[0], (int)a[0]), __hiloint2double((int)b[1], (int)a[1]));
}

__device__ __forceinline__ double wave_sum(double v) { return wave_allreduce<false>(v); }

// sum over the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48) with the gfx950 permlane swaps:
// after v_permlane16_swap(v, v) the two results hold rows (0,0,2,2) and (1,1,3,3) of v; their sum is the
// xor-16 butterfly.  v_permlane32_swap does the same for the two 32-lane halves.  No LDS crossbar traffic.
__device__ __forceinline__ double xrow_sum(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  double s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  lo = (unsigned)__double2loint(s);
  hi = (unsigned)__double2hiint(s);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// 1/sqrt(a) for a > 0 (normal range): v_rsq_f64 seed (measured 5.2e-8 relative, tools/probe_rsq.hip) + ONE third-order
// step  y (1 + e (1 + 1.5 e)),  e = 0.5 - 0.5 a y^2  -> 2.6e-16, the same as two Newton steps with a dependency
// chain two operations shorter (the pivot chain is the critical path of every factorisation here).
// LAPACK's dpotf2 also scales the column by a rounded reciprocal.
__device__ __forceinline__ double rsqrt_nr(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  const double e = fma(-(0.5 * a * y), y, 0.5);
  return fma(y * e, fma(1.5, e, 1.0), y);
}

// exp(-h) for four independent arguments h >= 0 at once (the RBF entries of E and K**).  The library exp() is one long dependent
// chain: ~650 cycles per call at one wave per SIMD, and the 37 calls per wave of the k_pairs prologue were 24 k of its 31 k cycles
// (in-kernel stamps).  Here: n = rint(-h log2 e), r = -h - n ln 2 (two-term Cody-Waite, n ln2_hi exact for |n| < 2^11), a degree-13
// Taylor polynomial of e^r on |r| <= 0.347 (truncation 4e-18, Horner), v_ldexp_f64; the four chains are interleaved instruction
// by instruction, so the FMAs issue back to back (~100 cycles per value).  Relative error <= 2.3e-16 against exp() over h in
// [0, 700] (tests/test_gpu_parity.py); arguments beyond 800 (the padding sentinels) return 0.
__device__ __forceinline__ void exp_neg4(const double (&h)[4], double (&o)[4]) {
  double n[4], r[4], p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double t = -fmin(h[i], 800.0);
    n[i] = __builtin_rint(t * 1.4426950408889634);
    r[i] = fma(n[i], -6.93147180369123816490e-01, t);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = fma(n[i], -1.90821492927058770002e-10, r[i]);
    p[i] = 1.6059043836821613e-10;   // 1/13!
  }
  constexpr double ck[13] = {1.0, 1.0, 0.5, 1.6666666666666666e-01, 4.1666666666666664e-02, 8.3333333333333332e-03,
                             1.3888888888888889e-03, 1.9841269841269841e-04, 2.4801587301587302e-05, 2.7557319223985893e-06,
                             2.7557319223985888e-07, 2.5052108385441720e-08, 2.0876756987868100e-09};   // 1/k!, k = 0 .. 12
#pragma unroll
  for (int k = 12; k >= 0; --k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fma(p[i], r[i], ck[k]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_ldexp(p[i], (int)n[i]);
}

// log-determinant accumulator (product of pivots kept as mantissa * 2^ex) + LAPACK-style info
struct PivotAcc {
  double mant;
  int ex;
  int info;  // 0 = ok, j > 0 = pivot j (1-based) was not positive
#ifdef HGP_STAMPS
  unsigned long long diag_cycles, cf[5];
  __device__ __forceinline__ void init() { mant = 1.0; ex = 0; info = 0; diag_cycles = 0; for (int i = 0; i < 5; ++i) cf[i] = 0; }
#else
  __device__ __forceinline__ void init() { mant = 1.0; ex = 0; info = 0; }
#endif
  __device__ __forceinline__ void renorm() {
    ex += __builtin_amdgcn_frexp_exp(mant);
    mant = __builtin_amdgcn_frexp_mant(mant);
  }
  // log of the product of the squared pivots u_kk^2 = log det(A)
  __device__ __forceinline__ double logdet() const { return log(mant) + (double)ex * 0.6931471805599453; }
};

// value of lane `src` (compile-time constant after unrolling) broadcast to the whole wave through SGPRs
__device__ __forceinline__ double lane_bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// diag16: Cholesky of one 16x16 diagonal block AND the inverse of its factor, by one wave.
//   in : X  (acc layout, the full symmetric block: the lower triangle must hold finite numbers)
//   out: returns W = L^{-1} (L = U^T lower) in A-operand layout, w[s] = W[c][4s+g];
//        if Lout != nullptr the wave writes L (zeros above the diagonal) for rows/cols < nvalid.
// Blocked 4 x 4 inside the tile so that the O(16^3) part runs on the matrix core too.  Register r of an acc tile holds
// rows 4r..4r+3 (row 4r+g in lane group g), which is at once the B operand "rows 4r.. of X" and the A operand
// "columns 4r.. of X^T".  Round r:
//   (1) the 10 upper entries of the 4x4 pivot block go to SGPRs (v_readlane); every lane factors it and lane group g
//       solves column g of its inverse W4 = L44^{-1} (uniform instruction stream, 4 dependent rsqrt);
//   (2) one MFMA per tile with A = W4 placed at rows 4r.. replaces rows 4r.. of X (and of Z, which starts as I) by
//       the finished rows of U (of L^{-1});
//   (3) one MFMA per tile subtracts the rank-4 product from the rows below (A operand masked to those rows).
// 16 MFMA + 4 short scalar chains per block instead of 16 elimination steps of 15 v_readlane + 15 v_fma each:
// 4.2k cycles per block in isolation (tools/probe_diag16.hip), bounded by the pivot chain (about 38 dependent fp64 ops per round).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ d4 diag16(const d4& Xin, double* scr, int lane, PivotAcc& pa, int col0,
                                     double* Lout, int ldl, int nvalid) {
  const int g = lane >> 4, c = lane & 15;
#ifdef HGP_EXP_NODIAG   // in-situ knock-out experiment (diagnostic builds only): W = diag(1 / sqrt(x_ii)), no pivot chain
  {
    d4 w;
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = (4 * s + g == c) ? 1.0 / sqrt(fabs(Xin[s]) + 1.0) : 0.0;
    pa.mant *= 1.0 + 1e-300 * Xin[0];
    return w;
  }
#endif
#ifdef HGP_STAMPS
  const unsigned long long td0 = __builtin_readcyclecounter();
#endif
  d4 X = Xin, Z;
#pragma unroll
  for (int r = 0; r < 4; ++r) Z[r] = (4 * r + g == c) ? 1.0 : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b0 = 4 * r;
    // X[4r+i][4r+j] lives in register r of lane 16 i + 4r + j
    const double x00 = lane_bcast(X[r], b0), x01 = lane_bcast(X[r], b0 + 1), x02 = lane_bcast(X[r], b0 + 2),
                 x03 = lane_bcast(X[r], b0 + 3), x11 = lane_bcast(X[r], 16 + b0 + 1), x12 = lane_bcast(X[r], 16 + b0 + 2),
                 x13 = lane_bcast(X[r], 16 + b0 + 3), x22 = lane_bcast(X[r], 32 + b0 + 2),
                 x23 = lane_bcast(X[r], 32 + b0 + 3), x33 = lane_bcast(X[r], 48 + b0 + 3);
    const double p0 = x00;
    const double r0 = rsqrt_nr(p0);
    const double u01 = x01 * r0, u02 = x02 * r0, u03 = x03 * r0;
    const double p1 = fma(-u01, u01, x11);
    const double r1 = rsqrt_nr(p1);
    const double u12 = fma(-u01, u02, x12) * r1, u13 = fma(-u01, u03, x13) * r1;
    const double p2 = fma(-u12, u12, fma(-u02, u02, x22));
    const double r2 = rsqrt_nr(p2);
    const double u23 = fma(-u12, u13, fma(-u02, u03, x23)) * r2;
    const double p3 = fma(-u23, u23, fma(-u13, u13, fma(-u03, u03, x33)));
    const double r3 = rsqrt_nr(p3);
    // (a division-free Bareiss form of this block - scaled Schur complements by multiplications only, the four 1/sqrt side by
    // side - was measured in round 2: 5.6 k instead of 4.2 k cycles per block.  The block is bound by the NUMBER of fp64
    // instructions, ~4.6 cycles each at one wave per SIMD, not by the latency of the pivot chain.)
    // NaN compares false; a bad pivot then poisons the outputs with NaN, info says where
    if (pa.info == 0) {
      if (!(p0 > 0.0)) pa.info = col0 + b0 + 1;
      else if (!(p1 > 0.0)) pa.info = col0 + b0 + 2;
      else if (!(p2 > 0.0)) pa.info = col0 + b0 + 3;
      else if (!(p3 > 0.0)) pa.info = col0 + b0 + 4;
    }
    pa.mant *= (p0 * p1) * (p2 * p3);   // four pivots between renormalisations: no over/underflow for |log2 piv| < 250
    pa.renorm();
    // column g of W4 = L44^{-1}: forward substitution of e_g (L44 = U44^T)
    const double e0 = (g == 0) ? 1.0 : 0.0, e1 = (g == 1) ? 1.0 : 0.0, e2 = (g == 2) ? 1.0 : 0.0, e3 = (g == 3) ? 1.0 : 0.0;
    const double w0 = e0 * r0;
    const double w1 = fma(-u01, w0, e1) * r1;
    const double w2 = fma(-u12, w1, fma(-u02, w0, e2)) * r2;
    const double w3 = fma(-u23, w2, fma(-u13, w1, fma(-u03, w0, e3))) * r3;
    // A operand: A[i][kk] = W4[i - 4r][kk] for 4r <= i < 4r + 4, else 0; lane (g, c) holds A[c][g]
    const int ci = c & 3;
    double aw = (ci == 0) ? w0 : (ci == 1) ? w1 : (ci == 2) ? w2 : w3;
    aw = ((c >> 2) == r) ? aw : 0.0;
    d4 tX = X, tZ = Z;
    tX[r] = 0.0;
    tZ[r] = 0.0;
    tX = mfma(aw, X[r], tX);   // rows 4r.. := W4 * rows 4r..  (the other rows pass through)
    tZ = mfma(aw, Z[r], tZ);
    if (r < 3) {
      const double am = (c >= b0 + 4) ? tX[r] : 0.0;   // rows below the pivot block only
      X = mfma_sub(am, tX[r], tX);
      Z = mfma_sub(am, tZ[r], tZ);
    } else {
      X = tX;
      Z = tZ;
    }
  }
  if (Lout != nullptr && c < nvalid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + g;                                  // L[c][i] = U[i][c]
      if (i < nvalid) Lout[(size_t)c * ldl + i] = (i <= c) ? X[r] : 0.0;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) scr[(4 * r + g) * DIAG_LD + c] = Z[r];   // Z[i][j], i = 4r+g, j = c
  __builtin_amdgcn_wave_barrier();
  d4 w;
#pragma unroll
  for (int s = 0; s < 4; ++s) w[s] = scr[c * DIAG_LD + 4 * s + g];                // W[c][4s+g]
  __builtin_amdgcn_wave_barrier();
#ifdef HGP_STAMPS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pa.diag_cycles += __builtin_readcyclecounter() - td0;
#endif
  return w;
}

// The all-VALU form of the same block (column j of U in lane j, column j of Z = L^{-1} in lane 16 + j, 16 elimination
// steps of 15 v_readlane + 15 v_fma): the same ~4.2k cycles in isolation (tools/probe_diag16.hip) with ~20 fewer live
// registers, which is what the T = 128 single-matrix kernel (k_wave_score1<8>, 36 resident tiles) needs to avoid spills.
// Round 2 measured two rewrites of the elimination loop, both correct, neither faster (isolation 4.19 / 4.38 k cycles against
// 4.27 k; inside k_pairs<8> 1.548 / 1.573 ms against 1.535 ms) and both removed: (a) the multiplier broadcast inside the
// FMA (v_fmac_f64_dpp row_newbcast, the pivot row copied into row 1 of the wave by v_permlane16_swap): one instruction per
// element instead of two v_readlane_b32 + one v_fma_f64, 35 % fewer instructions; (b) on top of it the division-free
// (Bareiss) form with the sixteen 1/sqrt taken after the loop, one per lane.  The block costs ~260 cycles per pivot whatever
// sits on the chain: a dependent fp64 VALU instruction issues ~25 cycles after its producer at one wave per SIMD.
__device__ __forceinline__ d4 diag16_valu(const d4& X, double* scr, int lane, PivotAcc& pa, int col0,
                                     double* Lout, int ldl, int nvalid) {
  const int g = lane >> 4, c = lane & 15;
#ifdef HGP_EXP_NODIAG   // in-situ knock-out experiment (diagnostic builds only)
  {
    d4 w;
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = (4 * s + g == c) ? 1.0 / sqrt(fabs(X[s]) + 1.0) : 0.0;
    pa.mant *= 1.0 + 1e-300 * X[0];
    return w;
  }
#endif
#ifdef HGP_STAMPS
  const unsigned long long td0 = __builtin_readcyclecounter();
#endif
#pragma unroll
  for (int r = 0; r < 4; ++r) scr[(g + 4 * r) * DIAG_LD + c] = X[r];
  __builtin_amdgcn_wave_barrier();
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double x = scr[i * DIAG_LD + c];                       // X[i][c]: column c (upper part is what matters)
    double e = (c == i && lane < 32) ? 1.0 : 0.0;          // identity column for the Z lanes
    v[i] = (lane < 16) ? x : e;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double piv = lane_bcast(v[k], k);                    // u_kk^2, fully updated
    if (!(piv > 0.0) && pa.info == 0) pa.info = col0 + k + 1;  // NaN compares false; a bad pivot then poisons
    pa.mant *= piv;                                            // the outputs with NaN, info says where
    if ((k & 3) == 3) pa.renorm();       // four pivots between renormalisations: no over/underflow for |log2 piv| < 250
    v[k] *= rsqrt_nr(piv);
    // all broadcasts of the step first (distinct SGPR pairs), then the FMAs: the VALU->SGPR->VALU hazard of a
    // readlane that feeds the very next instruction is paid once per step instead of once per element
    double sb[16];
#pragma unroll
    for (int kp = k + 1; kp < 16; ++kp) sb[kp] = lane_bcast(v[k], kp);   // U[k][k'] (= L[k'][k])
#pragma unroll
    for (int kp = k + 1; kp < 16; ++kp) v[kp] = fma(-sb[kp], v[k], v[kp]);
  }
  if (Lout != nullptr && lane < 16 && lane < nvalid) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < nvalid) Lout[(size_t)lane * ldl + i] = (i <= lane) ? v[i] : 0.0;   // L[j][i] = U[i][j]
  }
  if (lane >= 16 && lane < 32) {
#pragma unroll
    for (int i = 0; i < 16; ++i) scr[i * DIAG_LD + c] = v[i];                     // Z[i][j], j = c
  }
  __builtin_amdgcn_wave_barrier();
  d4 w;
#pragma unroll
  for (int s = 0; s < 4; ++s) w[s] = scr[c * DIAG_LD + 4 * s + g];                // W[c][4s+g]
  __builtin_amdgcn_wave_barrier();
#ifdef HGP_STAMPS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pa.diag_cycles += __builtin_readcyclecounter() - td0;
#endif
  return w;
}

// 1/a (a != 0, normal range): v_rcp_f64 seed + ONE third-order step  y (1 + e + e^2),  e = 1 - a y  (seed error < 2^-22 ->
// < 2^-66), three dependent operations after the seed instead of the four of two Newton steps.
__device__ __forceinline__ double rcp_nr(double a) {
  const double y = __builtin_amdgcn_rcp(a);
  const double e = fma(-a, y, 1.0);
  const double ye = y * e;
  return fma(ye, e, y + ye);
}

// product over the 16 lanes of each row of the wave (every lane gets its row's result); DPP rotations only
__device__ __forceinline__ double row16_prod(double v) {
  v *= dpp_f64<0x128>(v);   // row_ror:8
  v *= dpp_f64<0x124>(v);   // row_ror:4
  v *= dpp_f64<0x122>(v);   // row_ror:2
  v *= dpp_f64<0x121>(v);   // row_ror:1
  return v;
}
__device__ __forceinline__ int row16_sum_i32(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x122, 0xf, 0xf, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, false);
  return v;
}

// ---------------------------------------------------------------------------------------------
// diag16_col (round 3): the column form of diag16_valu with a third of its instructions.  Measured first (tools/probe_lat.hip,
// profiles/r03_probe_lat.txt): a dependent v_fma_f64 issues 9.5 cycles after its producer, v_rsq/v_rcp_f64 21, a v_readlane pair
// feeding an FMA ~10, an LDS write -> read round trip 77; independent f64 VALU instructions issue every ~5 cycles.  The 4.2 k
// cycles of diag16_valu are therefore its ~950 INSTRUCTIONS (two v_readlane_b32 per multiplier, a 7-instruction 1/sqrt, pivot
// bookkeeping in every step), not its dependency chain (~100 cycles per pivot).  Here:
//   * the rows stay UNSCALED Schur-complement rows S[k][.] during the elimination: step k publishes row k in LDS (one ds_write),
//     multiplies the pivot row by 1/pivot (v_rcp + one third-order step: t = S[k][.] / p_k) and updates v[k'] -= S[k][k'] t with
//     the multipliers S[k][k'] read back as LDS BROADCASTS (one ds_read2_b64 per two rows; only the multiplier of row k + 1,
//     which carries the next pivot, comes through v_readlane so that the LDS round trip stays off the pivot chain);
//   * the sixteen 1/sqrt(p_k) are ONE rsqrt sequence after the loop (lane c takes p_c = S[c][c] from the published rows), the row
//     scaling of Z = L^{-1} happens in the final transposition (row c of W is lane-local), and the pivot product / the first bad
//     pivot are one DPP reduction and one ballot instead of sixteen multiply / compare / renormalise groups.
// Same interface and the same results up to rounding (the multipliers are S/p instead of (S/sqrt p)(S/sqrt p)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ d4 diag16_col(const d4& X, double* scr, int lane, PivotAcc& pa, int col0,
                                         double* Lout, int ldl, int nvalid) {
  const int g = lane >> 4, c = lane & 15;
#ifdef HGP_EXP_NODIAG   // in-situ knock-out experiment (diagnostic builds only)
  {
    d4 w;
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = (4 * s + g == c) ? 1.0 / sqrt(fabs(X[s]) + 1.0) : 0.0;
    pa.mant *= 1.0 + 1e-300 * X[0];
    return w;
  }
#endif
#ifdef HGP_STAMPS
  const unsigned long long td0 = __builtin_readcyclecounter();
#endif
  const bool zrole = (g & 1) != 0;   // lanes 16-31: column c of Z (starts as I); lanes 0-15: column c of X; 32-63 mirror 0-31
#pragma unroll
  for (int r = 0; r < 4; ++r) scr[(g + 4 * r) * DIAG_LD + c] = X[r];
  if (lane < 16) scr[lane * DIAG_LD + 16] = 0.0;           // padding column 16 of the staging tile: what the Z lanes load
  __builtin_amdgcn_wave_barrier();
  double v[16];
  {
    const int cz = zrole ? 16 : c;                         // Z lanes read zeros, then get their 1.0 (one compare + one select per row)
    const int ci = zrole ? c : -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double x = scr[i * DIAG_LD + cz];              // X[i][c]: column c (the upper part is what matters)
      v[i] = __hiloint2double((ci == i) ? 0x3FF00000 : __double2hiint(x), __double2loint(x));
    }
  }
  __builtin_amdgcn_wave_barrier();
  // the staging tile is dead: row k of the multipliers goes to scr[16 k ..] (LDS operations of one wave complete in order)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (!zrole) scr[16 * k + c] = v[k];                    // S[k][c], unscaled
    __builtin_amdgcn_wave_barrier();
    const double piv = lane_bcast(v[k], k);
    const double t = v[k] * rcp_nr(piv);
    if (k + 1 < 16) {
      const double m1 = lane_bcast(v[k], k + 1);           // the next pivot's row first, without the LDS round trip
      v[k + 1] = fma(-m1, t, v[k + 1]);
    }
#pragma unroll
    for (int kp = k + 2; kp < 16; ++kp) v[kp] = fma(-scr[16 * k + kp], t, v[kp]);
  }
  __builtin_amdgcn_wave_barrier();
  const double pv = scr[16 * c + c];                       // p_c = S[c][c]
  const double rc = rsqrt_nr(pv);                          // NaN for a pivot <= 0 (or NaN): poisons row c of W, info says where
  {
    const unsigned long long bad = __ballot(!(pv > 0.0)) & 0xffffull;
    if (bad != 0 && pa.info == 0) pa.info = col0 + __ffsll((long long)bad);
    pa.mant *= row16_prod(__builtin_amdgcn_frexp_mant(pv));      // sixteen mantissas in [0.5, 1): no underflow
    pa.ex += row16_sum_i32(__builtin_amdgcn_frexp_exp(pv));
    pa.renorm();
  }
  if (Lout != nullptr) {
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) scr[256 + lane] = rc;
    __builtin_amdgcn_wave_barrier();
    if (lane < 16 && lane < nvalid) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < nvalid) Lout[(size_t)lane * ldl + i] = (i <= lane) ? v[i] * scr[256 + i] : 0.0;   // L[j][i] = U[i][j] = r_i S[i][j]
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane >= 16 && lane < 32) {
#pragma unroll
    for (int i = 0; i < 16; ++i) scr[i * DIAG_LD + c] = v[i];                       // unscaled Z rows: Zs[i][j], j = c
  }
  __builtin_amdgcn_wave_barrier();
  d4 w;
#pragma unroll
  for (int s = 0; s < 4; ++s) w[s] = scr[c * DIAG_LD + 4 * s + g] * rc;             // W[c][4s+g] = r_c Zs[c][4s+g]
  __builtin_amdgcn_wave_barrier();
#ifdef HGP_STAMPS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pa.diag_cycles += __builtin_readcyclecounter() - td0;
#endif
  return w;
}

// ---------------------------------------------------------------------------------------------
// diag16_acc (round 3, third session): the elimination IN the accumulator layout, all 64 lanes busy.  The other VALU forms hold one
// column per lane (16 + 16 lanes, sixteen registers, 15 multiplier broadcasts + 15 FMAs per step: ~950 instructions per block, which
// is what their ~4.2 k cycles are).  Here lane (g, c) keeps S[4r + g][c] in register r exactly as the MFMA left it - no staging
// through LDS on the way in - and step k is a rank-1 update of the whole tile in (4 - k/4) FMAs:
//   * the pivot p_k = S[k][k] goes through v_readlane (uniform, SGPR pair), 1/p_k by v_rcp_f64 + one third-order step;
//   * row k is copied to the four 16-lane rows of the wave by v_permlane16_swap + v_permlane32_swap (s_k[c] in every lane (., c));
//   * the multipliers S[i][k] of the rows below come from v_mov_dpp row_newbcast:k of the tile registers themselves (the Schur
//     complement stays symmetric, so column k IS the multiplier column); rows of the pivot's own register that are already done
//     get a zero multiplier through the DPP row mask;
//   * Z (= I at the start) takes the same row operations with the same multipliers: Z[i][.] -= (S[i][k] / p_k) Z[k][.];
//   * rows stay UNSCALED during the elimination (as in diag16_col): the sixteen 1/sqrt(p_k) are one rsqrt sequence at the end
//     (lane c holds p_c), the scaling of W = D^{-1/2} Z happens in the final transposition to the A-operand layout, the pivot
//     product / first bad pivot are one DPP reduction and one ballot.
// Finished rows are not protected: row k of S becomes ~0 after its step and only ever feeds itself again (dead rows and columns
// of the symmetric Schur complement never reach a live entry).  ~470 instructions per block, dependency chain ~80 cycles per pivot.
// Same interface as the other forms; results agree to rounding (multipliers S/p instead of (S/sqrt p)(S/sqrt p)).
// ---------------------------------------------------------------------------------------------
#ifndef HGP_DIAG_ROWCOPY
#define HGP_DIAG_ROWCOPY 1   // 1 (shipped since round 4) = the three row copies of a pivot step through LDS memory: one ds_write2_b64, one
                             // ds_read2_b64, one ds_read_b64 in the wave's scratch; 0 = six ds_bpermute_b32 (crossbar, no memory; round 3).
                             // tools/probe_diag16: 2 811 vs 3 088 cycles per block; in the kernels (tools/ab_rowcopy.sh) k_pairs<8> 1.004 ->
                             // 0.991 ms, k_pairs<6> 0.642 -> 0.616, cooph<16> 13.60 -> 13.37; results identical bit for bit
#endif
#ifndef HGP_DIAG_SCHED
#define HGP_DIAG_SCHED 0     // 0 = the compiler's schedule; 4 = the pinned order below.  (Three other pinned orders, built on the ds_bpermute
                             // form in round 4 - off-chain FMAs behind the broadcasts / crossbar instructions in one run / dealt two by
                             // two into the reciprocal's gaps - measured 3 004 / 3 144 / 3 388 cycles per block against 3 088 and made no
                             // difference inside the kernels: profiles/r04_ab_sched.txt; their code is gone.)
#endif
template <int G0>
__device__ __forceinline__ double row_to_all(double v) {   // row G0 (16 lanes) of v copied to all four rows of the wave
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // a[0] = rows (0,0,2,2), a[1] = rows (1,1,3,3)
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const unsigned lo1 = a[G0 & 1], hi1 = b[G0 & 1];
  const auto a2 = __builtin_amdgcn_permlane32_swap(lo1, lo1, false, false);   // [0] = lower half twice, [1] = upper half twice
  const auto b2 = __builtin_amdgcn_permlane32_swap(hi1, hi1, false, false);
  return __hiloint2double((int)b2[G0 >> 1], (int)a2[G0 >> 1]);
}
// the same through the LDS crossbar (ds_bpermute_b32 x 2, no LDS memory): two instructions that do not occupy the VALU, ~30 cycles
__device__ __forceinline__ double row_to_all_bperm(double v, int addr) {   // addr = 4 (16 G0 + c)
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// One elimination step on the registers of S and of Z:  X[r] += bcast_{4r}(mrot) * t  for the rows below pivot K.  mrot is row K
// of S copied to the four rows of the wave with row g rotated by g lanes, so that lane 4r of row g holds S[K][4r + g] - the
// multiplier of row 4r + g - and ONE DPP control (row_newbcast:4r inside v_fmac_f64_dpp) serves all four rows; the DPP row mask
// keeps the finished rows of the pivot's own register.  The multipliers are taken from ROW K (upper triangle), never from column
// K: rounding makes the two differ, and a first version that read column K lost two digits on ill-conditioned blocks
// (k_pairs_acc at length-scale 3: 1e-7 instead of 5e-10).  (s_nop 1: a DPP source written by the preceding VALU instruction
// needs two wait states, and the hazard recogniser does not look inside inline assembly.)
template <int RS, int RMF>   // register RS (the one that holds row K + 1: on the pivot chain); row mask RMF
__device__ __forceinline__ void elim_crit(double (&S)[4], double (&Z)[4], double mrot, double nt, double ntz) {
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:%6 bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%5 row_mask:%6 bank_mask:0xf"
      : "+v"(S[RS]), "+v"(Z[RS])
      : "v"(mrot), "v"(nt), "v"(ntz), "n"(4 * RS), "n"(RMF));
}
template <int RS>   // registers RS + 1 .. 3 (all rows): off the chain, issued behind the next pivot's broadcasts
__device__ __forceinline__ void elim_rest(double (&S)[4], double (&Z)[4], double mrot, double nt, double ntz) {
#define HGP_FD(dst, mul, lane) "v_fmac_f64_dpp %" #dst ", %6, %" #mul " row_newbcast:" #lane " row_mask:0xf bank_mask:0xf\n\t"
  if constexpr (RS == 0) {
    asm("s_nop 1\n\t" HGP_FD(0, 7, 4) HGP_FD(1, 7, 8) HGP_FD(2, 7, 12) HGP_FD(3, 8, 4) HGP_FD(4, 8, 8) HGP_FD(5, 8, 12)
        : "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(Z[1]), "+v"(Z[2]), "+v"(Z[3])
        : "v"(mrot), "v"(nt), "v"(ntz));
  } else if constexpr (RS == 1) {
    asm("s_nop 1\n\t" HGP_FD(1, 7, 8) HGP_FD(2, 7, 12) HGP_FD(4, 8, 8) HGP_FD(5, 8, 12)
        : "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(Z[1]), "+v"(Z[2]), "+v"(Z[3])
        : "v"(mrot), "v"(nt), "v"(ntz));
  } else if constexpr (RS == 2) {
    asm("s_nop 1\n\t" HGP_FD(2, 7, 12) HGP_FD(5, 8, 12)
        : "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(Z[1]), "+v"(Z[2]), "+v"(Z[3])
        : "v"(mrot), "v"(nt), "v"(ntz));
  }
#undef HGP_FD
}

// elim_rest one instruction at a time (I = 0 .. 5: S[RS+1], Z[RS+1], S[RS+2], Z[RS+2], ...; nothing beyond register 3), so that a
// pinned order can place each where the pivot chain waits.  No s_nop: the DPP source (mrot) comes from an LDS read, never from
// the VALU instruction in front (tools/check_dpp_hazard.py looks at the compiled stream).
template <int RS, int I>
__device__ __forceinline__ void rest_one(double (&S)[4], double (&Z)[4], double mrot, double nt, double ntz) {
  constexpr int reg = RS + 1 + I / 2;
  if constexpr (reg <= 3) {
    if constexpr (I % 2 == 0)
      asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(S[reg]) : "v"(mrot), "v"(nt), "n"(4 * reg));
    else
      asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(Z[reg]) : "v"(mrot), "v"(ntz), "n"(4 * reg));
  }
}

struct NoBg {
  template <class KC>
  __device__ __forceinline__ void operator()(KC) const {}
};
// bg(integral_constant<int, k>), k = 0 .. 14: work of the CALLER that does not depend on this block (wave_factor hands in the
// trailing-update MFMAs of the previous block step), issued once per pivot step where the pivot chain waits for its LDS round trip.
template <class BG = NoBg>
__device__ __forceinline__ d4 diag16_acc(const d4& X, double* scr, int lane, PivotAcc& pa, int col0,
                                         double* Lout, int ldl, int nvalid, BG&& bg = BG{}) {
  const int g = lane >> 4, c = lane & 15;
#ifdef HGP_EXP_NODIAG   // in-situ knock-out experiment (diagnostic builds only)
  {
    d4 w;
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = (4 * s + g == c) ? 1.0 / sqrt(fabs(X[s]) + 1.0) : 0.0;
    pa.mant *= 1.0 + 1e-300 * X[0];
    return w;
  }
#endif
#ifdef HGP_STAMPS
  const unsigned long long td0 = __builtin_readcyclecounter();
#endif
  double S[4], Z[4];
  d4 Us;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    S[r] = X[r];
    Z[r] = (4 * r + g == c) ? 1.0 : 0.0;
    Us[r] = 0.0;
  }
  int pvlo = 0, pvhi = 0;   // p_k in lane k (row 0 of the wave), collected with v_writelane
  // software pipeline: the pivot and the row copies of step k + 1 are requested right behind the two FMAs that finish row
  // k + 1 (elim_crit); the other FMAs of step k (elim_rest) issue under their latency
  const int crot = (c + g) & 15;
  double p = lane_bcast(S[0], 0);
  double sk = row_to_all_bperm(S[0], 4 * c), mrot = row_to_all_bperm(S[0], 4 * crot), zk = row_to_all_bperm(Z[0], 4 * c);
#if HGP_DIAG_SCHED == 4
  // The step's instruction ORDER is pinned (a wave issues in order; every statement below is fenced by a scheduling barrier):
  // behind the two FMAs that finish row k + 1 come its LDS copy and the pivot read, then v_rcp_f64 at once, and the off-chain FMAs of
  // step k (one instruction each, rest_one) are dealt into the latency gaps of the reciprocal's dependent chain (v_rcp 21 cycles,
  // each dependent f64 operation ~9.5, tools/probe_lat.hip) instead of sitting in one run in front of it.
#define HGP_SB __builtin_amdgcn_sched_barrier(0)
  asm("v_writelane_b32 %0, %1, %2" : "+v"(pvlo) : "s"(__double2loint(p)), "n"(0));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(pvhi) : "s"(__double2hiint(p)), "n"(0));
  double rp = rcp_nr(p);
  static_for<0, 15>([&](auto kc) {
    constexpr int k = decltype(kc)::value, r0 = k >> 2, g0 = k & 3;
    constexpr int RS = (g0 == 3) ? r0 + 1 : r0;                          // register of row k + 1
    constexpr int RMF = (g0 == 3) ? 0xf : (0xf << (g0 + 1)) & 0xf;      // its rows below the pivot
    if (Lout != nullptr) Us[r0] = (g == g0) ? sk : Us[r0];
    const double nt = -(sk * rp), ntz = -(zk * rp), mr = mrot;
    HGP_SB;
    elim_crit<RS, RMF>(S, Z, mr, nt, ntz);
    HGP_SB;
    constexpr int k1 = k + 1, g1 = k1 & 3;
    if constexpr (k1 < 15) {
      scr[lane] = S[RS];
      scr[64 + lane] = Z[RS];
      __builtin_amdgcn_wave_barrier();
    }
    p = lane_bcast(S[RS], 16 * g1 + k1);
    HGP_SB;
    if constexpr (k1 < 15) {
      sk = scr[16 * g1 + c];
      zk = scr[64 + 16 * g1 + c];
      mrot = scr[16 * g1 + crot];
      __builtin_amdgcn_wave_barrier();
      HGP_SB;
      const double y = __builtin_amdgcn_rcp(p);
      HGP_SB;
      bg(kc);
      HGP_SB;
      rest_one<RS, 0>(S, Z, mr, nt, ntz);
      rest_one<RS, 1>(S, Z, mr, nt, ntz);
      HGP_SB;
      const double e = fma(-p, y, 1.0);
      HGP_SB;
      rest_one<RS, 2>(S, Z, mr, nt, ntz);
      HGP_SB;
      const double ye = y * e, t = fma(y, e, y);
      HGP_SB;
      rest_one<RS, 3>(S, Z, mr, nt, ntz);
      asm("v_writelane_b32 %0, %1, %2" : "+v"(pvlo) : "s"(__double2loint(p)), "n"(k1));
      asm("v_writelane_b32 %0, %1, %2" : "+v"(pvhi) : "s"(__double2hiint(p)), "n"(k1));
      HGP_SB;
      rp = fma(ye, e, t);
      HGP_SB;
      rest_one<RS, 4>(S, Z, mr, nt, ntz);
      rest_one<RS, 5>(S, Z, mr, nt, ntz);
      HGP_SB;
    } else {
      asm("v_writelane_b32 %0, %1, %2" : "+v"(pvlo) : "s"(__double2loint(p)), "n"(k1));
      asm("v_writelane_b32 %0, %1, %2" : "+v"(pvhi) : "s"(__double2hiint(p)), "n"(k1));
      bg(kc);
      elim_rest<RS>(S, Z, mr, nt, ntz);
    }
  });
#undef HGP_SB
#else
  static_for<0, 15>([&](auto kc) {
    constexpr int k = decltype(kc)::value, r0 = k >> 2, g0 = k & 3;
    constexpr int RS = (g0 == 3) ? r0 + 1 : r0;                          // register of row k + 1
    constexpr int RMF = (g0 == 3) ? 0xf : (0xf << (g0 + 1)) & 0xf;      // its rows below the pivot
    asm("v_writelane_b32 %0, %1, %2" : "+v"(pvlo) : "s"(__double2loint(p)), "n"(k));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(pvhi) : "s"(__double2hiint(p)), "n"(k));
    const double rp = rcp_nr(p);
    if (Lout != nullptr) Us[r0] = (g == g0) ? sk : Us[r0];
    const double nt = -(sk * rp), ntz = -(zk * rp), mr = mrot;
    elim_crit<RS, RMF>(S, Z, mr, nt, ntz);
    constexpr int k1 = k + 1, g1 = k1 & 3;
    p = lane_bcast(S[RS], 16 * g1 + k1);
    if constexpr (k1 < 15) {
#if HGP_DIAG_ROWCOPY == 1
      // the row copies through LDS memory instead of the crossbar: one ds_write2_b64 + one ds_read2_b64 + one ds_read_b64 (64-bit)
      // in place of six ds_bpermute_b32 - three instructions fewer per pivot, at 77 instead of ~25 cycles of latency
      scr[lane] = S[RS];
      scr[64 + lane] = Z[RS];
      __builtin_amdgcn_wave_barrier();
      sk = scr[16 * g1 + c];
      zk = scr[64 + 16 * g1 + c];
      mrot = scr[16 * g1 + crot];
      __builtin_amdgcn_wave_barrier();
#else
      sk = row_to_all_bperm(S[RS], 4 * (16 * g1 + c));
      mrot = row_to_all_bperm(S[RS], 4 * (16 * g1 + crot));
      zk = row_to_all_bperm(Z[RS], 4 * (16 * g1 + c));
#endif
    }
    bg(kc);
    elim_rest<RS>(S, Z, mr, nt, ntz);
  });
  asm("v_writelane_b32 %0, %1, %2" : "+v"(pvlo) : "s"(__double2loint(p)), "n"(15));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(pvhi) : "s"(__double2hiint(p)), "n"(15));
#endif
  if (Lout != nullptr) Us[3] = (g == 3) ? S[3] : Us[3];
  // the pivots: lane c of row 0 holds p_c; every row needs it for the scaling of its part of W
  const double pv = row_to_all<0>(__hiloint2double(pvhi, pvlo));
  const double rc = rsqrt_nr(pv);                          // NaN for a pivot <= 0 (or NaN): poisons row c of W, info says where
  {
    const unsigned long long bad = __ballot(!(pv > 0.0)) & 0xffffull;
    if (bad != 0 && pa.info == 0) pa.info = col0 + __ffsll((long long)bad);
    pa.mant *= row16_prod(__builtin_amdgcn_frexp_mant(pv));      // sixteen mantissas in [0.5, 1): no underflow
    pa.ex += row16_sum_i32(__builtin_amdgcn_frexp_exp(pv));
    pa.renorm();
  }
  if (Lout != nullptr) {   // L[j][i] = U[i][j] = r_i S_i[j] (row i as it stood at its own step)
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(4 * r + g) * DIAG_LD + c] = Us[r];
    if (g == 0) scr[c * DIAG_LD + 16] = rc;                  // padding column of the staging tile
    __builtin_amdgcn_wave_barrier();
    if (lane < 16 && lane < nvalid) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < nvalid) Lout[(size_t)lane * ldl + i] = (i <= lane) ? scr[i * DIAG_LD + lane] * scr[i * DIAG_LD + 16] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) scr[(4 * r + g) * DIAG_LD + c] = Z[r];   // unscaled Z rows: Zs[i][j], i = 4r+g, j = c
  __builtin_amdgcn_wave_barrier();
  d4 w;
#pragma unroll
  for (int s = 0; s < 4; ++s) w[s] = scr[c * DIAG_LD + 4 * s + g] * rc;             // W[c][4s+g] = r_c Zs[c][4s+g]
  __builtin_amdgcn_wave_barrier();
#ifdef HGP_STAMPS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pa.diag_cycles += __builtin_readcyclecounter() - td0;
#endif
  return w;
}

// Which 16 x 16 diagonal-block routine the factorisations use (A/B builds: -DHGP_DIAG_IMPL=0 restores rounds 1-2:
// diag16_valu where the register budget asks for it, the MFMA-blocked diag16 elsewhere).
#ifndef HGP_DIAG_IMPL
#define HGP_DIAG_IMPL 3   // 3 = diag16_acc everywhere (shipped: k_pairs<8> 1.265 -> 1.121 ms, k_pairs<6> 0.757 -> 0.677, k_pairs_cooph<16> 7.39 -> 6.91 per 16 384 pairs);
                          // 0 = rounds 1-2 (diag16_valu / MFMA-blocked diag16); 1 / 2 = diag16_col (measured SLOWER in the kernels: see diag16_col); 4 = diag16_acc for the VALU users only
#endif
template <bool VALU, class BG = NoBg>
__device__ __forceinline__ d4 diag16_sel(const d4& X, double* scr, int lane, PivotAcc& pa, int col0, double* Lout, int ldl,
                                         int nvalid, BG&& bg = BG{}) {
#if HGP_DIAG_IMPL == 3
  return diag16_acc(X, scr, lane, pa, col0, Lout, ldl, nvalid, bg);
#elif HGP_DIAG_IMPL == 4
  return VALU ? diag16_acc(X, scr, lane, pa, col0, Lout, ldl, nvalid) : diag16(X, scr, lane, pa, col0, Lout, ldl, nvalid);
#elif HGP_DIAG_IMPL == 2
  return diag16_col(X, scr, lane, pa, col0, Lout, ldl, nvalid);
#elif HGP_DIAG_IMPL == 1
  return VALU ? diag16_col(X, scr, lane, pa, col0, Lout, ldl, nvalid) : diag16(X, scr, lane, pa, col0, Lout, ldl, nvalid);
#else
  return VALU ? diag16_valu(X, scr, lane, pa, col0, Lout, ldl, nvalid) : diag16(X, scr, lane, pa, col0, Lout, ldl, nvalid);
#endif
}

// ---------------------------------------------------------------------------------------------
// Upper-form blocked Cholesky of an NB x NB tile matrix held in registers (upper tiles only),
// right-looking, with ONE block column of 16 right-hand sides eliminated in the same sweep:
//   on exit R[K] = Z_K with L Z = R_in (L = U^T), so  rhs^T A^{-1} rhs = column sums of Z.^2 .
// The inverses of the diagonal factors, W_K = U_KK^{-T} (A-operand layout), are transient unless
// Wlds != nullptr, in which case they are kept in LDS ([K][s][lane]) for later wave_fwd_solve calls.
// If Lout != nullptr the lower factor L = U^T is written row-major (ld = ldl) for rows/cols < n.
// ---------------------------------------------------------------------------------------------
// RHSMODE: 0 = none; 1 = one block column of 16 right-hand sides as MFMA tiles R[K]; 2 = ONE right-hand side kept
// as a vector in LDS (dvec[16 NB], per wave) and eliminated on the VALU next to the MFMA stream: on exit dvec
// holds z = L^{-1} d and the return value is z^T z (valid in every lane).
#ifndef HGP_LOOKAHEAD
#define HGP_LOOKAHEAD 1   // shipped since round 4 (tools/ab_look.sh: k_pairs<8> 0.998 -> 0.980 ms, k_pairs<6> 0.625 -> 0.612; bit-identical results); 0 = trailing update in one run
#endif
// The trailing-update MFMAs of block step Kp that wave_factor hands to diag16_acc of block Kp + 1 (look-ahead): every tile
// (I, J), Kp < I <= J < NB, except (Kp + 1, Kp + 1) itself, which the next diagonal block needs first.  Operation q of the 4 NT:
// k-step q / NT of tile q % NT (consecutive operations go to different accumulators); tiles in row-major order.
template <int NB, int Kp>
struct Pending {
  static constexpr int M = NB - 1 - Kp, NT = M * (M + 1) / 2 - 1, N = 4 * NT, CH = (N + 14) / 15;
  static constexpr int row(int t) {
    int u = t + 1, I = Kp + 1;                 // u: index among all tiles of the trailing matrix, (Kp+1, Kp+1) being 0
    while (u >= NB - I) { u -= NB - I; ++I; }
    return I;
  }
  static constexpr int col(int t) {
    int u = t + 1, I = Kp + 1;
    while (u >= NB - I) { u -= NB - I; ++I; }
    return I + u;
  }
};
template <int NB, int RHSMODE, bool DIAG_VALU = false, bool RHS_DEFER = false, bool LOOK = (HGP_LOOKAHEAD != 0)>
__device__ __forceinline__ double wave_factor(d4 (&U)[NB * (NB + 1) / 2], d4 (&R)[NB], double* scr, double* Wlds,
                                              double* dvec, int lane_in, PivotAcc& pa, double* Lout, int ldl, int n) {
  constexpr bool RHS = (RHSMODE == 1);
  double zq = 0.0;
  // RHSMODE 2: what the steps K' < K have to subtract from d_K is kept per lane as the partial sum over the lane's own rows
  // (dp[K], lane (g, c): rows g + 4r of the tiles U_K'K) and reduced over the four 16-lane rows ONCE, when block K becomes the
  // pivot block - one cross-row sum and one LDS update per block instead of one per (K', K) pair (28 -> 8 at NB = 8).
  double dp[NB];
#pragma unroll
  for (int K = 0; K < NB; ++K) dp[K] = 0.0;
  static_for<0, NB>([&](auto Kc) {
    constexpr int K = decltype(Kc)::value;
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    double* Ld = (Lout != nullptr) ? Lout + (size_t)(16 * K) * ldl + 16 * K : nullptr;
    d4 W;
    if constexpr (LOOK && K > 0 && K + 1 < NB) {
      // look-ahead: the pivot chain of the diagonal block leaves the f64 pipe idle while it waits for its LDS round trips
      // (~130 of ~165 cycles per pivot); the trailing-update MFMAs of step K - 1 that this block does not depend on fill it
      using P = Pending<NB, K - 1>;
      W = diag16_sel<DIAG_VALU>(U[tix(K, K, NB)], scr, lane, pa, 16 * K, Ld, ldl, n - 16 * K, [&](auto kc) {
        constexpr int k = decltype(kc)::value, q0 = k * P::CH, q1 = (q0 + P::CH < P::N) ? q0 + P::CH : P::N;
        static_for<q0, q1>([&](auto qc) {
          constexpr int q = decltype(qc)::value, s = q / P::NT, t = q % P::NT, I = P::row(t), J = P::col(t);
          U[tix(I, J, NB)] = mfma_sub(U[tix(K - 1, I, NB)][s], U[tix(K - 1, J, NB)][s], U[tix(I, J, NB)]);
        });
      });
    } else {
      W = diag16_sel<DIAG_VALU>(U[tix(K, K, NB)], scr, lane, pa, 16 * K, Ld, ldl, n - 16 * K);
    }
    if (Wlds != nullptr) {
#pragma unroll
      for (int s = 0; s < 4; ++s) Wlds[(K * 4 + s) * 64 + lane] = W[s];
    }
    if (RHSMODE == 2) {   // z_K = W d_K : lane (g,c) sums W[c][4s+g] d[4s+g] over s, rows are summed over g
      if (RHS_DEFER && K > 0) {   // d_K -= sum over K' < K of U_K'K^T z_K' (deferred, see above)
        const double q = xrow_sum(dp[K]);
        if (g == 0) dvec[16 * K + c] -= q;
        __builtin_amdgcn_wave_barrier();
      }
      double p = 0.0;
#pragma unroll
      for (int s = 0; s < 4; ++s) p = fma(W[s], dvec[16 * K + 4 * s + g], p);
      p = xrow_sum(p);
      __builtin_amdgcn_wave_barrier();
      if (g == 0) {
        dvec[16 * K + c] = p;
        zq = fma(p, p, zq);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // panel: U_KJ = U_KK^{-T} A_KJ = W * A_KJ  for J > K, and the rhs tile (J == NB); two tiles in flight
#pragma unroll
    for (int J0 = K + 1; J0 <= NB; J0 += 2) {
      d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
      const int J1 = J0 + 1;
      const bool has0 = (J0 < NB) || RHS, has1 = (J1 < NB) || (RHS && J1 == NB);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (has0) acc0 = mfma(W[s], (J0 < NB) ? U[tix(K, J0 < NB ? J0 : K, NB)][s] : R[K][s], acc0);
        if (has1) acc1 = mfma(W[s], (J1 < NB) ? U[tix(K, J1 < NB ? J1 : K, NB)][s] : R[K][s], acc1);
      }
      if (J0 < NB) U[tix(K, J0 < NB ? J0 : K, NB)] = acc0;
      else if (RHS) R[K] = acc0;
      if (J1 < NB) U[tix(K, J1 < NB ? J1 : K, NB)] = acc1;
      else if (RHS && J1 == NB) R[K] = acc1;
    }
    if (Lout != nullptr) {
#pragma unroll
      for (int J = K + 1; J < NB; ++J) {
        // L[16J + c][16K + g + 4r] = U_KJ[g + 4r][c]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = 16 * J + c, col = 16 * K + g + 4 * r;
          if (row < n && col < n) Lout[(size_t)row * ldl + col] = U[tix(K, J, NB)][r];
        }
      }
    }
    if (RHSMODE == 2 && K + 1 < NB) {   // d_I -= U_KI^T z_K for I > K
      double zr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) zr[r] = dvec[16 * K + g + 4 * r];
      if (RHS_DEFER) {
#pragma unroll
        for (int I = K + 1; I < NB; ++I) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[I] = fma(U[tix(K, I, NB)][r], zr[r], dp[I]);
        }
      } else {   // reduced and subtracted at once (the mask-driven k_pairs<8, false> has no registers for the partial sums)
#pragma unroll
        for (int I = K + 1; I < NB; ++I) {
          double q = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) q = fma(U[tix(K, I, NB)][r], zr[r], q);
          q = xrow_sum(q);
          if (g == 0) dvec[16 * I + c] -= q;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // trailing update: A_IJ -= U_KI^T U_KJ  for K < I <= J  (and the rhs tiles I > K).  No look-ahead: on gfx950 the
    // f64 MFMA and f64 VALU share the DP pipe (tools/probe_coexec.hip: 1 MFMA + 12 independent v_fma_f64 = 64 + 64 clk),
    // so deferring these MFMAs into the VALU stream of the next diagonal block gains nothing (tried, measured).
    // With LOOK only the next diagonal tile (and the rhs tiles) are updated here; the rest rides in the next diag16_acc (above).
#pragma unroll
    for (int I = K + 1; I < NB; ++I) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int J = I; J < NB; ++J) {
          if (LOOK && K + 2 < NB && !(I == K + 1 && J == K + 1)) continue;
          U[tix(I, J, NB)] = mfma_sub(U[tix(K, I, NB)][s], U[tix(K, J, NB)][s], U[tix(I, J, NB)]);
        }
        if (RHS) R[I] = mfma_sub(U[tix(K, I, NB)][s], R[K][s], R[I]);
      }
    }
  });
  return (RHSMODE == 2) ? wave_sum(zq) : 0.0;
}

// Forward substitution  L Z = R  (L = U^T) on one more block column of 16 right-hand sides, with the
// stored factor: U tiles in registers, W_K in LDS (written by wave_factor).  Tiles K < K0 of the
// right-hand side are taken as zero (used for the columns of L^{-1}).
template <int NB>
__device__ __forceinline__ void wave_fwd_solve(const d4 (&U)[NB * (NB + 1) / 2], const double* Wlds, d4 (&R)[NB],
                                               int lane, int K0 = 0) {
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    if (K < K0) continue;
    d4 t = R[K];
#pragma unroll
    for (int I = 0; I < K; ++I) {
      if (I < K0) continue;
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma_sub(U[tix(I, K, NB)][s], R[I][s], t);     // t -= U_IK^T Z_I
    }
    d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) z = mfma(Wlds[(K * 4 + s) * 64 + lane], t[s], z);   // Z_K = W_K t
    R[K] = z;
  }
}

// sum of squares of every rhs column: returns in each lane the value for ITS column c (all g agree)
template <int NB>
__device__ __forceinline__ double wave_colnorm2(const d4 (&Z)[NB]) {
  double s = 0.0;
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) s = fma(Z[K][r], Z[K][r], s);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// ---------------------------------------------------------------------------------------------
// Loaders.  A is row-major with leading dimension ld, logical size n x n (n <= 16 NB); rows/cols
// >= n are padded with the identity so the padded factorisation leaves quad/logdet unchanged.
// ---------------------------------------------------------------------------------------------
// symmetrised upper tiles:  0.5 (A + A^T).  Every global access is coalesced: tile (I, J) AND tile (J, I) are read in
// their natural row-major order (16 lanes = 128 contiguous bytes) and the second is transposed through the wave's
// 16 x 18 LDS staging tile.  (The first version read A[j][i] directly: 16 cache lines per load instruction; PMC showed
// 56-65 % of the wave cycles of k_wave_inv / k_wave_score1 in s_waitcnt.)  One block row at a time: its 2 (NB - I) - 1
// tile loads are all in flight before the first transpose.
template <int NB>
__device__ __forceinline__ void load_sym_upper(d4 (&U)[NB * (NB + 1) / 2], const double* __restrict__ A, int ld, int n,
                                               int lane_in, double* scr) {
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    d4 nat[NB], trn[NB];
#pragma unroll
    for (int J = I; J < NB; ++J) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;          // natural element of tile (I, J)
        nat[J][r] = (i < n && j < n) ? A[(size_t)i * ld + j] : 0.0;
        const int i2 = 16 * J + g + 4 * r, j2 = 16 * I + c;        // natural element of tile (J, I)
        trn[J][r] = (J > I && i2 < n && j2 < n) ? A[(size_t)i2 * ld + j2] : 0.0;
      }
    }
#pragma unroll
    for (int J = I; J < NB; ++J) {
      const d4 src = (J == I) ? nat[I] : trn[J];
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[(g + 4 * r) * DIAG_LD + c] = src[r];
      __builtin_amdgcn_wave_barrier();
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;
        const double t = scr[c * DIAG_LD + g + 4 * r];               // element (c, g + 4r) of src = A[j][i]
        double x = 0.5 * (nat[J][r] + t);
        if (!(i < n && j < n)) x = (i == j) ? 1.0 : 0.0;
        v[r] = x;
      }
      __builtin_amdgcn_wave_barrier();
      U[tix(I, J, NB)] = v;
    }
  }
}

// load_sym_upper with ALL global loads in flight at once (the upper tiles land in U, the NB (NB - 1) / 2 transposed partners
// in temporaries: 36 tiles = 288 VGPRs at NB = 6), then the LDS transposes: ONE exposed load latency instead of NB.
// For the latency-bound one-wave-per-SIMD kernels at NB <= 6.  Same result as load_sym_upper, bit for bit.
template <int NB>
__device__ __forceinline__ void load_sym_upper_burst(d4 (&U)[NB * (NB + 1) / 2], const double* __restrict__ A, int ld, int n,
                                                     int lane_in, double* scr) {
  constexpr int NO = NB > 1 ? NB * (NB - 1) / 2 : 1;
  d4 trn[NO];
  {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
      for (int J = I; J < NB; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + g + 4 * r, j = 16 * J + c;          // natural element of tile (I, J)
          U[tix(I, J, NB)][r] = (i < n && j < n) ? A[(size_t)i * ld + j] : 0.0;
          if (J > I) {
            const int i2 = 16 * J + g + 4 * r, j2 = 16 * I + c;      // natural element of tile (J, I)
            trn[tix(I, J, NB) - (I + 1)][r] = (i2 < n && j2 < n) ? A[(size_t)i2 * ld + j2] : 0.0;
          }
        }
  }
#pragma unroll
  for (int I = 0; I < NB; ++I) {
#pragma unroll
    for (int J = I; J < NB; ++J) {
      const int lane = launder(lane_in);
      const int g = lane >> 4, c = lane & 15;
      const d4 nat = U[tix(I, J, NB)];
      const d4 src = (J == I) ? nat : trn[(J == I) ? 0 : tix(I, J, NB) - (I + 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[(g + 4 * r) * DIAG_LD + c] = src[r];
      __builtin_amdgcn_wave_barrier();
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;
        const double t = scr[c * DIAG_LD + g + 4 * r];               // element (c, g + 4r) of src = A[j][i]
        double x = 0.5 * (nat[r] + t);
        if (!(i < n && j < n)) x = (i == j) ? 1.0 : 0.0;
        v[r] = x;
      }
      __builtin_amdgcn_wave_barrier();
      U[tix(I, J, NB)] = v;
    }
  }
}

// upper tiles of a matrix the caller guarantees to be symmetric: only the tiles I <= J are read (about half the
// bytes), every access coalesced (no transposed partner read)
template <int NB>
__device__ __forceinline__ void load_upper_only(d4 (&U)[NB * (NB + 1) / 2], const double* __restrict__ A, int ld, int n,
                                                int lane_in) {
#pragma unroll
  for (int I = 0; I < NB; ++I) {
#pragma unroll
    for (int J = I; J < NB; ++J) {
      const int lane = launder(lane_in);
      const int g = lane >> 4, c = lane & 15;
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = 16 * I + g + 4 * r, j = 16 * J + c;
        double x = 0.0;
        if (i < n && j < n) x = A[(size_t)i * ld + j];
        else if (i == j) x = 1.0;
        v[r] = x;
      }
      U[tix(I, J, NB)] = v;
    }
  }
}

// add `shift` to the diagonal entries i < n (padded rows keep their 1)
template <int NB>
__device__ __forceinline__ void add_diag(d4 (&U)[NB * (NB + 1) / 2], double shift, int n, int lane_in) {
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i = 16 * I + g + 4 * r;
      if (g + 4 * r == c && i < n) U[tix(I, I, NB)][r] += shift;
    }
  }
}

// mean over i < n of |A_ii + shift| taken from the diagonal tiles
template <int NB>
__device__ __forceinline__ double diag_abs_mean(const d4 (&U)[NB * (NB + 1) / 2], int n, int lane_in, double shift = 0.0) {
  double s = 0.0;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int i = 16 * I + g + 4 * r;
      if (g + 4 * r == c && i < n) s += fabs(U[tix(I, I, NB)][r] + shift);
    }
  }
  return wave_sum(s) / (double)n;
}

// =============================================================================================
// 4-wave cooperative version for 128 < T <= 256 (NB <= 16 tiles): the upper tiles are dealt
// column-wise to the 4 waves of a workgroup (tile (I, J) lives in wave Coop::owner(J), snake order), 40 tiles =
// 320 VGPR per wave at NB = 16.  Step K: the owner of column K factors the diagonal block and
// publishes W through LDS; every wave solves the row-K tiles of its own columns and publishes them
// (rowbuf); every wave updates its own trailing tiles reading U_KI from rowbuf.  Two workgroup
// barriers per step.  Right-hand sides (one block column of 16) live in LDS (Rbuf) and are updated
// by the waves round-robin.
// =============================================================================================
template <int NB>
struct Coop {
  static_assert(NB % 4 == 0 && NB <= 16, "cooperative factor: NB in {4,8,12,16}");
  static constexpr int NQ = NB / 4;
  static constexpr int NT = 2 * NQ * (NQ + 1);   // tiles per wave (shape of the wave that owns the most)
  __host__ __device__ static constexpr int loc(int I, int q) { return 2 * q * (q + 1) + I; }   // tile (I, col(q, wave))
  // Block columns are dealt to the waves in SNAKE order (0 1 2 3 | 3 2 1 0 | 0 1 2 3 ...): column J costs J + 1 tiles,
  // so plain cyclic dealing gives the last wave 40 tiles and the first 28 at NB = 16; the snake gives every wave 34,
  // and the trailing updates of each elimination step are balanced the same way.
  __host__ __device__ static constexpr int col(int q, int wave) { return 4 * q + ((q & 1) ? 3 - wave : wave); }
  __host__ __device__ static constexpr int owner(int J) { return ((J >> 2) & 1) ? 3 - (J & 3) : (J & 3); }
  static constexpr int LDS_DOUBLES = NB * 256 /*rowbuf*/ + NB * 256 /*Rbuf*/ + 256 /*Wbuf*/ + DIAG_SCR + 16;
};

__device__ __forceinline__ d4 lds_tile_load(const double* buf, int tile, int lane) {
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = buf[(tile * 4 + r) * 64 + lane];
  return v;
}
__device__ __forceinline__ void lds_tile_store(double* buf, int tile, int lane, const d4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) buf[(tile * 4 + r) * 64 + lane] = v[r];
}

template <int NB>
__device__ __forceinline__ void coop_load_sym_upper(d4 (&U)[Coop<NB>::NT], const double* __restrict__ A, int ld, int n,
                                                    int wave, int lane_in, double* scr_w) {
  // scr_w: a 16 x 18 LDS staging tile private to this wave.  As in load_sym_upper both triangles are read coalesced
  // and the lower one is transposed through LDS; a block column (the tiles this wave owns) at a time.
#pragma unroll
  for (int q = 0; q < Coop<NB>::NQ; ++q) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    const int J = Coop<NB>::col(q, wave);
    d4 nat[NB], trn[NB];
#pragma unroll
    for (int I = 0; I < 4 * q + 4; ++I) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;          // natural element of tile (I, J)
        nat[I][r] = (I <= J && i < n && j < n) ? A[(size_t)i * ld + j] : 0.0;
        const int i2 = 16 * J + g + 4 * r, j2 = 16 * I + c;        // natural element of tile (J, I)
        trn[I][r] = (I < J && i2 < n && j2 < n) ? A[(size_t)i2 * ld + j2] : 0.0;
      }
    }
#pragma unroll
    for (int I = 0; I < 4 * q + 4; ++I) {
      const d4 src = (I == J) ? nat[I] : trn[I];
#pragma unroll
      for (int r = 0; r < 4; ++r) scr_w[(g + 4 * r) * DIAG_LD + c] = src[r];
      __builtin_amdgcn_wave_barrier();
      d4 v = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;
        const double t = scr_w[c * DIAG_LD + g + 4 * r];
        double x = 0.5 * (nat[I][r] + t);
        if (!(i < n && j < n)) x = (i == j) ? 1.0 : 0.0;
        if (I <= J) v[r] = x;
      }
      __builtin_amdgcn_wave_barrier();
      U[Coop<NB>::loc(I, q)] = v;
    }
  }
}

// sum over i < n of |A_ii + shift| / n, from the diagonal tiles spread over the waves (LDS reduction in red[4])
template <int NB>
__device__ __forceinline__ double coop_diag_abs_mean(const d4 (&U)[Coop<NB>::NT], int n, int wave, int lane_in, double shift,
                                                     double* red) {
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < Coop<NB>::NQ; ++q) {
#pragma unroll
    for (int I = 4 * q; I < 4 * q + 4; ++I) {
      const int lane = launder(lane_in);
      const int g = lane >> 4, c = lane & 15;
      if (I == Coop<NB>::col(q, wave)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + g + 4 * r;
          if (g + 4 * r == c && i < n) s += fabs(U[Coop<NB>::loc(I, q)][r] + shift);
        }
      }
    }
  }
  s = wave_sum(s);
  if (lane_in == 0) red[wave] = s;
  __syncthreads();
  const double tot = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return tot / (double)n;
}

template <int NB>
__device__ __forceinline__ void coop_add_diag(d4 (&U)[Coop<NB>::NT], double shift, int n, int wave, int lane_in) {
#pragma unroll
  for (int q = 0; q < Coop<NB>::NQ; ++q) {
#pragma unroll
    for (int I = 4 * q; I < 4 * q + 4; ++I) {
      const int lane = launder(lane_in);
      const int g = lane >> 4, c = lane & 15;
      if (I == Coop<NB>::col(q, wave)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + g + 4 * r;
          if (g + 4 * r == c && i < n) U[Coop<NB>::loc(I, q)][r] += shift;
        }
      }
    }
  }
}

// Factor (and eliminate the 16 right-hand sides held as tiles in Rbuf when RHS).  On exit Rbuf holds Z = L^{-1} R.
// pa accumulates only the pivots of the columns this wave owns; combine with coop_logdet_info.
// RHSMODE: 0 none; 1 = 16 right-hand sides as tiles in Rbuf; 2 = ONE right-hand side as a vector dvec[16 NB] in LDS,
// eliminated on the VALU by the wave that owns the matching block column (on exit dvec = z = L^{-1} d and the return
// value is this wave's share of z^T z: sum the four waves' values).
template <int NB, int RHSMODE>
__device__ __forceinline__ double coop_factor(d4 (&U)[Coop<NB>::NT], double* rowbuf, double* Rbuf, double* Wbuf, double* scr,
                                              int wave, int lane_in, PivotAcc& pa, double* Lout, int ldl, int n,
                                              double* dvec = nullptr, double* Wout = nullptr, int ldw = 0,
                                              double* Lpack = nullptr, double* Wpack = nullptr, int kstop = NB) {
  // Lpack / Wpack (optional): the factor in MFMA OPERAND order for a consumer that streams it (hgp_matlik_coop.hip) - the
  // accumulator tile of U_KJ (K < J) IS the A operand of L[J, K] = U_KJ^T, so it is stored as it stands, 32 bytes per lane, at
  // tile index J (J - 1) / 2 + K; the inverses W_K of the diagonal blocks (A-operand order already) at tile index K of Wpack.
  // Wout (optional): the inverses W_K = L_KK^{-1} of the diagonal blocks go to the diagonal blocks of this [n, ldw] matrix -
  // they ARE the diagonal blocks of L^{-1}; k_trtri (hgp_kernels.hip) fills in the rest from L.
  // kstop (optional, the same in every thread): block steps K >= kstop are not taken - for callers whose blocks from kstop on are
  // identity padding (their factor is the identity, nothing of it is read) and who only want the right-hand side / the pivots.
  using C = Coop<NB>;
  constexpr bool RHS = (RHSMODE == 1);
  double zq = 0.0;
#ifdef HGP_STAMPS
  unsigned long long cf_t = __builtin_readcyclecounter();
#define HGP_CF(i) do { unsigned long long n_ = __builtin_readcyclecounter(); pa.cf[i] += n_ - cf_t; cf_t = n_; } while (0)
#else
#define HGP_CF(i)
#endif
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    if (K >= kstop) break;
    const int qK = K / 4, wK = C::owner(K);
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    if (wave == wK) {
      double* Ld = (Lout != nullptr) ? Lout + (size_t)(16 * K) * ldl + 16 * K : nullptr;
      const d4 Wd = diag16_sel<false>(U[C::loc(K, qK)], scr, lane, pa, 16 * K, Ld, ldl, n - 16 * K);
#pragma unroll
      for (int s = 0; s < 4; ++s) Wbuf[s * 64 + lane] = Wd[s];
      if (Wout != nullptr) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int row = 16 * K + c, col = 16 * K + 4 * s + g;      // Wd[s] = W[c][4 s + g]
          if (row < n && col < n) Wout[(size_t)row * ldw + col] = Wd[s];
        }
      }
      if (Wpack != nullptr) *reinterpret_cast<d4*>(Wpack + ((size_t)K * 64 + lane) * 4) = Wd;
      if (RHS) {   // Z_K = W R_K
        const d4 rk = lds_tile_load(Rbuf, K, lane);
        d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) z = mfma(Wd[s], rk[s], z);
        lds_tile_store(Rbuf, K, lane, z);
      }
      if (RHSMODE == 2) {   // z_K = W d_K : lane (g,c) sums W[c][4s+g] d[4s+g] over s, rows are summed over g
        double p = 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s) p = fma(Wd[s], dvec[16 * K + 4 * s + g], p);
        p = xrow_sum(p);
        __builtin_amdgcn_wave_barrier();
        if (g == 0) {
          dvec[16 * K + c] = p;
          zq = fma(p, p, zq);
        }
      }
    }
    HGP_CF(0);
    __syncthreads();
    HGP_CF(1);
    d4 W;
#pragma unroll
    for (int s = 0; s < 4; ++s) W[s] = Wbuf[s * 64 + lane];
    // panel: U_KJ = W A_KJ for my columns J > K; publish them
#pragma unroll
    for (int q = qK; q < C::NQ; ++q) {
      const int J = Coop<NB>::col(q, wave);
      if (J > K && J < NB) {
        const d4 t = U[C::loc(K, q)];
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(W[s], t[s], acc);
        U[C::loc(K, q)] = acc;
        lds_tile_store(rowbuf, J, lane, acc);
        if (Lout != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * J + c, col = 16 * K + g + 4 * r;
            if (row < n && col < n) Lout[(size_t)row * ldl + col] = acc[r];
          }
        }
        if (Lpack != nullptr) *reinterpret_cast<d4*>(Lpack + ((size_t)(J * (J - 1) / 2 + K) * 64 + lane) * 4) = acc;
      }
    }
    if (RHSMODE == 2) {   // d_J -= U_KJ^T z_K for my columns J > K (nobody else touches those 16 entries)
      double zr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) zr[r] = dvec[16 * K + g + 4 * r];
#pragma unroll
      for (int q = qK; q < C::NQ; ++q) {
        const int J = Coop<NB>::col(q, wave);
        if (J > K && J < NB) {
          double t = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) t = fma(U[C::loc(K, q)][r], zr[r], t);
          t = xrow_sum(t);
          if (g == 0) dvec[16 * J + c] -= t;
        }
      }
    }
    HGP_CF(2);
    __syncthreads();
    HGP_CF(3);
    // trailing: A_IJ -= U_KI^T U_KJ for my columns J >= I > K.  Row tile I of the panel is read from LDS once and
    // applied to all my columns; the next one is requested before the MFMAs of the current one are issued.
    if (K + 1 < NB) {
      d4 ucur = lds_tile_load(rowbuf, K + 1, lane);
#pragma unroll
      for (int I = K + 1; I < NB; ++I) {
        d4 unext = ucur;
        if (I + 1 < NB) unext = lds_tile_load(rowbuf, I + 1, lane);
#pragma unroll
        for (int q = I / 4; q < C::NQ; ++q) {
          if (C::col(q, wave) >= I) {
#pragma unroll
            for (int s = 0; s < 4; ++s) U[C::loc(I, q)] = mfma_sub(ucur[s], U[C::loc(K, q)][s], U[C::loc(I, q)]);
          }
        }
        ucur = unext;
      }
    }
    if (RHS) {   // R_I -= U_KI^T Z_K for I > K.  Row tile I belongs to the wave that OWNS block column I: that wave reads
      // R_I at the top of step I (Z_I = W R_I) with no workgroup barrier in between, so nobody else may write it.
      // (Until round 2 the rows were dealt round-robin, (I & 3) == wave, which differs from the snake ownership for
      // columns 4-7 and 12-15: a write/read race across waves that ~4 k cycles of diag16 slack hid, and that a build
      // slowed by register spills lost - the "wrong rows in L^-1" of DESIGN 7; `make raceprobe` + tools/probe_coop_race.py
      // reproduce it with an injected delay.)
      const d4 zk = lds_tile_load(Rbuf, K, lane);
#ifdef HGP_RACE_PROBE_DELAY
      if (K + 1 < NB && ((K + 1) & 3) == wave && C::owner(K + 1) != wave)
        for (int spin = 0; spin < 400; ++spin) __builtin_amdgcn_s_sleep(127);
#endif
#pragma unroll
      for (int I = K + 1; I < NB; ++I) {
#ifdef HGP_RACE_PROBE_ROUNDROBIN
        if ((I & 3) == wave) {
#else
        if (C::owner(I) == wave) {
#endif
          const d4 uki = lds_tile_load(rowbuf, I, lane);
          d4 ri = lds_tile_load(Rbuf, I, lane);
#pragma unroll
          for (int s = 0; s < 4; ++s) ri = mfma_sub(uki[s], zk[s], ri);
          lds_tile_store(Rbuf, I, lane, ri);
        }
      }
    }
    HGP_CF(4);
  }
  __syncthreads();
  return zq;
}
#undef HGP_CF

// combine the per-wave pivot accumulators: returns log det in every thread, info = first failing column (or 0)
__device__ __forceinline__ double coop_logdet_info(const PivotAcc& pa, int wave, int lane, double* red, int* redi, int& info) {
  if (lane == 0) {
    red[wave] = pa.logdet();
    redi[wave] = pa.info;
  }
  __syncthreads();
  const double ld = red[0] + red[1] + red[2] + red[3];
  int inf = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (redi[w] != 0 && (inf == 0 || redi[w] < inf)) inf = redi[w];
  info = inf;
  __syncthreads();
  return ld;
}

// =============================================================================================
// NB/2-wave cooperative version (CoopH<NB>: 8 waves at NB = 16, 6 at NB = 12, 4 at NB = 8): more than one wave per SIMD,
// so that the LDS/scalar latencies and the barrier waits of one wave sit under the MFMAs of another.  Wave w owns block
// columns w ("A": tiles (I, w), I <= w) and NB-1-w ("B": tiles (I, NB-1-w), I <= NB-1-w): EXACTLY NB + 1 tiles per wave,
// with slots that do not depend on w -
//     slotA(I) = NB - I,   slotB(I) = I       (A uses NB-w..NB, B uses 0..NB-1-w: disjoint for every w)
// so the register array is indexed statically.  Same step structure as coop_factor (diagonal block by the column owner,
// row panel and trailing update by column owner, two barriers per step); the right-hand side is ONE vector in LDS.
// =============================================================================================
template <int NB_>
struct CoopH {
  static_assert(NB_ % 2 == 0 && NB_ <= 16, "CoopH: even NB <= 16");
  static constexpr int NB = NB_, NT = NB_ + 1, NW = NB_ / 2;
  __host__ __device__ static constexpr int slotA(int I) { return NB - I; }
  __host__ __device__ static constexpr int slotB(int I) { return I; }
  __host__ __device__ static constexpr int owner(int J) { return J < NW ? J : NB - 1 - J; }
  __host__ __device__ static constexpr int diag_slot(int K) { return K < NW ? slotA(K) : slotB(K); }
};

// mean over i < n of |A_ii + shift| (LDS reduction in red[8])
template <int NB>
__device__ __forceinline__ double cooph_diag_abs_mean(const d4 (&U)[CoopH<NB>::NT], int n, int wave, int lane_in, double shift,
                                                      double* red) {
  double s = 0.0;
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    if (wave == CoopH<NB>::owner(K)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + g + 4 * r;
        if (g + 4 * r == c && i < n) s += fabs(U[CoopH<NB>::diag_slot(K)][r] + shift);
      }
    }
  }
  s = wave_sum(s);
  if (lane_in == 0) red[wave] = s;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < CoopH<NB>::NW; ++w) tot += red[w];
  __syncthreads();
  return tot / (double)n;
}

template <int NB>
__device__ __forceinline__ void cooph_add_diag(d4 (&U)[CoopH<NB>::NT], double shift, int n, int wave, int lane_in) {
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    if (wave == CoopH<NB>::owner(K)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + g + 4 * r;
        if (g + 4 * r == c && i < n) U[CoopH<NB>::diag_slot(K)][r] += shift;
      }
    }
  }
}

// 0.5 (A + A^T) of a row-major [n, ld] matrix into the CoopH tile layout of this wave (identity padding); scr_w: a 16 x 18 LDS
// staging tile private to the wave
template <int NB>
__device__ __forceinline__ void cooph_load_sym_upper(d4 (&U)[CoopH<NB>::NT], const double* __restrict__ A, int ld, int n, int wave,
                                                     int lane_in, double* scr_w) {
  using C = CoopH<NB>;
  // my block columns JA = wave (tiles I <= JA in slotA(I)) and JB = NB - 1 - wave (slotB(I)); both triangles are read as they lie
  // (coalesced rows) and the lower one is transposed through a per-wave 16 x 18 LDS tile: U = 0.5 (A + A^T), identity padding
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    const int J = h == 0 ? wave : NB - 1 - wave;
    d4 nat[NB], trn[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      if (h == 0 && I >= C::NW) continue;           // column A only reaches block rows < NW
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;          // natural element of tile (I, J)
        nat[I][r] = (I <= J && i < n && j < n) ? A[(size_t)i * ld + j] : 0.0;
        const int i2 = 16 * J + g + 4 * r, j2 = 16 * I + c;        // natural element of tile (J, I)
        trn[I][r] = (I < J && i2 < n && j2 < n) ? A[(size_t)i2 * ld + j2] : 0.0;
      }
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      if (h == 0 && I >= C::NW) continue;
      const d4 src = (I == J) ? nat[I] : trn[I];
#pragma unroll
      for (int r = 0; r < 4; ++r) scr_w[(g + 4 * r) * DIAG_LD + c] = src[r];
      __builtin_amdgcn_wave_barrier();
      d4 v = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + g + 4 * r, j = 16 * J + c;
        const double t = scr_w[c * DIAG_LD + g + 4 * r];
        double x = 0.5 * (nat[I][r] + t);
        if (!(i < n && j < n)) x = (i == j) ? 1.0 : 0.0;
        if (I <= J) v[r] = x;
      }
      __builtin_amdgcn_wave_barrier();
      if (h == 0) U[C::slotA(I < C::NW ? I : 0)] = (I <= J) ? v : U[C::slotA(I < C::NW ? I : 0)];
      else if (I <= J) U[C::slotB(I)] = v;
    }
  }
}

// Factor + eliminate the single right-hand side dvec (on exit z = L^{-1} d); returns this wave's share of z^T z.
template <int NB>
__device__ __forceinline__ double cooph_factor(d4 (&U)[CoopH<NB>::NT], double* rowbuf, double* Wbuf, double* scr, int wave,
                                               int lane_in, PivotAcc& pa, int n, double* dvec) {
  using C = CoopH<NB>;
  double zq = 0.0;
  const int JA = wave, JB = NB - 1 - wave;   // my block columns
#ifdef HGP_STAMPS
  unsigned long long cf_t = __builtin_readcyclecounter();
#define HGP_CF(i) do { unsigned long long n_ = __builtin_readcyclecounter(); pa.cf[i] += n_ - cf_t; cf_t = n_; } while (0)
#else
#define HGP_CF(i)
#endif
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    if (wave == C::owner(K)) {
      const d4 Wd = diag16_sel<false>(U[C::diag_slot(K)], scr, lane, pa, 16 * K, nullptr, 0, n - 16 * K);
#pragma unroll
      for (int s = 0; s < 4; ++s) Wbuf[s * 64 + lane] = Wd[s];
      double p = 0.0;   // z_K = W d_K
#pragma unroll
      for (int s = 0; s < 4; ++s) p = fma(Wd[s], dvec[16 * K + 4 * s + g], p);
      p = xrow_sum(p);
      __builtin_amdgcn_wave_barrier();
      if (g == 0) {
        dvec[16 * K + c] = p;
        zq = fma(p, p, zq);
      }
    }
    HGP_CF(0);
    __syncthreads();
    HGP_CF(1);
    d4 W;
#pragma unroll
    for (int s = 0; s < 4; ++s) W[s] = Wbuf[s * 64 + lane];
    double zr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zr[r] = dvec[16 * K + g + 4 * r];
    // panel + right-hand side for my columns J > K: slot of tile (K, J) is slotA(K) for J = JA, slotB(K) for J = JB
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0 && K >= C::NW) continue;         // column A has rows 0..w <= NW - 1 only
      const int J = h == 0 ? JA : JB;
      const int sl = h == 0 ? C::slotA(K) : C::slotB(K);
      if (J > K) {
        const d4 t = U[sl];
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(W[s], t[s], acc);
        U[sl] = acc;
        lds_tile_store(rowbuf, J, lane, acc);
        double tq = 0.0;                          // d_J -= U_KJ^T z_K
#pragma unroll
        for (int r = 0; r < 4; ++r) tq = fma(acc[r], zr[r], tq);
        tq = xrow_sum(tq);
        if (g == 0) dvec[16 * J + c] -= tq;
      }
    }
    HGP_CF(2);
    __syncthreads();
    HGP_CF(3);
    // trailing: A_IJ -= U_KI^T U_KJ for my columns J >= I > K
    if (K + 1 < NB) {
      d4 ucur = lds_tile_load(rowbuf, K + 1, lane);
#pragma unroll
      for (int I = K + 1; I < NB; ++I) {
        d4 unext = ucur;
        if (I + 1 < NB) unext = lds_tile_load(rowbuf, I + 1, lane);
        if (I < C::NW && K < C::NW && JA >= I) {   // tile (I, JA): needs I <= w <= NW - 1
#pragma unroll
          for (int s = 0; s < 4; ++s) U[C::slotA(I)] = mfma_sub(ucur[s], U[C::slotA(K)][s], U[C::slotA(I)]);
        }
        if (JB >= I) {
#pragma unroll
          for (int s = 0; s < 4; ++s) U[C::slotB(I)] = mfma_sub(ucur[s], U[C::slotB(K)][s], U[C::slotB(I)]);
        }
        ucur = unext;
      }
    }
    HGP_CF(4);
  }
  __syncthreads();
  return zq;
}
#undef HGP_CF

// ---------------------------------------------------------------------------------------------
// cooph_factor with DATAFLOW synchronisation instead of two workgroup barriers per block step.
//
// What a wave really waits for in step K is (i) W_K (and z_K) from the owner of the diagonal block, (ii) the row tiles
// (K, I) of exactly those block columns I it still updates, from their owners.  With barriers every wave also waits for the
// slowest wave's whole phase, and the 16 serial diag16 calls (4.2 k cycles each, one wave busy, seven idle) cannot overlap
// anybody's trailing update: the stamps show 74 k of the 154 k cycles of the factorisation spent at barriers.
// Here every produced item has a monotone counter in LDS:
//     wdone        number of diagonal blocks whose W (own buffer per K) and z_K are published,
//     rowpub[J]    number of row tiles (K, J), K = 0, 1, .. published by the owner of block column J  (tile K sits in the row
//                  buffer K % 3: three buffers, so a writer of row K must know that everybody has finished row K - 3:)
//     tdone[w]     number of block steps whose trailing update wave w has completed,
// producers store with release, consumers spin (s_sleep) with acquire.  The owner of diagonal block K + 1 factors it as soon
// as it has updated that one tile in step K - before the rest of its own trailing update - so W_{K+1} is usually there when
// the other waves finish step K: the diag16 chain runs under the trailing updates of the other waves (two waves share a SIMD:
// the VALU-only diag16 of one leaves the matrix pipe to the other).  Same arithmetic, same order per tile: bit-identical.
// A spin that exceeds SPIN_LIMIT sets `err` and falls through (results are then garbage and info reports it) - a bug here
// must fail a test, not hang the GPU.
// ---------------------------------------------------------------------------------------------
constexpr int COOPH_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ void df_publish(int* p, int v, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef HGP_RACE_STRESS
  {
    const unsigned h = ((threadIdx.x >> 6) + 1u) * 2654435761u ^ (unsigned)(__builtin_readcyclecounter() >> 7);
    const unsigned n = __builtin_amdgcn_readfirstlane((h >> 5) & 7u);
    for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
  }
#endif
}
__device__ __forceinline__ void df_wait_ge(int* p, int v, int* err, bool hot = false) {
  int it = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
    if (!hot) __builtin_amdgcn_s_sleep(1);
    if (++it > COOPH_SPIN_LIMIT) {
      *err = 1;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int NW>
__device__ __forceinline__ void df_wait_all_ge(int* tdone, int v, int lane, int* err) {
  int it = 0;
  while (!__all(__hip_atomic_load(&tdone[lane & (NW - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= v)) {
    __builtin_amdgcn_s_sleep(1);
    if (++it > COOPH_SPIN_LIMIT) {
      *err = 1;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// row0 / row1 / row2: three row buffers [NB][4][64]; Wall: [NB][4][64]; flags: 16 + NB ints (zeroed here).
// Two refinements on the critical path (the chain diag16(K) -> W_K -> tile (K, K+1) -> tile (K+1, K+1) -> diag16(K+1)):
//  * the owner of block K + 1 solves ONLY that panel tile first, updates its diagonal tile straight from the accumulator (no LDS
//    round trip), factors it and publishes W_{K+1} - its other panel tile and the rest of its trailing update come afterwards;
//  * the wave that will factor the next diagonal block polls for W_K without sleeping.
// (Measured and dropped: the SIMD-mate of the wave on the pivot chain holding its MFMAs back while the chain runs - f64 MFMA and
//  VALU instructions of two waves on one SIMD do not overlap - is SLOWER, 3.93 vs 3.69 ms per 8 192 pairs at T = 256: the polls
//  cost every tile update an LDS round trip; s_setprio(3) on the chain wave is worth 0.7 %.  Round 4 tried it again with the flag
//  read issued together with the LDS tile the update needs anyway (no extra round trip), the mate sleeping only while diag16_acc
//  runs: 14.16 vs 13.34 ms per 32 768 pairs - the MFMA time the mate loses is not given back by the shorter chain.)
// RP (round 4): instead of the single vector dvec, a block column of 16 right-hand sides rides the factorisation (the inversions
// of the member step, k_cooph_inv_rhs).  Tile I of it lives in the registers of the wave that owns block column I (*RAp for its column
// JA, *RBp for JB); Z_K = W_K R_K is formed by the pivot wave right behind diag16_acc and published in zbuf ([NB] tiles of LDS)
// together with W_K; R_J -= U_KJ^T Z_K follows each panel tile while it is still in its accumulator.  On exit the tiles hold L^-1 R.
// RHS = 2: no right-hand side at all (a8 / a9 above T = 128: the factor is wanted, packed).  Lpack / Wpack (optional, any mode): the
// factor in MFMA OPERAND order for a consumer that streams it, as coop_factor writes it - tile U_KJ (K < J) at index J (J - 1) / 2 + K
// of Lpack, W_K at index K of Wpack, 32 bytes per lane.
template <int NB, int RHS = 0>
__device__ __forceinline__ double cooph_factor_df(d4 (&U)[CoopH<NB>::NT], double* row0, double* row1, double* row2, double* Wall,
                                                  double* scr, int* flags, int wave, int lane_in, PivotAcc& pa, int n, double* dvec,
                                                  d4* RAp = nullptr, d4* RBp = nullptr, double* zbuf = nullptr,
                                                  double* Lpack = nullptr, double* Wpack = nullptr) {
  constexpr bool RP = (RHS == 1), NORHS = (RHS == 2);
  using C = CoopH<NB>;
  int* wdone = flags;
  int* tdone = flags + 1;        // [8]
  int* err = flags + 9;
  int* rowpub = flags + 16;      // [NB]
  if (threadIdx.x < 16 + NB) flags[threadIdx.x] = 0;
  __syncthreads();
  double zq = 0.0;
  const int JA = wave, JB = NB - 1 - wave;   // my block columns
  // what the steps so far have to subtract from d_JA / d_JB, as per-lane partial sums over the lane's own rows: reduced over the
  // four 16-lane rows once, when the column becomes the pivot block (do_diag), instead of once per step inside panel_col -
  // one cross-row sum, one LDS read-modify-write and their latency less on every step of the critical chain
  double dpA = 0.0, dpB = 0.0;
#ifdef HGP_STAMPS
  unsigned long long cf_t = __builtin_readcyclecounter();
#define HGP_DF(i) do { unsigned long long n_ = __builtin_readcyclecounter(); pa.cf[i] += n_ - cf_t; cf_t = n_; } while (0)
#else
#define HGP_DF(i)
#endif

  auto do_diag = [&](auto Kc) {
    constexpr int K = decltype(Kc)::value;
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    HGP_DF(4);
    __builtin_amdgcn_s_setprio(3);
    const d4 Wd = diag16_sel<false>(U[C::diag_slot(K)], scr, lane, pa, 16 * K, nullptr, 0, n - 16 * K);
#pragma unroll
    for (int s = 0; s < 4; ++s) Wall[(K * 4 + s) * 64 + lane] = Wd[s];
    if (Wpack != nullptr) *reinterpret_cast<d4*>(Wpack + ((size_t)K * 64 + lane) * 4) = Wd;
    if constexpr (NORHS) {
    } else if constexpr (RP) {      // Z_K = W R_K
      d4& rk = (K < C::NW) ? *RAp : *RBp;
      d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) z = mfma(Wd[s], rk[s], z);
      rk = z;
      lds_tile_store(zbuf, K, lane, z);
    } else {
      if constexpr (K > 0) {   // d_K -= sum over K' < K of U_K'K^T z_K' (deferred)
        const double q = xrow_sum((K < C::NW) ? dpA : dpB);
        if (g == 0) dvec[16 * K + c] -= q;
        __builtin_amdgcn_wave_barrier();
      }
      double p = 0.0;   // z_K = W d_K
#pragma unroll
      for (int s = 0; s < 4; ++s) p = fma(Wd[s], dvec[16 * K + 4 * s + g], p);
      p = xrow_sum(p);
      __builtin_amdgcn_wave_barrier();
      if (g == 0) {
        dvec[16 * K + c] = p;
        zq = fma(p, p, zq);
      }
    }
    df_publish(wdone, K + 1, lane_in);
    __builtin_amdgcn_s_setprio(0);
    HGP_DF(0);
  };
  if (wave == C::owner(0)) do_diag(std::integral_constant<int, 0>{});
  static_for<0, NB>([&](auto Kc) {
    constexpr int K = decltype(Kc)::value;
    const int lane = launder(lane_in);
    const int g = lane >> 4, c = lane & 15;
    HGP_DF(4);
    df_wait_ge(wdone, K + 1, err, (K + 1 < NB) && wave == C::owner(K + 1 < NB ? K + 1 : 0));
    HGP_DF(1);
    d4 W;
#pragma unroll
    for (int s = 0; s < 4; ++s) W[s] = Wall[(K * 4 + s) * 64 + lane];
    double zr[4] = {0.0, 0.0, 0.0, 0.0};
    d4 zk = (d4){0.0, 0.0, 0.0, 0.0};
    if constexpr (NORHS) {
    } else if constexpr (RP) {
      zk = lds_tile_load(zbuf, K, lane);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) zr[r] = dvec[16 * K + g + 4 * r];
    }
    double* rb = (K % 3 == 0) ? row0 : (K % 3 == 1 ? row1 : row2);      // three row buffers: the writer of row K needs row K - 3 dead
    const bool haveA = (K < C::NW) && (JA > K), haveB = JB > K;
    if constexpr (K >= 3) {
      if (haveA || haveB) {
        if constexpr ((C::NW & (C::NW - 1)) == 0) df_wait_all_ge<C::NW>(tdone, K - 2, lane_in, err);
        else {
          for (int w = 0; w < C::NW; ++w) df_wait_ge(&tdone[w], K - 2, err);
        }
      }
    }
    HGP_DF(3);
    // panel tile (K, J) of my column h (0: JA, 1: JB) + the right-hand side; returns the tile
    auto panel_col = [&](auto hc) -> d4 {
      constexpr int h = decltype(hc)::value;
      constexpr int sl = (h == 0) ? C::slotA(K < C::NW ? K : 0) : C::slotB(K);
      const int J = (h == 0) ? JA : JB;
      const d4 t = U[sl];
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma(W[s], t[s], acc);
      U[sl] = acc;
      lds_tile_store(rb, J, lane, acc);
      if (Lpack != nullptr) *reinterpret_cast<d4*>(Lpack + ((size_t)(J * (J - 1) / 2 + K) * 64 + lane) * 4) = acc;
      if constexpr (NORHS) {
      } else if constexpr (RP) {                // R_J -= U_KJ^T Z_K (the accumulator tile IS the A operand of its transpose)
        d4& rj = (h == 0) ? *RAp : *RBp;
#pragma unroll
        for (int s = 0; s < 4; ++s) rj = mfma_sub(acc[s], zk[s], rj);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {           // d_J -= U_KJ^T z_K (partial sums, see dpA / dpB)
          if (h == 0) dpA = fma(acc[r], zr[r], dpA);
          else dpB = fma(acc[r], zr[r], dpB);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane_in == 0) __hip_atomic_store(&rowpub[J], K + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return acc;
    };
    // is this wave the owner of the NEXT diagonal block?  then that block's column goes first (critical path)
    constexpr int HC = (K + 1 < C::NW) ? 0 : 1;                 // column (A / B) of block K + 1 in its owner's registers
    bool crit = false;
    if constexpr (K + 1 < NB) crit = (wave == C::owner(K + 1));
#ifdef HGP_DF_NOCRIT
    crit = false;
#endif
    if constexpr (K + 1 < NB) {
      if (crit) {
        const d4 acc = panel_col(std::integral_constant<int, HC>{});
        HGP_DF(2);
        constexpr int sd = C::diag_slot(K + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) U[sd] = mfma_sub(acc[s], acc[s], U[sd]);      // tile (K+1, K+1) -= U_K,K+1^T U_K,K+1
        do_diag(std::integral_constant<int, K + 1>{});
      }
    }
    if (!(crit && HC == 0) && haveA) {
      if constexpr (K < C::NW) {
        panel_col(std::integral_constant<int, 0>{});
      }
    }
    if (!(crit && HC == 1) && haveB) {
      panel_col(std::integral_constant<int, 1>{});
    }
    HGP_DF(2);
    // trailing: A_IJ -= U_KI^T U_KJ for my columns J >= I > K
    static_for<K + 1, NB>([&](auto Ic) {
      constexpr int I = decltype(Ic)::value;
      bool needA = (I < C::NW) && (K < C::NW) && (JA >= I), needB = JB >= I;
      if constexpr (I == K + 1) {               // the owner of block K + 1 has already updated its diagonal tile
        if (crit && HC == 0) needA = false;
        if (crit && HC == 1) needB = false;
      }
      if (needA || needB) {
        HGP_DF(4);
        df_wait_ge(&rowpub[I], K + 1, err);
        HGP_DF(3);
        const d4 ucur = lds_tile_load(rb, I, lane);
        if constexpr (I < C::NW && K < C::NW) {
          if (needA) {
#pragma unroll
            for (int s = 0; s < 4; ++s) U[C::slotA(I)] = mfma_sub(ucur[s], U[C::slotA(K)][s], U[C::slotA(I)]);
          }
        }
        if (needB) {
#pragma unroll
          for (int s = 0; s < 4; ++s) U[C::slotB(I)] = mfma_sub(ucur[s], U[C::slotB(K)][s], U[C::slotB(I)]);
        }
      }
#ifdef HGP_DF_NOCRIT
      if constexpr (I == K + 1) {
        if (wave == C::owner(I)) do_diag(std::integral_constant<int, I>{});
      }
#endif
    });
    df_publish(&tdone[wave], K + 1, lane_in);
  });
  __syncthreads();
  if (*err) pa.info = 1;
  return zq;
}
#undef HGP_DF

template <int NB>
__device__ __forceinline__ double cooph_logdet_info(const PivotAcc& pa, int wave, int lane, double* red, int* redi, int& info) {
  if (lane == 0) {
    red[wave] = pa.logdet();
    redi[wave] = pa.info;
  }
  __syncthreads();
  double ld = 0.0;
  int inf = 0;
#pragma unroll
  for (int w = 0; w < CoopH<NB>::NW; ++w) {
    ld += red[w];
    if (redi[w] != 0 && (inf == 0 || redi[w] < inf)) inf = redi[w];
  }
  info = inf;
  __syncthreads();
  return ld;
}

}  // namespace hgp
