// a8 (GPI_model.log_lat_error, GPI_model.py:288-323) and a9 (matrix_normal_inv_wishart.log_likelihood_MNIW,
// GPI_model.py:1346-1362) as ONE kernel each, one wavefront per item, T <= 128: the matrix-valued terms are traces of the
// form  tr(X^T G^{-1} Y) = sum (L^{-1} X) o (L^{-1} Y)  with G = L L^T, and a forward solve acts on the columns of its
// right-hand side independently - so the wave factors G once in registers (tile_f64.hpp), then walks the 16-column panels:
// builds the two right-hand-side panels (GEMMs on the matrix core straight from global memory), forward-solves both with
// the stored factor and accumulates their element-wise product.  No intermediate matrix ever goes to memory (the first
// version was a 5-kernel composition through 4 T^2 doubles of workspace per item: 1.2 M evals/s at T = 90).
//   a8:  r = f_cur - A f_prev;  out = -0.5 (|L^{-1} r|^2 + sum (L^{-1} (A P)) o (L^{-1} A)),         G = _chol_spd(Gamma)
//   a9:  D = M - m_mean;        out = -0.5 sum (L^{-1} (D R)) o (L^{-1} D) - 0.5 sum (L^{-1} S) o (L^{-1} I),  G = Sigma + 1e-8 I
// Algorithmic bytes per item: 3 T^2 + 2 T doubles (a8), 2-4 T^2 (a9, the prior shared by all items).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

namespace {

// tile (I, J) of a row-major [n, n] matrix (ld = n) in accumulator layout, zero outside; optionally minus a second matrix
__device__ __forceinline__ d4 load_acc_tile(const double* __restrict__ A, const double* __restrict__ B, int n, int I, int J, int lane) {
  const int g = lane >> 4, c = lane & 15;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 16 * I + g + 4 * r, j = 16 * J + c;
    double x = 0.0;
    if (i < n && j < n) {
      x = A[(size_t)i * n + j];
      if (B) x -= B[(size_t)i * n + j];
    }
    v[r] = x;
  }
  return v;
}

// the same tile as the A operand of a product (lane (g, c) holds X[16 I + c][16 K + 4 s + g] in element s)
__device__ __forceinline__ d4 load_aop_tile(const double* __restrict__ A, const double* __restrict__ B, int n, int I, int K, int lane) {
  const int g = lane >> 4, c = lane & 15;
  d4 v;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int i = 16 * I + c, k = 16 * K + 4 * s + g;
    double x = 0.0;
    if (i < n && k < n) {
      x = A[(size_t)i * n + k];
      if (B) x -= B[(size_t)i * n + k];
    }
    v[s] = x;
  }
  return v;
}

template <int NB>
__device__ __forceinline__ double tiles_dot(const d4 (&X)[NB], const d4 (&Y)[NB], int K0 = 0) {
  double s = 0.0;
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    if (K < K0) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) s = fma(X[K][r], Y[K][r], s);
  }
  return s;
}

// R[:, J] = (X - X2) Pm[:, J]  (X2, Pm2 optional subtrahends; Pm tile rows beyond n are zero)
template <int NB>
__device__ __forceinline__ void panel_gemm(d4 (&R)[NB], const double* __restrict__ X, const double* __restrict__ X2,
                                           const double* __restrict__ Pm, int n, int J, int lane_in) {
#pragma unroll
  for (int K = 0; K < NB; ++K) R[K] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma nounroll
  for (int Kk = 0; Kk < NB; ++Kk) {
    const int lane = launder(lane_in);
    if (16 * Kk >= n) break;
    const d4 pb = load_acc_tile(Pm, nullptr, n, Kk, J, lane);      // B operand: rows 16Kk + 4s + g, column 16J + c
#pragma unroll
    for (int K = 0; K < NB; ++K) {
      const d4 xa = load_aop_tile(X, X2, n, K, Kk, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s) R[K] = mfma(xa[s], pb[s], R[K]);
    }
  }
}

struct LatArgs {
  const double* f_cur;
  const double* f_prev;
  const double* A;
  const double* Gamma;
  const double* P;
  int T, b;
  double* out;
  int32_t* info;
};

template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_wave_lat(LatArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  __shared__ __attribute__((aligned(16))) double w_all[WAVES * NB * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int it = blockIdx.x * WAVES + wave;
  if (it >= a.b) return;
  double* scr = scr_all + wave * DIAG_SCR;
  double* Wl = w_all + wave * NB * 256;
  const int T = a.T;
  const size_t tt = (size_t)T * T;
  const double* A = a.A + it * tt;
  const double* P = a.P + it * tt;
  d4 U[NB * (NB + 1) / 2];
  d4 R1[NB], R2[NB];
  load_sym_upper<NB>(U, a.Gamma + it * tt, T, T, lane, scr);
  {
    const double dm = diag_abs_mean<NB>(U, T, lane);
    add_diag<NB>(U, 1e-8 * fmax(dm, F64_EPS), T, lane);           // _chol_spd, GPI_model.py:83-87
  }
  PivotAcc pa;
  pa.init();
  wave_factor<NB, 0, (NB >= 8)>(U, R1, scr, Wl, nullptr, lane, pa, nullptr, 0, T);
  double acc = 0.0;
  // vector panel: column 0 = r = f_cur - A f_prev (the product rides the matrix core with f_prev as a one-column operand)
  {
    const double* fc = a.f_cur + (size_t)it * T;
    const double* fp = a.f_prev + (size_t)it * T;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + g + 4 * r;
        R1[K][r] = (c == 0 && i < T) ? fc[i] : 0.0;
      }
#pragma nounroll
    for (int Kk = 0; Kk < NB; ++Kk) {
      const int ln = launder(lane);
      if (16 * Kk >= T) break;
      d4 fb;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 16 * Kk + 4 * s + (ln >> 4);
        fb[s] = ((ln & 15) == 0 && k < T) ? fp[k] : 0.0;
      }
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const d4 xa = load_aop_tile(A, nullptr, T, K, Kk, ln);
#pragma unroll
        for (int s = 0; s < 4; ++s) R1[K] = mfma_sub(xa[s], fb[s], R1[K]);
      }
    }
    wave_fwd_solve<NB>(U, Wl, R1, lane);
    acc += tiles_dot<NB>(R1, R1);                                  // |L^{-1} r|^2 (the other columns are zero)
  }
#pragma nounroll
  for (int J = 0; J < NB; ++J) {
    const int ln = launder(lane);
    if (16 * J >= T) break;
    panel_gemm<NB>(R1, A, nullptr, P, T, J, ln);                   // (A P)[:, J]
#pragma unroll
    for (int K = 0; K < NB; ++K) R2[K] = load_acc_tile(A, nullptr, T, K, J, ln);
    wave_fwd_solve<NB>(U, Wl, R1, ln);
    wave_fwd_solve<NB>(U, Wl, R2, ln);
    acc += tiles_dot<NB>(R1, R2);                                  // tr(A^T Gamma^{-1} A P), columns of panel J
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    a.out[it] = -0.5 * acc;
    if (a.info) a.info[it] = pa.info;
  }
}


// T <= 96 (NB <= 6): the Gram form of a8.  All NB^2 tiles of Y = L^{-1} A fit the register file next to the factor (36 + 21 tiles
// at NB = 6), so the trace is taken as tr(A^T Gamma^{-1} A P) = tr((Y^T Y) P) = sum_{I <= J} G_IJ o (P_IJ + P_JI^T): the NB panel
// solves of A only, one Gram sweep (upper tiles) and an element-wise product with P straight from global memory - 1 300 MFMAs per
// item at T = 90 instead of 2 500 (no A P product, no second set of panel solves).
// One wave per SIMD (456 VGPRs), so every exposed load latency is idle time; the order of the memory traffic is the design:
//   * Gamma arrives in ONE burst (load_sym_upper_burst: both triangles of every tile in flight, then the LDS transposes), the
//     NB^2 tiles of A in a second one (they land in the registers that will hold Y): two exposed latencies instead of 13;
//   * r = f_cur - A f_prev is formed on the VALU from those tiles (4 FMAs per tile + a 16-lane row sum) and rides the
//     factorisation as a vector (wave_factor RHSMODE 2, |L^-1 r|^2 comes back): no second pass over A, no extra panel;
//   * in the Gram sweep the P values of the NEXT tile are requested before the MFMAs of the current one.
// (first fused version: A read three times, 48 dependent load round trips per item: 6.9 M evals/s at T = 90.)
__device__ __forceinline__ double row16_sum(double v) {   // sum over the 16 lanes of a row (same g), valid in every lane
  v += dpp_f64<0x128>(v);
  v += dpp_f64<0x124>(v);
  v += dpp_f64<0x122>(v);
  v += dpp_f64<0x121>(v);
  return v;
}

// P_IJ + P_JI^T (or P_II) in accumulator layout; indices clamped into the matrix instead of predicated: the padded
// entries multiply exact zeros of the Gram tile
__device__ __forceinline__ d4 load_psym_tile(const double* __restrict__ P, int n, int I, int J, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int j = min(16 * J + c, n - 1);
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(16 * I + g + 4 * r, n - 1);
    double x = P[(size_t)i * n + j];
    if (I != J) x += P[(size_t)j * n + i];
    v[r] = x;
  }
  return v;
}

template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_wave_lat_gram(LatArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  __shared__ __attribute__((aligned(16))) double w_all[WAVES * NB * 256];
  __shared__ __attribute__((aligned(16))) double d_all[WAVES * NB * 32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int it = blockIdx.x * WAVES + wave;
  if (it >= a.b) return;
  double* scr = scr_all + wave * DIAG_SCR;
  double* Wl = w_all + wave * NB * 256;
  double* dvec = d_all + wave * NB * 32;                           // f_cur, then r, then L^{-1} r
  double* fprev = dvec + NB * 16;
  const int T = a.T;
  const size_t tt = (size_t)T * T;
  const double* A = a.A + it * tt;
  const double* P = a.P + it * tt;
  d4 U[NB * (NB + 1) / 2];
  d4 Y[NB][NB];                                                    // Y[J][K] = tile (K, J) of A, then of L^{-1} A
  load_sym_upper_burst<NB>(U, a.Gamma + it * tt, T, T, lane, scr);   // one burst: Gamma, both triangles
  {                                                                   // second burst: A (into Y) and the two vectors (into LDS)
    const int ln = launder(lane);
    const double* fc = a.f_cur + (size_t)it * T;
    const double* fp = a.f_prev + (size_t)it * T;
#pragma unroll
    for (int J = 0; J < NB; ++J)
#pragma unroll
      for (int K = 0; K < NB; ++K) Y[J][K] = load_acc_tile(A, nullptr, T, K, J, ln);
    for (int i = ln; i < 16 * NB; i += 64) {
      dvec[i] = (i < T) ? fc[i] : 0.0;
      fprev[i] = (i < T) ? fp[i] : 0.0;
    }
  }
  {
    const double dm = diag_abs_mean<NB>(U, T, lane);
    add_diag<NB>(U, 1e-8 * fmax(dm, F64_EPS), T, lane);           // _chol_spd, GPI_model.py:83-87
  }
  {   // r = f_cur - A f_prev -> dvec (row 16 K + g + 4 r by the lanes c == 0)
    const int ln = launder(lane);
    const int g = ln >> 4, c = ln & 15;
    double fpv[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) fpv[J] = fprev[16 * J + c];
#pragma unroll
    for (int K = 0; K < NB; ++K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double sacc = 0.0;
#pragma unroll
        for (int J = 0; J < NB; ++J) sacc = fma(Y[J][K][r], fpv[J], sacc);
        sacc = row16_sum(sacc);
        if (c == 0) dvec[16 * K + g + 4 * r] -= sacc;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  PivotAcc pa;
  pa.init();
  double acc = wave_factor<NB, 2>(U, Y[0], scr, Wl, dvec, lane, pa, nullptr, 0, T);   // |L^{-1} r|^2 (Y[0] is not touched)
  acc *= (1.0 / 64.0);                                             // valid in every lane; the final wave_sum adds it 64 times
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    const int ln = launder(lane);
    wave_fwd_solve<NB>(U, Wl, Y[J], ln);
  }
  // Gram sweep + trace with P, the P tile of the next (I, J) in flight under the MFMAs of the current one
  d4 pn = load_psym_tile(P, T, 0, 0, launder(lane));
#pragma unroll
  for (int I = 0; I < NB; ++I) {
#pragma unroll
    for (int J = I; J < NB; ++J) {
      const int ln = launder(lane);
      const d4 pc = pn;
      const int In = (J + 1 < NB) ? I : I + 1, Jn = (J + 1 < NB) ? J + 1 : I + 1;
      if (In < NB) pn = load_psym_tile(P, T, In, Jn, ln);
      d4 G = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int K = 0; K < NB; ++K)
#pragma unroll
        for (int s = 0; s < 4; ++s) G = mfma(Y[I][K][s], Y[J][K][s], G);          // (Y_KI)^T Y_KJ
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = fma(G[r], pc[r], acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    a.out[it] = -0.5 * acc;
    if (a.info) a.info[it] = pa.info;
  }
}

struct MniwArgs {
  const double* M;
  const double* Sigma;
  const double* m_mean;
  const double* R;        // may be NULL = identity
  const double* S;
  int s_diag;             // the scale matrix is diagonal (the hot path: the prior's sigma I): tr(Sigma^-1 S) = sum_j S_jj |L^-1 e_j|^2
  long prior_stride;
  int T, b;
  double* out;
  int32_t* info;
};

// HASR / SDIAG are compile-time so that the hot call (identity right covariance, diagonal prior scale) carries ONE panel next to
// the factor: 27 tiles instead of 33, which is what lets two waves share a SIMD at NB = 6.
template <int NB, bool HASR, bool SDIAG>
__global__ __launch_bounds__(64 * WAVES, (NB <= 6 && !HASR && SDIAG) ? 2 : 1) void k_wave_mniw(MniwArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  __shared__ __attribute__((aligned(16))) double w_all[WAVES * NB * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int it = blockIdx.x * WAVES + wave;
  if (it >= a.b) return;
  double* scr = scr_all + wave * DIAG_SCR;
  double* Wl = w_all + wave * NB * 256;
  const int T = a.T;
  const size_t tt = (size_t)T * T;
  const double* M = a.M + it * tt;
  const double* mm = a.m_mean + (size_t)it * a.prior_stride;
  const double* Rm = HASR ? a.R + (size_t)it * a.prior_stride : nullptr;
  const double* Sm = a.S + (size_t)it * a.prior_stride;
  d4 U[NB * (NB + 1) / 2];
  d4 R1[(HASR || !SDIAG) ? NB : 1], R2[NB];
  load_sym_upper<NB>(U, a.Sigma + it * tt, T, T, lane, scr);
  add_diag<NB>(U, 1e-8, T, lane);                                  // chol(0.5 (S + S^T) + 1e-8 I), GPI_model.py:1353
  PivotAcc pa;
  pa.init();
  wave_factor<NB, 0, (NB >= 8)>(U, R2, scr, Wl, nullptr, lane, pa, nullptr, 0, T);
  double acc = 0.0;
#pragma nounroll
  for (int J = 0; J < NB; ++J) {
    const int ln = launder(lane);
    if (16 * J >= T) break;
    // mean term: sum (L^{-1} (D R)) o (L^{-1} D), D = M - m_mean
#pragma unroll
    for (int K = 0; K < NB; ++K) R2[K] = load_acc_tile(M, mm, T, K, J, ln);
    wave_fwd_solve<NB>(U, Wl, R2, ln);
    if constexpr (HASR) {
      panel_gemm<NB>(R1, M, mm, Rm, T, J, ln);
      wave_fwd_solve<NB>(U, Wl, R1, ln);
      acc += tiles_dot<NB>(R1, R2);
    } else {
      acc += tiles_dot<NB>(R2, R2);
    }
    // scale term: tr(Sigma^{-1} S) = sum (L^{-1} S) o (L^{-1} I); the identity panel is zero above block J
#pragma unroll
    for (int K = 0; K < NB; ++K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) R2[K][r] = (K == J && (ln >> 4) + 4 * r == (ln & 15)) ? 1.0 : 0.0;
    }
    wave_fwd_solve<NB>(U, Wl, R2, ln, J);
    if constexpr (SDIAG) {   // diagonal S: column j of L^{-1} S is S_jj times column j of L^{-1}
      const int j = 16 * J + (ln & 15);
      const double sj = (j < T) ? Sm[(size_t)j * T + j] : 0.0;
      acc = fma(sj, tiles_dot<NB>(R2, R2, J), acc);
    } else {
#pragma unroll
      for (int K = 0; K < NB; ++K) R1[K] = load_acc_tile(Sm, nullptr, T, K, J, ln);
      wave_fwd_solve<NB>(U, Wl, R1, ln);
      acc += tiles_dot<NB>(R1, R2, J);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    a.out[it] = -0.5 * acc;
    if (a.info) a.info[it] = pa.info;
  }
}

}  // namespace

int hgp_internal_lat_error_wave(const double* f_cur, const double* f_prev, const double* A, const double* Gamma, const double* covprev,
                                int T, int b, double* out, int32_t* info, hipStream_t st) {
  LatArgs a{f_cur, f_prev, A, Gamma, covprev, T, b, out, info};
  dim3 grid((b + WAVES - 1) / WAVES), blk(64 * WAVES);
  switch (nb_for(T)) {
    case 2: hipLaunchKernelGGL(k_wave_lat_gram<2>, grid, blk, 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_wave_lat_gram<4>, grid, blk, 0, st, a); break;
    case 6: hipLaunchKernelGGL(env_on("HGP_LAT_PANEL") ? k_wave_lat<6> : k_wave_lat_gram<6>, grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL(k_wave_lat<8>, grid, blk, 0, st, a); break;   // 64 tiles of Y do not fit next to the factor: panel form
  }
  return launch_status();
}

int hgp_internal_mniw_wave(const double* M, const double* Sigma, const double* m_mean, const double* m_r_cov, const double* scale,
                           int scale_is_diagonal, long prior_stride, int T, int b, double* out, int32_t* info, hipStream_t st) {
  MniwArgs a{M, Sigma, m_mean, m_r_cov, scale, scale_is_diagonal, prior_stride, T, b, out, info};
  dim3 grid((b + WAVES - 1) / WAVES), blk(64 * WAVES);
  const bool hasr = m_r_cov != nullptr, sd = scale_is_diagonal != 0;
#define HGP_MNIW(NB_)                                                                                         \
  do {                                                                                                        \
    if (hasr && sd) hipLaunchKernelGGL((k_wave_mniw<NB_, true, true>), grid, blk, 0, st, a);                  \
    else if (hasr) hipLaunchKernelGGL((k_wave_mniw<NB_, true, false>), grid, blk, 0, st, a);                  \
    else if (sd) hipLaunchKernelGGL((k_wave_mniw<NB_, false, true>), grid, blk, 0, st, a);                    \
    else hipLaunchKernelGGL((k_wave_mniw<NB_, false, false>), grid, blk, 0, st, a);                           \
  } while (0)
  switch (nb_for(T)) {
    case 2: HGP_MNIW(2); break;
    case 4: HGP_MNIW(4); break;
    case 6: HGP_MNIW(6); break;
    default: HGP_MNIW(8); break;
  }
#undef HGP_MNIW
  return launch_status();
}
