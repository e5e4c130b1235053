// libhdpgpc_hip.so - HIP kernels (gfx950) and the C-ABI of include/hdpgpc_hip.h.
// One wavefront owns one SPD matrix (<= 128 x 128) in MFMA accumulator registers; see tile_f64.hpp.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/hdpgpc_hip.h"
#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

namespace {

// ------------------------------------------------------------------------------------------ a1
__global__ void k_gram_rbf(const double* __restrict__ x, int nx, const double* __restrict__ y, int ny, double c,
                           double ell, double noise, int one_arg, double* __restrict__ K) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)nx * ny) return;
  int i = (int)(idx / ny), j = (int)(idx % ny);
  double u = x[i] / ell - (one_arg ? x[j] : y[j]) / ell;   // sklearn divides by the length-scale first
  double v = c * exp(-0.5 * (u * u));
  if (one_arg && i == j) v = c + noise;
  K[idx] = v;
}

// diagnostics: one 16x16x16 product through the operand / accumulator lane maps tile_f64.hpp assumes
__global__ void k_mfma_probe(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = mfma(A[c * 16 + 4 * s + g], B[(4 * s + g) * 16 + c], acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(g + 4 * r) * 16 + c] = acc[r];
}

// diagnostics: the kernels' own exp(-h), four values per lane
__global__ void k_exp_probe(const double* __restrict__ h, int n, double* __restrict__ out) {
  const int i = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 3 >= n + 0 && i >= n) return;
  double hv[4], ev[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) hv[j] = (i + j < n) ? h[i + j] : 0.0;
  exp_neg4(hv, ev);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (i + j < n) out[i + j] = ev[j];
}

// ------------------------------------------------------------------------------------------ a3
struct PotrfArgs {
  double* A;
  int T, b;
  double jitter_rel, add;
  double* Linv;
  double* logdet;
  int32_t* info;
  int inv_info = 0;   // k_wave_inv also reports info (used when no in-place factor follows)
  int symmetric = 0;  // the caller guarantees A == A^T bit for bit: only the upper tiles are read
  double* Aout = nullptr;   // cooperative factor only: L goes here instead of over A (hgp_chol_inverse_ws_f64)
};

template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_wave_potrf(PotrfArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int m = blockIdx.x * WAVES + wave;
  if (m >= a.b) return;
  double* scr = scr_all + wave * DIAG_SCR;
  double* A = a.A + (size_t)m * a.T * a.T;
  const int T = a.T;
  d4 U[NB * (NB + 1) / 2];
  d4 R[NB];
  load_sym_upper<NB>(U, A, T, T, lane, scr);
  if (a.add != 0.0) add_diag<NB>(U, a.add, T, lane);
  if (a.jitter_rel != 0.0) {
    double dm = diag_abs_mean<NB>(U, T, lane);
    add_diag<NB>(U, a.jitter_rel * fmax(dm, F64_EPS), T, lane);
  }
  PivotAcc pa;
  pa.init();
  wave_factor<NB, 0>(U, R, scr, nullptr, nullptr, lane, pa, A, T, T);
  // zero the strictly upper blocks of the in-place result (torch.linalg.cholesky returns zeros there)
#pragma unroll
  for (int I = 0; I < NB; ++I)
#pragma unroll
    for (int J = I + 1; J < NB; ++J) {
      const int ln = launder(lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = 16 * I + (ln >> 4) + 4 * r, j = 16 * J + (ln & 15);
        if (i < T && j < T) A[(size_t)i * T + j] = 0.0;
      }
    }
  if (lane == 0) {
    if (a.info) a.info[m] = pa.info;
    if (a.logdet) a.logdet[m] = pa.logdet();
  }
}

// L^{-1} of the regularised matrix, one wave per (matrix, block column): the identity block column Jc rides along
// the factorisation as its 16 right-hand sides.  Reads A (never writes it), so it runs BEFORE an in-place k_wave_potrf.
template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_wave_inv(PotrfArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int w = blockIdx.x * WAVES + wave;
  const int m = w / NB, Jc = w % NB;
  const int T = a.T;
  if (m >= a.b || 16 * Jc >= T) return;
  double* scr = scr_all + wave * DIAG_SCR;
  const double* A = a.A + (size_t)m * T * T;
  d4 U[NB * (NB + 1) / 2];
  d4 R[NB];
  if (a.symmetric) load_upper_only<NB>(U, A, T, T, lane);
  else if constexpr (NB <= 6) load_sym_upper_burst<NB>(U, A, T, T, lane, scr);   // latency-bound: all loads in flight at once
  else load_sym_upper<NB>(U, A, T, T, lane, scr);
  {
    double sh = a.add;
    if (a.jitter_rel != 0.0) sh += a.jitter_rel * fmax(diag_abs_mean<NB>(U, T, lane, a.add), F64_EPS);
    if (sh != 0.0) add_diag<NB>(U, sh, T, lane);
  }
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) R[K][r] = (K == Jc && g + 4 * r == c) ? 1.0 : 0.0;
  PivotAcc pa;
  pa.init();
  wave_factor<NB, 1>(U, R, scr, nullptr, nullptr, lane, pa, nullptr, 0, T);
  double* Z = a.Linv + (size_t)m * T * T;
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (i < T && j < T) Z[(size_t)i * T + j] = (K >= Jc) ? R[K][r] : 0.0;
    }
  if (a.inv_info && Jc == 0 && lane == 0 && a.info) a.info[m] = pa.info;   // block column 0 sees every pivot
}

template <int NB>
void launch_wave_inv(const PotrfArgs& a, hipStream_t st) {
  const int waves = a.b * NB;
  hipLaunchKernelGGL(k_wave_inv<NB>, dim3((waves + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, st, a);
}

// -------------------------------------------------------------------------------------- a4 + a6
struct ScoreArgs {
  const double* Y;
  int ldy;
  const double* mean;
  long mean_stride;
  const double* Sigma;
  long sigma_stride;
  int T;
  int ld_sigma;               // leading dimension of every Sigma matrix (= T for the public entry point)
  const int32_t* item_mat;
  const int32_t* item_mean;   // optional: row of `mean` per item (default: item_mat)
  const double* item_add;
  const int32_t* item_off;
  const int32_t* item_cnt;
  int n_items;
  const int32_t* seg_ids;
  double jitter_rel;
  double* out_quad;
  double* out_logdet;
  int32_t* out_info;
};

template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_wave_score(ScoreArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  __shared__ __attribute__((aligned(16))) double w_all[WAVES * NB * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int it = blockIdx.x * WAVES + wave;
  if (it >= a.n_items) return;
  double* scr = scr_all + wave * DIAG_SCR;
  const int T = a.T;
  const int mat = a.item_mat[it];
  const double* S = a.Sigma + (size_t)mat * a.sigma_stride;
  const double* mu = a.mean ? a.mean + (size_t)(a.item_mean ? a.item_mean[it] : mat) * a.mean_stride : nullptr;
  d4 U[NB * (NB + 1) / 2];
  d4 R[NB];
  double* Wl = w_all + wave * NB * 256;
  load_sym_upper<NB>(U, S, a.ld_sigma, T, lane, scr);
  const double add = a.item_add ? a.item_add[it] : 0.0;
  if (add != 0.0) add_diag<NB>(U, add, T, lane);
  if (a.jitter_rel != 0.0) {
    double dm = diag_abs_mean<NB>(U, T, lane);
    add_diag<NB>(U, a.jitter_rel * fmax(dm, F64_EPS), T, lane);
  }
  const int off = a.item_off[it], cnt = a.item_cnt[it];
  PivotAcc pa;
  pa.init();
  double ld = 0.0;
  // the first 16 segments ride along with the factorisation; further chunks reuse the stored factor
  for (int base = 0; base < cnt; base += 16) {
    const int j = base + c;
    const bool live = j < cnt;
    const int seg = live ? (a.seg_ids ? a.seg_ids[off + j] : off + j) : 0;
    const double* yr = a.Y + (size_t)seg * a.ldy;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = 16 * K + g + 4 * r;
        double v = 0.0;
        if (live && i < T) v = yr[i] - (mu ? mu[i] : 0.0);
        R[K][r] = v;
      }
    if (base == 0) {
      wave_factor<NB, 1, true>(U, R, scr, cnt > 16 ? Wl : nullptr, nullptr, lane, pa, nullptr, 0, T);
      ld = pa.logdet();
    } else {
      wave_fwd_solve<NB>(U, Wl, R, lane);
    }
    double q = wave_colnorm2<NB>(R);
    if (live && g == 0) {
      a.out_quad[seg] = q;
      if (a.out_logdet) a.out_logdet[seg] = ld;
      if (a.out_info) a.out_info[seg] = pa.info;
    }
  }
}

// ------------------------------------------------------------------------- a6, one state per segment
// The reference's real dataflow on a shared grid: every member segment i of a cluster is scored against ITS OWN
// Sigma_i (GPI_model.py:508-531), i.e. one factorisation per segment with a single right-hand side.  Lean variant
// of k_wave_score: the right-hand side is an LDS vector eliminated on the VALU (no RHS tiles), which brings the
// NB <= 6 instantiations under 256 registers -> two waves per SIMD, so one matrix's pivot chain overlaps another's
// loads and MFMAs.  HBM-bound in principle: 8 T^2 + 16 T + 8 bytes per evaluation.
struct EachArgs {
  const double* Y;
  int ldy;
  const double* mean;
  long mean_stride;
  const double* Sigma;
  long sigma_stride;
  int T, n;
  const int32_t* seg_mat;    // [n] Sigma index of segment i
  const int32_t* seg_mean;   // [n] mean row of segment i (NULL: seg_mat)
  const double* seg_add;     // [n] additive diagonal (NULL: 0)
  double jitter_rel;
  double* out_quad;
  double* out_logdet;
  int32_t* out_info;
  int symmetric;             // caller guarantees Sigma == Sigma^T bit for bit: read the upper triangle only
};

// (SYM: one instantiation per loader - with both in one function the NB = 8 kernel spilled 290 VGPRs)
template <int NB, bool SYM>
__global__ __launch_bounds__(64 * WAVES, (NB <= 6) ? 2 : 1) void k_wave_score1(EachArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  __shared__ __attribute__((aligned(16))) double dv_all[WAVES * 16 * NB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seg = blockIdx.x * WAVES + wave;
  if (seg >= a.n) return;
  double* scr = scr_all + wave * DIAG_SCR;
  double* dv = dv_all + wave * 16 * NB;
  const int T = a.T;
  const int mat = a.seg_mat[seg];
  const double* S = a.Sigma + (size_t)mat * a.sigma_stride;
  const double* mu = a.mean ? a.mean + (size_t)(a.seg_mean ? a.seg_mean[seg] : mat) * a.mean_stride : nullptr;
  const double* yr = a.Y + (size_t)seg * a.ldy;
  for (int i = lane; i < 16 * NB; i += 64) dv[i] = (i < T) ? yr[i] - (mu ? mu[i] : 0.0) : 0.0;
  d4 U[NB * (NB + 1) / 2];
  d4 Rnone[NB];
  if constexpr (SYM) load_upper_only<NB>(U, S, T, T, lane);
  else load_sym_upper<NB>(U, S, T, T, lane, scr);
  {
    double sh = a.seg_add ? a.seg_add[seg] : 0.0;
    if (a.jitter_rel != 0.0) sh += a.jitter_rel * fmax(diag_abs_mean<NB>(U, T, lane, sh), F64_EPS);
    if (sh != 0.0) add_diag<NB>(U, sh, T, lane);
  }
  __builtin_amdgcn_wave_barrier();
  PivotAcc pa;
  pa.init();
  const double q = wave_factor<NB, 2, (NB >= 8), (NB < 8)>(U, Rnone, scr, nullptr, dv, lane, pa, nullptr, 0, T);
  if (lane == 0) {
    a.out_quad[seg] = q;
    if (a.out_logdet) a.out_logdet[seg] = pa.logdet();
    if (a.out_info) a.out_info[seg] = pa.info;
  }
}

// ------------------------------------------------------------------ 128 < T <= 256: cooperative kernels
// One workgroup (4 waves) per matrix / work item; see Coop<> in tile_f64.hpp.
template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_coop_score(ScoreArgs a) {
  using C = Coop<NB>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* rowbuf = smem;
  double* Rbuf = rowbuf + NB * 256;
  double* Wbuf = Rbuf + NB * 256;
  double* scr = Wbuf + 256;
  double* red = scr + DIAG_SCR;
  int* redi = reinterpret_cast<int*>(red + 8);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int it = blockIdx.x;
  const int T = a.T;
  const int mat = a.item_mat[it];
  const double* S = a.Sigma + (size_t)mat * a.sigma_stride;
  const double* mu = a.mean ? a.mean + (size_t)(a.item_mean ? a.item_mean[it] : mat) * a.mean_stride : nullptr;
  const double add = a.item_add ? a.item_add[it] : 0.0;
  const int off = a.item_off[it], cnt = a.item_cnt[it];
  d4 U[C::NT];
  for (int base = 0; base < cnt; base += 16) {     // every chunk of 16 segments refactors (rare for T > 128)
    coop_load_sym_upper<NB>(U, S, a.ld_sigma, T, wave, lane, rowbuf + wave * DIAG_SCR);
    __syncthreads();   // rowbuf served as per-wave staging for the loader
    {
      double sh = add;
      if (a.jitter_rel != 0.0) {
        const double dm = coop_diag_abs_mean<NB>(U, T, wave, lane, add, red);
        sh += a.jitter_rel * fmax(dm, F64_EPS);
      }
      if (sh != 0.0) coop_add_diag<NB>(U, sh, T, wave, lane);
    }
    const int j = base + c;
    const bool live = j < cnt;
    const int seg = live ? (a.seg_ids ? a.seg_ids[off + j] : off + j) : 0;
    const double* yr = a.Y + (size_t)seg * a.ldy;
    for (int K = wave; K < NB; K += WAVES) {
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + g + 4 * r;
        v[r] = (live && i < T) ? yr[i] - (mu ? mu[i] : 0.0) : 0.0;
      }
      lds_tile_store(Rbuf, K, lane, v);
    }
    __syncthreads();
    PivotAcc pa;
    pa.init();
    coop_factor<NB, true>(U, rowbuf, Rbuf, Wbuf, scr, wave, lane, pa, nullptr, 0, T);
    int info;
    const double ld = coop_logdet_info(pa, wave, lane, red, redi, info);
    // quad_j = sum over all tiles of Z^2 in column j
    double q = 0.0;
    for (int K = wave; K < NB; K += WAVES) {
      const d4 z = lds_tile_load(Rbuf, K, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) q = fma(z[r], z[r], q);
    }
    q = xrow_sum(q);
    if (g == 0) Wbuf[wave * 16 + c] = q;
    __syncthreads();
    if (wave == 0 && g == 0 && live) {
      a.out_quad[seg] = Wbuf[c] + Wbuf[16 + c] + Wbuf[32 + c] + Wbuf[48 + c];
      if (a.out_logdet) a.out_logdet[seg] = ld;
      if (a.out_info) a.out_info[seg] = info;
    }
    __syncthreads();
  }
}

// in-place factor: A <- L (zeros above), info, logdet
template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_coop_potrf(PotrfArgs a) {
  using C = Coop<NB>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* rowbuf = smem;
  double* Rbuf = rowbuf + NB * 256;
  double* Wbuf = Rbuf + NB * 256;
  double* scr = Wbuf + 256;
  double* red = scr + DIAG_SCR;
  int* redi = reinterpret_cast<int*>(red + 8);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int m = blockIdx.x;
  const int T = a.T;
  double* A = (a.Aout ? a.Aout : a.A) + (size_t)m * T * T;     // where L goes
  d4 U[C::NT];
  coop_load_sym_upper<NB>(U, a.A + (size_t)m * T * T, T, T, wave, lane, rowbuf + wave * DIAG_SCR);
  __syncthreads();   // rowbuf served as per-wave staging for the loader
  {
    double sh = a.add;
    if (a.jitter_rel != 0.0) {
      const double dm = coop_diag_abs_mean<NB>(U, T, wave, lane, a.add, red);
      sh += a.jitter_rel * fmax(dm, F64_EPS);
    }
    if (sh != 0.0) coop_add_diag<NB>(U, sh, T, wave, lane);
  }
  __syncthreads();   // every wave has loaded its tiles before anyone overwrites A
  PivotAcc pa;
  pa.init();
  coop_factor<NB, false>(U, rowbuf, Rbuf, Wbuf, scr, wave, lane, pa, A, T, T, nullptr,
                         a.Linv ? a.Linv + (size_t)m * T * T : nullptr, T);   // + the diagonal blocks of L^-1 (k_trtri does the rest)
  int info;
  const double ld = coop_logdet_info(pa, wave, lane, red, redi, info);
  if (threadIdx.x == 0) {
    if (a.info) a.info[m] = info;
    if (a.logdet) a.logdet[m] = ld;
  }
  for (int idx = threadIdx.x; idx < T * T; idx += 64 * WAVES) {   // zeros above the diagonal blocks
    const int i = idx / T, j = idx % T;
    if ((j >> 4) > (i >> 4)) A[idx] = 0.0;
  }
}

// Z = L^-1 from L and the inverses of its diagonal blocks (already sitting in Z's diagonal blocks): block column Kc of Z by ONE
// wave - forward substitution by blocks,  Z_IK = -W_I sum_{K <= j < I} L_Ij Z_jK  (I = K + 1 .. NB - 1), the column's tiles in
// registers (accumulator layout = the B operand of the next product), L and W read from memory in A-operand order.  The block
// columns are independent: 16 waves per 256 x 256 matrix, each one pass over its part of L - instead of one more cooperative
// factorisation per block column (k_coop_inv: NB redundant factorisations per matrix, 3.4 ms for 256 matrices of 256).
template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_trtri(const double* __restrict__ Lall, double* __restrict__ Zall, int T, int b,
                                                      const int32_t* __restrict__ info) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int w = blockIdx.x * WAVES + wave;
  const int m = w / NB, Kc = w % NB;
  if (m >= b || 16 * Kc >= T) return;
  const double* L = Lall + (size_t)m * T * T;
  double* Z = Zall + (size_t)m * T * T;
  const int nb = (T + 15) >> 4;
  d4 Zc[NB];                                       // Z_IK, I = Kc .. nb - 1 (statically indexed: slot I)
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    if (I == Kc) {                                 // diagonal block: W_K, read back in accumulator layout v[r] = X[g + 4 r][c]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, col = 16 * I + c;
        Zc[I][r] = (row < T && col < T) ? Z[(size_t)row * T + col] : ((row == col) ? 1.0 : 0.0);
      }
    } else {
      Zc[I] = (d4){0.0, 0.0, 0.0, 0.0};
    }
  }
#pragma unroll
  for (int I = 1; I < NB; ++I) {
    if (I > Kc && I < nb) {
      d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < I; ++j) {
        if (j >= Kc) {
          double av[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {            // A operand of L_Ij: lane (g, c) holds L[16 I + c][16 j + 4 s + g]
            const int row = 16 * I + c, col = 16 * j + 4 * s + g;
            av[s] = (row < T) ? L[(size_t)row * T + col] : 0.0;
          }
          if (j & 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc1 = mfma(av[s], Zc[j][s], acc1);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc0 = mfma(av[s], Zc[j][s], acc0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
      double wv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {                // A operand of W_I (diagonal block I of Z)
        const int row = 16 * I + c, col = 16 * I + 4 * s + g;
        wv[s] = (row < T && col < T) ? Z[(size_t)row * T + col] : ((row == col) ? 1.0 : 0.0);
      }
      d4 z = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) z = mfma_sub(wv[s], acc0[s], z);      // -W_I acc
      Zc[I] = z;
    }
  }
  const bool bad = info && info[m] != 0;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + g + 4 * r, col = 16 * Kc + c;
      if (row < T && col < T && I != Kc) Z[(size_t)row * T + col] = bad ? __builtin_nan("") : ((I > Kc) ? Zc[I][r] : 0.0);
    }
  }
}

// Linv[:, 16 Jc ..] = L^{-1} e for block column Jc = blockIdx.y: factor again with the identity block as right-hand
// side (reads A, which must still hold the input: launched BEFORE k_coop_potrf on the same stream).
template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_coop_inv(PotrfArgs a) {
  using C = Coop<NB>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* rowbuf = smem;
  double* Rbuf = rowbuf + NB * 256;
  double* Wbuf = Rbuf + NB * 256;
  double* scr = Wbuf + 256;
  double* red = scr + DIAG_SCR;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int m = blockIdx.x, Jc = blockIdx.y;
  const int T = a.T;
  if (16 * Jc >= T) return;
  const double* A = a.A + (size_t)m * T * T;
  d4 U[C::NT];
  coop_load_sym_upper<NB>(U, A, T, T, wave, lane, rowbuf + wave * DIAG_SCR);
  __syncthreads();   // rowbuf served as per-wave staging for the loader
  {
    double sh = a.add;
    if (a.jitter_rel != 0.0) {
      const double dm = coop_diag_abs_mean<NB>(U, T, wave, lane, a.add, red);
      sh += a.jitter_rel * fmax(dm, F64_EPS);
    }
    if (sh != 0.0) coop_add_diag<NB>(U, sh, T, wave, lane);
  }
  for (int K = wave; K < NB; K += WAVES) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (K == Jc && g + 4 * r == c) ? 1.0 : 0.0;
    lds_tile_store(Rbuf, K, lane, v);
  }
  __syncthreads();
  PivotAcc pa;
  pa.init();
  coop_factor<NB, true>(U, rowbuf, Rbuf, Wbuf, scr, wave, lane, pa, nullptr, 0, T);
  double* Z = a.Linv + (size_t)m * T * T;
  for (int K = wave; K < NB; K += WAVES) {
    const d4 z = lds_tile_load(Rbuf, K, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (i < T && j < T) Z[(size_t)i * T + j] = (K >= Jc) ? z[r] : 0.0;
    }
  }
  if (a.inv_info && Jc == 0) {   // inverse-only call: block column 0 saw every pivot
    int info;
    (void)coop_logdet_info(pa, wave, lane, red, reinterpret_cast<int*>(red + 8), info);
    if (threadIdx.x == 0 && a.info) a.info[m] = info;
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once for every (kernel, device)
// this process launches on, under a lock (the launchers are called from any host thread).
}  // namespace

int hgp_internal_ensure_dynamic_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dv = 0;
  if (hipGetDevice(&dv) != hipSuccess) return launch_status();
  std::lock_guard<std::mutex> lk(mu);
  for (const auto& d : done)
    if (d.first == fn && d.second == dv) return 0;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return 1000 + (int)e;
  done.emplace_back(fn, dv);
  return 0;
}

namespace {

template <int NB>
int launch_coop_score(const ScoreArgs& a, hipStream_t st) {
  const size_t lds = sizeof(double) * Coop<NB>::LDS_DOUBLES;
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_score<NB>), lds)) return rc_;
  hipLaunchKernelGGL(k_coop_score<NB>, dim3(a.n_items), dim3(64 * WAVES), lds, st, a);
  return launch_status();
}

template <int NB>
int launch_coop_potrf(const PotrfArgs& a, hipStream_t st) {
  const size_t lds = sizeof(double) * Coop<NB>::LDS_DOUBLES;
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_potrf<NB>), lds)) return rc_;
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_inv<NB>), lds)) return rc_;
  hipLaunchKernelGGL(k_coop_potrf<NB>, dim3(a.b), dim3(64 * WAVES), lds, st, a);
  if (a.Linv) {   // L^-1 from L: the factor kernel left the diagonal blocks' inverses in Linv, k_trtri fills in the block columns
    const int waves = a.b * NB;
    hipLaunchKernelGGL(k_trtri<NB>, dim3((waves + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, st, a.Aout ? a.Aout : a.A, a.Linv, a.T, a.b,
                       a.info);
  }
  return launch_status();
}

template <int NB>
int launch_coop_inv_only(const PotrfArgs& a, hipStream_t st) {   // L^-1 without the in-place factor (A untouched)
  const size_t lds = sizeof(double) * Coop<NB>::LDS_DOUBLES;
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_inv<NB>), lds)) return rc_;
  hipLaunchKernelGGL(k_coop_inv<NB>, dim3(a.b, NB), dim3(64 * WAVES), lds, st, a);
  return launch_status();
}

// --------------------------------------------------------------------------- batched tile GEMM
// C[b] = alpha op(A[b]) op(B[b]) + beta C[b], any M x N x Kd, one wave per 16x16 tile of C, operands straight
// from global memory (L2).  Used for per-cluster operators and for the matrix-valued likelihood terms (a8, a9),
// never per (segment, cluster) pair.
struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int M, N, Kd, lda, ldb, ldc;
  long sA, sB, sC;
  double alpha, beta;
  int tA, tB;
  int nb2 = 1;                 // optional second batch level: item b = b1 * nb2 + b2 uses offsets b1 * s?  + b2 * s?2
  long sA2 = 0, sB2 = 0, sC2 = 0;
  const double* D = nullptr;   // optional addend (instead of C): C = alpha op(A) op(B) + beta D
  int ldd = 0;
  long sD = 0;
  int boff = 0;                // first batch item of this launch (gridDim.y is capped at 65535: larger batches go in chunks)
  int triA = 0;                // A (not transposed) is lower triangular: row tile ti only needs k < 16 (ti + 1)
};

// TRIP = k-steps whose operand loads are issued before the first MFMA of a trip.  TRIP = 24 covers Kd <= 96 in ONE
// trip (the LDS recursion's 90 x 90 products: one load latency instead of three per tile).
template <int TRIP>
__global__ __launch_bounds__(64 * WAVES) void k_gemm(GemmArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int ntn = (a.N + 15) / 16, ntm = (a.M + 15) / 16;
  const int tile = blockIdx.x * WAVES + wave;
  if (tile >= ntm * ntn) return;
  const int ti = tile / ntn, tj = tile % ntn;
  const int by = (int)blockIdx.y + a.boff;
  const int b1 = by / a.nb2, b2 = by % a.nb2;
  const double* A = a.A + (size_t)b1 * a.sA + (size_t)b2 * a.sA2;
  const double* B = a.B + (size_t)b1 * a.sB + (size_t)b2 * a.sB2;
  double* C = a.C + (size_t)b1 * a.sC + (size_t)b2 * a.sC2;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  const int row = 16 * ti + c, col = 16 * tj + c;
  int nk = (a.Kd + 3) / 4;
  if (a.triA && !a.tA) nk = min(nk, 4 * (ti + 1));
  for (int k0 = 0; k0 < nk; k0 += TRIP) {   // TRIP k-steps per trip: their 2 TRIP operand loads are issued before the first MFMA
    double av[TRIP], bv[TRIP];
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      const int k = 4 * (k0 + u) + g;
      av[u] = 0.0;
      bv[u] = 0.0;
      if (k < a.Kd) {
        if (row < a.M) av[u] = a.tA ? A[(size_t)k * a.lda + row] : A[(size_t)row * a.lda + k];
        if (col < a.N) bv[u] = a.tB ? B[(size_t)col * a.ldb + k] : B[(size_t)k * a.ldb + col];
      }
    }
#pragma unroll
    for (int u = 0; u < TRIP; ++u) acc = mfma(av[u], bv[u], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 16 * ti + g + 4 * r;
    if (i < a.M && col < a.N) {
      double v = a.alpha * acc[r];
      if (a.beta != 0.0)
        v += a.beta * (a.D ? a.D[(size_t)b1 * a.sD + (size_t)i * a.ldd + col] : C[(size_t)i * a.ldc + col]);
      C[(size_t)i * a.ldc + col] = v;
    }
  }
}

// Large products (M, N, Kd multiples of 32, Kd > 128, no transposes: the a8 / a9 compositions at 128 < T <= 256): one wave per
// 32 x 32 block of C - four accumulator tiles fed by two A and two B fragments per k-step, i.e. half the L2 operand traffic per
// MFMA of k_gemm (which is bound by exactly that traffic at these sizes: 14 TFLOP/s on the 2 T^3 product at T = 256).
__global__ __launch_bounds__(64 * WAVES) void k_gemm22(GemmArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int ntn = a.N / 32, ntm = a.M / 32;
  const int tile = blockIdx.x * WAVES + wave;
  if (tile >= ntm * ntn) return;
  const int ti = tile / ntn, tj = tile % ntn;
  const int by = (int)blockIdx.y + a.boff;
  const int b1 = by / a.nb2, b2 = by % a.nb2;
  const double* A = a.A + (size_t)b1 * a.sA + (size_t)b2 * a.sA2 + (size_t)(32 * ti + c) * a.lda + g;
  const double* B = a.B + (size_t)b1 * a.sB + (size_t)b2 * a.sB2 + (size_t)g * a.ldb + 32 * tj + c;
  double* C = a.C + (size_t)b1 * a.sC + (size_t)b2 * a.sC2;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  int nk = a.Kd / 4;
  if (a.triA) nk = min(nk, 8 * (ti + 1));   // lower-triangular A: rows 32 ti .. 32 ti + 31 only reach k < 32 (ti + 1)
  constexpr int TR = 8;                      // k-steps per trip (Kd is a multiple of 32 here)
  for (int k0 = 0; k0 < nk; k0 += TR) {
    double av[TR][2], bv[TR][2];
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const size_t k = 4 * (size_t)(k0 + u);
      av[u][0] = A[k];
      av[u][1] = A[k + 16 * (size_t)a.lda];
      bv[u][0] = B[k * a.ldb];
      bv[u][1] = B[k * a.ldb + 16];
    }
#pragma unroll
    for (int u = 0; u < TR; ++u) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma(av[u][i], bv[u][j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * ti + 16 * i + g + 4 * r, col = 32 * tj + 16 * j + c;
        double v = a.alpha * acc[i][j][r];
        if (a.beta != 0.0) v += a.beta * (a.D ? a.D[(size_t)b1 * a.sD + (size_t)row * a.ldd + col] : C[(size_t)row * a.ldc + col]);
        C[(size_t)row * a.ldc + col] = v;
      }
}

int launch_gemm(const GemmArgs& a0, int batch, hipStream_t st) {
  const int nt = ((a0.M + 15) / 16) * ((a0.N + 15) / 16);
  for (int b0 = 0; b0 < batch; b0 += 65535) {   // gridDim.y <= 65535
    GemmArgs a = a0;
    a.boff = b0;
    const int nb = std::min(65535, batch - b0);
    // (only for launches that fill the chip: a single 256^3 product is 64 waves here against 256 in k_gemm - latency-bound,
    // 25.8 vs ~12 us in the member step of the online path at T = 256)
    if (!a.tA && !a.tB && a.M % 32 == 0 && a.N % 32 == 0 && a.Kd % 32 == 0 && a.Kd > 128 &&
        (long)(a.M / 32) * (a.N / 32) * nb >= 2048 && !env_on("HGP_GEMM_PLAIN")) {
      const int nt2 = (a.M / 32) * (a.N / 32);
      hipLaunchKernelGGL(k_gemm22, dim3((nt2 + WAVES - 1) / WAVES, nb), dim3(64 * WAVES), 0, st, a);
      continue;
    }
    if (a.Kd <= 96 && a.Kd > 32)
      hipLaunchKernelGGL(k_gemm<24>, dim3((nt + WAVES - 1) / WAVES, nb), dim3(64 * WAVES), 0, st, a);
    else if (a.Kd <= 128 && a.Kd > 96)
      hipLaunchKernelGGL(k_gemm<32>, dim3((nt + WAVES - 1) / WAVES, nb), dim3(64 * WAVES), 0, st, a);
    else
      hipLaunchKernelGGL(k_gemm<8>, dim3((nt + WAVES - 1) / WAVES, nb), dim3(64 * WAVES), 0, st, a);
  }
  return launch_status();
}

// out[b] = scale * sum_i X[b][i] * Y[b][i]  (+ out[b] if accumulate)
__global__ __launch_bounds__(256) void k_dot_batched(const double* __restrict__ X, const double* __restrict__ Y, long sX,
                                                      long sY, long n, double scale, int accumulate, double* out) {
  __shared__ double red[256];
  const double* x = X + (size_t)blockIdx.x * sX;
  const double* y = Y + (size_t)blockIdx.x * sY;
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) s = fma(x[i], y[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (accumulate ? out[blockIdx.x] : 0.0) + scale * red[0];
}

// out[b] (+)= scale * sum_ij Z[b][i][j]^2 S[j][j]: trace(Sigma^-1 S) for a DIAGONAL prior scale S from Z = L^-1 alone
__global__ __launch_bounds__(256) void k_colnorm_diag(const double* __restrict__ Z, const double* __restrict__ S, long sS, int T,
                                                       double scale, int accumulate, double* out) {
  __shared__ double red[256];
  const double* z = Z + (size_t)blockIdx.x * T * T;
  const double* sd = S + (size_t)blockIdx.x * sS;
  double s = 0.0;
  for (long i = threadIdx.x; i < (long)T * T; i += 256) {
    const int r = (int)(i / T), cidx = (int)(i % T);
    if (cidx <= r) {
      const double v = z[i];
      s = fma(v * v, sd[(size_t)cidx * T + cidx], s);
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (accumulate ? out[blockIdx.x] : 0.0) + scale * red[0];
}

// C[b] = A[b] - B[b]  (elementwise, n per item)
__global__ void k_sub_batched(const double* __restrict__ A, const double* __restrict__ B, long sA, long sB, long n,
                              double* __restrict__ C, int b) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n) return;
  for (int m = blockIdx.y; m < b; m += gridDim.y) C[(size_t)m * n + i] = A[(size_t)m * sA + i] - B[(size_t)m * sB + i];
}

// r[b] = f_cur[b] - A[b] f_prev[b]   (a8 residual), one workgroup per item
__global__ __launch_bounds__(256) void k_lat_resid(const double* __restrict__ f_cur, const double* __restrict__ f_prev,
                                                    const double* __restrict__ A, int T, double* __restrict__ r) {
  const int b = blockIdx.x;
  const double* Ab = A + (size_t)b * T * T;
  const double* fp = f_prev + (size_t)b * T;
  for (int i = threadIdx.x; i < T; i += 256) {
    double s = 0.0;
    for (int k = 0; k < T; ++k) s = fma(Ab[(size_t)i * T + k], fp[k], s);
    r[(size_t)b * T + i] = f_cur[(size_t)b * T + i] - s;
  }
}

// a11: omega^2 exp(-0.5 dx^2 / rho^2) + diag_add I on the (optionally [0,1]-normalised) grid
__global__ void k_warp_cov(const double* __restrict__ x, int T, double rho, double omega, double diag_add, int normalize,
                           double* __restrict__ K) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)T * T) return;
  const int i = (int)(idx / T), j = (int)(idx % T);
  double xi = x[i], xj = x[j];
  if (normalize) {
    const double x0 = x[0];
    const double rng = fabs((x[T - 1] - x0) - (x0 - x0)) + 1e-12;   // amtgp_warping_system.py:163-166
    xi = (xi - x0) / rng;
    xj = (xj - x0) / rng;
  }
  const double dx = xi - xj;
  double v = (omega * omega) * exp(-0.5 * (dx * dx) / (rho * rho));
  if (i == j) v += diag_add;
  K[idx] = v;
}

// config 5: L <- chol(alpha L L^T + beta v v^T) by a rank-1 update, O(T^2) instead of O(T^3); T <= 256.
// One workgroup per matrix, thread i owns ROW i of L and x_i.  Columns are processed in blocks of 16:
//   - every thread still below the block loads its 16-entry row segment (the next block's segment is requested
//     before the current one is processed);
//   - the wave that holds the 16 pivot rows runs the 16 dependent steps  r = hypot(l_kk, x_k), c = r / l_kk,
//     s = x_k / l_kk  with the pivot row's values broadcast by v_readlane, applying each rotation to all of its
//     own rows on the way (one rsqrt on the chain per step; the reciprocals 1 / l_kk are taken off the chain);
//   - the 16 rotations go to LDS (double-buffered, one barrier per block) and the other waves apply them to their rows.
// Row segments are read and written once: 8 T^2 bytes of HBM traffic per update (lower triangle in and out).
struct Rank1Args {
  double* L;
  const double* v;
  const double* alpha;
  const double* beta;
  int T, b;
  int32_t* info;
  unsigned long long* stamps;   // diagnostic builds (HGP_STAMPS) only
};

// COAL (T even): the row segments move as full 128-byte lines - lane l of a wave takes 16 bytes (chunk l & 7) of row 8 u + (l >> 3),
// u = 0 .. 7, so one load / store instruction covers 8 complete lines instead of 16 bytes of 64 different ones - and are
// transposed to "thread i owns row i" through a per-wave LDS tile.  (The direct form re-fetched every line eight times: eight
// waves x 64 lines x 128 bytes per block do not stay in the 32 KB L1.)
constexpr int R1_LD = 18;   // doubles per row of the transpose tile (16 + 2: 144-byte stride)
template <bool COAL>
__global__ __launch_bounds__(256, 2) void k_chol_rank1(Rank1Args a) {
  __shared__ double rot[2][16][4];   // (1/c, s, c) of the 16 steps of a block
  __shared__ __attribute__((aligned(16))) double tile_all[COAL ? 4 * 64 * R1_LD : 2];
  __shared__ int s_info;
  const int i = threadIdx.x, lane = i & 63;
  const int wave = __builtin_amdgcn_readfirstlane(i >> 6);
  const int m = blockIdx.x;
  const int T = a.T;
  double* L = a.L + (size_t)m * T * T;
  const double al = a.alpha ? a.alpha[m] : 1.0, be = a.beta ? a.beta[m] : 1.0;
  const double sa = sqrt(al), sb = sqrt(be);
  double x = (i < T) ? sb * a.v[(size_t)m * T + i] : 0.0;
  if (i == 0) s_info = (al > 0.0 && be >= 0.0) ? 0 : -1;
  int info = 0;
  const int nblk = (T + 15) >> 4;
  double cur[16], nxt[16];
  double* tile = tile_all + (COAL ? wave * 64 * R1_LD : 0);
  const int q8 = lane >> 3, ch = lane & 7;          // COAL: my sub-row and 16-byte chunk
  // COAL: block kc of my wave's 64 rows -> nxt[2 u], nxt[2 u + 1] = columns 16 kc + 2 ch, + 1 of row 64 wave + 8 u + q8
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = 64 * wave + 8 * u + q8, col = 16 * kc + 2 * ch;
      double2 v = make_double2(0.0, 0.0);   // (branches: lanes with nothing to fetch issue nothing - faster here than the branch-free
      if (row < T && row >= 16 * kc && col <= row && col + 1 < T) v = *reinterpret_cast<const double2*>(L + (size_t)row * T + col);   // form, 0.051 vs 0.063 ms at T = 128)
      else if (row < T && row >= 16 * kc && col <= row && col < T) v.x = L[(size_t)row * T + col];
      nxt[2 * u] = v.x;
      nxt[2 * u + 1] = v.y;
    }
  };
  auto to_rows = [&](int kc) {                      // nxt (line layout) -> cur (thread i owns row i), upper part zero
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<double2*>(tile + (8 * u + q8) * R1_LD + 2 * ch) = make_double2(nxt[2 * u], nxt[2 * u + 1]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = (16 * kc + j <= i) ? tile[lane * R1_LD + j] : 0.0;
    __builtin_amdgcn_wave_barrier();
  };
  if constexpr (COAL) {
    fetch(0);
    to_rows(0);
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = (i < T && j < T && j <= i) ? L[(size_t)i * T + j] : 0.0;
  }
  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = 16 * kb, k1 = k0 + 16;
    const bool below = i < T && i >= k0;            // rows above the block are final
    if (kb + 1 < nblk) {
      if constexpr (COAL) {
        fetch(kb + 1);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) nxt[j] = (i < T && i >= k1 && k1 + j <= i) ? L[(size_t)i * T + k1 + j] : 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] *= sa;
    const int wd = k0 >> 6;                         // the wave that owns the pivot rows k0 .. k0 + 15
    double (*R)[4] = rot[kb & 1];
    if (wave == wd) {
      const int pl0 = k0 & 63;
      // reciprocal of MY pivot entry (lane pl0 + j holds l_jj in cur[j]), off the dependent chain
      double mine = 1.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) mine = (lane == pl0 + j) ? cur[j] : mine;
      const double myinv = 1.0 / mine;
      double rc[16][3];
#pragma unroll
      for (int k = 0; k < 16; ++k) {   // (columns >= T of the last block hold zeros: l_kk = 0 -> NaN rotations that no row uses)
        const int pl = pl0 + k;
        const double lkk = lane_bcast(cur[k], pl), xk = lane_bcast(x, pl), ilkk = lane_bcast(myinv, pl);
        const double t = fma(lkk, lkk, xk * xk);
        const double rinv = rsqrt_nr(t);
        const double r = t * rinv, cinv = lkk * rinv, c = r * ilkk, sn = xk * ilkk;
        const bool live = k0 + k < T;
        info = (live && !(r > 0.0) && info == 0) ? k0 + k + 1 : info;
        // selects, not branches: an exec-mask change per step would sit on the dependent chain
        const bool upd = live && lane > pl && below;
        const double ln = fma(sn, x, cur[k]) * cinv;
        const double xn = fma(c, x, -sn * ln);
        x = upd ? xn : x;
        cur[k] = (live && lane == pl) ? r : (upd ? ln : cur[k]);
        rc[k][0] = cinv;
        rc[k][1] = sn;
        rc[k][2] = c;
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          R[k][0] = rc[k][0];
          R[k][1] = rc[k][1];
          R[k][2] = rc[k][2];
        }
      }
    }
    __syncthreads();
    if (wave > wd && below) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k0 + k < T) {
          const double cinv = R[k][0], sn = R[k][1], c = R[k][2];
          const double ln = fma(sn, x, cur[k]) * cinv;
          x = fma(c, x, -sn * ln);
          cur[k] = ln;
        }
      }
    }
    if constexpr (COAL) {
      if (64 * wave + 63 >= k0) {                     // (wave-uniform) some of my rows are at or below the block
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[lane * R1_LD + j] = cur[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = 64 * wave + 8 * u + q8, col = k0 + 2 * ch;
          const double2 v = *reinterpret_cast<const double2*>(tile + (8 * u + q8) * R1_LD + 2 * ch);
          if (row < T && row >= k0 && col + 1 <= row) *reinterpret_cast<double2*>(L + (size_t)row * T + col) = v;
          else if (row < T && row >= k0 && col <= row) L[(size_t)row * T + col] = v.x;
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (kb + 1 < nblk) to_rows(kb + 1);
    } else {
      if (below) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (k0 + j <= i) L[(size_t)i * T + k0 + j] = cur[j];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[j] = nxt[j];
    }
  }
  if (info != 0 && lane == 0) {                 // the earliest bad pivot over the waves wins (a bad alpha/beta stays -1)
    const int old = atomicCAS(&s_info, 0, info);
    if (old > info) atomicMin(&s_info, info);
  }
  __syncthreads();
  if (i == 0 && a.info) a.info[m] = s_info;
}

// The pipelined form (T even, shipped).  What k_chol_rank1<true> does per block - pivot chain, barrier, everybody applies the 16
// rotations, stores, transposes the next segment - is one serial 12 k cycles (5 us) per block: 83 us per factor with the CU to
// itself (in-kernel stamps, tools/stamps_rank1.py).  But block kb's chain needs nothing from block kb - 1 except x of ITS OWN
// sixteen rows, which the pivot wave updates itself inside the chain; the fresh columns of L do not depend on earlier rotations.
// So in iteration kb the pivot wave runs chain(kb) while the other waves apply the rotations of block kb - 1 (published at the
// previous barrier; one buffer per block, no reuse hazard), store that block and bring in the next one; the pivot wave does its own
// loads and stores BEHIND its chain.  Critical path per block: two LDS transposes + the 16 dependent steps + one barrier.
// Same arithmetic per element as the other two forms: results identical bit for bit.
__global__ __launch_bounds__(256, 2) void k_chol_rank1_pipe(Rank1Args a) {
  __shared__ __attribute__((aligned(16))) double rot[16][16][4];   // [block][step] (1/c, s, c, -)
  __shared__ __attribute__((aligned(16))) double tile_all[4 * 64 * R1_LD];
  __shared__ int s_info;
  const int i = threadIdx.x, lane = i & 63;
  const int wave = __builtin_amdgcn_readfirstlane(i >> 6);
  const int m = blockIdx.x;
  const int T = a.T;
  double* L = a.L + (size_t)m * T * T;
  const double al = a.alpha ? a.alpha[m] : 1.0, be = a.beta ? a.beta[m] : 1.0;
  const double sa = sqrt(al), sb = sqrt(be);
  double x = (i < T) ? sb * a.v[(size_t)m * T + i] : 0.0;
  if (i == 0) s_info = (al > 0.0 && be >= 0.0) ? 0 : -1;
  int info = 0;
  const int nblk = (T + 15) >> 4;
  double cur[16];
  double2 ln8[8];   // eight 16-byte pieces in line layout: the fetched block on its way in
  double* tile = tile_all + wave * 64 * R1_LD;
  const int q8 = lane >> 3, ch = lane & 7;
  const int last = min(nblk - 1, (64 * wave + 63) >> 4);   // the last block that reaches rows of this wave
  // piece u of a block: columns 16 kc + 2 ch, + 1 of row 64 wave + 8 u + q8
  size_t off[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) off[u] = (size_t)min(64 * wave + 8 * u + q8, T - 1) * T + 2 * ch;
  // no branches: a lane with nothing to fetch reads L[0] (a conditional load is waited for at once - eight serial round trips)
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = 64 * wave + 8 * u + q8, col = 16 * kc + 2 * ch;
      const bool on = row < T && row >= 16 * kc && col <= row && col < T;      // (T even: col + 1 < T as well)
      const double2 v = *reinterpret_cast<const double2*>(L + (on ? off[u] + 16 * kc : (size_t)0));
      ln8[u] = on ? v : make_double2(0.0, 0.0);
    }
  };
  // ln8 -> cur (thread i owns row i) scaled by sqrt(alpha).  Entries right of the diagonal inside a fetched piece are whatever the
  // caller's upper triangle holds: no step reads them (rotation k touches rows > k only) and put_rows never stores them
  auto to_rows = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<double2*>(tile + (8 * u + q8) * R1_LD + 2 * ch) = ln8[u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) cur[j] = tile[lane * R1_LD + j] * sa;
    __builtin_amdgcn_wave_barrier();
  };
  double2 out8[8];   // the finished block on its way out
  auto rows_to_lines = [&]() {   // cur (final) -> out8
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[lane * R1_LD + j] = cur[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 8; ++u) out8[u] = *reinterpret_cast<const double2*>(tile + (8 * u + q8) * R1_LD + 2 * ch);
    __builtin_amdgcn_wave_barrier();
  };
  auto put_rows = [&](int kc) {
    const int k0 = 16 * kc;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = 64 * wave + 8 * u + q8, col = k0 + 2 * ch;
      if (row < T && row >= k0 && col + 1 <= row) *reinterpret_cast<double2*>(L + off[u] + k0) = out8[u];
      else if (row < T && row >= k0 && col <= row) L[off[u] + k0] = out8[u].x;
    }
  };
#ifdef HGP_STAMPS   // where do the iterations go?  per wave: [0] apply + transposes, [1] chain, [2] loads / stores issued, [3] barrier wait
  unsigned long long st_acc[4] = {0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#define HGP_R1(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_readcyclecounter(); st_acc[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define HGP_R1(i)
#endif
  fetch(0);
  to_rows();
  if (1 <= last) fetch(1);
  __syncthreads();   // s_info
  HGP_R1(0);
#pragma nounroll
  for (int kb = 0; kb <= nblk; ++kb) {
    const bool have_prev = kb >= 1 && kb - 1 <= last;
    // A. block kb - 1: apply its rotations (its pivot wave did so inside the chain), turn it into lines; block kb into rows
    if (have_prev) {
      const int kp = kb - 1, wdp = (16 * kp) >> 6;
      if (wave > wdp) {
        const double4* R = reinterpret_cast<const double4*>(&rot[kp][0][0]);
        double4 rr[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) rr[k] = R[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (16 * kp + k < T) {   // (uniform)
            const double lnv = fma(rr[k].y, x, cur[k]) * rr[k].x;
            x = fma(rr[k].z, x, -rr[k].y * lnv);
            cur[k] = lnv;
          }
        }
      }
      rows_to_lines();
      if (kb <= last) to_rows();
    }
    HGP_R1(0);
    const bool pivot = kb < nblk && wave == ((16 * kb) >> 6);
    if (!pivot && have_prev) {
      put_rows(kb - 1);
      if (kb + 1 <= last) fetch(kb + 1);
      HGP_R1(2);
    }
    // B. the pivot chain of block kb
    if (pivot) {
      const int k0 = 16 * kb, pl0 = k0 & 63;
      double (*R)[4] = rot[kb];
      double mine = 1.0;   // reciprocal of MY pivot entry (lane pl0 + j holds l_jj in cur[j]), off the dependent chain
#pragma unroll
      for (int j = 0; j < 16; ++j) mine = (lane == pl0 + j) ? cur[j] : mine;
      const double myinv = 1.0 / mine;
      const unsigned rbase = (unsigned)(unsigned long long)(&R[0][0]);   // LDS offset = low half of the flat address
#pragma unroll
      for (int k = 0; k < 16; ++k) {   // (columns >= T of the last block hold zeros: l_kk = 0 -> NaN rotations that no row uses)
        const int pl = pl0 + k;
        const double lkk = lane_bcast(cur[k], pl), xk = lane_bcast(x, pl), ilkk = lane_bcast(myinv, pl);
        const double t = fma(lkk, lkk, xk * xk);
        const double rinv = rsqrt_nr(t);
        const double r = t * rinv, cinv = lkk * rinv, c = r * ilkk, sn = xk * ilkk;
        const bool live = k0 + k < T;
        info = (live && !(r > 0.0) && info == 0) ? k0 + k + 1 : info;
        const bool upd = live && lane > pl && i < T;
        const double lnv = fma(sn, x, cur[k]) * cinv;
        const double xn = fma(c, x, -sn * lnv);
        x = upd ? xn : x;
        cur[k] = (live && lane == pl) ? r : (upd ? lnv : cur[k]);
        // lane 0 publishes the rotation (the same number in every lane): three stores under exec = 1, in assembly - the compiler's
        // form of "if (lane < 2) store" (two selects per value, s_and_saveexec, branch) was 1.8 k of the 5.3 k cycles of a block,
        // and keeping the values by selects for one store per block measured slower still (0.190 vs 0.182 ms)
        {
          unsigned long long ex_;
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\t"
                       "ds_write_b64 %1, %2 offset:%5\n\tds_write_b64 %1, %3 offset:%6\n\tds_write_b64 %1, %4 offset:%7\n\t"
                       "s_mov_b64 exec, %0"
                       : "=&s"(ex_)
                       : "v"(rbase), "v"(cinv), "v"(sn), "v"(c), "n"(32 * k), "n"(32 * k + 8), "n"(32 * k + 16)
                       : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stores above are invisible to the compiler's counters
      HGP_R1(1);
      if (have_prev) put_rows(kb - 1);
      if (kb + 1 <= last) fetch(kb + 1);
      HGP_R1(2);
    }
    __syncthreads();
    HGP_R1(3);
  }
#ifdef HGP_STAMPS
  if (a.stamps && m == 0 && lane == 0)
    for (int q = 0; q < 4; ++q) atomicAdd(&a.stamps[4 * wave + q], st_acc[q]);
#endif
#undef HGP_R1
  if (info != 0 && lane == 0) {                 // the earliest bad pivot over the waves wins (a bad alpha/beta stays -1)
    const int old = atomicCAS(&s_info, 0, info);
    if (old > info) atomicMin(&s_info, info);
  }
  __syncthreads();
  if (i == 0 && a.info) a.info[m] = s_info;
}

// ------------------------------------------------------------------ 8f-1: glue of the LDS chain step, fused
// The captured per-member step (hdpgpc_amd/GPI_model.py: _chain_step) is a chain of small GEMMs and inverses; what sits
// between them was ~75 element-wise / index launches of 3-5 us each.  Two kernels replace most of them.
//
// k_chain_gather: rows `pos` of the eight state stacks -> one contiguous workspace (A, G, C, S, Psm, P [T,T]; F, Fsm [T]),
// and the observation of the member this step includes, Y[pos - y_row0], into y_out.
struct ChainGatherArgs {
  const double* st[8];   // A, G, C, S, Psm, P (T*T each), F, Fsm (T each)
  const int64_t* pos;
  double* out;           // [6 T T + 2 T]
  int T;
  const double* Y;       // [n,T] observations of the run (may be NULL)
  long y_row0;
  double* y_out;         // [T]
};

__global__ __launch_bounds__(256) void k_chain_gather(ChainGatherArgs a) {
  const long tt = (long)a.T * a.T, p = a.pos[0];
  const long total = 6 * tt + 2 * a.T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    double v;
    if (i < 6 * tt) {
      const int w = (int)(i / tt);
      v = a.st[w][p * tt + (i - w * tt)];
    } else {
      const long j = i - 6 * tt;
      const int w = 6 + (int)(j / a.T);
      v = a.st[w][p * a.T + (j - (long)(w - 6) * a.T)];
    }
    a.out[i] = v;
  }
  if (a.Y && blockIdx.x == 0)
    for (int i = threadIdx.x; i < a.T; i += 256) a.y_out[i] = a.Y[(p - a.y_row0) * a.T + i];
}

// k_chain_scatter: the new filtered state and the re-smoothed previous one into the stacks (rows pos + 1 and pos).
struct ChainScatterArgs {
  const double* f_post;     // [T]
  const double* c_post;     // [T,T]
  const double* f_sm_prev;  // [T]
  const double* P_sm_prev;  // [T,T]
  double* stF;
  double* stFsm;
  double* stP;
  double* stPsm;
  const int64_t* pos;
  int T;
};

__global__ __launch_bounds__(256) void k_chain_scatter(ChainScatterArgs a) {
  const long tt = (long)a.T * a.T, p = a.pos[0], nx = p + 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tt; i += (long)gridDim.x * 256) {
    const double c = a.c_post[i];
    a.stP[nx * tt + i] = c;
    a.stPsm[nx * tt + i] = c;
    a.stPsm[p * tt + i] = a.P_sm_prev[i];
    if (i < a.T) {
      const double f = a.f_post[i];
      a.stF[nx * a.T + i] = f;
      a.stFsm[nx * a.T + i] = f;
      a.stFsm[p * a.T + i] = a.f_sm_prev[i];
    }
  }
}

// out[b] = R[b] + factor * max(mean |diag S[b]|, eps) I   (the jitter of matrix_normal_inv_wishart.posterior,
// GPI_model.py:1312-1316, taken from the CURRENT scale matrix)
__global__ __launch_bounds__(256) void k_add_diag_mean(const double* __restrict__ R, const double* __restrict__ S, int T,
                                                       double factor, double* __restrict__ out) {
  __shared__ double red[256];
  const long tt = (long)T * T;
  const double* Sb = S + (size_t)blockIdx.x * tt;
  double s = 0.0;
  for (int i = threadIdx.x; i < T; i += 256) s += fabs(Sb[(size_t)i * T + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double jit = factor * fmax(red[0] / T, F64_EPS);
  const double* Rb = R + (size_t)blockIdx.x * tt;
  double* ob = out + (size_t)blockIdx.x * tt;
  for (long i = threadIdx.x; i < tt; i += 256) ob[i] = Rb[i] + ((i / T == i % T) ? jit : 0.0);
}

// k_rts_chain: the sequential part of the RTS smoother (GPI.backward, GPI.py:240-270) for ALL steps in one launch.
// The gains J_t, the predictive covariances P_t and A_t m_t only depend on the filtered states and are batched by the
// caller; what remains is, for t = n-2 .. 0:
//     m_t <- m_t + J_t (m_{t+1} - A_t m_t),      C_t <- C_t + J_t (C_{t+1} - P_t) J_t^T.
// One workgroup of 12 waves walks the chain; wave w owns the output tiles w, w + 12, w + 24 of the 6 x 6 tile grid.
// What crosses a step stays on chip: the new C_t tiles are still in the registers of the wave that formed them when the
// next step needs them as C_{t+1} (D = C_{t+1} - P_t is formed tile-wise in registers and written to LDS), m_t sits in
// LDS; and everything the NEXT step reads from memory (J, P, C, A m, m of step t - 1: filtered quantities, independent of
// the recursion) is requested before the two product phases of the current step.  Per step: stage J and D in LDS (row
// pitch 100 doubles), X = J D on the matrix core (kept in registers until D is dead, then written over it), C_t + X J^T.
// The floor is the matrix core of ONE compute unit: 2 x 36 tiles x 24 MFMAs x 64 cycles / 4 SIMDs = 27.6 k cycles per step.
struct RtsArgs {
  const double* J;    // [n-1,T,T]
  const double* P;    // [n-1,T,T]
  const double* AM;   // [n-1,T]
  double* M;          // [n,T]   in/out
  double* Cv;         // [n,T,T] in/out
  int n, T;
};

constexpr int RTS_WAVES = 12;   // three waves per SIMD: the LDS operand latency of one hides under the MFMAs of the others

__global__ __launch_bounds__(64 * RTS_WAVES) void k_rts_chain(RtsArgs a) {
  constexpr int NBR = 6, PITCH = 100;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Jl = smem;                    // [96][PITCH]
  double* Dl = Jl + 96 * PITCH;         // [96][PITCH]  D, then X
  double* vl = Dl + 96 * PITCH;         // [96]  m_{t+1} - A_t m_t
  double* ml = vl + 96;                 // [96]  m_{t+1} (smoothed), then m_t
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  const long tt = (long)T * T;
  const int kpad = (T + 3) & ~3;          // k-steps beyond T multiply the zero padding
  for (int i = tid; i < 96 * PITCH; i += 64 * RTS_WAVES) {   // zero padding once (rows/cols >= T are never written below)
    Jl[i] = 0.0;
    Dl[i] = 0.0;
  }
  if (tid < 96) {
    vl[tid] = 0.0;
    ml[tid] = (tid < T) ? a.M[(size_t)(a.n - 1) * T + tid] : 0.0;
  }
  // my tiles (I, Jc) = (wave / 2, 3 (wave % 2) + i), i = 0 .. 2, of a row-major [T,T] matrix, accumulator layout, zero outside
  auto load_tiles = [&](const double* __restrict__ X, d4 (&v)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int I = wave >> 1, Jc = 3 * (wave & 1) + i;   // my three tiles share block row I (one A operand per k-step for three MFMAs)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, col = 16 * Jc + c;
        v[i][r] = (row < T && col < T) ? X[(size_t)row * T + col] : 0.0;
      }
    }
  };
  // rows wave, wave + 12, ... of J (8 rows per wave), columns lane and lane + 64
  auto load_rows = [&](const double* __restrict__ X, double (&v)[8][2]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = wave + RTS_WAVES * u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        v[u][h] = (r < T && q < T) ? X[(size_t)r * T + q] : 0.0;
      }
    }
  };
  d4 cn[3], pt[3], ct[3];
  double jv[8][2];
  double am = 0.0, mt = 0.0;
  load_tiles(a.Cv + (size_t)(a.n - 1) * tt, cn);          // C_{n-1}: the last filtered state is its own smoothed state
  {
    const int t = a.n - 2;
    load_rows(a.J + (size_t)t * tt, jv);
    load_tiles(a.P + (size_t)t * tt, pt);
    load_tiles(a.Cv + (size_t)t * tt, ct);
    if (tid < T) {
      am = a.AM[(size_t)t * T + tid];
      mt = a.M[(size_t)t * T + tid];
    }
  }
  __syncthreads();
  for (int t = a.n - 2; t >= 0; --t) {
    // ---- stage J_t (rows) and D = C_{t+1} - P_t (my tiles), v = m_{t+1} - A_t m_t
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = wave + RTS_WAVES * u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        if (r < T && q < T) Jl[r * PITCH + q] = jv[u][h];
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int I = wave >> 1, Jc = 3 * (wave & 1) + i;   // my three tiles share block row I (one A operand per k-step for three MFMAs)
#pragma unroll
      for (int r = 0; r < 4; ++r) Dl[(16 * I + g + 4 * r) * PITCH + 16 * Jc + c] = cn[i][r] - pt[i][r];
    }
    const double mt_cur = mt;
    if (tid < T) vl[tid] = ml[tid] - am;
    __syncthreads();
    // ---- requests of step t - 1 (filtered quantities only): in flight under both product phases
    if (t > 0) {
      load_rows(a.J + (size_t)(t - 1) * tt, jv);
      load_tiles(a.P + (size_t)(t - 1) * tt, pt);
      if (tid < T) {
        am = a.AM[(size_t)(t - 1) * T + tid];
        mt = a.M[(size_t)(t - 1) * T + tid];
      }
    }
    // ---- X = J D
    d4 X[3];
    {
      const int I = wave >> 1, J0 = 3 * (wave & 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) X[i] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
      for (int k = 0; k < kpad; k += 4) {
        const double av = Jl[(16 * I + c) * PITCH + k + g];
#pragma unroll
        for (int i = 0; i < 3; ++i) X[i] = mfma(av, Dl[(k + g) * PITCH + 16 * (J0 + i) + c], X[i]);
      }
    }
    // ---- m_t += J v : eight lanes per row, 12 columns each, summed inside the 8-lane group
    double mnew = 0.0;
    {
      const int row = tid >> 3, part = tid & 7;
      double sv = 0.0;
      if (row < T)
        for (int j = part; j < T; j += 8) sv = fma(Jl[row * PITCH + j], vl[j], sv);
      sv += __shfl_xor(sv, 1, 64);
      sv += __shfl_xor(sv, 2, 64);
      sv += __shfl_xor(sv, 4, 64);
      mnew = sv;                             // valid in the lanes with part == 0
    }
    __syncthreads();                         // every wave has finished reading D, v and m_{t+1}
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int I = wave >> 1, Jc = 3 * (wave & 1) + i;   // my three tiles share block row I (one A operand per k-step for three MFMAs)
#pragma unroll
      for (int r = 0; r < 4; ++r) Dl[(16 * I + g + 4 * r) * PITCH + 16 * Jc + c] = X[i][r];
    }
    if ((tid & 7) == 0 && (tid >> 3) < T) vl[tid >> 3] = mnew;        // J v, row-indexed (v is dead)
    __syncthreads();
    if (tid < T) {                           // m_t = (filtered m_t) + J v: out to memory, and kept for the next step
      const double m = mt_cur + vl[tid];
      ml[tid] = m;
      a.M[(size_t)t * T + tid] = m;
    }
    // ---- C_t = (filtered C_t) + X J^T : the result stays in cn for the next step
    double* Ct = a.Cv + (size_t)t * tt;
    {
      const int I = wave >> 1, J0 = 3 * (wave & 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) cn[i] = ct[i];
#pragma unroll 8
      for (int k = 0; k < kpad; k += 4) {
        const double av = Dl[(16 * I + c) * PITCH + k + g];
#pragma unroll
        for (int i = 0; i < 3; ++i) cn[i] = mfma(av, Jl[(16 * (J0 + i) + c) * PITCH + k + g], cn[i]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + g + 4 * r, col = 16 * (J0 + i) + c;
          if (row < T && col < T) Ct[(size_t)row * T + col] = cn[i][r];
        }
      }
    }
    if (t > 0) load_tiles(a.Cv + (size_t)(t - 1) * tt, ct);          // consumed one whole step later
    __syncthreads();                         // LDS free for the next step; m_t visible
  }
}

// k_chain_finish: matrix_normal_inv_wishart.posterior's element-wise tail for BOTH updates (GPI_model.py:1326-1336),
// the keep-previous rule on a failed factorisation (GPI_model.py:1068-1071), the annealed scales
// (GPI_model.py:1083-1091), the append of A, Gamma, C, Sigma and the counters - one workgroup, one launch.
struct ChainFinishArgs {
  int T;
  const double* part;     // [2,T,T]  S_ S__^{-1}
  const double* ee;       // [2,T,T]  (y1 - y2)(y1 - y2)^T
  const double* Snew;     // [2,T,T]  S__ (the new right covariance)
  const int32_t* info1;   // [2]
  const int32_t* info2;   // [2]
  const int32_t* info0;   // [2] or NULL: status of the Kalman / pair-smoother factorisations of this step
  double* W;              // [3,2,T,T] means, R, scales (in/out)
  double* n0;             // device scalars (in/out)
  double* Nf;
  int32_t* bad_count;
  double* stA;            // stacks [L,T,T]: row pos + 1 is written
  double* stG;
  double* stC;
  double* stS;
  int64_t* pos;           // in/out: += 1
  int annealing;
  int32_t* sync;          // one zero-initialised counter (left zero)
};

#pragma clang fp contract(off)   // the reference's op order, no fused multiply-adds
__global__ __launch_bounds__(256) void k_chain_finish(ChainFinishArgs a) {
  __shared__ int last;
  const long tt = (long)a.T * a.T;
  const bool bad = (a.info1[0] | a.info1[1] | a.info2[0] | a.info2[1]) != 0;
  const double n0 = a.n0[0], Nf = a.Nf[0] + 1.0;
  const long nxt = a.pos[0] + 1;
  const double n0n = bad ? n0 : n0 + 1.0;
  const double scl = n0n / (n0n - 2.0);
  const double ann = a.annealing ? 1.0 / (Nf * Nf) : 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2 * tt; i += (long)gridDim.x * 256) {
    double m = a.W[i], r = a.W[2 * tt + i], sc = a.W[4 * tt + i];
    if (!bad) {
      m = ((n0 - 2.0) * m + a.part[i]) / (n0 - 1.0);
      r = a.Snew[i];
      sc = ((n0 - 2.0) * sc + a.ee[i]) / (n0 - 1.0);
      a.W[i] = m;
      a.W[2 * tt + i] = r;
      a.W[4 * tt + i] = sc;
    }
    const bool obs = i >= tt;           // item 0 = internal (A, Gamma), item 1 = observation (C, Sigma)
    const long e = obs ? i - tt : i;
    (obs ? a.stC : a.stA)[nxt * tt + e] = m;
    double* sg = obs ? a.stS : a.stG;
    sg[nxt * tt + e] = sc * scl + sg[e] * ann;
  }
  // the scalars are rewritten by whichever block finishes last: every block has read them by then
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = (atomicAdd(a.sync, 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    a.n0[0] = n0n;
    a.Nf[0] = Nf;
    a.bad_count[0] += bad ? 1 : 0;
    // bad_count[1]: 1-based index of the first step whose Kalman / smoother factorisation failed (0 = none) - the info
    // tensors themselves are overwritten by every graph replay
    if (a.info0 && a.bad_count[1] == 0 && (a.info0[0] | a.info0[1]) != 0) a.bad_count[1] = (int32_t)nxt;
    a.pos[0] = nxt;
    a.sync[0] = 0;
  }
}
#pragma clang fp contract(on)

// a10 (reference as written, GPI.py:1043): || G^{-1} y ||^2 with G = tril(K) used as if it were a Cholesky factor.
// One workgroup, column-oriented forward substitution in LDS; T <= 2048.
__global__ __launch_bounds__(256) void k_trsv_lower_quad(const double* __restrict__ G, int ld, const double* __restrict__ y,
                                                          int T, double* __restrict__ out, double* __restrict__ alpha) {
  extern __shared__ double w[];
  for (int i = threadIdx.x; i < T; i += 256) w[i] = y[i];
  __syncthreads();
  for (int k = 0; k < T; ++k) {
    if (threadIdx.x == 0) w[k] = w[k] / G[(size_t)k * ld + k];
    __syncthreads();
    const double wk = w[k];
    for (int i = k + 1 + threadIdx.x; i < T; i += 256) w[i] = fma(-G[(size_t)i * ld + k], wk, w[i]);
    __syncthreads();
  }
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < T; i += 256) s = fma(w[i], w[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && out) out[0] = red[0];
  if (alpha) {   // alpha = G^{-T} w  (cho_solve((G, True), y), the second half of the reference's call)
    for (int k = T - 1; k >= 0; --k) {
      __syncthreads();
      if (threadIdx.x == 0) w[k] = w[k] / G[(size_t)k * ld + k];
      __syncthreads();
      const double wk = w[k];
      for (int i = threadIdx.x; i < k; i += 256) w[i] = fma(-G[(size_t)k * ld + i], wk, w[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += 256) alpha[i] = w[i];
  }
}

// a10 gradient (GPI.py:1046-1051): out[k] = 0.5 tr((alpha alpha^T - Kinv) dK/dtheta_k) for theta = (log c, log ell, log noise)
// with scikit-learn's kernel gradients: c R, c R d^2 / ell^2, noise I  (R_ij = exp(-0.5 d^2 / ell^2)).
__global__ __launch_bounds__(256) void k_lml_grad(const double* __restrict__ x, const double* __restrict__ alpha,
                                                  const double* __restrict__ Kinv, int T, double c, double ell,
                                                  double noise, double* __restrict__ out) {
  __shared__ double red[3][256];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const long tt = (long)T * T;
  for (long idx = threadIdx.x; idx < tt; idx += 256) {
    const int i = (int)(idx / T), j = (int)(idx % T);
    const double t = alpha[i] * alpha[j] - Kinv[idx];
    const double u = x[i] / ell - x[j] / ell, d2 = u * u;
    const double cr = c * exp(-0.5 * d2);
    s0 = fma(t, cr, s0);
    s1 = fma(t, cr * d2, s1);
    if (i == j) s2 += t;
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = 0.5 * red[0][0];
    out[1] = 0.5 * red[1][0];
    out[2] = 0.5 * noise * red[2][0];
  }
}

// ------------------------------------------------------------------ messages of the switching variable
// GPI_HDP.forward / backward / coupled_state_coef (GPI_HDP.py:3546-3700) on the device, so that the [N, K] score matrix
// never leaves HBM between evaluation and assignment.  Sequential in N, K <= 64 states: one wave per direction, lane i
// = state i, its row of the (clamped, max-shifted) transition matrix in LDS, the message vector broadcast through LDS.
__device__ __forceinline__ double hmm_exp(double x, double m) {   // the reference's safe_exp element: NaN -> 1e-8
  const double e = exp(x - m);
  return (e != e) ? 1e-8 : e;
}
struct HmmArgs {
  const double* q;          // [N,K] log-observations
  const double* log_pi;     // [K]
  const double* log_trans;  // [K,K]
  int N, K;
  double* fmsg;             // [N,K]
  double* marg;             // [N]
  double* bmsg;             // [N,K]
};

__global__ __launch_bounds__(64) void k_hmm_messages(HmmArgs a) {
  extern __shared__ double sm[];
  const int K = a.K, N = a.N, i = threadIdx.x, LD = K + 1;
  double* P = sm;            // [K][K+1]
  double* f = P + K * LD;    // [K]
  const bool fwd = blockIdx.x == 0;
  {                          // blockIdx.y = variant of a batch of score matrices sharing log_pi / log_trans
    const size_t vo = (size_t)blockIdx.y * N * K;
    a.q += vo;
    a.fmsg += vo;
    a.bmsg += vo;
    a.marg += (size_t)blockIdx.y * N;
  }
  const bool live = i < K;
  const double ninf = -__builtin_inf();
  // my row of the transition operator: forward uses safe_exp(log_trans^T) clamped at 1e-6, backward safe_exp(log_trans)
  // clamped at 1e-5 (GPI_HDP.py:3586-3589, 3637-3642)
  if (live) {
    double m = ninf;
    for (int j = 0; j < K; ++j) m = fmax(m, fwd ? a.log_trans[(size_t)j * K + i] : a.log_trans[(size_t)i * K + j]);
    for (int j = 0; j < K; ++j) {
      double e = hmm_exp(fwd ? a.log_trans[(size_t)j * K + i] : a.log_trans[(size_t)i * K + j], m);
      if (e < (fwd ? 1e-6 : 1e-5)) e += 1e-4;
      P[i * LD + j] = e;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (fwd) {
    double pi_ = live ? exp(a.log_pi[i]) : 0.0;
    if (live && pi_ < 1e-10) pi_ += 1e-4;
    for (int t0 = 0; t0 < N; t0 += 8) {   // the observations of 8 steps are requested together: one load latency per 8 steps
      double qv8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) qv8[u] = (live && t0 + u < N) ? a.q[(size_t)(t0 + u) * K + i] : ninf;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u;
        if (t >= N) break;
        const double qv = qv8[u];
        const double qm = wave_allreduce<true>(qv);                     // (all 64 lanes take part in the shuffles)
        const double qe = live ? hmm_exp(qv, qm) : 0.0;
        double g = pi_;
        if (t > 0) {
          g = 0.0;
          if (live)
            for (int j = 0; j < K; ++j) g = fma(P[i * LD + j], f[j], g);
        }
        const double v = live ? g * qe : 0.0;
        const double mg = wave_allreduce<false>(v);
        const double fi = v / mg;
        __builtin_amdgcn_wave_barrier();
        if (live) {
          f[i] = fi;
          a.fmsg[(size_t)t * K + i] = fi;
        }
        if (i == 0) a.marg[t] = mg;
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
    double b = 1.0;
    if (live) a.bmsg[(size_t)(N - 1) * K + i] = 1.0;
    for (int t0 = N - 2; t0 >= 0; t0 -= 8) {
      double qv8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) qv8[u] = (live && t0 - u >= 0) ? a.q[(size_t)(t0 - u + 1) * K + i] : ninf;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 - u;
        if (t < 0) break;
        const double qv = qv8[u];
        const double qm = wave_allreduce<true>(qv);
        const double qe = live ? hmm_exp(qv, qm) : 0.0;
        if (live) f[i] = b * qe;
        __builtin_amdgcn_wave_barrier();
        double v = 0.0;
        if (live)
          for (int j = 0; j < K; ++j) v = fma(P[i * LD + j], f[j], v);
        const double nrm = wave_allreduce<false>((live && i < K - 1) ? v : 0.0);   // the reference leaves the last state out (:3645)
        b = v / nrm;
        if (live) a.bmsg[(size_t)t * K + i] = b;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// log of the normalised pair responsibilities (coupled_state_coef): one workgroup per step t
__global__ __launch_bounds__(256) void k_hmm_pair(const double* __restrict__ q, const double* __restrict__ log_trans,
                                                  const double* __restrict__ alpha, const double* __restrict__ beta, int N,
                                                  int K, double* __restrict__ out) {
  extern __shared__ double sm[];
  double* soft = sm;          // [K]  safe_exp(q[t]) * beta[t]
  double* rmax = soft + K;    // [K]  row maxima of log_trans
  __shared__ double red[256];
  const int t = blockIdx.x, tid = threadIdx.x;
  double* o = out + (size_t)t * K * K;
  if (t == 0) {               // respPair[0] = 0 -> log 0
    for (int e = tid; e < K * K; e += 256) o[e] = -__builtin_inf();
    return;
  }
  double qm = -__builtin_inf();
  for (int j = 0; j < K; ++j) qm = fmax(qm, q[(size_t)t * K + j]);
  for (int j = tid; j < K; j += 256) {
    soft[j] = hmm_exp(q[(size_t)t * K + j], qm) * beta[(size_t)t * K + j];
    double m = -__builtin_inf();
    for (int l = 0; l < K; ++l) m = fmax(m, log_trans[(size_t)j * K + l]);
    rmax[j] = m;
  }
  __syncthreads();
  double s = 0.0;
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    s += alpha[(size_t)(t - 1) * K + i] * soft[j] * hmm_exp(log_trans[e], rmax[i]);
  }
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  double den = red[0];
  if (den == 0.0) den = 1e-10;
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    o[e] = log(alpha[(size_t)(t - 1) * K + i] * soft[j] * hmm_exp(log_trans[e], rmax[i]) / den);
  }
}

// The same table reduced on the spot to what the hard assignment keeps of it: the FIRST arg-max of row t over the flattened
// K x K entries (GPI_HDP._safe_exp on the pair table; row 0 is all -inf -> 0; a row holding a NaN -> 0, as the host layer's
// first-arg-max did).  Same arithmetic, element by element, as k_hmm_pair; grid (N, variants).
__global__ __launch_bounds__(256) void k_hmm_pair_first(const double* __restrict__ q_all, const double* __restrict__ log_trans,
                                                        const double* __restrict__ alpha_all, const double* __restrict__ beta_all,
                                                        int N, int K, int64_t* __restrict__ first_all) {
  extern __shared__ double sm[];
  double* soft = sm;
  double* rmax = soft + K;
  __shared__ double red[256];
  __shared__ int redi[256];
  __shared__ int any_nan;
  const int t = blockIdx.x, tid = threadIdx.x;
  const size_t vo = (size_t)blockIdx.y * N * K;
  const double* q = q_all + vo;
  const double* alpha = alpha_all + vo;
  const double* beta = beta_all + vo;
  int64_t* first = first_all + (size_t)blockIdx.y * N;
  if (t == 0) {
    if (tid == 0) first[0] = 0;
    return;
  }
  if (tid == 0) any_nan = 0;
  double qm = -__builtin_inf();
  for (int j = 0; j < K; ++j) qm = fmax(qm, q[(size_t)t * K + j]);
  for (int j = tid; j < K; j += 256) {
    soft[j] = hmm_exp(q[(size_t)t * K + j], qm) * beta[(size_t)t * K + j];
    double m = -__builtin_inf();
    for (int l = 0; l < K; ++l) m = fmax(m, log_trans[(size_t)j * K + l]);
    rmax[j] = m;
  }
  __syncthreads();
  double s = 0.0;
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    s += alpha[(size_t)(t - 1) * K + i] * soft[j] * hmm_exp(log_trans[e], rmax[i]);
  }
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  double den = red[0];
  if (den == 0.0) den = 1e-10;
  __syncthreads();
  double bv = -__builtin_inf();
  int bi = K * K;                       // K * K = "nothing yet": an all -inf row keeps index 0 below
  bool nan = false;
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    const double v = log(alpha[(size_t)(t - 1) * K + i] * soft[j] * hmm_exp(log_trans[e], rmax[i]) / den);
    nan |= (v != v);
    if (v > bv || (bi == K * K && v == bv)) {
      bv = v;
      bi = e;
    }
  }
  if (nan) atomicOr(&any_nan, 1);
  red[tid] = bv;
  redi[tid] = bi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      const double ov = red[tid + w];
      const int oi = redi[tid + w];
      if (ov > red[tid] || (ov == red[tid] && oi < redi[tid])) {
        red[tid] = ov;
        redi[tid] = oi;
      }
    }
    __syncthreads();
  }
  if (tid == 0) first[t] = (any_nan || redi[0] >= K * K) ? 0 : redi[0];
}

// ------------------------------------------------------------------ per-cluster operators (plan)
// scal[k*8 + ..] : 0 c, 1 ell, 2 noise, 3 iso flag, 4 mean(diag Sigma), 5 jitter of K~, 6 ||K~^{-1}||_inf
struct PrepArgs {
  const double* xb;
  const double* mean;
  const double* Sigma;
  int T, TP, K;
  const double* theta;  // [K,3] device copy
  double* scal;         // [K,8]
  double* A;            // [K,TP,TP] K~ (identity padded)
  double* S;            // [K,TP,TP] 0.5 (Sigma + Sigma^T) (zero padded)
  double* xb_copy;      // [TP]
};

__global__ __launch_bounds__(256) void k_prep_build(PrepArgs a) {
  __shared__ double red[256];
  __shared__ int redi[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int T = a.T, TP = a.TP;
  const double* Sg = a.Sigma + (size_t)k * T * T;
  const double c = a.theta[3 * k], ell = a.theta[3 * k + 1], noise = a.theta[3 * k + 2];
  double s_abs = 0.0, s_sgn = 0.0;
  for (int i = tid; i < T; i += 256) {
    double d = Sg[(size_t)i * T + i];
    s_abs += fabs(d);
    s_sgn += d;
  }
  red[tid] = s_abs;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mean_abs = red[0] / T;
  __syncthreads();
  red[tid] = s_sgn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mS = red[0] / T;
  int bad = 0;
  for (int i = tid; i < T; i += 256) {
    double d = Sg[(size_t)i * T + i];
    if (!(fabs(d - mS) <= 1e-8 + 1e-5 * fabs(mS))) bad = 1;   // torch.isclose defaults (GPI.py:497)
  }
  redi[tid] = bad;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) redi[tid] |= redi[tid + o];
    __syncthreads();
  }
  const double jit = 1e-4 * fmax(mean_abs, F64_EPS);           // GPI.py:488
  if (tid == 0 && blockIdx.y == 0) {
    double* sc = a.scal + 8 * k;
    sc[0] = c;
    sc[1] = ell;
    sc[2] = noise;
    sc[3] = redi[0] ? 0.0 : 1.0;
    sc[4] = mS;
    sc[5] = jit;
    sc[6] = 0.0;   // ||K~^{-1}||_inf: accumulated by k_prep_final with an atomic maximum
  }
  if (k == 0 && blockIdx.y == 0)
    for (int i = tid; i < TP; i += 256) a.xb_copy[i] = (i < T) ? a.xb[i] : 0.0;
  double* Ak = a.A + (size_t)k * TP * TP;
  double* Sk = a.S + (size_t)k * TP * TP;
  for (int idx = blockIdx.y * 256 + tid; idx < TP * TP; idx += gridDim.y * 256) {
    int i = idx / TP, j = idx % TP;
    double av, sv = 0.0;
    if (i < T && j < T) {
      double u = a.xb[i] / ell - a.xb[j] / ell;
      av = c * exp(-0.5 * (u * u));
      if (i == j) av += jit;
      sv = 0.5 * (Sg[(size_t)i * T + j] + Sg[(size_t)j * T + i]);
    } else {
      av = (i == j) ? 1.0 : 0.0;
    }
    Ak[idx] = av;
    Sk[idx] = sv;
  }
}

struct PrepFinalArgs {
  const double* Q;      // Kinv * S * Kinv
  const double* Kinv;
  const double* mean;   // [K,T]
  double* scal;
  int T, TP;
  double* Mp;           // [K,TP,TP]
  double* ap;           // [K,TP]
  int interleave;       // 1: tile-pair interleaved columns (fused pairs kernel); 0: plain row-major (cooperative pairs kernel)
  int32_t* fb;          // fall-back list of k_pairs: its two counters start every plan state at zero (a killed launch cannot leave them set)
};

__global__ __launch_bounds__(256) void k_prep_final(PrepFinalArgs a) {
  const int k = blockIdx.x, tid = threadIdx.x;
  if (a.fb && k == 0 && blockIdx.y == 0 && tid == 0) {
    a.fb[0] = 0;
    a.fb[1 + PAIRS_FB_CAP] = 0;
  }
  const int T = a.T, TP = a.TP;
  const double c = a.scal[8 * k];
  const double* Q = a.Q + (size_t)k * TP * TP;
  const double* Ki = a.Kinv + (size_t)k * TP * TP;
  double* Mp = a.Mp + (size_t)k * TP * TP;
  // M' = c^2 (sym(Q) - sym(Kinv)) by 32 x 32 tiles: tile (bi, bj) and its mirror (bj, bi) are both read row-wise
  // (coalesced) and the mirror is transposed through LDS (pitch 33).
  // Column order of M' (logical j -> physical jp).  The pairs kernel reads row k of M' as the A operands of the NH
  // row tiles of one half h (tiles NH h .. NH h + NH - 1): inside a half, tiles are interleaved two by two so that ONE
  // 16-byte load per lane (lane cc) yields the operands of tiles 2q and 2q + 1; an odd last tile stays contiguous.
  __shared__ double tq[32][33], tk[32][33];
  const int NHh = (TP / 16) / 2, nt = TP / 32;
  const int tx = tid & 31, ty = tid >> 5;   // 32 x 8 threads, 4 rows each
  for (int t = blockIdx.y; t < nt * nt; t += gridDim.y) {
    const int bi = t / nt, bj = t % nt;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // mirror tile, natural order: element (32 bj + y, 32 bi + tx)
      const int y = ty + 8 * r;
      tq[y][tx] = Q[(size_t)(32 * bj + y) * TP + 32 * bi + tx];
      tk[y][tx] = Ki[(size_t)(32 * bj + y) * TP + 32 * bi + tx];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ty + 8 * r;
      const int i = 32 * bi + y, j = 32 * bj + tx;
      double v = 0.0;
      if (i < T && j < T)
        v = (c * c) * (0.5 * (Q[(size_t)i * TP + j] + tq[tx][y]) - 0.5 * (Ki[(size_t)i * TP + j] + tk[tx][y]));
      int jp = j;
      if (a.interleave) {
        const int hh = j / (16 * NHh), tl = (j / 16) % NHh, cc = j % 16;
        const int loc = (tl < 2 * (NHh / 2)) ? 32 * (tl / 2) + 2 * cc + (tl & 1) : 16 * (NHh - 1) + cc;
        jp = 16 * NHh * hh + loc;
      }
      Mp[(size_t)i * TP + jp] = v;
    }
  }
  // a' = c Kinv mean and the row sums of |Kinv| (rows dealt to the gridDim.y blocks of the cluster)
  const double* mu = a.mean + (size_t)k * T;
  __shared__ double red[256];
  double rmax = 0.0;
  for (int i = blockIdx.y * 256 + tid; i < TP; i += gridDim.y * 256) {
    double s = 0.0, rs = 0.0;
    if (i < T) {
#pragma unroll 8
      for (int j = 0; j < T; ++j) {
        const double kij = Ki[(size_t)j * TP + i];   // K~^{-1} = Z^T Z is symmetric: read column-wise, coalesced
        s = fma(kij, mu[j], s);
        rs += fabs(kij);
      }
    }
    a.ap[(size_t)k * TP + i] = c * s;
    rmax = fmax(rmax, rs);
  }
  red[tid] = rmax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
    __syncthreads();
  }
  // ||K~^{-1}||_inf >= ||K~^{-1}||_2: maximum over the blocks (non-negative doubles order like their bit patterns;
  // k_prep_build zeroes the slot)
  if (tid == 0) atomicMax(reinterpret_cast<unsigned long long*>(a.scal + 8 * k + 6), (unsigned long long)__double_as_longlong(red[0]));
}

}  // namespace

#ifdef HGP_STAMPS
unsigned long long* hgp_internal_stamp_dev = nullptr;   // host-side handle of the diagnostic counters (8 x u64)
#endif

// =============================================================================================
// C-ABI
// =============================================================================================
static int tp_for(int n) {   // padded size: wave kernels {32,64,96,128}, cooperative kernels {192,256}
  if (n <= HGP_MAX_T_WAVE) return 16 * nb_for(n);
  return n <= 192 ? 192 : 256;
}
// Diagnostic switch: HGP_PAIRS_COOP=1 (read when a plan is created) runs the cooperative kernels for T <= 128 too.
static int tp_plan(int n) {
  if (n <= HGP_MAX_T_WAVE && env_on("HGP_PAIRS_COOP")) return 128;   // (k_pairs_cooph<8>; a <4> instance spilled 316 VGPRs and had no use)
  return tp_for(n);
}

static size_t plan_bytes(int T, int Ts_max, int K, size_t* offs /*[24]*/) {
  const size_t TP = (size_t)tp_plan(std::max(T, Ts_max));
  const size_t mat = (size_t)K * TP * TP * sizeof(double);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += (bytes + 255) & ~(size_t)255;
    return at;
  };
  size_t tmp[24] = {0};
  tmp[0] = take((size_t)K * 3 * sizeof(double));   // theta
  tmp[1] = take((size_t)K * 8 * sizeof(double));   // scal
  tmp[2] = take(mat);                              // A  (K~ then L)
  tmp[3] = take(mat);                              // S
  tmp[4] = take(mat);                              // Z
  tmp[5] = take(mat);                              // Kinv
  tmp[6] = take(mat);                              // P
  tmp[7] = take(mat);                              // Q
  tmp[8] = take(mat);                              // Mp
  tmp[9] = take((size_t)K * TP * sizeof(double));  // ap
  tmp[10] = take((size_t)K * sizeof(int32_t));     // perm
  tmp[11] = take(TP * sizeof(double));             // x_basis copy
  if (TP > HGP_MAX_T_WAVE || env_on("HGP_PAIRS_COOP")) {   // cooperative kernel: overflow areas for dense grids
    const int nb = (int)TP / 16;
    const size_t cap = (nb >= 12) ? 48 : (nb == 8 ? 24 : 16);
    const size_t over = (size_t)nb * nb > cap ? (size_t)nb * nb - cap : 0;
    const size_t nscr = over ? (nb >= 12 ? 512 : 1024) : 0;
    tmp[22] = take(nscr * over * 256 * sizeof(double));
    tmp[23] = take((nscr + 1) * sizeof(int32_t));
  }
  {   // solve-based kernel (hgp_pairs_acc.hip): packed operands of L, Sigma; mean copy; per-workgroup S areas; flag list
    size_t sz[6];
    hgp_internal_acc_bytes((int)TP, K, sz);
    for (int i = 0; i < 6; ++i) tmp[12 + i] = take(sz[i]);
    tmp[18] = take((size_t)(K + 1) * sizeof(int32_t));
  }
  tmp[19] = take((size_t)(PAIRS_FB_CAP + 2) * sizeof(int32_t));   // fall-back list of k_pairs
  if (offs) memcpy(offs, tmp, sizeof(tmp));
  return o;
}

extern "C" {

int hgp_abi_version(void) { return HGP_ABI_VERSION; }

int hgp_debug_mfma_f64(const double* A, const double* B, double* C, void* stream) {
  if (!A || !B || !C) return -1;
  hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C);
  return launch_status();
}

int hgp_debug_exp_neg_f64(const double* h, int n, double* out, void* stream) {
  if (!h || !out || n < 0) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_exp_probe, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, h, n, out);
  return launch_status();
}

int hgp_gram_rbf_f64(const double* x, int nx, const double* y, int ny, double c, double ell, double noise,
                     double* K_out, void* stream) {
  if (!x || !K_out || nx <= 0 || ell <= 0.0) return -1;
  const int one = (y == nullptr);
  if (one) ny = nx;
  if (ny <= 0) return -1;
  size_t tot = (size_t)nx * ny;
  hipLaunchKernelGGL(k_gram_rbf, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, nx, y, ny, c,
                     ell, noise, one, K_out);
  return launch_status();
}

int hgp_potrf_batched_f64(double* A, int T, int b, double jitter_rel, double add_diag, double* Linv, double* logdet,
                          int32_t* info, void* stream) {
  if (b == 0) return 0;
  if (!A || T <= 0 || b < 0) return -1;
  if (T > HGP_MAX_T_COOP) return -2;
  PotrfArgs a{A, T, b, jitter_rel, add_diag, Linv, logdet, info};
  if (T > HGP_MAX_T_WAVE) return T <= 192 ? launch_coop_potrf<12>(a, (hipStream_t)stream) : launch_coop_potrf<16>(a, (hipStream_t)stream);
  dim3 grid((b + WAVES - 1) / WAVES), blk(64 * WAVES);
  hipStream_t st = (hipStream_t)stream;
  switch (nb_for(T)) {
    case 2: if (Linv) launch_wave_inv<2>(a, st); hipLaunchKernelGGL(k_wave_potrf<2>, grid, blk, 0, st, a); break;
    case 4: if (Linv) launch_wave_inv<4>(a, st); hipLaunchKernelGGL(k_wave_potrf<4>, grid, blk, 0, st, a); break;
    case 6: if (Linv) launch_wave_inv<6>(a, st); hipLaunchKernelGGL(k_wave_potrf<6>, grid, blk, 0, st, a); break;
    default: if (Linv) launch_wave_inv<8>(a, st); hipLaunchKernelGGL(k_wave_potrf<8>, grid, blk, 0, st, a); break;
  }
  return launch_status();
}

int hgp_score_groups_f64(const double* Y, int ldy, const double* mean, long mean_stride, const double* Sigma,
                         long sigma_stride, int T, const int32_t* item_mat, const int32_t* item_mean,
                         const double* item_add, const int32_t* item_off, const int32_t* item_cnt, int n_items,
                         const int32_t* seg_ids, double jitter_rel, double* out_quad, double* out_logdet,
                         int32_t* out_info, void* stream) {
  if (n_items == 0) return 0;
  if (!Y || !Sigma || !item_mat || !item_off || !item_cnt || !out_quad || T <= 0 || ldy < T || n_items < 0) return -1;
  if (T > HGP_MAX_T_COOP) return -2;
  ScoreArgs a{Y, ldy, mean, mean_stride, Sigma, sigma_stride, T, T, item_mat, item_mean, item_add, item_off, item_cnt, n_items,
              seg_ids, jitter_rel, out_quad, out_logdet, out_info};
  if (T > HGP_MAX_T_WAVE) return T <= 192 ? launch_coop_score<12>(a, (hipStream_t)stream) : launch_coop_score<16>(a, (hipStream_t)stream);
  dim3 grid((n_items + WAVES - 1) / WAVES), blk(64 * WAVES);
  hipStream_t st = (hipStream_t)stream;
  switch (nb_for(T)) {
    case 2: hipLaunchKernelGGL(k_wave_score<2>, grid, blk, 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_wave_score<4>, grid, blk, 0, st, a); break;
    case 6: hipLaunchKernelGGL(k_wave_score<6>, grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL(k_wave_score<8>, grid, blk, 0, st, a); break;
  }
  return launch_status();
}

size_t hgp_pairs_plan_device_bytes(int T, int Ts_max, int K) {
  if (T <= 0 || Ts_max <= 0 || K <= 0) return 0;
  return plan_bytes(T, Ts_max, K, nullptr);
}

int hgp_pairs_plan_create(hgp_pairs_plan** plan, int T, int Ts_max, int K, const double* theta_host, void* dev_buf,
                          size_t dev_bytes) {
  if (!plan || !theta_host || !dev_buf || T <= 0 || Ts_max <= 0 || K <= 0) return -1;
  if (T > HGP_MAX_T_COOP || Ts_max > HGP_MAX_T_COOP) return -2;
  size_t offs[24];
  if (dev_bytes < plan_bytes(T, Ts_max, K, offs)) return -1;
  for (int k = 0; k < K; ++k)
    if (!(theta_host[3 * k] > 0.0) || !(theta_host[3 * k + 1] > 0.0)) return -1;
  hgp_pairs_plan* p = new (std::nothrow) hgp_pairs_plan();
  if (!p) return -1;
  p->T = T;
  p->K = K;
  p->TP = tp_plan(std::max(T, Ts_max));
  p->NB = p->TP / 16;
  p->coop = p->TP > HGP_MAX_T_WAVE || env_on("HGP_PAIRS_COOP");
  p->theta.assign(theta_host, theta_host + 3 * (size_t)K);
  p->perm.resize(K);
  for (int k = 0; k < K; ++k) p->perm[k] = k;
  std::stable_sort(p->perm.begin(), p->perm.end(),
                   [&](int a, int b) { return p->theta[3 * a + 1] < p->theta[3 * b + 1]; });
  for (int i = 0; i < K;) {
    int j = i;
    double ell = p->theta[3 * p->perm[i] + 1];
    while (j < K && p->theta[3 * p->perm[j] + 1] == ell) ++j;
    p->grp_beg.push_back(i);
    p->grp_end.push_back(j);
    p->grp_ell.push_back(ell);
    i = j;
  }
  char* base = (char*)dev_buf;
  p->d_theta = (double*)(base + offs[0]);
  p->d_scal = (double*)(base + offs[1]);
  p->d_A = (double*)(base + offs[2]);
  p->d_S = (double*)(base + offs[3]);
  p->d_Z = (double*)(base + offs[4]);
  p->d_Kinv = (double*)(base + offs[5]);
  p->d_P = (double*)(base + offs[6]);
  p->d_Q = (double*)(base + offs[7]);
  p->d_Mp = (double*)(base + offs[8]);
  p->d_ap = (double*)(base + offs[9]);
  p->d_perm = (int32_t*)(base + offs[10]);
  p->d_xb = (double*)(base + offs[11]);
  p->d_Lop = (double*)(base + offs[12]);
  p->d_LTop = (double*)(base + offs[13]);
  p->d_Dop = (double*)(base + offs[14]);
  p->d_Sop = (double*)(base + offs[15]);
  p->d_mu = (double*)(base + offs[16]);
  p->d_sscr = (double*)(base + offs[17]);
  p->d_acc_list = (int32_t*)(base + offs[18]);
  p->d_fb = (int32_t*)(base + offs[19]);
  if (hipMemset(p->d_fb, 0, (size_t)(PAIRS_FB_CAP + 2) * sizeof(int32_t)) != hipSuccess) {
    delete p;
    return 1000 + (int)hipGetLastError();
  }
  if (p->coop) {
    const size_t cap = (p->NB >= 12) ? 48 : (p->NB == 8 ? 24 : 16);
    const size_t over = (size_t)p->NB * p->NB > cap ? (size_t)p->NB * p->NB - cap : 0;
    p->nscr = over ? (p->NB >= 12 ? 512 : 1024) : 0;
    p->escr_stride = (long)(over * 256);
    p->d_escr = (double*)(base + offs[22]);
    p->d_eflags = (int32_t*)(base + offs[23]);
    if (hipMemset(p->d_eflags, 0, (p->nscr + 1) * sizeof(int32_t)) != hipSuccess) {
      delete p;
      return 1000 + (int)hipGetLastError();
    }
  }
  if (hipMemcpy(p->d_theta, p->theta.data(), sizeof(double) * 3 * K, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->d_perm, p->perm.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice) != hipSuccess) {
    delete p;
    return 1000 + (int)hipGetLastError();
  }
  *plan = p;
  return 0;
}

void hgp_pairs_plan_destroy(hgp_pairs_plan* plan) { delete plan; }

const double* hgp_pairs_plan_scalars(const hgp_pairs_plan* plan) { return plan ? plan->d_scal : nullptr; }

int hgp_pairs_plan_update(hgp_pairs_plan* p, const double* x_basis, const double* mean, const double* Sigma,
                          int32_t* info, void* stream) {
  if (!p || !x_basis || !mean || !Sigma) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int K = p->K, T = p->T, TP = p->TP;
  PrepArgs pa{x_basis, mean, Sigma, T, TP, K, p->d_theta, p->d_scal, p->d_A, p->d_S, p->d_xb};
  hipLaunchKernelGGL(k_prep_build, dim3(K, 8), dim3(256), 0, st, pa);
  // Z = chol(K~)^{-1}  (the factor itself is not needed)
  PotrfArgs fa{p->d_A, TP, K, 0.0, 0.0, p->d_Z, nullptr, info};
  fa.inv_info = 1;
  fa.symmetric = 1;   // k_prep_build writes K~ from (x_i - x_j)^2: exactly symmetric
  switch (p->NB) {
    case 2: launch_wave_inv<2>(fa, st); break;
    case 4: launch_wave_inv<4>(fa, st); break;
    case 6: launch_wave_inv<6>(fa, st); break;
    case 8:   // one workgroup per block column (the trailing updates split over its four waves) halves the latency at T = 128
      if (env_on("HGP_PLAN_WAVE_INV")) launch_wave_inv<8>(fa, st);
      else launch_coop_inv_only<8>(fa, st);
      break;
    case 12: launch_coop_potrf<12>(fa, st); break;
    default: launch_coop_potrf<16>(fa, st); break;
  }
  const long sm = (long)TP * TP;
  GemmArgs g1{p->d_Z, p->d_Z, p->d_Kinv, TP, TP, TP, TP, TP, TP, sm, sm, sm, 1.0, 0.0, 1, 0};     // Kinv = Z^T Z
  GemmArgs g2{p->d_S, p->d_Kinv, p->d_P, TP, TP, TP, TP, TP, TP, sm, sm, sm, 1.0, 0.0, 0, 0};    // P = S Kinv
  GemmArgs g3{p->d_Kinv, p->d_P, p->d_Q, TP, TP, TP, TP, TP, TP, sm, sm, sm, 1.0, 0.0, 0, 0};    // Q = Kinv S Kinv
  launch_gemm(g1, K, st);
  launch_gemm(g2, K, st);
  launch_gemm(g3, K, st);
  PrepFinalArgs fin{p->d_Q, p->d_Kinv, mean, p->d_scal, T, TP, p->d_Mp, p->d_ap, p->coop ? 0 : 1, p->d_fb};
  hipLaunchKernelGGL(k_prep_final, dim3(K, 8), dim3(256), 0, st, fin);
  // overflow-area flags: a launch that was killed mid-flight must not leave areas marked busy for the next one
  if (p->d_eflags && p->nscr > 0 && hipMemsetAsync(p->d_eflags, 0, (p->nscr + 1) * sizeof(int32_t), st) != hipSuccess) return launch_status();
  // clusters whose explicit operator would lose digits take the solve-based kernel: flags + packed operands of L, Sigma
  int rc = hgp_internal_acc_prep(p, mean, st);
  return rc ? rc : launch_status();
}

int hgp_pairs_plan_set_accuracy(hgp_pairs_plan* p, double tol) {
  if (!p || tol != tol) return -1;
  p->acc_tol = tol;
  return 0;
}

int hgp_pairs_plan_set_score_output(hgp_pairs_plan* p, int on) {
  if (!p) return -1;
  p->score_out = on ? 1 : 0;
  return 0;
}

int hgp_loglik_pairs_f64(const hgp_pairs_plan* p, const double* x, const double* y, int N, int Ts,
                         const double* first_noise, const int32_t* sel, double* out_quad, double* out_logdet,
                         int32_t* out_info, void* stream) {
  if (N == 0) return 0;   // an empty batch is a no-op (its pointers may legitimately be null)
  if (!p || !x || !y || !out_quad || N < 0 || Ts <= 0) return -1;
  if (Ts > HGP_MAX_T_COOP) return -2;
  if (Ts > p->TP) return -2;   // plan was created with a smaller Ts_max
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  // the hand-out flags of the overflow areas start every call at "free": a launch that was killed mid-flight cannot leave the
  // next one spinning (a plan serves ONE stream at a time; see the header)
  if (p->coop && p->d_eflags && p->nscr > 0 &&
      hipMemsetAsync(p->d_eflags, 0, (p->nscr + 1) * sizeof(int32_t), st) != hipSuccess)
    return launch_status();
  for (size_t gi = 0; gi < p->grp_beg.size() && rc == 0; ++gi) {
    PairsArgs a{x, y, N, Ts, p->d_xb, p->T, p->d_Mp, p->d_ap, p->d_scal, p->d_perm, p->grp_beg[gi], p->grp_end[gi],
                p->grp_ell[gi], first_noise, sel,
#ifdef HGP_STAMPS
                hgp_internal_stamp_dev,
#endif
                p->K, out_quad, out_logdet, out_info, p->d_escr, p->d_eflags, p->nscr, p->escr_stride,
                env_on("HGP_PAIRS_GENERIC") ? 1 : 0, -0.5 * (double)Ts * 1.8378770664093453, p->score_out, p->d_fb};
    rc = hgp_internal_pairs_fast(a, p->NB, p->coop, st);
  }
  if (rc == 0) rc = hgp_internal_pairs_acc(p, x, y, N, Ts, first_noise, sel, out_quad, out_logdet, out_info, st);
  return rc;
}

int hgp_gemm_batched_f64(int transA, int transB, int M, int N, int Kd, double alpha, const double* A, int lda, long strideA,
                         const double* B, int ldb, long strideB, double beta, double* C, int ldc, long strideC, int batch,
                         void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || Kd <= 0 || batch < 0) return -1;
  if (batch == 0) return 0;
  GemmArgs g{A, B, C, M, N, Kd, lda, ldb, ldc, strideA, strideB, strideC, alpha, beta, transA, transB};
  return launch_gemm(g, batch, (hipStream_t)stream);
}

size_t hgp_matrix_lik_ws_bytes(int T, int b) { return (size_t)b * ((size_t)4 * T * T + T) * sizeof(double) + 256; }

int hgp_lat_error_f64(const double* f_cur, const double* f_prev, const double* A, const double* Gamma, const double* covprev,
                      int T, int b, double* out, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
  if (!f_cur || !f_prev || !A || !Gamma || !covprev || !out || T <= 0 || b < 0) return -1;
  if (b == 0) return 0;
  if (T > HGP_MAX_T_COOP) return -2;
  if (T <= HGP_MAX_T_WAVE)   // fused: one wavefront per item, nothing goes through the workspace
    return hgp_internal_lat_error_wave(f_cur, f_prev, A, Gamma, covprev, T, b, out, info, (hipStream_t)stream);
  if (!ws || ws_bytes < hgp_matrix_lik_ws_bytes(T, b)) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (!env_on("HGP_MATLIK_COMPOSE"))   // 128 < T <= 256: one fused cooperative kernel per item, Gram form (hgp_matlik_coop.hip)
    return hgp_internal_lat_coop(f_cur, f_prev, A, Gamma, covprev, T, b, out, info, (double*)ws, st);
  // (kept for A/B runs: the composition of the batched kernels)
  const long tt = (long)T * T;
  double* Gc = (double*)ws;          // copy of Gamma -> L
  double* Z = Gc + (size_t)b * tt;   // L^{-1}
  double* Y = Z + (size_t)b * tt;    // Z A
  double* Y2 = Y + (size_t)b * tt;   // Y P  (and z = Z r in its first T entries per item afterwards)
  double* r = Y2 + (size_t)b * tt;   // residuals [b,T]
  if (hipMemcpyAsync(Gc, Gamma, sizeof(double) * b * tt, hipMemcpyDeviceToDevice, st) != hipSuccess) return launch_status();
  int rc = hgp_potrf_batched_f64(Gc, T, b, 1e-8, 0.0, Z, nullptr, info, stream);   // _chol_spd(Gamma), GPI_model.py:312
  if (rc) return rc;
  hipLaunchKernelGGL(k_lat_resid, dim3(b), dim3(256), 0, st, f_cur, f_prev, A, T, r);
  GemmArgs g1{Z, A, Y, T, T, T, T, T, T, tt, tt, tt, 1.0, 0.0, 0, 0};
  g1.triA = 1;                                                                       // Z = L^-1 is lower triangular
  if ((rc = launch_gemm(g1, b, st))) return rc;
  GemmArgs g2{Y, covprev, Y2, T, T, T, T, T, T, tt, tt, tt, 1.0, 0.0, 0, 0};
  if ((rc = launch_gemm(g2, b, st))) return rc;
  // trace(A^T Gamma^{-1} A P) = sum (Y P) o Y
  hipLaunchKernelGGL(k_dot_batched, dim3(b), dim3(256), 0, st, Y2, Y, tt, tt, tt, -0.5, 0, out);
  // mahal = || Z r ||^2 : z = Z r as a T x 1 GEMM into Y2
  GemmArgs g3{Z, r, Y2, T, 1, T, T, 1, 1, tt, (long)T, tt, 1.0, 0.0, 0, 0};
  g3.triA = 1;
  if ((rc = launch_gemm(g3, b, st))) return rc;
  hipLaunchKernelGGL(k_dot_batched, dim3(b), dim3(256), 0, st, Y2, Y2, tt, tt, (long)T, -0.5, 1, out);
  return launch_status();
}

int hgp_mniw_loglik_f64(const double* M, const double* Sigma, const double* m_mean, const double* m_r_cov,
                        const double* scale, int scale_is_diagonal, long prior_stride, int T, int b, double* out, int32_t* info,
                        void* ws, size_t ws_bytes, void* stream) {
  if (!M || !Sigma || !m_mean || !scale || !out || T <= 0 || b < 0) return -1;
  if (b == 0) return 0;
  if (T > HGP_MAX_T_COOP) return -2;
  if (T <= HGP_MAX_T_WAVE)   // fused: one wavefront per item, nothing goes through the workspace
    return hgp_internal_mniw_wave(M, Sigma, m_mean, m_r_cov, scale, scale_is_diagonal, prior_stride, T, b, out, info, (hipStream_t)stream);
  if (!ws || ws_bytes < hgp_matrix_lik_ws_bytes(T, b)) return -1;
  hipStream_t st = (hipStream_t)stream;
  // the hot path's call (identity right covariance, diagonal prior scale): one fused cooperative kernel per item (hgp_matlik_coop.hip)
  if (!m_r_cov && scale_is_diagonal && !env_on("HGP_MATLIK_COMPOSE"))
    return hgp_internal_mniw_coop(M, Sigma, m_mean, scale, prior_stride, T, b, out, info, (double*)ws, st);
  // everything else at 128 < T <= 256: composition of the batched kernels
  const long tt = (long)T * T;
  double* Sc = (double*)ws;
  double* Z = Sc + (size_t)b * tt;
  double* D = Z + (size_t)b * tt;
  double* Y = D + (size_t)b * tt;
  if (hipMemcpyAsync(Sc, Sigma, sizeof(double) * b * tt, hipMemcpyDeviceToDevice, st) != hipSuccess) return launch_status();
  int rc = hgp_potrf_batched_f64(Sc, T, b, 0.0, 1e-8, Z, nullptr, info, stream);   // chol(0.5(S+S^T) + 1e-8 I), GPI_model.py:1353
  if (rc) return rc;
  hipLaunchKernelGGL(k_sub_batched, dim3((unsigned)((tt + 255) / 256), std::min(b, 65535)), dim3(256), 0, st, M, m_mean, tt, prior_stride, tt, D, b);
  GemmArgs g1{Z, D, Y, T, T, T, T, T, T, tt, tt, tt, 1.0, 0.0, 0, 0};              // Y = L^{-1} D
  g1.triA = 1;
  if ((rc = launch_gemm(g1, b, st))) return rc;
  if (m_r_cov) {                                                                     // sum (D R) o Sigma^{-1} D = sum (Y R) o Y
    GemmArgs g2{Y, m_r_cov, D, T, T, T, T, T, T, tt, prior_stride, tt, 1.0, 0.0, 0, 0};
    if ((rc = launch_gemm(g2, b, st))) return rc;
    hipLaunchKernelGGL(k_dot_batched, dim3(b), dim3(256), 0, st, D, Y, tt, tt, tt, -0.5, 0, out);
  } else {
    hipLaunchKernelGGL(k_dot_batched, dim3(b), dim3(256), 0, st, Y, Y, tt, tt, tt, -0.5, 0, out);
  }
  if (scale_is_diagonal) {   // the hot path's prior scale sigma I: trace(Sigma^{-1} S) = sum_j S_jj |Z e_j|^2, no product
    hipLaunchKernelGGL(k_colnorm_diag, dim3(b), dim3(256), 0, st, Z, scale, prior_stride, T, -0.5, 1, out);
    return launch_status();
  }
  GemmArgs g3{Z, scale, Y, T, T, T, T, T, T, tt, prior_stride, tt, 1.0, 0.0, 0, 0};  // trace(Sigma^{-1} S) = sum (Z S) o Z
  g3.triA = 1;
  if ((rc = launch_gemm(g3, b, st))) return rc;
  hipLaunchKernelGGL(k_dot_batched, dim3(b), dim3(256), 0, st, Y, Z, tt, tt, tt, -0.5, 1, out);
  return launch_status();
}

int hgp_warp_cov_f64(const double* x, int T, double rho, double omega, double diag_add, int normalize, double* K_out,
                     void* stream) {
  if (!x || !K_out || T <= 0 || !(rho > 0.0)) return -1;
  const size_t tot = (size_t)T * T;
  hipLaunchKernelGGL(k_warp_cov, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, T, rho, omega,
                     diag_add, normalize, K_out);
  return launch_status();
}

int hgp_chol_rank1_f64(double* L, const double* v, const double* alpha, const double* beta, int T, int b, int32_t* info,
                       void* stream) {
  if (!L || !v || T <= 0 || b < 0) return -1;
  if (b == 0) return 0;
  if (T > 256) return -2;
  Rank1Args a{L, v, alpha, beta, T, b, info, nullptr};
#ifdef HGP_STAMPS
  a.stamps = hgp_internal_stamp_dev;
#endif
  if (T % 2 == 0 && !env_on("HGP_RANK1_DIRECT") && !env_on("HGP_RANK1_COAL") && (T >= 192 || env_on("HGP_RANK1_PIPE")))
    hipLaunchKernelGGL(k_chol_rank1_pipe, dim3(b), dim3(64 * ((T + 63) / 64)), 0, (hipStream_t)stream, a);
  else if (T % 2 == 0 && !env_on("HGP_RANK1_DIRECT"))
    hipLaunchKernelGGL(k_chol_rank1<true>, dim3(b), dim3(64 * ((T + 63) / 64)), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_chol_rank1<false>, dim3(b), dim3(64 * ((T + 63) / 64)), 0, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_lds_chain_gather_f64(const double* stA, const double* stG, const double* stC, const double* stS, const double* stPsm,
                             const double* stP, const double* stF, const double* stFsm, const int64_t* pos, int T,
                             double* out, const double* Y, long y_row0, double* y_out, void* stream) {
  if (!stA || !stG || !stC || !stS || !stP || !stPsm || !stF || !stFsm || !pos || !out || T <= 0 || (Y && !y_out)) return -1;
  ChainGatherArgs a{{stA, stG, stC, stS, stPsm, stP, stF, stFsm}, pos, out, T, Y, y_row0, y_out};
  const long total = 6L * T * T + 2L * T;
  hipLaunchKernelGGL(k_chain_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_lds_chain_scatter_f64(const double* f_post, const double* c_post, const double* f_sm_prev, const double* P_sm_prev,
                              double* stF, double* stFsm, double* stP, double* stPsm, const int64_t* pos, int T, void* stream) {
  if (!f_post || !c_post || !f_sm_prev || !P_sm_prev || !stF || !stFsm || !stP || !stPsm || !pos || T <= 0) return -1;
  ChainScatterArgs a{f_post, c_post, f_sm_prev, P_sm_prev, stF, stFsm, stP, stPsm, pos, T};
  hipLaunchKernelGGL(k_chain_scatter, dim3((unsigned)(((long)T * T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_add_diag_mean_f64(const double* R, const double* S, int T, int b, double factor, double* out, void* stream) {
  if (!R || !S || !out || T <= 0 || b < 0) return -1;
  if (b == 0) return 0;
  hipLaunchKernelGGL(k_add_diag_mean, dim3(b), dim3(256), 0, (hipStream_t)stream, R, S, T, factor, out);
  return launch_status();
}

int hgp_gemm_add_batched_f64(int transA, int transB, int M, int N, int Kd, double alpha, const double* A, int lda, long strideA,
                             const double* B, int ldb, long strideB, double beta, const double* D, int ldd, long strideD,
                             double* C, int ldc, long strideC, int batch, void* stream) {
  if (!A || !B || !C || !D || M <= 0 || N <= 0 || Kd <= 0 || batch < 0) return -1;
  if (batch == 0) return 0;
  GemmArgs g{A, B, C, M, N, Kd, lda, ldb, ldc, strideA, strideB, strideC, alpha, beta, transA, transB};
  g.D = D;
  g.ldd = ldd;
  g.sD = strideD;
  return launch_gemm(g, batch, (hipStream_t)stream);
}

int hgp_rts_chain_f64(const double* J, const double* P, const double* AM, double* M, double* Cv, int n, int T, void* stream) {
  if (!J || !P || !AM || !M || !Cv || n < 0 || T <= 0) return -1;
  if (T > 96) return -2;
  if (n < 2) return 0;
  RtsArgs a{J, P, AM, M, Cv, n, T};
  const size_t lds = sizeof(double) * (2 * 96 * 100 + 2 * 96);
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_rts_chain), lds)) return rc_;
  hipLaunchKernelGGL(k_rts_chain, dim3(1), dim3(64 * RTS_WAVES), lds, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_lds_chain_finish_f64(int T, const double* part, const double* ee, const double* Snew, const int32_t* info1,
                             const int32_t* info2, const int32_t* info0, double* W, double* n0, double* Nf, int32_t* bad_count,
                             double* stA, double* stG, double* stC, double* stS, int64_t* pos, int annealing, int32_t* sync,
                             void* stream) {
  if (!part || !ee || !Snew || !info1 || !info2 || !W || !n0 || !Nf || !bad_count || !stA || !stG || !stC || !stS || !pos ||
      !sync || T <= 0)
    return -1;
  ChainFinishArgs a{T, part, ee, Snew, info1, info2, info0, W, n0, Nf, bad_count, stA, stG, stC, stS, pos, annealing, sync};
  const long n2 = 2L * T * T;
  hipLaunchKernelGGL(k_chain_finish, dim3((unsigned)std::min<long>(64, (n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_trsv_lower_quad_f64(const double* G, int ld, const double* y, int T, double* out, void* stream) {
  if (!G || !y || !out || T <= 0 || ld < T) return -1;
  if (T > 2048) return -2;
  hipLaunchKernelGGL(k_trsv_lower_quad, dim3(1), dim3(256), sizeof(double) * T, (hipStream_t)stream, G, ld, y, T, out,
                     (double*)nullptr);
  return launch_status();
}

int hgp_hmm_messages_f64(const double* q, const double* log_pi, const double* log_trans, int N, int K, double* fmsg,
                         double* marg, double* bmsg, double* log_resp_pair, void* stream) {
  if (N == 0) return 0;
  if (!q || !log_pi || !log_trans || !fmsg || !marg || !bmsg || N < 0 || K <= 0) return -1;
  if (K > 64) return -2;
  HmmArgs a{q, log_pi, log_trans, N, K, fmsg, marg, bmsg};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_hmm_messages, dim3(2), dim3(64), sizeof(double) * ((size_t)K * (K + 1) + K), st, a);
  if (log_resp_pair)
    hipLaunchKernelGGL(k_hmm_pair, dim3(N), dim3(256), sizeof(double) * 2 * K, st, q, log_trans, (const double*)fmsg,
                       (const double*)bmsg, N, K, log_resp_pair);
  return launch_status();
}

int hgp_hmm_local_terms_f64(const double* q, const double* log_pi, const double* log_trans, int N, int K, int B, double* qnorm,
                            double* fmsg, double* marg, double* bmsg, int64_t* labels, int64_t* pair_first, double* last_log,
                            void* stream) {
  if (N == 0 || B == 0) return 0;
  if (!q || !log_pi || !log_trans || !qnorm || !fmsg || !marg || !bmsg || !labels || N < 0 || K <= 0 || B < 0) return -1;
  if (K > 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = hgp_internal_loglik_rows_b(q, N, K, B, qnorm, st)) return rc;
  HmmArgs a{qnorm, log_pi, log_trans, N, K, fmsg, marg, bmsg};
  hipLaunchKernelGGL(k_hmm_messages, dim3(2, B), dim3(64), sizeof(double) * ((size_t)K * (K + 1) + K), st, a);
  if (int rc = hgp_internal_assign_b(fmsg, bmsg, N, K, B, labels, last_log, st)) return rc;
  if (pair_first)
    hipLaunchKernelGGL(k_hmm_pair_first, dim3(N, B), dim3(256), sizeof(double) * 2 * K, st, (const double*)qnorm, log_trans,
                       (const double*)fmsg, (const double*)bmsg, N, K, pair_first);
  return launch_status();
}

int hgp_trsv_lower_solve_f64(const double* G, int ld, const double* y, int T, double* alpha, double* quad, void* stream) {
  if (!G || !y || !alpha || T <= 0 || ld < T) return -1;
  if (T > 2048) return -2;
  hipLaunchKernelGGL(k_trsv_lower_quad, dim3(1), dim3(256), sizeof(double) * T, (hipStream_t)stream, G, ld, y, T, quad, alpha);
  return launch_status();
}

int hgp_lml_grad_f64(const double* x, const double* alpha, const double* Kinv, int T, double c, double ell, double noise,
                     double* out3, void* stream) {
  if (!x || !alpha || !Kinv || !out3 || T <= 0 || !(ell > 0.0)) return -1;
  hipLaunchKernelGGL(k_lml_grad, dim3(1), dim3(256), 0, (hipStream_t)stream, x, alpha, Kinv, T, c, ell, noise, out3);
  return launch_status();
}

#ifdef HGP_STAMPS
// diagnostic build: read and reset the phase cycle sums (d/f*, sweep 1, K** init, sweep 2, regularise, factor, -, diag16)
int hgp_debug_stamps(unsigned long long* out8_host) {
  if (!hgp_internal_stamp_dev) {
    if (hipMalloc(&hgp_internal_stamp_dev, 128) != hipSuccess) return 1;
    (void)hipMemset(hgp_internal_stamp_dev, 0, 128);
  }
  if (hipMemcpy(out8_host, hgp_internal_stamp_dev, 128, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  (void)hipMemset(hgp_internal_stamp_dev, 0, 128);
  return 0;
}
#endif

int hgp_score_each_f64(const double* Y, int ldy, const double* mean, long mean_stride, const double* Sigma,
                       long sigma_stride, int T, const int32_t* seg_mat, const int32_t* seg_mean, const double* seg_add,
                       int n, double jitter_rel, int symmetric, double* out_quad, double* out_logdet, int32_t* out_info,
                       void* stream) {
  if (n == 0) return 0;
  if (!Y || !Sigma || !seg_mat || !out_quad || T <= 0 || ldy < T || n < 0) return -1;
  if (T > HGP_MAX_T_WAVE) return -2;   // larger T: hgp_score_groups_f64 with one segment per item
  EachArgs a{Y, ldy, mean, mean_stride, Sigma, sigma_stride, T, n, seg_mat, seg_mean, seg_add, jitter_rel, out_quad, out_logdet,
             out_info, symmetric};
  dim3 grid((n + WAVES - 1) / WAVES), blk(64 * WAVES);
  hipStream_t st = (hipStream_t)stream;
  switch (nb_for(T)) {
    case 2: if (a.symmetric) hipLaunchKernelGGL((k_wave_score1<2, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((k_wave_score1<2, false>), grid, blk, 0, st, a); break;
    case 4: if (a.symmetric) hipLaunchKernelGGL((k_wave_score1<4, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((k_wave_score1<4, false>), grid, blk, 0, st, a); break;
    case 6: if (a.symmetric) hipLaunchKernelGGL((k_wave_score1<6, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((k_wave_score1<6, false>), grid, blk, 0, st, a); break;
    default: if (a.symmetric) hipLaunchKernelGGL((k_wave_score1<8, true>), grid, blk, 0, st, a); else hipLaunchKernelGGL((k_wave_score1<8, false>), grid, blk, 0, st, a); break;
  }
  return launch_status();
}

int hgp_chol_inverse_batched_f64(const double* A, int T, int b, double jitter_rel, double add_diag, double* Linv,
                                 int32_t* info, void* stream) {
  if (b == 0) return 0;
  if (!A || !Linv || T <= 0 || b < 0) return -1;
  if (T > HGP_MAX_T_COOP) return -2;
  PotrfArgs a{const_cast<double*>(A), T, b, jitter_rel, add_diag, Linv, nullptr, info};
  a.inv_info = 1;
  hipStream_t st = (hipStream_t)stream;
  if (T > HGP_MAX_T_WAVE) return T <= 192 ? launch_coop_inv_only<12>(a, st) : launch_coop_inv_only<16>(a, st);
  switch (nb_for(T)) {
    case 2: launch_wave_inv<2>(a, st); break;
    case 4: launch_wave_inv<4>(a, st); break;
    case 6: launch_wave_inv<6>(a, st); break;
    default: launch_wave_inv<8>(a, st); break;
  }
  return launch_status();
}

// The same inverse with a caller-provided workspace work[b,T,T] (T > 128 only; may be NULL): for batches that would fill the
// chip several times over with the per-block-column kernel (b * NB workgroups, every one a full factorisation) the matrix is
// factored ONCE into the workspace and L^-1 follows from L by block columns (k_trtri).  Small batches keep the per-block-column
// kernel: one launch, 97 us at T = 256 against 146 + 73 us for factor + k_trtri.
int hgp_chol_inverse_ws_f64(const double* A, int T, int b, double jitter_rel, double add_diag, double* Linv, double* work,
                            int32_t* info, void* stream) {
  if (b == 0) return 0;
  if (!A || !Linv || T <= 0 || b < 0) return -1;
  if (T > HGP_MAX_T_COOP) return -2;
  const int nb = T <= 192 ? 12 : 16;
  if (T <= HGP_MAX_T_WAVE || !work || (long)b * nb <= 512) return hgp_chol_inverse_batched_f64(A, T, b, jitter_rel, add_diag, Linv, info, stream);
  PotrfArgs a{const_cast<double*>(A), T, b, jitter_rel, add_diag, Linv, nullptr, info};
  a.Aout = work;
  return T <= 192 ? launch_coop_potrf<12>(a, (hipStream_t)stream) : launch_coop_potrf<16>(a, (hipStream_t)stream);
}

}  // extern "C"
