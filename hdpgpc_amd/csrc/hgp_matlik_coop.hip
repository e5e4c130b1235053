// a9 (matrix_normal_inv_wishart.log_likelihood_MNIW, GPI_model.py:1346-1362) and a8 (GPI_model.log_lat_error,
// GPI_model.py:288-323) for 128 < T <= 256 as ONE launch each: one workgroup of four waves per item.
//
// Until round 4 these sizes ran as compositions of the batched kernels (cooperative Cholesky, L^-1 by block columns, one or two
// T^3 products, element-wise reductions: 5-7 launches, 4 T^2 doubles of workspace per item, 0.08 / 0.15 of the fp64 MFMA peak,
// and ~0.5 ms of pure launch latency for the handful of items the online step asks for).  Both terms are traces
//     tr(X^T G^-1 Y) = sum (L^-1 X) o (L^-1 Y),   G = L L^T,
// and a forward solve acts on the columns of its right-hand side independently.  So:
//   1. the workgroup factors G cooperatively (Coop<NB>, tile_f64.hpp) and leaves L in the workspace in MFMA OPERAND order
//      (the accumulator tile of U_KJ is the A operand of L[J, K]: stored as it stands, 2 KB per tile, tile J (J - 1) / 2 + K)
//      next to the inverses W_K of its diagonal blocks - 272 KB per item at T = 256, L2-resident;
//   2. every wave then takes PAIRS of 16-column panels of the right-hand side and forward-solves them by blocks,
//          Y[K] = W_K (B[K] - sum_{K' < K} L[K, K'] Y[K']),
//      with the two panels' 32 tiles of Y in registers and L streamed once per pair through a ring of prefetched tiles (the stream
//      is the storage order: one contiguous read of the packed factor per panel pair, 8 MFMAs per 2 KB);
//   3. what the trace needs is reduced on the spot - Y never goes to memory for a9.
//        a9:  D = M - m_mean, R = I, diagonal prior scale S (the hot path's case; everything else keeps the composition):
//             out = -0.5 |L^-1 D|_F^2 - 0.5 sum_j S_jj |L^-1 e_j|^2:  nb/2 pairs of D panels + nb/2 pairs of identity panels
//             (an identity panel J starts at block row J; its tiles above are zeros and are skipped), dealt to the waves in
//             snake order.  5/3 T^3 flops executed = the yardstick's.
//        a8:  Y = L^-1 A goes to the workspace (packed tiles), z = L^-1 r rides as one more panel, then the Gram form
//             tr(A^T G^-1 A P) = sum_{I <= J} (Y^T Y)_IJ o (P_IJ + P_JI^T)  (7/3 T^3 flops instead of the 11/3 T^3 of solving A and A P).
// Numerics: same factorisation kernel, same regularisation and the same operation order per tile as the composition's
// potrf -> L^-1 -> product chain up to the association of the sums (forward substitution instead of an explicit inverse).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

namespace {

struct MniwCoopArgs {
  const double* M;        // [b,T,T]
  const double* Sigma;    // [b,T,T]
  const double* m_mean;   // [T,T] or [b,T,T] (prior_stride)
  const double* scale;    // diagonal prior scale, same stride
  long prior_stride;
  int T, b;
  double* out;
  int32_t* info;
  double* ws;             // per item: (NB (NB - 1) / 2 + NB) packed tiles of 256 doubles
};

template <int NB>
constexpr size_t packed_doubles() { return (size_t)(NB * (NB - 1) / 2 + NB) * 256; }

// Forward solve of the panel pair (J0, J0 + 1) against the packed factor.  RHS(K, p) -> tile K of panel p (accumulator layout);
// DONE(K, Y0, Y1) consumes the finished tiles of block row K.  kmin: the first block row with a non-zero right-hand side.
template <int NB, int PAN, class RhsFn, class DoneFn>
__device__ __forceinline__ void solve_panels(const double* __restrict__ Lp, const double* __restrict__ Wp, int nb, int kmin, int lane,
                                             RhsFn rhs, DoneFn done) {
  constexpr int RING = 4;
  constexpr int NTL = NB * (NB - 1) / 2;
  d4 Y[PAN][NB];
  d4 ring[RING];
  d4 nx[PAN];                                                // right-hand-side tiles one block row ahead
#pragma unroll
  for (int p = 0; p < PAN; ++p) nx[p] = rhs(kmin, p);
  // Block rows below kmin hold zeros in Y and are skipped altogether; from row kmin on EVERY tile of the packed factor is
  // streamed in storage order (tile q = K (K - 1) / 2 + K'), so that the ring slot of a tile is a compile-time constant; the tiles
  // (K, K' < kmin) of an identity panel are loaded and not multiplied (2 KB from L2 against 4 PAN MFMAs saved).
#pragma unroll
  for (int K = 0; K < NB; ++K) {
    if (K < nb && K >= kmin) {
      const int qrow = K * (K - 1) / 2;
      const d4* Lt = reinterpret_cast<const d4*>(Lp) + launder(lane);     // tile q: Lt[64 q]
      const d4* Wt = reinterpret_cast<const d4*>(Wp) + launder(lane);
      if (K == kmin) {
#pragma unroll
        for (int i = 0; i < RING; ++i)
          if (qrow + i < NTL) ring[(qrow + i) % RING] = Lt[(size_t)64 * (qrow + i)];
      }
      d4 a[PAN];
#pragma unroll
      for (int p = 0; p < PAN; ++p) a[p] = nx[p];
      if (K + 1 < nb) {
#pragma unroll
        for (int p = 0; p < PAN; ++p) nx[p] = rhs(K + 1, p);
      }
      const d4 wk = Wt[(size_t)64 * K];
#pragma unroll
      for (int Kp = 0; Kp < K; ++Kp) {
        const int q = qrow + Kp;
        const d4 lt = ring[q % RING];
        if (q + RING < NTL) ring[q % RING] = Lt[(size_t)64 * (q + RING)];
        if (Kp >= kmin) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int p = 0; p < PAN; ++p) a[p] = mfma_sub(lt[s], Y[p][Kp][s], a[p]);
          }
        }
      }
      d4 y[PAN];
#pragma unroll
      for (int p = 0; p < PAN; ++p) y[p] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int p = 0; p < PAN; ++p) y[p] = mfma(wk[s], a[p][s], y[p]);
      }
#pragma unroll
      for (int p = 0; p < PAN; ++p) Y[p][K] = y[p];
      done(K, y);
    } else {
#pragma unroll
      for (int p = 0; p < PAN; ++p) Y[p][K] = (d4){0.0, 0.0, 0.0, 0.0};
    }
  }
}

// Which factorisation the two kernels run on (round 4, last session): DF = the eight-wave (NB / 2) dataflow factorisation of the
// pair kernels with a packed output (cooph_factor_df<NB, 2>) instead of the four-wave barrier version - the factor was 146 of the
// 385 / 515 us of a call, and the panel solves behind it are spread over twice the waves.  HGP_MATLIK_COOP4=1 keeps the four-wave kernels.
template <int NB, bool DF>
struct MatlikLds {
  static constexpr int NWK = DF ? CoopH<NB>::NW : WAVES;
  // DF: three row buffers + W of every block + per-wave diag16 scratch + red[16] + flags;  else Coop<NB>'s layout
  static constexpr int DOUBLES = DF ? (4 * NB * 256 + NWK * DIAG_SCR + 16 + (16 + NB + 16) / 2 + 2) : Coop<NB>::LDS_DOUBLES;
};

// factor 0.5 (S + S^T) + shift into the packed workspace; returns info (first bad pivot or 0) in every thread
template <int NB, bool DF>
__device__ __forceinline__ int matlik_factor(const double* __restrict__ S, int T, double add, double jitter_rel, double* smem, int wave,
                                             int lane, double* Lp, double* Wp) {
  int info = 0;
  if constexpr (DF) {
    using C = CoopH<NB>;
    double* row0 = smem;
    double* Wall = row0 + 3 * NB * 256;
    double* scr = Wall + NB * 256 + wave * DIAG_SCR;
    double* red = Wall + NB * 256 + C::NW * DIAG_SCR;
    int* flags = reinterpret_cast<int*>(red + 16);
    d4 U[C::NT];
#pragma unroll
    for (int i_ = 0; i_ < C::NT; ++i_) U[i_] = (d4){0.0, 0.0, 0.0, 0.0};
    cooph_load_sym_upper<NB>(U, S, T, T, wave, lane, scr);
    double sh = add;
    if (jitter_rel != 0.0) sh += jitter_rel * fmax(cooph_diag_abs_mean<NB>(U, T, wave, lane, 0.0, red), F64_EPS);
    if (sh != 0.0) cooph_add_diag<NB>(U, sh, T, wave, lane);
    PivotAcc pa;
    pa.init();
    cooph_factor_df<NB, 2>(U, row0, row0 + NB * 256, row0 + 2 * NB * 256, Wall, scr, flags, wave, lane, pa, T, nullptr, nullptr, nullptr,
                           nullptr, Lp, Wp);
    int* redi = flags + 16 + NB;
    if (lane == 0) redi[wave] = pa.info;
    __syncthreads();
    for (int w = 0; w < C::NW; ++w)
      if (redi[w] != 0 && (info == 0 || redi[w] < info)) info = redi[w];
    __syncthreads();
  } else {
    using C = Coop<NB>;
    double* rowbuf = smem;
    double* Rbuf = rowbuf + NB * 256;
    double* Wbuf = Rbuf + NB * 256;
    double* scr = Wbuf + 256;
    double* red = scr + DIAG_SCR;
    int* redi = reinterpret_cast<int*>(red + 8);
    d4 U[C::NT];
    coop_load_sym_upper<NB>(U, S, T, T, wave, lane, rowbuf + wave * DIAG_SCR);
    __syncthreads();   // rowbuf served as per-wave staging for the loader
    double sh = add;
    if (jitter_rel != 0.0) sh += jitter_rel * fmax(coop_diag_abs_mean<NB>(U, T, wave, lane, 0.0, red), F64_EPS);
    if (sh != 0.0) coop_add_diag<NB>(U, sh, T, wave, lane);
    __syncthreads();
    PivotAcc pa;
    pa.init();
    coop_factor<NB, 0>(U, rowbuf, Rbuf, Wbuf, scr, wave, lane, pa, nullptr, 0, T, nullptr, nullptr, 0, Lp, Wp);
    (void)coop_logdet_info(pa, wave, lane, red, redi, info);
  }
  return info;
}

template <int NB, bool DF>
__global__ __launch_bounds__((64 * MatlikLds<NB, DF>::NWK)) void k_coop_mniw(MniwCoopArgs a) {
  constexpr int NWK = MatlikLds<NB, DF>::NWK;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* red = smem + MatlikLds<NB, DF>::DOUBLES;   // [NWK] partial sums (behind whatever the factorisation uses)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int m = blockIdx.x;
  const int T = a.T;
  const long tt = (long)T * T;
  double* Lp = a.ws + (size_t)m * packed_doubles<NB>();
  double* Wp = Lp + (size_t)(NB * (NB - 1) / 2) * 256;
  // chol(0.5 (S + S^T) + 1e-8 I), GPI_model.py:1353
  const int info = matlik_factor<NB, DF>(a.Sigma + (size_t)m * tt, T, 1e-8, 0.0, smem, wave, lane, Lp, Wp);
  __threadfence();     // the packed factor is read back through L2 by all four waves
  __syncthreads();
  const int nb = (T + 15) >> 4;
  const int g = lane >> 4, c = lane & 15;
  const double* Mm = a.M + (size_t)m * tt;
  const double* mean = a.m_mean + (size_t)m * a.prior_stride;
  const double* scale = a.scale + (size_t)m * a.prior_stride;
  constexpr int PAN = DF ? 1 : 2;     // panels solved side by side by a wave (eight waves: 256 registers each, one panel) (the spills of the NB = 16 instance are the factorisation's, as in k_coop_potrf<16>)
  const int ngrp = (nb + PAN - 1) / PAN;
  double acc = 0.0;
  // D panels: groups w, w + 4, ...
  for (int pp = wave; pp < ngrp; pp += NWK) {
    const int J0 = PAN * pp;
    auto rhs = [&](int K, int p) {
      d4 v;
      const int ln = launder(lane);         // keeps the tile addresses of the unrolled solve out of the preheader
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + (ln >> 4) + 4 * r, j = 16 * (J0 + p) + (ln & 15);
        v[r] = (i < T && j < T) ? Mm[(size_t)i * T + j] - mean[(size_t)i * T + j] : 0.0;
      }
      return v;
    };
    auto done = [&](int, const d4 (&y)[PAN]) {
#pragma unroll
      for (int p = 0; p < PAN; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fma(y[p][r], y[p][r], acc);
    };
    solve_panels<NB, PAN>(Lp, Wp, nb, 0, lane, rhs, done);
  }
  // identity panels: sum_j S_jj |L^-1 e_j|^2; panel J costs (nb - J)^2 / 2 tile products: groups dealt in snake order
  for (int t = 0; t < (ngrp + NWK - 1) / NWK; ++t) {
    const int pp = (t & 1) ? (t + 1) * NWK - 1 - wave : t * NWK + wave;
    if (pp >= ngrp) continue;
    const int J0 = PAN * pp;
    double sj[PAN];
#pragma unroll
    for (int p = 0; p < PAN; ++p) {
      const int j = 16 * (J0 + p) + c;
      sj[p] = (j < T) ? scale[(size_t)j * T + j] : 0.0;
    }
    auto rhs = [&](int K, int p) {
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (K == J0 + p && g + 4 * r == c) ? 1.0 : 0.0;
      return v;
    };
    auto done = [&](int, const d4 (&y)[PAN]) {
#pragma unroll
      for (int p = 0; p < PAN; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fma(sj[p] * y[p][r], y[p][r], acc);
    };
    solve_panels<NB, PAN>(Lp, Wp, nb, J0, lane, rhs, done);
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = ((red[0] + red[1]) + red[2]) + red[3];
    for (int w = 4; w < NWK; ++w) tot += red[w];
    a.out[m] = (info != 0) ? __builtin_nan("") : -0.5 * tot;
    if (a.info) a.info[m] = info;
  }
}

template <int NB, bool DF>
int launch_coop_mniw_v(const MniwCoopArgs& a, hipStream_t st) {
  const size_t lds = sizeof(double) * (MatlikLds<NB, DF>::DOUBLES + 8);
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_mniw<NB, DF>), lds)) return rc_;
  hipLaunchKernelGGL((k_coop_mniw<NB, DF>), dim3(a.b), dim3(64 * MatlikLds<NB, DF>::NWK), lds, st, a);
  return launch_status();
}
template <int NB>
int launch_coop_mniw(const MniwCoopArgs& a, hipStream_t st) {
  static const bool four = env_on("HGP_MATLIK_COOP4");
  return four ? launch_coop_mniw_v<NB, false>(a, st) : launch_coop_mniw_v<NB, true>(a, st);
}

// ------------------------------------------------------------------------------------------------------------ a8
struct LatCoopArgs {
  const double* f_cur;    // [b,T]
  const double* f_prev;   // [b,T]
  const double* A;        // [b,T,T]
  const double* Gamma;    // [b,T,T]
  const double* P;        // [b,T,T]  covariance of the previous state
  int T, b;
  double* out;
  int32_t* info;
  double* ws;             // per item: packed factor (NB (NB - 1) / 2 + NB tiles) + NB * NB packed tiles of Y = L^-1 A
};

template <int NB>
constexpr size_t lat_ws_doubles() { return packed_doubles<NB>() + (size_t)NB * NB * 256; }

template <int NB, bool DF>
__global__ __launch_bounds__((64 * MatlikLds<NB, DF>::NWK)) void k_coop_lat(LatCoopArgs a) {
  constexpr int NWK = MatlikLds<NB, DF>::NWK;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* red = smem + MatlikLds<NB, DF>::DOUBLES;   // [NWK] partial sums
  double* rvec = red + 8;                            // [16 NB]: r = f_cur - A f_prev (zero padded)
  double* stage = smem + (DF ? 4 * NB * 256 : 0);    // [NWK][DIAG_SCR]: per-wave 16 x 18 staging tiles of the Gram phase (the factorisation's diag16 scratch / row buffer, free by then)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int m = blockIdx.x;
  const int T = a.T;
  const long tt = (long)T * T;
  const int nb = (T + 15) >> 4;
  double* Lp = a.ws + (size_t)m * lat_ws_doubles<NB>();
  double* Wp = Lp + (size_t)(NB * (NB - 1) / 2) * 256;
  double* Yp = Lp + packed_doubles<NB>();            // tile (K, J) at K * NB + J, accumulator layout
  const double* Am = a.A + (size_t)m * tt;
  // r = f_cur - A f_prev on the VALU: wave w takes the rows w, w + 4, ... (64 consecutive elements per load, one wave reduction per row)
  {
    const double* fp = a.f_prev + (size_t)m * T;
    const double* fc = a.f_cur + (size_t)m * T;
    double fpv[NB / 4];
#pragma unroll
    for (int u = 0; u < NB / 4; ++u) fpv[u] = (64 * u + lane < T) ? fp[64 * u + lane] : 0.0;
    for (int i = wave; i < 16 * NB; i += NWK) {
      double sacc = 0.0;
      if (i < T) {
#pragma unroll
        for (int u = 0; u < NB / 4; ++u) sacc = fma((64 * u + lane < T) ? Am[(size_t)i * T + 64 * u + lane] : 0.0, fpv[u], sacc);
      }
      sacc = wave_sum(sacc);
      if (lane == 0) rvec[i] = (i < T) ? fc[i] - sacc : 0.0;
    }
  }
  // (no barrier here: r is consumed behind the factorisation's barriers, and the waves that finish their rows of r early start on Gamma)
  // _chol_spd(Gamma): + 1e-8 max(mean |diag|, eps) I  (GPI_model.py:83-87,312)
  int info;
  if constexpr (DF) {
    info = matlik_factor<NB, true>(a.Gamma + (size_t)m * tt, T, 0.0, 1e-8, smem, wave, lane, Lp, Wp);
  } else {   // (written out: through the helper the NB = 16 instance compiles 8 % slower - 0.54 vs 0.50 ms per 256 items)
    using C = Coop<NB>;
    double* rowbuf = smem;
    double* Rbuf = rowbuf + NB * 256;
    double* Wbuf = Rbuf + NB * 256;
    double* scr = Wbuf + 256;
    double* redf = scr + DIAG_SCR;
    int* redi = reinterpret_cast<int*>(redf + 8);
    d4 U[C::NT];
    coop_load_sym_upper<NB>(U, a.Gamma + (size_t)m * tt, T, T, wave, lane, rowbuf + wave * DIAG_SCR);
    __syncthreads();   // rowbuf served as per-wave staging for the loader
    {
      const double dm = coop_diag_abs_mean<NB>(U, T, wave, lane, 0.0, redf);
      coop_add_diag<NB>(U, 1e-8 * fmax(dm, F64_EPS), T, wave, lane);
    }
    __syncthreads();
    PivotAcc pa;
    pa.init();
    coop_factor<NB, 0>(U, rowbuf, Rbuf, Wbuf, scr, wave, lane, pa, nullptr, 0, T, nullptr, nullptr, 0, Lp, Wp);
    (void)coop_logdet_info(pa, wave, lane, redf, redi, info);
  }
  __threadfence();
  __syncthreads();
  constexpr int PAN = DF ? 1 : 2;
  const int ngrp = (nb + PAN - 1) / PAN;
  double acc = 0.0;
  // Y = L^-1 A by panel pairs, stored as packed accumulator tiles
  for (int pp = wave; pp < ngrp; pp += NWK) {
    const int J0 = PAN * pp;
    auto rhs = [&](int K, int p) {
      d4 v;
      const int ln = launder(lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + (ln >> 4) + 4 * r, j = 16 * (J0 + p) + (ln & 15);
        v[r] = (i < T && j < T) ? Am[(size_t)i * T + j] : 0.0;
      }
      return v;
    };
    auto done = [&](int K, const d4 (&y)[PAN]) {
      const int ln = launder(lane);
#pragma unroll
      for (int p = 0; p < PAN; ++p)
        if (J0 + p < NB) *reinterpret_cast<d4*>(Yp + ((size_t)(K * NB + J0 + p) * 64 + ln) * 4) = y[p];
    };
    solve_panels<NB, PAN>(Lp, Wp, nb, 0, lane, rhs, done);
  }
  // z = L^-1 r (one more panel, column 0; the wave with the fewest panel pairs takes it)
  if (wave == (ngrp % NWK)) {
    auto rhs = [&](int K, int) {
      d4 v;
      const int ln = launder(lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = ((ln & 15) == 0) ? rvec[16 * K + (ln >> 4) + 4 * r] : 0.0;
      return v;
    };
    auto done = [&](int, const d4 (&y)[1]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = fma(y[0][r], y[0][r], acc);
    };
    solve_panels<NB, 1>(Lp, Wp, nb, 0, lane, rhs, done);
  }
  __threadfence();
  __syncthreads();
  // Gram form: tr(A^T G^-1 A P) = sum_{I <= J} (Y^T Y)_IJ o Pw_IJ,  Pw_IJ = P_IJ + P_JI^T (I < J), P_II (I = J).
  // 2 x 2 blocks of tile pairs per step: four tiles of Y per block row feed sixteen MFMAs.
  {
    const double* Pm = a.P + (size_t)m * tt;
    const int nb2 = (nb + 1) >> 1;
    const int nblk = nb2 * (nb2 + 1) / 2;
    double* scw = stage + wave * DIAG_SCR;           // this wave's 16 x 18 staging tile (transposes of P)
    for (int t = wave; t < nblk; t += NWK) {
      int Ib = 0, rem = t;                           // block (Ib, Jb), Ib <= Jb, row-major over the upper triangle
      while (rem >= nb2 - Ib) {
        rem -= nb2 - Ib;
        ++Ib;
      }
      const int Jb = Ib + rem;
      const int I0 = 2 * Ib, J0 = 2 * Jb;
      d4 G[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) G[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
      const d4* Yt = reinterpret_cast<const d4*>(Yp) + lane;
      d4 yi[2], yj[2], ni[2], nj[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        ni[x] = Yt[(size_t)64 * (0 * NB + I0 + x)];
        nj[x] = Yt[(size_t)64 * (0 * NB + J0 + x)];
      }
      for (int K = 0; K < nb; ++K) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          yi[x] = ni[x];
          yj[x] = nj[x];
        }
        if (K + 1 < nb) {
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            ni[x] = Yt[(size_t)64 * ((K + 1) * NB + I0 + x)];
            nj[x] = Yt[(size_t)64 * ((K + 1) * NB + J0 + x)];
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) G[x][y] = mfma(yi[x][s], yj[y][s], G[x][y]);
      }
      const int g = lane >> 4, c = lane & 15;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          const int I = I0 + x, J = J0 + y;
          if (I > J || J >= nb) continue;            // lower tile of a diagonal block / padding
          d4 pn, pt;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + g + 4 * r, j = 16 * J + c;
            pn[r] = (i < T && j < T) ? Pm[(size_t)i * T + j] : 0.0;
            const int i2 = 16 * J + g + 4 * r, j2 = 16 * I + c;
            pt[r] = (I < J && i2 < T && j2 < T) ? Pm[(size_t)i2 * T + j2] : 0.0;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) scw[(g + 4 * r) * DIAG_LD + c] = pt[r];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = fma(G[x][y][r], pn[r] + scw[c * DIAG_LD + g + 4 * r], acc);
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  acc = wave_sum(acc);
  __syncthreads();
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = ((red[0] + red[1]) + red[2]) + red[3];
    for (int w = 4; w < NWK; ++w) tot += red[w];
    a.out[m] = (info != 0) ? __builtin_nan("") : -0.5 * tot;
    if (a.info) a.info[m] = info;
  }
}

template <int NB, bool DF>
int launch_coop_lat_v(const LatCoopArgs& a, hipStream_t st) {
  const size_t lds = sizeof(double) * (MatlikLds<NB, DF>::DOUBLES + 8 + 16 * NB);
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_lat<NB, DF>), lds)) return rc_;
  hipLaunchKernelGGL((k_coop_lat<NB, DF>), dim3(a.b), dim3(64 * MatlikLds<NB, DF>::NWK), lds, st, a);
  return launch_status();
}
template <int NB>
int launch_coop_lat(const LatCoopArgs& a, hipStream_t st) {
  // the eight-wave form is the faster CALL (281 vs 341 us for a few items); with a workgroup on every CU the four-wave form's larger
  // register budget wins for a8 (0.515 vs 0.543 ms per 256 items at T = 256)
  static const bool four = env_on("HGP_MATLIK_COOP4");
  return (four || a.b >= 200) ? launch_coop_lat_v<NB, false>(a, st) : launch_coop_lat_v<NB, true>(a, st);
}

}  // namespace

size_t hgp_internal_lat_coop_ws_doubles(int T) { return T <= 192 ? lat_ws_doubles<12>() : lat_ws_doubles<16>(); }

// a8, 128 < T <= 256
int hgp_internal_lat_coop(const double* f_cur, const double* f_prev, const double* A, const double* Gamma, const double* P, int T, int b,
                          double* out, int32_t* info, double* ws, hipStream_t st) {
  LatCoopArgs a{f_cur, f_prev, A, Gamma, P, T, b, out, info, ws};
  return T <= 192 ? launch_coop_lat<12>(a, st) : launch_coop_lat<16>(a, st);
}

size_t hgp_internal_matlik_coop_ws_doubles(int T) { return T <= 192 ? packed_doubles<12>() : packed_doubles<16>(); }

// a9, identity right covariance and diagonal prior scale, 128 < T <= 256
int hgp_internal_mniw_coop(const double* M, const double* Sigma, const double* m_mean, const double* scale, long prior_stride, int T, int b,
                           double* out, int32_t* info, double* ws, hipStream_t st) {
  MniwCoopArgs a{M, Sigma, m_mean, scale, prior_stride, T, b, out, info, ws};
  return T <= 192 ? launch_coop_mniw<12>(a, st) : launch_coop_mniw<16>(a, st);
}
