// Solve-based evaluation of the per-(segment, cluster) path (a2 + a5) for clusters whose kernel matrix K~ is
// ill-conditioned (length-scale >> grid spacing, e.g. the drivers' ini_lengthscale = 3.0 on a unit-spaced grid).
//
// The explicit operator M' = c^2 (K~^-1 Sigma K~^-1 - K~^-1) of the fast kernels (hgp_kernels.hip) squares the condition
// number of K~: its rounding error relative to the reference's triangular solves grows like eps (c ||K~^-1||)^2 - 1e-11 at
// ell = 1.2 but 1e-3 at ell = 3.  Here the reference's own operation order is kept (GPI.py:489-501):
//     L L^T = K~;   S = L^-T (L^-1 K*)   (cholesky_solve, GPI.py:492);   f* = S^T m;
//     cov = K** - K*^T S + S^T Sigma S = K** + S^T (Sigma S - K*)
// with the two triangular solves done by SUBSTITUTION, 4 rows at a time (the k-dimension of v_mfma_f64_16x16x4_f64):
// inside a 16-row block, sub-step r multiplies rows 4r..4r+3 by the inverse of their 4x4 diagonal block (one MFMA whose
// A operand is that inverse placed at rows 4r..) and eliminates them from the rows below (one MFMA, A operand = the
// masked columns 4r..4r+3 of L_KK) - the scheme of diag16 in tile_f64.hpp with the right-hand side riding along.
// Multiplying by explicit 16x16 (or whole) inverses instead loses 2-3 digits at ell = 3 (measured against the reference:
// 5e-9 vs 5e-11); 4x4 inverses do not.
//
// Work split: one workgroup of NB/2 waves per pair (CoopH<NB>, as k_pairs_cooph).  The solves are independent per column
// of K*, so wave w solves the two 16-column panels it owns (w and NB-1-w) entirely in registers, publishes its panels of
// S in the workgroup's scratch area (L2-resident; [tile][lane][4] = MFMA operand order), and after one barrier builds the
// covariance tiles of its own block columns: cov[I][J] = K**[I][J] + sum_K S[K,I]^T Q[K,J], Q[K,J] = (Sigma S - K*)[K,J].
// Then the cooperative factorisation of tile_f64.hpp (cooph_factor) with the single right-hand side d = y - f*.
// Workgroups are persistent (grid = acc_grid_for(NB)) and walk the flagged (segment, cluster) pairs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

namespace {

__host__ __device__ constexpr int ntl(int NB) { return NB * (NB - 1) / 2; }          // strictly lower tiles
__host__ __device__ constexpr int tl(int K, int Kp) { return K * (K - 1) / 2 + Kp; }   // K > Kp

// ------------------------------------------------------------------------------------------------ flags
// scal[8k+7] = 1 when cluster k takes the solve-based kernel; acc_list = [count, ids...] (ascending).
__global__ void k_acc_flags(double* scal, int K, double tol, int32_t* acc_list) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0;
  for (int k = 0; k < K; ++k) {
    const double ck = scal[8 * k] * scal[8 * k + 6];
    const double bound = F64_EPS * ck * ck;
    const bool flag = (tol == 0.0) || (tol > 0.0 && !(bound <= tol));   // NaN bound -> solve-based
    scal[8 * k + 7] = flag ? 1.0 : 0.0;
    if (flag) acc_list[1 + n++] = k;
  }
  acc_list[0] = n;
}

// ------------------------------------------------------------------------------------------------ factor (T <= 128)
// L = chol(K~) for the flagged clusters, one wave per cluster, written over the lower triangle of A (the upper
// triangle keeps K~; nothing reads it afterwards).  For T > 128 the plan update has already factored in place.
template <int NB>
__global__ __launch_bounds__(64) void k_acc_factor(double* A, int TP, const double* scal, int32_t* info) {
  __shared__ __attribute__((aligned(16))) double scr[DIAG_SCR];
  const int k = blockIdx.x, lane = threadIdx.x;
  if (scal[8 * k + 7] == 0.0) return;
  double* Ak = A + (size_t)k * TP * TP;
  d4 U[NB * (NB + 1) / 2];
  d4 R[NB];
  load_upper_only<NB>(U, Ak, TP, TP, lane);
  PivotAcc pa;
  pa.init();
  wave_factor<NB, 0>(U, R, scr, nullptr, nullptr, lane, pa, Ak, TP, TP);
  if (lane == 0 && info && pa.info != 0) info[k] = pa.info;
}

// ------------------------------------------------------------------------------------------------ operand packing
struct AccPrepArgs {
  const double* L;       // [K,TP,TP] lower factor of K~ (identity padded)
  const double* S;       // [K,TP,TP] 0.5 (Sigma + Sigma^T), zero padded, exactly symmetric
  const double* mean;    // [K,T]
  const double* scal;
  int T, TP, NB;
  double* Lop;
  double* LTop;
  double* Dop;
  double* Sop;
  double* mu;            // [K,TP]
};

__global__ __launch_bounds__(256) void k_acc_prep(AccPrepArgs a) {
  const int k = blockIdx.x;
  if (a.scal[8 * k + 7] == 0.0) return;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int wv = blockIdx.y * 4 + (threadIdx.x >> 6), nwv = gridDim.y * 4;
  const int NB = a.NB, TP = a.TP;
  const double* L = a.L + (size_t)k * TP * TP;
  const double* S = a.S + (size_t)k * TP * TP;
  const int nl = ntl(NB);
  double* Lop = a.Lop + (size_t)k * nl * 256;
  double* LTop = a.LTop + (size_t)k * nl * 256;
  double* Dop = a.Dop + (size_t)k * 4 * NB * 256;
  double* Sop = a.Sop + (size_t)k * NB * NB * 256;
  if (blockIdx.y == 0)
    for (int i = threadIdx.x; i < TP; i += 256) a.mu[(size_t)k * TP + i] = (i < a.T) ? a.mean[(size_t)k * a.T + i] : 0.0;
  const int items = nl + NB + NB * NB;
  for (int it = wv; it < items; it += nwv) {
    if (it < nl) {                       // strictly lower tile (Kb, Kp)
      int Kb = 1;
      while (tl(Kb + 1, 0) <= it) ++Kb;
      const int Kp = it - tl(Kb, 0);
      d4 lo, lt;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        lo[s] = L[(size_t)(16 * Kb + c) * TP + 16 * Kp + 4 * s + g];      // A operand of L[Kb,Kp]:   A[c][4s+g]
        lt[s] = L[(size_t)(16 * Kb + 4 * s + g) * TP + 16 * Kp + c];      // A operand of L[Kb,Kp]^T: A[c][4s+g] = L[4s+g][c]
      }
      *reinterpret_cast<d4*>(Lop + ((size_t)it * 64 + lane) * 4) = lo;
      *reinterpret_cast<d4*>(LTop + ((size_t)it * 64 + lane) * 4) = lt;
    } else if (it < nl + NB) {           // diagonal block Kb: 4x4 inverses and masked columns / rows
      const int Kb = it - nl;
      const double* D = L + (size_t)(16 * Kb) * TP + 16 * Kb;
      d4 a1, a2, a1t, a2t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* B = D + (size_t)(4 * r) * TP + 4 * r;
        const double l00 = B[0], l10 = B[TP], l11 = B[TP + 1], l20 = B[2 * TP], l21 = B[2 * TP + 1], l22 = B[2 * TP + 2],
                     l30 = B[3 * (size_t)TP], l31 = B[3 * (size_t)TP + 1], l32 = B[3 * (size_t)TP + 2], l33 = B[3 * (size_t)TP + 3];
        double w[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) w[i][j] = 0.0;
        // column j of W4 = L44^{-1}: forward substitution of e_j
        const double r0 = 1.0 / l00, r1 = 1.0 / l11, r2 = 1.0 / l22, r3 = 1.0 / l33;
        w[0][0] = r0;
        w[1][0] = -(l10 * w[0][0]) * r1;
        w[2][0] = -(l20 * w[0][0] + l21 * w[1][0]) * r2;
        w[3][0] = -(l30 * w[0][0] + l31 * w[1][0] + l32 * w[2][0]) * r3;
        w[1][1] = r1;
        w[2][1] = -(l21 * w[1][1]) * r2;
        w[3][1] = -(l31 * w[1][1] + l32 * w[2][1]) * r3;
        w[2][2] = r2;
        w[3][2] = -(l32 * w[2][2]) * r3;
        w[3][3] = r3;
        const int ci = c & 3;
        double wf = 0.0, wt = 0.0;   // W4[ci][g] and W4[g][ci]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (i == ci && j == g) wf = w[i][j];
            if (i == g && j == ci) wt = w[i][j];
          }
        const bool mine = (c >> 2) == r;
        a1[r] = mine ? wf : 0.0;
        a1t[r] = mine ? wt : 0.0;
        a2[r] = (c >= 4 * r + 4) ? D[(size_t)c * TP + 4 * r + g] : 0.0;          // L_KK[c][4r+g], rows below
        a2t[r] = (c < 4 * r) ? D[(size_t)(4 * r + g) * TP + c] : 0.0;            // L_KK^T[c][4r+g], rows above
      }
      *reinterpret_cast<d4*>(Dop + ((size_t)(0 * NB + Kb) * 64 + lane) * 4) = a1;
      *reinterpret_cast<d4*>(Dop + ((size_t)(1 * NB + Kb) * 64 + lane) * 4) = a2;
      *reinterpret_cast<d4*>(Dop + ((size_t)(2 * NB + Kb) * 64 + lane) * 4) = a1t;
      *reinterpret_cast<d4*>(Dop + ((size_t)(3 * NB + Kb) * 64 + lane) * 4) = a2t;
    } else {                             // tile (Kb, Kp) of the symmetrised Sigma as an A operand
      const int t = it - nl - NB, Kb = t / NB, Kp = t % NB;
      d4 so;
#pragma unroll
      for (int s = 0; s < 4; ++s) so[s] = S[(size_t)(16 * Kp + 4 * s + g) * TP + 16 * Kb + c];   // = S[16Kb+c][16Kp+4s+g]
      *reinterpret_cast<d4*>(Sop + ((size_t)t * 64 + lane) * 4) = so;
    }
  }
}

// ------------------------------------------------------------------------------------------------ the pair kernel
struct AccArgs {
  const double* x;
  const double* y;
  int N, Ts;
  const double* xb;
  int T;
  const double* scal;
  const double* mu;
  const double* Lop;
  const double* LTop;
  const double* Dop;
  const double* Sop;
  const int32_t* acc_list;
  const double* first_noise;
  const int32_t* sel;
  int K;
  double* out_quad;
  double* out_logdet;
  int32_t* out_info;
  double* sscr;
  double score_add;   // out_quad = score_on ? -0.5 quad + score_add : quad (hgp_pairs_plan_set_score_output)
  int score_on;
};

template <int NB>
struct AccLds {
  static constexpr int TP = 16 * NB;
  static constexpr size_t DOUBLES = (size_t)NB * 256 /*rowbuf*/ + 256 /*Wbuf*/ + 5 * TP /*dvec xs ys xbs mus*/ + DIAG_SCR + 32;
  static constexpr size_t BYTES = DOUBLES * sizeof(double) + 16 * sizeof(int);
};

__device__ __forceinline__ d4 ld4(const double* base, unsigned off) { return *reinterpret_cast<const d4*>(base + off); }

template <int NB>
__global__ __launch_bounds__(64 * CoopH<NB>::NW, (NB <= 8) ? 2 : 1) void k_pairs_acc(AccArgs a) {
  using H = CoopH<NB>;
  constexpr int NW = H::NW, TP = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* rowbuf = smem;                // [NB][4][64]
  double* Wbuf = rowbuf + NB * 256;
  double* dvec = Wbuf + 256;            // d, then z = L^{-1} d
  double* xs = dvec + TP;               // segment grid / ell (sentinel padded)
  double* ys = xs + TP;
  double* xbs = ys + TP;                // basis grid / ell (sentinel padded)
  double* mus = xbs + TP;               // prior mean on the basis grid (zero padded)
  double* scr = mus + TP;
  double* red = scr + DIAG_SCR;         // 32 doubles
  int* redi = reinterpret_cast<int*>(red + 32);
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, Ts = a.Ts;
  const int nflag = a.acc_list[0];
  if (nflag == 0) return;
  const long npairs = a.sel ? (long)a.N : (long)a.N * nflag;
  double* Ssc = a.sscr + (size_t)blockIdx.x * NB * NB * 256;

  for (long p = blockIdx.x; p < npairs; p += gridDim.x) {
    int n, kc;
    if (a.sel) {
      n = (int)p;
      kc = a.sel[n];
      if (a.scal[8 * kc + 7] == 0.0) continue;   // uniform over the workgroup
    } else {
      n = (int)(p / nflag);
      kc = a.acc_list[1 + (int)(p % nflag)];
    }
    kc = __builtin_amdgcn_readfirstlane(kc);
    const double* sc = a.scal + 8 * kc;
    const double cc = sc[0], ell = sc[1], noise = sc[2];
    const bool iso = sc[3] != 0.0;
    const size_t oidx = a.sel ? (size_t)n : (size_t)n * a.K + kc;
    const double fn = a.first_noise ? a.first_noise[oidx] : 0.0;
    __syncthreads();   // the previous pair is done with the LDS vectors
    for (int i = tid; i < TP; i += 64 * NW) {   // sentinel padding: every kernel entry that touches a padded point is exp(-huge) = 0
      xs[i] = (i < Ts) ? a.x[(size_t)n * Ts + i] / ell : 1e150 * (double)(1 + i);
      ys[i] = (i < Ts) ? a.y[(size_t)n * Ts + i] : 0.0;
      xbs[i] = (i < T) ? a.xb[i] / ell : -1e150 * (double)(1 + i);
      mus[i] = a.mu[(size_t)kc * TP + i];
    }
    __syncthreads();
    // uniform (scalar) base pointers + one per-lane offset: the tile loads then address as s[base] + v_off + imm instead
    // of one hoisted 64-bit VGPR address per tile of the unrolled loops (which spilled hundreds of registers)
    const double* Lop = a.Lop + (size_t)kc * ntl(NB) * 256;
    const double* LTop = a.LTop + (size_t)kc * ntl(NB) * 256;
    const double* Dop = a.Dop + (size_t)kc * 4 * NB * 256;
    const double* Sop = a.Sop + (size_t)kc * NB * NB * 256;
    double* Sl = Ssc;

    // ---- phase A: S[:, J] = L^-T L^-1 K*[:, J] for my two column panels, d_J = y_J - S[:, J]^T m
    double dsq = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int J = (q == 0) ? wave : NB - 1 - wave;
      d4 X[NB];
      // forward substitution  L V = K*
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const unsigned l4 = (unsigned)launder(lane) * 4u;   // (opaque: keeps the per-tile addresses out of the loop preheader)
        {   // right-hand side tile K*[16K + 4r + g][16J + c] (two-argument kernel call: no white noise), built right before its use
          const int ln = launder(lane);
          const double xj = xs[16 * J + (ln & 15)];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double u = xbs[16 * K + 4 * r + (ln >> 4)] - xj;
            X[K][r] = cc * exp(-0.5 * (u * u));
          }
        }
#pragma unroll
        for (int Kp = 0; Kp < K; ++Kp) {
          const d4 lo = ld4(Lop + tl(K, Kp) * 256, l4);
#pragma unroll
          for (int s = 0; s < 4; ++s) X[K] = mfma_sub(lo[s], X[Kp][s], X[K]);
        }
        const d4 a1 = ld4(Dop + (0 * NB + K) * 256, l4), a2 = ld4(Dop + (1 * NB + K) * 256, l4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d4 t = X[K];
          t[r] = 0.0;
          t = mfma(a1[r], X[K][r], t);                        // rows 4r.. := W4 rows 4r..
          X[K] = (r < 3) ? mfma_sub(a2[r], t[r], t) : t;      // rows below -= L_KK[., 4r..] new rows
        }
      }
      // backward substitution  L^T S = V
#pragma unroll
      for (int K = NB - 1; K >= 0; --K) {
        const unsigned l4 = (unsigned)launder(lane) * 4u;
#pragma unroll
        for (int Kp = NB - 1; Kp > K; --Kp) {
          const d4 lt = ld4(LTop + tl(Kp, K) * 256, l4);
#pragma unroll
          for (int s = 0; s < 4; ++s) X[K] = mfma_sub(lt[s], X[Kp][s], X[K]);
        }
        const d4 a1 = ld4(Dop + (2 * NB + K) * 256, l4), a2 = ld4(Dop + (3 * NB + K) * 256, l4);
#pragma unroll
        for (int r = 3; r >= 0; --r) {
          d4 t = X[K];
          t[r] = 0.0;
          t = mfma(a1[r], X[K][r], t);                        // rows 4r.. := W4^T rows 4r..
          X[K] = (r > 0) ? mfma_sub(a2[r], t[r], t) : t;      // rows above -= L_KK[4r.., .]^T new rows
        }
      }
      double pj = 0.0;
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const int ln = launder(lane);
        const unsigned l4 = (unsigned)ln * 4u;
#pragma unroll
        for (int r = 0; r < 4; ++r) pj = fma(X[K][r], mus[16 * K + 4 * r + (ln >> 4)], pj);
        *reinterpret_cast<d4*>(Sl + (K * NB + J) * 256 + l4) = X[K];
      }
      pj = xrow_sum(pj);
      if (g == 0) {
        const int j = 16 * J + c;
        const double d = ys[j] - pj;     // padded entries: 0 - 0
        dvec[j] = d;
        dsq = fma(d, d, dsq);
      }
    }
    __syncthreads();   // S (global scratch, same CU) and dvec are visible to the whole workgroup

    if (iso) {   // GPI.py:497-498: cov_f = mean(diag Sigma) I
      dsq = wave_sum(dsq);
      if (lane == 0) red[wave] = dsq;
      __syncthreads();
      if (tid == 0) {
        const double v = sc[4] + fn;
        const double v2 = v + 1e-8 * fmax(fabs(v), F64_EPS);
        double tot_ = 0.0;
        for (int w_ = 0; w_ < NW; ++w_) tot_ += red[w_];
        a.out_quad[oidx] = a.score_on ? fma(-0.5, tot_ / v2, a.score_add) : (tot_ / v2);
        if (a.out_logdet) a.out_logdet[oidx] = (double)Ts * log(v2);
        if (a.out_info) a.out_info[oidx] = (v2 > 0.0) ? 0 : 1;
      }
      continue;
    }

    // ---- phase B: cov[I][J] = K**[I][J] + sum_K S[K,I]^T Q[K,J],  Q = Sigma S - K*,  for my columns
    d4 U[H::NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int J = (q == 0) ? wave : NB - 1 - wave;
      const double xj = xs[16 * J + c];
#pragma unroll
      for (int I = 0; I < (q == 0 ? NW : NB); ++I) {
        const int ln = launder(lane);
        d4 kt = (d4){0.0, 0.0, 0.0, 0.0};
        if (I <= J) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double u = xs[16 * I + (ln >> 4) + 4 * r] - xs[16 * J + (ln & 15)];
            kt[r] = cc * exp(-0.5 * (u * u));
          }
          if (I == J) {   // exact diagonal of the one-argument kernel call (GPI.py:476); identity on the padding
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((ln >> 4) + 4 * r == (ln & 15)) kt[r] = (16 * I + (ln & 15) < Ts) ? cc + noise : 1.0;
          }
          U[q == 0 ? H::slotA(I) : H::slotB(I)] = kt;
        }
      }
#pragma nounroll
      for (int K = 0; K < NB; ++K) {
        const unsigned l4 = (unsigned)launder(lane) * 4u;
        d4 Q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double u = xbs[16 * K + 4 * r + g] - xj;
          Q[r] = -(cc * exp(-0.5 * (u * u)));
        }
#pragma unroll
        for (int Kp = 0; Kp < NB; ++Kp) {
          const d4 so = ld4(Sop + (K * NB + Kp) * 256, l4);
          const d4 sb = ld4(Sl + (Kp * NB + J) * 256, l4);
#pragma unroll
          for (int s = 0; s < 4; ++s) Q = mfma(so[s], sb[s], Q);
        }
#pragma unroll
        for (int I = 0; I < (q == 0 ? NW : NB); ++I) {
          if (I <= J) {
            const d4 sa = ld4(Sl + (K * NB + I) * 256, l4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
              U[q == 0 ? H::slotA(I) : H::slotB(I)] = mfma(sa[s], Q[s], U[q == 0 ? H::slotA(I) : H::slotB(I)]);
          }
        }
      }
    }

    // regularisation of the reference: +1e-6 I (GPI.py:501), + first, + 1e-8 mean|diag| I (GPI_model.py:83-87)
    {
      const double sh = 1e-6 + fn;
      const double dm = cooph_diag_abs_mean<NB>(U, Ts, wave, lane, sh, red);
      cooph_add_diag<NB>(U, sh + 1e-8 * fmax(dm, F64_EPS), Ts, wave, lane);
    }
    PivotAcc pa;
    pa.init();
    double zq = cooph_factor<NB>(U, rowbuf, Wbuf, scr, wave, lane, pa, Ts, dvec);
    int info;
    const double ld = cooph_logdet_info<NB>(pa, wave, lane, red, redi, info);
    zq = wave_sum(zq);
    if (lane == 0) red[8 + wave] = zq;
    __syncthreads();
    if (tid == 0) {
      double tot_ = 0.0;
      for (int w_ = 0; w_ < NW; ++w_) tot_ += red[8 + w_];
      a.out_quad[oidx] = a.score_on ? fma(-0.5, tot_, a.score_add) : (tot_);
      if (a.out_logdet) a.out_logdet[oidx] = ld;
      if (a.out_info) a.out_info[oidx] = info;
    }
  }
}

template <int NB>
int launch_pairs_acc(const AccArgs& a, hipStream_t st) {
  const size_t lds = AccLds<NB>::BYTES;
  hipLaunchKernelGGL(k_pairs_acc<NB>, dim3(acc_grid_for(NB)), dim3(64 * CoopH<NB>::NW), lds, st, a);
  return launch_status();
}

}  // namespace

// sizes[0..5]: Lop, LTop, Dop, Sop, mu, sscr (bytes)
size_t hgp_internal_acc_bytes(int TP, int K, size_t* sizes) {
  const int NB = TP / 16;
  size_t s[6];
  s[0] = (size_t)K * ntl(NB) * 256 * sizeof(double);
  s[1] = s[0];
  s[2] = (size_t)K * 4 * NB * 256 * sizeof(double);
  s[3] = (size_t)K * NB * NB * 256 * sizeof(double);
  s[4] = (size_t)K * TP * sizeof(double);
  s[5] = (size_t)acc_grid_for(NB) * NB * NB * 256 * sizeof(double);
  size_t tot = 0;
  for (int i = 0; i < 6; ++i) {
    if (sizes) sizes[i] = s[i];
    tot += (s[i] + 255) & ~(size_t)255;
  }
  return tot;
}

int hgp_internal_acc_prep(hgp_pairs_plan* p, const double* mean, hipStream_t st) {
  const int K = p->K, TP = p->TP, NB = p->NB;
  hipLaunchKernelGGL(k_acc_flags, dim3(1), dim3(64), 0, st, p->d_scal, K, p->acc_tol, p->d_acc_list);
  if (p->acc_tol < 0.0) return launch_status();   // explicit operator everywhere: nothing else to prepare
  switch (NB) {   // T <= 128: the plan update computed L^{-1} only; T > 128: d_A already holds L
    case 2: hipLaunchKernelGGL(k_acc_factor<2>, dim3(K), dim3(64), 0, st, p->d_A, TP, p->d_scal, (int32_t*)nullptr); break;
    case 4: hipLaunchKernelGGL(k_acc_factor<4>, dim3(K), dim3(64), 0, st, p->d_A, TP, p->d_scal, (int32_t*)nullptr); break;
    case 6: hipLaunchKernelGGL(k_acc_factor<6>, dim3(K), dim3(64), 0, st, p->d_A, TP, p->d_scal, (int32_t*)nullptr); break;
    case 8: hipLaunchKernelGGL(k_acc_factor<8>, dim3(K), dim3(64), 0, st, p->d_A, TP, p->d_scal, (int32_t*)nullptr); break;
    default: break;
  }
  AccPrepArgs a{p->d_A, p->d_S, mean, p->d_scal, p->T, TP, NB, p->d_Lop, p->d_LTop, p->d_Dop, p->d_Sop, p->d_mu};
  hipLaunchKernelGGL(k_acc_prep, dim3(K, 8), dim3(256), 0, st, a);
  return launch_status();
}

int hgp_internal_pairs_acc(const hgp_pairs_plan* p, const double* x, const double* y, int N, int Ts, const double* first_noise,
                           const int32_t* sel, double* out_quad, double* out_logdet, int32_t* out_info, hipStream_t st) {
  if (p->acc_tol < 0.0) return 0;
  AccArgs a{x, y, N, Ts, p->d_xb, p->T, p->d_scal, p->d_mu, p->d_Lop, p->d_LTop, p->d_Dop, p->d_Sop, p->d_acc_list,
            first_noise, sel, p->K, out_quad, out_logdet, out_info, p->d_sscr, -0.5 * (double)Ts * 1.8378770664093453, p->score_out};
  switch (p->NB) {
    case 2: return launch_pairs_acc<2>(a, st);
    case 4: return launch_pairs_acc<4>(a, st);
    case 6: return launch_pairs_acc<6>(a, st);
    case 8: return launch_pairs_acc<8>(a, st);
    case 12: return launch_pairs_acc<12>(a, st);
    default: return launch_pairs_acc<16>(a, st);
  }
}
