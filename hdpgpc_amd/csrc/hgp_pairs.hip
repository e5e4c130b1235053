// a2 + a5, the per-pair kernels of hgp_loglik_pairs_f64 (explicit-operator evaluation of cov_f): k_pairs<NB> (T <= 128, one
// wavefront per pair) and k_pairs_cooph<NB> (NB/2 waves per pair) for T <= 256.  (The 4-wave cooperative kernel of round 1,
// k_pairs_coop, 1.28x slower, was removed in round 3.)
// The plan (per-cluster operators) and the C-ABI live in hgp_kernels.hip; the solve-based kernel in hgp_pairs_acc.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

#ifndef HGP_PAIRS_DIAG_MFMA
#define HGP_PAIRS_DIAG_MFMA 0   // 1 = the MFMA-blocked diag16 also at NB = 8: measured 1.5 % faster, but 43 spilled VGPRs turn into 59 MB of scratch writes per launch (WRITE_SIZE): not kept
#endif
// in-situ knock-out experiments (diagnostic builds only; results are wrong by construction)
#ifdef HGP_EXP_NOEXP
#define HGP_EXP4(h, o) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) (o)[i_] = 1.0 / (1.0 + (h)[i_])
#else
#define HGP_EXP4(h, o) exp_neg4(h, o)
#endif

namespace {

// -------------------------------------------------------------------------------------- a2 + a5

// PAIRS_CUT (block-wise cut-off of E and K**): hgp_internal.hpp

constexpr int PAIRS_DCOLS = 16;   // clusters whose d = y - E^T a' one pass of the workgroup prepares (one MFMA column block)

// Waves per workgroup of k_pairs<NB, BAND>.  One workgroup per segment, its waves take the clusters round-robin; E (16 NB)^2 doubles
// of LDS) allows one workgroup per CU from NB = 6 on, i.e. ONE wave per SIMD at four waves: every latency of the pair - the pivot
// chain of the six diag16 first of all - is idle time.  The band kernel at NB = 6 (the records' T = 90) fits 256 registers once
// the sweep-1 operand ring holds two half-blocks instead of eight, so EIGHT waves share the segment's E: two per SIMD.
#ifndef HGP_PAIRS_W8_NB
#define HGP_PAIRS_W8_NB 6   // the NB whose band kernel runs eight waves per workgroup (0: none)
#endif
template <int NB, bool BAND>
constexpr int pairs_waves() { return (BAND && NB == HGP_PAIRS_W8_NB) ? 8 : WAVES; }
template <int NB, int PW = WAVES>
constexpr size_t pairs_lds_bytes() {
  return sizeof(double) * ((size_t)(16 * NB) * (16 * NB) + 3 * 16 * NB + PW * DIAG_SCR + PAIRS_DCOLS * 16 * NB) +
         sizeof(int) * 32;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The two sweeps of k_pairs for the BLOCK-TRIDIAGONAL E of the reference's setting (length-scale 1.2 on a unit-spaced grid, any
// jitter below the cut-off radius: blocks |Kt - J| <= 1 of E and tiles J - 1 <= I <= J of K** active, nothing else), as
// straight-line code.  The workgroup checks its masks against this pattern (from the data, as ever) and takes the generic
// mask-driven sweeps below for anything else.  What the static schedule buys (round 3):
//  * sweep 1 computes only the row tiles of B_J = (M'E)[:, J] that sweep 2 reads (Kt <= J + 1): 476 instead of 704 MFMAs per pair
//    at NB = 8 - the generic code can only drop whole halves, predicating single row tiles costs more in branches than it saves;
//  * no bit scans, no scalar branches, no conditional refills around the operand ring: one flat list of (column, half, k-block,
//    half-block) items, item t multiplied from ring slot t % 4 and the slot refilled with item t + 4 (exact s_waitcnt counts);
//  * the K** seeds and sweep 2 of a pass follow its last item with compile-time tile lists.
// Same operations in the same order per accumulator as the generic sweeps: bit-identical covariances, factors and
// log-determinants (HGP_PAIRS_GENERIC=1 runs the generic code for A/B; tests/test_gpu_edge_cases.py compares the two); the
// quadratic form agrees to rounding (the band kernel defers the cross-row sums of the right-hand side, wave_factor RHS_DEFER).
// ---------------------------------------------------------------------------------------------------------------------------
template <int NB>
struct PairsBand {
  static constexpr int NH = NB / 2;
  static constexpr int lo(int J) { return J > 0 ? J - 1 : 0; }
  static constexpr int hi(int J) { return J + 1 < NB ? J + 1 : NB - 1; }
  static constexpr int emask(int J) { return ((1 << (hi(J) + 1)) - 1) & ~((1 << lo(J)) - 1); }   // blocks (Kt, J) of E
  static constexpr int kmask(int J) { return ((1 << (J + 1)) - 1) & ~((1 << lo(J)) - 1); }        // tiles (I, J) of K**
  // k-step s (rows 4 s .. 4 s + 3) of block (Kt, J) of E can hold an entry above the cut-off: with the cut at 2^-80 the band is
  // |k - j| <= 12 grid points (length-scale 1.2, unit spacing, any jitter below 0.3), so the first k-step of the block above the
  // diagonal one and the last k-step of the block below it are zero: 10 of 12 k-steps per column panel (-1/6 of both sweeps)
  static constexpr bool alive(int Kt, int J, int s) { return Kt == J || (Kt == J - 1 && s >= 1) || (Kt == J + 1 && s <= 2); }
  static constexpr unsigned emask4(int J) {   // bit 4 Kt + s
    unsigned m = 0;
    for (int Kt = lo(J); Kt <= hi(J); ++Kt)
      for (int s = 0; s < 4; ++s)
        if (alive(Kt, J, s)) m |= 1u << (4 * Kt + s);
    return m;
  }
  static constexpr int rows(int J, int h) {   // row tiles NH h + i of B[:, J] that sweep 2 reads: NH h + i <= J + 1
    int m = 0;
    for (int i = 0; i < NH; ++i)
      if (NH * h + i <= hi(J)) m |= 1 << i;
    return m;
  }
  struct Item {
    int J, h, Lt, hb, first, last, valid;
  };
  static constexpr Item item(int t) {
    int cnt = 0;
    for (int J = 0; J < NB; ++J)
      for (int h = 0; h < 2; ++h) {
        if (!rows(J, h)) continue;
        for (int Lt = lo(J); Lt <= hi(J); ++Lt)
          for (int hb = 0; hb < 2; ++hb) {
            if (cnt == t) return Item{J, h, Lt, hb, Lt == lo(J) && hb == 0, Lt == hi(J) && hb == 1, 1};
            ++cnt;
          }
      }
    return Item{0, 0, 0, 0, 0, 0, 0};
  }
  // sweep 2 of pass (J, h) as a flat list of groups (row tile i of the panel, column I <= J) with block (NH h + i, I) of E active
  struct Group {
    int i, I;
  };
  static constexpr int ngroups(int J, int h) {
    int cnt = 0;
    for (int i = 0; i < NH; ++i)
      if ((rows(J, h) >> i) & 1)
        for (int I = 0; I <= J; ++I)
          if ((emask(I) >> (NH * h + i)) & 1) ++cnt;
    return cnt;
  }
  static constexpr Group group(int J, int h, int n) {
    int cnt = 0;
    for (int i = 0; i < NH; ++i)
      if ((rows(J, h) >> i) & 1)
        for (int I = 0; I <= J; ++I)
          if ((emask(I) >> (NH * h + i)) & 1) {
            if (cnt == n) return Group{i, I};
            ++cnt;
          }
    return Group{0, 0};
  }
  static constexpr int nitems() {
    int cnt = 0;
    for (int J = 0; J < NB; ++J)
      for (int h = 0; h < 2; ++h)
        if (rows(J, h)) cnt += 2 * (hi(J) - lo(J) + 1);
    return cnt;
  }
};

template <int NB, int RD_>
__device__ __forceinline__ void band_sweeps(d4 (&cov)[NB * (NB + 1) / 2], const double* __restrict__ Mu, const double* E,
                                            int lane_in, double cc, double noise, int Ts
#ifdef HGP_STAMPS
                                            , unsigned long long& hgp_t_, unsigned long long (&hgp_acc_)[12]
#endif
) {
  using PB = PairsBand<NB>;
  constexpr int TP = 16 * NB, NH = NB / 2, NI = PB::nitems();
  constexpr int RD = RD_;   // ring slots (half-blocks in flight): NB <= 6 has the registers for eight at one wave per SIMD
  double ra[RD][2][NH], re[RD][2];
  d4 BJ[NH];
  // Every operand address of the sweeps is  (uniform base + compile-time constant) + ONE of two lane offsets: Mu is uniform
  // (the cluster index comes through v_readfirstlane), so the M' loads take the scalar-base form of global_load and the E reads
  // an immediate offset - the ~14 VALU instructions of address arithmetic per ring fill (1.5 k per pair) are gone.
  const unsigned moff0 = (unsigned)((lane_in >> 4) * TP + 2 * (lane_in & 15));   // row g of M', tile-pair interleaved column 2c
  const unsigned eoff0 = (unsigned)((lane_in >> 4) * TP + (lane_in & 15));       // row g, column c (E, and the odd last tile of M')
  auto fill = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr auto it = PB::item(t);
    if constexpr (it.valid) {
      constexpr int slot = t % RD, rm = PB::rows(it.J, it.h);
      const unsigned moff = (unsigned)launder((int)moff0), eoff = (unsigned)launder((int)eoff0);   // opaque per fill: no merging / hoisting of the loads of different fills
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        if (!PB::alive(it.Lt, it.J, 2 * it.hb + s_)) continue;
        const double* row_ = Mu + 16 * NH * it.h + (size_t)(16 * it.Lt + 4 * (2 * it.hb + s_)) * TP;
#pragma unroll
        for (int P_ = 0; P_ < NH / 2; ++P_) {
          if ((rm >> (2 * P_)) & 3) {
            const d2 t_ = *reinterpret_cast<const d2*>(row_ + 32 * P_ + moff);
            ra[slot][s_][2 * P_] = t_[0];
            ra[slot][s_][2 * P_ + 1] = t_[1];
          }
        }
        if ((NH & 1) && ((rm >> (NH - 1)) & 1)) ra[slot][s_][NH - 1] = (row_ + 16 * (NH - 1))[eoff];
        re[slot][s_] = (E + (16 * it.Lt + 4 * (2 * it.hb + s_)) * TP + 16 * it.J)[eoff];
      }
    }
  };
  static_for<0, RD>([&](auto tc) { fill(tc); });
  static_for<0, NI>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr auto it = PB::item(t);
    constexpr int slot = t % RD, rm = PB::rows(it.J, it.h), J = it.J, h = it.h;
    if constexpr (it.first) {
#pragma unroll
      for (int i = 0; i < NH; ++i) BJ[i] = (d4){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      if (!PB::alive(it.Lt, it.J, 2 * it.hb + s_)) continue;
#pragma unroll
      for (int i = 0; i < NH; ++i)
        if ((rm >> i) & 1) BJ[i] = mfma(ra[slot][s_][i], re[slot][s_], BJ[i]);
    }
    fill(std::integral_constant<int, t + RD>{});
    if constexpr (it.last) {
      const int g = lane_in >> 4, c = lane_in & 15;
      const unsigned eoff = (unsigned)launder((int)eoff0);
      HGP_ACC(1);
      constexpr int NG = PB::ngroups(J, h);
      double af[2][4];
      auto ldaf = [&](auto nc) {
        constexpr int n_ = decltype(nc)::value;
        if constexpr (n_ < NG) {
          constexpr auto gp = PB::group(J, h, n_);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (PB::alive(NH * h + gp.i, gp.I, r)) af[n_ & 1][r] = (E + (16 * (NH * h + gp.i) + 4 * r) * TP + 16 * gp.I)[eoff];
        }
      };
      ldaf(std::integral_constant<int, 0>{});
      if constexpr (h == 0) {   // K** seeds of column J (cached tiles, see k_pairs) + the exact diagonal
#pragma unroll
        for (int I = 0; I <= J; ++I) {
          d4 kt = (d4){0.0, 0.0, 0.0, 0.0};
          if ((PB::kmask(J) >> I) & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kt[r] = cc * (E + (16 * ((I + NH) % NB) + 4 * r) * TP + 16 * J)[eoff];
          }
          if (I == J) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (g + 4 * r == c) kt[r] = (16 * I + c < Ts) ? cc + noise : 1.0;
          }
          cov[tix(I, J, NB)] = kt;
        }
      }
      HGP_ACC(2);
      // sweep 2: cov[I][J] += E[Kt, I]^T BJ[Kt] over the blocks |Kt - I| <= 1, I <= J.  The A operand of group n + 1 is requested
      // from LDS before the four MFMAs of group n are issued (the compiler left the reads one MFMA ahead of their use:
      // 41 k cycles for 396 MFMAs, 25 k at the issue rate)
      {
        static_for<0, NG>([&](auto nc) {
          constexpr int n_ = decltype(nc)::value;
          constexpr auto gp = PB::group(J, h, n_);
          ldaf(std::integral_constant<int, n_ + 1>{});
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (PB::alive(NH * h + gp.i, gp.I, r)) cov[tix(gp.I, J, NB)] = mfma(af[n_ & 1][r], BJ[gp.i][r], cov[tix(gp.I, J, NB)]);
        });
      }
      HGP_ACC(3);
    }
  });
}

// One workgroup per segment n; its 4 waves take the clusters of the length-scale group round-robin.
// E_n = exp(-0.5 ((xb_k - x_j)/ell)^2) is built once per workgroup in LDS (active 16x16 blocks only) and shared by
// the waves; each wave then evaluates one (segment, cluster) pair entirely in its own registers:
//   cov = c R_n + noise I + E^T M'_k E   (two MFMA sweeps per column panel, the first result feeding the second
//   straight from its accumulators), regularise, factor (wave_factor), eliminate d on the VALU, reduce.
//
// Two instantiations (round 3): k_pairs<NB, true> holds ONLY the static sweeps of a block-tridiagonal E (band_sweeps); a workgroup
// whose masks do not match that pattern appends its segment to the fall-back list a.fb and leaves, and k_pairs<NB, false> (the
// mask-driven sweeps, launched right behind on the same grid) takes the listed segments; its other workgroups leave at once.  With both sweep codes in one
// kernel the allocator spilled 126 VGPRs at NB = 8 (284 B of scratch per lane, 189 MB of scratch writes per launch, WRITE_SIZE);
// apart they need none.  Without a list (a.fb == nullptr: NB < 6, HGP_PAIRS_GENERIC=1) the generic kernel takes every segment.
#ifndef HGP_PAIRS_OCC2_NB
#define HGP_PAIRS_OCC2_NB 4   // largest NB built for two workgroups per CU (<= 256 VGPRs)
#endif
template <int NB, bool BAND>
__global__ __launch_bounds__((64 * pairs_waves<NB, BAND>()), ((NB <= HGP_PAIRS_OCC2_NB) ? 2 : 1)) void k_pairs(PairsArgs a) {   // T <= 64: two workgroups per CU (the kernel sat 3 registers above that limit)
  constexpr int PW = pairs_waves<NB, BAND>();
  constexpr int TP = 16 * NB;
  constexpr int NH = NB / 2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* E = smem;               // [TP][TP]: row k = basis point, column j = segment point
  double* xs = E + TP * TP;       // segment grid / ell
  double* ys = xs + TP;
  double* xbs = ys + TP;          // basis grid / ell
  const int tid = threadIdx.x, wave = tid >> 6;
  double* scr = xbs + TP + wave * DIAG_SCR;
  double* dall = xbs + TP + PW * DIAG_SCR;   // [PAIRS_DCOLS][TP]: d = y - E^T a' of the clusters of the current chunk, then z = L^{-1} d
  int* amask = reinterpret_cast<int*>(dall + PAIRS_DCOLS * TP);   // bit Kt of amask[J]: block (Kt, J) of E active
  const int T = a.T, Ts = a.Ts;
  HGP_STAMP_DECL
  HGP_T0();
  int n = blockIdx.x, total = 0;
  if constexpr (!BAND) {
    if (a.fb) {   // the list is complete (previous kernel on the stream); workgroups beyond it have nothing to do
      total = min(a.fb[0], PAIRS_FB_CAP);
      if (n >= total) return;
      n = a.fb[1 + n];
    }
  }

  // Padding (i >= Ts, k >= T) uses far-apart sentinels instead of bounds predicates: every kernel entry that
  // involves a padded point is then exp(-huge) = 0 by itself (all differences stay finite: < 3e152).
  for (int i = tid; i < TP; i += 64 * PW) {
    xs[i] = (i < Ts) ? a.x[(size_t)n * Ts + i] / a.ell : 1e150 * (double)(1 + i);
    ys[i] = (i < Ts) ? a.y[(size_t)n * Ts + i] : 0.0;
    xbs[i] = (i < T) ? a.xb[i] / a.ell : -1e150 * (double)(1 + i);
  }
  int* kmask = amask + 8;   // bit I of kmask[J]: tile (I, J) of K** has an entry above the cut-off (I <= J)
  int* amask4 = amask + 16;  // bit 4 Kt + s of amask4[J]: k-step s of block (Kt, J) of E has an entry above the cut-off
  if (tid < 24) amask[tid] = 0;
  __syncthreads();
#ifdef HGP_STAMPS
  { unsigned long long n_ = __builtin_readcyclecounter(); hgp_acc_[8] += n_ - hgp_t_; }
#endif
  {
    // Wave w owns the column blocks Jb = w, w + PW, ...: it tests every block (Kt, Jb) of E and every tile (I <= Jb, Jb) of K**
    // against the cut-off and builds the active blocks.  The lane's rows of both grids sit in registers after ONE LDS round trip
    // (the loop over blocks used to wait for five LDS reads per block: 19 k cycles for the two test loops, in-kernel stamps),
    // and the masks of a column are plain stores by its owner (no atomics).
    const int lane = tid & 63, g = lane >> 4, c = lane & 15;
    double xbr[NB][4], xsr[NB][4];
#pragma unroll
    for (int Kt = 0; Kt < NB; ++Kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xbr[Kt][r] = xbs[16 * Kt + g + 4 * r];
        xsr[Kt][r] = xs[16 * Kt + g + 4 * r];
      }
    }
    for (int Jb = wave; Jb < NB; Jb += PW) {
      const int j = 16 * Jb + c;
      const double xj = xs[j];
      unsigned am = 0, a4 = 0, km = 0;
#pragma unroll
      for (int Kt = 0; Kt < NB; ++Kt) {
        double h[4];
        bool near = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double u = xbr[Kt][r] - xj;
          h[r] = 0.5 * (u * u);
          near = near || (h[r] < PAIRS_CUT);
        }
        if (__any(near)) {   // entries below the cut-off are exact zeros: a k-step of the block made of zeros only can be skipped bit for bit
          double ev[4];
          HGP_EXP4(h, ev);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            E[(16 * Kt + g + 4 * r) * TP + j] = (h[r] < PAIRS_CUT) ? ev[r] : 0.0;
            if (__any(h[r] < PAIRS_CUT)) a4 |= 1u << ((4 * Kt + r) & 31);
          }
          am |= 1u << Kt;
        }
        if (Kt <= Jb) {   // the same test for tile (Kt, Jb) of K** (upper tiles)
          bool nk = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double u = xsr[Kt][r] - xj;
            nk = nk || (0.5 * (u * u) < PAIRS_CUT);
          }
          if (__any(nk)) km |= 1u << Kt;
        }
      }
      if (lane == 0) {
        amask[Jb] = (int)am;
        kmask[Jb] = (int)km;
        if (NB <= 8) amask4[Jb] = (int)a4;
      }
    }
  }
  __syncthreads();

#ifdef HGP_STAMPS
  { unsigned long long n_ = __builtin_readcyclecounter(); hgp_acc_[9] += n_ - hgp_t_; }
#endif
  // K** = c exp(-0.5 (x_i - x_j)^2 / ell^2) depends on the segment only: its active tiles (without the factor c) are built ONCE
  // per workgroup and kept in blocks of the E array that E itself does not use - tile (I, J) in block ((I + NB/2) % NB, J) -
  // instead of 4 exp per lane and tile in every pair (16 k of a pair's 246 k cycles at T = 128).  Decided from the
  // data: if any home block is an active block of E (dense grids, NB < 6) the pairs compute the tiles themselves as before.
  // (all masks are read first and combined with bit operations: the short-circuit form made 24 dependent LDS round trips, 3.7 k cycles)
  int kbad = 0, bbad = 0;
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    const int km = kmask[J], am = amask[J], a4 = (NB <= 8) ? amask4[J] : 0;
    const int rot = ((km << NH) | (km >> (NB - NH))) & ((1 << NB) - 1);
    kbad |= rot & am;
    // block-tridiagonal E and K** (the reference's setting) with every live k-step one that the static sweeps multiply
    bbad |= (am ^ PairsBand<NB>::emask(J)) | (km ^ PairsBand<NB>::kmask(J)) | (int)((unsigned)a4 & ~PairsBand<NB>::emask4(J));
  }
  const bool kcache = (NB >= 6) && __builtin_amdgcn_readfirstlane(kbad) == 0;
  // -> the static sweeps (band_sweeps) instead of the mask-driven ones
  const bool band = (NB >= 6) && kcache && !(a.flags & 1) && __builtin_amdgcn_readfirstlane(bbad) == 0;
  if constexpr (BAND) {
    if (!band) {   // not the static pattern: this segment goes to the generic kernel
      if (tid == 0) {   // (the count is clamped on both sides: counters left over from a launch that died must not index past the list)
        const int slot = atomicAdd(&a.fb[0], 1);
        if (slot < PAIRS_FB_CAP) a.fb[1 + slot] = n;
      }
      return;
    }
  }
#ifdef HGP_STAMPS
  { unsigned long long n_ = __builtin_readcyclecounter(); hgp_acc_[10] += n_ - hgp_t_; }
#endif
  if (kcache) {
    const int lane = tid & 63, g = lane >> 4, c = lane & 15;
    for (int t = wave; t < NB * NB; t += PW) {
      const int I = t / NB, J = t % NB;
      if (I > J || !((kmask[J] >> I) & 1)) continue;
      const int Kh = (I + NH) % NB;
      double hk[4], ev[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double u = xs[16 * I + g + 4 * r] - xs[16 * J + c];
        hk[r] = 0.5 * (u * u);
      }
      HGP_EXP4(hk, ev);
#pragma unroll
      for (int r = 0; r < 4; ++r) E[(16 * Kh + g + 4 * r) * TP + 16 * J + c] = (hk[r] < PAIRS_CUT) ? ev[r] : 0.0;
    }
  }

  HGP_ACC(6);   // prologue: loads, E / K** build, masks (stamps build only)
  const int Kg = a.kend - a.kbeg;
  for (int ch = 0; ch < Kg; ch += PAIRS_DCOLS) {
  // d = y - E^T a'  (a' = c K~^{-1} mean) for the clusters ch .. ch + 15 of the group at once, on the matrix core: column c of
  // the B operand is cluster c's a', the A operand is the transposed block of E - one MFMA chain per block column of E,
  // shared by all the workgroup's pairs (was: 8 LDS-fed dot products per pair, 10 k cycles each).
  if (ch) __syncthreads();
  {
    HGP_T0();
    const int lane = launder(tid) & 63;
    const int g = lane >> 4, c = lane & 15;
    const bool col = ch + c < Kg;
    const double* apc = a.ap + (size_t)(col ? a.perm[a.kbeg + ch + c] : 0) * TP + g;
    for (int Jb = wave; Jb < NB; Jb += PW) {
      const int mJ = __builtin_amdgcn_readfirstlane(amask[Jb]);
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int Kt = 0; Kt < NB; ++Kt) {
        if (mJ & (1 << Kt)) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double bv = apc[16 * Kt + 4 * s];
            acc = mfma(E[(16 * Kt + 4 * s + g) * TP + 16 * Jb + c], col ? bv : 0.0, acc);
          }
        }
      }
      if (col) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dall[c * TP + 16 * Jb + g + 4 * r] = ys[16 * Jb + g + 4 * r] - acc[r];   // padded entries: 0 - 0
      }
    }
    HGP_ACC(0);
  }
  __syncthreads();
  const int kk_end = (a.kbeg + ch + PAIRS_DCOLS < a.kend) ? a.kbeg + ch + PAIRS_DCOLS : a.kend;
  for (int kk = a.kbeg + ch + wave; kk < kk_end; kk += PW) {
    const int lane = launder(tid) & 63;
    const int g = lane >> 4, c = lane & 15;
    const int kc = BAND ? __builtin_amdgcn_readfirstlane(a.perm[kk]) : a.perm[kk];   // band kernel: uniform, the per-cluster bases are scalar
    if (a.sel && a.sel[n] != kc) continue;
    HGP_T0();
    const double* sc = a.scal + 8 * kc;
    if (sc[7] != 0.0) continue;   // ill-conditioned K~: this cluster is scored by the solve-based kernel (hgp_pairs_acc.hip)
    const double cc = sc[0], noise = sc[2];
    const bool iso = sc[3] != 0.0;
    const size_t oidx = a.sel ? (size_t)n : (size_t)n * a.K + kc;
    const double fn = a.first_noise ? a.first_noise[oidx] : 0.0;
    int msk[NB], kmsk[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      msk[J] = __builtin_amdgcn_readfirstlane(amask[J]);
      kmsk[J] = __builtin_amdgcn_readfirstlane(kmask[J]);
    }
    double* dv = dall + (kk - a.kbeg - ch) * TP;

    if (iso) {   // GPI.py:497-498: cov_f = mean(diag Sigma) I
      double dsq = 0.0;
#pragma unroll
      for (int i = 0; i < TP; i += 64) {
        const double d = (i + lane < TP) ? dv[(i + lane < TP) ? i + lane : 0] : 0.0;
        dsq = fma(d, d, dsq);
      }
      double v = sc[4] + fn;
      double v2 = v + 1e-8 * fmax(fabs(v), F64_EPS);
      double q = wave_sum(dsq) / v2;
      if (lane == 0) {
        a.out_quad[oidx] = a.score_on ? fma(-0.5, q, a.score_add) : (q);
        if (a.out_logdet) a.out_logdet[oidx] = (double)Ts * log(v2);
        if (a.out_info) a.out_info[oidx] = (v2 > 0.0) ? 0 : 1;
      }
      continue;
    }

    // cov tiles (upper).  Each tile starts as K** = c exp(-0.5 (x_i - x_j)^2 / ell^2) + noise I (the one-argument
    // kernel call, GPI.py:476), computed right before the first MFMA that accumulates into it.
    d4 cov[NB * (NB + 1) / 2];
    // cov[I][J] += sum_h E[rows h, I]^T (M'[rows h, :] E[:, J]) : the basis index is split in two halves
    // so the intermediate panel is 4 tiles; it feeds the second sweep straight from its accumulators.
#ifdef HGP_EXP_SHARED_M   // experiment: every wave of the chip streams the SAME M' (wrong results; upper bound of M' locality)
    const double* Mbase = a.Mp + (size_t)g * TP;
#else
    const double* Mbase = a.Mp + (size_t)kc * TP * TP + (size_t)g * TP;   // + column offset inside HGP_FILL (interleaved)
#endif
    if constexpr (BAND) {
      band_sweeps<NB, (PW > WAVES) ? 2 : ((NB <= 6) ? 8 : 4)>(cov, a.Mp + (size_t)kc * TP * TP, E, lane, cc, noise, Ts
#ifdef HGP_STAMPS
                      , hgp_t_, hgp_acc_
#endif
      );
    } else {
    // Sweep-1 operand ring: 4 slots of half a k-block each (2 k-steps: 2 x NH rows of M' from L2 + 2 values of E
    // from LDS).  A slot is refilled right after its MFMAs are issued, i.e. three half-blocks (about 1.5k cycles
    // of MFMA) before it is used again; the first two blocks of the NEXT sweep are requested at the end of a sweep.
    double ra[4][2][NH], re[4][2];
    // Row half h of the panel B_J = (M'E)[:, J] is only read by sweep 2 through the active blocks (NH h + Kt, I) of E with
    // I <= J: with a banded E the lower half is dead for the first panels.  Such half-sweeps run with an empty block mask
    // (-5.5 %; predicating the individual row tiles inside a half costs more in scalar branches than the MFMAs it saves).
    int pm[NB];
    pm[0] = msk[0];
#pragma unroll
    for (int J = 1; J < NB; ++J) pm[J] = pm[J - 1] | msk[J];
    constexpr int HALF = (1 << NH) - 1;
    int kA = -1, kB = -1, m = (pm[0] & HALF) ? msk[0] : 0;
#ifdef HGP_EXP_NOFILL   // knock-out: sweep-1 operands are constants (no global loads, no LDS reads, no address arithmetic)
#define HGP_FILL(slot, half, blk, Mptr, Jcol)                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                        \
    _Pragma("unroll") for (int P_ = 0; P_ < NH; ++P_) ra[slot][s_][P_] = 1e-3 * (double)(blk);              \
    re[slot][s_] = 1e-3 * (double)(Jcol);                                                                   \
  }
#else
#define HGP_FILL(slot, half, blk, Mptr, Jcol)                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                        \
    const double* row_ = (Mptr) + (size_t)(16 * (blk) + 4 * (2 * (half) + s_)) * TP;                        \
    _Pragma("unroll") for (int P_ = 0; P_ < NH / 2; ++P_) {                                                 \
      const d2 t_ = *reinterpret_cast<const d2*>(row_ + 32 * P_ + 2 * c);                                   \
      ra[slot][s_][2 * P_] = t_[0];                                                                         \
      ra[slot][s_][2 * P_ + 1] = t_[1];                                                                     \
    }                                                                                                       \
    if (NH & 1) ra[slot][s_][NH - 1] = row_[16 * (NH - 1) + c];                                             \
    re[slot][s_] = E[(16 * (blk) + 4 * (2 * (half) + s_) + g) * TP + 16 * (Jcol) + c];                      \
  }
#endif
#define HGP_MMA(slot)                                                                                       \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                        \
    _Pragma("unroll") for (int I_ = 0; I_ < NH; ++I_) BJ[I_] = mfma(ra[slot][s_][I_], re[slot][s_], BJ[I_]); \
  }
    if (m) {
      kA = __builtin_ctz(m);
      m &= m - 1;
      HGP_FILL(0, 0, kA, Mbase, 0)
      HGP_FILL(1, 1, kA, Mbase, 0)
    }
    if (m) {
      kB = __builtin_ctz(m);
      m &= m - 1;
      HGP_FILL(2, 0, kB, Mbase, 0)
      HGP_FILL(3, 1, kB, Mbase, 0)
    }
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // sweep 1: BJ = M'[16 NH h .. , :] E[:, J] over the ACTIVE k-blocks of column panel J.  M' is symmetric, so
        // row-tile I of the A operand is read as M'[k][16 I + c]: 128 contiguous bytes per 16 lanes, from L2.
        const double* Mk = Mbase + 16 * NH * h;
        d4 BJ[NH];
#pragma unroll
        for (int I = 0; I < NH; ++I) BJ[I] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma nounroll
        while (kA >= 0) {
          int kC = -1, kD = -1;
          if (m) {
            kC = __builtin_ctz(m);
            m &= m - 1;
          }
          HGP_MMA(0)
          if (kC >= 0) { HGP_FILL(0, 0, kC, Mk, J) }
          HGP_MMA(1)
          if (kC >= 0) { HGP_FILL(1, 1, kC, Mk, J) }
          if (kB < 0) {
            kA = kC;
            break;
          }
          if (m) {
            kD = __builtin_ctz(m);
            m &= m - 1;
          }
          HGP_MMA(2)
          if (kD >= 0) { HGP_FILL(2, 0, kD, Mk, J) }
          HGP_MMA(3)
          if (kD >= 0) { HGP_FILL(3, 1, kD, Mk, J) }
          kA = kC;
          kB = kD;
        }
        kA = -1;
        kB = -1;
        if (!(J == NB - 1 && h == 1)) {   // request the first two active blocks of the next sweep now
          const int Jn = (h == 0) ? J : J + 1, hn = (h == 0) ? 1 : 0;
          const int Jc = Jn < NB ? Jn : 0;
          const double* Mn = Mbase + 16 * NH * hn;
          m = (pm[Jc] & (HALF << (NH * hn))) ? msk[Jc] : 0;
          if (m) {
            kA = __builtin_ctz(m);
            m &= m - 1;
            HGP_FILL(0, 0, kA, Mn, Jc)
            HGP_FILL(1, 1, kA, Mn, Jc)
          }
          if (m) {
            kB = __builtin_ctz(m);
            m &= m - 1;
            HGP_FILL(2, 0, kB, Mn, Jc)
            HGP_FILL(3, 1, kB, Mn, Jc)
          }
        }
        HGP_ACC(1);
        if (h == 0) {
#pragma unroll
          for (int I = 0; I <= J; ++I) {
            const int ln = launder(lane);
            d4 kt = (d4){0.0, 0.0, 0.0, 0.0};
            if (kmsk[J] & (1 << I)) {
              if (kcache) {
#pragma unroll
                for (int r = 0; r < 4; ++r) kt[r] = cc * E[(16 * ((I + NH) % NB) + (ln >> 4) + 4 * r) * TP + 16 * J + (ln & 15)];
              } else {
                double hk[4], ev[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const double u = xs[16 * I + (ln >> 4) + 4 * r] - xs[16 * J + (ln & 15)];
                  hk[r] = 0.5 * (u * u);
                }
                HGP_EXP4(hk, ev);
#pragma unroll
                for (int r = 0; r < 4; ++r) kt[r] = (hk[r] < PAIRS_CUT) ? cc * ev[r] : 0.0;
              }
            }
            if (I == J) {   // exact diagonal of the one-argument kernel call; identity on the padding
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if ((ln >> 4) + 4 * r == (ln & 15)) kt[r] = (16 * I + (ln & 15) < Ts) ? cc + noise : 1.0;
            }
            cov[tix(I, J, NB)] = kt;
          }
        }
        HGP_ACC(2);
        // sweep 2: cov[I][J] += E[rows h, I]^T BJ over the active blocks (Kt, I) of E; the B operand is the
        // accumulator of sweep 1, untouched.
#pragma unroll
        for (int Kt = 0; Kt < NH; ++Kt) {
#pragma unroll
          for (int I = 0; I <= J; ++I) {
            // operands are read unconditionally (an inactive block holds stale LDS bytes that are never multiplied):
            // the reads can then be issued ahead of the branch and overlap the previous block's MFMAs
            double af[4];
#pragma unroll
#ifdef HGP_EXP_NOAF   // knock-out: sweep-2 A operands are constants (no LDS reads, no address arithmetic)
            for (int r = 0; r < 4; ++r) af[r] = 1e-3 * (double)(Kt + I + r);
#else
            for (int r = 0; r < 4; ++r) af[r] = E[(16 * (NH * h + Kt) + 4 * r + g) * TP + 16 * I + c];
#endif
            if (msk[I] & (1 << (NH * h + Kt))) {
#pragma unroll
              for (int r = 0; r < 4; ++r) cov[tix(I, J, NB)] = mfma(af[r], BJ[Kt][r], cov[tix(I, J, NB)]);
            }
          }
        }
        HGP_ACC(3);
      }
    }
    }   // generic (mask-driven) sweeps

    // regularisation of the reference: +1e-6 I (GPI.py:501), + first, + 1e-8 mean|diag| I (GPI_model.py:83-87)
    {
      const double sh = 1e-6 + fn;
      const double dm = diag_abs_mean<NB>(cov, Ts, lane, sh);
      add_diag<NB>(cov, sh + 1e-8 * fmax(dm, F64_EPS), Ts, lane);
    }
    PivotAcc pa;
    pa.init();
    d4 Rnone[NB];
    HGP_ACC(4);
#ifdef HGP_EXP_NORHS   // knock-out: factor only, no right-hand side (diagnostic builds only)
    wave_factor<NB, 0, (NB >= 8)>(cov, Rnone, scr, nullptr, dv, lane, pa, nullptr, 0, Ts);
    const double q = dv[lane];
#elif defined(HGP_EXP_NOFACTOR)   // knock-out: no factorisation at all
    double q = dv[lane];
#pragma unroll
    for (int i_ = 0; i_ < NB * (NB + 1) / 2; ++i_) q += cov[i_][0] + cov[i_][1] + cov[i_][2] + cov[i_][3];
#else
    const double q = wave_factor<NB, 2, (NB >= 8) && !HGP_PAIRS_DIAG_MFMA, BAND || (NB < 8)>(cov, Rnone, scr, nullptr, dv, lane, pa, nullptr, 0, Ts);
#endif
    if (lane == 0) {
      a.out_quad[oidx] = a.score_on ? fma(-0.5, q, a.score_add) : (q);
      if (a.out_logdet) a.out_logdet[oidx] = pa.logdet();
      if (a.out_info) a.out_info[oidx] = pa.info;
    }
    HGP_ACC(5);
#ifdef HGP_STAMPS
    hgp_acc_[7] += pa.diag_cycles;
#endif
  }
  }   // chunks of PAIRS_DCOLS clusters
#ifdef HGP_STAMPS
  if ((tid & 63) == 0 && a.stamps)
    for (int i = 0; i < 12; ++i) atomicAdd(&a.stamps[i], hgp_acc_[i]);
#endif
  if constexpr (!BAND) {   // the last LISTED workgroup to finish hands the list back empty; an unlisted one that reads the count after that
                           // sees 0 and leaves as it would have before
    if (a.fb && tid == 0) {
      __threadfence();
      if (atomicAdd(&a.fb[1 + PAIRS_FB_CAP], 1) == total - 1) {
        a.fb[0] = 0;
        a.fb[1 + PAIRS_FB_CAP] = 0;
      }
    }
  }
}

// ------------------------------------------------------------ a2 + a5, cooperative: one workgroup per pair
// For 128 < T <= 256 a pair does not fit one wave (136 tiles at T = 256): the NB/2 waves of a workgroup share ONE (segment,
// cluster) pair (k_pairs_cooph below).  E_n lives in LDS in COMPACT form: only the 16x16 blocks with an entry above the
// cut-off get a slot (2 KB each, [slot][k-step][lane] = the MFMA operand order of both sweeps).  With the reference's
// length-scale (1.2 on a unit-spaced grid) 3 NB - 2 blocks are active (46 at T = 256, 92 KB).  Blocks beyond the CAP slots
// go to a global scratch area of the workgroup - slower, but any grid / length-scale stays correct.
// Row tiles of B = M' E[:, J] that no active block (Kt, I <= J) of sweep 2 reads are not computed at all.
template <int NB>
struct PairsCoop {
  static constexpr int TP = 16 * NB;
  static constexpr int CAP = (NB >= 12) ? 48 : (NB >= 8 ? 24 : 16);   // >= 2 NB: the factorisation reuses the E slots (row buffer, W)
  static constexpr size_t LDS_BYTES =
      sizeof(double) * ((size_t)CAP * 256 + NB * 256 /*rowbuf*/ + 256 /*Wbuf*/ + TP /*dvec*/ + 3 * TP + DIAG_SCR + 16) +
      sizeof(int) * (8 + 16 + 16 + 16 + 20 + NB * NB) + sizeof(double) * 4 * NB;
};

// The same pipeline with NB/2 waves per pair (CoopH<NB>, tile_f64.hpp: 8 waves and 17 tiles per wave at NB = 16 instead
// of 4 waves and 34-40 tiles): more than one wave per SIMD, so the latencies and barrier waits of one wave sit under the
// MFMAs of another.
template <int NB>
__global__ __launch_bounds__(64 * CoopH<NB>::NW, (NB <= 8) ? 2 : 1) void k_pairs_cooph(PairsArgs a) {
  using PC = PairsCoop<NB>;
  using H = CoopH<NB>;
  constexpr int NW = H::NW;
  constexpr int TP = 16 * NB, CAP = PC::CAP;
  constexpr int CH = (NB <= 8 && NB % 4 == 0) ? 4 : 2;   // row tiles of B[:, J] per pass (must divide NB; register budget)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ec = smem;                    // [CAP][4][64]
  double* rowbuf = Ec + CAP * 256;      // [NB][4][64]
  double* Wbuf = rowbuf + NB * 256;
  double* dvec = Wbuf + 256;            // d, then z = L^{-1} d
  double* xs = dvec + TP;
  double* ys = xs + TP;
  double* xbs = ys + TP;
  double* scr = xbs + TP;
  double* red = scr + DIAG_SCR;         // 16 doubles
  int* redi = reinterpret_cast<int*>(red + 16);   // 8
  int* amask = redi + 8;                // bit Kt of amask[J]: block (Kt, J) of E active
  int* kmask = amask + 16;              // bit I of kmask[J]: tile (I, J) of K** above the cut-off (I <= J)
  int* pneed = kmask + 16;              // OR of amask[0..J]: row tiles of B[:, J] that sweep 2 reads
  int* base = pneed + 16;               // first slot of column J (prefix sum of popcounts), base[NB] = total
  int* slotblk = base + 20;             // slot -> (Kt << 8) | J
  double* rng = reinterpret_cast<double*>(slotblk + NB * NB);   // [2 NB][lo, hi] of the real points of each 16-block
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, Ts = a.Ts;
  const int Kg = a.kend - a.kbeg;

  // block -> (segment, cluster).  Without `sel` the Kg clusters of the group are spread over the 8 XCDs (block b runs
  // on XCD b % 8) so that each XCD's L2 keeps the operators of Kg / 8 clusters only.
  int n, kc;
  if (a.sel) {
    n = blockIdx.x;
    kc = a.sel[n];
    bool mine = false;
    for (int kk = a.kbeg; kk < a.kend; ++kk) mine = mine || (a.perm[kk] == kc);
    if (!mine) return;
  } else {
    const int b = blockIdx.x;
    int kk;
    if ((Kg & 7) == 0) {
      const int cpx = Kg >> 3, s = b >> 3;
      kk = (b & 7) + 8 * (s % cpx);
      n = s / cpx;
    } else {
      kk = b % Kg;
      n = b / Kg;
    }
    kc = a.perm[a.kbeg + kk];
  }
  if (a.scal[8 * kc + 7] != 0.0) return;   // ill-conditioned K~: scored by the solve-based kernel (hgp_pairs_acc.hip)

  HGP_STAMP_DECL
  HGP_T0();
  for (int i = tid; i < TP; i += 64 * NW) {   // sentinel padding as in k_pairs
    xs[i] = (i < Ts) ? a.x[(size_t)n * Ts + i] / a.ell : 1e150 * (double)(1 + i);
    ys[i] = (i < Ts) ? a.y[(size_t)n * Ts + i] : 0.0;
    xbs[i] = (i < T) ? a.xb[i] / a.ell : -1e150 * (double)(1 + i);
  }
  if (tid < 16) {
    amask[tid] = 0;
    kmask[tid] = 0;
  }
  __syncthreads();
  // Which 16x16 blocks of E (and tiles of K**) can hold an entry above the cut-off?  Decided from the data: the
  // range [min, max] of the REAL points of every 16-block (padding excluded), then block (Kt, J) is active iff the
  // two ranges are closer than the cut-off radius.  Exact for sorted grids, a superset otherwise (never drops a block).
  for (int b16 = wave; b16 < 2 * NB; b16 += NW) {
    const int B = (b16 < NB) ? b16 : b16 - NB;
    const int i = 16 * B + c;
    const bool real = (b16 < NB) ? (i < T) : (i < Ts);
    const double v = (b16 < NB) ? xbs[i] : xs[i];
    double lo = real ? v : __builtin_inf(), hi = real ? v : -__builtin_inf();
    // all-reduce over the 16 lanes of a row by DPP rotations (was: four ds_bpermute round trips per value)
    lo = fmin(lo, dpp_f64<0x128>(lo));
    hi = fmax(hi, dpp_f64<0x128>(hi));
    lo = fmin(lo, dpp_f64<0x124>(lo));
    hi = fmax(hi, dpp_f64<0x124>(hi));
    lo = fmin(lo, dpp_f64<0x122>(lo));
    hi = fmax(hi, dpp_f64<0x122>(hi));
    lo = fmin(lo, dpp_f64<0x121>(lo));
    hi = fmax(hi, dpp_f64<0x121>(hi));
    if (lane == 0) {
      rng[2 * b16] = lo;
      rng[2 * b16 + 1] = hi;
    }
  }
  __syncthreads();
  bool actE = false;
  int myKt = 0, myJb = 0;
  if (tid < NB * NB) {
    myKt = tid / NB;
    myJb = tid % NB;
    const double xlo = rng[2 * (NB + myJb)], xhi = rng[2 * (NB + myJb) + 1];
    const double gE = fmax(0.0, fmax(rng[2 * myKt] - xhi, xlo - rng[2 * myKt + 1]));
    actE = 0.5 * (gE * gE) < PAIRS_CUT;
    if (actE) atomicOr(&amask[myJb], 1 << myKt);
    if (myKt <= myJb) {
      const double gK = fmax(0.0, fmax(rng[2 * (NB + myKt)] - xhi, xlo - rng[2 * (NB + myKt) + 1]));
      if (0.5 * (gK * gK) < PAIRS_CUT) atomicOr(&kmask[myJb], 1 << myKt);
    }
  }
  __syncthreads();
  if (tid <= NB) {   // thread J: first slot of column J and the rows of B[:, J] that sweep 2 reads (prefix over the columns before it)
    int s = 0, o = 0;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
      const int am = amask[J];
      if (J < tid) s += __popc(am);
      if (J <= tid) o |= am;
    }
    base[tid] = s;
    if (tid < NB) pneed[tid] = o;
  }
  __syncthreads();
  if (actE) slotblk[base[myJb] + __popc(amask[myJb] & ((1 << myKt) - 1))] = (myKt << 8) | myJb;
  const int nslot = __builtin_amdgcn_readfirstlane(base[NB]);
  // Dense grids: the blocks beyond the LDS slots go to a global scratch area of this workgroup.  Areas are handed out
  // with a compare-and-swap on a flag array that has more entries than workgroups can be resident at once.
  double* Eov = nullptr;
  int my_area = -1;
  if (nslot > CAP) {
    if (tid == 0) {
      int sidx = blockIdx.x % a.nscr;
      while (atomicCAS(&a.eflags[sidx], 0, 1) != 0) sidx = (sidx + 1 == a.nscr) ? 0 : sidx + 1;
      redi[7] = sidx;
    }
    __syncthreads();
    my_area = redi[7];
    Eov = a.escr + (size_t)my_area * a.escr_stride;
  }
  __syncthreads();
  for (int slot = wave; slot < nslot; slot += NW) {
    const int kj = __builtin_amdgcn_readfirstlane(slotblk[slot]);
    const int Kt = (kj >> 8) & 255, Jb = kj & 255;
    double* dst = (slot < CAP) ? Ec + slot * 256 : Eov + (size_t)(slot - CAP) * 256;
    double hk[4], ev[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double u = xbs[16 * Kt + 4 * s + g] - xs[16 * Jb + c];
      hk[s] = 0.5 * (u * u);
    }
    exp_neg4(hk, ev);
    int ks = 0;   // k-steps of the block with an entry above the cut-off (the others are exact zeros: the sweeps skip their MFMAs)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      dst[s * 64 + lane] = (hk[s] < PAIRS_CUT) ? ev[s] : 0.0;
      if (__any(hk[s] < PAIRS_CUT)) ks |= 1 << s;
    }
    if (lane == 0) slotblk[slot] = kj | (ks << 16);
  }
  __syncthreads();
  // E[16 Kt + 4 s + g][16 Jb + c] of the block in `slot`: the B operand of sweep 1 (k-step s) and the A operand of sweep 2
  // (the overflow read is a volatile global load on purpose: with two plain loads the compiler merges the branches
  //  into ONE flat_load through a selected generic pointer, which costs the LDS path its ds_read and its wait counter)
  auto e_op = [&](int slot, int s) -> double {
    if (slot < CAP) return Ec[slot * 256 + s * 64 + lane];
    return *reinterpret_cast<const volatile double*>(Eov + (size_t)(slot - CAP) * 256 + s * 64 + lane);
  };
  auto release_area = [&]() {
    if (my_area >= 0) {
      __syncthreads();
      if (tid == 0) {
        __threadfence();
        atomicExch(&a.eflags[my_area], 0);
      }
    }
  };

  HGP_ACC(0);
  const double* sc = a.scal + 8 * kc;
  const double cc = sc[0], noise = sc[2];
  const bool iso = sc[3] != 0.0;
  const size_t oidx = a.sel ? (size_t)n : (size_t)n * a.K + kc;
  const double fn = a.first_noise ? a.first_noise[oidx] : 0.0;

  // d = y - E^T a'  (block columns dealt to the waves)
  const double* apk = a.ap + (size_t)kc * TP;
  double dsq = 0.0;
  for (int Jb = wave; Jb < NB; Jb += NW) {
    int m = __builtin_amdgcn_readfirstlane(amask[Jb]);
    int slot = __builtin_amdgcn_readfirstlane(base[Jb]);
    double p = 0.0;
    while (m) {
      const int Kt = __builtin_ctz(m);
      m &= m - 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) p = fma(e_op(slot, s), apk[16 * Kt + 4 * s + g], p);
      ++slot;
    }
    p = xrow_sum(p);
    if (g == 0) {
      const int j = 16 * Jb + c;
      const double d = ys[j] - p;
      dvec[j] = d;
      dsq = fma(d, d, dsq);
    }
  }
  HGP_ACC(1);
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
    release_area();
    return;
  }

  int msk[NB], bas[NB];   // uniform copies for the statically indexed uses of sweep 2
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    msk[I] = __builtin_amdgcn_readfirstlane(amask[I]);
    bas[I] = __builtin_amdgcn_readfirstlane(base[I]);
  }
  const double* Mk = a.Mp + (size_t)kc * TP * TP;   // plain row-major here (no tile-pair interleave)
  d4 U[H::NT];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int J = (q == 0) ? wave : NB - 1 - wave;   // my block columns: A (rows 0..wave) and B (rows 0..NB-1-wave)
    const int mJ = __builtin_amdgcn_readfirstlane(amask[J]), bJ = __builtin_amdgcn_readfirstlane(base[J]);
    const int kmJ = __builtin_amdgcn_readfirstlane(kmask[J]), need = __builtin_amdgcn_readfirstlane(pneed[J]);
    // K** = c exp(-0.5 (x_i - x_j)^2) + noise I on my column (the one-argument kernel call, GPI.py:476)
#pragma unroll
    for (int I = 0; I < (q == 0 ? NW : NB); ++I) {
      const int ln = launder(lane);
      d4 kt = (d4){0.0, 0.0, 0.0, 0.0};
      if (I <= J) {
        if (kmJ & (1 << I)) {
          double hk[4], ev[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double u = xs[16 * I + (ln >> 4) + 4 * r] - xs[16 * J + (ln & 15)];
            hk[r] = 0.5 * (u * u);
          }
          exp_neg4(hk, ev);
#pragma unroll
          for (int r = 0; r < 4; ++r) kt[r] = (hk[r] < PAIRS_CUT) ? cc * ev[r] : 0.0;
        }
        if (I == J) {   // exact diagonal; identity on the padding
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((ln >> 4) + 4 * r == (ln & 15)) kt[r] = (16 * I + (ln & 15) < Ts) ? cc + noise : 1.0;
        }
      }
      if (I <= J) U[q == 0 ? H::slotA(I) : H::slotB(I)] = kt;   // (slots of rows I > J belong to the other column)
    }
    HGP_ACC(2);
#pragma nounroll
    for (int h = 0; h < NB / CH; ++h) {
      const int nb4 = (need >> (CH * h)) & ((1 << CH) - 1);
      if (!nb4) continue;
      // sweep 1: BJ[i] = M'[rows of tile 4h + i, :] E[:, J] over the active blocks of column J.  M' is symmetric:
      // the A operand of output row tile i, k-step s of block Kt is M'[16 Kt + 4 s + g][16 (4h + i) + c] (coalesced).
      d4 BJ[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) BJ[i] = (d4){0.0, 0.0, 0.0, 0.0};
      const double* Mh = Mk + (size_t)g * TP + 16 * CH * h + c;
      double ra[2][4][CH], re[2][4];
      int ksv[2] = {0, 0};   // live k-steps of the block in each buffer (uniform)
#define HGP_CFILL(buf, Kt_, slot_)                                                                  \
  ksv[buf] = __builtin_amdgcn_readfirstlane(slotblk[(slot_)]) >> 16;                                \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                \
    const double* row_ = Mh + (size_t)(16 * (Kt_) + 4 * s_) * TP;                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) ra[buf][s_][i_] = row_[16 * i_];               \
    re[buf][s_] = e_op((slot_), s_);                                                                 \
  }
#define HGP_CMMA(buf)                                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                \
    if (ksv[buf] & (1 << s_)) {                                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_)                                             \
        if (nb4 & (1 << i_)) BJ[i_] = mfma(ra[buf][s_][i_], re[buf][s_], BJ[i_]);                   \
    }                                                                                               \
  }
      int m = mJ, slot = bJ;
      int kA = -1, kB = -1;
      if (m) {
        kA = __builtin_ctz(m);
        m &= m - 1;
        HGP_CFILL(0, kA, slot)
        ++slot;
      }
#pragma nounroll
      while (kA >= 0) {
        kB = -1;
        if (m) {
          kB = __builtin_ctz(m);
          m &= m - 1;
          HGP_CFILL(1, kB, slot)
          ++slot;
        }
        HGP_CMMA(0)
        if (kB < 0) break;
        kA = -1;
        if (m) {
          kA = __builtin_ctz(m);
          m &= m - 1;
          HGP_CFILL(0, kA, slot)
          ++slot;
        }
        HGP_CMMA(1)
      }
      HGP_ACC(3);
      // sweep 2: U[I][J] += E[Kt, I]^T BJ[i] over the active blocks (Kt = 4h + i, I <= J) of E.  Per tile the operands
      // of all its active blocks in this chunk are requested first, then multiplied: one LDS latency per tile.
#pragma unroll
      for (int I = 0; I < (q == 0 ? NW : NB); ++I) {
        const int m4 = (msk[I] >> (CH * h)) & nb4;
        if (I <= J && m4) {
          const int below = __popc(msk[I] & ((1 << (CH * h)) - 1));
          double af[CH][4];
          int ks2[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            ks2[i] = 0;
            if (m4 & (1 << i)) {
              const int slot2 = bas[I] + below + __popc(m4 & ((1 << i) - 1));
              ks2[i] = slotblk[slot2];
#pragma unroll
              for (int s = 0; s < 4; ++s) af[i][s] = e_op(slot2, s);
            }
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            if (m4 & (1 << i)) {
              const int ks = __builtin_amdgcn_readfirstlane(ks2[i]) >> 16;
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (ks & (1 << s)) U[q == 0 ? H::slotA(I) : H::slotB(I)] = mfma(af[i][s], BJ[i][s], U[q == 0 ? H::slotA(I) : H::slotB(I)]);
            }
          }
        }
      }
      HGP_ACC(4);
    }
  }
#undef HGP_CFILL
#undef HGP_CMMA

  // regularisation of the reference: +1e-6 I (GPI.py:501), + first, + 1e-8 mean|diag| I (GPI_model.py:83-87)
  {
    const double sh = 1e-6 + fn;
    const double dm = cooph_diag_abs_mean<NB>(U, Ts, wave, lane, sh, red);   // (also orders the dvec writes: barrier)
    cooph_add_diag<NB>(U, sh + 1e-8 * fmax(dm, F64_EPS), Ts, wave, lane);
  }
  PivotAcc pa;
  pa.init();
  HGP_ACC(5);
#ifdef HGP_COOPH_BARRIERS   // the barrier-synchronised factorisation of rounds 1-2 (A/B builds only)
  double zq = cooph_factor<NB>(U, rowbuf, Wbuf, scr, wave, lane, pa, Ts, dvec);
#else
  // dataflow-synchronised (tile_f64.hpp, cooph_factor_df): the E slots are dead now - second row buffer and the W of every step
  static_assert(CAP >= 3 * NB, "E slots too small for the factorisation's buffers");
  double zq = cooph_factor_df<NB>(U, rowbuf, Ec, Ec + NB * 256, Ec + 2 * NB * 256, scr, slotblk, wave, lane, pa, Ts, dvec);
#endif
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
  release_area();
  HGP_ACC(6);
#ifdef HGP_STAMPS
  if (tid == 64 * (WAVES - 1) && a.stamps) {   // the view of the last wave
    for (int i = 0; i < 8; ++i) atomicAdd(&a.stamps[i], hgp_acc_[i]);
    for (int i = 0; i < 5; ++i) atomicAdd(&a.stamps[8 + i], pa.cf[i]);
  }
#endif
}

template <int NB>
int launch_pairs_cooph(const PairsArgs& a, hipStream_t st) {
  const size_t lds = PairsCoop<NB>::LDS_BYTES;
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_pairs_cooph<NB>), lds)) return rc_;
  const int blocks = a.sel ? a.N : a.N * (a.kend - a.kbeg);
  hipLaunchKernelGGL(k_pairs_cooph<NB>, dim3(blocks), dim3(64 * CoopH<NB>::NW), lds, st, a);
  return launch_status();
}

template <int NB>
int launch_pairs(const PairsArgs& a0, hipStream_t st) {
  size_t lds = pairs_lds_bytes<NB>();
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_pairs<NB, false>), lds)) return rc_;
  if (NB < 6 || (a0.flags & 1) || !a0.fb) {   // mask-driven sweeps for every segment
    PairsArgs a = a0;
    a.fb = nullptr;
    hipLaunchKernelGGL((k_pairs<NB, false>), dim3(a.N), dim3(64 * WAVES), lds, st, a);
    return launch_status();
  }
  constexpr int PWB = pairs_waves<NB, true>();
  const size_t ldsb = pairs_lds_bytes<NB, PWB>();
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_pairs<NB, true>), ldsb)) return rc_;
  for (int off = 0; off < a0.N; off += PAIRS_FB_CAP) {   // the fall-back list holds PAIRS_FB_CAP segments
    PairsArgs a = a0;
    a.N = std::min(PAIRS_FB_CAP, a0.N - off);
    const size_t oo = a.sel ? (size_t)off : (size_t)off * a.K;
    a.x += (size_t)off * a.Ts;
    a.y += (size_t)off * a.Ts;
    if (a.first_noise) a.first_noise += oo;
    if (a.sel) a.sel += off;
    a.out_quad += oo;
    if (a.out_logdet) a.out_logdet += oo;
    if (a.out_info) a.out_info += oo;
    hipLaunchKernelGGL((k_pairs<NB, true>), dim3(a.N), dim3(64 * PWB), ldsb, st, a);
    const int rc1 = launch_status();
    hipLaunchKernelGGL((k_pairs<NB, false>), dim3(a.N), dim3(64 * WAVES), lds, st, a);
    const int rc2 = launch_status();
    if (rc1 || rc2) {   // the generic kernel hands the list back empty; if either launch failed, do it here
      (void)hipMemsetAsync(a.fb, 0, sizeof(int32_t), st);
      (void)hipMemsetAsync(a.fb + 1 + PAIRS_FB_CAP, 0, sizeof(int32_t), st);
      return rc1 ? rc1 : rc2;
    }
  }
  return 0;
}

}  // namespace

// dispatch by padded size / kernel family (see hgp_loglik_pairs_f64)
int hgp_internal_pairs_fast(const PairsArgs& a, int NB, bool coop, hipStream_t st) {
  if (coop) {   // NB/2 waves per pair (CoopH)
    switch (NB) {
      case 8: return launch_pairs_cooph<8>(a, st);
      case 12: return launch_pairs_cooph<12>(a, st);
      default: return launch_pairs_cooph<16>(a, st);
    }
  }
  switch (NB) {
    case 2: return launch_pairs<2>(a, st);
    case 4: return launch_pairs<4>(a, st);
    case 6: return launch_pairs<6>(a, st);
    default: return launch_pairs<8>(a, st);
  }
}
