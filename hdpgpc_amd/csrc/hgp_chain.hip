// SURVEY.md 8f-1, the per-member step of the LDS recursion with ONE launch per dependency level.
//
// The step (Kalman update, pair smoother, two MNIW updates; GPI.py:72-151,272-300, GPI_model.py:1300-1344) is ~30 small dense
// products at T = 90.  Measured (rocprofv3, record 100): every dependent launch costs ~4.5 us whatever it does - a 90^3
// product adds ~2 us of work on top - so the step's ~40 launches (one per product, plus element-wise glue) WERE its 0.28 ms.
// Here every dependency level is one launch of k_gemm_list: a device-resident list of heterogeneous items
//     C = alpha op(A) op(B) + beta D        (any M x N x K <= 256; vectors are N = 1; D may alias nothing or be a vector)
// with all pointers fixed for the life of the chain (the state rows of the step are gathered into one workspace first), and
// the two Cholesky inversions of the step carry their right-hand sides along (k_wave_inv_rhs: Z = L^-1 and Y = L^-1 op(B) from
// one factorisation), which removes a product level behind each of them.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

namespace {

// -------------------------------------------------------------------------------------------- list GEMM
// tile_map (may be NULL): tile_map[t] = (item << 16) | tile-within-item for the t-th output tile of the list.  Without it a
// wave finds its item by walking the list - one dependent global load per item, fine for the 1-5 items of one chain's level,
// 25 us for the 50 items of ten chains side by side.
__global__ __launch_bounds__(64 * WAVES) void k_gemm_list(const hgp_gemm_item* __restrict__ items, int n_items,
                                                          const uint32_t* __restrict__ tile_map, int total_tiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  int tile = blockIdx.x * WAVES + wave;
  int it = 0;
  if (tile_map) {
    if (tile >= total_tiles) return;
    const uint32_t e = tile_map[tile];
    it = (int)(e >> 16);
    tile = (int)(e & 0xffffu);
  } else {
    for (; it < n_items; ++it) {           // which item does this wave's tile belong to?
      const int nt = ((items[it].M + 15) >> 4) * ((items[it].N + 15) >> 4);
      if (tile < nt) break;
      tile -= nt;
    }
    if (it >= n_items) return;
  }
  const hgp_gemm_item q = items[it];
  const int ntn = (q.N + 15) >> 4;
  const int ti = tile / ntn, tj = tile % ntn;
  const int row = 16 * ti + c, col = 16 * tj + c;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  double dv[4] = {0.0, 0.0, 0.0, 0.0};   // the addend travels with the operands (behind the products it was one more round trip)
  if (q.D) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ti + g + 4 * r;
      if (i < q.M && col < q.N) dv[r] = q.D[(size_t)i * q.ldd + col];
    }
  }
  const int nk = (q.K + 3) >> 2;
  constexpr int TRIP = 24;               // K <= 96 in one trip: all operand loads are in flight before the first MFMA
  for (int k0 = 0; k0 < nk; k0 += TRIP) {
    double av[TRIP], bv[TRIP];
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      const int k = 4 * (k0 + u) + g;
      av[u] = 0.0;
      bv[u] = 0.0;
      if (k < q.K) {
        if (row < q.M) av[u] = q.tA ? q.A[(size_t)k * q.lda + row] : q.A[(size_t)row * q.lda + k];
        if (col < q.N) bv[u] = q.tB ? q.B[(size_t)col * q.ldb + k] : q.B[(size_t)k * q.ldb + col];
      }
    }
#pragma unroll
    for (int u = 0; u < TRIP; ++u) acc = mfma(av[u], bv[u], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 16 * ti + g + 4 * r;
    if (i < q.M && col < q.N) {
      double v = q.alpha * acc[r];
      if (q.D) v += q.beta * dv[r];
      if (q.add_eye && i == col) v += q.add_eye;
      q.C[(size_t)i * q.ldc + col] = v;
      if (q.C2) q.C2[(size_t)i * q.ldc + col] = v;
    }
  }
}

// -------------------------------------------------------------------------------------------- inverse with riding RHS
struct InvRhsArgs {
  const double* A;       // [b,T,T]
  int T, b;
  double jitter_rel, add;
  double* Linv;          // [b,T,T]  Z = L^-1 (may be NULL: only the solves are wanted)
  const double* rhs;     // [b,T,T] or NULL
  const int32_t* rhs_on; // [b] or NULL: item m carries a right-hand side iff rhs_on[m] != 0 (NULL: all do when rhs != NULL)
  int rhs_trans;         // solve against rhs^T
  double* rhs_out;       // [b,T,T]  Y = L^-1 op(rhs)
  int32_t* info;
};

template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_wave_inv_rhs(InvRhsArgs a) {
  __shared__ __attribute__((aligned(16))) double scr_all[WAVES * DIAG_SCR];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int w = blockIdx.x * WAVES + wave;
  const int m = w / (2 * NB), Jq = w % (2 * NB);
  const int T = a.T;
  if (m >= a.b) return;
  const bool is_rhs = Jq >= NB;
  const int Jc = is_rhs ? Jq - NB : Jq;
  if (16 * Jc >= T) return;
  const bool skip = (is_rhs && (!a.rhs || (a.rhs_on && a.rhs_on[m] == 0))) || (!is_rhs && !a.Linv);
  if (skip) {     // an item nobody factors (no Linv wanted, right-hand side switched off) still reports a defined status
    if (Jq == 0 && !a.Linv && lane == 0 && a.info && (!a.rhs || (a.rhs_on && a.rhs_on[m] == 0))) a.info[m] = 0;
    return;
  }
  double* scr = scr_all + wave * DIAG_SCR;
  const double* A = a.A + (size_t)m * T * T;
  d4 U[NB * (NB + 1) / 2];
  d4 R[NB];
  if constexpr (NB <= 6) load_sym_upper_burst<NB>(U, A, T, T, lane, scr);   // one exposed load latency instead of NB
  else load_sym_upper<NB>(U, A, T, T, lane, scr);
  {
    double sh = a.add;
    if (a.jitter_rel != 0.0) sh += a.jitter_rel * fmax(diag_abs_mean<NB>(U, T, lane, a.add), F64_EPS);
    if (sh != 0.0) add_diag<NB>(U, sh, T, lane);
  }
  if (!is_rhs) {
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int r = 0; r < 4; ++r) R[K][r] = (K == Jc && g + 4 * r == c) ? 1.0 : 0.0;
  } else {
    const double* B = a.rhs + (size_t)m * T * T;
#pragma unroll
    for (int K = 0; K < NB; ++K)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
        R[K][r] = (i < T && j < T) ? (a.rhs_trans ? B[(size_t)j * T + i] : B[(size_t)i * T + j]) : 0.0;
      }
  }
  PivotAcc pa;
  pa.init();
  wave_factor<NB, 1>(U, R, scr, nullptr, nullptr, lane, pa, nullptr, 0, T);
  double* Z = (is_rhs ? a.rhs_out : a.Linv) + (size_t)m * T * T;
#pragma unroll
  for (int K = 0; K < NB; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (i < T && j < T) Z[(size_t)i * T + j] = (is_rhs || K >= Jc) ? R[K][r] : 0.0;
    }
  if (Jq == (a.Linv ? 0 : NB) && lane == 0 && a.info) a.info[m] = pa.info;   // the first panel sees every pivot
}

template <int NB>
void launch_inv_rhs(const InvRhsArgs& a, hipStream_t st) {
  const int waves = a.b * 2 * NB;
  hipLaunchKernelGGL(k_wave_inv_rhs<NB>, dim3((waves + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, st, a);
}

// The same job for SMALL batches (the member step of a few chains: 4 or 2 matrices per chain), where the launch is one exposed
// latency and a wave that factors the whole matrix by itself is the whole of it (31 us at T = 90: six serial diag16 plus ~280
// dependent MFMAs on one SIMD).  Here a WORKGROUP of four waves takes each (matrix, panel): the tiles are dealt over the waves
// (Coop<NB>), only the pivot chain stays serial.  NB is padded to a multiple of four (identity blocks); the padded steps are not
// taken (kstop).  Arithmetic per tile is that of wave_factor (same operations in the same order): results identical bit for bit.
template <int NB>
__global__ __launch_bounds__(64 * WAVES) void k_coop_inv_rhs(InvRhsArgs a) {
  using C = Coop<NB>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* rowbuf = smem;
  double* Rbuf = rowbuf + NB * 256;
  double* Wbuf = Rbuf + NB * 256;
  double* scr = Wbuf + 256;
  double* red = scr + DIAG_SCR;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int T = a.T, nblk = (T + 15) >> 4;
  const int m = blockIdx.x, Jq = blockIdx.y;            // gridDim.y = 2 nblk: panels of L^-1, then panels of L^-1 op(B)
  const bool is_rhs = Jq >= nblk;
  const int Jc = is_rhs ? Jq - nblk : Jq;
  const bool skip = (is_rhs && (!a.rhs || (a.rhs_on && a.rhs_on[m] == 0))) || (!is_rhs && !a.Linv);
  if (skip) {
    if (Jq == 0 && !a.Linv && threadIdx.x == 0 && a.info && (!a.rhs || (a.rhs_on && a.rhs_on[m] == 0))) a.info[m] = 0;
    return;
  }
  const double* A = a.A + (size_t)m * T * T;
  d4 U[C::NT];
  coop_load_sym_upper<NB>(U, A, T, T, wave, lane, rowbuf + wave * DIAG_SCR);
  __syncthreads();   // rowbuf served as per-wave staging for the loader
  {
    double sh = a.add;
    if (a.jitter_rel != 0.0) sh += a.jitter_rel * fmax(coop_diag_abs_mean<NB>(U, T, wave, lane, a.add, red), F64_EPS);
    if (sh != 0.0) coop_add_diag<NB>(U, sh, T, wave, lane);
  }
  const double* B = is_rhs ? a.rhs + (size_t)m * T * T : nullptr;
  for (int K = wave; K < NB; K += WAVES) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (!is_rhs) v[r] = (K == Jc && g + 4 * r == c) ? 1.0 : 0.0;
      else v[r] = (i < T && j < T) ? (a.rhs_trans ? B[(size_t)j * T + i] : B[(size_t)i * T + j]) : 0.0;
    }
    lds_tile_store(Rbuf, K, lane, v);
  }
  __syncthreads();
  PivotAcc pa;
  pa.init();
  coop_factor<NB, true>(U, rowbuf, Rbuf, Wbuf, scr, wave, lane, pa, nullptr, 0, T, nullptr, nullptr, 0, nullptr, nullptr, nblk);
  double* Z = (is_rhs ? a.rhs_out : a.Linv) + (size_t)m * T * T;
  for (int K = wave; K < nblk; K += WAVES) {
    const d4 z = lds_tile_load(Rbuf, K, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (i < T && j < T) Z[(size_t)i * T + j] = (is_rhs || K >= Jc) ? z[r] : 0.0;
    }
  }
  if (Jq == (a.Linv ? 0 : nblk) && a.info) {   // the first panel sees every pivot
    int info;
    (void)coop_logdet_info(pa, wave, lane, red, reinterpret_cast<int*>(red + 8), info);
    if (threadIdx.x == 0) a.info[m] = info;
  }
}

// ... and with the DATAFLOW factorisation (cooph_factor_df<NB, 1>, NB / 2 waves per workgroup, NB = the even number of 16-blocks:
// no padding at T = 90): no workgroup barrier inside the factorisation - the wave that owns block K + 1 solves only that panel tile,
// updates its diagonal tile from the accumulator, factors it and publishes W_{K+1} and Z_{K+1} while the others are still in the
// trailing update of step K.  The barrier version above spends 4.5 us per block step for 1.2 us of diag16_acc.
template <int NB>
__global__ __launch_bounds__(64 * CoopH<NB>::NW) void k_cooph_inv_rhs(InvRhsArgs a) {
  using C = CoopH<NB>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* row0 = smem;                       // three row buffers, W of every block, Z of every block: [NB] tiles each
  double* Wall = row0 + 3 * NB * 256;
  double* zbuf = Wall + NB * 256;
  double* scr_all = zbuf + NB * 256;         // [NW][DIAG_SCR]
  double* red = scr_all + C::NW * DIAG_SCR;  // [8]
  int* flags = reinterpret_cast<int*>(red + 8);   // [16 + NB] + [8] status words
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  const int T = a.T, nblk = (T + 15) >> 4;
  const int m = blockIdx.x, Jq = blockIdx.y;            // gridDim.y = 2 nblk: panels of L^-1, then panels of L^-1 op(B)
  const bool is_rhs = Jq >= nblk;
  const int Jc = is_rhs ? Jq - nblk : Jq;
  const bool skip = (is_rhs && (!a.rhs || (a.rhs_on && a.rhs_on[m] == 0))) || (!is_rhs && !a.Linv);
  if (skip) {
    if (Jq == 0 && !a.Linv && threadIdx.x == 0 && a.info && (!a.rhs || (a.rhs_on && a.rhs_on[m] == 0))) a.info[m] = 0;
    return;
  }
  double* scr = scr_all + wave * DIAG_SCR;
  const double* A = a.A + (size_t)m * T * T;
  d4 U[C::NT];
#pragma unroll
  for (int i_ = 0; i_ < C::NT; ++i_) U[i_] = (d4){0.0, 0.0, 0.0, 0.0};
  cooph_load_sym_upper<NB>(U, A, T, T, wave, lane, scr);
  {
    double sh = a.add;
    if (a.jitter_rel != 0.0) sh += a.jitter_rel * fmax(cooph_diag_abs_mean<NB>(U, T, wave, lane, a.add, red), F64_EPS);
    if (sh != 0.0) cooph_add_diag<NB>(U, sh, T, wave, lane);
  }
  // my two tiles of the right-hand side panel: block rows JA = wave and JB = NB - 1 - wave
  const double* B = is_rhs ? a.rhs + (size_t)m * T * T : nullptr;
  d4 RA, RB;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int K = h == 0 ? wave : NB - 1 - wave;
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (!is_rhs) v[r] = (K == Jc && g + 4 * r == c) ? 1.0 : 0.0;
      else v[r] = (i < T && j < T) ? (a.rhs_trans ? B[(size_t)j * T + i] : B[(size_t)i * T + j]) : 0.0;
    }
    if (h == 0) RA = v;
    else RB = v;
  }
  PivotAcc pa;
  pa.init();
  cooph_factor_df<NB, 1>(U, row0, row0 + NB * 256, row0 + 2 * NB * 256, Wall, scr, flags, wave, lane, pa, T, nullptr, &RA, &RB, zbuf);
  double* Z = (is_rhs ? a.rhs_out : a.Linv) + (size_t)m * T * T;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int K = h == 0 ? wave : NB - 1 - wave;
    const d4 z = h == 0 ? RA : RB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * K + g + 4 * r, j = 16 * Jc + c;
      if (i < T && j < T) Z[(size_t)i * T + j] = (is_rhs || K >= Jc) ? z[r] : 0.0;
    }
  }
  if (Jq == (a.Linv ? 0 : nblk) && a.info) {   // the first panel sees every pivot: the earliest bad one over the waves
    int* redi = flags + 16 + NB;
    if (lane == 0) redi[wave] = pa.info;
    __syncthreads();
    if (threadIdx.x == 0) {
      int inf = 0;
      for (int w = 0; w < C::NW; ++w)
        if (redi[w] != 0 && (inf == 0 || redi[w] < inf)) inf = redi[w];
      a.info[m] = inf;
    }
  }
}

template <int NB>
int launch_cooph_inv_rhs(const InvRhsArgs& a, hipStream_t st) {
  const size_t lds = sizeof(double) * ((size_t)5 * NB * 256 + CoopH<NB>::NW * DIAG_SCR + 8) + sizeof(int) * (16 + NB + 8);
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_cooph_inv_rhs<NB>), lds)) return rc_;
  hipLaunchKernelGGL(k_cooph_inv_rhs<NB>, dim3(a.b, 2 * ((a.T + 15) >> 4)), dim3(64 * CoopH<NB>::NW), lds, st, a);
  return launch_status();
}

template <int NB>
int launch_coop_inv_rhs(const InvRhsArgs& a, hipStream_t st) {
  const size_t lds = sizeof(double) * Coop<NB>::LDS_DOUBLES;
  if (int rc_ = hgp_internal_ensure_dynamic_lds(reinterpret_cast<const void*>(&k_coop_inv_rhs<NB>), lds)) return rc_;
  hipLaunchKernelGGL(k_coop_inv_rhs<NB>, dim3(a.b, 2 * ((a.T + 15) >> 4)), dim3(64 * WAVES), lds, st, a);
  return launch_status();
}

// -------------------------------------------------------------------------------------------- fused glue of the step
// gather (rows `pos` of the stacks -> workspace, the member's observation) + the jittered right covariances of the two MNIW
// updates  R' = R + 1e-2 max(mean |diag scale|, eps) I  (GPI_model.py:1312-1316), one launch.
using Gather2Args = hgp_chain_gather_desc;   // include/hdpgpc_hip.h

__device__ __forceinline__ void chain_gather2_body(const Gather2Args& a, double (&red)[2][4]) {
  const long tt = (long)a.T * a.T, p = a.pos[0];
  const long total = 6 * tt + 2 * a.T;
  // mean |diag scale| of both MNIW distributions, by every block (2 T strided loads, one LDS reduction)
  {
    double s0 = 0.0, s1 = 0.0;
    for (int d = threadIdx.x; d < a.T; d += 256) {
      s0 += fabs(a.W[4 * tt + (size_t)d * a.T + d]);
      s1 += fabs(a.W[5 * tt + (size_t)d * a.T + d]);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = s0;
      red[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total + 2 * tt; i += (long)gridDim.x * 256) {
    if (i < total) {
      if (i < 6 * tt) {
        const int k = (int)(i / tt);
        a.out[i] = a.st[k][p * tt + (i - k * tt)];
      } else {
        const long j = i - 6 * tt;
        const int k = 6 + (int)(j / a.T);
        a.out[i] = a.st[k][p * a.T + (j % a.T)];
      }
      if (a.Y && i < a.T) a.y_out[i] = a.Y[(a.y_row0 < 0 ? 0 : p - a.y_row0) * a.T + i];   // y_row0 < 0: Y is the observation itself
    } else {
      const long e = i - total;                 // element of R' [2,T,T]
      const int mtx = (int)(e / tt);
      const long ij = e % tt;
      double v = a.W[2 * tt + e];               // R
      if (ij / a.T == ij % a.T) {
        const double s = ((red[mtx][0] + red[mtx][1]) + red[mtx][2]) + red[mtx][3];
        v += 1e-2 * fmax(s / a.T, F64_EPS);
      }
      a.Rp[e] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_chain_gather2(Gather2Args a) {
  __shared__ double red[2][4];
  chain_gather2_body(a, red);
}

// the same for a BATCH of independent chains: blockIdx.y = chain, one descriptor per chain in device memory
__global__ __launch_bounds__(256) void k_chain_gather2_b(const Gather2Args* __restrict__ descs) {
  __shared__ double red[2][4];
  const Gather2Args a = descs[blockIdx.y];
  chain_gather2_body(a, red);
}

// scatter + finish: append the new filtered state, overwrite the re-smoothed previous one (GPI_model.py:317,705-716), the
// element-wise tail of both MNIW updates with (y1 - y2)(y1 - y2)^T formed here, the annealed scales, the append of
// A, Gamma, C, Sigma and the counters (GPI_model.py:1068-1106,1326-1336).
using Finish2Args = hgp_chain_finish_desc;   // include/hdpgpc_hip.h

#pragma clang fp contract(off)   // the reference's op order, no fused multiply-adds
__device__ __forceinline__ void chain_finish2_body(const Finish2Args& a, int& last) {
  const int T = a.T;
  const long tt = (long)T * T;
  const bool bad = (a.info1[2] | a.info1[3] | a.info2[0] | a.info2[1]) != 0;
  const double n0 = a.n0[0], Nf = a.Nf[0] + 1.0;
  const long p = a.pos[0], nxt = p + 1;
  const double n0n = bad ? n0 : n0 + 1.0;
  const double scl = n0n / (n0n - 2.0);
  const double ann = (a.annealing & 1) ? 1.0 / (Nf * Nf) : 0.0;
  // online path (hdpgpc_amd/online_chain.py): bit 1 = dry run - a CANDIDATE step: the new rows are written behind the chain's
  // end (row pos + 1) but the distributions W, the counters and the position stay as they are; bit 2 = the previous smoothed
  // state is not rewritten (a step without backwards_pair: the committed online step, GPI_HDP.py:2186-2196)
  const bool dry = (a.annealing & 2) != 0, keep_prev = (a.annealing & 4) != 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2 * tt; i += (long)gridDim.x * 256) {
    const bool obs = i >= tt;           // item 0 = internal (A, Gamma): (f_post, f_sm_prev); item 1 = observation (C, Sigma): (y, f_post)
    const long e = obs ? i - tt : i;
    const int r_ = (int)(e / T), c_ = (int)(e % T);
    double m = a.W[i], r = a.W[2 * tt + i], sc = a.W[4 * tt + i];
    if (!bad) {
      const double er = obs ? a.y[r_] - a.f_post[r_] : a.f_post[r_] - a.f_sm_prev[r_];
      const double ec = obs ? a.y[c_] - a.f_post[c_] : a.f_post[c_] - a.f_sm_prev[c_];
      m = ((n0 - 2.0) * m + a.part[i]) / (n0 - 1.0);
      r = a.Snew[i];
      sc = ((n0 - 2.0) * sc + er * ec) / (n0 - 1.0);
      if (!dry) {
        a.W[i] = m;
        a.W[2 * tt + i] = r;
        a.W[4 * tt + i] = sc;
      }
    }
    (obs ? a.stC : a.stA)[nxt * tt + e] = m;
    double* sg = obs ? a.stS : a.stG;
    sg[nxt * tt + e] = sc * scl + sg[e] * ann;
    if (!obs) {                          // the state lists (scatter)
      const double cp = a.c_post[e];
      a.stP[nxt * tt + e] = cp;
      a.stPsm[nxt * tt + e] = cp;
      if (!keep_prev) a.stPsm[p * tt + e] = a.P_sm_prev[e];
      if (e < T) {
        const double f = a.f_post[e];
        a.stF[nxt * T + e] = f;
        a.stFsm[nxt * T + e] = f;
        if (!keep_prev) a.stFsm[p * T + e] = a.f_sm_prev[e];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = (atomicAdd(a.sync, 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    if (!dry) {
      a.n0[0] = n0n;
      a.Nf[0] = Nf;
      a.pos[0] = nxt;
    }
    a.bad_count[0] += bad ? 1 : 0;
    if (a.bad_count[1] == 0 && (a.info1[0] | a.info1[1]) != 0) a.bad_count[1] = (int32_t)nxt;
    a.sync[0] = 0;
  }
}

__global__ __launch_bounds__(256) void k_chain_finish2(Finish2Args a) {
  __shared__ int last;
  chain_finish2_body(a, last);
}

__global__ __launch_bounds__(256) void k_chain_finish2_b(const Finish2Args* __restrict__ descs) {
  __shared__ int last;
  const Finish2Args a = descs[blockIdx.y];
  chain_finish2_body(a, last);
}
#pragma clang fp contract(on)

// -------------------------------------------------------------------------------------------- list of copies
// The online step assembles the inputs of ONE batched a8 / a9 launch from rows of many clusters' stacks (three members per
// candidate, two parameter pairs): dozens of T- and T^2-sized copies as one launch instead of one hipMemcpyAsync each.
__global__ __launch_bounds__(256) void k_copy_list(const hgp_copy_item* __restrict__ items) {
  const hgp_copy_item q = items[blockIdx.y];
  const long n2 = q.n >> 1;                                   // pairs of doubles (src / dst 16-byte aligned when n is even)
  const bool vec = ((reinterpret_cast<uintptr_t>(q.src) | reinterpret_cast<uintptr_t>(q.dst)) & 15) == 0;
  if (vec) {
    const double2* s2 = reinterpret_cast<const double2*>(q.src);
    double2* d2 = reinterpret_cast<double2*>(q.dst);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) d2[i] = s2[i];
    if ((q.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) q.dst[q.n - 1] = q.src[q.n - 1];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < q.n; i += (long)gridDim.x * 256) q.dst[i] = q.src[i];
  }
}

}  // namespace

extern "C" {

int hgp_copy_list_f64(const hgp_copy_item* items_dev, int n_items, long max_n, void* stream) {
  if (n_items == 0) return 0;
  if (!items_dev || n_items < 0 || n_items > 65535 || max_n <= 0) return -1;
  const long blocks = std::min<long>(64, (max_n / 2 + 255) / 256 + 1);
  hipLaunchKernelGGL(k_copy_list, dim3((unsigned)blocks, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
  return launch_status();
}

int hgp_gemm_list_f64(const hgp_gemm_item* items_dev, int n_items, int total_tiles, void* stream) {
  if (n_items == 0) return 0;
  if (!items_dev || n_items < 0 || total_tiles <= 0) return -1;
  hipLaunchKernelGGL(k_gemm_list, dim3((total_tiles + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, (hipStream_t)stream, items_dev, n_items,
                     (const uint32_t*)nullptr, total_tiles);
  return launch_status();
}

int hgp_gemm_list_mapped_f64(const hgp_gemm_item* items_dev, int n_items, const uint32_t* tile_map_dev, int total_tiles, void* stream) {
  if (n_items == 0) return 0;
  if (!items_dev || !tile_map_dev || n_items < 0 || n_items > 65535 || total_tiles <= 0) return -1;
  hipLaunchKernelGGL(k_gemm_list, dim3((total_tiles + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, (hipStream_t)stream, items_dev, n_items,
                     tile_map_dev, total_tiles);
  return launch_status();
}

int hgp_chol_inverse_rhs_batched_f64(const double* A, int T, int b, double jitter_rel, double add_diag, double* Linv, const double* rhs,
                                     const int32_t* rhs_on, int rhs_trans, double* rhs_out, int32_t* info, void* stream) {
  if (b == 0) return 0;
  if (!A || T <= 0 || b < 0 || (!Linv && !rhs) || (rhs && !rhs_out)) return -1;
  if (T > HGP_MAX_T_WAVE) return -2;
  InvRhsArgs a{A, T, b, jitter_rel, add_diag, Linv, rhs, rhs_on, rhs_trans, rhs_out, info};
  hipStream_t st = (hipStream_t)stream;
  // few matrices (the member step of a few chains): one workgroup per (matrix, panel) - latency; many: one wave each - throughput
  // (tools/time_inv_rhs.py; while the workgroups fit the chip in one round - beyond that the one-wave kernel wins.
  // HGP_INV_COOP_MAX_WG overrides the round size, 0 = never.)
  static const int coop_max_wg = getenv("HGP_INV_COOP_MAX_WG") ? atoi(getenv("HGP_INV_COOP_MAX_WG")) : 256;
  // T > 64: the dataflow form (T = 90: 21.7 us, barrier form 27.1, one wave per panel 30.3; T = 128: 28.0 / 30.6 / 51.5; two
  // workgroups per CU at NB = 6); 32 < T <= 64: the barrier form (T = 50: 13.1 us against 14.7 / 16.5); T <= 32 is one wave anyway
  const long wgs = (long)b * 2 * ((T + 15) >> 4);
  if (T > 64 && wgs <= (nb_for(T) == 6 ? 2 : 1) * (long)coop_max_wg) {
    static const bool barrier_form = env_on("HGP_INV_COOP_BARRIER");
    if (barrier_form && wgs <= coop_max_wg) return launch_coop_inv_rhs<8>(a, st);
    if (!barrier_form) return nb_for(T) == 6 ? launch_cooph_inv_rhs<6>(a, st) : launch_cooph_inv_rhs<8>(a, st);
  } else if (T > 32 && T <= 64 && wgs <= coop_max_wg) {
    return launch_coop_inv_rhs<4>(a, st);
  }
  switch (nb_for(T)) {
    case 2: launch_inv_rhs<2>(a, st); break;
    case 4: launch_inv_rhs<4>(a, st); break;
    case 6: launch_inv_rhs<6>(a, st); break;
    default: launch_inv_rhs<8>(a, st); break;
  }
  return launch_status();
}

int hgp_lds_chain_gather2_f64(const double* stA, const double* stG, const double* stC, const double* stS, const double* stPsm,
                              const double* stP, const double* stF, const double* stFsm, const int64_t* pos, int T, double* out,
                              const double* Y, long y_row0, double* y_out, const double* W, double* Rp, void* stream) {
  if (!stA || !stG || !stC || !stS || !stP || !stPsm || !stF || !stFsm || !pos || !out || !W || !Rp || T <= 0 || (Y && !y_out)) return -1;
  Gather2Args a{{stA, stG, stC, stS, stPsm, stP, stF, stFsm}, pos, out, Y, y_out, W, Rp, y_row0, T};
  const long total = 8L * T * T + 2L * T;
  hipLaunchKernelGGL(k_chain_gather2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_lds_chain_finish2_f64(int T, const double* f_post, const double* c_post, const double* f_sm_prev, const double* P_sm_prev,
                              const double* y, const double* part, const double* Snew, const int32_t* info1, const int32_t* info2,
                              double* W, double* n0, double* Nf, int32_t* bad_count, double* stA, double* stG, double* stC, double* stS,
                              double* stF, double* stFsm, double* stP, double* stPsm, int64_t* pos, int annealing, int32_t* sync,
                              void* stream) {
  if (!f_post || !c_post || !f_sm_prev || !P_sm_prev || !y || !part || !Snew || !info1 || !info2 || !W || !n0 || !Nf || !bad_count ||
      !stA || !stG || !stC || !stS || !stF || !stFsm || !stP || !stPsm || !pos || !sync || T <= 0)
    return -1;
  Finish2Args a{f_post, c_post, f_sm_prev, P_sm_prev, y, part, Snew, info1, info2, W, n0, Nf, bad_count,
                stA, stG, stC, stS, stF, stFsm, stP, stPsm, pos, sync, T, annealing};
  const long n2 = 2L * T * T;
  hipLaunchKernelGGL(k_chain_finish2, dim3((unsigned)std::min<long>(64, (n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int hgp_lds_chain_gather2_batched_f64(const hgp_chain_gather_desc* descs_dev, int n_chains, int T, void* stream) {
  if (n_chains == 0) return 0;
  if (!descs_dev || n_chains < 0 || T <= 0) return -1;
  const long total = 8L * T * T + 2L * T;
  // a grid-stride loop per chain: every block pays the two diagonal means (2 T strided loads + a barrier) before it copies, so a
  // block per 256 elements (2 050 blocks per chain at T = 256) spent more time there than copying: 288 us for 23 chains
  const long blocks = std::min<long>((total + 255) / 256, n_chains >= 8 ? 64 : 256);
  hipLaunchKernelGGL(k_chain_gather2_b, dim3((unsigned)blocks, (unsigned)n_chains), dim3(256), 0, (hipStream_t)stream, descs_dev);
  return launch_status();
}

int hgp_lds_chain_finish2_batched_f64(const hgp_chain_finish_desc* descs_dev, int n_chains, int T, void* stream) {
  if (n_chains == 0) return 0;
  if (!descs_dev || n_chains < 0 || T <= 0) return -1;
  const long n2 = 2L * T * T;
  hipLaunchKernelGGL(k_chain_finish2_b, dim3((unsigned)std::min<long>(64, (n2 + 255) / 256), (unsigned)n_chains), dim3(256), 0,
                     (hipStream_t)stream, descs_dev);
  return launch_status();
}

}  // extern "C"
