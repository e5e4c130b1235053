// Host-side declarations shared by the translation units of libhdpgpc_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "../../include/hdpgpc_hip.h"

constexpr int WAVES = 4;  // waves per workgroup of the one-wave-per-item kernels: one per SIMD of a CU

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : 1000 + (int)e;
}

static inline int nb_for(int n) {  // tile count, rounded to the instantiated sizes {2,4,6,8}
  int nb = (n + 15) / 16;
  nb = (nb + 1) & ~1;
  return nb < 2 ? 2 : nb;
}

static inline bool env_on(const char* name) {
  const char* v = getenv(name);
  return v && v[0] && v[0] != '0';
}

// Entries of E = exp(-h) and of K** / c with h > PAIRS_CUT (value < 2^-80 = 8.3e-25) are exact zeros, and a 16x16 block (k_pairs'
// static sweeps: a 4-row k-step of a block) whose entries are ALL below that is neither built nor multiplied.  What is dropped
// changes a covariance entry by less than 2 T max|M'| 2^-80 - below one ulp of the entry for every cluster the explicit-operator
// kernels score (the plan routes c^2 |K~^-1|^2 > 1e7 to the solve-based kernel).  With the reference's length-scale 1.2 on a
// unit-spaced grid the band is |k - j| <= 12 points: only the blocks |Kt - J| <= 1 survive (~60 % of the MFMA work at T = 128 gone)
// and, of their 12 k-steps per column panel, 10.  (Rounds 1-3 cut at 1e-36: |k - j| <= 15, no dead k-step.)  The decision is taken
// from the data (any grid), never from an assumed band structure.
constexpr double PAIRS_CUT = 55.45177444479562;   // 80 ln 2
// segments per launch pair of k_pairs<NB, true> / k_pairs<NB, false> (capacity of the fall-back list of the plan)
constexpr int PAIRS_FB_CAP = 65536;

// number of workgroups (and S scratch areas) of the solve-based pairs kernel (hgp_pairs_acc.hip)
static inline int acc_grid_for(int nb) { return nb > 8 ? 256 : 512; }

struct hgp_pairs_plan {
  int T, K, TP, NB;
  std::vector<double> theta;           // host copy [K,3]
  std::vector<int32_t> perm;           // clusters sorted by length-scale
  std::vector<int> grp_beg, grp_end;   // ranges of `perm` sharing one length-scale
  std::vector<double> grp_ell;
  // device carve-up
  double *d_theta, *d_scal, *d_A, *d_S, *d_Z, *d_Kinv, *d_P, *d_Q, *d_Mp, *d_ap, *d_xb;
  int32_t* d_perm;
  bool coop = false;   // one workgroup per pair (k_pairs_coop): always for T > 128
  // cooperative kernel: overflow areas for the E blocks of dense grids
  double* d_escr = nullptr;
  int32_t* d_eflags = nullptr;
  int nscr = 0;
  long escr_stride = 0;
  // solve-based evaluation (hgp_pairs_acc.hip) for the clusters whose explicit operator M' would lose digits:
  double acc_tol = 1e-9;        // cluster k takes the solve-based kernel when eps (c ||K~^-1||_inf)^2 > acc_tol
  double* d_Lop = nullptr;      // [K][NB(NB-1)/2 strictly lower tiles][64][4]: A operand of L[K,K']
  double* d_LTop = nullptr;     // same tiles: A operand of L[K,K']^T
  double* d_Dop = nullptr;      // [K][4 kinds][NB][64][4]: diagonal-block operands (W4, masked L_KK; plain and transposed)
  double* d_Sop = nullptr;      // [K][NB][NB][64][4]: A operand of tile (K,K') of 0.5 (Sigma + Sigma^T)
  double* d_mu = nullptr;       // [K][TP] prior means on the basis grid (zero padded)
  int32_t* d_acc_list = nullptr;   // [1 + K]: number of flagged clusters, then their ids
  double* d_sscr = nullptr;     // [acc_grid][NB*NB][64][4]: per-workgroup storage of S = K~^{-1} K*
  int32_t* d_fb = nullptr;      // [PAIRS_FB_CAP + 2]: fall-back list of k_pairs (see PairsArgs::fb); zero between launches
  int score_out = 0;            // hgp_pairs_plan_set_score_output: out_quad receives -0.5 quad - 0.5 Ts log(2 pi)
};

// hgp_pairs_acc.hip
int hgp_internal_acc_prep(hgp_pairs_plan* p, const double* mean, hipStream_t st);
int hgp_internal_pairs_acc(const hgp_pairs_plan* p, const double* x, const double* y, int N, int Ts, const double* first_noise,
                           const int32_t* sel, double* out_quad, double* out_logdet, int32_t* out_info, hipStream_t st);
size_t hgp_internal_acc_bytes(int TP, int K, size_t* sizes /*[6]*/);

// hgp_assign.hip: batched LogLik normalisation / arg-max of the state posterior (B score matrices [N, K] back to back)
int hgp_internal_loglik_rows_b(const double* q, int N, int K, int B, double* out, hipStream_t st);
int hgp_internal_assign_b(const double* fmsg, const double* bmsg, int N, int K, int B, int64_t* labels, double* last_log, hipStream_t st);

// hgp_matlik.hip: fused one-wave-per-item kernels of a8 / a9 (T <= HGP_MAX_T_WAVE)
int hgp_internal_lat_error_wave(const double* f_cur, const double* f_prev, const double* A, const double* Gamma, const double* covprev,
                                int T, int b, double* out, int32_t* info, hipStream_t st);
int hgp_internal_mniw_wave(const double* M, const double* Sigma, const double* m_mean, const double* m_r_cov, const double* scale,
                           int scale_is_diagonal, long prior_stride, int T, int b, double* out, int32_t* info, hipStream_t st);

// hgp_matlik_coop.hip: the same two terms for 128 < T <= HGP_MAX_T_COOP, one workgroup per item (factor once, packed factor in ws)
size_t hgp_internal_matlik_coop_ws_doubles(int T);
size_t hgp_internal_lat_coop_ws_doubles(int T);
int hgp_internal_lat_coop(const double* f_cur, const double* f_prev, const double* A, const double* Gamma, const double* P, int T, int b,
                          double* out, int32_t* info, double* ws, hipStream_t st);
int hgp_internal_mniw_coop(const double* M, const double* Sigma, const double* m_mean, const double* scale, long prior_stride, int T, int b,
                           double* out, int32_t* info, double* ws, hipStream_t st);

// arguments of the explicit-operator pair kernels (hgp_pairs.hip)
struct PairsArgs {
  const double* x;
  const double* y;
  int N, Ts;
  const double* xb;
  int T;
  const double* Mp;
  const double* ap;
  const double* scal;
  const int32_t* perm;   // sorted position -> original cluster id
  int kbeg, kend;        // range of sorted positions sharing one length-scale
  double ell;
  const double* first_noise;
  const int32_t* sel;    // optional [N]: segment n is scored against cluster sel[n] only; outputs are then [N]
#ifdef HGP_STAMPS
  unsigned long long* stamps;
#endif
  int K;
  double* out_quad;
  double* out_logdet;
  int32_t* out_info;
  // cooperative kernel only: global scratch for the E blocks that do not fit its LDS slots (dense grids)
  double* escr;          // [nscr][escr_stride]
  int32_t* eflags;       // [nscr] 0 = free
  int nscr;
  long escr_stride;
  int flags;             // bit 0: generic (mask-driven) sweeps even for a block-tridiagonal E (HGP_PAIRS_GENERIC=1: A/B runs and tests)
  double score_add;      // score output (hgp_pairs_plan_set_score_output): out_quad = score_on ? -0.5 quad + score_add : quad
  int score_on;
  int32_t* fb;           // k_pairs: fall-back list [0] = count, [1 .. PAIRS_FB_CAP] = segments, [1 + PAIRS_FB_CAP] = finished workgroups
};

#ifdef HGP_STAMPS
extern unsigned long long* hgp_internal_stamp_dev;
#endif
int hgp_internal_pairs_fast(const PairsArgs& a, int NB, bool coop, hipStream_t st);
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device), under a lock
int hgp_internal_ensure_dynamic_lds(const void* fn, size_t bytes);
