// Warping_system.compute_warp_batch (amtgp_warping_system.py:548-735) - B independent monotone time-warps
// fitted by Adam - as ONE kernel: one wavefront per sample, the whole optimisation (forward pass, hand-derived reverse pass,
// Adam update; 50-250 iterations) inside the kernel with every array in LDS; nothing returns to the host between
// iterations (the reference runs torch autograd + torch.optim.Adam on [B, n_ctrl] parameters).
//
//   u_T   = linear interpolation of the n_ctrl control values onto the T grid points (F.interpolate, align_corners)
//   inc   = softplus(u_T) + 1e-6;  g_raw = cumsum(inc);  g = x_0 + (x_end - x_0) (g_raw - g_raw[0]) / (g_raw[-1] - g_raw[0] + 1e-12)
//   Y_w   = linear interpolation of the target y at g;  x_warp = g - x
//   loss  = 0.5 |Y_w - Y_model|^2 / (noise + 1e-12) + lam_s |D2 x_warp|^2 + lam_a |x_warp|^2, weighted mean over the batch
// The two scans (cumsum and its adjoint) run on lane 0 in index order, as torch.cumsum does; everything else is one
// element per lane and wave reductions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hgp_internal.hpp"
#include "tile_f64.hpp"

using namespace hgp;

namespace {

constexpr int WARP_MAX_T = 256, WARP_MAX_D = 4, WARP_MAX_CTRL = 32;

struct WarpArgs {
  const double* x;        // [T] strictly increasing
  const double* Yt;       // [B,T,D] observations to be warped
  const double* Ym;       // [B or 1,T,D] model mean (stride ym_stride: 0 = shared)
  long ym_stride;
  int T, B, D, n_ctrl, iters;
  double noise, lam_s, lam_a, lr;
  const double* weights;  // [B] or NULL
  const double* u0;       // [n_ctrl] warm start or NULL
  double* u_out;          // [B,n_ctrl]
  double* xw_out;         // [B,T]
  double* yw_out;         // [B,T,D]
  double* loss_out;       // [B,iters,4] (loss, data, smooth, amp per sample, unweighted) or NULL
};

__global__ __launch_bounds__(64) void k_warp_batch(WarpArgs a) {
  __shared__ double xg[WARP_MAX_T], yt[WARP_MAX_T * WARP_MAX_D], ym[WARP_MAX_T * WARP_MAX_D];
  __shared__ double uT[WARP_MAX_T], zs[WARP_MAX_T], graw[WARP_MAX_T], xw[WARP_MAX_T], dg[WARP_MAX_T], d2[WARP_MAX_T];
  __shared__ double u[WARP_MAX_CTRL], m1[WARP_MAX_CTRL], m2[WARP_MAX_CTRL], du[WARP_MAX_CTRL];
  __shared__ double red[4];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int T = a.T, D = a.D, n = a.n_ctrl;
  const double* Ytb = a.Yt + (size_t)b * T * D;
  const double* Ymb = a.Ym + (size_t)b * a.ym_stride;
  for (int i = lane; i < T; i += 64) xg[i] = a.x[i];
  for (int i = lane; i < T * D; i += 64) {
    yt[i] = Ytb[i];
    ym[i] = Ymb[i];
  }
  if (lane < n) {
    u[lane] = a.u0 ? a.u0[lane] : 0.0;
    m1[lane] = 0.0;
    m2[lane] = 0.0;
  }
  // weight of this sample in the batch-mean loss: w_b / (sum w + 1e-12)
  double wsum = 0.0;
  for (int i = lane; i < a.B; i += 64) wsum += a.weights ? fmax(a.weights[i], 0.0) : 1.0;
  wsum = wave_sum(wsum);
  const double wb = (a.weights ? fmax(a.weights[b], 0.0) : 1.0) / (wsum + 1e-12);
  __syncthreads();
  const double x0 = xg[0], xN = xg[T - 1], span = xN - x0;
  const double scale = (T > 1) ? (double)(n - 1) / (double)(T - 1) : 0.0;
  const double inv_noise = 1.0 / (a.noise + 1e-12);

  for (int it = 0; it <= a.iters; ++it) {
    const bool last = (it == a.iters);   // final forward pass with the optimised controls (no update)
    // ---- forward: controls -> increments
    for (int t = lane; t < T; t += 64) {
      const double src = scale * (double)t;
      int i0 = (int)src;
      i0 = i0 < n - 1 ? i0 : n - 1;
      const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
      double l1 = src - (double)i0;
      l1 = fmin(fmax(l1, 0.0), 1.0);
      const double v = u[i0] * (1.0 - l1) + u[i1] * l1;
      uT[t] = v;
      double z = 0.0, sp;
      if (v > 20.0) sp = v;
      else {
        z = exp(v);
        sp = log1p(z);
      }
      zs[t] = (v > 20.0) ? 1.0 : z / (z + 1.0);      // d softplus / d v
      graw[t] = sp + 1e-6;                            // inc, scanned in place below
    }
    __syncthreads();
    if (lane == 0) {                                  // cumsum in index order
      double acc = 0.0;
      for (int t = 0; t < T; ++t) {
        acc += graw[t];
        graw[t] = acc;
      }
    }
    __syncthreads();
    const double g0 = graw[0], S = graw[T - 1] - g0 + 1e-12;
    // ---- forward: grid, interpolation, residuals; backward seeds w.r.t. g
    double sse = 0.0, ap = 0.0;
    for (int t = lane; t < T; t += 64) {
      const double q = (graw[t] - g0) / S;
      const double g = x0 + span * q;
      const double w = g - xg[t];
      xw[t] = w;
      ap = fma(w, w, ap);
      const double xq = fmin(fmax(g, x0), xN);
      int lo = 0, hi = T;                             // lower bound: first index with xg[idx] >= xq
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xg[mid] < xq) lo = mid + 1;
        else hi = mid;
      }
      int ih = lo < 1 ? 1 : (lo > T - 1 ? T - 1 : lo);
      const int il = ih - 1;
      const double den = xg[ih] - xg[il] + 1e-12;
      const double tt = (xq - xg[il]) / den;
      double dxq = 0.0;
      for (int d = 0; d < D; ++d) {
        const double ylo = yt[il * D + d], yhi = yt[ih * D + d];
        const double yw = (1.0 - tt) * ylo + tt * yhi;
        if (last) a.yw_out[((size_t)b * T + t) * D + d] = yw;
        const double r = yw - ym[t * D + d];
        sse = fma(r, r, sse);
        dxq = fma(r * inv_noise, (yhi - ylo) / den, dxq);
      }
      dg[t] = (g >= x0 && g <= xN) ? dxq : 0.0;       // clamp passes the gradient inside the domain only
      if (last) a.xw_out[(size_t)b * T + t] = w;
    }
    if (last) break;
    __syncthreads();
    double spn = 0.0;
    for (int t = lane; t < T; t += 64) {
      double v = 0.0;
      if (t + 2 < T) {
        v = xw[t] - 2.0 * xw[t + 1] + xw[t + 2];
        spn = fma(v, v, spn);
      }
      d2[t] = v;
    }
    __syncthreads();
    sse = wave_sum(sse);
    ap = wave_sum(ap);
    spn = wave_sum(spn);
    if (a.loss_out && lane == 0) {
      double* lo_ = a.loss_out + ((size_t)b * a.iters + it) * 4;
      const double data = 0.5 * sse * inv_noise;
      lo_[0] = data + a.lam_s * spn + a.lam_a * ap;
      lo_[1] = data;
      lo_[2] = a.lam_s * spn;
      lo_[3] = a.lam_a * ap;
    }
    // ---- backward: d loss / d g (penalties through x_warp = g - x), then through the normalisation
    double sdq = 0.0, sdqq = 0.0;
    for (int t = lane; t < T; t += 64) {
      double gsm = d2[t];                              // adjoint of the second difference
      if (t >= 1) gsm -= 2.0 * d2[t - 1];
      if (t >= 2) gsm += d2[t - 2];
      const double dgt = dg[t] + 2.0 * a.lam_a * xw[t] + 2.0 * a.lam_s * gsm;
      const double dq = span * dgt;                    // g = x0 + span q
      const double q = (graw[t] - g0) / S;
      sdq += (t >= 1) ? dq / S : 0.0;                  // sum over t >= 1 of dR[t]
      sdqq = fma(dq, q, sdqq);
      dg[t] = dq / S;                                  // dR[t]
    }
    sdq = wave_sum(sdq);
    sdqq = wave_sum(sdqq);
    __syncthreads();
    if (lane == 0) {
      const double dS = -sdqq / S;
      dg[T - 1] += dS;                                 // S = g_raw[T-1] - g_raw[0] + eps
      dg[0] = -sdq - dS;                               // d g_raw[0]
      double acc = 0.0;                                // adjoint of the cumsum: suffix sums
      for (int t = T - 1; t >= 0; --t) {
        acc += dg[t];
        dg[t] = acc;
      }
    }
    __syncthreads();
    // ---- backward: increments -> controls (adjoint of the interpolation), Adam step
    for (int i = 0; i < n; ++i) {
      double part = 0.0;
      for (int t = lane; t < T; t += 64) {
        const double src = scale * (double)t;
        int i0 = (int)src;
        i0 = i0 < n - 1 ? i0 : n - 1;
        const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
        double l1 = src - (double)i0;
        l1 = fmin(fmax(l1, 0.0), 1.0);
        const double duT = dg[t] * zs[t];
        if (i0 == i) part = fma(1.0 - l1, duT, part);
        if (i1 == i) part = fma(l1, duT, part);
      }
      part = wave_sum(part);
      if (lane == 0) du[i] = part * wb;
    }
    __syncthreads();
    if (lane < n) {   // torch.optim.Adam (defaults): lerp / addcmul moments, bias-corrected step
      const double gk = du[lane];
      const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
      const double step = (double)(it + 1);
      m1[lane] = m1[lane] + (gk - m1[lane]) * (1.0 - b1);
      m2[lane] = m2[lane] * b2 + (1.0 - b2) * gk * gk;
      const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
      const double denom = sqrt(m2[lane]) / sqrt(bc2) + eps;
      u[lane] = u[lane] - (a.lr / bc1) * (m1[lane] / denom);
    }
    __syncthreads();
  }
  __syncthreads();
  if (lane < n) a.u_out[(size_t)b * n + lane] = u[lane];
}

}  // namespace

extern "C" int hgp_warp_batch_f64(const double* x, const double* Yt, const double* Ym, long ym_stride, int T, int B, int D, int n_ctrl,
                                  int iters, double noise, double lam_s, double lam_a, double lr, const double* weights,
                                  const double* u0, double* u_out, double* xw_out, double* yw_out, double* loss_out, void* stream) {
  if (B == 0) return 0;
  if (!x || !Yt || !Ym || !u_out || !xw_out || !yw_out || T < 2 || B < 0 || D < 1 || n_ctrl < 2 || iters < 0 || !(noise >= 0.0))
    return -1;
  if (T > WARP_MAX_T || D > WARP_MAX_D || n_ctrl > WARP_MAX_CTRL) return -2;
  WarpArgs a{x, Yt, Ym, ym_stride, T, B, D, n_ctrl, iters, noise, lam_s, lam_a, lr, weights, u0, u_out, xw_out, yw_out, loss_out};
  hipLaunchKernelGGL(k_warp_batch, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
  return launch_status();
}
