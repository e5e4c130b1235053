// Assignment tail of the switching variable on the device: GPI_HDP.LogLik (GPI_HDP.py:632-661) and the
// one-hot arg-max GPI_HDP._safe_exp (GPI_HDP.py:338-350) applied to log(alpha * beta), so that the [N,K] score matrix goes
// from the pair kernels through the message kernel to the label vector without leaving the GPU.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "hgp_internal.hpp"

namespace {

// LogLik(axis = 1): out[n,:] = q[n,:] - max_k q[n,k], rowmax[n] = that maximum - unless ANY row maximum is infinite, in
// which case the reference returns its input unchanged (GPI_HDP.py:646-648).  One workgroup: the any-infinite test spans all rows.
__global__ __launch_bounds__(1024) void k_loglik_rows(const double* __restrict__ q, int N, int K, double* __restrict__ out,
                                                      double* __restrict__ rowmax) {
  __shared__ int any_inf;
  q += (size_t)blockIdx.x * N * K;            // blockIdx.x = variant of a batch (the rule below is per score matrix)
  out += (size_t)blockIdx.x * N * K;
  if (rowmax) rowmax += (size_t)blockIdx.x * N;
  if (threadIdx.x == 0) any_inf = 0;
  __syncthreads();
  int bad = 0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    double m = q[(size_t)n * K];
    for (int k = 1; k < K; ++k) {
      const double v = q[(size_t)n * K + k];
      m = (v > m || v != v) ? ((m != m) ? m : v) : m;      // torch.max: a NaN anywhere in the row is the row's maximum
    }
    if (rowmax) rowmax[n] = m;
    bad |= isinf(m) ? 1 : 0;
  }
  if (bad) atomicOr(&any_inf, 1);
  __syncthreads();
  const bool keep = any_inf != 0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    double m = 0.0;
    if (!keep) {
      m = q[(size_t)n * K];
      for (int k = 1; k < K; ++k) {
        const double v = q[(size_t)n * K + k];
        m = (v > m || v != v) ? ((m != m) ? m : v) : m;
      }
    }
    for (int k = 0; k < K; ++k) out[(size_t)n * K + k] = q[(size_t)n * K + k] - m;
  }
}

// labels[n] = first arg-max over k of log(fmsg[n,k] * bmsg[n,k]); resp (optional) = its one-hot row.  torch.argmax treats NaN
// as the maximum (first NaN wins): a failed factorisation upstream lands on the NaN column here as it does in the reference.
__global__ __launch_bounds__(256) void k_assign(const double* __restrict__ fmsg, const double* __restrict__ bmsg, int N, int K,
                                                int64_t* __restrict__ labels, double* __restrict__ resp) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int best = 0;
  double bv = log(fmsg[(size_t)n * K] * bmsg[(size_t)n * K]);
  for (int k = 1; k < K; ++k) {
    const double v = log(fmsg[(size_t)n * K + k] * bmsg[(size_t)n * K + k]);
    if ((v > bv || v != v) && bv == bv) {      // a NaN already held is never replaced; the first NaN replaces any number
      bv = v;
      best = k;
    }
  }
  if (labels) labels[n] = best;
  if (resp)
    for (int k = 0; k < K; ++k) resp[(size_t)n * K + k] = (k == best) ? 1.0 : 0.0;
}

// last_log[v, k] = log(fmsg[v, N-1, k] * bmsg[v, N-1, k]) (what variational_local_terms hands back for the newest segment)
__global__ __launch_bounds__(64) void k_last_log(const double* __restrict__ fmsg, const double* __restrict__ bmsg, int N, int K,
                                                 double* __restrict__ out) {
  const size_t o = ((size_t)blockIdx.x * N + (N - 1)) * K;
  for (int k = threadIdx.x; k < K; k += 64) out[(size_t)blockIdx.x * K + k] = log(fmsg[o + k] * bmsg[o + k]);
}

}  // namespace

int hgp_internal_loglik_rows_b(const double* q, int N, int K, int B, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_loglik_rows, dim3(B), dim3(1024), 0, st, q, N, K, out, (double*)nullptr);
  return launch_status();
}

int hgp_internal_assign_b(const double* fmsg, const double* bmsg, int N, int K, int B, int64_t* labels, double* last_log, hipStream_t st) {
  const long rows = (long)N * B;
  hipLaunchKernelGGL(k_assign, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, fmsg, bmsg, (int)rows, K, labels, (double*)nullptr);
  if (last_log) hipLaunchKernelGGL(k_last_log, dim3(B), dim3(64), 0, st, fmsg, bmsg, N, K, last_log);
  return launch_status();
}

extern "C" {

int hgp_loglik_rows_f64(const double* q, int N, int K, double* out, double* rowmax, void* stream) {
  if (N == 0) return 0;
  if (!q || !out || N < 0 || K <= 0) return -1;
  hipLaunchKernelGGL(k_loglik_rows, dim3(1), dim3(1024), 0, (hipStream_t)stream, q, N, K, out, rowmax);
  return launch_status();
}

int hgp_assign_f64(const double* fmsg, const double* bmsg, int N, int K, int64_t* labels, double* resp, void* stream) {
  if (N == 0) return 0;
  if (!fmsg || !bmsg || (!labels && !resp) || N < 0 || K <= 0) return -1;
  hipLaunchKernelGGL(k_assign, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, fmsg, bmsg, N, K, labels, resp);
  return launch_status();
}

}  // extern "C"
