"""Kernel hyper-parameter fit of a new cluster: Adam on the exact marginal log-likelihood of ONE segment,
restating what IterativeGaussianProcess.fit_torch asks gpytorch for on the shared grid (GPI.py:610-770):

* model: constant mean m, ScaleKernel(RBFKernel) -> K = s exp(-0.5 d^2 / l^2) + noise I, Gaussian likelihood;
* parameterisation (gpytorch defaults): s = softplus(raw_s), l = softplus(raw_l), noise = lo + (hi - lo) sigmoid(raw_n) with
  (lo, hi) = the white-kernel bounds (GPI.py:629,655-656); every raw parameter and m start at 0;
* loss = - log p(y | theta) / T (ExactMarginalLogLikelihood divides by the number of points), Adam(lr = 0.1), at most 4000
  iterations, early stop once 1000 iterations have run and the last ten loss increments sum to 0 within 1e-4 (GPI.py:689-693);
* afterwards the length-scale is OVERWRITTEN with 1.2 (GPI.py:711); output-scale and noise are kept.

The value and the gradient of the log-likelihood come from the a10 kernels (Gram, Cholesky score, Cholesky inverse,
hgp_lml_grad_f64); only four scalars per iteration reach the host.  gpytorch is absent in the build container, so this
step has no golden vector ("parity unpinned"): tests check it against a NumPy restatement of the same
optimiser on the oracle's log-likelihood, and that the bounds hold and the loss settles.
"""
import math

import numpy as np
import torch

from . import ops

f64 = torch.float64


def _softplus(x):
    return math.log1p(math.exp(-abs(x))) + max(x, 0.0)


def _sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x)) if x >= 0 else math.exp(x) / (1.0 + math.exp(x))


def lml_value_grad(x, r, c, ell, noise, ones_row):
    """log N(r | 0, c RBF(ell) + noise I) and its gradient w.r.t. (log c, log ell, log noise), plus sum(K^-1 r).
    x [T], r [1,T], ones_row [1,T] device tensors.  Returns (value, grad[3], sum_alpha) as host floats, or None if K is not PD."""
    T = x.numel()
    K = ops.gram_rbf(x, None, c, ell, noise)
    items = ops.build_items([0], [0.0], [1])
    quad, logdet, info = ops.score_groups(r, None, K, *items, jitter_rel=0.0, want_logdet=True)
    Z, _ = ops.chol_inverse(K)
    Kinv = ops.gemm_batched(Z[0], Z[0], transA=True)
    alpha = ops.gemm_batched(Kinv, r.reshape(-1, 1))
    grad = ops.lml_grad(x, alpha.reshape(-1), Kinv, c, ell, noise)
    sa = ops.gemm_batched(ones_row, alpha)
    host = torch.cat([quad.reshape(1), logdet.reshape(1), grad.reshape(3), sa.reshape(1), info.to(f64).reshape(1)]).cpu().numpy()
    if host[6] != 0:
        return None
    val = -0.5 * host[0] - 0.5 * host[1] - 0.5 * T * math.log(2.0 * math.pi)
    return val, host[2:5], host[5]


def fit_kernel_adam(x, y, noise_bounds, device="cuda", max_iter=4000, lr=0.1, min_iter=1000, return_trace=False):
    """Returns (outputscale, lengthscale, noise, mean) at the end of the optimisation (the caller applies GPI.py:711)."""
    xh = np.asarray(x, dtype=np.float64).reshape(-1)
    yh = np.asarray(y, dtype=np.float64).reshape(-1)
    T = xh.size
    xd = torch.as_tensor(xh, dtype=f64, device=device)
    ones_row = torch.ones((1, T), dtype=f64, device=device)
    lo, hi = float(noise_bounds[0]), float(noise_bounds[1])
    p = np.zeros(4)                        # raw noise, mean, raw outputscale, raw lengthscale (gpytorch's parameter order)
    m1, m2 = np.zeros(4), np.zeros(4)
    b1, b2, eps = 0.9, 0.999, 1e-8          # torch.optim.Adam defaults
    losses = []
    for it in range(1, max_iter + 1):
        s_n = _sigmoid(p[0])
        noise = lo + (hi - lo) * s_n
        c, ell = _softplus(p[2]), _softplus(p[3])
        r = torch.as_tensor((yh - p[1])[None, :], dtype=f64, device=device)
        out = lml_value_grad(xd, r, c, ell, noise, ones_row)
        if out is None:
            raise torch.linalg.LinAlgError("kernel fit: K(theta) is not positive-definite")
        val, glog, sum_alpha = out
        losses.append(-val / T)
        g = np.array([glog[2] / noise * (hi - lo) * s_n * (1.0 - s_n),        # d noise / d raw
                      sum_alpha,                                                 # d L / d mean = 1^T K^-1 (y - m)
                      glog[0] / c * _sigmoid(p[2]),                             # d softplus = sigmoid
                      glog[1] / ell * _sigmoid(p[3])]) * (-1.0 / T)
        m1 = b1 * m1 + (1 - b1) * g
        m2 = b2 * m2 + (1 - b2) * g * g
        p = p - lr * (m1 / (1 - b1 ** it)) / (np.sqrt(m2 / (1 - b2 ** it)) + eps)
        if len(losses) > min_iter and abs(float(np.sum(np.subtract(losses[-10:], losses[-11:-1])))) <= 1e-4:
            break
    theta = (_softplus(p[2]), _softplus(p[3]), lo + (hi - lo) * _sigmoid(p[0]), p[1])
    return (theta, np.asarray(losses)) if return_trace else theta
