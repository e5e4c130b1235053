"""Host-side mirror of hdpgpc/hdpgpc/GPI.py for the GP-emission hot path (same names, argument meaning and
error behaviour); all arithmetic runs in HIP kernels through hdpgpc_amd.ops.

Built: the kernel object (scikit-learn's ConstantKernel*RBF + WhiteKernel, GPI_HDP.py:164-166),
IterativeGaussianProcess.pred_dist (GPI.py:457-503), pred_latent_dist (GPI.py:505-560),
log_marginal_likelihood (GPI.py:976-1056, value and gradient).  Not built here: posterior / backward (the LDS recursion,
SURVEY.md 8f-1) and fit_torch (gpytorch, 8f-2).
"""
import math

import numpy as np
import torch

from . import ops

torch_f64 = torch.float64


class RBFWhiteKernel:
    """ConstantKernel(c) * RBF(ell) + WhiteKernel(noise) with scikit-learn's call semantics:
    k(X) adds the white noise on the diagonal, k(X, Y) does not (GPI.py:136-139 relies on that)."""

    def __init__(self, constant_value=1.0, length_scale=1.0, noise_level=1.0, device="cuda"):
        self.constant_value = float(constant_value)
        self.length_scale = float(length_scale)
        self.noise_level = float(noise_level)
        self.device = device

    @property
    def theta(self):  # log-transformed, as sklearn
        return np.log([self.constant_value, self.length_scale, self.noise_level])

    @theta.setter
    def theta(self, t):
        self.constant_value, self.length_scale, self.noise_level = (float(v) for v in np.exp(np.asarray(t, dtype=np.float64)))

    def clone_with_theta(self, theta):
        k = RBFWhiteKernel(device=self.device)
        k.theta = theta
        return k

    def params(self):
        return (self.constant_value, self.length_scale, self.noise_level)

    def _dev(self, x):
        return torch.as_tensor(np.asarray(x.detach().cpu()) if torch.is_tensor(x) and not x.is_cuda else x,
                               dtype=torch_f64, device=self.device).reshape(-1).contiguous()

    def __call__(self, X, Y=None):
        return ops.gram_rbf(self._dev(X), None if Y is None else self._dev(Y), self.constant_value, self.length_scale,
                            self.noise_level)


class IterativeGaussianProcess:
    """GPI.py:25-70: holds the kernel and the basis grid; the scoring half only needs pred_dist."""

    def __init__(self, kernel, x_basis, cuda=True, verbose=False):
        self.kernel = kernel
        self.device = kernel.device
        self.x_basis = self.cond_to_torch(x_basis).reshape(-1, 1)
        self.cuda = cuda
        self.verbose = verbose
        self.fitted = False

    def __getstate__(self):
        d = dict(self.__dict__)
        d["x_basis"] = self.x_basis.detach().cpu()
        return d

    def cond_to_torch(self, x):
        if x is None:
            return None
        return torch.as_tensor(x, dtype=torch_f64).to(self.device)

    @staticmethod
    def _iso(Sigma):
        d = torch.diagonal(Sigma)
        return bool(torch.all(torch.isclose(d, torch.mean(d))))     # GPI.py:497 (torch defaults)

    @staticmethod
    def _spd_solve(Z, Kt, B):
        """Kt^{-1} B from the explicit Z = chol(Kt)^{-1}: X = Z^T (Z B), then ONE step of iterative refinement,
        X += Z^T Z (B - Kt X).  The reference calls cholesky_solve (two triangular solves, GPI.py:492); with the
        kernels' length-scale Kt is ill-conditioned and the bare product of explicit inverses is 10-100 x further
        from that result than the refined one (gate of tests/test_gpu_mirror_api.py).  GEMMs only."""
        X = ops.gemm_batched(Z, ops.gemm_batched(Z, B), transA=True)
        R = ops.gemm_batched(Kt, X, alpha=-1.0, add=B)
        return ops.gemm_batched(Z, ops.gemm_batched(Z, R), transA=True, add=X)

    def pred_dist(self, x_post, x_fixed, mean_prior, Sigma):
        """GPI.py:457-503.  Returns (f_star [T*,1], cov_f [T*,T*]) - the explicit predictive distribution.
        The N x K scoring path does not call this (it never materialises cov_f in HBM): see ops.PairsPlan."""
        x_p = self.cond_to_torch(x_post).reshape(-1, 1)
        x_f = self.cond_to_torch(x_fixed).reshape(-1, 1)
        mean_prior = self.cond_to_torch(mean_prior).reshape(-1, 1)
        Sigma = self.cond_to_torch(Sigma)
        if x_f.shape == x_p.shape and torch.equal(x_f, x_p):        # GPI.py:467-468
            return mean_prior, Sigma
        c, ell, noise = self.kernel.params()
        K_X_X = ops.gram_rbf(x_f, x_f, c, ell)                      # two-argument calls: no white noise
        K_X_Xs = ops.gram_rbf(x_f, x_p, c, ell)
        jitter = 1e-4 * max(float(torch.mean(torch.diagonal(Sigma).abs())), np.finfo(np.float64).eps)
        L, info, Linv = ops.potrf_batched(K_X_X, 0.0, jitter, want_inv=True)
        ops.raise_on_info(info, "pred_dist")
        K_solve = self._spd_solve(Linv[0], ops.gram_rbf(x_f, None, c, ell, jitter), K_X_Xs)   # K~^{-1} K*
        f_star = ops.gemm_batched(K_solve, mean_prior, transA=True)
        m = x_p.shape[0]
        if self._iso(Sigma):
            cov_f = torch.mean(torch.diagonal(Sigma)) * torch.eye(m, dtype=torch_f64, device=self.device)
        else:
            K_Xs_Xs = ops.gram_rbf(x_p, None, c, ell, noise)        # one-argument call: white noise included
            SK = ops.gemm_batched(Sigma.contiguous(), K_solve)
            cov_f = K_Xs_Xs - ops.gemm_batched(K_X_Xs, K_solve, transA=True) + ops.gemm_batched(K_solve, SK, transA=True)
            cov_f = 0.5 * (cov_f + cov_f.T) + 1e-6 * torch.eye(m, dtype=torch_f64, device=self.device)
        return f_star, cov_f

    def pred_latent_dist(self, x_post, x_fixed, mean_prior, cov_prior):
        """GPI.py:505-560."""
        x_p = self.cond_to_torch(x_post).reshape(-1, 1)
        x_f = self.cond_to_torch(x_fixed).reshape(-1, 1)
        mean_prior = self.cond_to_torch(mean_prior).reshape(-1, 1)
        cov_prior = self.cond_to_torch(cov_prior).contiguous()
        if x_f.shape == x_p.shape and torch.equal(x_f, x_p):
            return mean_prior, cov_prior
        c, ell, _ = self.kernel.params()
        K_X_X = ops.gram_rbf(x_f, x_f, c, ell)
        K_X_Xs = ops.gram_rbf(x_f, x_p, c, ell)
        K_Xs_Xs = ops.gram_rbf(x_p, x_p, c, ell)
        L, info, Z = ops.potrf_batched(K_X_X, 0.0, 1e-4, want_inv=True)
        ops.raise_on_info(info, "pred_latent_dist")
        Z = Z[0]
        Kt = ops.gram_rbf(x_f, None, c, ell, 1e-4)

        def solve(B):   # (K + 1e-4 I)^{-1} B
            return self._spd_solve(Z, Kt, B)

        f_star = ops.gemm_batched(K_X_Xs, solve(mean_prior), transA=True)
        sol_K = solve(K_X_Xs)
        term_data = ops.gemm_batched(K_X_Xs, sol_K, transA=True)
        term_prior = ops.gemm_batched(K_X_Xs, solve(ops.gemm_batched(cov_prior, sol_K)), transA=True)
        return f_star, K_Xs_Xs - term_data + term_prior

    def log_marginal_likelihood(self, x_train, y_train, alpha_ini=None, theta=None, eval_gradient=False,
                                clone_kernel=True, faithful=True):
        """GPI.py:976-1056 (value, and with eval_gradient the gradient w.r.t. the log-parameters, GPI.py:1046-1051).
        ``faithful=True`` reproduces the reference as written (it hands K, not L, to cho_solve, GPI.py:1043);
        ``faithful=False`` is the textbook value."""
        kernel = self.kernel if theta is None else self.kernel.clone_with_theta(theta)
        if eval_gradient and theta is None:
            raise ValueError("Gradient can only be evaluated for theta!=None")     # GPI.py:1017-1020
        x = self.cond_to_torch(x_train).reshape(-1)
        y = self.cond_to_torch(y_train).reshape(1, -1).contiguous()
        K = kernel(x)
        T = x.numel()
        items = ops.build_items([0], [0.0], [1])
        quad, logdet, info = ops.score_groups(y, None, K, *items, jitter_rel=0.0, want_logdet=True)
        if int(info[0]) != 0:
            return (-np.inf, np.zeros(3)) if eval_gradient else -np.inf      # GPI.py:1035-1037
        alpha = None
        if faithful:
            alpha, quad0 = ops.trsv_lower_solve(K, y) if eval_gradient else (None, ops.trsv_lower_quad(K, y))
        else:
            quad0 = quad[0]
        val = float(-0.5 * quad0 - 0.5 * logdet[0] - T / 2.0 * math.log(2.0 * math.pi))
        if not eval_gradient:
            return val
        # gradient w.r.t. the log-parameters (GPI.py:1046-1051): K^{-1} = Z^T Z with Z = chol(K)^{-1}
        if T <= 128:
            Z, _ = ops.chol_inverse(K)
        else:
            _, _, Z = ops.potrf_batched(K, 0.0, 0.0, want_inv=True)
        Kinv = ops.gemm_batched(Z[0], Z[0], transA=True)
        if alpha is None:
            alpha = ops.gemm_batched(Kinv, y.reshape(-1, 1)).reshape(-1)
        c, ell, noise = kernel.params()
        grad = ops.lml_grad(x, alpha, Kinv, c, ell, noise)
        return val, grad.cpu().numpy()
