"""Host-side mirror of hdpgpc/hdpgpc/amtgp_warping_system.py: WarpPriorAMTGP's GP-prior score (:106-264, row a11) and the
batched warp fit of Warping_system (:266-735) - the optimisation itself runs as one HIP launch
(hgp_warp_batch_f64); this file keeps the reference's argument handling."""
import math

import numpy as np
import torch

from . import ops


class WarpPriorAMTGP:
    def __init__(self, noise_warp, bound_noise_warp=(1e-8, 1e2), jitter=1e-6, default_rho=1.0, default_omega=1.0,
                 normalize_x=True, device="cuda"):
        self.noise_warp = float(noise_warp)
        self.noise_bounds = tuple(bound_noise_warp)
        self.jitter = float(jitter)
        self.default_rho, self.default_omega = float(default_rho), float(default_omega)
        self.normalize_x = bool(normalize_x)
        self.theta = None
        self.device = device
        self._cache_key = None
        self._cache_K = None

    def _parse_theta(self):   # :141-153
        rho, omega = self.default_rho, self.default_omega
        th = self.theta
        try:
            if isinstance(th, (tuple, list)) and len(th) >= 2:
                rho, omega = float(th[0]), float(th[1])
            elif isinstance(th, dict):
                rho, omega = float(th.get("rho", rho)), float(th.get("omega", omega))
        except Exception:
            pass
        return max(rho, 1e-12), max(omega, 1e-12)

    def _clamped_noise(self):  # :155-158
        lo, hi = self.noise_bounds
        return min(max(self.noise_warp, lo), hi)

    def _cov(self, x):
        rho, omega = self._parse_theta()
        noise2 = self._clamped_noise()
        key = (int(x.numel()), rho, omega, noise2, float(x[0]), float(x[-1]), self.normalize_x)   # :176-186
        if self._cache_key != key:
            self._cache_K = ops.warp_cov(x, rho, omega, noise2 + self.jitter, self.normalize_x)
            self._cache_key = key
        return self._cache_K

    def log_sq_error_batch(self, x_model, x_warp_batch):
        """:224-264  W (B,T) / (T,B) / (B,T,1) -> (B,): -0.5 (w^T K^{-1} w + logdet K + T log 2pi)."""
        x = torch.as_tensor(x_model, dtype=torch.float64).to(self.device).reshape(-1).contiguous()
        W = x_warp_batch
        if isinstance(W, list):
            W = torch.stack([torch.as_tensor(w) for w in W], dim=0)
        W = torch.as_tensor(W, dtype=torch.float64).to(self.device)
        if W.ndim == 3 and W.shape[-1] == 1:
            W = W[..., 0]
        if W.shape[0] == x.numel() and W.shape[1] != x.numel():
            W = W.transpose(0, 1)
        assert W.ndim == 2 and W.shape[1] == x.numel(), f"Expected (B,T), got {tuple(W.shape)}"
        W = W.contiguous()
        K = self._cov(x)
        items = ops.build_items([0], [0.0], [W.shape[0]])
        quad, logdet, info = ops.score_groups(W, None, K, *items, jitter_rel=0.0, want_logdet=True)
        ops.raise_on_info(info, "WarpPriorAMTGP")
        return -0.5 * (quad + logdet + x.numel() * math.log(2.0 * math.pi))

    def log_sq_error(self, x_model, x_warp):
        """:197-221"""
        w = torch.as_tensor(x_warp, dtype=torch.float64).reshape(1, -1)
        return self.log_sq_error_batch(x_model, w)[0]


class Warping_system:
    """amtgp_warping_system.py:266-735: monotone time-warps by positive increments on a coarse control grid, MAP objective
    data fit + smoothness + amplitude, warm start across calls.  ``compute_warp_batch`` is the vectorised entry point
    (GPI_HDP.warp_batch_by_resp_amtgp_cached / compute_warp_actual_state_amtgp use it); ``compute_warp`` fits one sample
    through the same kernel."""

    def __init__(self, x_basis_warp, noise_warp=1e-2, bound_noise_warp=(1e-6, 1e2), recursive=True, cuda=True, bayesian=True,
                 mode="balanced", n_ctrl=8, lr=5e-2, lambda_smooth=200.0, lambda_amp=1e-3, device="cuda"):
        self.device = device
        self.x_basis = torch.as_tensor(np.asarray(x_basis_warp, dtype=np.float64)).reshape(-1).to(device)
        self.T = self.x_basis.numel()
        self.noise_warp_default = float(noise_warp)
        self.noise_bounds = tuple(bound_noise_warp)
        self.recursive, self.bayesian, self.mode = bool(recursive), bool(bayesian), str(mode)
        self.n_ctrl = int(max(4, min(n_ctrl, self.T)))
        self.lr = float(lr)
        self.lambda_smooth_base, self.lambda_amp_base = float(lambda_smooth), float(lambda_amp)
        self._u_ctrl_prev = None
        self.warp_gp = WarpPriorAMTGP(noise_warp=noise_warp, bound_noise_warp=bound_noise_warp, default_rho=1.0,
                                      default_omega=1.0, device=device)

    def _theta_to_lambdas(self, theta):   # :361-395
        lam_s, lam_a = self.lambda_smooth_base, self.lambda_amp_base
        try:
            if isinstance(theta, (tuple, list)) and len(theta) >= 2:
                rho, omg = float(theta[0]), float(theta[1])
            elif isinstance(theta, dict):
                rho, omg = float(theta.get("rho", 1.0)), float(theta.get("omega", 1.0))
            else:
                return lam_s, lam_a
            return self.lambda_smooth_base / (rho * rho + 1e-12), self.lambda_amp_base / (omg * omg + 1e-12)
        except Exception:
            return lam_s, lam_a

    def compute_warp_batch(self, x_model, y_target_batch, y_model, theta=None, noise=None, weights=None, visualize=False,
                           verbose=False, train_iter=50):
        """:548-735.  Returns (x_warp (B,T,1), y_warp (B,T,D), lik_warp (B,), losses dict of batch-mean traces)."""
        dev = self.device
        x = torch.as_tensor(np.asarray(x_model.cpu() if torch.is_tensor(x_model) else x_model, dtype=np.float64)).reshape(-1).to(dev)
        T = x.numel()
        Yt = torch.as_tensor(y_target_batch, dtype=torch.float64).to(dev)
        if Yt.ndim == 2:
            Yt = Yt[:, :, None]
        elif Yt.ndim == 1:
            Yt = Yt[None, :, None]
        B, D = Yt.shape[0], Yt.shape[2]
        assert Yt.shape[1] == T, f"y_target_batch length mismatch: got {Yt.shape[1]} expected {T}"
        Ym = torch.as_tensor(y_model, dtype=torch.float64).to(dev)
        if Ym.ndim == 1:
            Ym = Ym[:, None]
        if Ym.ndim == 3 and Ym.shape[0] == 1:
            Ym = Ym[0]
        assert Ym.shape[-2] == T, f"y_model length mismatch: got {Ym.shape[-2]} expected {T}"
        assert Ym.ndim == 2 or Ym.shape[0] == B, f"y_model batch mismatch: got {Ym.shape[0]} expected {B}"
        Ym = Ym[..., :D].contiguous()
        if T != self.T:   # :601-610
            self.x_basis, self.T = x, T
            self.n_ctrl = int(max(4, min(self.n_ctrl, self.T)))
            self.warp_gp = WarpPriorAMTGP(noise_warp=self.noise_warp_default, bound_noise_warp=self.noise_bounds, default_rho=1.0,
                                          default_omega=1.0, device=dev)
        self.warp_gp.theta = theta
        if noise is None:
            n = self.noise_warp_default
        else:
            nz = torch.as_tensor(noise, dtype=torch.float64)
            n = float(nz.mean()) if nz.numel() > 1 else float(nz.reshape(()))
            n = min(max(n, self.noise_bounds[0]), self.noise_bounds[1])
        lam_s, lam_a = self._theta_to_lambdas(theta)
        w = None if weights is None else torch.clamp(torch.as_tensor(weights, dtype=torch.float64).reshape(-1), min=0.0).to(dev)
        u0 = None
        if self.recursive and self._u_ctrl_prev is not None and self._u_ctrl_prev.numel() == self.n_ctrl:
            u0 = self._u_ctrl_prev
        u, xw, yw, tr = ops.warp_batch(x, Yt.contiguous(), Ym, self.n_ctrl, int(train_iter), n, lam_s, lam_a, self.lr, weights=w, u0=u0)
        lik = self.warp_gp.log_sq_error_batch(x, xw)
        if self.recursive:
            self._u_ctrl_prev = u.mean(dim=0)            # warm start of the next call: the mean control vector (:727-729)
        wn = (torch.ones(B, dtype=torch.float64, device=dev) if w is None else w)
        mean_tr = (torch.einsum("b,bik->ik", wn, tr) / (wn.sum() + 1e-12)).cpu().numpy()
        losses = {"loss": list(mean_tr[:, 0]), "data": list(mean_tr[:, 1]), "smooth": list(mean_tr[:, 2]), "amp": list(mean_tr[:, 3])}
        return xw[:, :, None], yw, lik, losses

    def compute_warp(self, x_model, y_target, y_model, theta=None, noise=None, visualize=False, verbose=False, train_iter=50):
        """One sample through the batched kernel; returns (x_warp (T,1), y_warp (T,D), lik, losses)."""
        yt = torch.as_tensor(y_target, dtype=torch.float64)
        yt = yt.reshape(1, yt.shape[0], -1)
        xw, yw, lik, losses = self.compute_warp_batch(x_model, yt, y_model, theta=theta, noise=noise, train_iter=train_iter)
        return xw[0], yw[0], lik[0], losses
