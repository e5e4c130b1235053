"""Host-side mirror of hdpgpc/hdpgpc/amtgp_warping_system.py: WarpPriorAMTGP's GP-prior score (:106-264, row a11) and the
batched warp fit of Warping_system (:266-735) - the optimisation itself runs as one HIP launch
(hgp_warp_batch_f64); this file keeps the reference's argument handling."""
import math

import numpy as np
import torch

from . import ops


def _rho_omega(theta, rho, omega):
    """theta as (rho, omega, ...) or {"rho": .., "omega": ..}; anything else (or unparsable) keeps the defaults."""
    if isinstance(theta, dict):
        pair = (theta.get("rho", rho), theta.get("omega", omega))
    elif isinstance(theta, (tuple, list)) and len(theta) >= 2:
        pair = tuple(theta[:2])
    else:
        return rho, omega, False
    try:
        return float(pair[0]), float(pair[1]), True
    except (TypeError, ValueError):
        return rho, omega, False


def _f64(a, device):
    return torch.as_tensor(a, dtype=torch.float64).to(device)


class WarpPriorAMTGP:
    def __init__(self, noise_warp, bound_noise_warp=(1e-8, 1e2), jitter=1e-6, default_rho=1.0, default_omega=1.0,
                 normalize_x=True, device="cuda"):
        self.device = device
        self.theta = None
        self.noise_warp, self.jitter = float(noise_warp), float(jitter)
        self.noise_bounds = (bound_noise_warp[0], bound_noise_warp[1])
        self.default_rho, self.default_omega = float(default_rho), float(default_omega)
        self.normalize_x = normalize_x is True or bool(normalize_x)
        self._cache_key = self._cache_K = None

    def _parse_theta(self):   # amtgp_warping_system.py:141-153
        rho, omega, _ = _rho_omega(self.theta, self.default_rho, self.default_omega)
        return max(rho, 1e-12), max(omega, 1e-12)

    def _clamped_noise(self):  # :155-158
        return float(np.clip(self.noise_warp, self.noise_bounds[0], self.noise_bounds[1]))

    def _cov(self, x):
        """Prior covariance on the grid x, cached on everything it depends on (:176-186)."""
        (rho, omega), noise2 = self._parse_theta(), self._clamped_noise()
        key = (int(x.numel()), rho, omega, noise2, float(x[0]), float(x[-1]), self.normalize_x)
        if key != self._cache_key:
            self._cache_key, self._cache_K = key, ops.warp_cov(x, rho, omega, noise2 + self.jitter, self.normalize_x)
        return self._cache_K

    def log_sq_error_batch(self, x_model, x_warp_batch):
        """:224-264  W (B,T) / (T,B) / (B,T,1) -> (B,): -0.5 (w^T K^{-1} w + logdet K + T log 2pi)."""
        x = _f64(x_model, self.device).reshape(-1).contiguous()
        n = x.numel()
        W = x_warp_batch
        W = _f64(torch.stack([torch.as_tensor(w) for w in W]) if isinstance(W, list) else W, self.device)
        if W.ndim == 3 and W.shape[2] == 1:
            W = W.squeeze(2)
        if W.ndim == 2 and W.shape[0] == n and W.shape[1] != n:      # (T, B) -> (B, T)
            W = W.t()
        if W.ndim != 2 or W.shape[1] != n:
            raise AssertionError(f"Expected (B,T), got {tuple(W.shape)}")
        W = W.contiguous()
        items = ops.build_items([0], [0.0], [W.shape[0]])
        quad, logdet, info = ops.score_groups(W, None, self._cov(x), *items, jitter_rel=0.0, want_logdet=True)
        ops.raise_on_info(info, "WarpPriorAMTGP")
        return -0.5 * (quad + logdet + n * math.log(2.0 * math.pi))

    def log_sq_error(self, x_model, x_warp):
        """:197-221"""
        return self.log_sq_error_batch(x_model, torch.as_tensor(x_warp, dtype=torch.float64).reshape(1, -1))[0]


class Warping_system:
    """amtgp_warping_system.py:266-735: monotone time-warps by positive increments on a coarse control grid, MAP objective
    data fit + smoothness + amplitude, warm start across calls.  ``compute_warp_batch`` is the vectorised entry point
    (GPI_HDP.warp_batch_by_resp_amtgp_cached / compute_warp_actual_state_amtgp use it); ``compute_warp`` fits one sample
    through the same kernel."""

    def __init__(self, x_basis_warp, noise_warp=1e-2, bound_noise_warp=(1e-6, 1e2), recursive=True, cuda=True, bayesian=True,
                 mode="balanced", n_ctrl=8, lr=5e-2, lambda_smooth=200.0, lambda_amp=1e-3, device="cuda"):
        self.device = device
        self.recursive, self.bayesian, self.mode = bool(recursive), bool(bayesian), str(mode)
        self.lr, self.noise_warp_default = float(lr), float(noise_warp)
        self.noise_bounds = (bound_noise_warp[0], bound_noise_warp[1])
        self.lambda_smooth_base, self.lambda_amp_base = float(lambda_smooth), float(lambda_amp)
        self._u_ctrl_prev = None                                         # warm start (mean control vector of the last call)
        self._n_ctrl_asked = n_ctrl
        self._set_grid(_f64(np.asarray(x_basis_warp, dtype=np.float64), device).reshape(-1))

    def _set_grid(self, x):
        """(Re)bind the grid: at least 4 and at most T control points, a fresh prior (:601-610)."""
        self.x_basis, self.T = x, int(x.numel())
        self.n_ctrl = int(min(max(getattr(self, "n_ctrl", self._n_ctrl_asked), 4), self.T)) if self.T >= 4 else 4
        self.warp_gp = WarpPriorAMTGP(noise_warp=self.noise_warp_default, bound_noise_warp=self.noise_bounds, default_rho=1.0,
                                      default_omega=1.0, device=self.device)

    def _theta_to_lambdas(self, theta):   # :361-395
        rho, omg, given = _rho_omega(theta, 1.0, 1.0)
        if not given:
            return self.lambda_smooth_base, self.lambda_amp_base
        return self.lambda_smooth_base / (rho * rho + 1e-12), self.lambda_amp_base / (omg * omg + 1e-12)

    def compute_warp_batch(self, x_model, y_target_batch, y_model, theta=None, noise=None, weights=None, visualize=False,
                           verbose=False, train_iter=50):
        """:548-735.  Returns (x_warp (B,T,1), y_warp (B,T,D), lik_warp (B,), losses dict of batch-mean traces)."""
        dev = self.device
        x = _f64(np.asarray(x_model.cpu() if torch.is_tensor(x_model) else x_model, dtype=np.float64), dev).reshape(-1)
        T = int(x.numel())
        Yt = _f64(y_target_batch, dev)
        Yt = Yt.reshape(1, -1, 1) if Yt.ndim == 1 else (Yt.unsqueeze(2) if Yt.ndim == 2 else Yt)      # -> (B, T, D)
        B, Tt, D = Yt.shape
        if Tt != T:
            raise AssertionError(f"y_target_batch length mismatch: got {Tt} expected {T}")
        Ym = _f64(y_model, dev)
        Ym = Ym.unsqueeze(1) if Ym.ndim == 1 else (Ym[0] if (Ym.ndim == 3 and Ym.shape[0] == 1) else Ym)   # (T, D) or (B, T, D)
        if Ym.shape[-2] != T:
            raise AssertionError(f"y_model length mismatch: got {Ym.shape[-2]} expected {T}")
        if Ym.ndim == 3 and Ym.shape[0] != B:
            raise AssertionError(f"y_model batch mismatch: got {Ym.shape[0]} expected {B}")
        Ym = Ym[..., :D].contiguous()
        if self.T != T:
            self._set_grid(x)
        self.warp_gp.theta = theta
        n = self.noise_warp_default
        if noise is not None:
            n = float(np.clip(float(torch.as_tensor(noise, dtype=torch.float64).mean()), self.noise_bounds[0], self.noise_bounds[1]))
        lam_s, lam_a = self._theta_to_lambdas(theta)
        w = None if weights is None else torch.clamp(torch.as_tensor(weights, dtype=torch.float64).reshape(-1), min=0.0).to(dev)
        prev = self._u_ctrl_prev
        u0 = prev if (self.recursive and prev is not None and prev.numel() == self.n_ctrl) else None
        u, xw, yw, tr = ops.warp_batch(x, Yt.contiguous(), Ym, self.n_ctrl, int(train_iter), n, lam_s, lam_a, self.lr, weights=w, u0=u0)
        lik = self.warp_gp.log_sq_error_batch(x, xw)
        if self.recursive:
            self._u_ctrl_prev = u.mean(dim=0)            # :727-729
        wn = torch.ones(B, dtype=torch.float64, device=dev) if w is None else w
        mean_tr = (torch.einsum("b,bik->ik", wn, tr) / (wn.sum() + 1e-12)).cpu().numpy()
        losses = {name: list(mean_tr[:, col]) for col, name in enumerate(("loss", "data", "smooth", "amp"))}
        return xw[:, :, None], yw, lik, losses

    def compute_warp(self, x_model, y_target, y_model, theta=None, noise=None, visualize=False, verbose=False, train_iter=50):
        """One sample through the batched kernel; returns (x_warp (T,1), y_warp (T,D), lik, losses)."""
        yt = torch.as_tensor(y_target, dtype=torch.float64)
        xw, yw, lik, losses = self.compute_warp_batch(x_model, yt.reshape(1, yt.shape[0], -1), y_model, theta=theta, noise=noise,
                                                      train_iter=train_iter)
        return xw[0], yw[0], lik[0], losses
