"""Host-side mirror of WarpPriorAMTGP's GP-prior score (hdpgpc/hdpgpc/amtgp_warping_system.py:106-264, row a11).
The warp optimiser itself (Warping_system, :266-735) is outside this hot path."""
import math

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
