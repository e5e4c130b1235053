"""Host-side mirror of the SCORING half of hdpgpc/hdpgpc/GPI_model.py.

Same method names, argument meaning and error behaviour as the reference; the per-cluster lists of the
reference (f_star[i], Sigma[i], ... one entry per LDS step) are held as STACKED device tensors
([steps, T, 1] / [steps, T, T]) so that a whole batch of steps is scored by one kernel launch.  Indexing
``model.Sigma[i]`` / ``[-1]`` / ``len(model.Sigma)`` behaves like the reference's lists.

The producer of that state (the Kalman / RTS / MNIW recursion, GPI_model.py:325-406,966-1115) is outside this
hot path; state arrives through ``load_state``.
"""
import math

import numpy as np
import torch

from . import ops
from .GPI import IterativeGaussianProcess, RBFWhiteKernel

f64 = torch.float64
LOG2PI = math.log(2.0 * math.pi)   # GPI_model.py:89-90


class matrix_normal_inv_wishart:
    """GPI_model.py:1281-1298 (container) + log_likelihood_MNIW (GPI_model.py:1346-1362)."""

    def __init__(self, m_mean, m_r_cov, n0, scale):
        self.m_mean = m_mean
        self.m_r_cov = m_r_cov
        self.n0 = n0
        self.scale = scale

    def log_likelihood_MNIW(self, M, Sigma, n0=None):
        T = M.shape[0]
        dev = M.device
        eye_like = self.m_r_cov is None or bool(torch.equal(self.m_r_cov, torch.eye(T, dtype=f64, device=dev)))
        out, info = ops.mniw_loglik(M.reshape(1, T, T).contiguous(), Sigma.reshape(1, T, T).contiguous(),
                                    self.m_mean.contiguous(), None if eye_like else self.m_r_cov.contiguous(),
                                    self.scale.contiguous())
        ops.raise_on_info(info, "log_likelihood_MNIW")
        return out[0]


class GPI_model:
    def __init__(self, kernel, x_basis, annealing=True, bayesian=False, cuda=True, inducing_points=False,
                 estimation_limit=None, free_deg_MNIV=5, verbose=False):
        if not isinstance(kernel, RBFWhiteKernel):
            raise TypeError("kernel must be a hdpgpc_amd.GPI.RBFWhiteKernel")
        self.gp = IterativeGaussianProcess(kernel, x_basis, cuda=cuda, verbose=verbose)
        self.device = kernel.device
        self.x_basis = self.gp.x_basis
        self.indexes = []
        self.N = 0
        self.estimation_limit = np.inf if estimation_limit is None else estimation_limit
        self.free_deg_MNIV = free_deg_MNIV
        self.bayesian = bayesian
        self.annealing = annealing
        self.verbose = verbose
        self.f_star = self.f_star_sm = self.cov_f_sm = self.A = self.Gamma = self.C = self.Sigma = None
        self.A_def = self.Gamma_def = self.C_def = self.Sigma_def = None
        self.internal_params = None

    # ------------------------------------------------------------------ state (a12)
    def cond_to_torch(self, x):
        if x is None:
            return None
        return torch.as_tensor(x, dtype=f64).to(self.device)

    def load_state(self, f_star, Sigma, C, indexes, f_star_sm=None, cov_f_sm=None, A=None, Gamma=None, A_def=None,
                   Gamma_def=None, C_def=None, Sigma_def=None, n0=None):
        """Stacked state: f_star [S,T] or [S,T,1]; matrices [S,T,T]; indexes = member segment ids in order."""
        T = self.x_basis.shape[0]

        def vec(a):
            return None if a is None else self.cond_to_torch(a).reshape(-1, T, 1).contiguous()

        def mat(a):
            return None if a is None else self.cond_to_torch(a).reshape(-1, T, T).contiguous()

        self.f_star, self.f_star_sm = vec(f_star), vec(f_star_sm)
        self.Sigma, self.C, self.A, self.Gamma, self.cov_f_sm = mat(Sigma), mat(C), mat(A), mat(Gamma), mat(cov_f_sm)
        self.indexes = [int(i) for i in indexes]
        self.N = len(self.indexes)
        one = lambda a: None if a is None else self.cond_to_torch(a).reshape(T, T).contiguous()  # noqa: E731
        self.A_def, self.Gamma_def, self.C_def, self.Sigma_def = one(A_def), one(Gamma_def), one(C_def), one(Sigma_def)
        if n0 is not None:
            self.internal_params = matrix_normal_inv_wishart(self.A_def, None, n0, self.Gamma_def)
        return self

    # ------------------------------------------------------------------ a7: which state does step t read?
    def _select(self, t):
        """(C/Sigma index, f_star index) of GPI_model.observe with params=None, proj=False (GPI_model.py:626-669)."""
        nC, nf = len(self.C), len(self.f_star)
        if len(self.indexes) == 0:
            return 0, 0
        if len(self.indexes) <= t:
            return nC - 1, nf - 1
        if self.estimation_limit <= t:
            return nC - 1, t % nf
        ci = t if t < nC else nC - 1
        return ci % nC, t % nf

    def get_params(self, t):
        ind = t if t < len(self.C) else -1
        return self.A[ind], self.Gamma[ind], self.C[ind], self.Sigma[ind]

    def _mean_of(self, ci, fi):
        return ops.gemm_batched(self.C[ci], self.f_star[fi])

    def observe(self, x_post, t, params=None, proj=False):
        """GPI_model.py:626-662.  Returns the explicit (f_star, cov_f)."""
        if params is None:
            ci, fi = self._select(t)
            Sigma = self.Sigma[ci]
            if proj and len(self.indexes) > t:
                Sigma = Sigma + self.Gamma[ci]
            mean = self._mean_of(ci, fi)
        else:
            mean = ops.gemm_batched(self.cond_to_torch(params[2]).contiguous(), self.cond_to_torch(params[0]).reshape(-1, 1))
            Sigma = self.cond_to_torch(params[3])
        return self.gp.pred_dist(x_post, self.x_basis, mean, Sigma)

    def observe_last(self, x_post):
        """GPI_model.py:617-624."""
        mean = ops.gemm_batched(self.C[-1], self.f_star_sm[-1])
        return self.gp.pred_dist(x_post, self.x_basis, mean, self.Sigma[-1])

    def step_forward_last(self, x_post, params=None):
        """GPI_model.py:595-615."""
        if params is None:
            C, Sigma, mean = self.C[-1], self.Sigma[-1], self.f_star_sm[-1]
        else:
            mean, C, Sigma = (self.cond_to_torch(params[0]).reshape(-1, 1), self.cond_to_torch(params[2]).contiguous(),
                              self.cond_to_torch(params[3]))
        return self.gp.pred_dist(x_post, self.x_basis, ops.gemm_batched(C, mean), Sigma)

    # ------------------------------------------------------------------ a3 / a4
    def _chol_spd(self, M, jitter_scale=1e-8):
        """GPI_model.py:83-87; raises torch.linalg.LinAlgError like torch.linalg.cholesky."""
        L, info = ops.potrf_batched(self.cond_to_torch(M).contiguous(), jitter_scale, 0.0)
        ops.raise_on_info(info, "_chol_spd")
        return L[0]

    def _gaussian_score_shared_cov(self, Y, mean, cov):
        """GPI_model.py:92-113: Y (B,T,1)/(B,T), mean (T,1)/(T,), cov (T,T) -> (B,)."""
        Y = self.cond_to_torch(Y)
        Y2 = (Y[..., 0] if Y.ndim == 3 else Y).contiguous()
        B, T = Y2.shape
        items = ops.build_items([0], [0.0], [B])
        quad, _, info = ops.score_groups(Y2, self.cond_to_torch(mean).reshape(1, T).contiguous(),
                                         self.cond_to_torch(cov).reshape(1, T, T).contiguous(), *items)
        ops.raise_on_info(info, "_gaussian_score_shared_cov")
        return -0.5 * quad - 0.5 * T * LOG2PI

    # ------------------------------------------------------------------ a5
    def log_sq_error(self, x_train, y, mean=None, cov=None, C=None, Sigma=None, i=None, proj=False, first=False):
        """GPI_model.py:250-286: score of ONE segment (no log-determinant)."""
        x = self.x_basis if x_train is None else self.cond_to_torch(x_train).reshape(-1, 1)
        y = self.cond_to_torch(y).reshape(1, -1).contiguous()
        if mean is not None:
            m = ops.gemm_batched(self.cond_to_torch(C).contiguous(), self.cond_to_torch(mean).reshape(-1, 1))
            S = self.cond_to_torch(Sigma)
        elif i is not None:
            ci, fi = self._select(i)
            m, S = self._mean_of(ci, fi), self.Sigma[ci]
            if proj and len(self.indexes) > i:
                S = S + self.Gamma[ci]
        else:
            m, S = ops.gemm_batched(self.C[-1], self.f_star_sm[-1]), self.Sigma[-1]
        fn = 1e-2 * float(torch.mean(torch.diagonal(self.Sigma[0]))) if first else 0.0
        T = self.x_basis.shape[0]
        if x.shape == self.x_basis.shape and torch.equal(x, self.x_basis):
            items = ops.build_items([0], [fn], [1])
            quad, _, info = ops.score_groups(y, m.reshape(1, T).contiguous(), S.reshape(1, T, T).contiguous(), *items)
        else:
            plan = ops.PairsPlan(T, x.shape[0], [self.gp.kernel.params()], device=self.device)
            plan.update(self.x_basis.reshape(-1).contiguous(), m.reshape(1, T).contiguous(), S.reshape(1, T, T).contiguous())
            ops.raise_on_info(plan.info, "pred_dist")
            fnt = torch.full((1, 1), fn, dtype=f64, device=self.device) if first else None
            quad, _, info = plan.loglik(x.reshape(1, -1).contiguous(), y, first_noise=fnt, want_logdet=False)
            quad = quad.reshape(-1)
        ops.raise_on_info(info, "log_sq_error")
        return -0.5 * quad[0] - 0.5 * y.shape[1] * LOG2PI

    # ------------------------------------------------------------------ a6
    def _steps(self, n_samps, no_first):
        """GPI_model.py:497-513 (host logic on the index list)."""
        idx = np.asarray(self.indexes, dtype=np.int64)
        pos = np.full(n_samps, -1, dtype=np.int64)
        pos[idx] = np.arange(idx.size)
        exact = pos >= 0
        closest = np.maximum(np.searchsorted(idx, np.arange(n_samps), side="right") - 1, 0)
        i_vals = np.where(exact, pos + 1, np.maximum(closest, 1))
        first = exact & (i_vals == 1) & (not no_first)
        return i_vals, first

    def compute_sq_err_all(self, x_trains, y_trains, no_first=False):
        """GPI_model.py:488-547: score of every segment of the batch under this cluster -> (N,) on the device."""
        x_trains = self.cond_to_torch(x_trains)
        y_trains = self.cond_to_torch(y_trains)
        X = (x_trains[..., 0] if x_trains.ndim == 3 else x_trains).contiguous()
        Y = (y_trains[..., 0] if y_trains.ndim == 3 else y_trains).contiguous()
        n, Ts = X.shape
        T = self.x_basis.shape[0]
        out = torch.zeros(n, dtype=f64, device=self.device)
        if len(self.indexes) == 0:
            return out
        i_vals, first = self._steps(n, no_first)
        sel = np.array([self._select(int(t)) for t in np.unique(i_vals)])           # distinct steps -> (ci, fi)
        step_pos = np.searchsorted(np.unique(i_vals), i_vals)
        ci_seg, fi_seg = sel[step_pos, 0], sel[step_pos, 1]
        ini_noise = 1e-2 * float(torch.mean(torch.diagonal(self.Sigma[0])))
        shared = bool(torch.equal(X, X[0:1].expand_as(X)))
        on_basis = shared and Ts == T and bool(torch.equal(X[0], self.x_basis.reshape(-1)))
        if on_basis:
            # one Cholesky per (step, first) group, every member of the group solved against it (GPI_model.py:519-533)
            code = i_vals * 2 + first
            order = np.argsort(code, kind="stable")
            codes, start, counts = np.unique(code[order], return_index=True, return_counts=True)
            rep = order[start]
            g_ci, g_fi = ci_seg[rep], fi_seg[rep]
            pairs, inv = np.unique(np.stack([g_ci, g_fi], 1), axis=0, return_inverse=True)
            means = ops.gemm_batched(self.C[torch.as_tensor(pairs[:, 0], device=self.device)],
                                     self.f_star[torch.as_tensor(pairs[:, 1], device=self.device)]).reshape(-1, T)
            adds = np.where(codes % 2 == 1, ini_noise, 0.0)
            inv = inv.reshape(-1)
            quad = torch.zeros(n, dtype=f64, device=self.device)
            multi = counts > 1
            if multi.any():   # several segments share one (step, first) state: one factorisation, many right-hand sides
                gm = np.nonzero(multi)[0]
                seg_list = np.concatenate([order[start[g]:start[g] + counts[g]] for g in gm]).astype(np.int32)
                im, ia, io, ic = ops.build_items(g_ci[gm].tolist(), adds[gm].tolist(), counts[gm].tolist())
                grp_of_item = np.repeat(np.arange(len(gm)), [-(-c // ops.MAX_CHUNK) for c in counts[gm]])
                quad, _, info = ops.score_groups(Y, means.contiguous(), self.Sigma, im, ia, io, ic, seg_ids=seg_list,
                                                 item_mean=inv[gm][grp_of_item].astype(np.int32))
                ops.raise_on_info(info, "compute_sq_err_all")
            single = ~multi
            if single.any():  # member segments: each has its own Sigma_i and a single right-hand side
                gs = np.nonzero(single)[0]
                segs = torch.as_tensor(rep[gs], device=self.device)
                q1, _, info1 = ops.score_each(Y[segs].contiguous(), means.contiguous(), self.Sigma, g_ci[gs].astype(np.int32),
                                              inv[gs].astype(np.int32), adds[gs])
                ops.raise_on_info(info1, "compute_sq_err_all")
                quad[segs] = q1
            return -0.5 * quad - 0.5 * T * LOG2PI
        # general path (GPI_model.py:535-545): every segment against the state of ITS step, on its own grid
        pairs, col = np.unique(np.stack([ci_seg, fi_seg], 1), axis=0, return_inverse=True)
        col = col.reshape(-1)
        means = ops.gemm_batched(self.C[torch.as_tensor(pairs[:, 0], device=self.device)],
                                 self.f_star[torch.as_tensor(pairs[:, 1], device=self.device)]).reshape(-1, T)
        Sig = self.Sigma[torch.as_tensor(pairs[:, 0], device=self.device)].contiguous()
        plan = ops.PairsPlan(T, Ts, np.repeat(np.asarray(self.gp.kernel.params())[None], len(pairs), 0), device=self.device)
        plan.update(self.x_basis.reshape(-1).contiguous(), means.contiguous(), Sig)
        ops.raise_on_info(plan.info, "pred_dist")
        fn = torch.as_tensor(np.where(first, ini_noise, 0.0), dtype=f64, device=self.device)
        score, info = plan.score(X, Y, first_noise=fn, sel=col.astype(np.int32))
        ops.raise_on_info(info, "compute_sq_err_all")
        return score

    # ------------------------------------------------------------------ a8
    def _lat_indices(self):
        nG = len(self.Gamma)
        cur, prev, par, cov = [], [], [], []
        for j in range(len(self.indexes)):
            if j == 0:
                prev.append(1), cov.append(1), par.append(nG - 1)
            else:
                prev.append(j), cov.append(j), par.append(j + 1 if j + 1 < nG else nG - 1)
            cur.append(j + 1)
        return cur, prev, par, cov

    def _lat_all(self, h_ini=1.0, only=None):
        cur, prev, par, cov = self._lat_indices()
        if only is not None:
            cur, prev, par, cov = [cur[only]], [prev[only]], [par[only]], [cov[only]]
        T = self.x_basis.shape[0]
        ix = lambda a: torch.as_tensor(a, device=self.device)  # noqa: E731
        Gam = self.Gamma[ix(par)].clone()
        if only is None or only == 0:
            Gam[0] = Gam[0] * h_ini                                  # GPI_model.py:293
        out, info = ops.lat_error(self.f_star_sm[ix(cur)].reshape(-1, T).contiguous(),
                                  self.f_star_sm[ix(prev)].reshape(-1, T).contiguous(), self.A[ix(par)].contiguous(), Gam,
                                  self.cov_f_sm[ix(cov)].contiguous())
        ops.raise_on_info(info, "log_lat_error")
        return out - 0.5 * T * LOG2PI

    def log_lat_error(self, i, h_ini):
        """GPI_model.py:288-323."""
        return self._lat_all(h_ini, only=i)[0]

    def compute_q_lat_all(self, x_trains, h_ini=1.0):
        """GPI_model.py:549-559: all members of the cluster in one batch."""
        n = x_trains.shape[0]
        out = torch.zeros(n, dtype=f64, device=self.device)
        if self.N == 0 or not bool(torch.any(self.Gamma[-1] != 0)):
            return out
        out[torch.as_tensor(self.indexes, device=self.device)] = self._lat_all(h_ini)
        return out

    # ------------------------------------------------------------------ a9
    def return_LDS_param_likelihood(self, first=False):
        """GPI_model.py:459-486."""
        T = self.x_basis.shape[0]
        A_, Gam_, C_, Sig_ = self.A[-1], self.Gamma[-1], self.C[-1], self.Sigma[-1]
        if first:
            eye = torch.eye(T, dtype=f64, device=self.device)
            Gam_ = Gam_ + 2.0 * torch.mean(torch.diagonal(Gam_)) * eye
            Sig_ = Sig_ + 2.0 * torch.mean(torch.diagonal(Sig_)) * eye
        Ms, Ss, means, scales = [C_], [Sig_], [self.C_def], [self.Sigma_def]
        if bool(torch.any(self.Gamma_def != 0)):
            Ms.append(A_), Ss.append(Gam_), means.append(self.A_def), scales.append(self.Gamma_def)
        out, info = ops.mniw_loglik(torch.stack(Ms).contiguous(), torch.stack(Ss).contiguous(), torch.stack(means).contiguous(),
                                    None, torch.stack(scales).contiguous())
        ops.raise_on_info(info, "return_LDS_param_likelihood")
        return torch.sum(out) / T * 100.0
