"""Host-side mirror of the SCORING half of hdpgpc/hdpgpc/GPI_model.py.

Same method names, argument meaning and error behaviour as the reference; the per-cluster lists of the
reference (f_star[i], Sigma[i], ... one entry per LDS step) are held as STACKED device tensors
([steps, T, 1] / [steps, T, T]) so that a whole batch of steps is scored by one kernel launch.  Indexing
``model.Sigma[i]`` / ``[-1]`` / ``len(model.Sigma)`` behaves like the reference's lists.

State arrives either through ``load_state`` (arrays captured elsewhere) or from the producer half below
(first "next" row): ``full_pass_weighted`` = per member Kalman update (GPI.py:72-151), two-step RTS
(GPI.py:272-300), MNIW conjugate update (GPI_model.py:966-1115,1300-1344), then the full RTS pass (GPI.py:240-270).
It is a composition of the batched MFMA GEMM and Cholesky-inverse kernels, one launch per matrix operation - the
persistent chain kernel of SURVEY.md H5 is the next step.  Every linear solve of the reference (LU ``torch.linalg.solve``
/ ``inv`` of symmetric positive-definite matrices) is done through the Cholesky inverse; covered cases: dynamic
model, shared grid, h = 1 (what hdpgpc/tests run with warp=False).  The gpytorch hyper-parameter fit is NOT built:
``fit_kernel_params`` takes theta from ``GPI_model.fixed_theta`` (parity-unpinned piece).
"""
import math
from collections.abc import MutableSequence

import numpy as np
import torch

from . import ops
from .GPI import IterativeGaussianProcess, RBFWhiteKernel

f64 = torch.float64
LOG2PI = math.log(2.0 * math.pi)   # GPI_model.py:89-90


class StackList(MutableSequence):
    """One per-step list of a model (f_star[i], Sigma[i], ... one entry per LDS step in the reference, a12) held as ONE stacked
    device tensor [L, ...].  Reads behave like the reference's Python list of tensors (len, [i], [-1], slices, iteration,
    ``+``); the first mutation (item assignment, append) turns it into a real list (copy-on-write), so a model and its
    copies can share one.  A 2 272-step chain is 8 such lists: building 18 000 tensor objects per pass, and re-stacking
    them for every batched kernel, was 10 % of the offline loop's wall-clock.  It is a collections.abc.MutableSequence (pop,
    extend, insert, del, reversed, ``in`` come with it); torch.stack / torch.cat only take real lists and tuples: pass
    ``list(m.f_star)`` or use ``m.f_star.stack()``."""
    __slots__ = ("_st", "_ls")

    def __init__(self, stack):
        self._st, self._ls = stack, None

    def _list(self):
        if self._ls is None:
            self._ls, self._st = list(self._st.unbind(0)), None
        return self._ls

    def stack(self):
        return self._st if self._ls is None else torch.stack(self._ls).contiguous()

    def __len__(self):
        return self._st.shape[0] if self._ls is None else len(self._ls)

    def __getitem__(self, i):
        if self._ls is not None:
            return self._ls[i]
        if isinstance(i, slice):
            return list(self._st[i].unbind(0))
        return self._st[i]

    def __setitem__(self, i, v):
        self._list()[i] = v

    def __delitem__(self, i):
        del self._list()[i]

    def insert(self, i, v):
        self._list().insert(i, v)

    def append(self, v):
        self._list().append(v)

    def __radd__(self, other):
        return list(other) + list(self)

    def __iter__(self):
        return iter(self._st.unbind(0) if self._ls is None else self._ls)

    def __add__(self, other):
        return list(self) + list(other)

    def copy(self):
        return StackList(self._st) if self._ls is None else list(self._ls)


def _copy_list(lst):
    return lst.copy() if isinstance(lst, StackList) else list(lst)


class matrix_normal_inv_wishart:
    """GPI_model.py:1281-1298 (container) + log_likelihood_MNIW (GPI_model.py:1346-1362)."""

    def __init__(self, m_mean, m_r_cov, n0, scale):
        self.m_mean = m_mean
        self.m_r_cov = m_r_cov
        self.n0 = n0
        self.scale = scale

    def get_mean(self):
        return self.m_mean

    def get_scale(self, final=False):
        return self.scale if final else self.scale * self.n0 / (self.n0 - 2)

    def set_scale(self, scale):
        self.scale = scale

    def posterior(self, n_k, y1, y2, cov=None, cov_=None, cov_cross=None, sse_matrix=None, annealing=False, defer=None):
        """GPI_model.py:1300-1344 for n_k = 1, zero covariance corrections and no projection (the only call the
        one-step estimation makes on a shared grid, GPI_model.py:995-998,1034-1036)."""
        if n_k != 1 or sse_matrix is not None:
            raise NotImplementedError("MNIW posterior: only the one-step, shared-grid update is built")
        T = self.scale.shape[0]
        dev = self.scale.device
        eye = torch.eye(T, dtype=f64, device=dev)
        new_n0 = self.n0 + n_k
        jitter = 1e-2 * torch.clamp_min(torch.mean(torch.diagonal(self.scale).abs()), np.finfo(np.float64).eps)
        m_r_cov = eye if self.m_r_cov is None else self.m_r_cov
        Z, info = ops.chol_inverse((0.5 * (m_r_cov + m_r_cov.T) + jitter * eye).contiguous())   # :1313-1316
        (defer.append(info) if defer is not None else ops.raise_on_info(info, "MNIW.posterior"))
        scale_inv = ops.gemm_batched(Z[0], Z[0], transA=True)
        y1, y2 = y1.reshape(T, 1), y2.reshape(T, 1)
        S__ = ops.gemm_batched(y2, y2, transB=True) + scale_inv                                 # :1321,1325
        S_ = ops.gemm_batched(y1, y2, transB=True) + ops.gemm_batched(self.m_mean.contiguous(), scale_inv)
        Zs, info = ops.chol_inverse(S__.contiguous(), 0.0, 1e-8)                                # :1329
        (defer.append(info) if defer is not None else ops.raise_on_info(info, "MNIW.posterior"))
        part_mean = ops.gemm_batched(ops.gemm_batched(S_.contiguous(), Zs[0], transB=True), Zs[0])   # S_ S__^{-1}
        new_m_mean = ((self.n0 - 2) * self.m_mean + part_mean) / (new_n0 - 2)                   # :1332-1336
        e = y1 - y2
        new_scale = ((self.n0 - 2) * self.scale + ops.gemm_batched(e, e, transB=True)) / (new_n0 - 2)
        return matrix_normal_inv_wishart(new_m_mean, S__, new_n0, new_scale)

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
        self.f_star, self.f_star_sm, self.cov_f, self.cov_f_sm = [], [], [], []
        self.A, self.Gamma, self.C, self.Sigma = [], [], [], []
        self.x_train, self.y_train = [], []
        self.A_def = self.Gamma_def = self.C_def = self.Sigma_def = None
        self.ini_cov_def = None
        self.internal_params = self.observation_params = None
        self.fitted = False
        self.fixed_theta = None        # (c, ell, noise) taken by fit_kernel_params instead of the gpytorch fit
        self.theta_source = None       # ... or an owner (GPI_HDP) whose `fixed_theta` is looked up at fit time
        # BASELINE configs[4]: carry chol(scale) of the observation MNIW through its rank-1 recursion (GPI_model.py:1332-1336)
        # instead of refactoring Sigma at every score.  Only exact when Sigma is a multiple of the scale, i.e. annealing off
        # (the annealed Sigma_i adds a time-varying diagonal shift); off by default, see rank1_scale_factor below.
        self.rank1_scoring = False
        self._Lobs = None              # (observation_params object the factor belongs to, lower Cholesky factor of its scale)
        self.noise_bounds = (1e-10, 1e10)
        self._stk = {}
        self._pending = []
        self._defer_checks = False

    # ------------------------------------------------------------------ pickling (GPI_HDP.save_swgp)
    _TENSOR_LISTS = ("f_star", "f_star_sm", "cov_f", "cov_f_sm", "A", "Gamma", "C", "Sigma", "x_train", "y_train")

    def __getstate__(self):
        """Host copies of the state; caches, pending status words and the online pool's slot are dropped."""
        d = {k: v for k, v in self.__dict__.items() if k not in ("_stk", "_pending", "_graph_keepalive", "_slot", "_dyn", "theta_source",
                                                                  "_Lobs", "_def_diag_key", "_def_diag", "_dyn_def")}
        cpu = lambda t: t.detach().cpu() if torch.is_tensor(t) else t           # noqa: E731
        for name in self._TENSOR_LISTS:
            d[name] = [cpu(t) for t in getattr(self, name)]
        for name in ("A_def", "Gamma_def", "C_def", "Sigma_def", "ini_cov_def", "x_basis"):
            d[name] = cpu(d.get(name))
        for name in ("internal_params", "observation_params"):
            m = d.get(name)
            if m is not None:
                d[name] = matrix_normal_inv_wishart(cpu(m.m_mean), cpu(m.m_r_cov), m.n0, cpu(m.scale))
        d["theta_owner_fixed"] = getattr(self.theta_source, "fixed_theta", None)
        return d

    def __setstate__(self, d):
        fixed = d.pop("theta_owner_fixed", None)
        self.__dict__.update(d)
        self._stk, self._pending, self._Lobs, self.theta_source = {}, [], None, None
        if self.fixed_theta is None:
            self.fixed_theta = fixed

    def _set_device(self, dev):
        """Move the (unpickled, host-resident) state to `dev`."""
        mv = lambda t: t.to(dev) if torch.is_tensor(t) else t                   # noqa: E731
        self.device = dev
        self.gp.device = self.gp.kernel.device = dev
        self.gp.x_basis = mv(self.gp.x_basis)
        for name in self._TENSOR_LISTS:
            setattr(self, name, [mv(t) for t in getattr(self, name)])
        for name in ("A_def", "Gamma_def", "C_def", "Sigma_def", "ini_cov_def", "x_basis"):
            setattr(self, name, mv(getattr(self, name)))
        for name in ("internal_params", "observation_params"):
            m = getattr(self, name)
            if m is not None:
                setattr(self, name, matrix_normal_inv_wishart(mv(m.m_mean), mv(m.m_r_cov), m.n0, mv(m.scale)))
        self._stk = {}

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

        unb = lambda t: [] if t is None else list(t.unbind(0))  # noqa: E731
        self.f_star, self.f_star_sm = unb(vec(f_star)), unb(vec(f_star_sm))
        self.Sigma, self.C, self.A, self.Gamma, self.cov_f_sm = (unb(mat(Sigma)), unb(mat(C)), unb(mat(A)), unb(mat(Gamma)),
                                                                 unb(mat(cov_f_sm)))
        self._stk = {}
        self.indexes = [int(i) for i in indexes]
        self.N = len(self.indexes)
        one = lambda a: None if a is None else self.cond_to_torch(a).reshape(T, T).contiguous()  # noqa: E731
        self.A_def, self.Gamma_def, self.C_def, self.Sigma_def = one(A_def), one(Gamma_def), one(C_def), one(Sigma_def)
        if n0 is not None:
            self.internal_params = matrix_normal_inv_wishart(self.A_def, None, n0, self.Gamma_def)
        return self

    def _S(self, name):
        """Stacked [steps, ...] device tensor of one of the per-step lists (cached until the list changes; every method
        that rewrites an entry in place resets ``_stk``)."""
        lst = getattr(self, name)
        key = self._stk.get(name)
        if isinstance(lst, StackList) and lst._ls is None:          # already one tensor: no copy
            if key is None or key[2] is not lst._st:
                self._stk[name] = [len(lst), None, lst._st, None]
            return lst._st
        if key is None or key[0] != len(lst) or key[1] is not lst[-1]:
            self._stk[name] = [len(lst), lst[-1], torch.stack(list(lst)).contiguous(), None]
        return self._stk[name][2]

    def _S_symmetric(self, name):
        """Is every matrix of the stack equal to its transpose bit for bit?  One device check per stack, stored WITH the
        stack (it dies with it)."""
        self._S(name)
        ent = self._stk[name]
        if ent[3] is None:
            ent[3] = bool(torch.equal(ent[2], ent[2].transpose(1, 2)))
        return ent[3]

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

    # ------------------------------------------------------------------ 8f-1: the producer of the state
    def _eye(self):
        T = self.x_basis.shape[0]
        return torch.eye(T, dtype=f64, device=self.device)

    def compute_mean(self):
        return torch.zeros((self.x_basis.shape[0], 1), dtype=f64, device=self.device)

    def GPR_dynamic(self, gamma=None, sigma=None):
        """GPI_model.py:191-204."""
        eye = self._eye()
        return eye.clone(), (0.01 if gamma is None else gamma) * eye, eye.clone(), (0.25 if sigma is None else sigma) * eye

    def GPR_static(self, ini_Sigma=None):
        """GPI_model.py:178-189."""
        eye = self._eye()
        return eye.clone(), torch.zeros_like(eye), eye.clone(), (0.25 if ini_Sigma is None else ini_Sigma) * eye

    def initial_conditions(self, ini_mean=None, ini_cov=None, ini_A=None, ini_Gamma=None, ini_C=None, ini_Sigma=None):
        """GPI_model.py:115-175."""
        K = self.gp.kernel(self.x_basis, self.x_basis)
        m0 = self.compute_mean() if ini_mean is None else self.cond_to_torch(ini_mean).reshape(-1, 1)
        c0 = K if ini_cov is None else self.cond_to_torch(ini_cov)
        self.f_star, self.f_star_sm = [m0], [m0]
        self.cov_f, self.cov_f_sm = [c0], [c0]
        self.ini_cov_def = c0.clone()
        if ini_A is None and ini_Gamma is None and ini_C is None and ini_Sigma is None:
            ini_A, ini_Gamma, ini_C, ini_Sigma = self.GPR_dynamic()
        ini_A, ini_Gamma, ini_C, ini_Sigma = (self.cond_to_torch(m).contiguous() for m in (ini_A, ini_Gamma, ini_C, ini_Sigma))
        self.A, self.Gamma, self.C, self.Sigma = [ini_A], [ini_Gamma], [ini_C], [ini_Sigma]
        self.A_def, self.Gamma_def, self.C_def, self.Sigma_def = ini_A, ini_Gamma, ini_C, ini_Sigma
        self.indexes, self.N, self.x_train, self.y_train, self._stk = [], 0, [], [], {}
        self.internal_params = matrix_normal_inv_wishart(ini_A, None, self.free_deg_MNIV, ini_Gamma)
        self.observation_params = matrix_normal_inv_wishart(ini_C, None, self.free_deg_MNIV, ini_Sigma)

    def fit_kernel_params(self, x_train, y, alpha_ini, gamma_ini, valid=True):
        """GPI_model.py:207-241.  The gpytorch fit (GPI.py:610-770) is either replaced by ``self.fixed_theta`` (what the
        golden fixtures inject) or restated by kernel_fit.fit_kernel_adam: what the fit leaves behind is outputscale, a
        length-scale forced to 1.2 (GPI.py:711) and a noise level inside its bounds - and none of them enters Sigma, which
        is reset to the INITIAL sigma (GPI_model.py:215-219)."""
        if valid:
            lo, hi = self.noise_bounds
            fixed = self.fixed_theta if self.fixed_theta is not None else getattr(self.theta_source, "fixed_theta", None)
            if fixed is None:                 # Adam on the exact MLL of this first member (kernel_fit.py)
                from .kernel_fit import fit_kernel_adam
                c, _, noise, _ = fit_kernel_adam(self.x_basis.reshape(-1).cpu().numpy(), self.cond_to_torch(y).reshape(-1).cpu().numpy(),
                                                 (lo, hi), device=self.device)
                ell = 1.2                     # GPI.py:711 overwrites the fitted length-scale
            else:
                c, ell, noise = fixed
            # written through log-parameters, as the reference does (GPI.py:707-714: `kernel.k1.k1.theta = np.log([...])`): the
            # stored value is exp(log(v)), and it is THAT value whose Gram matrix GPI.py:136 later compares bit for bit with
            # the prior covariance of a copied model (gpmodel_deepcopy clones the kernel through theta again)
            self.gp.kernel.theta = np.log(np.array([float(c), float(ell), float(min(max(noise, lo), hi))]))
            self.gp.fitted = True
        eye = self._eye()
        alph = alpha_ini[0][0]
        self.Sigma[-1] = alph * eye
        self.Sigma_def = self.Sigma[-1].clone()
        self.C[-1], self.A[-1] = eye.clone(), eye.clone()
        self.Gamma[-1] = torch.mean(torch.diagonal(self.Gamma[-1])) * eye
        self.f_star[-1] = self.f_star_sm[-1] = self.compute_mean()
        ini_cov = self.gp.kernel(self.x_basis, self.x_basis)
        self.ini_cov_def = ini_cov
        self.cov_f[-1] = self.cov_f_sm[-1] = ini_cov
        self.observation_params.set_scale(alph * eye)
        self.observation_params.m_mean = self.C[-1]
        self.internal_params.set_scale(self.Gamma[-1])
        self.internal_params.m_mean = self.A[-1]
        self.fitted = True
        self._stk = {}
        return self.x_basis, ini_cov

    def _spd_inv(self, S, what):
        """S^{-1} of a symmetric positive-definite matrix through the Cholesky inverse: Z^T Z with Z = chol(S)^{-1}.
        The LAPACK info is collected and checked once per pass (one host sync instead of one per solve)."""
        Z, info = ops.chol_inverse(S.contiguous())
        self._pending.append((what, info))
        return ops.gemm_batched(Z[0], Z[0], transA=True)

    def _raise(self, info, what):
        """raise_on_info now, or - with _defer_checks set (the online step's would-be new cluster: five small evaluations in a row,
        each of which would wait for the device) - at the caller's next _check_pending()."""
        if self._defer_checks:
            self._pending.append((what, info))
        else:
            ops.raise_on_info(info, what)

    def _check_pending(self):
        """One host sync for all the LAPACK infos collected since the last check; raises like torch.linalg.cholesky."""
        if self._pending:
            pending, self._pending = self._pending, []
            infos = torch.cat([i.reshape(-1) for _, i in pending])
            if bool(infos.any()):
                flat = int(torch.nonzero(infos)[0, 0])
                sizes = np.cumsum([i.numel() for _, i in pending])
                what = pending[int(np.searchsorted(sizes, flat, side="right"))][0]      # flat index -> its entry
                raise torch.linalg.LinAlgError(f"{what}: the input is not positive-definite")

    def _posterior(self, mean_prior, cov_prior, y, A, Gamma, C, Sigma, first_step, h=1.0, x_warped=None):
        """GPI.posterior (GPI.py:72-151).  x_warped None / equal to the basis: K_cov = I (shared grid); otherwise the
        observation sits on its own grid and K_cov = K(x, xb) (K(xb, xb) + 1e-4 I)^{-1} interpolates the basis state onto it
        (GPI.py:124-133), f*, cov_f come from pred_dist (GPI.py:141-142)."""
        mm = ops.gemm_batched
        xm = mm(A, mean_prior)
        on_basis = x_warped is None or (x_warped.shape == self.x_basis.shape and bool(torch.equal(x_warped, self.x_basis)))
        k = self.gp.kernel
        if first_step:   # cov_prior is the kernel Gram itself (GPI.py:136-139): prior predictive, white-noise observation
            P = cov_prior
            n_obs = self.x_basis.shape[0] if on_basis else x_warped.shape[0]
            f_star = torch.zeros((n_obs, 1), dtype=f64, device=self.device)
            # kernel(x) - kernel(x, x): the white-noise level as the reference's subtraction leaves it
            cov_f = (((k.constant_value + k.noise_level) - k.constant_value) / h) * torch.eye(n_obs, dtype=f64, device=self.device)
        else:
            P = mm(mm(A, cov_prior), A, transB=True) + Gamma
            if on_basis:
                f_star, cov_f = mm(C, xm), Sigma              # pred_dist short-circuits on the shared grid (GPI.py:467-468)
            else:
                f_star, cov_f = self.gp.pred_dist(x_warped, self.x_basis, mm(C, xm), Sigma)
        if on_basis:
            KC = C
        else:
            T = self.x_basis.shape[0]
            Kxx = k(self.x_basis, self.x_basis) + 1e-4 * self._eye()
            Kinv = self._spd_inv(Kxx, "posterior (K_cov)")
            KC = mm(mm(k(x_warped, self.x_basis), Kinv), C)    # K_cov C, [T*, T]
        S = mm(mm(KC, P), KC, transB=True) + cov_f
        K_t = mm(mm(P, KC, transB=True), self._spd_inv(S, "posterior"))   # P KC^T S^{-1} (GPI.py:144-145); the inverse symmetrises on load
        mean_post = xm + mm(K_t, y - f_star)
        IKC = self._eye() - mm(K_t, KC)
        cov_post = mm(mm(IKC, P), IKC, transB=True) + mm(mm(K_t, cov_f.contiguous()), K_t, transB=True)   # Joseph form
        return mean_post, cov_post

    def posterior_weighted(self, x_train, y, h, t=None):
        """GPI_model.py:561-582: the filtered state the model would have after absorbing (x, y) with responsibility h -
        from the last FILTERED state (or the one of step t), with Gamma / h and Sigma / h."""
        y = self.cond_to_torch(y).reshape(-1, 1)
        x = self.cond_to_torch(x_train).reshape(-1, 1)
        if not h > 0.0:
            return self.f_star[-1].clone(), self.cov_f[-1].clone()
        if t is not None and len(self.indexes) > t:
            f, c = self.f_star[t], self.cov_f[t]
            A, Gamma, C, Sigma = self.get_params(t)
        else:
            f, c, A, Gamma, C, Sigma = self.f_star[-1], self.cov_f[-1], self.A[-1], self.Gamma[-1], self.C[-1], self.Sigma[-1]
        first = bool(torch.equal(c, self.gp.kernel(self.x_basis, self.x_basis)))
        out = self._posterior(f, c, y, A, Gamma / h, C, Sigma / h, first, h=h, x_warped=x)
        if not self._defer_checks:
            self._check_pending()
        return out

    def find_closest_lower(self, t):
        """GPI_model.py:584-593: position of the last member at or before segment t (the index list is sorted)."""
        idx = int(np.searchsorted(np.asarray(self.indexes, dtype=np.int64), t, side="right"))
        return idx - 1 if idx else 0

    def smoother_weighted(self, x_train, y, h):
        """GPI_model.py:726-738: the state lists as they would be with the sample added (nothing is stored)."""
        f_aux, c_aux = self.posterior_weighted(x_train, y, h)
        return self.f_star + [f_aux], self.cov_f + [c_aux], self.C + [self.C[-1]], self.Sigma + [self.Sigma[-1]]

    def smoother_weighted_index(self, x_train, y, h, t):
        """GPI_model.py:740-745."""
        f_aux, c_aux = self.posterior_weighted(x_train, y, h, t)
        _, _, C, Sigma = self.get_params(t)
        return f_aux, c_aux, C, Sigma

    def estimate_new_and_include(self, index, x_train, y):
        """GPI_HDP.estimate_new (GPI_HDP.py:2830-2842) followed by include_weighted_sample(h = 1) of the SAME segment, as the
        online step does for its would-be new cluster (GPI_HDP.py:1990-1996).  The reference runs the Kalman update twice - from
        the last filtered state for the score, from the last smoothed state for the inclusion; on a model without history
        (after reinit_GP) both start from the same prior and take the same first-step branch, so one update serves both.
        Returns the score."""
        x = self.cond_to_torch(x_train).reshape(-1, 1)
        fresh = (self.N == 0 and self.fitted and len(self.f_star) == 1 and self.f_star[0] is self.f_star_sm[0]
                 and x.shape == self.x_basis.shape and bool(torch.equal(x, self.x_basis)))
        if not fresh:
            mean_, cov_, C_, Sigma_ = self.smoother_weighted(x, y, 1.0)
            score = self.log_sq_error(x, y, mean=mean_[-1], cov=cov_[-1], C=C_[-1], Sigma=Sigma_[-1], i=-1, first=len(self.indexes) == 1)
            self.include_weighted_sample(index, x, x, y, 1.0)
            return score
        f, c = self.posterior_weighted(x, y, 1.0)
        score = self.log_sq_error(x, y, mean=f, cov=c, C=self.C[-1], Sigma=self.Sigma[-1], i=-1, first=len(self.indexes) == 1)
        self.N += 1
        self.indexes.append(int(index))
        self.x_train.append(x)
        self.y_train.append(self.cond_to_torch(y).reshape(-1, 1))
        self.f_star.append(f), self.f_star_sm.append(f), self.cov_f.append(c), self.cov_f_sm.append(c)
        return score

    def reinit_GP(self, save_last=False, save_index=False):
        """GPI_model.py:408-434: drop the filtered / smoothed history (keep the first entry, or first and last)."""
        if save_last:
            self.f_star = [self.f_star[0], self.f_star[-1]]
            self.f_star_sm = [self.f_star[0].clone(), self.f_star[-1].clone()]
            self.cov_f = [self.cov_f[0], self.cov_f[-1]]
            self.cov_f_sm = [self.cov_f_sm[0], self.cov_f_sm[-1]]
            if not save_index:
                self.indexes = [0]
        else:
            self.f_star = self.f_star[:1]
            self.f_star_sm = list(self.f_star)
            self.cov_f = [self.ini_cov_def.clone()]
            self.cov_f_sm = [self.ini_cov_def.clone()]
            self.indexes = []
        self.y_train, self.x_train = [], []
        self.N = 0
        self._stk = {}

    def reinit_LDS(self, save_last=False, save_last_diag=False):
        """GPI_model.py:437-457: back to the default LDS parameters (or keep the last ones) and fresh MNIW priors."""
        if save_last:
            if save_last_diag:
                ini = (self.A_def, torch.diag(torch.diagonal(self.Gamma[-1])) * 3.0, self.C_def, torch.diag(torch.diagonal(self.Sigma[-1])) * 3.0)
            else:
                ini = (self.A[-1], self.Gamma[-1], self.C[-1], self.Sigma[-1])
        else:
            ini = (self.A_def, self.Gamma_def, self.C_def, self.Sigma_def)
        self.A, self.Gamma, self.C, self.Sigma = [ini[0]], [ini[1]], [ini[2]], [ini[3]]
        eye = self._eye()
        self.internal_params = matrix_normal_inv_wishart(ini[0], eye, self.free_deg_MNIV, ini[1])
        self.observation_params = matrix_normal_inv_wishart(ini[2], eye, self.free_deg_MNIV, ini[3])
        self._stk = {}

    def include_sample(self, index, x_train, y, x_warped=None, h=1.0, posterior=True, embedding=True, include_index=False):
        """GPI_model.py:325-351."""
        y = self.cond_to_torch(y).reshape(-1, 1)
        if posterior:
            self.N += 1
            self.indexes.append(int(index))
            self.x_train.append(x_train)
            self.y_train.append(y)
            # GPI.py:136 tests cov_prior == ker(xb, xb) on every call (two Gram builds per step); it can only hold for
            # the first member after the kernel fit, so it is only evaluated there
            first = self.N == 1 and bool(torch.equal(self.cov_f_sm[-1], self.gp.kernel(self.x_basis, self.x_basis)))
            xw = None if x_warped is None else self.cond_to_torch(x_warped).reshape(-1, 1)
            f, c = self._posterior(self.f_star_sm[-1], self.cov_f_sm[-1], y, self.A[-1], self.Gamma[-1], self.C[-1],
                                   self.Sigma[-1] / h, first, h, x_warped=xw)
        elif include_index:
            self.indexes.append(int(index))
            self.x_train.append(x_train)
            self.y_train.append(y)
            f, c = self.f_star_sm[-1], self.cov_f_sm[-1]
        else:
            return self.f_star_sm[-1], self.cov_f_sm[-1]
        self.f_star.append(f), self.f_star_sm.append(f), self.cov_f.append(c), self.cov_f_sm.append(c)
        return f, c

    def include_weighted_sample(self, index, x_train, x_warped, y, h, snr=None):
        """GPI_model.py:353-375 (h = 1 or h < 1; snr gating not built)."""
        if snr is not None:
            raise NotImplementedError("snr-gated inclusion (multi-lead) is not part of this path")
        x_train = self.cond_to_torch(x_train).reshape(-1, 1)
        if h == 1.0:
            if self.N == 0 and not self.fitted:
                self.fit_kernel_params(x_train, y, self.Sigma[-1], self.Gamma[-1], valid=True)
            self.include_sample(index, x_train, y, x_warped, h=1.0)
        else:
            self.include_sample(index, x_train, y, x_warped, posterior=False)
        return self.x_basis

    def backwards_pair(self, h, snr=None):
        """GPI_model.py:705-716 + GPI.backward_notrange (GPI.py:272-300): smooth the last two filtered states."""
        if len(self.indexes) > 1 and h == 1.0:
            mm = ops.gemm_batched
            A, Gam = self.A[-1], self.Gamma[-1]
            m0, m1, c0, c1 = self.f_star[-2], self.f_star[-1], self.cov_f[-2], self.cov_f[-1]
            P = mm(mm(A, c0), A, transB=True) + Gam
            J = mm(mm(c0, A, transB=True), self._spd_inv(P, "backwards_pair"))     # c0 A^T P^{-1}; the inverse symmetrises on load
            self.f_star_sm[-2] = m0 + mm(J, m1 - mm(A, m0))
            self.cov_f_sm[-2] = c0 + mm(mm(J, c1 - P), J, transB=True)
            self.f_star_sm[-1], self.cov_f_sm[-1] = m1, c1
            self._stk = {}

    def bayesian_new_params(self, h, model_type="dynamic", full_data=False, q=None, force=False, snr=1.0):
        """GPI_model.py:966-1115, one-step estimation (full_data=False), dynamic model, shared grid."""
        if h != 1.0:          # the reference's whole body sits under `if h == 1.0:` (GPI_model.py:972): a soft member is skipped
            return
        if full_data:
            raise NotImplementedError("bayesian_new_params: only the one-step update (full_data=False) is built")
        self._stk = {}
        if 1 < self.N < self.estimation_limit or force:
            infos = []
            new_int = self.internal_params.posterior(1, self.f_star_sm[-1], self.f_star_sm[-2], defer=infos)
            new_obs = self.observation_params.posterior(1, self.y_train[-1], self.f_star_sm[-1], defer=infos)
            if bool(torch.cat(infos).any()):             # GPI_model.py:1068-1071: keep the previous distributions
                new_int, new_obs = self.internal_params, self.observation_params
            elif self._rank1_on():
                self._rank1_advance(new_obs)
        else:
            new_int, new_obs = self.internal_params, self.observation_params
        self.internal_params, self.observation_params = new_int, new_obs
        if 1 < self.N:
            Gamma_, Sigma_ = new_int.get_scale(), new_obs.get_scale()
        else:
            Gamma_, Sigma_ = self.Gamma[-1], self.Sigma[-1]
        if self.annealing:                               # GPI_model.py:1083-1091
            Gamma_ = Gamma_ + self.Gamma[0] / (self.N ** 2)
            Sigma_ = Sigma_ + self.Sigma[0] / (self.N ** 2)
        if self.N < self.estimation_limit:
            self.A.append(new_int.get_mean())
            self.Gamma.append(Gamma_)
            self.C.append(new_obs.get_mean())
            self.Sigma.append(Sigma_)

    def _rank1_on(self):
        return (self.rank1_scoring or bool(getattr(self.theta_source, "rank1_scoring", False))) and not self.annealing

    def _rank1_advance(self, new_obs):
        """scale' = ((n0 - 2) scale + e e^T) / (n0 - 1), e = y - f_sm  (GPI_model.py:1332-1336): its Cholesky factor from the
        previous one by ONE rank-1 update, O(T^2) (hgp_chol_rank1_f64) - the consumer of BASELINE configs[4]'s kernel."""
        old = self.observation_params
        n0 = float(old.n0)
        if self._Lobs is None or self._Lobs[0] is not old:      # (re)start the recursion from a full factorisation
            L, info = ops.potrf_batched(old.scale.contiguous(), 0.0, 0.0)
            if bool(info.any()):
                self._Lobs = None
                return
            self._Lobs = (old, L[0])
        e = (self.y_train[-1] - self.f_star_sm[-1]).reshape(1, -1).contiguous()
        L1, info = ops.chol_rank1(self._Lobs[1].contiguous(), e, alpha=(n0 - 2.0) / (n0 - 1.0), beta=1.0 / (n0 - 1.0))
        self._Lobs = None if bool(info.any()) else (new_obs, L1)

    def rank1_scale_factor(self):
        """(L, c) with Sigma[-1] = c L L^T when the tracked factor belongs to the current observation distribution and the last
        Sigma is exactly a multiple of its scale (N > 1, no annealing); None otherwise (callers refactor)."""
        if (not self._rank1_on() or self._Lobs is None or self._Lobs[0] is not self.observation_params
                or self.N <= 1 or not self.N < self.estimation_limit):
            return None
        n0 = float(self.observation_params.n0)
        return self._Lobs[1], n0 / (n0 - 2.0)

    def backwards(self, h=1.0):
        """GPI_model.py:687-703 + GPI.backward (GPI.py:240-270): full RTS pass over the filtered states."""
        if h != 1.0:
            return
        mm = ops.gemm_batched
        means, covs = list(self.f_star[1:]), list(self.cov_f[1:])
        A_list, G_list = self.A[1:], self.Gamma[1:]
        for t in range(len(means) - 2, -1, -1):
            A = A_list[t] if t < len(A_list) else A_list[-1]
            Gam = G_list[t] if t < len(G_list) else G_list[-1]
            P = mm(mm(A, covs[t]), A, transB=True) + Gam
            J = mm(mm(covs[t], A, transB=True), self._spd_inv(P, "backwards"))
            means[t] = means[t] + mm(J, means[t + 1] - mm(A, means[t]))
            covs[t] = covs[t] + mm(mm(J, covs[t + 1] - P), J, transB=True)
        for i in range(len(means)):
            self.f_star_sm[i + 1] = means[i]
            self.cov_f_sm[i + 1] = covs[i]
        self._stk = {}

    # ---- the same recursion as capture-safe, buffer-based steps replayed as hipGraphs ----------------------------
    # The eager methods above issue ~90 launches and one host sync per member: launch-bound.  For a run of members
    # (dynamic model, shared grid, h = 1, N >= 2) the step is restated on pre-allocated stacks with device-side
    # indices/counters and no host synchronisation, captured once with torch.cuda.CUDAGraph (hipGraph) and replayed per
    # member; the RTS backward pass likewise.  Same arithmetic, same order.
    def _chain_alloc(self, n_more):
        T = self.x_basis.shape[0]
        L = len(self.f_star)
        dev = self.device

        def stack(lst, shape):
            buf = torch.empty((L + n_more,) + shape, dtype=f64, device=dev)
            buf[:L] = lst.stack() if isinstance(lst, StackList) else torch.stack(lst)
            buf[L:].zero_()
            return buf

        ch = {"F": stack(self.f_star, (T, 1)), "Fsm": stack(self.f_star_sm, (T, 1)), "P": stack(self.cov_f, (T, T)),
              "Psm": stack(self.cov_f_sm, (T, T)), "A": stack(self.A, (T, T)), "G": stack(self.Gamma, (T, T)),
              "C": stack(self.C, (T, T)), "S": stack(self.Sigma, (T, T))}
        ch["pos"] = torch.tensor([L - 1], dtype=torch.int64, device=dev)
        ch["Nf"] = torch.tensor([float(self.N)], dtype=f64, device=dev)
        ch["n0"] = torch.tensor([float(self.internal_params.n0)], dtype=f64, device=dev)
        eye = self._eye()
        # the two MNIW distributions (internal, observation) as one tensor: W[0] = means, W[1] = right covariances,
        # W[2] = scales, each [2,T,T]
        mi, mo = self.internal_params, self.observation_params
        ch["W"] = torch.stack((torch.stack((mi.m_mean, mo.m_mean)),
                               torch.stack((eye if mi.m_r_cov is None else mi.m_r_cov, eye if mo.m_r_cov is None else mo.m_r_cov)),
                               torch.stack((mi.scale, mo.scale)))).contiguous()
        ch["ws"] = torch.empty(6 * T * T + 2 * T, dtype=f64, device=dev)     # gathered previous state (hgp_lds_chain_gather_f64)
        ch["X5"] = torch.empty((5, T, T), dtype=f64, device=dev)             # Kalman P_k + the 4 inputs of the batched inverse
        ch["y1s"] = torch.zeros((2, T, 1), dtype=f64, device=dev)            # (f_post, y) of the MNIW updates
        ch["y2s"] = torch.zeros((2, T, 1), dtype=f64, device=dev)            # (f_sm_prev, f_post)
        ch["I0"] = torch.stack((eye, torch.zeros_like(eye))).contiguous()    # addends of the batched (I - K C, -K Sigma)
        ch["bad"] = torch.zeros(2, dtype=torch.int32, device=dev)          # [MNIW updates skipped, first step whose filter failed]
        ch["sync"] = torch.zeros(1, dtype=torch.int32, device=dev)           # inter-block counter of hgp_lds_chain_finish_f64
        return ch

    def _chain_step(self, ch):
        """One member (N >= 2 after it): include_sample + backwards_pair + bayesian_new_params on the stacks.

        (Measured: running the three branches that only read the previous state - Kalman update, smoother gain, first
        MNIW inverse - on separate streams, i.e. as parallel branches of the captured hipGraph, is SLOWER on ROCm 7.2:
        1.03 ms instead of 0.77 ms per member.  One stream.)"""
        mm = ops.gemm_batched
        eye = self._eye()
        T = eye.shape[0]
        tt = T * T
        pos = ch["pos"]
        y1s, y2s, X5 = ch["y1s"], ch["y2s"], ch["X5"]
        ws = ops.lds_chain_gather(ch["A"], ch["G"], ch["C"], ch["S"], ch["Psm"], ch["P"], ch["F"], ch["Fsm"], pos, ch["ws"],
                                  Y=ch["Y"], y_row0=ch["y_row0"], y_out=y1s[1])
        A, G, C, S, Psm, c0 = (ws[i * tt:(i + 1) * tt].view(T, T) for i in range(6))
        m0, Fsm = ws[6 * tt:6 * tt + T].view(T, 1), ws[6 * tt + T:].view(T, 1)   # filtered / smoothed mean of the previous step
        y = y1s[1]                                                              # the member's observation
        n0 = ch["n0"]
        means, Rs, scales = ch["W"][0], ch["W"][1], ch["W"][2]
        # The three factorisations that only need the previous state go out as ONE batch of 4 single-matrix inverses
        # (a 90 x 90 inverse is latency-bound: the launch costs the same with one matrix or four): A c0 A^T + G of
        # backwards_pair, S of the Kalman update, and the two MNIW scale matrices.  X5 = [Pk, P, Sk, R0', R1'] holds the
        # Kalman predictive covariance followed by the four inverse inputs; additions ride in the GEMM epilogues.
        AP = mm(A, ws[4 * tt:6 * tt].view(2, T, T))                            # A P_sm and A c0 (A shared)
        mm(AP, A, transB=True, add=G, out=X5[0:2])                             # predictive covariances of both
        Pk, P = X5[0], X5[1]
        Amf = mm(A, ws[6 * tt:].view(2, T, 1))                                 # A m0 and A f_sm (A shared)
        Am0, xm = Amf[0], Amf[1]
        f_pred = mm(C, xm)                                                     # pred_dist short-circuits on the shared grid
        CPk = mm(C, Pk)
        mm(CPk, C, transB=True, add=S, out=X5[2])
        ops.add_diag_mean(Rs, scales, 1e-2, out=X5[3:5])
        Z4, i4 = ops.chol_inverse(X5[1:5])
        # (i4 is rewritten by every graph replay: hgp_lds_chain_finish_f64 latches a failure of i4[:2] in ch["bad"][1])
        inv4 = mm(Z4, Z4, transA=True)
        i1, scale_inv = i4[2:], inv4[2:]
        # Kalman update (GPI.py:140-151, Joseph form).  P C^T is the transpose of the C P already formed for S (P is a
        # covariance), so the gain is one transposed product: K = (C P)^T S^{-1}.
        K_t = mm(CPk, inv4[1], transA=True)
        f_post = mm(K_t, y - f_pred, add=xm, out=y1s[0])
        KCS = mm(K_t, ws[2 * tt:4 * tt].view(2, T, T), alpha=-1.0, add=ch["I0"])   # (I - K C, -K Sigma) in one launch
        IKC = KCS[0]
        c_post = mm(KCS[1], K_t, transB=True, alpha=-1.0, add=mm(mm(IKC, Pk), IKC, transB=True))
        # backwards_pair on the last two filtered states: J = c0 A^T P^{-1} = (A c0)^T P^{-1}
        J = mm(AP[1], inv4[0], transA=True)
        f_sm_prev = mm(J, f_post - Am0, add=m0, out=y2s[0])
        y2s[1].copy_(f_post)
        P_sm_prev = mm(mm(J, c_post - P), J, transB=True, add=c0)
        ops.lds_chain_scatter(f_post, c_post, f_sm_prev, P_sm_prev, ch["F"], ch["Fsm"], ch["P"], ch["Psm"], pos)
        # bayesian_new_params (one-step MNIW update; on a failed factorisation the previous distributions are kept):
        # y1s = (f_post, y), y2s = (f_sm_prev, f_post)
        S__ = mm(y2s, y2s, transB=True, add=scale_inv)
        S_ = mm(y1s, y2s, transB=True, add=mm(means, scale_inv))
        Zs, i2 = ops.chol_inverse(S__, 0.0, 1e-8)
        part = mm(mm(S_, Zs, transB=True), Zs)
        e = y1s - y2s
        ops.lds_chain_finish(part, mm(e, e, transB=True), S__, i1, i2, ch["W"], n0, ch["Nf"], ch["bad"], ch["A"], ch["G"],
                             ch["C"], ch["S"], pos, self.annealing, ch["sync"], info0=i4)

    def _chain_lists(self, ch, views=None):
        """The member step as ONE launch per dependency level (hgp_chain.hip): every product of the step is an item of a
        device-resident list whose pointers are fixed for the life of the chain; the two inversions carry their right-hand
        sides.  14 launches per member, no torch arithmetic, no allocation (measured: a dependent launch costs ~4.5 us
        whatever it does, so launches - not flops - were the step's time).  128 < T <= 256: the inversions are the
        cooperative inverse-only kernels and Z rhs is one more list level behind each of them (16 launches)."""
        T = self.x_basis.shape[0]
        riding = T <= 128
        tt = T * T
        dev = self.device
        new = ch.get("alloc") or (lambda *shape: torch.zeros(shape, dtype=f64, device=dev))      # noqa: E731  (a pool hands out slices of its arena)
        ws = ch["ws"]
        A, G, C, S, Psm, c0 = (ws[i * tt:(i + 1) * tt].view(T, T) for i in range(6))
        m0, Fsm = ws[6 * tt:6 * tt + T], ws[6 * tt + T:]
        v = views or {}     # chain_batch.run hands out slices of buffers shared by many chains (one batched inversion for all)
        X4, RH4, Z4, Y4 = (v[k] if k in v else new(4, T, T) for k in ("X4", "RH4", "Z4", "Y4"))   # [P, Sk, R0', R1'], riding RHS, Z, Z rhs
        S__, S_, Zs, Y3 = (v[k] if k in v else new(2, T, T) for k in ("S__", "S_", "Zs", "Y3"))
        part = new(2, T, T)
        AP0, Pk, K_t, J, SINV, IKC, KS, MS, T1, KKt, KKtmP, c_post, CmP, X, P_sm_prev = (
            new(T, T), new(T, T), new(T, T), new(T, T), new(2, T, T), new(T, T), new(T, T), new(2, T, T), new(T, T), new(T, T),
            new(T, T), new(T, T), new(T, T), new(T, T), new(T, T))
        y, xm, innov, f_post, w, f_sm_prev = new(T), new(T), new(T), new(T), new(T), new(T)
        means = ch["W"][0]
        P, Sk = X4[0], X4[1]
        lv = [ops.GemmList(dev) for _ in range(10)]
        # L1-L4: predictions (GPI.py:100-139; GPI.py:283-287 for the pair smoother's P = A c0 A^T + G)
        lv[0].add(A, Psm, AP0)
        lv[0].add(A, c0, RH4[0])                                  # A c0, the smoother gain's right-hand side
        lv[0].add(A, Fsm, xm)
        lv[1].add(AP0, A, Pk, D=G, transB=True)
        lv[1].add(RH4[0], A, P, D=G, transB=True)
        lv[1].add(C, xm, innov, D=y, alpha=-1.0)                  # y - C x_m
        lv[2].add(C, Pk, RH4[1])                                  # C P_k, the Kalman gain's right-hand side
        lv[3].add(RH4[1], C, Sk, D=S, transB=True)
        # after INV1 (Z = L^-1 of P, Sk, R0', R1';  Y = Z rhs):  K = (C Pk)^T Sk^-1 = Y1^T Z1,  J = (A c0)^T P^-1 = Y0^T Z0
        lv[4].add(Y4[1], Z4[1], K_t, transA=True)
        lv[4].add(Y4[0], Z4[0], J, transA=True)
        lv[4].add(Z4[2], Z4[2], SINV[0], transA=True)
        lv[4].add(Z4[3], Z4[3], SINV[1], transA=True)
        lv[5].add(K_t, innov, f_post, D=xm)
        lv[5].add(K_t, C, IKC, alpha=-1.0, add_eye=1.0)
        lv[5].add(K_t, S, KS)
        lv[5].add(means[0], SINV[0], MS[0])
        lv[5].add(means[1], SINV[1], MS[1])
        lv[6].add(IKC, Pk, T1)
        lv[6].add(KS, K_t, KKt, transB=True)
        lv[6].add(KS, K_t, KKtmP, D=P, transB=True, beta=-1.0)
        lv[6].add(A, m0, w, D=f_post, alpha=-1.0)                 # f_post - A m0
        lv[7].add(T1, IKC, c_post, D=KKt, transB=True)            # Joseph form (GPI.py:148-150)
        lv[7].add(T1, IKC, CmP, D=KKtmP, transB=True)             # c_post - P for the smoother
        lv[7].add(J, w, f_sm_prev, D=m0)
        lv[8].add(J, CmP, X)
        lv[8].add(f_sm_prev, f_sm_prev, S__[0], D=SINV[0], transB=True)      # y2 y2^T + R'^-1 (GPI_model.py:1317-1322)
        lv[8].add(f_post, f_post, S__[1], D=SINV[1], transB=True)
        lv[8].add(f_post, f_sm_prev, S_[0], D=MS[0], transB=True)            # y1 y2^T + M R'^-1
        lv[8].add(y, f_post, S_[1], D=MS[1], transB=True)
        # after INV2 (Zs of S__ + 1e-8 I;  Y3 = Zs S_^T):  S_ S__^-1 = Y3^T Zs
        lv[9].add(Y3[0], Zs[0], part[0], transA=True)
        lv[9].add(Y3[1], Zs[1], part[1], transA=True)
        lv[9].add(X, J, P_sm_prev, D=c0, transB=True)
        if not riding:
            lvy = [ops.GemmList(dev), ops.GemmList(dev)]
            lvy[0].add(Z4[0], RH4[0], Y4[0])                     # Z_P (A c0)
            lvy[0].add(Z4[1], RH4[1], Y4[1])                     # Z_S (C P_k)
            lvy[1].add(Zs[0], S_[0], Y3[0], transB=True)         # Z_s S_^T
            lvy[1].add(Zs[1], S_[1], Y3[1], transB=True)
            lv += lvy
        if views is None:
            for l_ in lv:
                l_.finalize()
        ch["lv"], ch["riding"] = lv, riding
        ch["bufs"] = dict(X4=X4, RH4=RH4, Z4=Z4, Y4=Y4, S__=S__, S_=S_, Zs=Zs, Y3=Y3, part=part, y=y, f_post=f_post, c_post=c_post,
                          f_sm_prev=f_sm_prev, P_sm_prev=P_sm_prev)
        ch["rhs_on"] = torch.tensor([1, 1, 0, 0], dtype=torch.int32, device=dev)
        ch["i4"] = v["i4"] if "i4" in v else torch.zeros(4, dtype=torch.int32, device=dev)
        ch["i2"] = v["i2"] if "i2" in v else torch.zeros(2, dtype=torch.int32, device=dev)

    def _chain_step2(self, ch):
        """_chain_step with one launch per dependency level; see _chain_lists."""
        lv, b = ch["lv"], ch["bufs"]
        ops.lds_chain_gather2(ch["A"], ch["G"], ch["C"], ch["S"], ch["Psm"], ch["P"], ch["F"], ch["Fsm"], ch["pos"], ch["ws"],
                              ch["Y"], ch["y_row0"], b["y"], ch["W"], b["X4"][2:4])
        for i in range(4):
            lv[i].run()
        if ch["riding"]:
            ops.chol_inverse_rhs(b["X4"], b["Z4"], b["RH4"], b["Y4"], ch["i4"], rhs_on=ch["rhs_on"])
        else:
            ops.chol_inverse(b["X4"], out=b["Z4"], info=ch["i4"])
            lv[10].run()
        for i in range(4, 9):
            lv[i].run()
        if ch["riding"]:
            ops.chol_inverse_rhs(b["S__"], b["Zs"], b["S_"], b["Y3"], ch["i2"], rhs_trans=True, add_diag=1e-8)
        else:
            ops.chol_inverse(b["S__"], 0.0, 1e-8, out=b["Zs"], info=ch["i2"])
            lv[11].run()
        lv[9].run()
        ops.lds_chain_finish2(b["f_post"], b["c_post"], b["f_sm_prev"], b["P_sm_prev"], b["y"], b["part"], b["S__"], ch["i4"], ch["i2"],
                              ch["W"], ch["n0"], ch["Nf"], ch["bad"], ch["A"], ch["G"], ch["C"], ch["S"], ch["F"], ch["Fsm"], ch["P"],
                              ch["Psm"], ch["pos"], self.annealing, ch["sync"])

    def _chain_commit(self, ch, members, x_trains, y_trains):
        L = int(ch["pos"][0]) + 1
        unb = lambda k: StackList(ch[k][:L])               # noqa: E731  (rows stay views of the chain's stacks)
        self.f_star, self.f_star_sm, self.cov_f, self.cov_f_sm = unb("F"), unb("Fsm"), unb("P"), unb("Psm")
        self.A, self.Gamma, self.C, self.Sigma = unb("A"), unb("G"), unb("C"), unb("S")
        n0 = float(ch["n0"])
        W = ch["W"]
        self.internal_params = matrix_normal_inv_wishart(W[0, 0], W[1, 0], n0, W[2, 0])
        self.observation_params = matrix_normal_inv_wishart(W[0, 1], W[1, 1], n0, W[2, 1])
        for idx in members:
            self.indexes.append(int(idx))
            self.x_train.append(x_trains[idx])
            self.y_train.append(y_trains[idx].reshape(-1, 1))
        self.N += len(members)
        self._stk = {}

    def _run_graphed(self, fn, n_iter, unroll=1):
        """Run fn() n_iter times: once eagerly (warm-up, counts as the first iteration), then `unroll` iterations captured ONCE
        as a hipGraph and replayed (a replay costs ~8 us of launch gap, amortised over the unrolled iterations); the remainder
        runs eagerly.  A failed capture or replay raises: silently re-running eagerly would both hide a 30x slow-down and,
        after a partial replay, apply steps twice."""
        if n_iter <= 0:
            return
        self._check_pending()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()                                        # warm-up iteration (counts as the first one)
        torch.cuda.current_stream().wait_stream(side)
        n_iter -= 1
        if n_iter == 0:
            return
        unroll = max(1, min(int(unroll), n_iter))
        graph = torch.cuda.CUDAGraph()
        keep = self._pending
        try:
            with torch.cuda.graph(graph):
                for _ in range(unroll):
                    fn()
        except RuntimeError as e:
            raise RuntimeError(f"hipGraph capture of the LDS step failed: {e}") from e
        self._pending = keep + self._pending            # info tensors written by every replay
        for _ in range(n_iter // unroll):
            graph.replay()
        for _ in range(n_iter % unroll):
            fn()
        self.graph_replays = getattr(self, "graph_replays", 0) + n_iter // unroll
        self._graph_keepalive = graph

    def _backwards_graphed(self):
        """GPI_model.backwards (full RTS) with the step captured once: index t runs down on the device."""
        mm = ops.gemm_batched
        L = len(self.f_star)
        if L - 1 < 2:
            return self.backwards()
        st = lambda lst: (lst.stack() if isinstance(lst, StackList) else torch.stack(lst))[1:]     # noqa: E731
        M, Cv = st(self.f_star).clone(), st(self.cov_f).clone()          # smoothed in place below: the filtered lists stay
        A, G = st(self.A).contiguous(), st(self.Gamma).contiguous()
        nA, n = A.shape[0], M.shape[0]
        # Everything that only reads the FILTERED states is batched over all steps at once (GPI.py:252-262 uses the
        # filtered cov_t for P_t and the gain J_t): P_t = A_t c_t A_t^T + G_t, its inverse, J_t = c_t A_t^T P_t^{-1} and
        # A_t m_t.  Only the two-line recursion itself stays sequential.
        if nA >= n - 1:
            Ab, Gb = A[:n - 1], G[:n - 1]
        else:
            ta_all = torch.clamp_max(torch.arange(n - 1, device=self.device), nA - 1)
            Ab, Gb = A.index_select(0, ta_all), G.index_select(0, ta_all)
        Pb = mm(mm(Ab, Cv[:n - 1]), Ab, transB=True, add=Gb.contiguous())
        Zb, infob = ops.chol_inverse(Pb)
        self._pending.append(("backwards", infob))
        Jb = mm(mm(Cv[:n - 1], Ab, transB=True), mm(Zb, Zb, transA=True))
        AMb = mm(Ab, M[:n - 1])
        if M.shape[1] <= 96:          # the whole recursion in one launch (one workgroup walks the chain)
            self._check_pending()
            Mv = M.reshape(n, -1)
            ops.rts_chain(Jb, Pb, AMb.reshape(n - 1, -1).contiguous(), Mv, Cv)
            self.f_star_sm = StackList(torch.cat([self.f_star_sm[0].unsqueeze(0), M]))
            self.cov_f_sm = StackList(torch.cat([self.cov_f_sm[0].unsqueeze(0), Cv]))
            self._stk = {}
            return
        # T > 96: the two-line recursion as product lists (hgp_gemm_list_f64), two dependent launches per step and no other
        # arithmetic: what only reads filtered quantities is folded into addends first,
        #     m_t <- J_t m_{t+1} + (m_t - J_t A_t m_t),      C_t <- (J_t C_{t+1} - J_t P_t) J_t^T + C_t.
        self._check_pending()
        T = M.shape[1]
        JP = mm(Jb, Pb)
        Mm = mm(Jb, AMb, alpha=-1.0, add=M[:n - 1].contiguous())
        X = torch.empty((T, T), dtype=f64, device=self.device)
        gl = ops.GemmList(self.device)
        for t in range(n - 2, -1, -1):
            gl.add(Jb[t], Cv[t + 1], X, D=JP[t], beta=-1.0)
            gl.add(Jb[t], M[t + 1], M[t], D=Mm[t])
            gl.add(X, Jb[t], Cv[t], D=Cv[t], transB=True)
        gl.finalize()
        for i in range(n - 1):
            gl.run_range(3 * i, 2)
            gl.run_range(3 * i + 2, 1)
        for i in range(M.shape[0]):
            self.f_star_sm[i + 1] = M[i]
            self.cov_f_sm[i + 1] = Cv[i]
        self._stk = {}

    def full_pass_weighted(self, x_trains, y_trains, resp, q=None, q_lat=None, snr=None, use_graphs=True):
        """GPI_model.py:377-406: filter / smooth / re-estimate over the members (resp > 0.99), then score everything.
        With use_graphs (default) members beyond the first are processed by hipGraph replays of the captured step."""
        x_trains = self.cond_to_torch(x_trains)
        y_trains = self.cond_to_torch(y_trains)
        resp = torch.as_tensor(resp)
        dynamic = bool(torch.any(self.Gamma[-1] != 0))
        active = torch.nonzero(resp > 0.99, as_tuple=False).reshape(-1).tolist()
        if len(active) == 0:
            return q, q_lat
        hs = [float(resp[i]) for i in active]
        X2 = x_trains[..., 0] if x_trains.ndim == 3 else x_trains
        graphable = (use_graphs and dynamic and all(h == 1.0 for h in hs) and len(active) >= 4 and
                     self.estimation_limit == np.inf and
                     bool(torch.equal(X2[active], self.x_basis.reshape(1, -1).expand(len(active), -1))))
        if graphable:
            head = 1 if self.N == 0 else 0                  # the first member of a fresh model takes the eager path
            for index in active[:head]:
                self.include_weighted_sample(index, x_trains[index], x_trains[index], y_trains[index], 1.0)
                self.backwards_pair(1.0)
                self.bayesian_new_params(1.0)
            rest = active[head:]
            ch = self._chain_alloc(len(rest))
            # observations of the run; the step reads row (pos - y_row0) inside its gather kernel
            ch["Y"] = (y_trains[rest][..., 0] if y_trains.ndim == 3 else y_trains[rest]).reshape(len(rest), -1).contiguous()
            ch["y_row0"] = int(ch["pos"][0])
            if not ops.env_flag("HGP_CHAIN_PER_PRODUCT"):
                self._chain_lists(ch)
                self._run_graphed(lambda: self._chain_step2(ch), len(rest), unroll=8)
            else:                                       # round-1 form, one launch per product: kept for comparison
                self._run_graphed(lambda: self._chain_step(ch), len(rest))
            self._chain_commit(ch, rest, x_trains, y_trains)
            bad = ch["bad"].tolist()
            if bad[1] != 0:      # torch.linalg.solve / inv of the reference would have raised at that member
                raise torch.linalg.LinAlgError(f"posterior / backwards_pair: the input is not positive-definite (LDS step {bad[1]})")
            if bad[0] != 0 and self.verbose:
                print("Alg error matrix ill conditioned.")     # GPI_model.py:1069
            self._backwards_graphed()
        else:
            for index, h in zip(active, hs):
                self.include_weighted_sample(index, x_trains[index], x_trains[index], y_trains[index], h)
                if dynamic:
                    self.backwards_pair(h)
                    self.bayesian_new_params(h)
            if dynamic:
                self.backwards()
        self._check_pending()
        self._stk = {}
        return self.compute_sq_err_all(x_trains, y_trains), self.compute_q_lat_all(x_trains)

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

    def _pairs_plan(self, Ts, K):
        """Per-pair plans (device buffer + handle) are reused across calls: the online loop calls log_sq_error 1 + 2M times per
        beat on irregular grids; only update() (the per-cluster operators) runs per call.  A plan is 37 MiB (T = 90, one
        cluster) to 410 MiB (T = 256, 16 clusters) of device memory and a GPI_HDP holds M x n_outputs models, so the plans live
        in ONE least-recently-used cache per process, bounded in bytes (ops.plan_cache), not per model."""
        theta = tuple(float(v) for v in self.gp.kernel.params())
        return ops.plan_cache(self.x_basis.shape[0], int(Ts), int(K), theta, self.device)

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
            last = mean is None and (i is None or i == -1 or i >= len(self.indexes)) and not first and not proj
            fac = self.rank1_scale_factor() if last else None
            if fac is not None:      # Sigma[-1] = c L L^T with L carried by rank-1 updates: |L^-1 d|^2 / c, no factorisation.
                # (the reference's 1e-8 mean|diag| jitter of _chol_spd is not applied: relative change of the score <= 1e-8 cond)
                quad = ops.trsv_lower_quad(fac[0], (y.reshape(-1) - m.reshape(-1)).contiguous()) / fac[1]
                return -0.5 * quad - 0.5 * y.shape[1] * LOG2PI
            items = ops.build_items([0], [fn], [1])
            quad, _, info = ops.score_groups(y, m.reshape(1, T).contiguous(), S.reshape(1, T, T).contiguous(), *items)
        else:
            plan = self._pairs_plan(x.shape[0], 1)
            plan.update(self.x_basis.reshape(-1).contiguous(), m.reshape(1, T).contiguous(), S.reshape(1, T, T).contiguous())
            ops.raise_on_info(plan.info, "pred_dist")
            fnt = torch.full((1, 1), fn, dtype=f64, device=self.device) if first else None
            quad, _, info = plan.loglik(x.reshape(1, -1).contiguous(), y, first_noise=fnt, want_logdet=False)
            quad = quad.reshape(-1)
        self._raise(info, "log_sq_error")
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
            means = ops.gemm_batched(self._S("C")[torch.as_tensor(pairs[:, 0], device=self.device)],
                                     self._S("f_star")[torch.as_tensor(pairs[:, 1], device=self.device)]).reshape(-1, T)
            adds = np.where(codes % 2 == 1, ini_noise, 0.0)
            inv = inv.reshape(-1)
            quad = torch.zeros(n, dtype=f64, device=self.device)
            multi = counts > 1
            if multi.any():   # several segments share one (step, first) state: one factorisation, many right-hand sides
                gm = np.nonzero(multi)[0]
                seg_list = np.concatenate([order[start[g]:start[g] + counts[g]] for g in gm]).astype(np.int32)
                im, ia, io, ic = ops.build_items(g_ci[gm].tolist(), adds[gm].tolist(), counts[gm].tolist())
                grp_of_item = np.repeat(np.arange(len(gm)), [-(-c // ops.MAX_CHUNK) for c in counts[gm]])
                quad, _, info = ops.score_groups(Y, means.contiguous(), self._S("Sigma"), im, ia, io, ic, seg_ids=seg_list,
                                                 item_mean=inv[gm][grp_of_item].astype(np.int32))
                ops.raise_on_info(info, "compute_sq_err_all")
            single = ~multi
            if single.any():  # member segments: each has its own Sigma_i and a single right-hand side
                gs = np.nonzero(single)[0]
                segs = torch.as_tensor(rep[gs], device=self.device)
                Sst = self._S("Sigma")
                sym = self._S_symmetric("Sigma")      # exactly symmetric stacks take the half-traffic path
                q1, _, info1 = ops.score_each(Y[segs].contiguous(), means.contiguous(), Sst, g_ci[gs].astype(np.int32),
                                              inv[gs].astype(np.int32), adds[gs], symmetric=sym)
                ops.raise_on_info(info1, "compute_sq_err_all")
                quad[segs] = q1
            return -0.5 * quad - 0.5 * T * LOG2PI
        # general path (GPI_model.py:535-545): every segment against the state of ITS step, on its own grid
        pairs, col = np.unique(np.stack([ci_seg, fi_seg], 1), axis=0, return_inverse=True)
        col = col.reshape(-1)
        means = ops.gemm_batched(self._S("C")[torch.as_tensor(pairs[:, 0], device=self.device)],
                                 self._S("f_star")[torch.as_tensor(pairs[:, 1], device=self.device)]).reshape(-1, T)
        Sig = self._S("Sigma")[torch.as_tensor(pairs[:, 0], device=self.device)].contiguous()
        plan = self._pairs_plan(Ts, len(pairs))
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
        ix = lambda a: ops.to_dev(a, torch.int64, self.device)  # noqa: E731
        Gam = self._S("Gamma")[ix(par)].clone()
        if only is None or only == 0:
            Gam[0] = Gam[0] * h_ini                                  # GPI_model.py:293
        out, info = ops.lat_error(self._S("f_star_sm")[ix(cur)].reshape(-1, T).contiguous(),
                                  self._S("f_star_sm")[ix(prev)].reshape(-1, T).contiguous(), self._S("A")[ix(par)].contiguous(), Gam,
                                  self._S("cov_f_sm")[ix(cov)].contiguous())
        self._raise(info, "log_lat_error")
        return out - 0.5 * T * LOG2PI

    def log_lat_error(self, i, h_ini):
        """GPI_model.py:288-323."""
        return self._lat_all(h_ini, only=i)[0]

    def compute_q_lat_all(self, x_trains, h_ini=1.0):
        """GPI_model.py:549-559: all members of the cluster in one batch."""
        n = x_trains.shape[0]
        if self.N == 0 or not self._is_dynamic():
            return torch.zeros(n, dtype=f64, device=self.device)
        ent = self._stk.get("_lat_all")          # members' scores are kept until the model changes (the online loop asks every
        key = (h_ini, len(self.indexes), len(self.Gamma), self.f_star_sm[-1].data_ptr(), self.cov_f_sm[-1].data_ptr())
        if ent is None or ent[0] != key:         # cluster for them at every beat, GPI_HDP.py:1972; only one cluster changes)
            ent = self._stk["_lat_all"] = (key, self._lat_all(h_ini))
        # ... and so is the column itself (zeros with the members' scores scattered in), with room to grow: while the model does
        # not change, a longer history only needs a longer view of it
        col = self._stk.get("_lat_col")
        if col is None or col[0] is not ent[1] or col[1].shape[0] < n:
            buf = torch.zeros(max(64, 2 * n), dtype=f64, device=self.device)
            buf[ops.to_dev(self.indexes, torch.int64, self.device)] = ent[1]
            col = self._stk["_lat_col"] = (ent[1], buf)
        return col[1][:n]

    def _is_dynamic(self):
        """bool(any(Gamma[-1] != 0)) (GPI_model.py:551), looked up once per state of the list (a device round trip otherwise)."""
        G = self.Gamma[-1]
        key = (len(self.Gamma), G.data_ptr())
        ent = getattr(self, "_dyn", None)
        if ent is None or ent[0] != key:
            ent = self._dyn = (key, bool(torch.any(G != 0)))
        return ent[1]

    def _dyn_prior(self):
        """bool(any(Gamma_def != 0)) (GPI_model.py:470), looked up once per prior object."""
        ent = getattr(self, "_dyn_def", None)
        if ent is None or ent[0] is not self.Gamma_def:
            ent = self._dyn_def = (self.Gamma_def, bool(torch.any(self.Gamma_def != 0)))
        return ent[1]

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
        if self._dyn_prior():
            Ms.append(A_), Ss.append(Gam_), means.append(self.A_def), scales.append(self.Gamma_def)
        # the prior scales are the initial sigma I / gamma I unless a caller replaced them: checked once per pair of objects
        key = (id(self.Sigma_def), id(self.Gamma_def))
        if getattr(self, "_def_diag_key", None) != key:
            self._def_diag = all(bool(torch.equal(s_, torch.diag(torch.diagonal(s_)))) for s_ in scales)
            self._def_diag_key = key
        out, info = ops.mniw_loglik(torch.stack(Ms).contiguous(), torch.stack(Ss).contiguous(), torch.stack(means).contiguous(),
                                    None, torch.stack(scales).contiguous(), scale_is_diagonal=self._def_diag)
        self._raise(info, "return_LDS_param_likelihood")
        return torch.sum(out) / T * 100.0

    def lds_param_likelihood_value(self):
        """return_LDS_param_likelihood() as a host float, evaluated once per state of the model: the variational loop asks for
        it at every evaluation of the bound (GPI_HDP.py:1838-1864), mostly for models that have not changed.  Stored with the
        stack cache, which every method that changes the per-step lists resets."""
        ent = self._stk.get("_lds_lik")
        key = (len(self.A), self.A[-1].data_ptr(), self.Sigma[-1].data_ptr(), self.Gamma[-1].data_ptr())
        if ent is None or ent[0] != key:
            ent = self._stk["_lds_lik"] = (key, float(self.return_LDS_param_likelihood()))
        return ent[1]

