"""Drop-in surface of hdpgpc/hdpgpc/GPI_HDP.py for the GP-emission hot path (Face 1).

Built: the constructor (same keyword arguments, GPI_HDP.py:100-111), the per-(lead, cluster) model table ``gpmodels`` and the
driver entry points -

* ``include_batch`` (GPI_HDP.py:805-943, ``warp=`` / the drivers' ``with_warp=``; offline_loop.py) and
  ``cluster_new_batch(learning=True)`` (GPI_HDP.py:3005-3145): the offline variational loop;
* ``include_sample`` (GPI_HDP.py:1906-2208; online_loop.py + online_chain.py): the online step, one lead, ``with_warp=False``;
* ``reload_model_from_labels`` (GPI_HDP.py:3952-4035): one ``GPI_model.full_pass_weighted`` per (lead, class) from given
  labels, then the HDP pseudo-counts (host-side, hdp_global.py);
* ``cluster_new_batch(learning=False)`` (GPI_HDP.py:2975-3003): the N x M batch of frozen-model scores
  (``log_sq_error(x_i, y_i, i=-1)`` for every pair, GPI_HDP.py:2981-2985) as ONE pair-kernel launch (irregular grids) or
  one shared-covariance launch per cluster (segments on the basis grid), then LogLik -> forward / backward -> one-hot
  arg-max, all on the device (ops.hmm_messages, ops.assign);

plus the small public helpers they and the drivers use (``LogLik``, ``_safe_exp``, ``weight_mean``, ``forward``,
``backward``, ``coupled_state_coef``, ``compute_trans_A/pi``, ``selected_gpmodels``).  NOT built (each raises
NotImplementedError): ``include_sample(with_warp=True)`` (the reference's own path raises at its second beat), ``classify=True``,
several leads online, static models, ``estimation_limit``, inducing points, ``kernels=`` objects, ``bayesian_params=False``.

Kernel hyper-parameters: the reference fits them with gpytorch on the first member of every cluster (GPI.py:610-770).
``GPI_HDP.fixed_theta = (outputscale, lengthscale, noise)`` injects them instead (what the golden fixtures do); with
``fixed_theta = None`` the mirror's own fit (hdpgpc_amd.kernel_fit, when present) runs.
"""
import numpy as np
import torch
from scipy.special import digamma as _digamma

from . import hdp_global, ops
from .GPI import RBFWhiteKernel
from .GPI_model import GPI_model, _copy_list
from .offline_loop import OfflineLoop
from .online_loop import OnlineLoop

f64 = torch.float64
_HDP_HYP = {"less": (0.01, 0.01, 0.01, 0.0), "balanced": (1.0, 1.0, 0.1, 0.0), "more": (10.0, 10.0, 1.0, 0.0)}


def _limit_host_threads():
    """The host side of this package only ever touches SMALL CPU tensors (one-hot tables, counts, K x K HDP terms); PyTorch's
    intra-op pool (one thread per core: 128 on an MI355X host) turns each of those 30 us operations into 0.2 - 1 ms of thread
    wake-ups (measured: tools/bench_host_ops.py).  One thread for CPU tensor ops unless HGP_HOST_THREADS says otherwise
    (0 = leave PyTorch's setting alone)."""
    import os
    n = int(os.environ.get("HGP_HOST_THREADS", "1"))
    if n > 0 and torch.get_num_threads() != n:
        torch.set_num_threads(n)


def _kernel_params(k):
    """(output-scale, length-scale, noise level, noise bounds or None) of ConstantKernel * RBF + WhiteKernel (scikit-learn's
    composition, GPI_HDP.py:164-166) or of an RBFWhiteKernel."""
    if isinstance(k, RBFWhiteKernel):
        return (k.constant_value, k.length_scale, k.noise_level, None)
    try:
        c = float(k.k1.k1.constant_value)
        ell = float(np.asarray(k.k1.k2.length_scale).reshape(-1)[0])
        nb = getattr(k.k2, "noise_level_bounds", None)
        nb = None if nb is None or isinstance(nb, str) else (float(nb[0]), float(nb[1]))
        return (c, ell, float(k.k2.noise_level), nb)
    except AttributeError as e:
        raise TypeError("kernels=: expected ConstantKernel * RBF + WhiteKernel (scikit-learn) or RBFWhiteKernel objects") from e


def _first(v):
    """Per-cluster options may arrive as one value or as a list with one value per initial cluster (GPI_HDP.py:123-156)."""
    return v[0] if isinstance(v, (list, np.ndarray)) and np.ndim(v) > 0 else v


class GPI_HDP(OfflineLoop, OnlineLoop):
    _default_device = "cuda"     # every tensor of this build lives on the GPU

    def __init__(self, x_basis, M=None, n_outputs=1, x_basis_warp=None, kernels=None, model_type='dynamic',
                 ini_lengthscale=None, bound_lengthscale=None, ini_gamma=None, ini_sigma=None, ini_outputscale=None,
                 bound_sigma=(1e-10, 1e+10), bound_gamma=(1e-1, 1e+2), bound_noise_warp=(1e-10, 1e+10),
                 reest_conditions=[1, 20, 5], noise_warp=0.05, recursive_warp=False, warp_updating=False,
                 method_compute_warp='greedy', mode_warp='rough', verbose=False, annealing=True, hmm_switch=True,
                 max_models=None, batch=None, check_var=False, bayesian_params=True, cuda=False, inducing_points=False,
                 estimation_limit=None, reestimate_initial_params=False, n_explore_steps=10, free_deg_MNIV=5,
                 share_gp=False, use_snr=True, reduce_outputs=False, reduce_outputs_ratio=1.0, hdp_hyp='balanced'):
        kern0 = None
        if kernels is not None:
            # GPI_HDP.py:160-168 builds ConstantKernel(c, (c, 5 c)) * RBF(ell, bounds) + WhiteKernel(noise, bounds) per cluster when no
            # kernels are given; explicit ones (scikit-learn objects of that shape, or RBFWhiteKernel) are read for their parameters.
            # One kernel for all initial clusters (what every driver passes): per-cluster initial kernels are not built
            ks = list(kernels) if isinstance(kernels, (list, tuple)) else [kernels]
            pars = [_kernel_params(k) for k in ks]
            if any(p_ != pars[0] for p_ in pars[1:]):
                raise NotImplementedError("kernels=: different initial kernels per cluster are not part of this build")
            kern0 = pars[0]
        if inducing_points or estimation_limit is not None:
            raise NotImplementedError("inducing points / estimation_limit are not part of this build")
        _limit_host_threads()
        self.M = 1 if M is None else int(M)
        self.n_outputs = int(n_outputs)
        self.verbose = verbose
        self.cuda = True                       # the flag is kept for the drivers
        self.device = self._default_device
        self.x_basis_ini = np.asarray(x_basis[0] if isinstance(x_basis, list) else x_basis, dtype=np.float64).reshape(-1, 1)
        self.x_basis = [self.x_basis_ini] * self.M
        self.x_basis_warp = x_basis_warp
        self.model_type_def = _first(model_type)
        if self.model_type_def != 'dynamic':
            raise NotImplementedError("static models are not part of this build")
        self.model_type = [self.model_type_def] * self.M
        self.ini_sigma_def, self.ini_gamma_def = float(_first(ini_sigma)), float(_first(ini_gamma))
        osc = _first(ini_outputscale)
        self.ini_outputscale_def = self.ini_sigma_def if osc is None else float(osc)      # GPI_HDP.py:157-158
        self.ini_lengthscale, self.bound_lengthscale = _first(ini_lengthscale), bound_lengthscale
        if kern0 is not None:      # explicit kernel: its output-scale, length-scale and white-noise bounds are the defaults
            self.ini_outputscale_def, self.ini_lengthscale = kern0[0], kern0[1]
            if kern0[3] is not None:
                bound_sigma = kern0[3]
        self.bound_sigma_def = tuple(bound_sigma[0]) if isinstance(bound_sigma, list) else tuple(bound_sigma)
        self.bound_gamma_def = tuple(bound_gamma[0]) if isinstance(bound_gamma, list) else tuple(bound_gamma)
        self.annealing_def = bool(_first(annealing))
        self.hmm_switch, self.max_models, self.batch = hmm_switch, max_models, batch
        self.use_snr, self.bayesian_params, self.free_deg_MNIV = use_snr, bayesian_params, free_deg_MNIV
        self.n_explore_steps, self.reestimate_initial_params = n_explore_steps, reestimate_initial_params
        self.share_gp, self.reduce_outputs, self.reduce_outputs_ratio = share_gp, reduce_outputs, reduce_outputs_ratio
        self.f_ind_old = torch.zeros(self.M, dtype=torch.int64)
        self.noise_warp, self.mode_warp, self.method_compute_warp = noise_warp, mode_warp, method_compute_warp
        self.bound_noise_warp_def = tuple(bound_noise_warp[0]) if isinstance(bound_noise_warp, list) else tuple(bound_noise_warp)
        self.recursive_warp_def = bool(_first(recursive_warp))
        self.x_basis_warp_def = self.x_basis_ini if x_basis_warp is None else np.asarray(
            x_basis_warp[0] if isinstance(x_basis_warp, list) else x_basis_warp, dtype=np.float64).reshape(-1, 1)
        self.warp = False
        self.wp_sys = [[self.create_wp_sys_default() for _ in range(self.M)] for _ in range(self.n_outputs)]   # GPI_HDP.py:221-226
        self._warp_cache_full = {}
        self.static_factor = self.dynamic_factor = 1.0                                        # GPI_HDP.py:181-182
        self.gamma, self.transAlpha, self.startAlpha, self.kappa = _HDP_HYP[hdp_hyp]           # GPI_HDP.py:274-291
        self.fixed_theta = None
        self.rank1_scoring = False     # annealing=False only: carry chol(scale) by rank-1 updates instead of refactoring (GPI_model.py)
        self.train_elbo, self.resp_assigned, self.q = [], [], []
        self.T = 0
        self.x_train, self.y_train, self.y = [], torch.tensor([]), []
        self.actual_state = 0
        self.fmsg = self.margPrObs = None
        self.elbo_last = None
        self.snr_norm = None
        self.gpmodels = [[self.create_gp_default() for _ in range(self.M)] for _ in range(self.n_outputs)]
        self.init_global_params(self.M)

    # ------------------------------------------------------------------ per-cluster models
    def create_gp_default(self, i=None):
        """A fresh cluster model with the default priors (GPI_HDP.py:496-572, without the warping systems)."""
        kern = RBFWhiteKernel(self.ini_outputscale_def, float(self.ini_lengthscale), self.bound_sigma_def[0], device=self.device)
        gp = GPI_model(kern, self.x_basis_ini, annealing=self.annealing_def, bayesian=self.bayesian_params,
                       free_deg_MNIV=self.free_deg_MNIV, verbose=self.verbose)
        gp.noise_bounds = self.bound_sigma_def
        cond = gp.GPR_dynamic(self.ini_gamma_def, self.ini_sigma_def)
        gp.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
        gp.theta_source = self
        # the prior scales are gamma I / sigma I by construction (GPR_dynamic): spare return_LDS_param_likelihood its device test
        gp._def_diag_key, gp._def_diag = (id(gp.Sigma_def), id(gp.Gamma_def)), True
        return gp

    def create_wp_sys_default(self):
        """GPI_HDP.py:573-583: a fresh time-warp fitter with the default options (one per lead and cluster)."""
        from .amtgp_warping_system import Warping_system
        return Warping_system(self.x_basis_warp_def, self.noise_warp, self.bound_noise_warp_def, recursive=self.recursive_warp_def,
                              bayesian=self.bayesian_params, mode=self.mode_warp, device=self._default_device)

    def gpmodel_deepcopy(self, gpmodel):
        """GPI_HDP.py:4037-4064: a new model object with the same kernel hyper-parameters, priors and (shared, immutable)
        state tensors; the lists themselves are copied."""
        g = GPI_model(gpmodel.gp.kernel.clone_with_theta(gpmodel.gp.kernel.theta), gpmodel.x_basis.clone(), annealing=gpmodel.annealing,
                      bayesian=gpmodel.bayesian, free_deg_MNIV=gpmodel.free_deg_MNIV, verbose=self.verbose)
        for name in ("y_train", "x_train", "f_star", "f_star_sm", "cov_f", "cov_f_sm", "A", "Gamma", "C", "Sigma", "indexes"):
            setattr(g, name, _copy_list(getattr(gpmodel, name)))
        g.N, g.fitted, g.ini_cov_def = gpmodel.N, gpmodel.fitted, gpmodel.ini_cov_def
        g.A_def, g.Gamma_def, g.C_def, g.Sigma_def = gpmodel.A_def, gpmodel.Gamma_def, gpmodel.C_def, gpmodel.Sigma_def
        g.internal_params, g.observation_params = gpmodel.internal_params, gpmodel.observation_params
        g.fixed_theta, g.noise_bounds, g.estimation_limit = gpmodel.fixed_theta, gpmodel.noise_bounds, gpmodel.estimation_limit
        g.theta_source = gpmodel.theta_source
        g.rank1_scoring, g._Lobs = gpmodel.rank1_scoring, gpmodel._Lobs
        g.gp.fitted = gpmodel.gp.fitted
        if getattr(gpmodel, "_def_diag_key", None) == (id(g.Sigma_def), id(g.Gamma_def)):     # same prior objects: same verdict
            g._def_diag_key, g._def_diag = gpmodel._def_diag_key, gpmodel._def_diag
        if getattr(gpmodel, "_dyn_def", None) is not None:
            g._dyn_def = gpmodel._dyn_def
        return g

    def keep_last_all(self):
        """GPI_HDP.py:460-466: drop every cluster's history but its first and last entries."""
        for ld in range(self.n_outputs):
            for gp in self.gpmodels[ld]:
                gp.reinit_LDS(save_last=True)
                gp.reinit_GP(save_last=True, save_index=True)
        self.__dict__.pop("_pools", None)       # the persistent chains of the online step held the dropped history

    def save_swgp(self, st):
        """GPI_HDP.py:3946-3950: keep_last_all, everything to host memory, pickle.  ``load_swgp`` (or pickle.load) brings it back
        onto the default device."""
        import pickle
        self.keep_last_all()
        with open(st, "wb") as f:
            pickle.dump(self, f)

    @staticmethod
    def load_swgp(st):
        import pickle
        with open(st, "rb") as f:
            return pickle.load(f)

    def __getstate__(self):
        # (the time-warp fitters are rebuilt on load: their warm-start controls are not part of the saved model)
        d = {k: v for k, v in self.__dict__.items() if k not in ("_pools", "_lin_memo", "_warp_cache_full", "wp_sys")}
        return _to_device(d, "cpu")

    def __setstate__(self, d):
        _limit_host_threads()
        dev = self._default_device if torch.cuda.is_available() else "cpu"
        self.__dict__.update(_to_device(d, dev))
        self.device = dev
        self._warp_cache_full = {}
        for lead in self.gpmodels:
            for g in lead:
                g._set_device(dev)
                g.theta_source = self
        self.wp_sys = [[self.create_wp_sys_default() for _ in range(len(lead))] for lead in self.gpmodels]

    def selected_gpmodels(self):
        return list(range(sum(1 for g in self.gpmodels[0] if len(g.indexes) > 0)))

    # ------------------------------------------------------------------ conversions (GPI_HDP.py:4085-4094)
    def cond_to_torch(self, x):
        return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=f64).to(self.device)

    # ------------------------------------------------------------------ HDP pseudo-counts (host side)
    def init_global_params(self, M):
        self.rho = hdp_global.create_initrho(M)
        self.omega = (1.0 + self.gamma) * np.ones(M)
        self.transTheta, self.startTheta = self._calcThetaFull(np.ones((M, M)), np.ones(M), M + 1)

    def reinit_global_params(self, M, transStateCount, startStateCount):
        self.rho = hdp_global.create_initrho(M)
        self.omega = (1.0 + self.gamma) * np.ones(M)
        self.transTheta, self.startTheta = self._calcThetaFull(transStateCount, startStateCount, M=M)

    def _calcThetaFull(self, transStateCount, startStateCount, M=None, rho=None, kappa=None):
        M = self.M + 1 if M is None else M
        return hdp_global.calc_theta_full(_np(transStateCount), _np(startStateCount), M, self.rho if rho is None else rho,
                                          self.transAlpha, self.startAlpha, self.kappa if kappa is None else kappa)

    def find_optimum_rhoOmega(self):
        """GPI_HDP.py:2752-2828: E[log pi] of the current pseudo-counts -> (rho, omega) maximising the surrogate bound."""
        tt, st = _np(self.transTheta), _np(self.startTheta)
        e_log_pi = _digamma(tt) - np.log(np.sum(np.exp(_digamma(tt)), axis=1) + 1e-5)[:, None]
        start_e_log_pi = _digamma(st) - np.log(np.sum(np.exp(_digamma(st))) + 1e-5)
        try:
            rho, omega, _ = hdp_global.find_optimum_rho_omega(np.sum(e_log_pi, axis=0), self.startAlpha * start_e_log_pi, self.M + 1,
                                                              self.gamma, self.transAlpha, self.kappa, self.rho, self.omega)
        except ValueError as err:                       # keep the current values, as the reference does
            print('***** Optim failed. Remain at cur val. ' + str(err))
            rho, omega = self.rho, self.omega
        return rho, omega

    def compute_trans_A(self, M):
        """log transition matrix from the Dirichlet pseudo-counts (GPI_HDP.py:3527-3535): the leading M x M block, each row
        normalised by digamma of the row sum over M + 1 columns (as many as the table has); a table one state short is
        padded with -inf."""
        tt = _np(self.transTheta)
        tp = _digamma(tt[:M, :M]) - _digamma(np.sum(tt[:M, :M + 1], axis=1))[:, None]
        if tp.shape[0] == M:
            return torch.as_tensor(tp, dtype=f64)
        out = np.full((M, M), -np.inf)
        out[:M - 1, :M - 1] = tp
        return torch.as_tensor(out, dtype=f64)

    def compute_trans_pi(self, M, pi):
        """GPI_HDP.py:3537-3543."""
        pi = torch.as_tensor(_np(pi), dtype=f64).reshape(-1)
        if pi.shape[0] == M:
            return pi
        out = torch.full((M,), -np.inf, dtype=f64)
        out[:M - 1] = pi
        return out

    # ------------------------------------------------------------------ assignment tail
    def LogLik(self, logSoftEv, axis=1):
        """GPI_HDP.py:632-661: row-wise (axis=1) max-normalisation on the device; other axes / NumPy inputs on the host."""
        if torch.is_tensor(logSoftEv) and logSoftEv.is_cuda and logSoftEv.dim() == 2 and axis == 1:
            return ops.loglik_rows(logSoftEv.contiguous())
        x = torch.as_tensor(_np(logSoftEv), dtype=f64)
        c = torch.max(x, dim=axis)[0]
        if bool(torch.any(torch.isinf(c))):
            return x, c
        return x - c.unsqueeze(axis), c

    def _safe_exp(self, x):
        """GPI_HDP.py:338-350: a ONE-HOT ARG-MAX, not an exponential (2-D: per row; 3-D: over the flattened K x K)."""
        flat = x.reshape(x.shape[0], -1)
        y = torch.zeros_like(flat)
        y.scatter_(1, flat.argmax(dim=-1, keepdim=True), 1.0)
        return y.reshape_as(x)

    def weight_mean(self, q, snr=None):
        """GPI_HDP.py:685-701: leads combined with soft-max weights of their signal-to-noise ratio ([N,M,D] -> [N,M])."""
        if q.dim() > 2:
            w = self.snr_norm.to(q.device) if snr is None else torch.softmax(torch.max(snr, dim=1)[0], dim=1)
            return torch.einsum('ijk,ik->ij', q, w).contiguous()
        w = self.snr_norm if snr is None else torch.softmax(torch.max(snr, dim=1)[0], dim=1)
        frac = torch.sum(w, dim=0) / torch.sum(w)
        return torch.einsum('ij,j->i', q, frac.to(q.device))

    def _messages(self, pi, q, want_pair):
        q = self.cond_to_torch(q).contiguous()
        K = q.shape[1]
        return ops.hmm_messages(q, self.compute_trans_pi(K, pi).to(self.device), self.compute_trans_A(K).to(self.device),
                                want_pair=want_pair)

    def forward(self, pi=None, trans_A=None, q=None):
        """GPI_HDP.py:3546-3610, full recursion (the transition matrix always comes from the pseudo-counts, as there)."""
        fmsg, marg, _, _ = self._messages(pi, q, False)
        return fmsg, marg

    def backward(self, trans_A=None, q=None, margprob=None):
        """GPI_HDP.py:3612-3649."""
        pi0 = torch.zeros(q.shape[1], dtype=f64)
        return self._messages(pi0, q, False)[2]

    def coupled_state_coef(self, alpha=None, beta=None, trans_A=None, q=None, margprobs=None):
        """GPI_HDP.py:3651-3700 (recomputed from q: the device kernel produces messages and pair terms together)."""
        pi0 = torch.zeros(q.shape[1], dtype=f64)
        return self._messages(pi0, q, True)[3]

    # ------------------------------------------------------------------ signal-to-noise weights (multi-lead only)
    def compute_snr(self, y_lead, gp):
        """GPI_HDP.py:732-748: per-segment SNR against the smoothed state the segment would read.  With one lead the
        weights are identically 1 (soft-max over a single lead), so nothing is computed."""
        n = y_lead.shape[0]
        if not self.use_snr or self.n_outputs == 1:
            return torch.ones(n, dtype=f64, device=self.device)
        idx = np.asarray(gp.indexes, dtype=np.int64)
        pos = np.searchsorted(idx, np.arange(n), side="right") - 1            # find_closest_lower(t): idx-1 if idx else 0
        j = np.minimum(np.maximum(np.maximum(pos, 0), 1), len(gp.f_star_sm) - 1)
        F = gp._S("f_star_sm")[torch.as_tensor(j, device=self.device)][..., 0]
        eps = torch.finfo(f64).eps
        return 10.0 * torch.log10((torch.sum(F ** 2, -1) + eps) / (torch.sum((F - y_lead) ** 2, -1) + eps))

    def normalize_snr(self, snr):
        return torch.softmax(torch.max(snr, dim=1)[0], dim=1)

    # ------------------------------------------------------------------ hot-path wrappers
    def reload_model_from_labels(self, x_trains, y_trains, labels, M, warp=False):
        """GPI_HDP.py:3952-4035: rebuild every cluster model from given labels."""
        if warp:
            raise NotImplementedError("warping is not part of this build")
        y = self.cond_to_torch(y_trains)
        x = self.cond_to_torch(x_trains)
        assert y.shape[2] == self.n_outputs
        N = y.shape[0]
        labels = np.asarray(_np(labels), dtype=np.int64).reshape(-1)
        self.M, self.T = int(M), N
        self.x_basis = [self.x_basis_ini] * self.M
        self.model_type = [self.model_type_def] * self.M
        self.y_train = self.y = y
        self.x_train = x
        resp = torch.zeros((N, M), dtype=f64)
        resp[torch.arange(N), torch.as_tensor(labels)] = 1.0
        respPair = torch.zeros((N, M, M), dtype=f64)
        respPair[np.arange(N - 1), labels[:-1], labels[1:]] = 1.0
        q = torch.zeros((N, M, self.n_outputs), dtype=f64, device=self.device)
        q_lat = torch.zeros_like(q)
        snr = torch.zeros_like(q)
        # one fresh default model per (lead, class) (= the reference's deep copy of model 0 re-initialised, reinit_LDS / reinit_GP);
        # the D x M chains are independent: they run side by side (chain_batch.py)
        self.gpmodels = [[self.create_gp_default() for _ in range(M)] for _ in range(self.n_outputs)]
        outs = iter(self._passes(x, y, [(self.gpmodels[ld][m], y[:, :, [ld]], resp[:, m]) for ld in range(self.n_outputs) for m in range(M)]))
        for ld in range(self.n_outputs):
            for m in range(M):
                out = next(outs)
                if out is not None:
                    q[:, m, ld], q_lat[:, m, ld] = out
                snr[:, m, ld] = self.compute_snr(y[:, :, ld], self.gpmodels[ld][m])
        self.q.append(q)
        startStateCount, transStateCount = resp[0].numpy().copy(), torch.sum(respPair, dim=0).numpy()
        per_group = resp.sum(dim=0)
        print("Group responsability estimated: " + str(per_group.numpy().astype(np.int64)), flush=True)
        self.reinit_global_params(M, transStateCount, startStateCount)
        for _ in range(2):
            self.transTheta, self.startTheta = self._calcThetaFull(transStateCount, startStateCount, M + 1)
            self.rho, self.omega = self.find_optimum_rhoOmega()
        tt = _np(self.transTheta)
        self.trans_A = torch.as_tensor(_digamma(tt[:M, :M]) - np.log(np.sum(np.exp(_digamma(tt[:M, :M + 1])), axis=1) + 1e-5)[:, None])
        self.resp_assigned.append(torch.where(resp == 1.0)[1])
        self.q_last, self.q_lat_last, self.snr_last = q, q_lat, snr
        self.startStateCount_last, self.transStateCount_last = startStateCount, transStateCount
        self.resp_last, self.respPair_last = resp, respPair
        self.snr_norm = self.normalize_snr(snr)
        wq = self.weight_mean(q, snr)
        self.f_ind_old = torch.zeros(M, dtype=torch.int64)
        for m in range(M):
            ind = torch.as_tensor(self.gpmodels[0][m].indexes, device=self.device)
            if ind.numel():
                self.f_ind_old[m] = int(ind[torch.argmax(wq[ind, m])])

    def frozen_scores(self, x, y):
        """q[N, M, D]: GPI_model.log_sq_error(x_n, y_n[:, ld], i=-1) of every (segment, cluster, lead) against the LAST state
        of each frozen model (GPI_HDP.py:2981-2985).  Segments on the basis grid share the covariance Sigma_m (pred_dist
        short-circuits, GPI.py:467-468): one factorisation per cluster; all others go through the per-pair kernels in ONE
        launch per lead."""
        N, Ts = x.shape[0], x.shape[1]
        T = self.x_basis_ini.shape[0]
        X = (x[..., 0] if x.dim() == 3 else x).contiguous()
        xb = self.cond_to_torch(self.x_basis_ini).reshape(-1)
        on_basis = torch.zeros(N, dtype=torch.bool, device=self.device) if Ts != T else (X == xb.unsqueeze(0)).all(dim=1)
        rows_b = torch.nonzero(on_basis).reshape(-1)
        rows_p = torch.nonzero(~on_basis).reshape(-1)
        q = torch.zeros((N, self.M, self.n_outputs), dtype=f64, device=self.device)
        for ld in range(self.n_outputs):
            Y = y[:, :, ld].contiguous()
            sel = [g._select(-1) for g in self.gpmodels[ld]]
            means = torch.stack([g._mean_of(ci, fi).reshape(-1) for g, (ci, fi) in zip(self.gpmodels[ld], sel)]).contiguous()
            Sig = torch.stack([g.Sigma[ci] for g, (ci, _) in zip(self.gpmodels[ld], sel)]).contiguous()
            if rows_b.numel():
                Yb = Y[rows_b].contiguous()
                nb = Yb.shape[0]
                for m in range(self.M):
                    items = ops.build_items([m], [0.0], [nb])
                    quad, _, info = ops.score_groups(Yb, means, Sig, *items)
                    ops.raise_on_info(info, "cluster_new_batch")
                    q[rows_b, m, ld] = -0.5 * quad - 0.5 * T * ops.LOG2PI
            if rows_p.numel():
                theta = np.array([g.gp.kernel.params() for g in self.gpmodels[ld]])
                plan = ops.PairsPlan(T, Ts, theta, device=self.device).update(xb, means, Sig)
                ops.raise_on_info(plan.info, "pred_dist")
                score, info = plan.score(X[rows_p].contiguous(), Y[rows_p].contiguous())
                ops.raise_on_info(info, "cluster_new_batch")
                q[rows_p, :, ld] = score
        return q

    def cluster_new_batch(self, x_trains, y_trains, learning=False, it_limit=None, warp=False):
        """GPI_HDP.py:2975-3003 (learning=False): classify a batch with the frozen models; returns the label tensor."""
        x = self.cond_to_torch(x_trains)
        y = self.cond_to_torch(y_trains)
        if learning:
            if warp:
                raise NotImplementedError("warping is not part of this build")
            return self.cluster_new_batch_learning(x, y, it_limit=it_limit)
        M = self.M
        q = self.frozen_scores(x, y)
        snr = torch.stack([torch.stack([self.compute_snr(y[:, :, ld], self.gpmodels[ld][m]) for ld in range(self.n_outputs)], dim=-1)
                           for m in range(M)], dim=1)
        tt, st = _np(self.transTheta), _np(self.startTheta)
        startPi = _digamma(st[:M]) - _digamma(np.sum(st[:M + 1]) + 1e-5)
        q_norm, _ = self.LogLik(self.weight_mean(q, snr))
        fmsg, marg, bmsg, _ = self._messages(startPi, q_norm, False)
        self.last_messages = (fmsg, marg, bmsg)
        return ops.assign(fmsg, bmsg).cpu()           # = torch.where(_safe_exp(LogLik(log(alpha beta))) == 1)[1]


def _to_device(o, dev, _seen=None):
    """Tensors of a (nested) structure of dicts / lists / tuples moved to `dev`; everything else as it is."""
    if torch.is_tensor(o):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: _to_device(v, dev) for k, v in o.items()}
    if isinstance(o, list):
        return [_to_device(v, dev) for v in o]
    if isinstance(o, tuple):
        return tuple(_to_device(v, dev) for v in o)
    return o


def _np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
