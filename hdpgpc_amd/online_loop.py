"""The online variational step of GPI_HDP (hdpgpc/hdpgpc/GPI_HDP.py:1906-2208 ``include_sample`` with
``variational_local_terms`` :586, ``estimate_new`` :2830, ``reorder`` :1091) - one segment at a time: score it under every
cluster, compare "open a new cluster" against "give it to an existing one" through the one-sample bound, commit.

Host orchestration over the same kernels as the offline loop; per beat the device work is
* a5: the segment's score under the last state of every cluster (ONE shared-covariance launch for all clusters when the
  segment sits on the basis grid, the per-pair kernel otherwise),
* a8: the latent-transition scores of every cluster's members (kept per model until the model changes - only the cluster that
  absorbs the beat changes),
* a9 through the bound, the Kalman / MNIW update of the candidates (GPI_model.posterior_weighted, include_weighted_sample,
  backwards_pair, bayesian_new_params) and the switching-variable messages over the whole history (ops.hmm_messages).
``with_warp=True`` (the time-warp fit of every beat) and ``classify=True`` (no caller in the reference) are not built.
"""
import numpy as np
import torch
from scipy.special import digamma as _digamma

from . import ops

f64 = torch.float64


class OnlineLoop:
    """Mixin of GPI_HDP: the streaming (online) variational step."""

    def variational_local_terms(self, q, transTheta=None, startTheta=None, liks=None, classify=False):
        """GPI_HDP.py:586-630: hard state / pair assignments of the whole history from the score matrix q [T, K, D] (device).
        Returns (resp one-hot [T,K] host, log resp of the LAST row (numpy), respPair one-hot [T,K,K] host, None)."""
        M = self.M
        startTheta = self.startTheta if startTheta is None else startTheta
        st = _np(startTheta)
        startPi = torch.as_tensor(_digamma(st[:M]) - _digamma(np.sum(st[:M + 1])), dtype=f64)
        q = q.clone()
        if liks is not None:
            q[-1] = q[-1] + torch.as_tensor(np.asarray(liks, dtype=np.float64), device=q.device)[:, None]
        q_norm, _ = self.LogLik(self.weight_mean(q).contiguous())
        fmsg, _, bmsg, pair = self._messages(startPi, q_norm, True)
        resp, respPair = self._one_hot_tables(fmsg, bmsg, pair)
        last = torch.log(fmsg[-1] * bmsg[-1]).cpu().numpy()
        return resp, last - np.max(last) if np.isfinite(np.max(last)) else last, respPair, None

    def estimate_new(self, t, gpmodel, x_train, y, h=1.0):
        """GPI_HDP.py:2830-2842: the segment's score under the state the model would have after absorbing it."""
        mean_, cov_, C_, Sigma_ = gpmodel.smoother_weighted(x_train, y, h)
        return gpmodel.log_sq_error(x_train, y, mean=mean_[-1], cov=cov_[-1], C=C_[-1], Sigma=Sigma_[-1], i=-1,
                                    first=len(gpmodel.indexes) == 1)

    def reorder(self, resp, respPair, q, q_lat):
        """GPI_HDP.py:1091-1110: clusters sorted by size, largest first (tables, score matrices and the model list)."""
        order = torch.argsort(torch.sum(resp, dim=0), descending=True)
        resp = resp[:, order]
        respPair = respPair[:, order, :][:, :, order]
        od = order.to(q.device)
        q, q_lat = q[:, od], q_lat[:, od]
        for ld in range(self.n_outputs):
            self.gpmodels[ld] = [self.gpmodels[ld][int(order[i])] for i in range(self.M)]
        return resp, respPair, q, q_lat, order

    def _last_scores(self, x, y, ld):
        """log_sq_error(x, y, i=-1) of the segment under every cluster of lead ld (GPI_HDP.py:1973) -> [M] device.
        On the basis grid pred_dist short-circuits (GPI.py:467-468) and the M evaluations are one launch."""
        models = self.gpmodels[ld]
        xb = models[0].x_basis
        if x.shape == xb.shape and bool(torch.equal(x, xb)) and not any(g.rank1_scale_factor() is not None for g in models):
            sel = [g._select(-1) for g in models]
            means = torch.stack([g._mean_of(ci, fi).reshape(-1) for g, (ci, fi) in zip(models, sel)]).contiguous()
            Sig = torch.stack([g.Sigma[ci] for g, (ci, _) in zip(models, sel)]).contiguous()
            M, T = means.shape
            Y = y[:, ld].reshape(1, T).expand(M, T).contiguous()
            quad, _, info = ops.score_each(Y, means, Sig, np.arange(M, dtype=np.int32), np.arange(M, dtype=np.int32))
            ops.raise_on_info(info, "log_sq_error")
            return -0.5 * quad - 0.5 * T * ops.LOG2PI
        return torch.stack([g.log_sq_error(x, y[:, [ld]], i=-1) for g in models])

    def include_sample(self, x_train, y, with_warp=True, force_model=None, minibatch=0, classify=False):
        """GPI_HDP.py:1906-2208."""
        if with_warp:
            # The reference itself cannot run this path: with one cluster (the second beat of any run) compute_warp_y's greedy
            # branch takes torch.max of an empty tensor (GPI_HDP.py:3313, liks[:-1] with M = 1) and raises RuntimeError -
            # hdpgpc/tests/test_online_warp.py stops there (tests/golden/make_golden.py `onlinew` reproduces it).  There is no
            # reference behaviour to mirror; the batched warp fit itself (hgp_warp_batch_f64) serves include_batch(warp=True).
            raise NotImplementedError("include_sample(with_warp=True): the reference's own path raises at its second beat "
                                      "(GPI_HDP.py:3313); not built")
        if classify:
            raise NotImplementedError("include_sample(classify=True) has no caller in the reference and is not built")
        if self.n_outputs != 1:
            raise NotImplementedError("include_sample: one lead (the reference's reorder() aliases the per-lead model lists)")
        D, dev = self.n_outputs, self.device
        t = self.T
        self.T = self.T + 1
        T_all = self.T
        self.snr_norm = torch.ones((T_all, D), dtype=f64, device=dev)
        M = self.M
        y = self.cond_to_torch(y).reshape(-1, D)
        x = self.cond_to_torch(x_train).reshape(-1, 1)
        liks = np.zeros(M + 1)
        self.y.append(y)
        self.x_train.append(x)
        n_hist = torch.empty((T_all, 0))                                   # compute_q_lat_all only reads the history length
        q_aux = torch.full((T_all, M + 1, D), -np.inf, dtype=f64, device=dev)
        q_lat = torch.zeros((T_all, M + 1, D), dtype=f64, device=dev)
        if t > 0:
            prev = self.q[-1]
            q_aux[:-1, :prev.shape[1], :] = prev
        for ld in range(D):
            for m, gp in enumerate(self.gpmodels[ld]):
                q_lat[:, m, ld] = gp.compute_q_lat_all(n_hist, h_ini=1.0)
            q_aux[-1, :M, ld] = self._last_scores(x, y, ld)
        q_all = elbo = 0.0
        if t > 0:
            resp, _, respPair, _ = self.variational_local_terms(q_aux, self.transTheta, self.startTheta)
            q_all, elbo = self.compute_q_elbo(resp[:-1, :-1], respPair[:-1, :-1, :-1], self.weight_mean(q_aux)[:-1, :-1],
                                              self.weight_mean(q_lat)[:-1, :-1], self.gpmodels, self.M, snr='saved', post=False,
                                              one_sample=True, verb=self.verbose)
            # how well does each existing cluster explain the beat?  candidates are tried best first; the worst one lends its
            # kernel and priors to the would-be new cluster
            q_ord = torch.argsort(self.weight_mean(q_aux)[-1, :-1].cpu(), descending=True)
            m_w = int(q_ord[-1])
            q_prev, q_lat_prev = q_aux.clone(), q_lat.clone()
            for ld in range(D):
                prov = self.gpmodel_deepcopy(self.gpmodels[ld][m_w])
                prov.reinit_GP(save_last=False)
                prov.reinit_LDS(save_last=False)
                q_prev[-1, -1, ld] = self.estimate_new(t, prov, x, y[:, [ld]], h=1.0) + liks[-1]
                prov.include_weighted_sample(t, x, x, y[:, [ld]], 1.0)
                self.gpmodels[ld].append(prov)
                q_lat_prev[:, -1, ld] = prov.compute_q_lat_all(n_hist, h_ini=1.0)
            resp_prev, rl_prev, respPair_prev, _ = self.variational_local_terms(q_prev, self.transTheta, self.startTheta, liks)
            q_prev_post, elbo_prev_post = self.compute_q_elbo(resp_prev, respPair_prev, self.weight_mean(q_prev), self.weight_mean(q_lat_prev),
                                                              self.gpmodels, self.M, snr='saved', one_sample=True, post=True,
                                                              verb=self.verbose)
            elbo_prev_post -= elbo
            q_prev_post -= q_all
            for ld in range(D):
                self.gpmodels[ld].pop()
            self.M = M
            if int(torch.argmax(q_prev[-1])) == self.M:               # the new cluster scores the beat best: is it worth it?
                q_post, q_lat_post = q_aux.clone(), q_lat.clone()
                for m in q_ord.tolist():
                    saved = [self.gpmodels[ld][m] for ld in range(D)]
                    for ld in range(D):
                        cand = self.gpmodel_deepcopy(self.gpmodels[ld][m])
                        q_post[-1, m, ld] = self.estimate_new(t, cand, x, y[:, [ld]], h=1.0) + liks[m]
                        cand.include_weighted_sample(t, x, x, y[:, [ld]], 1.0)
                        self.gpmodels[ld][m] = cand
                        cand.backwards_pair(1.0)
                        cand.bayesian_new_params(1.0)
                        q_lat_post[:, m, ld] = cand.compute_q_lat_all(n_hist, h_ini=1.0)
                    resp_post, rl_post, respPair_post, _ = self.variational_local_terms(q_post, self.transTheta, self.startTheta, liks)
                    q_bas_post, elbo_bas_post = self.compute_q_elbo(resp_post[:, :-1], respPair_post[:, :-1, :-1],
                                                                    self.weight_mean(q_post)[:, :-1], self.weight_mean(q_lat_post)[:, :-1],
                                                                    self.gpmodels, self.M, snr='saved', post=False, one_sample=True,
                                                                    verb=self.verbose)
                    elbo_bas_post -= elbo
                    q_bas_post -= q_all
                    for ld in range(D):
                        self.gpmodels[ld][m] = saved[ld]
                    if q_bas_post + elbo_bas_post > q_prev_post + elbo_prev_post:
                        resp, resplog, respPair = resp_post, rl_post, respPair_post
                        q_chos, q_lat_chos = q_post, q_lat_post
                        break
                    q_chos, q_lat_chos = q_prev, q_lat_prev
                    resp, resplog, respPair = resp_prev, rl_prev, respPair_prev
            else:
                q_chos, q_lat_chos = q_aux, q_lat
                resp, resplog, respPair, _ = self.variational_local_terms(q_chos, self.transTheta, self.startTheta, liks)
        else:
            q_chos, q_lat_chos = q_aux, q_lat
            resp, resplog, respPair, _ = self.variational_local_terms(q_aux, self.transTheta, self.startTheta, liks)

        resp_mod = resp[-1].numpy()                      # a view: edits below land in the table, as in the reference
        model = int(np.argmax(resp_mod))
        if self.max_models is not None and model >= self.max_models:
            force_model = int(np.argmax(resplog[:-1]))
        if force_model is not None:
            resp_mod[:] = 0.0
            resp_mod[force_model] = 1.0
            model = int(force_model)
        order = torch.arange(resp.shape[1])
        if model == self.M:
            self._log("Birth of new model: ", self.M + 1)
            self.M = M = self.M + 1
            for ld in range(D):
                self.gpmodels[ld].append(self.create_gp_default())
            self.x_basis.append(self.x_basis_ini)
            resp, respPair, q_chos, q_lat_chos, order = self.reorder(resp, respPair, q_chos, q_lat_chos)
            startStateCount, transStateCount = resp[0].numpy().copy(), torch.sum(respPair, dim=0).numpy()
        else:
            if force_model is None:
                resp, respPair, q_chos, q_lat_chos, order = self.reorder(resp, respPair, q_chos, q_lat_chos)
            startStateCount, transStateCount = resp[0, :M].numpy().copy(), torch.sum(respPair[:, :M, :M], dim=0).numpy()
        if M > 2:
            self.reinit_global_params(M - 1, transStateCount, startStateCount)
        if M >= 2:
            for _ in range(4):
                self.transTheta, self.startTheta = self._calcThetaFull(transStateCount, startStateCount, M)
                self.rho, self.omega = self.find_optimum_rhoOmega()
        tt = _np(self.transTheta)
        self.trans_A = torch.as_tensor(_digamma(tt[:M, :M]) - np.log(np.sum(np.exp(_digamma(tt[:M, :M + 1])), axis=1))[:, None])
        resp_mod = resp[-1].numpy()
        model = int(np.argmax(resp_mod))
        if force_model is not None:
            model = int(force_model)
            resp_mod[:] = 0.0
            resp_mod[model] = 1.0
            q_chos[-1, model] = torch.max(q_chos[-1])
            q_lat_chos[-1, model] = torch.max(q_lat_chos[-1])
            respPair[-1, model, :] = 0.0
            respPair[-1, :, model] = 0.0
            respPair[-1, model, model] = 1.0
        self.actual_state = model
        self._log("Main model chosen:", model + 1)
        for ld in range(D):
            for m in range(M):
                h = float(resp_mod[m])
                gp = self.gpmodels[ld][m]
                gp.include_weighted_sample(t, x, x, y[:, [ld]], h)
                if h > 0.9 and ld == 0:
                    row = y.reshape(1, -1, D)
                    self.y_train = row if self.y_train.numel() == 0 else torch.cat([self.y_train, row])
                gp.bayesian_new_params(h, model_type=self.model_type_def)
        if self.verbose:
            self.compute_q_elbo(resp[:, :M], respPair[:, :M, :M], self.weight_mean(q_chos)[:, :M], self.weight_mean(q_lat_chos)[:, :M],
                                self.gpmodels, self.M, snr='saved', post=False, one_sample=True)
        self.resp_assigned.append(torch.argmax(resp, dim=1))
        self.q.append(q_chos)


def _np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
