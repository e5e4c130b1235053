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
        od = ops.to_dev(order, torch.int64, q.device)
        q, q_lat = q.index_select(1, od), q_lat.index_select(1, od)
        for ld in range(self.n_outputs):
            self.gpmodels[ld] = [self.gpmodels[ld][int(order[i])] for i in range(self.M)]
        return resp, respPair, q, q_lat, order

    def _online_pool(self, ld, x):
        """The persistent chains of lead ld's clusters (online_chain.OnlinePool), or None when the beat does not sit on the
        basis grid (the chain step covers the shared-grid case; other grids take the one-by-one path)."""
        xb = self.gpmodels[ld][0].x_basis if self.gpmodels[ld] else None
        if xb is None or x.shape != xb.shape or not bool(torch.equal(x, xb)) or ops.env_flag("HGP_ONLINE_EAGER"):
            return None
        pools = self.__dict__.setdefault("_pools", {})
        if ld not in pools:
            from .online_chain import OnlinePool
            pools[ld] = OnlinePool(xb.shape[0], self.device, self.annealing_def)
        pool = pools[ld]
        for g in self.gpmodels[ld]:
            if getattr(g, "_slot", None) is None or g._slot >= len(pool.slots) or pool.slots[g._slot].g is not g:
                if not pool.supports(g):
                    return None
                pool.adopt(g)
        return pool

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

    # ------------------------------------------------------------------ the local step for many score matrices at once
    def _local_terms_many(self, Qw, liks=None):
        """variational_local_terms (GPI_HDP.py:586-630) for a batch Qw [B, T, K] of lead-combined score matrices (device): one
        launch sequence for all of them (ops.hmm_local_terms), one host round trip.  Returns host arrays
        (labels [B, T], pair_first [B, T] - flat index into the K x K pair table, last [B, K] - log resp of the newest row,
        max-shifted as the reference's)."""
        M, dev = self.M, self.device
        B, N, K = Qw.shape
        st = _np(self.startTheta)
        startPi = torch.as_tensor(_digamma(st[:M]) - _digamma(np.sum(st[:M + 1])), dtype=f64)
        if liks is not None:
            Qw = Qw.clone()
            Qw[:, -1, :] += ops.to_dev(np.asarray(liks, dtype=np.float64), f64, dev)[None, :]
        labels, pairs, last = ops.hmm_local_terms(Qw.contiguous(), ops.to_dev(self.compute_trans_pi(K, startPi), f64, dev),
                                                  ops.to_dev(self.compute_trans_A(K), f64, dev))
        flat = torch.cat([labels.reshape(-1).to(f64), pairs.reshape(-1).to(f64), last.reshape(-1)]).cpu().numpy()
        lab = flat[:B * N].astype(np.int64).reshape(B, N)
        prs = flat[B * N:2 * B * N].astype(np.int64).reshape(B, N)
        last = flat[2 * B * N:].reshape(B, K).copy()
        for b in range(B):
            mx = np.max(last[b])
            if np.isfinite(mx):
                last[b] -= mx
        return lab, prs, last, labels

    @staticmethod
    def _tables(labels, pairs, K):
        """One-hot host tables resp [T, K], respPair [T, K, K] of a hard assignment (what _one_hot_tables builds)."""
        N = labels.shape[0]
        resp = torch.zeros((N, K), dtype=f64)
        resp[torch.arange(N), torch.as_tensor(labels)] = 1.0
        respPair = torch.zeros((N, K * K), dtype=f64)
        respPair[torch.arange(N), torch.as_tensor(pairs)] = 1.0
        return resp, respPair.reshape(N, K, K)

    def _bound_from_labels(self, labels, pairs, K, n_rows, n_cols, q_sum, lat_sum, lds, post):
        """compute_q_elbo(resp[:n_rows, :n_cols], respPair[:n_rows, :n_cols, :n_cols], ..., one_sample=True) (GPI_HDP.py:1796-1836)
        from the hard assignment itself (labels / pair_first of ONE score matrix, host ints), the two device sums over the
        assigned entries and the clusters' MNIW parameter likelihoods lds [n_cols] - no [T, K, K] table is built."""
        lab, prs = labels[:n_rows], pairs[:n_rows]
        # the HDP terms depend on the hard assignment only: candidates that move a score without moving an assignment share them
        memo = self.__dict__.setdefault("_lin_memo", {})
        key = (lab.tobytes(), prs.tobytes(), n_cols, post, id(self.rho), id(self.transTheta))
        elbo_lin = memo.get(key)
        if elbo_lin is None:
            start = np.zeros(n_cols)
            if lab[0] < n_cols:
                start[lab[0]] = 1.0
            i, j = prs // K, prs % K
            ok = (i < n_cols) & (j < n_cols)
            trans = np.zeros((n_cols, n_cols))
            np.add.at(trans, (i[ok], j[ok]), 1.0)
            elbo_lin = memo[key] = self._elbo_linears_counts(start, trans, n_cols, post=post, one_sample=True) * 1
        sums = [float(v) for v in np.bincount(lab[lab < n_cols], minlength=n_cols)]
        tot = sum(sums)
        elbo_lds = 0.0
        for k in range(n_cols):
            if sums[k] > 0:
                elbo_lds += lds[k] * (sums[k] / tot)
        elbo_lds = elbo_lds * 1.0                    # one lead: its share of the n_points = 1 sample is 1 (frac of GPI_HDP.py:1820-1826)
        q_bas = q_sum * self.static_factor
        elbo_latent = lat_sum * self.dynamic_factor
        return q_bas, (elbo_lin + elbo_lds + elbo_latent) if self.hmm_switch else elbo_latent

    def _eager_candidates(self, ld, t, x, y, q_lat, n_hist):
        """The candidates one by one (a copy of every cluster takes the beat: GPI_HDP.py:2040-2056) - the path for beats off the
        basis grid; same return values as online_chain.OnlinePool.candidates, in CLUSTER order."""
        est, cols, lds = [], [], []
        for m, g in enumerate(self.gpmodels[ld]):
            cand = self.gpmodel_deepcopy(g)
            est.append(self.estimate_new(t, cand, x, y[:, [ld]], h=1.0))
            cand.include_weighted_sample(t, x, x, y[:, [ld]], 1.0)
            cand.backwards_pair(1.0)
            cand.bayesian_new_params(1.0)
            cols.append(cand.compute_q_lat_all(n_hist, h_ini=1.0))
            lds.append(cand.lds_param_likelihood_value())
        return torch.stack(est), torch.stack(cols, dim=1), lds

    def include_sample(self, x_train, y, with_warp=True, force_model=None, minibatch=0, classify=False):
        """GPI_HDP.py:1906-2208.  Same decisions in the same order as the reference; what it evaluates one candidate after the
        other - the clusters with the beat added, and the local step + bound of every candidate's score table - is computed side
        by side (online_chain.OnlinePool.candidates, _local_terms_many) before the accept / reject walk."""
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
        if not self.bayesian_params or minibatch:
            raise NotImplementedError("include_sample: only the Bayesian one-step parameter update is built (bayesian_params=True, "
                                      "minibatch=0); the reference's new_params_weighted path (GPI_HDP.py:2195) is not")
        D, dev, ld = self.n_outputs, self.device, 0
        _tick("outside")
        self._lin_memo = {}
        t = self.T
        self.T = self.T + 1
        T_all = self.T
        self.snr_norm = torch.ones((T_all, D), dtype=f64, device=dev)
        M = self.M
        K = M + 1
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
        mods = self.gpmodels[ld]
        pool = self._online_pool(ld, x) if t > 0 else None                 # the clusters as persistent chains (online_chain.py)
        if M > 0:
            q_lat[:, :M, ld] = torch.stack([gp.compute_q_lat_all(n_hist, h_ini=1.0) for gp in mods], dim=1)
        info0 = None
        if pool is not None:
            slot_of = [g._slot for g in mods]
            sl_dev = ops.to_dev(slot_of, torch.int64, dev)
            sc, info0 = pool.begin_beat(y[:, ld])
            q_aux[-1, :M, ld] = sc[sl_dev]
        elif M > 0:
            q_aux[-1, :M, ld] = self._last_scores(x, y, ld)
        _tick("scores")
        if t > 0:
            # how well does each existing cluster explain the beat?  candidates are tried best first; the worst one lends its
            # kernel and priors to the would-be new cluster
            last_row = self.weight_mean(q_aux)[-1, :-1]
            if info0 is not None:                                          # the scores' LAPACK status rides the same round trip
                host = torch.cat([last_row, info0.to(f64)]).cpu()
                if bool(host[M:].any()):
                    ops.raise_on_info(info0, "log_sq_error")
                last_row = host[:M]
            q_ord = torch.argsort(last_row.cpu(), descending=True)
            m_w = int(q_ord[-1])
            q_prev, q_lat_prev = q_aux.clone(), q_lat.clone()
            prov = self.gpmodel_deepcopy(self.gpmodels[ld][m_w])
            prov.reinit_GP(save_last=False)
            prov.reinit_LDS(save_last=False)
            prov._defer_checks = True                                # its LAPACK statuses are read together, below
            q_prev[-1, -1, ld] = prov.estimate_new_and_include(t, x, y[:, [ld]]) + liks[-1]
            birth_best = int(torch.argmax(q_prev[-1])) == M          # the new cluster scores the beat best: is it worth it?
            order = q_ord.tolist() if birth_best else []
            lds_cur = [g.lds_param_likelihood_value() for g in self.gpmodels[ld]]
            lds_cand = lds_prov = None
            if order:
                if pool is not None:
                    # the new cluster's own latent-transition score and parameter likelihoods ride the candidates' batched calls
                    m_of = np.argsort(slot_of)                         # cluster index of every slot
                    res = pool.candidates(t, q_lat[:, ops.to_dev(m_of, torch.int64, dev), ld].contiguous(), [mods[m].indexes for m in m_of],
                                          extra=prov)
                    est, cols, lds_s = res[:3]
                    est, cols, lds_cand = est[sl_dev], cols[:, sl_dev], [float(lds_s[c]) for c in slot_of]
                    if len(res) == 5:
                        q_lat_prev[t, -1, ld] = res[3]
                        lds_prov = res[4]
                else:
                    est, cols, lds_cand = self._eager_candidates(ld, t, x, y, q_lat, n_hist)
            if lds_prov is None:
                q_lat_prev[:, -1, ld] = prov.compute_q_lat_all(n_hist, h_ini=1.0)
                lds_prov = float(prov.return_LDS_param_likelihood())
            prov._defer_checks = False
            prov._check_pending()
            _tick("candidates")
            # score tables of every evaluation of this beat: [0] current clusters, [1] with the new cluster, [2 + r] the r best
            # clusters tried so far with the beat added (the reference's q_post is cumulative over its loop, GPI_HDP.py:2040-2075)
            Qb, Lb = self.weight_mean(q_aux), self.weight_mean(q_lat)
            Qs, Ls = [Qb, self.weight_mean(q_prev)], [Lb, self.weight_mean(q_lat_prev)]
            if order:
                R = len(order)
                od = ops.to_dev(order, torch.int64, dev)
                tried = torch.zeros((R, K), dtype=torch.bool, device=dev)             # tried[r, m]: cluster m is among the r + 1 best
                tried[:, od] = torch.tril(torch.ones((R, R), dtype=torch.bool, device=dev))
                new_last = Qb[-1].clone()
                new_last[:M] = est + ops.to_dev(liks[:M], f64, dev)
                Qc = Qb.unsqueeze(0).repeat(R, 1, 1)
                Qc[:, -1, :] = torch.where(tried, new_last[None, :], Qb[-1][None, :])
                cols_full = Lb.clone()
                cols_full[:, :M] = cols
                Lc = torch.where(tried[:, None, :], cols_full[None], Lb[None])
                Qs += list(Qc.unbind(0))
                Ls += list(Lc.unbind(0))
            Qall, Lall = torch.stack(Qs), torch.stack(Ls)
            B = Qall.shape[0]
            lab, prs, last, lab_dev = self._local_terms_many(Qall, liks)
            _tick("local_terms")
            # the sums of the bound over the assigned entries (all rows; all but the newest for the current clusters)
            n_cols = torch.full((B, 1), M, dtype=torch.int64, device=dev)
            n_cols[1] = M + 1
            valid = lab_dev < n_cols
            idx = torch.clamp(lab_dev, max=M)[..., None]
            zero = torch.zeros((), dtype=f64, device=dev)
            gq = torch.where(valid, Qall.gather(2, idx)[..., 0], zero)
            gl = torch.where(valid, Lall.gather(2, idx)[..., 0], zero)
            sums = torch.stack([torch.sum(gq[0, :-1]), torch.sum(gl[0, :-1])] + [v for b in range(1, B) for v in (torch.sum(gq[b]), torch.sum(gl[b]))])
            sums = sums.cpu().numpy().reshape(B, 2)
            q_all, elbo = self._bound_from_labels(lab[0], prs[0], K, T_all - 1, M, sums[0, 0], sums[0, 1], lds_cur, post=False)
            q_prev_post, elbo_prev_post = self._bound_from_labels(lab[1], prs[1], K, T_all, M + 1, sums[1, 0], sums[1, 1],
                                                                  lds_cur + [lds_prov], post=True)
            elbo_prev_post -= elbo
            q_prev_post -= q_all
            chosen = 0                                                  # index into Qs of the table that is kept
            if birth_best:
                chosen = 1
                for r, m in enumerate(order):
                    lds_r = list(lds_cur)
                    lds_r[m] = lds_cand[m]                              # only the cluster being tried is the candidate's (GPI_HDP.py:2071)
                    q_bas_post, elbo_bas_post = self._bound_from_labels(lab[2 + r], prs[2 + r], K, T_all, M, sums[2 + r, 0], sums[2 + r, 1],
                                                                        lds_r, post=False)
                    elbo_bas_post -= elbo
                    q_bas_post -= q_all
                    if q_bas_post + elbo_bas_post > q_prev_post + elbo_prev_post:
                        chosen = 2 + r
                        break
            resp, respPair = self._tables(lab[chosen], prs[chosen], K)
            _tick("bounds")
            resplog = last[chosen]
            q_chos, q_lat_chos = Qall[chosen].unsqueeze(-1).clone(), Lall[chosen].unsqueeze(-1).clone()
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
        _tick("reorder")
        # the members' update of this beat (GPI_HDP.py:2186-2196, after the global step there): nothing below up to that point
        # reads the cluster models, and nothing in the update reads the HDP parameters, so its launches go out first and run
        # under the host-side optimisation of (rho, omega)
        resp_mod = resp[-1].numpy()
        if force_model is not None:
            resp_mod = resp_mod.copy()
            resp_mod[:] = 0.0
            resp_mod[int(force_model)] = 1.0
        committed = None
        for m in range(M):
            h = float(resp_mod[m])
            gp = self.gpmodels[ld][m]
            chained = pool is not None and h == 1.0 and gp.N >= 1 and getattr(gp, "_slot", None) is not None
            if chained:               # the cluster's persistent chain takes the member step (Kalman + both MNIW updates)
                pool.commit(gp, t, x, y[:, [ld]])
                committed = pool
            else:
                gp.include_weighted_sample(t, x, x, y[:, [ld]], h)
            if h > 0.9:
                row = y.reshape(1, -1, D)
                self.y_train = row if self.y_train.numel() == 0 else torch.cat([self.y_train, row])
            if not chained:
                gp.bayesian_new_params(h, model_type=self.model_type_def)
        if M > 2:
            self.reinit_global_params(M - 1, transStateCount, startStateCount)
        if M >= 2:
            for _ in range(4):
                self.transTheta, self.startTheta = self._calcThetaFull(transStateCount, startStateCount, M)
                self.rho, self.omega = self.find_optimum_rhoOmega()
        _tick("rho_omega")
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
        if committed is not None:
            committed.finish_commit()
        _tick("commit")
        if self.verbose:
            self.compute_q_elbo(resp[:, :M], respPair[:, :M, :M], self.weight_mean(q_chos)[:, :M], self.weight_mean(q_lat_chos)[:, :M],
                                self.gpmodels, self.M, snr='saved', post=False, one_sample=True)
        self.resp_assigned.append(torch.argmax(resp, dim=1))
        self.q.append(q_chos)


def _np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)


# phase timing of include_sample (tools/time_online.py --phases): HGP_ONLINE_TIMING=1 synchronises at every phase boundary
TIMING = {}
_T_ON = ops.env_flag("HGP_ONLINE_TIMING")
_t_last = [0.0]


def _tick(name):
    if not _T_ON:
        return
    import time
    torch.cuda.synchronize()
    now = time.perf_counter()
    TIMING[name] = TIMING.get(name, 0.0) + now - _t_last[0]
    _t_last[0] = now
