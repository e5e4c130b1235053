"""The offline variational loop of GPI_HDP (hdpgpc/hdpgpc/GPI_HDP.py:805-943 ``include_batch`` and what it calls:
``refill`` :1076, ``refill_resp`` :1141, ``new_group`` :1112, ``remove_last_group`` :1133,
``variational_local_terms_batch`` :1170, ``estimate_q_first`` :1243, ``estimate_q_all`` :2844, ``compute_q_elbo`` :1796,
``full_LDS_elbo`` :1838, ``elbo_Linears`` :1025, ``redefine_default`` :1866, ``compute_snr_ini`` :715) - host
orchestration over the HIP kernels: every number that depends on the data comes from

* ``GPI_model.full_pass_weighted``  (member chain kernels + a6 / a8 scores),
* ``GPI_model.compute_sq_err_all``  (a6: one shared-covariance launch for the N scores of a one-member proposal),
* ``GPI_model.return_LDS_param_likelihood``  (a9),
* ``ops.loglik_rows`` / ``ops.hmm_messages`` / ``ops.assign``  (switching-variable messages and the hard assignment),

and the score matrices ``q``, ``q_lat``, ``snr`` ([N, M, n_outputs]) stay on the device between them.  What runs on the host
is the decision logic (which clusters changed, accept / reject a proposal, bookkeeping of the one-hot responsibilities - N x M
and N x M x M tables of zeros and ones) and the O(M) HDP terms (hdp_global.py).

The reference's loop is written as three long methods that repeat the same blocks (assign -> rebuild changed clusters -> ELBO
-> iterate); here those blocks are helpers (``_assign``, ``_rebuild``, ``_elbo_of``, ``_pick_representatives``).  Quirks of
the reference that change results are kept and marked ``# quirk``.  With ``warp=False`` the reference's 4-D ``y_trains_w`` is a
broadcast view of ``y_trains`` and its ``liks`` are zeros (GPI_HDP.py:3441-3446), so neither is materialised; ``warp=True`` warps every
segment onto every cluster's representative (``warp_batch_by_resp_amtgp_cached``, GPI_HDP.py:3412-3525: hgp_warp_batch_f64 + the a11
prior score) and carries the 4-D tensor through the same helpers.
"""
import numpy as np
import torch

from . import chain_batch, hdp_global, ops

f64 = torch.float64


def _isclose(a, b, rtol=1e-5, atol=1e-8):
    """torch.isclose on two scalars."""
    return abs(a - b) <= atol + rtol * abs(b)


class OfflineLoop:
    """Mixin of GPI_HDP: the batch (offline) variational EM."""

    # ------------------------------------------------------------------ small host-side pieces
    def _log(self, *a):
        if self.verbose:
            print(*a, flush=True)

    def compute_snr_ini(self, y):
        """GPI_HDP.py:715-730: per-lead signal-to-noise ratio of every segment against the lead's mean segment, soft-max over
        the leads -> ``snr_norm`` [N, n_outputs] (identically 1 with one lead)."""
        N, _, D = y.shape
        if not self.use_snr:
            self.snr_norm = torch.ones((N, D), dtype=f64, device=self.device)
            return
        eps = torch.finfo(f64).eps
        target = torch.mean(y, dim=0)                                         # [T, D]
        num = torch.sum(target ** 2, dim=0) + eps                             # torchmetrics SNR(preds=y, target=mean)
        den = torch.sum((target.unsqueeze(0) - y) ** 2, dim=1) + eps          # [N, D]
        self.snr_norm = torch.softmax(10.0 * torch.log10(num.unsqueeze(0) / den), dim=1)

    def redefine_default(self, x, y, resp=None):
        """GPI_HDP.py:1866-1904: default observation / transition noise from the batch itself (median over the first 10
        grid points of lead 0), new bounds, and every model replaced by a fresh default one."""
        n_f = y.shape[0] - 1
        a = y[:n_f, :10, 0].T                                                 # [10, n_f]
        b = y[1:n_f + 1, :10, 0].T
        dev = a - torch.mean(a, dim=1, keepdim=True)
        var_y = torch.median(torch.sum(dev * dev, dim=1) / n_f)
        var_d = torch.median(torch.sum((b - a) ** 2, dim=1) / n_f)
        sig, gam = float(var_y) * 0.02, float(var_d) * 0.025
        self.ini_sigma_def, self.ini_gamma_def = sig, gam
        self.bound_sigma_def = (sig * 1e-5, sig * 2.0)
        self.bound_gamma_def = (gam * 1e-5, gam * 2.0)
        self._log("Redefining default LDS priors.  Sigma:", sig, " Gamma:", gam)
        for ld in range(self.n_outputs):
            for m in range(len(self.gpmodels[ld])):
                self.gpmodels[ld][m] = self.create_gp_default()

    @staticmethod
    def _counts(resp):
        return torch.sum(resp, dim=0)

    def refill(self, resp, respPair, startStateCount, transStateCount, q, q_lat, snr):
        """GPI_HDP.py:1076-1089: an empty cluster in front of a filled last one is swapped with it; an empty last one next to
        another empty one ends the loop."""
        per = self._counts(resp)
        if self.verbose:
            self._log("Group responsability estimated: " + str(per.numpy().astype(np.int64)))
        if bool(torch.any(per[:-1] < 1.0)):
            if per[-1] >= 1.0:
                resp, respPair = self.refill_resp(resp, respPair)
            else:
                self._log("Empty group detected, new iteration.")
                return resp, respPair, q, q_lat, snr, True
        return resp, respPair, q, q_lat, snr, False

    def refill_resp(self, resp, respPair=None):
        """GPI_HDP.py:1141-1168: swap the first empty cluster with the last one, in place.  The pair table is moved by the
        reference's own sequence of row / column copies (not a clean transposition of the two states: rows at or after the empty
        one receive the column entries of their predecessor) - kept as is."""
        per = self._counts(resp)
        if bool(torch.any(per[:-1] < 1.0)):
            e = int(torch.where(per < 1.0)[0][0])
            last = resp[:, -1].clone()
            resp[:, -1] = resp[:, e].clone()
            resp[:, e] = last
            if respPair is not None:
                M = respPair.shape[1]
                row_l = respPair[:, -1, :].clone()
                corner = row_l[:, -1].clone()
                row_l[:, -1] = row_l[:, e].clone()
                row_l[:, e] = corner
                col_l = respPair[:, :-1, -1].clone()
                col_l_e = col_l[:, e].clone()
                respPair[:, -1, :] = respPair[:, e, :].clone()
                respPair[:, :-1, -1] = respPair[:, :-1, e].clone()
                respPair[:, e, :] = row_l
                others = [i for i in range(M) if i != e]
                respPair[:, others, e] = col_l                                 # quirk: M - 1 rows, shifted after e
                respPair[:, -1, e] = col_l_e
        return (resp, respPair) if respPair is not None else resp

    def new_group(self, resp, respPair, q, q_lat, snr):
        """GPI_HDP.py:1112-1131: one more (empty) column everywhere; its snr sits at twice the smallest one, negated."""
        N, M = resp.shape
        D = self.n_outputs
        dev = self.device
        resp_ = torch.zeros((N, M + 1), dtype=f64)
        resp_[:, :-1] = resp
        pair_ = torch.zeros((N, M + 1, M + 1), dtype=f64)
        pair_[:, :-1, :-1] = respPair
        q_ = torch.zeros((N, M + 1, D), dtype=f64, device=dev)
        q_[:, :-1, :] = q
        ql_ = torch.zeros((N, M + 1, D), dtype=f64, device=dev)
        ql_[:, :-1, :] = q_lat
        snr_ = torch.zeros((N, M + 1, D), dtype=f64, device=dev) - torch.abs(torch.min(snr, dim=1)[0])[:, None] * 2.0
        snr_[:, :-1, :] = snr
        return resp_, pair_, q_, ql_, snr_

    @staticmethod
    def remove_last_group(resp, respPair, q, q_lat, snr):
        return resp[:, :-1], respPair[:, :-1, :-1], q[:, :-1, :], q_lat[:, :-1, :], snr[:, :-1, :]

    # ------------------------------------------------------------------ HDP terms of the bound (host, O(M^2))
    def temp_reinit_global_params(self, M, transStateCount, startStateCount, rho=None, omega=None):
        """GPI_HDP.py:365-375."""
        rho = self.rho if rho is None else rho
        omega = self.omega if omega is None else omega
        rho_ = hdp_global.create_initrho(M)
        rho_[:len(rho)] = rho
        omega_ = (1.0 + self.gamma) * np.ones(M)
        omega_[:len(omega)] = omega
        tt, st = self._calcThetaFull(transStateCount, startStateCount, M + 1, rho_)
        return rho_, omega_, tt, st

    def _calcThetaPost(self, transStateCount, startStateCount, M, rho):
        """GPI_HDP.py:383-398: pseudo-counts of a proposal with one more state - 0.8 of the current table, 0.2 of the counts."""
        ebeta = hdp_global.rho2beta(rho, "K+1")
        tt = np.zeros((M, M)) + self.transAlpha * ebeta[None, :]
        tt[:M - 1, :M - 1] += _np(self.transTheta) * 0.8
        tt[:M, :M] += _np(transStateCount)[:M, :M] * 0.2 + self.kappa * np.eye(M)
        st = self.startAlpha * ebeta
        st[:M - 1] += _np(self.startTheta)
        st[:M] += _np(startStateCount)[:M]
        return tt, st

    def elbo_Linears(self, resp, respPair, post=False, one_sample=False):
        """GPI_HDP.py:1025-1074: the HDP part of the bound for a given hard assignment."""
        return self._elbo_linears_counts(resp[0].numpy().copy(), torch.sum(respPair, dim=0).numpy().copy(), resp.shape[1], post, one_sample)

    def _elbo_linears_counts(self, start, trans, M, post=False, one_sample=False):
        """elbo_Linears from the counts themselves: start [M] = one-hot of the first assignment, trans [M, M] = pair counts."""
        if start.shape[0] == M:
            start = np.append(start, 0.0)
        if trans.shape[-1] == M:
            trans = np.pad(trans, ((0, 1), (0, 1)))
        if len(self.rho) == M:
            rho_, omega_ = np.array(self.rho, dtype=np.float64), np.array(self.omega, dtype=np.float64)
        else:
            rho_, omega_, _, _ = self.temp_reinit_global_params(M, trans.copy(), start.copy())
        if post and not one_sample:
            tt, st = self._calcThetaPost(trans.copy(), start.copy(), M + 1, rho_)
        else:
            tt, st = self._calcThetaFull(trans.copy(), start.copy(), M + 1, rho=rho_)
        return hdp_global.elbo_linear_terms(rho_, omega_, self.transAlpha, self.startAlpha, self.kappa, self.gamma,
                                            _np(tt), _np(st), start, trans)

    @staticmethod
    def calcELBO_NonlinearTerms(resp, respPair):
        return hdp_global.elbo_entropy(_np(resp), _np(respPair))

    def full_LDS_elbo(self, gpmodels, sum_resp, one_sample=False):
        """GPI_HDP.py:1838-1864: a9 of every non-empty cluster against its prior, weighted by the cluster's share."""
        sums = [float(v) for v in sum_resp]
        tot = sum(sums)
        M_ = sum(1 for v in sums if v > 0)
        elb = 0.0
        for i, gp in enumerate(gpmodels):
            if sums[i] > 0:
                elb += gp.lds_param_likelihood_value() * (sums[i] / tot)
        return elb if one_sample else elb / M_

    def compute_q_elbo(self, resp, respPair, q, q_lat, gpmodels, M, new_indexes=None, snr=None, post=False, one_sample=False,
                       verb=True):
        """GPI_HDP.py:1796-1836 -> (emission term, everything else).  q, q_lat: [N, M] device (already combined over the
        leads); resp / respPair: host one-hot tables."""
        n_points = 1 if one_sample else self.x_basis_ini.shape[0]
        rows, cols = torch.where(resp == 1.0)            # q[where(resp == 1)]: rows left without a 1 by a slice contribute nothing
        rows, cols = rows.to(q.device), cols.to(q.device)
        q_bas = float(torch.sum(q[rows, cols])) * self.static_factor
        elbo_latent = float(torch.sum(q_lat[rows, cols])) * self.dynamic_factor
        elbo_lin = self.elbo_Linears(resp, respPair, post=post, one_sample=one_sample) * n_points
        if isinstance(snr, str):                                              # 'saved'
            frac = torch.sum(self.snr_norm, dim=0)
        else:
            frac = torch.sum(torch.softmax(torch.max(snr, dim=1)[0], dim=1), dim=0)
        frac = (frac / torch.sum(frac) * n_points).cpu().numpy()
        sum_resp = self._counts(resp)
        elbo_lds = 0.0
        for i in range(self.n_outputs):
            elbo_lds += self.full_LDS_elbo(gpmodels[i], sum_resp, one_sample=one_sample) * float(frac[i])
        if verb and self.verbose:
            self._log("Sum resp_temp: " + str(sum_resp.int().numpy()) + " - Total samples: " + str(int(sum_resp.sum())))
            self._log(f"Q_em: {q_bas:.2f}, Q_lat: {elbo_latent:.2f}, Elbo_linear: {elbo_lin:.2f}, Elbo_LDS: {elbo_lds:.2f}")
        return q_bas, (elbo_lin + elbo_lds + elbo_latent) if self.hmm_switch else elbo_latent

    # ------------------------------------------------------------------ building blocks of the proposals
    def _assign(self, q_w, startPi):
        """LogLik -> forward / backward -> one-hot arg-max of the state and pair posteriors (GPI_HDP.py:1306-1312,
        1587-1595, 2856-2862), kernels only: returns host one-hot tables resp [N,K], respPair [N,K,K]."""
        q_norm, _ = self.LogLik(q_w.contiguous())
        fmsg, _, bmsg, pair = self._messages(startPi, q_norm, True)
        return self._one_hot_tables(fmsg, bmsg, pair)

    @staticmethod
    def _one_hot_tables(fmsg, bmsg, pair):
        """_safe_exp (GPI_HDP.py:338-350) of the state and pair posteriors as host tables of zeros and ones."""
        N, K = fmsg.shape
        labels = ops.assign(fmsg, bmsg)
        flat = pair.reshape(N, K * K)
        # first arg-max per row (torch.argmax of the reference on the host; an all -inf row - the first one - gives 0)
        mx = torch.max(flat, dim=1, keepdim=True)[0]
        ar = torch.arange(K * K, device=flat.device).expand_as(flat)
        first = torch.min(torch.where(flat == mx, ar, torch.full_like(ar, K * K)), dim=1)[0]
        first = torch.where(first == K * K, torch.zeros_like(first), first)   # NaN rows: arg-max of the reference undefined
        lab_h, first_h = labels.cpu(), first.cpu()
        resp = torch.zeros((N, K), dtype=f64)
        resp[torch.arange(N), lab_h] = 1.0
        respPair = torch.zeros((N, K * K), dtype=f64)                         # the reference's table is float32: 0 / 1 either way
        respPair[torch.arange(N), first_h] = 1.0
        return resp, respPair.reshape(N, K, K)

    def _fresh_copy(self, gp):
        """gpmodel_deepcopy + reinit of a fitted model (GPI_HDP.py:1289-1292 and its repeats): same kernel and priors, empty
        history."""
        g = self.gpmodel_deepcopy(gp)
        if g.fitted:
            g.reinit_LDS(save_last=False)
            g.reinit_GP(save_last=False)
        return g

    def _one_member_scores(self, gp_src, x, y, ld, beat, include=True, ycol=None):
        """Scores of all segments (``ycol``: as warped for this column; default the raw lead) under a copy of ``gp_src`` that has
        seen only the RAW segment ``beat`` (as member number 0: GPI_HDP.py:1294, 1581): one Kalman step + one shared-covariance
        launch.  Returns (q [N], the model)."""
        g = self._fresh_copy(gp_src)
        if include:
            g.include_weighted_sample(0, x[beat], x[beat], y[beat, :, [ld]], h=1.0)
        return g.compute_sq_err_all(x, y[:, :, [ld]] if ycol is None else ycol), g

    def _pick_representatives(self, resp_temp, q_w_simple, M, f_ind_old):
        """GPI_HDP.py:1404-1429 / 1760-1785: per cluster the member with the best one-member score that is not already the
        representative of an earlier cluster."""
        out = torch.full_like(f_ind_old, -1)
        used = set()
        q_h = q_w_simple.cpu()
        for k in range(M):
            members = torch.where(resp_temp[:, k] == 1.0)[0]
            order = torch.argsort(q_h[members, k], descending=True)
            cand = members[order]
            pick = next((int(c) for c in cand if int(c) not in used), None)
            if pick is None:
                pick = int(cand[0])
            out[k] = pick
            used.add(pick)
        return out

    # ------------------------------------------------------------------ estimate_q_all
    def estimate_q_all(self, M, x_trains, y_trains, resp, respPair, q_, q_lat_, snr_, startPi, q_def, elbo_def, gpmodels=None,
                       reparam=False, post=True, f_ind_old=None):
        """GPI_HDP.py:2844-2973: re-assign with the current scores, rebuild every cluster whose member set changed, keep the
        new state if the bound improves.  Returns (resp, respPair, q, q_lat, snr, gpmodels)."""
        gpmodels = self.gpmodels if gpmodels is None else gpmodels
        N, D, dev = y_trains.shape[0], self.n_outputs, self.device
        q = torch.zeros((N, M, D), dtype=f64, device=dev) + torch.min(q_) * 2.0
        q_lat = torch.zeros((N, M, D), dtype=f64, device=dev)
        snr_aux = snr_.clone()
        resp_temp, respPair_temp = self._assign(self.weight_mean(q_, snr_aux), startPi)
        reorder = torch.argsort(self._counts(resp_temp), descending=True)
        resp_temp = resp_temp[:, reorder].clone()                              # quirk: the pair table keeps the old order
        f_ind_old = self.f_ind_old if f_ind_old is None else f_ind_old
        Yw, liks = self.warp_batch_by_resp_amtgp_cached(x_trains, y_trains, resp_temp, f_ind_old)   # columns in f_ind_old's order
        gpmodels_temp = [[] for _ in range(D)]
        plan = []                                     # (ld, m, r, model, kind): kind 0 unchanged, 1 changed, 2 new with members, 3 new empty
        for ld in range(D):
            for m in range(M):
                r = int(reorder[m])
                members = torch.where(resp_temp[:, m] == 1.0)[0].tolist()
                if len(gpmodels[ld]) > r:
                    gp = gpmodels[ld][r]
                    kind = 0
                    if members != [int(i) for i in gp.indexes]:
                        kind = 1
                        if gp.fitted:
                            gp = self.gpmodel_deepcopy(gp)
                            gp.reinit_LDS(save_last=not reparam)
                            gp.reinit_GP(save_last=False)
                        else:
                            gp = self.create_gp_default(i=r)
                else:
                    gp = self.create_gp_default(i=r)
                    kind = 2 if len(members) > 0 else 3
                plan.append((ld, m, r, gp, kind))
                gpmodels_temp[ld].append(gp)
        outs = iter(self._passes(x_trains, y_trains, [(gp, self._ycol(Yw, y_trains, ld, r), resp_temp[:, m])
                                                      for ld, m, r, gp, kind in plan if kind in (1, 2)]))
        for ld, m, r, gp, kind in plan:
            if kind in (1, 2):
                out = next(outs)
                self._note_full_pass(resp_temp[:, m], out)
                if out is None:                       # no members: the previous columns (quirk: of the table being filled, for q_lat of a new model)
                    out = (q_[:, r, ld], (q_lat_ if kind == 1 else q_lat)[:, r, ld])
                q[:, m, ld], q_lat[:, m, ld] = out
                if liks is not None:
                    q[:, m, ld] += liks[:, r, ld]
                snr_aux[:, m, ld] = self.compute_snr(self._ylead(Yw, y_trains, ld, r), gp)
            elif kind == 0:
                q[:, m, ld] = q_[:, r, ld]
                q_lat[:, m, ld] = q_lat_[:, r, ld]
                snr_aux[:, m, ld] = snr_[:, m, ld].clone()                     # quirk: column m, not r
            else:
                q[:, m, ld] = q_[:, m, ld]
                q_lat[:, m, ld] = q_lat_[:, m, ld]
                snr_aux[:, m, ld] = 0.0
        self._log(">>> Q_all_loop -------")
        q_bas, elbo_bas = self._elbo_of(resp, respPair, q_, q_lat_, snr_, gpmodels, self.M, post)
        q_bas_post, elbo_post = self._elbo_of(resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, M, post)
        if bool(torch.all(self._counts(resp_temp) >= 1.0)):
            if q_bas + elbo_bas < q_bas_post + elbo_post:
                if reorder.shape[0] == self.f_ind_old.shape[0]:
                    self.f_ind_old = self.f_ind_old[reorder]
                self.snr_norm = self.normalize_snr(snr_aux)
                return resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp
            return resp, respPair, q_, q_lat_, snr_, gpmodels
        self._log(f">>> Possible emergency reallocation. Prev ----: Q_em: {q_def}, Elbo: {elbo_def}")
        if (q_def + elbo_def < q_bas_post + elbo_post) and (q_bas + elbo_bas < q_bas_post + elbo_post):
            self._log("Emergency reallocation and removing last group.")
            for ld in range(D):
                gpmodels_temp[ld] = gpmodels_temp[ld][:-1]
            self.gpmodels = gpmodels_temp
            self.snr_norm = self.normalize_snr(snr_aux)
            resp_temp, respPair_temp, q, q_lat, snr_aux = self.remove_last_group(resp_temp, respPair_temp, q, q_lat, snr_aux)
            reorder = torch.argsort(self._counts(resp_temp), descending=True)
            self.f_ind_old = self.f_ind_old[reorder]
            return resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp
        self._log("Bad estimation")
        return resp, respPair, q_, q_lat_, snr_, gpmodels

    # ------------------------------------------------------------------ warping inside the loop (GPI_HDP.py:3412-3525)
    def warp_batch_by_resp_amtgp_cached(self, x, y, resp_temp, f_ind_old=None, train_iter=50, batch_size=128):
        """Every segment warped onto the representative segment of every cluster column: (Yw [N, T, D, M], liks [N, M, D]); with
        ``warp=False`` both are None (the reference's 4-D tensor is then a broadcast view of y and its liks are zeros).
        One template = the representative f_ind_old[m] of column m on lead ld; its N warps are fitted in batches of 128 by
        hgp_warp_batch_f64 (Warping_system.compute_warp_batch) and scored by the warp prior (a11); templates are cached by
        (lead, representative) for the life of the model, as in the reference."""
        if not self.warp:
            return None, None
        f_ind_old = self.f_ind_old if f_ind_old is None else f_ind_old
        N, T, D = y.shape
        M = resp_temp.shape[1]
        Yw = torch.empty((N, T, D, M), dtype=f64, device=self.device)
        liks = torch.zeros((N, M, D), dtype=f64, device=self.device)
        theta = float(self.ini_lengthscale)                      # kernel_def's length-scale (GPI_HDP.py:3498)
        noise = float(np.sqrt(self.ini_sigma_def))
        for ld in range(D):
            for m in range(M):
                ref = int(f_ind_old[m])
                key = (ld, ref)
                if key not in self._warp_cache_full:
                    warper = self.wp_sys[ld][min(m, len(self.wp_sys[ld]) - 1)]
                    x0 = x[ref].reshape(-1)
                    y_model = y[ref, :, [ld]]
                    yw_all = torch.zeros((N, T), dtype=f64, device=self.device)
                    lik_all = torch.zeros(N, dtype=f64, device=self.device)
                    for s0 in range(0, N, batch_size):
                        idx = slice(s0, min(s0 + batch_size, N))
                        xwB, ywB, likB, _ = warper.compute_warp_batch(x0, y[idx][:, :, [ld]], y_model, theta=theta,
                                                                      noise=torch.full((T,), noise, dtype=f64), train_iter=train_iter)
                        base = self.wp_sys[ld][-1].warp_gp.log_sq_error_batch(x0, xwB[:, :, 0])
                        yw_all[idx] = ywB[:, :, 0]
                        lik_all[idx] = likB + base
                    self._warp_cache_full[key] = (yw_all, lik_all)
                yw_all, lik_all = self._warp_cache_full[key]
                Yw[:, :, ld, m] = yw_all
                liks[:, m, ld] = lik_all
        return Yw, liks

    @staticmethod
    def _ycol(Yw, y, ld, m):
        """Observations cluster column m scores / absorbs on lead ld, [N, T, 1]."""
        return y[:, :, [ld]] if Yw is None else Yw[:, :, [ld], m]

    @staticmethod
    def _ylead(Yw, y, ld, m):
        return y[:, :, ld] if Yw is None else Yw[:, :, ld, m]

    @staticmethod
    def select_assigned_warp(Yw, y, resp, order=None):
        """GPI_HDP.py:3519-3525: every segment as warped onto ITS cluster's representative, [N, T, D]."""
        if Yw is None:
            return y
        if order is not None:
            Yw = Yw[:, :, :, order.to(Yw.device)]
        z = torch.argmax(resp, dim=1).to(Yw.device)
        return torch.gather(Yw, 3, z[:, None, None, None].expand(-1, Yw.shape[1], Yw.shape[2], 1)).squeeze(3)

    def _passes(self, x, y, specs):
        """full_pass_weighted of many (model, observations [N, T, 1], membership column) at once: the chains do not depend on each other, so they
        advance side by side (chain_batch.py).  Returns per spec (q, q_lat), or None for a cluster without members (the caller
        keeps its previous columns, GPI_model.py:385-386).  ``_note_full_pass`` sees every result in spec order."""
        jobs = [chain_batch.Job(gp, x, ycol, col) for gp, ycol, col in specs]
        chain_batch.run(jobs)
        return [j.out if len(j.active) else None for j in jobs]

    def _note_full_pass(self, resp_col, out):
        """Hook (tests): called once per full pass, in the order the reference's loop makes them."""

    def _elbo_of(self, resp, respPair, q, q_lat, snr, gpmodels, M, post):
        return self.compute_q_elbo(resp, respPair, self.weight_mean(q, snr), self.weight_mean(q_lat, snr), gpmodels, M,
                                   snr=snr, post=post)

    def _converge(self, M, x, y, resp_temp, respPair_temp, q, q_lat, snr_aux, startPi, q_def, elbo_def, gpmodels_temp, reparam,
                  base, limit, post_all, post_elbo, f_ind_old=None):
        """The inner 'estimate_q_all until the bound stops moving' loops of GPI_HDP.py:1352-1379 and 1701-1729."""
        i = 0
        while True:
            resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp = self.estimate_q_all(
                M, x, y, resp_temp, respPair_temp, q, q_lat, snr_aux, startPi, q_def, elbo_def, gpmodels=gpmodels_temp,
                reparam=reparam, post=post_all, f_ind_old=f_ind_old)
            q_post, elbo_post = self._elbo_of(resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, M, post_elbo)
            self._log("ELBO_reduction: " + str((q_post + elbo_post) - base))
            if (_isclose(base, q_post + elbo_post) and i > 0) or i == limit:
                break
            base = q_post + elbo_post
            i += 1
        return resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, base

    # ------------------------------------------------------------------ estimate_q_first
    def estimate_q_first(self, M, x_trains, y_trains, resp, respPair, q_, q_lat_, snr_, startPi, reallocate_=False, reparam=False):
        """GPI_HDP.py:1243-1794: first try to move segments between the existing clusters; failing that, propose births
        seeded by badly explained segments and keep the first one that improves the bound.
        Returns (resp, respPair, q, q_lat, snr, reallocate)."""
        x, y = x_trains, y_trains
        N, D, dev = y.shape[0], self.n_outputs, self.device
        empty_estimation = False
        Yw, liks = self.warp_batch_by_resp_amtgp_cached(x, y, resp, self.f_ind_old)
        if float(torch.mean(q_)) == 0.0:                     # nothing scored yet: cluster 0 takes the whole batch
            snr_ = torch.zeros((N, M, D), dtype=f64, device=dev)
            if self.share_gp and D > 1:
                raise NotImplementedError("share_gp with several leads is not part of this build")
            firsts = [self.create_gp_default() for _ in range(D)]
            for ld, out in enumerate(self._passes(x, y, [(firsts[ld], self._ycol(Yw, y, ld, 0), resp[:, 0]) for ld in range(D)])):
                self._note_full_pass(resp[:, 0], out)
                q_[:, 0, ld], q_lat_[:, 0, ld] = out
                if liks is not None:
                    q_[:, 0, ld] += liks[:, 0, ld]
                snr_[:, 0, ld] = self.compute_snr(self._ylead(Yw, y, ld, 0), firsts[ld])
                self.gpmodels[ld][0] = firsts[ld]
        reallocate = False
        indexes_ = []
        for m in range(M):
            idx = torch.as_tensor(self.gpmodels[0][m].indexes, dtype=torch.int64)
            indexes_.append(idx if idx.numel() else torch.where(resp[:, m] == 1.0)[0])
        f_ind_old = self.f_ind_old.clone()

        # every cluster re-seeded with its representative segment only: how well does that one segment explain the batch?
        q_simple = q_.clone()
        for ld in range(D):
            for m in range(M):
                q_simple[:, m, ld], _ = self._one_member_scores(self.gpmodels[ld][m], x, y, ld, int(f_ind_old[m]),
                                                                include=len(indexes_[m]) > 0, ycol=self._ycol(Yw, y, ld, m))
                if liks is not None:
                    q_simple[:, m, ld] += liks[:, m, ld]

        if M > 1:
            q_aux, snr_aux = q_simple.clone(), snr_.clone()
            if self._counts(resp)[-1] == 0:
                q_aux[:, -1, :] = torch.min(q_aux) * 2.0
                snr_aux[:, -1, :] = torch.min(snr_aux) * 2.0
            resp_temp, respPair_temp = self._assign(self.weight_mean(q_aux, snr_aux), startPi)
            reorder = torch.argsort(self._counts(resp_temp), descending=True)
            resp_temp = resp_temp[:, reorder]
            q, q_lat = q_.clone(), q_lat_.clone()
            gpmodels_temp = [[] for _ in range(D)]
            plan = []
            for ld in range(D):
                for m in range(M):
                    r = int(reorder[m])
                    changed = not torch.equal(resp[:, r].long(), resp_temp[:, m].long())
                    gp = self._fresh_copy(self.gpmodels[ld][r]) if changed else self.gpmodels[ld][r]
                    plan.append((ld, m, r, gp, changed))
                    gpmodels_temp[ld].append(gp)
            outs = iter(self._passes(x, y, [(gp, self._ycol(Yw, y, ld, r), resp_temp[:, m]) for ld, m, r, gp, changed in plan if changed]))
            for ld, m, r, gp, changed in plan:
                if changed:
                    out = next(outs)
                    self._note_full_pass(resp_temp[:, m], out)
                    q[:, m, ld], q_lat[:, m, ld] = out if out is not None else (q[:, r, ld], q_lat[:, r, ld])
                    if liks is not None:
                        q[:, m, ld] += liks[:, r, ld]
                    snr_aux[:, m, ld] = self.compute_snr(self._ylead(Yw, y, ld, r), gp)
                else:
                    q[:, m, ld] = q_[:, r, ld].clone()                    # quirk: q_lat keeps column m of the old table
                    snr_aux[:, m, ld] = snr_[:, r, ld].clone()
            q_b, e_b = self._elbo_of(resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, M, False)
            q_def__, elbo_def__ = self._elbo_of(resp, respPair, q_, q_lat_, snr_, self.gpmodels, M, False)
            resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, base_ = self._converge(
                M, x, y, resp_temp, respPair_temp, q, q_lat, snr_aux, startPi, q_def__, elbo_def__, gpmodels_temp, reparam,
                q_b + e_b, 20, False, False)
            self._log(">>> Prev -------")
            q_bas, elbo_bas = self._elbo_of(resp, respPair, q_, q_lat_, snr_, self.gpmodels, M, False)
            self._log(">>> Post -------")
            q_bas_post, elbo_post = self._elbo_of(resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, M, False)
            if int(torch.sum(self._counts(resp_temp) < 1.0)) == 0:
                if q_bas + elbo_bas < q_bas_post + elbo_post and q_bas != q_bas_post:
                    self._log("Reallocating beats into existing groups.")
                    self.gpmodels = gpmodels_temp
                    self.y_train = self.select_assigned_warp(Yw, y, resp_temp, reorder)
                    self.f_ind_old = self._pick_representatives(resp_temp, self.weight_mean(q_simple, snr_aux), M, f_ind_old)
                    self.snr_norm = self.normalize_snr(snr_aux)
                    return resp_temp, respPair_temp, q, q_lat, snr_aux, True
                self._log("Not reallocating, trying to generate new group.")
            else:
                self._log(">>> Possible emergency reallocation. Prev ----")
                q_bas, elbo_bas = self._elbo_of(resp, respPair, q_, q_lat_, snr_, self.gpmodels, self.M, False)
                if q_bas + elbo_bas < base_:
                    self._log("Emergency reallocation and removing last group.")
                    for ld in range(D):
                        gpmodels_temp[ld] = gpmodels_temp[ld][:-1]
                    self.gpmodels = gpmodels_temp
                    self.snr_norm = self.normalize_snr(snr_aux)
                    resp_temp, respPair_temp, q, q_lat, snr_aux = self.remove_last_group(resp_temp, respPair_temp, q, q_lat, snr_aux)
                    reorder = torch.argsort(self._counts(resp_temp), descending=True)
                    self.f_ind_old = self.f_ind_old[reorder]
                    return resp_temp, respPair_temp, q, q_lat, snr_aux, True
                self._log("Bad estimation")
                empty_estimation = True

        # ---- birth proposals: candidate seeds = segments their own cluster explains worst ---------------------------
        lab = torch.argmax(resp, dim=1)

        def own(t3):          # weight_mean(t)[where(resp == 1)], scaled to [-1, 0] (GPI_HDP.py:1462-1467)
            v = self.weight_mean(t3).cpu().gather(1, lab[:, None]).reshape(-1)
            return (v - torch.max(v)) / (torch.max(v) - torch.min(v))

        q_rank = own(q_simple)
        by_rank = torch.argsort(q_rank)
        by_total = torch.argsort(own(q_) + own(q_lat_))
        near_cache = {}

        def near(i):          # segments whose rank is within 1 % of segment i's (GPI_HDP.py:1474)
            if i not in near_cache:
                near_cache[i] = set(torch.where(torch.isclose(q_rank, q_rank[i], rtol=0.01))[0].tolist())
            return near_cache[i]

        def cluster_of(i, n_search):
            for m in range(n_search):
                if bool((indexes_[m] == i).any()):
                    return m
            return int(torch.argmax(resp[i]))

        n_steps = self.n_explore_steps
        seeds = torch.zeros(n_steps, dtype=torch.int64)
        half = int(max(n_steps // 2, 1))

        def fill(order, j_, stop):
            last = {-1}
            for f in order.tolist():
                if j_ == stop:
                    break
                if f != int(f_ind_old[cluster_of(f, M - 1)]) and not last <= near(f):
                    last = near(f)
                    seeds[j_] = f
                    j_ += 1

        fill(by_rank, 0, half)
        fill(by_total, half, n_steps)

        resp_, respPair_, q_def, q_lat_def, snr_aux_def = self.new_group(resp, respPair, q_simple.clone(), q_lat_.clone(), snr_.clone())
        _, _, q__def, q_lat__def, snr__def = self.new_group(resp, respPair, q_.clone(), q_lat_.clone(), snr_.clone())
        M = M + 1
        f_ind_old = torch.zeros(M, dtype=torch.int64)
        f_ind_old[:self.f_ind_old.shape[0]] = self.f_ind_old
        # The candidates of one round do not depend on each other's outcome (the reference tries them one by one and stops at the
        # first it accepts; nothing it computes for candidate j feeds candidate j + 1).  They are therefore PREPARED together
        # (host logic + one one-member score and one message pass each), all the cluster rebuilds they need run side by side as
        # one batch of independent chains, and the accept / reject decisions are then taken in the reference's order.
        props = []
        step, last = 0, {-1}
        for f_new in seeds.tolist():
            if step == n_steps:
                break
            m_chosen = cluster_of(f_new, M - 1)
            if f_new == int(f_ind_old[m_chosen]) or last <= near(f_new):
                continue
            P = {"f_new": f_new, "m_chosen": m_chosen, "f_ind_old_temp": None, "q_simple_": None}
            if not empty_estimation:
                P["f_ind_old_temp"] = f_ind_old.clone()
                P["f_ind_old_temp"][-1] = f_new
                q_simple_ = q_def.clone()
                q, q_lat, snr_aux = q_def.clone(), q_lat_def.clone(), snr_aux_def.clone()
                q__, q_lat__, snr__ = q__def.clone(), q_lat__def.clone(), snr__def.clone()
                last = near(f_new)
                step += 1
                self._log(f"Step {step}/{n_steps}- Trying to divide: {m_chosen} with beat {f_new}")
                Yp, lp = self.warp_batch_by_resp_amtgp_cached(x, y, resp_, P["f_ind_old_temp"])     # + the column warped onto the seed
                for ld in range(D):
                    q_simple_[:, -1, ld], g1 = self._one_member_scores(self.gpmodels[ld][m_chosen], x, y, ld, f_new,
                                                                       ycol=self._ycol(Yp, y, ld, -1))
                    if lp is not None:
                        q_simple_[:, -1, ld] += lp[:, -1, ld]
                    snr_aux[:, -1, ld] = self.compute_snr(self._ylead(Yp, y, ld, -1), g1)
                P["q_simple_"] = q_simple_
                resp_temp, respPair_temp = self._assign(self.weight_mean(q_simple_, snr_aux), startPi)
            else:
                q, q_lat, snr_aux = q__def.clone(), q_lat__def.clone(), snr__def.clone()
                q__, q_lat__, snr__ = q__def.clone(), q_lat__def.clone(), snr__def.clone()
                q[:, -1, :] = torch.min(q) * 2.0
                q__[:, -1, :] = torch.min(q__) * 2.0
                snr_aux[:, -1, :] = torch.min(snr_aux) * 2.0
                q__[f_new, -1, :] = 0.0
                resp_temp, respPair_temp = self._assign(self.weight_mean(q__, snr_aux), startPi)
                Yp, lp = Yw, liks                    # (the reference keeps the round's warps on this path)
            reorder = torch.argsort(self._counts(resp_temp), descending=True)
            resp_temp = resp_temp[:, reorder]
            gpmodels_temp = [[] for _ in range(D)]
            plan = []
            for ld in range(D):
                for m in range(M):
                    r = int(reorder[m])
                    if r == M - 1:
                        gp = self._fresh_copy(self.gpmodels[ld][m_chosen]) if self.share_gp else self.create_gp_default()
                        rebuild = True
                    else:
                        rebuild = not torch.equal(resp[:, r].long(), resp_temp[:, m].long())
                        gp = self._fresh_copy(self.gpmodels[ld][r]) if rebuild else self.gpmodels[ld][r]
                    plan.append((ld, m, r, gp, rebuild))
                    gpmodels_temp[ld].append(gp)
            P.update(q=q, q_lat=q_lat, snr_aux=snr_aux, q__=q__, q_lat__=q_lat__, snr__=snr__, resp_temp=resp_temp,
                     respPair_temp=respPair_temp, reorder=reorder, gpmodels_temp=gpmodels_temp, plan=plan, Yp=Yp, lp=lp)
            props.append(P)
        # ... in two batches: the first candidate alone (it is the one most often accepted), then all the others.
        def rebuilds(ps):
            return iter(self._passes(x, y, [(gp, self._ycol(P["Yp"], y, ld, r), P["resp_temp"][:, m])
                                            for P in ps for ld, m, r, gp, rebuild in P["plan"] if rebuild]))

        outs = None
        for ip, P in enumerate(props):
            if ip < 2:
                outs = rebuilds(props[:1] if ip == 0 else props[1:])
            f_new, m_chosen, q_simple_, f_ind_old_temp = P["f_new"], P["m_chosen"], P["q_simple_"], P["f_ind_old_temp"]
            q, q_lat, snr_aux, q__, q_lat__, snr__ = P["q"], P["q_lat"], P["snr_aux"], P["q__"], P["q_lat__"], P["snr__"]
            resp_temp, respPair_temp, reorder, gpmodels_temp = P["resp_temp"], P["respPair_temp"], P["reorder"], P["gpmodels_temp"]
            Yp, lp = P["Yp"], P["lp"]
            for ld, m, r, gp, rebuild in P["plan"]:
                if rebuild:
                    out = next(outs)
                    self._note_full_pass(resp_temp[:, m], out)
                    q[:, m, ld], q_lat[:, m, ld] = out if out is not None else (q__[:, r, ld], q_lat__[:, r, ld])
                    if lp is not None:
                        q[:, m, ld] += lp[:, r, ld]
                    snr_aux[:, m, ld] = self.compute_snr(self._ylead(Yp, y, ld, r), gp)
                else:
                    q[:, m, ld] = q__[:, r, ld].clone()
                    q_lat[:, m, ld] = q_lat__[:, r, ld].clone()
                    snr_aux[:, m, ld] = snr__[:, r, ld].clone()
            q_b, e_b = self._elbo_of(resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, M, True)
            counts_t = self._counts(resp_temp)
            if int(torch.argmax(counts_t)) == M - 1:
                self._log("Bad estimation")
                continue
            if int(torch.sum(counts_t < 1.0)) > 0:
                self._log(">>> Possible emergency reallocation. Prev ----")
                q_bas, elbo_bas = self._elbo_of(resp, respPair, q_, q_lat_, snr_, self.gpmodels, self.M, False)
                if q_bas + elbo_bas < q_b + e_b:
                    self._log("Emergency reallocation and removing last group.")
                    for ld in range(D):
                        gpmodels_temp[ld] = gpmodels_temp[ld][:-1]
                    resp_temp, respPair_temp, q, q_lat, snr_aux = self.remove_last_group(resp_temp, respPair_temp, q, q_lat, snr_aux)
                    self.gpmodels = gpmodels_temp
                    for ld in range(D):
                        self.wp_sys[ld] = self.wp_sys[ld][:-1]                    # quirk: only this exit drops a warper
                    self.f_ind_old = f_ind_old[reorder]
                    self.y_train = self.select_assigned_warp(Yp, y, resp_temp, reorder)
                    self.snr_norm = self.normalize_snr(snr_aux)
                    return resp_temp, respPair_temp, q, q_lat, snr_aux, True
                self._log("Bad estimation")
                continue
            q_def__, elbo_def__ = self._elbo_of(resp, respPair, q_, q_lat_, snr_, self.gpmodels, self.M, False)
            resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, _ = self._converge(
                M, x, y, resp_temp, respPair_temp, q, q_lat, snr_aux, startPi, q_def__, elbo_def__, gpmodels_temp, reparam,
                q_b + e_b, 10, True, True, f_ind_old=f_ind_old_temp)
            self._log(f"- Trying to divide: {m_chosen} with beat {f_new}")
            self._log(">>> Prev -------")
            q_bas, elbo_bas = self._elbo_of(resp, respPair, q_, q_lat_, snr_, self.gpmodels, self.M, False)
            self._log(">>> Post -------")
            q_bas_post, elbo_post = self._elbo_of(resp_temp, respPair_temp, q, q_lat, snr_aux, gpmodels_temp, M, True)
            counts_t = self._counts(resp_temp)
            if bool(torch.all(counts_t >= 1.0)) and int(torch.argmax(counts_t)) != resp_temp.shape[1] - 1:
                if q_bas + elbo_bas < q_bas_post + elbo_post:
                    self._log(f"Chosen to divide: {m_chosen} with beat {f_new}")
                    self.gpmodels = gpmodels_temp
                    for ld in range(D):
                        self.wp_sys[ld].append(self.create_wp_sys_default())
                    self.y_train = self.select_assigned_warp(Yp, y, resp_temp, reorder)
                    self.f_ind_old = self._pick_representatives(resp_temp, self.weight_mean(q_simple_, snr_aux), M, f_ind_old)
                    self.snr_norm = self.normalize_snr(snr_aux)
                    return resp_temp, respPair_temp, q, q_lat, snr_aux, reallocate
            else:
                self._log("Bad estimation")
        return resp, respPair, q_, q_lat_, snr_, True

    # ------------------------------------------------------------------ one EM iteration
    def _log_pis(self, M, transTheta, startTheta):
        """log transition matrix / start vector handed to the proposals (GPI_HDP.py:1188-1194); the messages themselves
        always rebuild the transition matrix from ``self.transTheta`` (GPI_HDP.py:3580)."""
        tt, st = _np(transTheta), _np(startTheta)
        dg = hdp_global.digamma
        transPi = dg(tt[:M, :M]) - np.log(np.sum(np.exp(dg(tt[:M, :M + 1])), axis=1) + 1e-5)[:, None]
        startPi = dg(st[:M]) - np.log(np.sum(np.exp(dg(st[:M + 1]))) + 1e-5)
        return torch.as_tensor(transPi, dtype=f64), torch.as_tensor(startPi, dtype=f64)

    def variational_local_terms_batch(self, M, x_trains, y_trains, transTheta, startTheta, resp, respPair, q, q_lat, snr, reallocate):
        """GPI_HDP.py:1170-1241: proposals (estimate_q_first) when the last cluster is in use, then estimate_q_all until the
        bound converges."""
        transTheta = self.transTheta if transTheta is None else transTheta
        startTheta = self.startTheta if startTheta is None else startTheta
        self.trans_A, startPi = self._log_pis(M, transTheta, startTheta)
        i = 0
        reparam = True
        per = self._counts(resp)
        if per.shape[0] == 1 or per[-2] >= 1.0 or not self.gpmodels[0][0].fitted:
            resp, respPair, q, q_lat, snr, reallocate = self.estimate_q_first(M, x_trains, y_trains, resp, respPair, q, q_lat, snr,
                                                                              startPi, reallocate_=reallocate, reparam=reparam)
            q_bas, elbo_bas = self.compute_q_elbo(resp, respPair, self.weight_mean(q), self.weight_mean(q_lat), self.gpmodels,
                                                  self.M, snr='saved', post=resp.shape[1] > self.M)
            i += 1
            if self.verbose:
                self._log("First resp: " + str(self._counts(resp).int().numpy()))
        else:
            q_bas, elbo_bas = self.compute_q_elbo(resp, respPair, self.weight_mean(q), self.weight_mean(q_lat), self.gpmodels,
                                                  self.M, snr='saved', post=False)
            self._log("Not first estimated q.")
        q_def__, elbo_def__ = q_bas, elbo_bas
        if not reallocate:
            while True:
                M = resp.shape[1]
                resp, respPair, q, q_lat, snr, gpmodels = self.estimate_q_all(M, x_trains, y_trains, resp, respPair, q, q_lat, snr,
                                                                              startPi, q_def__, elbo_def__, reparam=reparam)
                self.gpmodels = gpmodels
                q_post, elbo_post = self.compute_q_elbo(resp, respPair, self.weight_mean(q), self.weight_mean(q_lat), self.gpmodels,
                                                        self.M, snr='saved', post=resp.shape[1] > self.M)
                self._log("ELBO_reduction: " + str((q_post + elbo_post) - (q_bas + elbo_bas)))
                if (_isclose(q_bas + elbo_bas, q_post + elbo_post) and i > 0) or i == 10:
                    break
                q_bas, elbo_bas = q_post, elbo_post
                i += 1
        return resp, respPair, q, q_lat, snr, reallocate

    # ------------------------------------------------------------------ the loop
    def include_batch(self, x_trains, y_trains, it_limit=None, warp=False, with_warp=None):
        """GPI_HDP.py:805-943.  ``with_warp`` is what the reference's own drivers pass (hdpgpc/tests/test_offline.py:79)."""
        if with_warp is not None:
            warp = with_warp
        if self.reduce_outputs:
            raise NotImplementedError("reduce_outputs is not part of this build")
        self.warp = bool(warp)
        self._log(f"------ HDP Hyperparameters ------\ngamma: {self.gamma}\ntransAlpha: {self.transAlpha}\n"
                  f"startAlpha: {self.startAlpha}\nkappa: {self.kappa}\n---------------------------------")
        y = self.cond_to_torch(y_trains)
        x = self.cond_to_torch(x_trains)
        N, _, D = y.shape
        assert D == self.n_outputs
        self.T = self.T + N
        self.compute_snr_ini(y)
        M = self.M
        self.x_train = x
        iteration = 0
        resp = torch.zeros((N, M), dtype=f64)
        respPair = torch.zeros((N, M, M), dtype=f64)
        respPair[:, 0, 0] = 1.0
        resp[:, 0] = 1.0
        q = torch.zeros((N, M, D), dtype=f64, device=self.device)
        q_lat = torch.zeros((N, M, D), dtype=f64, device=self.device)
        snr = self.snr_norm
        if self.reestimate_initial_params:
            self.redefine_default(x, y, resp)
        self._em_loop(x, y, resp, respPair, q, q_lat, snr, None, None, it_limit, first_batch=True)

    def _em_loop(self, x, y, resp, respPair, q, q_lat, snr, startStateCount, transStateCount, it_limit, first_batch):
        """The outer EM iterations shared by include_batch (GPI_HDP.py:861-947) and cluster_new_batch(learning=True)
        (GPI_HDP.py:3073-3151): proposals + re-assignment, HDP pseudo-counts, the bound, stop when the assignments repeat."""
        iteration = 0
        reallocate = False
        while True:
            resp, respPair, q, q_lat, snr, end = self.refill(resp, respPair, startStateCount, transStateCount, q, q_lat, snr)
            M = self.M
            if first_batch and resp.shape[1] == 1:
                startStateCount, transStateCount = resp[0].numpy().copy(), torch.sum(respPair, dim=0).numpy()
                self._update_global(M, transStateCount, startStateCount)
            if end:
                break
            resp, respPair, q, q_lat, snr, reallocate = self.variational_local_terms_batch(
                M, x, y, self.transTheta, self.startTheta, resp, respPair, q, q_lat, snr, reallocate)
            if resp.shape[1] > M:
                self.M = M = M + 1
            if self.hmm_switch:
                startStateCount, transStateCount = resp[0].numpy().copy(), torch.sum(respPair, dim=0).numpy()
            else:
                transStateCount, startStateCount = np.ones((M + 1, M + 1)), np.ones(M + 1)
            self._update_global(M, transStateCount, startStateCount)
            tt = _np(self.transTheta)
            dg = hdp_global.digamma
            self.trans_A = torch.as_tensor(dg(tt[:M, :M]) - np.log(np.sum(np.exp(dg(tt[:M, :M + 1])), axis=1))[:, None])
            if self.T <= 1:
                break
            elbo_ = self.calcELBO_NonlinearTerms(resp, respPair)
            self._log(f"\n-------End Lower Bound Iteration {iteration}-------")
            q_obs, elbo_lin = self.compute_q_elbo(resp, respPair, self.weight_mean(q), self.weight_mean(q_lat), self.gpmodels, self.M,
                                                  snr='saved', post=False)
            elbo_ = elbo_ + elbo_lin + q_obs
            self._log("ELBO + Nonlinear: " + str(elbo_))
            iteration += 1
            self._log(f"\n-------Start lower Bound Iteration {iteration}-------")
            labels = torch.argmax(resp, dim=1)               # = torch.where(resp == 1.0)[1] for one-hot rows
            if (it_limit is not None and iteration >= it_limit) or (first_batch and self.M == self.max_models):
                self.train_elbo.append(elbo_)
                self.resp_assigned.append(labels)
                break
            self.train_elbo.append(elbo_)
            self.resp_assigned.append(labels)
            if first_batch:
                self.q.append(q)
                self.elbo_last = elbo_
            self.q_last, self.q_lat_last, self.snr_last = q, q_lat, snr
            self.startStateCount_last, self.transStateCount_last = startStateCount, transStateCount
            self.resp_last, self.respPair_last = resp, respPair
            ra = self.resp_assigned
            if (int(torch.sum(self._counts(resp) == 0.0)) > 1 or
                    (len(ra) > 1 and ra[-2].shape[0] == ra[-1].shape[0] and bool(torch.all(ra[-2] == ra[-1])))):
                if not first_batch:
                    self.y_train = y
                break
            if not first_batch:
                self.y_train = y

    def cluster_new_batch_learning(self, x_new, y_new, it_limit=None):
        """GPI_HDP.py:3002-3151, ``cluster_new_batch(learning=True)``: classify the new segments with the frozen models, append
        them to the stored batch, rebuild every cluster from the joint assignment and run the EM iterations on the whole.
        (Where the reference's loop meets its stop condition it evaluates an undefined name, ``warp_computed``,
        GPI_HDP.py:3139 - a NameError; with ``warp=False`` the branch it guards is the plain ``break`` taken here.)"""
        D, dev = self.n_outputs, self.device
        q_new = self.frozen_scores(x_new, y_new)
        snr_new = torch.stack([torch.stack([self.compute_snr(y_new[:, :, ld], self.gpmodels[ld][m]) for ld in range(D)], dim=-1)
                               for m in range(self.M)], dim=1)
        _, startPi = self._log_pis(self.M, self.transTheta, self.startTheta)
        resp_new, respPair_new = self._assign(self.weight_mean(q_new, snr_new), startPi)
        x = torch.cat([self.x_train, x_new])
        y = torch.cat([self.y_train, y_new])
        N = self.T = y.shape[0]
        resp = torch.cat([self.resp_last, resp_new])
        respPair = torch.cat([self.respPair_last, respPair_new])
        self.snr_norm = torch.cat([self.snr_norm, self.normalize_snr(snr_new)])
        reorder = torch.argsort(self._counts(resp), descending=True)
        resp = resp[:, reorder]                                                # quirk: the pair table keeps the old order
        q = torch.zeros((N, self.M, D), dtype=f64, device=dev)
        q_lat, snr = torch.zeros_like(q), torch.zeros_like(q)
        models = [[self._fresh_copy(self.gpmodels[ld][int(reorder[m])]) for m in range(self.M)] for ld in range(D)]
        outs = iter(self._passes(x, y, [(models[ld][m], y[:, :, [ld]], resp[:, m]) for ld in range(D) for m in range(self.M)]))
        for ld in range(D):
            for m in range(self.M):
                out = next(outs)
                self._note_full_pass(resp[:, m], out)
                if out is not None:
                    q[:, m, ld], q_lat[:, m, ld] = out
                snr[:, m, ld] = self.compute_snr(y[:, :, ld], models[ld][m])
        self.gpmodels = models
        resp, respPair = self._assign(self.weight_mean(q, snr), startPi)
        self.x_train = x
        self._em_loop(x, y, resp, respPair, q, q_lat, snr, resp[0].numpy().copy(), torch.sum(respPair, dim=0).numpy(), it_limit,
                      first_batch=False)

    def _update_global(self, M, transStateCount, startStateCount):
        """GPI_HDP.py:868-873 / 897-902: pseudo-counts from the hard assignment, two rounds of the (rho, omega) optimiser."""
        self.reinit_global_params(M, transStateCount, startStateCount)
        for _ in range(2):
            self.transTheta, self.startTheta = self._calcThetaFull(transStateCount, startStateCount, M + 1)
            self.rho, self.omega = self.find_optimum_rhoOmega()


def _np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
