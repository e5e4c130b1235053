"""TEST INFRASTRUCTURE ONLY - a CPU stand-in for the handful of ``hdpgpc_amd.ops`` entry points the host-side control loop
(``GPI_HDP.include_batch`` and friends) reaches, written with NumPy / SciPy and the oracle, so that the HOST LOGIC can be
checked against the reference's traces in the CPU test tier (``-m "not gpu"``), where no kernel can run.

Nothing under ``hdpgpc_amd/`` imports this file; the product path has no CPU fallback (``hdpgpc_amd.ops`` calls the C-ABI
library and fails without a GPU).  ``install(monkeypatch)`` swaps the functions for the duration of one test.
"""
import numpy as np
import scipy.linalg
import torch

from oracle import hdpgpc_oracle as orc

f64 = torch.float64


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=f64)


def _n(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def gram_rbf(x, y, c, ell, noise=0.0):
    return _t(orc.gram_rbf(_n(x).reshape(-1, 1), None if y is None else _n(y).reshape(-1, 1), c, ell, noise))


def _shifted(A, jitter_rel, add_diag):
    A = 0.5 * (A + A.T)
    shift = add_diag + jitter_rel * max(np.mean(np.abs(np.diag(A))), np.finfo(np.float64).eps)
    return A + shift * np.eye(A.shape[0])


def _chol(A):
    try:
        return np.linalg.cholesky(A), 0
    except np.linalg.LinAlgError:
        return np.full_like(A, np.nan), 1


def potrf_batched(A, jitter_rel=1e-8, add_diag=0.0, want_inv=False, want_logdet=False):
    A3 = _n(A).reshape(-1, A.shape[-1], A.shape[-1])
    Ls, infos = zip(*[_chol(_shifted(a, jitter_rel, add_diag)) for a in A3])
    out = [_t(np.stack(Ls)), torch.tensor(infos, dtype=torch.int32)]
    if want_inv:
        out.append(_t(np.stack([scipy.linalg.solve_triangular(L, np.eye(L.shape[0]), lower=True) for L in Ls])))
    if want_logdet:
        out.append(_t([2.0 * np.sum(np.log(np.diag(L))) for L in Ls]))
    return tuple(out)


def chol_inverse(A, jitter_rel=0.0, add_diag=0.0, out=None, info=None, work=None):
    L, inf, Z = potrf_batched(A, jitter_rel, add_diag, want_inv=True)
    if out is not None:
        out.copy_(Z)
        Z = out
    if info is not None:
        info.copy_(inf)
        inf = info
    return Z, inf


def gemm_batched(A, B, transA=False, transB=False, alpha=1.0, add=None, beta=1.0, out=None):
    a = A.transpose(-1, -2) if transA else A
    b = B.transpose(-1, -2) if transB else B
    C = alpha * torch.matmul(a, b)
    if add is not None:
        C = C + beta * add
    if out is not None:
        out.copy_(C.reshape(out.shape))
        return out
    return C.contiguous()


def _quad(y, mean, cov, add, jitter_rel):
    L, info = _chol(_shifted(cov + add * np.eye(cov.shape[0]), jitter_rel, 0.0))
    if info:
        return np.nan, info
    z = scipy.linalg.solve_triangular(L, y - mean, lower=True)
    return float(z @ z), 0


def score_groups(Y, mean, Sigma, item_mat, item_add, item_off, item_cnt, seg_ids=None, jitter_rel=1e-8, want_logdet=False,
                 want_info=True, item_mean=None, strides=None):
    Yn, Sn = _n(Y), _n(Sigma).reshape(-1, Y.shape[1], Y.shape[1])
    mn = np.zeros((Sn.shape[0], Y.shape[1])) if mean is None else _n(mean).reshape(-1, Y.shape[1])
    quad, info = np.zeros(Yn.shape[0]), np.zeros(Yn.shape[0], dtype=np.int32)
    im, io, ic = _n(item_mat), _n(item_off), _n(item_cnt)
    ia = np.zeros(len(im)) if item_add is None else _n(item_add)
    ime = im if item_mean is None else _n(item_mean)
    sid = None if seg_ids is None else _n(seg_ids)
    for it in range(len(im)):
        for j in range(int(io[it]), int(io[it]) + int(ic[it])):
            n = j if sid is None else int(sid[j])
            quad[n], info[n] = _quad(Yn[n], mn[int(ime[it])], Sn[int(im[it])], float(ia[it]), jitter_rel)
    return _t(quad), None, torch.as_tensor(info)


def score_each(Y, mean, Sigma, seg_mat, seg_mean=None, seg_add=None, jitter_rel=1e-8, want_logdet=False, want_info=True,
               symmetric=False, strides=None):
    Yn, Sn, mn = _n(Y), _n(Sigma), _n(mean).reshape(-1, Y.shape[1])
    sm = _n(seg_mat)
    sme = sm if seg_mean is None else _n(seg_mean)
    sa = np.zeros(len(sm)) if seg_add is None else _n(seg_add)
    res = [_quad(Yn[i], mn[int(sme[i])], Sn[int(sm[i])], float(sa[i]), jitter_rel) for i in range(Yn.shape[0])]
    return _t([r[0] for r in res]), None, torch.tensor([r[1] for r in res], dtype=torch.int32)


def lat_error(f_cur, f_prev, A, Gamma, covprev):
    out = [orc.lat_error_terms(_n(f_cur[i]).reshape(-1, 1), _n(f_prev[i]).reshape(-1, 1), _n(A[i]), _n(Gamma[i]), _n(covprev[i]))
           for i in range(f_cur.shape[0])]
    T = f_cur.shape[1]                      # the oracle returns the full score; the kernel leaves -0.5 T log 2pi to its caller
    return _t([o + 0.5 * T * orc.LOG2PI for o in out]), torch.zeros(len(out), dtype=torch.int32)


def mniw_loglik(M, Sigma, m_mean, m_r_cov, scale, scale_is_diagonal=None):
    b, T, _ = M.shape
    pick = lambda a, i: None if a is None else (_n(a) if a.dim() == 2 else _n(a[i]))    # noqa: E731
    out = [orc.mniw_log_likelihood(_n(M[i]), _n(Sigma[i]), pick(m_mean, i), np.eye(T) if m_r_cov is None else pick(m_r_cov, i),
                                   pick(scale, i)) for i in range(b)]
    return _t(out), torch.zeros(b, dtype=torch.int32)


def hmm_messages(q, log_pi, log_trans, want_pair=True):
    qn, lt = _n(q), _n(log_trans)
    fmsg, marg = orc.hmm_forward(qn, _n(log_pi), lt)
    bmsg = orc.hmm_backward(qn, lt)
    pair = _t(orc.hmm_pair_coef(fmsg, bmsg, qn, lt)) if want_pair else None
    return _t(fmsg), _t(marg), _t(bmsg), pair


def hmm_local_terms(Q, log_pi, log_trans, want_pair=True):
    labels, pairs, last = [], [], []
    for b in range(Q.shape[0]):
        qn = loglik_rows(Q[b])[0]
        fmsg, _, bmsg, pair = hmm_messages(qn, log_pi, log_trans, True)
        lg = torch.log(fmsg * bmsg)
        labels.append(torch.argmax(lg, dim=1))
        flat = _n(pair).reshape(pair.shape[0], -1)
        first = np.array([0 if np.isnan(r).any() else int(np.argmax(r)) for r in flat], dtype=np.int64)
        pairs.append(torch.as_tensor(first))
        last.append(lg[-1])
    return torch.stack(labels), torch.stack(pairs), torch.stack(last)


def loglik_rows(q):
    c = torch.max(q, dim=1)[0]
    if bool(torch.any(torch.isinf(c))):
        return q, c
    return q - c[:, None], c


def assign(fmsg, bmsg, want_resp=False):
    lab = torch.argmax(torch.log(fmsg * bmsg), dim=1)
    if not want_resp:
        return lab
    resp = torch.zeros_like(fmsg)
    resp[torch.arange(fmsg.shape[0]), lab] = 1.0
    return lab, resp


class EagerPool:
    """Stand-in for hdpgpc_amd.online_chain.OnlinePool with the same interface: the candidates and the committed step computed
    with the one-by-one GPI_model methods (through the stand-ins above), so that the CPU tier exercises the host logic
    GPI_HDP.include_sample wraps around the pool (slot <-> cluster mapping, cumulative candidate tables, commit)."""

    def __init__(self, T, device, annealing, cap=8):
        self.slots = []

    @staticmethod
    def supports(g):
        return g.N >= 1

    def adopt(self, g):
        import types
        g._slot = len(self.slots)
        self.slots.append(types.SimpleNamespace(g=g))

    def begin_beat(self, y):
        self._y = y.reshape(-1, 1)
        sc = torch.stack([sl.g.log_sq_error(sl.g.x_basis, self._y, i=-1) for sl in self.slots])
        return sc, torch.zeros(len(self.slots), dtype=torch.int32)

    def candidates(self, t_new, q_lat_cols, indexes, extra=None):
        import types
        import hdpgpc_amd.GPI_HDP as H
        est, cols, lds = [], [], []
        hist = torch.empty((q_lat_cols.shape[0], 0))
        for sl in self.slots:
            cand = H.GPI_HDP.gpmodel_deepcopy(types.SimpleNamespace(verbose=False), sl.g)
            x, yy = cand.x_basis, self._y
            mean_, cov_, C_, Sigma_ = cand.smoother_weighted(x, yy, 1.0)
            est.append(cand.log_sq_error(x, yy, mean=mean_[-1], cov=cov_[-1], C=C_[-1], Sigma=Sigma_[-1], i=-1, first=len(cand.indexes) == 1))
            cand.include_weighted_sample(t_new, x, x, yy, 1.0)
            cand.backwards_pair(1.0)
            cand.bayesian_new_params(1.0)
            cols.append(cand.compute_q_lat_all(hist, h_ini=1.0))
            lds.append(cand.lds_param_likelihood_value())
        return torch.stack(est), torch.stack(cols, dim=1), lds

    def commit(self, g, index, x_train, y):
        g.include_weighted_sample(index, x_train, x_train, y, 1.0)
        g.bayesian_new_params(1.0)

    def finish_commit(self):
        pass


def install(monkeypatch):
    """Route the host layer to this file and to CPU tensors for one test."""
    import hdpgpc_amd.GPI_HDP as H
    import hdpgpc_amd.GPI_model as GM
    from hdpgpc_amd import ops

    for name in ("gram_rbf", "potrf_batched", "chol_inverse", "gemm_batched", "score_groups", "score_each", "lat_error",
                 "mniw_loglik", "hmm_messages", "hmm_local_terms", "loglik_rows", "assign"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(H.GPI_HDP, "_default_device", "cpu")
    from hdpgpc_amd import online_chain
    monkeypatch.setattr(online_chain, "OnlinePool", EagerPool)
    torch.set_num_threads(1)                # 90 x 90 products: one thread is 10x faster than eight
    from hdpgpc_amd import chain_batch
    monkeypatch.setattr(chain_batch, "_graphable", lambda job: False)      # no hipGraph chains on the CPU: one pass after the other
    real = GM.GPI_model.full_pass_weighted
    monkeypatch.setattr(GM.GPI_model, "full_pass_weighted",
                        lambda self, x, y, resp, q=None, q_lat=None, snr=None, use_graphs=True:
                        real(self, x, y, resp, q=q, q_lat=q_lat, snr=snr, use_graphs=False))
