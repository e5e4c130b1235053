"""Pin the CPU oracle (oracle/hdpgpc_oracle.py) against vectors produced by the reference itself.

Every .npz under tests/golden was written by tests/golden/make_golden.py, which imports the reference
and records its outputs.  Tolerance: 1e-9 relative; observed errors are
at the 1e-13 level or below.
"""
import numpy as np
import pytest

from conftest import golden, rel_err
from oracle import hdpgpc_oracle as orc

RTOL = 1e-9


def _state(g, prefix="st_"):
    return orc.ClusterState(
        g[prefix + "x_basis"], g[prefix + "theta"], list(g[prefix + "f_star"]), list(g[prefix + "Sigma"]),
        list(g[prefix + "C"]), list(g[prefix + "indexes"]), f_star_sm=list(g[prefix + "f_star_sm"]),
        cov_f_sm=list(g[prefix + "cov_f_sm"]), A=list(g[prefix + "A"]), Gamma=list(g[prefix + "Gamma"]))


def test_gram_a1():
    g = golden("gram.npz")
    for i in range(int(g["n_cases"])):
        c, ell, noise = g[f"c{i}_theta"]
        X, Y = g[f"c{i}_X"], g[f"c{i}_Y"]
        assert np.allclose(orc.gram_rbf(X, None, c, ell, noise), g[f"c{i}_K_one"], rtol=1e-13, atol=1e-300)
        assert np.allclose(orc.gram_rbf(X, Y, c, ell), g[f"c{i}_K_two"], rtol=1e-13, atol=1e-300)
        assert np.allclose(orc.gram_rbf(X, X, c, ell), g[f"c{i}_K_self"], rtol=1e-13, atol=1e-300)


def test_chol_and_shared_score_a3_a4():
    g = golden("score_shared.npz")
    for i in range(int(g["n_cases"])):
        cov, mean, Y = g[f"c{i}_cov"], g[f"c{i}_mean"], g[f"c{i}_Y"]
        assert rel_err(orc.chol_spd(cov)[np.tril_indices(cov.shape[0])],
                       g[f"c{i}_L"][np.tril_indices(cov.shape[0])]) < 1e-9 or \
            np.allclose(orc.chol_spd(cov), g[f"c{i}_L"], rtol=1e-9, atol=1e-12)
        assert rel_err(orc.gaussian_score_shared_cov(Y, mean, cov), g[f"c{i}_score"]) < RTOL
        # split form used by the kernels agrees with the reference's cholesky_solve form
        for b in range(min(Y.shape[0], 3)):
            quad, _ = orc.quad_logdet(Y[b] - mean, cov)
            ref = g[f"c{i}_score"][b]
            assert abs((-0.5 * quad - 0.5 * cov.shape[0] * orc.LOG2PI) - ref) <= RTOL * abs(ref)


def test_pred_dist_a2():
    g = golden("pred_dist.npz")
    for i in range(int(g["n_cases"])):
        th = tuple(g[f"c{i}_theta"])
        f, cov = orc.pred_dist(g[f"c{i}_xp"], g[f"c{i}_xb"], g[f"c{i}_mean"], g[f"c{i}_Sigma"], th)
        assert np.allclose(f[:, 0], g[f"c{i}_f"], rtol=RTOL, atol=1e-9 * np.abs(g[f"c{i}_f"]).max())
        assert np.allclose(cov, g[f"c{i}_cov"], rtol=RTOL, atol=1e-9 * np.abs(g[f"c{i}_cov"]).max())
        fl, covl = orc.pred_latent_dist(g[f"c{i}_xp"], g[f"c{i}_xb"], g[f"c{i}_mean"], g[f"c{i}_Sigma"], th)
        assert np.allclose(fl[:, 0], g[f"c{i}_f_lat"], rtol=1e-8, atol=1e-8 * np.abs(g[f"c{i}_f_lat"]).max())
        assert np.allclose(covl, g[f"c{i}_cov_lat"], rtol=1e-8, atol=1e-8 * np.abs(g[f"c{i}_cov_lat"]).max())


@pytest.mark.parametrize("tag", ["t30", "t45", "t90", "t45l3"])
def test_state_paths_a5_to_a9(tag):
    g = golden(f"state_{tag}.npz")
    st = _state(g)
    y = g["y"]
    n, T = y.shape
    xs = np.repeat(st.x_basis[None, :], n, axis=0)
    assert rel_err(orc.compute_sq_err_all(st, xs, y), g["q_shared"]) < RTOL
    assert rel_err(orc.compute_sq_err_all(st, xs, y, no_first=True), g["q_shared_nofirst"]) < RTOL
    assert rel_err(orc.compute_sq_err_all(st, g["x_irr"], y), g["q_irr"]) < RTOL
    assert rel_err(orc.compute_q_lat_all(st, n)[st.indexes], g["q_lat"][st.indexes]) < RTOL
    # i = -1 resolves to the last state by negative indexing (GPI_model.py:653-656)
    mean_last = st.C[-1] @ st.f_star[-1]
    lse = [orc.log_sq_error_state(g["x_irr"][j], y[j], st.x_basis, mean_last, st.Sigma[-1], st.theta)[0] for j in range(n)]
    assert rel_err(lse, g["lse_last"]) < RTOL
    lse = [orc.log_sq_error_state(xs[j], y[j], st.x_basis, mean_last, st.Sigma[-1], st.theta)[0] for j in range(n)]
    assert rel_err(lse, g["lse_last_shared"]) < RTOL
    # i=None -> step_forward_last: C[-1] f_star_sm[-1], Sigma[-1]  (GPI_model.py:595-615)
    mean_sm = st.C[-1] @ st.f_star_sm[-1]
    lse = [orc.log_sq_error_state(g["x_irr"][j], y[j], st.x_basis, mean_sm, st.Sigma[-1], st.theta)[0] for j in range(n)]
    assert rel_err(lse, g["lse_none"]) < RTOL
    # explicit params + first (GPI_HDP.py:2835-2841 style call)
    mean_p = st.C[-2] @ st.f_star_sm[-2]
    ini = 1e-2 * float(np.mean(np.diag(st.Sigma[0])))
    lse = [orc.log_sq_error_state(g["x_irr"][j], y[j], st.x_basis, mean_p, st.Sigma[-2], st.theta, ini)[0] for j in range(n)]
    assert rel_err(lse, g["lse_params_first"]) < RTOL
    # a9
    args = (st.A[-1], st.Gamma[-1], st.C[-1], st.Sigma[-1], g["st_A_def"], g["st_Gamma_def"], g["st_C_def"], g["st_Sigma_def"])
    assert abs(orc.lds_param_likelihood(*args) - float(g["lds_lik"])) <= RTOL * abs(float(g["lds_lik"]))
    assert abs(orc.lds_param_likelihood(*args, first=True) - float(g["lds_lik_first"])) <= RTOL * abs(float(g["lds_lik_first"]))
    m = orc.mniw_log_likelihood(st.C[-1], st.Sigma[-1], g["st_C_def"], np.eye(T), g["st_Sigma_def"])
    assert abs(m - float(g["mniw_obs"])) <= RTOL * abs(float(g["mniw_obs"]))
    # observe_last on the dense plotting grid (C[-1] f_star_sm[-1])
    f, cov = orc.pred_dist(g["x_dense"], st.x_basis, mean_sm, st.Sigma[-1], st.theta)
    assert np.allclose(f[:, 0], g["obs_last_f"], rtol=1e-8, atol=1e-8 * np.abs(g["obs_last_f"]).max())
    assert np.allclose(cov, g["obs_last_cov"], rtol=1e-8, atol=1e-8 * np.abs(g["obs_last_cov"]).max())


def test_lml_a10():
    g = golden("lml.npz")
    for i in range(int(g["n_cases"])):
        v = orc.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], tuple(g[f"c{i}_theta"]), faithful=True)
        assert abs(v - float(g[f"c{i}_lml"])) <= 1e-9 * abs(float(g[f"c{i}_lml"]))
        # value + gradient w.r.t. (log c, log ell, log noise), GPI.py:1046-1051
        v2, grad = orc.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], tuple(g[f"c{i}_theta"]), faithful=True,
                                               eval_gradient=True)
        assert v2 == v
        assert np.allclose(grad, g[f"c{i}_grad"], rtol=1e-9, atol=0.0)


def test_warp_prior_a11():
    g = golden("warp_prior.npz")
    for i in range(int(g["n_cases"])):
        rho, omega, noise2, jitter, norm = g[f"c{i}_par"]
        v = orc.warp_log_sq_error_batch(g[f"c{i}_x"], g[f"c{i}_W"], rho, omega, noise2, jitter, bool(norm))
        assert rel_err(v, g[f"c{i}_val"]) < RTOL
        assert abs(v[0] - float(g[f"c{i}_one"])) <= RTOL * abs(float(g[f"c{i}_one"]))


def test_offline_trace_assignments():
    """q matrix of a real include_batch run (record 102, 60 beats, T=45) and its hard assignments."""
    g = golden("offline_r102_t45.npz")
    y, xb = g["y"], g["x_basis"]
    N, M = y.shape[0], int(g["M"])
    q = np.zeros((N, M))
    for m in range(M):
        means, Sig, idx = g[f"m{m}_means"], g[f"m{m}_Sigma"], g[f"m{m}_indexes"]
        # state with identity C and f_star := C_i f_i (the fixture stores the observation mean directly)
        eye = np.eye(xb.size)
        st = orc.ClusterState(xb, g[f"m{m}_theta"], list(means), list(Sig), [eye] * Sig.shape[0], list(idx))
        q[:, m] = orc.compute_sq_err_all(st, np.repeat(xb[None], N, 0), y)
    assert rel_err(q, g["q"]) < RTOL
    # each beat's own cluster scores it best among the clusters it was assigned from
    assert np.array_equal(np.argmax(q, axis=1) == g["resp_assigned"], np.argmax(g["q"], axis=1) == g["resp_assigned"])


def test_hmm_messages_8f3():
    """forward / backward / coupled_state_coef of the switching variable (GPI_HDP.py:3546-3700) on the final
    variational observations of the record-102 run."""
    g = golden("hmm_r102_t45.npz")
    f, m = orc.hmm_forward(g["q"], g["log_pi"], g["log_trans"])
    b = orc.hmm_backward(g["q"], g["log_trans"])
    c = orc.hmm_pair_coef(g["fmsg"], g["bmsg"], g["q"], g["log_trans"])
    assert np.allclose(f, g["fmsg"], rtol=1e-12, atol=0) and np.allclose(m, g["margPrObs"], rtol=1e-12, atol=0)
    assert np.allclose(b, g["bmsg"], rtol=1e-12, atol=0)
    fin = np.isfinite(g["log_respPair"])
    assert np.array_equal(np.isfinite(c), fin) and np.allclose(c[fin], g["log_respPair"][fin], rtol=1e-12, atol=0)


def test_pairs_ill_conditioned_lengthscales():
    """a2 + a5 at the drivers' ini_lengthscale = 3.0 (and 2.5) on irregular grids, T = 45 / 90 / 128 and T* != T: the
    reference's own GPI_model.log_sq_error outputs (tests/golden/pairs_ill.npz)."""
    g = golden("pairs_ill.npz")
    for i in range(int(g["n_cases"])):
        x, y, xb, th = g[f"c{i}_x"], g[f"c{i}_y"], g[f"c{i}_xb"], g[f"c{i}_theta"]
        score, _, _ = orc.loglik_pairs(x, y, xb, th, g[f"c{i}_mean"], g[f"c{i}_Sigma"])
        assert rel_err(score, g[f"c{i}_score"]) < RTOL
        fn = np.full(score.shape, float(g[f"c{i}_ini_noise"]))
        score_f, _, _ = orc.loglik_pairs(x, y, xb, th, g[f"c{i}_mean"], g[f"c{i}_Sigma"], first_noise=fn)
        assert rel_err(score_f, g[f"c{i}_score_first"]) < RTOL
