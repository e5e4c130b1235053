"""GPU parity of the host-side mirror (hdpgpc_amd.GPI_model / GPI / amtgp_warping_system) against outputs of the
reference itself (tests/golden/*.npz): same method names, same numbers."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err, relclose
from oracle import hdpgpc_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hdpgpc_amd import ops
    from hdpgpc_amd.amtgp_warping_system import WarpPriorAMTGP
    from hdpgpc_amd.GPI import IterativeGaussianProcess, RBFWhiteKernel
    from hdpgpc_amd.GPI_model import GPI_model, matrix_normal_inv_wishart

RT = 1e-8


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def model_from(g, prefix="st_"):
    c, ell, noise = (float(v) for v in g[prefix + "theta"])
    m = GPI_model(RBFWhiteKernel(c, ell, noise), g[prefix + "x_basis"][:, None], bayesian=True)
    m.load_state(g[prefix + "f_star"], g[prefix + "Sigma"], g[prefix + "C"], g[prefix + "indexes"],
                 f_star_sm=g[prefix + "f_star_sm"], cov_f_sm=g[prefix + "cov_f_sm"], A=g[prefix + "A"],
                 Gamma=g[prefix + "Gamma"], A_def=g[prefix + "A_def"], Gamma_def=g[prefix + "Gamma_def"],
                 C_def=g[prefix + "C_def"], Sigma_def=g[prefix + "Sigma_def"], n0=float(g[prefix + "n0"]))
    return m


@pytest.mark.parametrize("tag", ["t30", "t45", "t90", "t45l3"])
def test_gpi_model_scoring_half(tag):
    g = golden(f"state_{tag}.npz")
    m = model_from(g)
    y = g["y"]
    n, T = y.shape
    xs = np.repeat(g["st_x_basis"][None, :, None], n, axis=0)
    yt = y[:, :, None]
    # a6 shared grid (with and without the `first` inflation) and irregular grids
    assert rel_err(m.compute_sq_err_all(xs, yt).cpu().numpy(), g["q_shared"]) < RT
    assert rel_err(m.compute_sq_err_all(xs, yt, no_first=True).cpu().numpy(), g["q_shared_nofirst"]) < RT
    assert rel_err(m.compute_sq_err_all(g["x_irr"][:, :, None], yt).cpu().numpy(), g["q_irr"]) < RT
    # a8
    ql = m.compute_q_lat_all(dev(xs)).cpu().numpy()
    assert rel_err(ql[m.indexes], g["q_lat"][m.indexes]) < RT
    assert np.all(np.delete(ql, m.indexes) == 0.0)
    assert abs(float(m.log_lat_error(1, 1.0)) - g["q_lat"][m.indexes[1]]) <= RT * abs(g["q_lat"][m.indexes[1]])
    # a5: the online-style calls
    for j in (0, n - 1):
        assert abs(float(m.log_sq_error(g["x_irr"][j][:, None], y[j][:, None], i=-1)) - g["lse_last"][j]) <= RT * abs(g["lse_last"][j])
        assert abs(float(m.log_sq_error(xs[j], y[j][:, None], i=-1)) - g["lse_last_shared"][j]) <= RT * abs(g["lse_last_shared"][j])
        assert abs(float(m.log_sq_error(g["x_irr"][j][:, None], y[j][:, None])) - g["lse_none"][j]) <= RT * abs(g["lse_none"][j])
        v = m.log_sq_error(g["x_irr"][j][:, None], y[j][:, None], mean=g["st_f_star_sm"][-2], cov=g["st_cov_f_sm"][-2],
                           C=g["st_C"][-2], Sigma=g["st_Sigma"][-2], i=0, first=True)
        assert abs(float(v) - g["lse_params_first"][j]) <= RT * abs(g["lse_params_first"][j])
    # a9
    assert abs(float(m.return_LDS_param_likelihood()) - float(g["lds_lik"])) <= RT * abs(float(g["lds_lik"]))
    assert abs(float(m.return_LDS_param_likelihood(first=True)) - float(g["lds_lik_first"])) <= RT * abs(float(g["lds_lik_first"]))
    prior = matrix_normal_inv_wishart(dev(g["st_C_def"]), torch.eye(T, dtype=torch.float64, device="cuda"), 5, dev(g["st_Sigma_def"]))
    v = prior.log_likelihood_MNIW(dev(g["st_C"][-1]), dev(g["st_Sigma"][-1]), float(g["st_n0"]))
    assert abs(float(v) - float(g["mniw_obs"])) <= RT * abs(float(g["mniw_obs"]))
    # a7 + a2: observe_last on the dense plotting grid (T* = 2T-1 > T)
    f, cov = m.observe_last(g["x_dense"][:, None])
    assert relclose(f.cpu().numpy()[:, 0], g["obs_last_f"], 1e-9)
    assert relclose(cov.cpu().numpy(), g["obs_last_cov"], 1e-9)


def test_pred_dist_and_latent_golden():
    g = golden("pred_dist.npz")
    for i in range(int(g["n_cases"])):
        c, ell, noise = (float(v) for v in g[f"c{i}_theta"])
        gp = IterativeGaussianProcess(RBFWhiteKernel(c, ell, noise), g[f"c{i}_xb"][:, None])
        f, cov = gp.pred_dist(g[f"c{i}_xp"][:, None], g[f"c{i}_xb"][:, None], g[f"c{i}_mean"][:, None], g[f"c{i}_Sigma"])
        assert relclose(f.cpu().numpy()[:, 0], g[f"c{i}_f"], 1e-8)
        assert relclose(cov.cpu().numpy(), g[f"c{i}_cov"], 1e-8)
        fl, covl = gp.pred_latent_dist(g[f"c{i}_xp"][:, None], g[f"c{i}_xb"][:, None], g[f"c{i}_mean"][:, None], g[f"c{i}_Sigma"])
        assert relclose(fl.cpu().numpy()[:, 0], g[f"c{i}_f_lat"], 1e-8)
        assert relclose(covl.cpu().numpy(), g[f"c{i}_cov_lat"], 1e-8)


def test_lml_a10_golden():
    g = golden("lml.npz")
    for i in range(int(g["n_cases"])):
        c, ell, noise = (float(v) for v in g[f"c{i}_theta"])
        gp = IterativeGaussianProcess(RBFWhiteKernel(c, ell, noise), g[f"c{i}_x"][:, None])
        v = gp.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], None)
        assert abs(v - float(g[f"c{i}_lml"])) <= 1e-8 * abs(float(g[f"c{i}_lml"]))
        v2 = gp.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], None, faithful=False)
        ref2 = orc.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], (c, ell, noise), faithful=False)
        assert abs(v2 - ref2) <= 1e-8 * abs(ref2)
        # value + gradient w.r.t. the log-parameters, as the reference forms it (GPI.py:1046-1051)
        theta = np.log([c, ell, noise])
        v3, grad = gp.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], None, theta=theta, eval_gradient=True)
        assert abs(v3 - float(g[f"c{i}_lml"])) <= 1e-8 * abs(float(g[f"c{i}_lml"]))
        assert np.allclose(grad, g[f"c{i}_grad"], rtol=1e-7, atol=0.0)
        v4, grad4 = gp.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], None, theta=theta, eval_gradient=True, faithful=False)
        ref4, gref4 = orc.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], (c, ell, noise), faithful=False, eval_gradient=True)
        assert abs(v4 - ref4) <= 1e-8 * abs(ref4) and np.allclose(grad4, gref4, rtol=1e-7, atol=1e-9)
        with pytest.raises(ValueError):
            gp.log_marginal_likelihood(g[f"c{i}_x"], g[f"c{i}_y"], None, eval_gradient=True)


def test_warp_prior_a11_golden():
    g = golden("warp_prior.npz")
    for i in range(int(g["n_cases"])):
        rho, omega, noise2, jitter, norm = (float(v) for v in g[f"c{i}_par"])
        wp = WarpPriorAMTGP(noise_warp=noise2, bound_noise_warp=(1e-10, 1e10), jitter=jitter, normalize_x=bool(norm))
        wp.theta = (rho, omega)
        v = wp.log_sq_error_batch(g[f"c{i}_x"], g[f"c{i}_W"])
        assert rel_err(v.cpu().numpy(), g[f"c{i}_val"]) < RT
        assert abs(float(wp.log_sq_error(g[f"c{i}_x"], g[f"c{i}_W"][0])) - float(g[f"c{i}_one"])) <= RT * abs(float(g[f"c{i}_one"]))


def test_offline_trace_q_matrix_and_assignments():
    """The q matrix of a real include_batch run of the reference (record 102, 60 beats, T=45, 5 clusters)."""
    g = golden("offline_r102_t45.npz")
    y, xb = g["y"], g["x_basis"]
    N, M, T = y.shape[0], int(g["M"]), xb.size
    q = np.zeros((N, M))
    xs = np.repeat(xb[None, :, None], N, axis=0)
    for m in range(M):
        c, ell, noise = (float(v) for v in g[f"m{m}_theta"])
        mod = GPI_model(RBFWhiteKernel(c, ell, noise), xb[:, None])
        S = g[f"m{m}_Sigma"].shape[0]
        mod.load_state(g[f"m{m}_means"], g[f"m{m}_Sigma"], np.repeat(np.eye(T)[None], S, 0), g[f"m{m}_indexes"])
        q[:, m] = mod.compute_sq_err_all(xs, y[:, :, None]).cpu().numpy()
    assert rel_err(q, g["q"]) < RT
    assert np.array_equal(np.argmax(q, axis=1), np.argmax(g["q"], axis=1))        # hard assignments bit-identical


def test_chol_rank1_config5():
    rng = np.random.default_rng(8)
    # every form of the kernel: odd T (one row segment per thread), even T < 192 (full lines through LDS), even T >= 192 (pipelined)
    for T, b in ((30, 3), (90, 4), (256, 2), (64, 5), (192, 3), (250, 2), (255, 2), (208, 9)):
        Q = rng.normal(size=(b, T, T))
        A = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
        L = np.linalg.cholesky(A)
        v = rng.normal(size=(b, T))
        al, be = rng.uniform(0.5, 1.5, b), rng.uniform(0.1, 2.0, b)
        Ln, info = ops.chol_rank1(dev(L), dev(v), al, be)
        assert int(info.abs().max()) == 0
        for k in range(b):
            ref = orc.chol_rank1_update(L[k], v[k], al[k], be[k])
            assert np.allclose(Ln[k].cpu().numpy(), ref, rtol=1e-10, atol=1e-11)
        # the strict upper triangle belongs to the caller: never read into a result, never written
        Lp = L + np.triu(np.full((T, T), np.nan), 1)
        Lq, info = ops.chol_rank1(dev(Lp), dev(v), al, be)
        assert int(info.abs().max()) == 0
        Lq = Lq.cpu().numpy()
        assert np.array_equal(np.tril(Lq), np.tril(Ln.cpu().numpy())) and np.all(np.isnan(Lq[:, np.triu_indices(T, 1)[0], np.triu_indices(T, 1)[1]]))
    # a factor that loses positive definiteness reports the first bad pivot, as the other forms do
    T = 256
    L = np.linalg.cholesky(np.eye(T) * 2.0)[None]
    Lb, info = ops.chol_rank1(dev(L), dev(np.ones((1, T))), np.array([1.0]), np.array([-1.0]))
    assert int(info[0]) == -1


def test_gemm_batched_shapes():
    rng = np.random.default_rng(2)
    for (m, n, k) in ((5, 7, 3), (16, 16, 16), (90, 1, 90), (33, 178, 90)):
        A, B = rng.normal(size=(3, m, k)), rng.normal(size=(3, k, n))
        assert np.allclose(ops.gemm_batched(dev(A), dev(B)).cpu().numpy(), A @ B, rtol=1e-12, atol=1e-12)
        At = np.ascontiguousarray(A.transpose(0, 2, 1))
        assert np.allclose(ops.gemm_batched(dev(At), dev(B), transA=True).cpu().numpy(), A @ B, rtol=1e-12, atol=1e-12)
        Bt = np.ascontiguousarray(B.transpose(0, 2, 1))
        assert np.allclose(ops.gemm_batched(dev(A), dev(Bt), transB=True).cpu().numpy(), A @ B, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag,members", [("t30", [2, 5, 6, 9, 12]), ("t45", [0, 1, 2, 3, 4, 7, 8, 11, 15, 16, 20]), ("t90", [0, 1, 3, 6])])
def test_producer_full_pass_weighted_reproduces_reference_state(tag, members):
    """the Kalman / RTS / MNIW recursion, driven exactly like the fixture generator drove the reference
    (tests/golden/make_golden.py: build_model), must reproduce every per-step list and the scores computed from them."""
    g = golden(f"state_{tag}.npz")
    y = g["y"]
    n, T = y.shape
    assert list(g["st_indexes"]) == members
    sigma, gamma = float(g["st_Sigma"][0][0, 0]), float(g["st_Gamma"][0][0, 0])
    m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), g["st_x_basis"][:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
    cond = m.GPR_dynamic(gamma, sigma)
    m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
    m.fixed_theta = tuple(float(v) for v in g["st_theta"])          # what the (out-of-scope) gpytorch fit left behind
    xs = np.repeat(g["st_x_basis"][None, :, None], n, axis=0)
    resp = np.zeros(n)
    resp[members] = 1.0
    q, q_lat = m.full_pass_weighted(xs, y[:, :, None], resp)
    assert m.indexes == members
    tol = 1e-9      # SURVEY section 7's gate; observed 1e-10 (gpurun_out/parity_observed.json)
    for name in ("f_star", "f_star_sm"):
        got = torch.stack(list(getattr(m, name))).cpu().numpy()[:, :, 0]
        assert relclose(got, g["st_" + name], tol), name
    for name in ("cov_f_sm", "A", "Gamma", "C", "Sigma"):
        got = torch.stack(list(getattr(m, name))).cpu().numpy()
        ref = g["st_" + name]
        assert got.shape == ref.shape, name
        assert relclose(got, ref, tol), name
    assert float(m.internal_params.n0) == float(g["st_n0"])
    assert rel_err(q.cpu().numpy(), g["q_shared"]) < 1e-9
    assert rel_err(q_lat.cpu().numpy()[members], g["q_lat"][members]) < 1e-9


def test_producer_step_forms_agree(monkeypatch):
    """The member step exists in three forms - one launch per dependency level (hgp_chain.hip, the default for T <= 128), one
    launch per product (128 < T <= 256; HGP_CHAIN_PER_PRODUCT=1 forces it) and the eager methods (use_graphs=False).  They
    order the same arithmetic differently (B^T A^-1 as (Z B)^T Z against B^T (Z^T Z)): every list must agree to 1e-9."""
    g = golden("state_t45.npz")
    y = g["y"]
    n, T = y.shape
    members = [int(v) for v in g["st_indexes"]]
    sigma, gamma = float(g["st_Sigma"][0][0, 0]), float(g["st_Gamma"][0][0, 0])
    xs = np.repeat(g["st_x_basis"][None, :, None], n, axis=0)
    resp = np.zeros(n)
    resp[members] = 1.0
    states = []
    for form in ("levels", "products", "eager"):
        if form == "products":
            monkeypatch.setenv("HGP_CHAIN_PER_PRODUCT", "1")
        else:
            monkeypatch.delenv("HGP_CHAIN_PER_PRODUCT", raising=False)
        m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), g["st_x_basis"][:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
        cond = m.GPR_dynamic(gamma, sigma)
        m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
        m.fixed_theta = tuple(float(v) for v in g["st_theta"])
        q, q_lat = m.full_pass_weighted(xs, y[:, :, None], resp, use_graphs=(form != "eager"))
        assert (getattr(m, "graph_replays", 0) > 0) == (form != "eager")
        states.append({k: torch.stack(list(getattr(m, k))).cpu().numpy() for k in ("f_star", "f_star_sm", "cov_f", "cov_f_sm", "A", "Gamma", "C", "Sigma")}
                      | {"q": q.cpu().numpy(), "q_lat": q_lat.cpu().numpy()[members], "n0": np.asarray(float(m.internal_params.n0))})
    for other in states[1:]:
        for k, ref in states[0].items():
            assert np.allclose(other[k], ref, rtol=1e-9, atol=1e-9 * max(np.abs(ref).max(), 1e-300)), k


def test_soft_members_are_skipped_like_the_reference():
    """A responsibility in (0.99, 1) (e.g. 0.9999 from the variational step) makes a segment 'active' but h != 1: the
    reference then skips it entirely - include_sample(posterior=False), backwards_pair and bayesian_new_params are no-ops
    (GPI_model.py:353-375,705-716,972).  The state must equal the hard members' state of the fixture, not crash."""
    g = golden("state_t45.npz")
    y = g["y"]
    n, T = y.shape
    members = [int(v) for v in g["st_indexes"]]
    sigma, gamma = float(g["st_Sigma"][0][0, 0]), float(g["st_Gamma"][0][0, 0])
    xs = np.repeat(g["st_x_basis"][None, :, None], n, axis=0)
    for use_graphs in (True, False):
        m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), g["st_x_basis"][:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
        cond = m.GPR_dynamic(gamma, sigma)
        m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
        m.fixed_theta = tuple(float(v) for v in g["st_theta"])
        resp = np.zeros(n)
        resp[members] = 1.0
        resp[[5, 9, 21]] = 0.995                       # soft members between hard ones
        q, q_lat = m.full_pass_weighted(xs, y[:, :, None], resp, use_graphs=use_graphs)
        assert m.indexes == members
        for name in ("f_star", "f_star_sm"):
            got = torch.stack(list(getattr(m, name))).cpu().numpy()[:, :, 0]
            assert relclose(got, g["st_" + name], 1e-9), name
        for name in ("A", "Gamma", "C", "Sigma"):
            got = torch.stack(list(getattr(m, name))).cpu().numpy()
            assert relclose(got, g["st_" + name], 1e-9), name
        assert rel_err(q.cpu().numpy(), g["q_shared"]) < 1e-9


def test_pending_info_names_the_failing_step():
    """_check_pending maps the flat index of the first non-zero LAPACK info back to the entry it belongs to and raises
    torch.linalg.LinAlgError (what callers of the reference catch, GPI_model.py:1068)."""
    m = GPI_model(RBFWhiteKernel(1.0, 1.0, 0.1), np.arange(8.0)[:, None])
    z = lambda k: torch.zeros(k, dtype=torch.int32, device="cuda")  # noqa: E731
    bad = z(3)
    bad[1] = 4
    m._pending = [("first", z(2)), ("second", z(5)), ("third", bad), ("fourth", z(1))]
    with pytest.raises(torch.linalg.LinAlgError, match="third"):
        m._check_pending()
    assert m._pending == []


def test_online_side_producer_pieces():
    """posterior_weighted / smoother_weighted (GPI_model.py:561-582,726-738: what GPI_HDP.estimate_new calls) on the shared
    grid and on an irregular grid (the K_cov path of GPI.posterior, GPI.py:124-133), h = 1 and h < 1, then
    reinit_LDS / reinit_GP and a second pass through the same fitted model - against the reference's outputs."""
    g, e = golden("state_t45.npz"), golden("producer_extra.npz")
    y = g["y"]
    n, T = y.shape
    members = [int(v) for v in g["st_indexes"]]
    sigma, gamma = float(g["st_Sigma"][0][0, 0]), float(g["st_Gamma"][0][0, 0])
    m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), g["st_x_basis"][:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
    cond = m.GPR_dynamic(gamma, sigma)
    m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
    m.fixed_theta = tuple(float(v) for v in g["st_theta"])
    xs = np.repeat(g["st_x_basis"][None, :, None], n, axis=0)
    resp = np.zeros(n)
    resp[members] = 1.0
    m.full_pass_weighted(xs, y[:, :, None], resp)
    tol = 1e-9

    def close(a, b):
        return relclose(a, b, tol)

    for tag, xx in (("shared", xs[21]), ("irr", e["x_irr"][:, None])):
        for h in (1.0, 0.6):
            f, c = m.posterior_weighted(xx, y[21][:, None], h)
            assert close(f.cpu().numpy()[:, 0], e[f"pw_{tag}_h{h}_f"]) and close(c.cpu().numpy(), e[f"pw_{tag}_h{h}_cov"]), (tag, h)
        f, c = m.posterior_weighted(xx, y[21][:, None], 1.0, t=3)
        assert close(f.cpu().numpy()[:, 0], e[f"pw_{tag}_t3_f"]) and close(c.cpu().numpy(), e[f"pw_{tag}_t3_cov"]), tag
    means, covs, C, Sigma = m.smoother_weighted(xs[21], y[21][:, None], 1.0)
    assert [len(means), len(covs), len(C), len(Sigma)] == list(e["sw_len"])
    assert close(means[-1].cpu().numpy()[:, 0], e["sw_last_f"]) and close(covs[-1].cpu().numpy(), e["sw_last_cov"])
    v = m.log_sq_error(xs[21], y[21][:, None], mean=means[-1], cov=covs[-1], C=C[-1], Sigma=Sigma[-1], i=0, first=True)
    assert abs(float(v) - float(e["lse_candidate"])) <= tol * abs(float(e["lse_candidate"]))
    assert m.find_closest_lower(0) == 0 and m.find_closest_lower(6) == 4 and m.find_closest_lower(23) == len(members) - 1
    m.reinit_LDS(save_last=False)
    m.reinit_GP(save_last=False)
    assert m.N == int(e["re_N"]) and int(m.fitted) == int(e["re_fitted"]) and m.indexes == []
    resp = np.zeros(n)
    resp[[5, 6, 9, 10, 12]] = 1.0
    q2, ql2 = m.full_pass_weighted(xs, y[:, :, None], resp)
    assert rel_err(q2.cpu().numpy(), e["re_q"]) < 1e-9
    assert rel_err(ql2.cpu().numpy()[[5, 6, 9, 10, 12]], e["re_q_lat"][[5, 6, 9, 10, 12]]) < 1e-9
    assert close(m.Sigma[-1].cpu().numpy(), e["re_Sigma_last"]) and close(m.f_star_sm[-1].cpu().numpy()[:, 0], e["re_f_sm_last"])


def test_replay_offline_trace_from_labels():
    """labels -> recursion -> q matrix -> hard assignments, on the reference's own include_batch result (record 102,
    60 beats, T = 45, 5 clusters): a fresh model per cluster is driven over the cluster's final members and must end
    in the reference's state, score matrix and arg-max assignments."""
    g = golden("offline_r102_t45.npz")
    y, xb = g["y"], g["x_basis"]
    N, M, T = y.shape[0], int(g["M"]), xb.size
    xs = np.repeat(xb[None, :, None], N, axis=0)
    q = np.zeros((N, M))
    for mi in range(M):
        members = [int(v) for v in g[f"m{mi}_indexes"]]
        Sig0, Gam0 = g[f"m{mi}_Sigma"][0], g[f"m{mi}_Gamma0"]
        mod = GPI_model(RBFWhiteKernel(300.0, 3.0, 1e-6), xb[:, None], annealing=True, bayesian=True,
                        free_deg_MNIV=int(g[f"m{mi}_free_deg"]))
        mod.initial_conditions(ini_A=g[f"m{mi}_A0"], ini_Gamma=Gam0, ini_C=g[f"m{mi}_C0"], ini_Sigma=Sig0)
        mod.fixed_theta = tuple(float(v) for v in g[f"m{mi}_theta"])
        resp = np.zeros(N)
        resp[members] = 1.0
        qm, qlat = mod.full_pass_weighted(xs, y[:, :, None], resp)
        S = len(mod.f_star)
        means = np.stack([(mod.C[min(i, len(mod.C) - 1)] @ mod.f_star[i]).cpu().numpy().reshape(-1) for i in range(S)])
        assert relclose(means, g[f"m{mi}_means"], 1e-9)
        Sg = torch.stack(list(mod.Sigma)).cpu().numpy()
        assert relclose(Sg, g[f"m{mi}_Sigma"], 1e-9)
        assert float(mod.internal_params.n0) == float(g[f"m{mi}_n0"])
        assert rel_err(qlat.cpu().numpy()[members], g[f"m{mi}_q_lat"][members]) < 1e-9
        q[:, mi] = qm.cpu().numpy()
    assert rel_err(q, g["q"]) < 1e-9
    assert np.array_equal(np.argmax(q, axis=1), np.argmax(g["q"], axis=1))


def test_gemm_addend_epilogue_and_out():
    rng = np.random.default_rng(21)
    A, B, D = rng.normal(size=(3, 50, 70)), rng.normal(size=(3, 70, 33)), rng.normal(size=(50, 33))
    out = torch.empty((3, 50, 33), dtype=torch.float64, device="cuda")
    r = ops.gemm_batched(dev(A), dev(B), alpha=-0.5, add=dev(D), beta=2.0, out=out)
    assert r is out
    assert np.allclose(out.cpu().numpy(), -0.5 * A @ B + 2.0 * D, rtol=1e-12, atol=1e-12)
    Dm = rng.normal(size=(3, 33, 33))
    r2 = ops.gemm_batched(dev(B), dev(B), transA=True, add=dev(Dm))
    assert np.allclose(r2.cpu().numpy(), B.transpose(0, 2, 1) @ B + Dm, rtol=1e-12, atol=1e-12)


def test_add_diag_mean_is_the_mniw_jitter():
    rng = np.random.default_rng(22)
    R, S = rng.normal(size=(2, 37, 37)), rng.normal(size=(2, 37, 37))
    got = ops.add_diag_mean(dev(R), dev(S), 1e-2).cpu().numpy()
    for b in range(2):
        jit = 1e-2 * max(np.mean(np.abs(np.diag(S[b]))), np.finfo(np.float64).eps)
        assert np.allclose(got[b], R[b] + jit * np.eye(37), rtol=0, atol=1e-15)


@pytest.mark.parametrize("T,n", [(90, 40), (33, 7), (96, 3)])
def test_rts_chain_kernel_matches_the_sequential_recursion(T, n):
    """hgp_rts_chain_f64: m_t += J_t (m_{t+1} - A_t m_t), C_t += J_t (C_{t+1} - P_t) J_t^T for t = n-2 .. 0, one launch."""
    rng = np.random.default_rng(T + n)
    J = rng.normal(size=(n - 1, T, T)) * 0.1
    P = rng.normal(size=(n - 1, T, T))
    AM = rng.normal(size=(n - 1, T))
    M = rng.normal(size=(n, T))
    Cv = rng.normal(size=(n, T, T))
    Mg, Cg = dev(M), dev(Cv)
    ops.rts_chain(dev(J), dev(P), dev(AM), Mg, Cg)
    Mr, Cr = M.copy(), Cv.copy()
    for t in range(n - 2, -1, -1):
        Mr[t] = Mr[t] + J[t] @ (Mr[t + 1] - AM[t])
        Cr[t] = Cr[t] + J[t] @ (Cr[t + 1] - P[t]) @ J[t].T
    assert np.allclose(Mg.cpu().numpy(), Mr, rtol=1e-11, atol=1e-11)
    assert np.allclose(Cg.cpu().numpy(), Cr, rtol=1e-11, atol=1e-11)


def test_hmm_messages_8f3_golden_and_large():
    """hgp_hmm_messages_f64 against the reference's forward / backward / coupled_state_coef outputs (record-102 run),
    then against the oracle on a long synthetic chain (N = 2 272, K = 9) with -inf entries as birth proposals leave them."""
    g = golden("hmm_r102_t45.npz")
    f, m, b, c = ops.hmm_messages(dev(g["q"]), dev(g["log_pi"]), dev(g["log_trans"]))
    assert np.allclose(f.cpu().numpy(), g["fmsg"], rtol=1e-11, atol=1e-300)
    assert np.allclose(m.cpu().numpy(), g["margPrObs"], rtol=1e-11, atol=0)
    assert np.allclose(b.cpu().numpy(), g["bmsg"], rtol=1e-11, atol=1e-300)
    cr, cg = g["log_respPair"], c.cpu().numpy()
    fin = np.isfinite(cr)
    assert np.array_equal(np.isfinite(cg), fin) and np.allclose(cg[fin], cr[fin], rtol=1e-10, atol=1e-10)
    assert np.array_equal(np.argmax(f.cpu().numpy() * b.cpu().numpy(), axis=1), np.argmax(g["fmsg"] * g["bmsg"], axis=1))
    rng = np.random.default_rng(9)
    N, K = 2272, 9
    q = rng.normal(size=(N, K)) * 30 - 100
    lt = np.log(rng.dirichlet(np.ones(K) * 0.3, size=K))
    lt[:, -1] = -np.inf                                   # a state nobody has entered yet (compute_trans_A pads with -inf)
    lp = np.log(rng.dirichlet(np.ones(K)))
    f, m, b, c = ops.hmm_messages(dev(q), dev(lp), dev(lt))
    fr, mr = orc.hmm_forward(q, lp, lt)
    br = orc.hmm_backward(q, lt)
    assert np.allclose(f.cpu().numpy(), fr, rtol=1e-9, atol=1e-300) and np.allclose(m.cpu().numpy(), mr, rtol=1e-9, atol=0)
    assert np.allclose(b.cpu().numpy(), br, rtol=1e-9, atol=1e-300)
    cr = orc.hmm_pair_coef(fr, br, q, lt)
    fin = np.isfinite(cr)
    cg = c.cpu().numpy()
    assert np.array_equal(np.isfinite(cg), fin) and np.allclose(cg[fin], cr[fin], rtol=1e-8, atol=1e-8)


def test_online_path_calls_at_T256():
    """BASELINE configs[4]: the online path on beats resampled to T = 256.  The mirror builds the cluster state itself
    (Kalman / RTS / MNIW recursion on the cooperative T > 128 kernels) and answers what one online step asks of a cluster
    (GPI_HDP.py:1970-2197): compute_q_lat_all, log_sq_error(i=-1) on the shared and on an irregular grid,
    return_LDS_param_likelihood, the candidate posterior - against the reference's own outputs."""
    g = golden("online_t256.npz")
    y = g["y"]
    n, T = y.shape
    assert T == 256
    members = [int(v) for v in g["members"]]
    sigma, gamma = float(g["sigma0"]), float(g["gamma0"])
    m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), np.arange(float(T))[:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
    cond = m.GPR_dynamic(gamma, sigma)
    m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
    m.fixed_theta = tuple(float(v) for v in g["theta"])
    xs = np.repeat(np.arange(float(T))[None, :, None], n, axis=0)
    resp = np.zeros(n)
    resp[members] = 1.0
    q, q_lat = m.full_pass_weighted(xs, y[:, :, None], resp)
    tol = 1e-8      # observed 1.0e-9 (T = 256, 2 x 256-step recursion behind every number)
    assert m.indexes == members and float(m.internal_params.n0) == float(g["n0"])
    assert rel_err(q.cpu().numpy(), g["q_shared"]) < tol
    assert rel_err(q_lat.cpu().numpy()[members], g["q_lat"][members]) < tol
    S = m.Sigma[-1].cpu().numpy()
    assert np.allclose(np.diag(S), g["Sigma_last_diag"], rtol=tol) and relclose(S[7], g["Sigma_last_row7"], tol)
    assert relclose(m.f_star_sm[-1].cpu().numpy()[:, 0], g["f_star_sm_last"], tol)
    assert abs(float(m.return_LDS_param_likelihood()) - float(g["lds_lik"])) <= tol * abs(float(g["lds_lik"]))
    for j in (0, 8, 9):
        assert abs(float(m.log_sq_error(xs[j], y[j][:, None], i=-1)) - g["lse_last_shared"][j]) <= tol * abs(g["lse_last_shared"][j])
    for k, j in enumerate((8, 9)):
        v = float(m.log_sq_error(g["x_irr"][j][:, None], y[j][:, None], i=-1))
        assert abs(v - g["lse_last_irr"][k]) <= tol * abs(g["lse_last_irr"][k])
    f, c = m.posterior_weighted(xs[8], y[8][:, None], 1.0)
    c = c.cpu().numpy()
    assert relclose(f.cpu().numpy()[:, 0], g["pw_f"], tol)
    assert np.allclose(np.diag(c), g["pw_cov_diag"], rtol=tol) and relclose(c[100], g["pw_cov_row100"], tol)
    means, covs, C, Sigma = m.smoother_weighted(xs[8], y[8][:, None], 1.0)
    v = m.log_sq_error(xs[8], y[8][:, None], mean=means[-1], cov=covs[-1], C=C[-1], Sigma=Sigma[-1], i=0, first=True)
    assert abs(float(v) - float(g["lse_candidate"])) <= tol * abs(float(g["lse_candidate"]))
