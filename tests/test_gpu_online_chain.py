"""hdpgpc_amd/online_chain.py (BASELINE configs[4]): the online step's candidates and commits on persistent chains against the
one-by-one GPI_model methods (GPI_model.py:325-375,705-716,966-1115 restated call by call) on the same state.

The end-to-end gate is tests/test_gpu_include_sample.py (the reference's own traces); here the pool's pieces are held to the
one-by-one path directly: estimate_new's score, the candidate's latent-transition column, its MNIW parameter likelihood, and the
state a commit leaves behind, for clusters with one, two and several members, at T = 90 (riding inversions) and T = 144 / 256
(cooperative inversions with the workspace)."""
import types

import numpy as np
import pytest
import torch

from conftest import golden, rel_err, relclose

pytestmark = pytest.mark.gpu


def _models(T, sizes, seed=3):
    """Clusters grown one beat at a time with the one-by-one methods (include_weighted_sample + bayesian_new_params, as the
    online commit does), from resampled beats of record 100."""
    import hdpgpc.GPI_HDP as hdpgp
    y90 = golden("mitbih100_lead0.npz")["y"][:64]
    tt = np.linspace(0, y90.shape[1] - 1, T)
    Y = np.stack([np.interp(tt, np.arange(y90.shape[1]), r) for r in y90])
    xb = np.arange(float(T))[:, None]
    std, std_dif = float(np.std(Y[:30], axis=0).mean()), float(np.std(np.diff(Y[:30], axis=0), axis=0).mean())
    sw = hdpgp.GPI_HDP(xb, x_basis_warp=xb[::2], n_outputs=1, ini_lengthscale=3.0, bound_lengthscale=(1.0, 20.0), ini_gamma=std_dif,
                       ini_sigma=std, ini_outputscale=300.0, noise_warp=std * 0.1, bound_sigma=(std * 1e-5, std * 2),
                       bound_gamma=(std_dif * 1e-5, std_dif * 2), bound_noise_warp=(std * 0.01, std * 0.02), verbose=False,
                       max_models=100, bayesian_params=True, free_deg_MNIV=20)
    sw.fixed_theta = (341.0, 1.2, 4.66)
    rng = np.random.default_rng(seed)
    x = torch.as_tensor(xb, device=sw.device)
    models, used = [], 0
    for n in sizes:
        g = sw.create_gp_default()
        for j in range(n):
            yy = torch.as_tensor(Y[used] + 0.05 * rng.standard_normal(T), device=sw.device).reshape(-1, 1)
            g.include_weighted_sample(used, x, x, yy, 1.0)
            g.bayesian_new_params(1.0)
            used += 1
        models.append(g)
    return sw, models, x, torch.as_tensor(Y[used], device=sw.device).reshape(-1, 1), used


def _one_by_one(sw, g, x, y, t_new, T_all):
    cand = sw.gpmodel_deepcopy(g)
    mean_, cov_, C_, Sigma_ = cand.smoother_weighted(x, y, 1.0)
    est = cand.log_sq_error(x, y, mean=mean_[-1], cov=cov_[-1], C=C_[-1], Sigma=Sigma_[-1], i=-1, first=len(cand.indexes) == 1)
    cand.include_weighted_sample(t_new, x, x, y, 1.0)
    cand.backwards_pair(1.0)
    cand.bayesian_new_params(1.0)
    return float(est), cand.compute_q_lat_all(torch.empty((T_all, 0))).cpu().numpy(), float(cand.lds_param_likelihood_value()), cand


@pytest.mark.parametrize("T", [90, 144, 256])
def test_candidates_and_commit_match_the_one_by_one_path(T):
    from hdpgpc_amd.online_chain import OnlinePool
    sizes = [1, 2, 5, 3] if T > 128 else [1, 2, 7, 3, 4, 2, 1, 6, 2]         # the second set grows the pool past its first capacity
    sw, models, x, y, t_new = _models(T, sizes)
    T_all = t_new + 1
    hist = torch.empty((T_all, 0))
    ref = [_one_by_one(sw, g, x, y, t_new, T_all) for g in models]
    pool = OnlinePool(T, sw.device, sw.annealing_def, cap=4)
    for g in models:
        assert pool.supports(g)
        pool.adopt(g)
    cols0 = torch.stack([g.compute_q_lat_all(hist) for g in models], dim=1).contiguous()
    before = [[t.clone() for t in (g.f_star_sm[-1], g.cov_f_sm[-1], g.A[-1], g.Sigma[-1], g.internal_params.scale)] for g in models]
    sc, info = pool.begin_beat(y[:, 0])
    assert int(info.abs().max()) == 0
    for c, g in enumerate(models):                 # the beat under every cluster's last state: log_sq_error(x, y, i=-1)
        want = float(g.log_sq_error(x, y, i=-1))
        assert abs(float(sc[c]) - want) <= 1e-9 * abs(want)
    est, cols, lds = pool.candidates(t_new, cols0, [g.indexes for g in models])
    est, cols = est.cpu().numpy(), cols.cpu().numpy()
    for c, (e_ref, col_ref, lds_ref, _) in enumerate(ref):
        assert abs(est[c] - e_ref) <= 1e-9 * abs(e_ref), (c, est[c], e_ref)
        assert rel_err(cols[:, c][col_ref != 0], col_ref[col_ref != 0]) <= 1e-8 and np.array_equal(cols[:, c] == 0, col_ref == 0)
        assert abs(lds[c] - lds_ref) <= 1e-8 * abs(lds_ref)
    # a dry run leaves every cluster exactly as it was
    for g, b in zip(models, before):
        for now, was in zip((g.f_star_sm[-1], g.cov_f_sm[-1], g.A[-1], g.Sigma[-1], g.internal_params.scale), b):
            assert torch.equal(now, was)
    # commit: the cluster that takes the beat, against include_weighted_sample + bayesian_new_params on a copy
    for c in (0, 2):
        g = models[c]
        twin = sw.gpmodel_deepcopy(g)
        twin.include_weighted_sample(t_new, x, x, y, 1.0)
        twin.bayesian_new_params(1.0)
        pool.commit(g, t_new, x, y)
        pool.finish_commit()
        assert g.N == twin.N and g.indexes == twin.indexes and len(g.f_star) == len(twin.f_star)
        for name in ("f_star", "f_star_sm", "cov_f", "cov_f_sm", "A", "Gamma", "C", "Sigma"):
            a, b = torch.stack(list(getattr(g, name))).cpu().numpy(), torch.stack(list(getattr(twin, name))).cpu().numpy()
            assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), (name, c)
        assert g.internal_params.n0 == twin.internal_params.n0
        for a, b in ((g.internal_params.scale, twin.internal_params.scale), (g.observation_params.scale, twin.observation_params.scale),
                     (g.observation_params.m_mean, twin.observation_params.m_mean)):
            assert relclose(a.cpu().numpy(), b.cpu().numpy(), 1e-9)
        la, lb = g.compute_q_lat_all(hist).cpu().numpy(), twin.compute_q_lat_all(hist).cpu().numpy()
        assert rel_err(la[lb != 0], lb[lb != 0]) <= 1e-8
        assert abs(g.lds_param_likelihood_value() - twin.lds_param_likelihood_value()) <= 1e-8 * abs(twin.lds_param_likelihood_value())
    # and a second round of candidates on the grown chains (rows were appended, one slot was committed twice over its lists)
    y2 = (y * 0.97 + 0.3).contiguous()
    ref2 = [_one_by_one(sw, g, x, y2, t_new + 1, T_all + 1) for g in models]
    hist2 = torch.empty((T_all + 1, 0))
    cols1 = torch.stack([g.compute_q_lat_all(hist2) for g in models], dim=1).contiguous()
    pool.begin_beat(y2[:, 0])
    est2, cols2, lds2 = pool.candidates(t_new + 1, cols1, [g.indexes for g in models])
    for c, (e_ref, col_ref, lds_ref, _) in enumerate(ref2):
        assert abs(float(est2[c]) - e_ref) <= 1e-9 * abs(e_ref)
        got = cols2[:, c].cpu().numpy()
        assert rel_err(got[col_ref != 0], col_ref[col_ref != 0]) <= 1e-8
        assert abs(lds2[c] - lds_ref) <= 1e-8 * abs(lds_ref)
