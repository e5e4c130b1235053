"""Shared by the CPU and GPU tests of GPI_HDP.include_batch: run the mirror the way hdpgpc/tests/test_offline.py:32-79 drives
the reference, record the same trace tests/golden/make_golden.py::gen_include_batch records there, and compare the two."""
import numpy as np
import torch


def build_model(g, y):
    import hdpgpc.GPI_HDP as hdpgp

    std, std_dif, bs0, bs1, bg0, bg1 = (float(v) for v in g["estimators"])
    data = np.asarray(y, dtype=np.float64)
    data = data[:, :, None] if data.ndim == 2 else data          # [N, T] one lead, [N, T, D] several
    N, T, D = data.shape
    xb = np.arange(float(T))[:, None]
    x_trains = np.array([xb] * N)
    sw = hdpgp.GPI_HDP(xb, x_basis_warp=xb[::2], n_outputs=D, kernels=None, model_type="dynamic", ini_lengthscale=3.0,
                       bound_lengthscale=(1.0, 20.0), ini_gamma=std_dif, ini_sigma=std, ini_outputscale=300.0, noise_warp=std * 0.1,
                       bound_sigma=(bs0, bs1), bound_gamma=(bg0, bg1), bound_noise_warp=(std * 0.01, std * 0.02), warp_updating=False,
                       method_compute_warp="greedy", verbose=False, hmm_switch=True, max_models=100, mode_warp="rough",
                       bayesian_params=True, inducing_points=False, reestimate_initial_params=True,
                       n_explore_steps=int(g["n_explore"]), free_deg_MNIV=5)
    sw.fixed_theta = tuple(float(v) for v in g["theta_inject"])
    return sw, x_trains, data


def run_model(g, y, it_limit=None):
    """include_batch with the keyword the reference's drivers use (hdpgpc/tests/test_offline.py:79)."""
    sw, x_trains, data = build_model(g, y)
    sw.include_batch(x_trains, data, with_warp=False, it_limit=it_limit)
    return sw


def traced(sw, run):
    """run() with the loop's entry points wrapped like tests/golden/make_golden.py::_trace_loop wraps the reference's."""
    tr = {"order": [], "elbo": [], "qall": [], "fpw": [], "em": []}
    o_elbo, o_qall, o_vltb = sw.compute_q_elbo, sw.estimate_q_all, sw.variational_local_terms_batch
    lab = lambda r: torch.argmax(r, dim=1).numpy().astype(np.int16)   # noqa: E731

    def w_elbo(resp, respPair, q, q_lat, gpmodels, M, *a, **k):
        out = o_elbo(resp, respPair, q, q_lat, gpmodels, M, *a, **k)
        tr["order"].append(0)
        tr["elbo"].append((torch.sum(resp, dim=0).numpy(), float(out[0]), float(out[1]), float(bool(k.get("post", False)))))
        return out

    def w_qall(M, *a, **k):
        out = o_qall(M, *a, **k)
        tr["order"].append(1)
        tr["qall"].append(lab(out[0]))
        return out

    def w_fpw(resp, out):
        # GPI_HDP._note_full_pass: one call per full pass, in the order the reference's loop makes them (the passes themselves run
        # in batches of independent chains, hdpgpc_amd/chain_batch.py)
        tr["order"].append(2)
        mem = torch.nonzero(torch.as_tensor(resp) > 0.99).reshape(-1).numpy()
        tr["fpw"].append((float(len(mem)), float(mem[0]) if len(mem) else -1.0, float(mem[-1]) if len(mem) else -1.0,
                          float(torch.sum(out[0])) if out is not None else np.nan, float(torch.sum(out[1])) if out is not None else np.nan))

    def w_vltb(*a, **k):
        out = o_vltb(*a, **k)
        tr["order"].append(3)
        tr["em"].append((lab(out[0]), out[2].cpu().numpy().copy(), out[3].cpu().numpy().copy(), bool(out[5])))
        return out

    sw.compute_q_elbo, sw.estimate_q_all, sw.variational_local_terms_batch, sw._note_full_pass = w_elbo, w_qall, w_vltb, w_fpw
    try:
        run()
    finally:
        sw.compute_q_elbo, sw.estimate_q_all, sw.variational_local_terms_batch = o_elbo, o_qall, o_vltb
        del sw._note_full_pass
    return tr


def run_traced(g, y, it_limit=None, warp=False):
    sw, x_trains, data = build_model(g, y)
    return sw, traced(sw, lambda: sw.include_batch(x_trains, data, with_warp=warp, it_limit=it_limit))   # the drivers' keyword


def run_cluster_learning(g):
    """hdpgpc/tests/test_offline_multi_output_load.py:81-85 on the fixture's beats: reload_model_from_labels on the first n0,
    cluster_new_batch(the rest, learning=True)."""
    import hdpgpc.GPI_HDP as hdpgp

    std, std_dif, bs0, bs1, bg0, bg1 = (float(v) for v in g["estimators"])
    data = np.asarray(g["y"], dtype=np.float64)
    N, T, D = data.shape
    n0 = int(g["n0"])
    xb = np.arange(float(T))[:, None]
    x_trains = np.array([xb] * N)
    sw = hdpgp.GPI_HDP(xb, x_basis_warp=xb[::2], n_outputs=D, kernels=None, model_type="dynamic", ini_lengthscale=3.0,
                       bound_lengthscale=(1.0, 20.0), ini_gamma=std_dif, ini_sigma=std, ini_outputscale=300.0, noise_warp=std * 0.1,
                       bound_sigma=(bs0, bs1), bound_gamma=(bg0, bg1), bound_noise_warp=(std * 0.01, std * 0.02), warp_updating=False,
                       method_compute_warp="greedy", verbose=False, hmm_switch=True, max_models=100, mode_warp="rough",
                       bayesian_params=True, inducing_points=False, reestimate_initial_params=True,
                       n_explore_steps=int(g["n_explore"]), free_deg_MNIV=5)
    sw.fixed_theta = tuple(float(v) for v in g["theta_inject"])
    sw.reload_model_from_labels(x_trains[:n0], data[:n0], g["labels"], int(g["M0"]))
    tr = traced(sw, lambda: sw.cluster_new_batch(x_trains[n0:], data[n0:], learning=True))
    return sw, tr


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300)) if a.size else 0.0


def compare_trace(g, sw, tr, q_tol=1e-8, n_em=None):
    """Every decision of the loop (the order of its calls, every assignment it produced) must be IDENTICAL to the reference's;
    every number it computed (scores, bound terms) within q_tol of the reference's."""
    from conftest import _note

    n_ref = len(g["order"]) if n_em is None else int(np.nonzero(np.cumsum(g["order"] == 3) == n_em)[0][0]) + 1
    order = np.array(tr["order"], dtype=np.int8)
    k = min(len(order), n_ref)
    first_bad = np.nonzero(order[:k] != g["order"][:k])[0]
    assert first_bad.size == 0 and len(order) == n_ref, \
        f"call sequence differs at event {first_bad[:1]} (ours {len(order)} events, reference {n_ref})"
    n_elbo, n_qall, n_fpw, n_emr = (int(np.sum(order == i)) for i in range(4))
    # assignments: exact
    for i in range(n_qall):
        assert np.array_equal(tr["qall"][i], g["qall_labels"][i]), f"estimate_q_all call {i}: assignments differ"
    for i in range(n_emr):
        assert np.array_equal(tr["em"][i][0], g[f"em{i}_labels"]), f"EM iteration {i}: assignments differ"
        assert tr["em"][i][3] == bool(g[f"em{i}_reallocate"])
    worst = 0.0
    for i in range(n_elbo):
        c = g["elbo_counts"][i]
        c = c[c >= 0]
        assert np.array_equal(tr["elbo"][i][0], c), f"bound evaluation {i}: cluster sizes {tr['elbo'][i][0]} vs {c}"
        assert tr["elbo"][i][3] == g["elbo_vals"][i][2]
        worst = max(worst, _rel(tr["elbo"][i][1:3], g["elbo_vals"][i][:2]))
    fp = np.array(tr["fpw"])
    assert np.array_equal(fp[:, :3], g["fpw"][:n_fpw, :3]), "full_pass_weighted: member sets differ"
    has = fp[:, 0] > 0                                   # a pass without members hands back its caller's columns: nothing to compare
    worst = max(worst, _rel(fp[has, 3:], g["fpw"][:n_fpw][has, 3:]))
    for i in range(n_emr):
        worst = max(worst, _rel(tr["em"][i][1], g[f"em{i}_q"]), _rel(tr["em"][i][2], g[f"em{i}_q_lat"]))
    _note(worst)
    assert worst <= q_tol, f"worst relative error {worst:.3e} > {q_tol:.1e}"
    if n_em is None and "q_last" in g:                       # cluster_new_batch(learning=True) fixture
        assert sw.M == int(g["M_final"])
        assert np.array_equal(np.array([[len(m.indexes) for m in lead] for lead in sw.gpmodels]), g["counts_final"])
        assert np.array_equal(sw.resp_assigned[-1].numpy().astype(np.int16), g["resp_last"])
        assert _rel(np.array(sw.train_elbo), g["train_elbo"]) <= q_tol
        assert _rel(sw.q_last.cpu().numpy(), g["q_last"]) <= q_tol and _rel(sw.q_lat_last.cpu().numpy(), g["q_lat_last"]) <= q_tol
    elif n_em is None:
        assert sw.M == int(g["M_final"])
        for lead in sw.gpmodels:
            assert np.array_equal(np.array([len(m.indexes) for m in lead]), g["counts_final"])
        ra = np.stack([r.numpy().astype(np.int16) for r in sw.resp_assigned])
        assert np.array_equal(ra, g["resp_assigned"])
        assert _rel(np.array(sw.train_elbo), g["train_elbo"]) <= q_tol
        assert np.array_equal(sw.f_ind_old.numpy(), g["f_ind_old"])
        for name in ("transTheta", "startTheta", "rho", "omega"):
            assert _rel(np.asarray(getattr(sw, name)), g[name]) <= 1e-6, name
    return worst
