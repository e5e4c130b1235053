"""Shared by the CPU and GPU tests of GPI_HDP.include_sample: drive the mirror as hdpgpc/tests/test_online.py:41-83 drives the
reference and compare, beat by beat, with the trace tests/golden/make_golden.py::gen_include_sample recorded there."""
import numpy as np


def run_online(g, n=None):
    import hdpgpc.GPI_HDP as hdpgp

    std, std_dif, bs0, bs1, bg0, bg1 = (float(v) for v in g["estimators"])
    data = np.asarray(g["y"], dtype=np.float64)[:, :, None]
    n = data.shape[0] if n is None else n
    T = data.shape[1]
    xb = np.arange(float(T))[:, None]
    sw = hdpgp.GPI_HDP(xb, x_basis_warp=xb[::2], n_outputs=1, kernels=None, model_type="dynamic", ini_lengthscale=3.0,
                       bound_lengthscale=(1.0, 20.0), ini_gamma=std_dif, ini_sigma=std, ini_outputscale=300.0, noise_warp=std * 0.1,
                       bound_sigma=(bs0, bs1), bound_gamma=(bg0, bg1), bound_noise_warp=(std * 0.01, std * 0.02), warp_updating=False,
                       method_compute_warp="greedy", verbose=False, hmm_switch=True, max_models=100, mode_warp="rough",
                       bayesian_params=True, inducing_points=False, estimation_limit=None, free_deg_MNIV=20)
    sw.fixed_theta = tuple(float(v) for v in g["theta_inject"])
    tr = []
    for i in range(n):
        sw.include_sample(xb, data[i], with_warp=False)
        tr.append((sw.actual_state, sw.M, sw.resp_assigned[-1].numpy().astype(np.int16), sw.q[-1].cpu().numpy().copy(),
                   np.array([len(m.indexes) for m in sw.gpmodels[0]])))
    return sw, tr


def compare_online(g, tr, tol):
    """Per beat: the chosen cluster, the number of clusters, the assignments of the whole history and the cluster sizes must
    be IDENTICAL to the reference's; the score matrix within tol (entries that are -inf there must be -inf here)."""
    from conftest import _note

    worst = 0.0
    for i, (state, M, labels, q, counts) in enumerate(tr):
        assert state == int(g["state"][i]) and M == int(g["M"][i]), f"beat {i}: cluster {state} of {M}, reference {g['state'][i]} of {g['M'][i]}"
        assert np.array_equal(labels, g[f"b{i}_labels"]), f"beat {i}: assignments of the history differ"
        assert np.array_equal(counts, g[f"b{i}_counts"]), f"beat {i}: cluster sizes {counts} vs {g[f'b{i}_counts']}"
        ref = g[f"b{i}_q"]
        assert q.shape == ref.shape and np.array_equal(np.isinf(q), np.isinf(ref)), f"beat {i}: score matrix layout"
        fin = np.isfinite(ref)
        if fin.any():
            worst = max(worst, float(np.max(np.abs(q[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300))))
    _note(worst)
    assert worst <= tol, f"worst relative error of the score matrices {worst:.3e} > {tol:.1e}"
    return worst
