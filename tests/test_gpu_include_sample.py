"""SURVEY.md 8b Face 1 / BASELINE configs[4]: GPI_HDP.include_sample - the online variational step - on the HIP kernels, driven as
hdpgpc/tests/test_online.py:41-83 drives the reference, against the traces of the reference's own runs
(tests/golden/include_sample_*.npz, make_golden.py online90 / "online256 trace"): after EVERY beat the chosen cluster, the number
of clusters, the hard assignments of the whole history and the cluster sizes are identical, the score matrix within 1e-8."""
import time

import pytest
import torch

from conftest import golden
from online_trace import compare_online, run_online

pytestmark = pytest.mark.gpu


def test_include_sample_r102_t90():
    g = golden("include_sample_r102_n40.npz")
    torch.cuda.synchronize()
    t0 = time.time()
    _, tr = run_online(g)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_online(g, tr, 1e-8)
    print(f"include_sample, 40 beats T=90: {wall:.2f} s (reference {float(g['secs'].sum()):.1f} s), worst {worst:.2e}")


def test_include_sample_r102_t256():
    """configs[4]: beats resampled to T = 256 (linear interpolation), same loop."""
    g = golden("include_sample_r102_t256_n24.npz")
    torch.cuda.synchronize()
    t0 = time.time()
    _, tr = run_online(g)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_online(g, tr, 1e-8)
    print(f"include_sample, 24 beats T=256: {wall:.2f} s (reference {float(g['secs'].sum()):.1f} s), worst {worst:.2e}")


def test_include_sample_two_records_t256():
    """configs[4] as worded: CONCATENATED records - 16 beats of record 100 followed by 16 of record 102, resampled to T = 256
    (make_golden.py online256x2).  Unlike the record-102 fixture, whose every beat opens a cluster, this one also walks the
    committed-member path of the persistent chains (two beats join an existing cluster; 30 clusters at the end)."""
    g = golden("include_sample_r100_r102_t256_n32.npz")
    torch.cuda.synchronize()
    t0 = time.time()
    _, tr = run_online(g)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_online(g, tr, 1e-8)
    assert sum(1 for i in range(1, 32) if int(g["M"][i]) == int(g["M"][i - 1])) >= 2       # beats that join an existing cluster
    print(f"include_sample, 16 + 16 beats of records 100 / 102 at T=256: {wall:.2f} s (reference {float(g['secs'].sum()):.1f} s), worst {worst:.2e}")


def test_online_rank1_factor_tracking_t256():
    """BASELINE configs[4] names the rank-1 Cholesky update kernel as part of the online path at T = 256.  Its consumer: with
    annealing off Sigma_i is an exact multiple of the observation MNIW's scale, whose factor then follows the recursion
    scale' = ((n0 - 2) scale + e e^T) / (n0 - 1) by one hgp_chol_rank1_f64 per absorbed beat, and the beat's score under a
    cluster's last state is |L^-1 d|^2 / c - no factorisation.  Stated tolerance: the reference's 1e-8 mean|diag|
    jitter of _chol_spd is not applied on this path, so scores agree with the refactoring path to 1e-6 relative, not 1e-9;
    the clustering decisions must be identical."""
    import numpy as np
    import hdpgpc.GPI_HDP as hdpgp
    g = golden("include_sample_r102_t256_n24.npz")
    std, std_dif, bs0, bs1, bg0, bg1 = (float(v) for v in g["estimators"])
    data = np.asarray(g["y"], dtype=np.float64)[:, :, None]
    T = data.shape[1]
    xb = np.arange(float(T))[:, None]

    def run(rank1):
        sw = hdpgp.GPI_HDP(xb, x_basis_warp=xb[::2], n_outputs=1, ini_lengthscale=3.0, bound_lengthscale=(1.0, 20.0), ini_gamma=std_dif,
                           ini_sigma=std, ini_outputscale=300.0, noise_warp=std * 0.1, bound_sigma=(bs0, bs1), bound_gamma=(bg0, bg1),
                           bound_noise_warp=(std * 0.01, std * 0.02), annealing=False, verbose=False, max_models=100,
                           bayesian_params=True, free_deg_MNIV=20)
        sw.fixed_theta = tuple(float(v) for v in g["theta_inject"])
        sw.rank1_scoring = rank1
        used = 0
        for i in range(data.shape[0]):
            sw.include_sample(xb, data[i], with_warp=False)
            used += sum(1 for m in sw.gpmodels[0] if m.rank1_scale_factor() is not None)
        return sw, used

    a, used_a = run(False)
    b, used_b = run(True)
    assert used_a == 0 and used_b > 0                        # the tracked factors were live on the second run
    assert [int(r[-1]) for r in a.resp_assigned] == [int(r[-1]) for r in b.resp_assigned] and a.M == b.M
    qa, qb = a.q[-1].cpu().numpy(), b.q[-1].cpu().numpy()
    fin = np.isfinite(qa)
    assert np.array_equal(fin, np.isfinite(qb))
    err = float(np.max(np.abs(qa[fin] - qb[fin]) / np.abs(qa[fin])))
    print(f"rank-1 tracked scoring vs refactoring, T=256, 24 beats: max relative difference {err:.2e}")
    assert err <= 1e-6
    # and the factor itself against a full factorisation of the scale it claims to factor
    for m in b.gpmodels[0]:
        fac = m.rank1_scale_factor()
        if fac is not None:
            L = fac[0].cpu().numpy()
            S = m.observation_params.scale.cpu().numpy()
            assert np.max(np.abs(L @ L.T - S)) <= 1e-10 * np.max(np.abs(S))
