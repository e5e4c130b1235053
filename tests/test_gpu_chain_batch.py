"""chain_batch.run: many independent member chains advanced side by side (one launch per dependency level for all of them)
must give BIT-identical results to running GPI_model.full_pass_weighted on each, whatever the mix of chain lengths."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _models(n, T, sigma=9.0, gamma=12.0):
    from hdpgpc_amd.GPI import RBFWhiteKernel
    from hdpgpc_amd.GPI_model import GPI_model
    out = []
    for _ in range(n):
        m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), np.arange(float(T))[:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
        cond = m.GPR_dynamic(gamma, sigma)
        m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
        m.fixed_theta = (341.0, 1.2, 4.66)
        m.noise_bounds = (sigma * 1e-5, sigma * 2.0)
        out.append(m)
    return out


@pytest.mark.parametrize("sizes", [[40, 7, 23, 23, 5], [3, 60], [12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12]])
def test_batched_chains_equal_sequential(sizes):
    from hdpgpc_amd import chain_batch
    y = golden("mitbih100_lead0.npz")["y"][:90]
    N, T = y.shape
    x = np.repeat(np.arange(float(T))[None, :, None], N, axis=0)
    rng = np.random.default_rng(len(sizes))
    resps = []
    for k in sizes:
        r = torch.zeros(N)
        r[torch.as_tensor(np.sort(rng.choice(N, k, replace=False)))] = 1.0
        resps.append(r)
    seq = [m.full_pass_weighted(x, y[:, :, None], r) for m, r in zip(_models(len(sizes), T), resps)]
    ms = _models(len(sizes), T)
    bat = chain_batch.run([chain_batch.Job(m, x, y[:, :, None], r) for m, r in zip(ms, resps)])
    ref = _models(len(sizes), T)
    for m, r in zip(ref, resps):
        m.full_pass_weighted(x, y[:, :, None], r)
    for (qa, la), (qb, lb), ma, mb in zip(seq, bat, ref, ms):
        assert torch.equal(qa, qb) and torch.equal(la, lb)
        for name in ("f_star", "f_star_sm", "cov_f_sm", "A", "Gamma", "C", "Sigma"):
            assert torch.equal(torch.stack(list(getattr(ma, name))), torch.stack(list(getattr(mb, name)))), name
        assert ma.indexes == mb.indexes and ma.N == mb.N and float(ma.internal_params.n0) == float(mb.internal_params.n0)


def test_batch_with_an_empty_and_a_tiny_chain():
    """A column without members hands back the previous scores; chains too short for the graphed path take the eager one."""
    from hdpgpc_amd import chain_batch
    y = golden("mitbih100_lead0.npz")["y"][:30]
    N, T = y.shape
    x = np.repeat(np.arange(float(T))[None, :, None], N, axis=0)
    r0, r1, r2 = torch.zeros(N), torch.zeros(N), torch.zeros(N)
    r1[[2, 9]] = 1.0
    r2[torch.arange(10, 30)] = 1.0
    ms = _models(3, T)
    prev = (torch.full((N,), -1.0, dtype=torch.float64, device="cuda"), torch.full((N,), -2.0, dtype=torch.float64, device="cuda"))
    out = chain_batch.run([chain_batch.Job(ms[0], x, y[:, :, None], r0, prev=prev), chain_batch.Job(ms[1], x, y[:, :, None], r1),
                           chain_batch.Job(ms[2], x, y[:, :, None], r2)])
    assert out[0][0] is prev[0] and out[0][1] is prev[1] and ms[0].N == 0
    assert ms[1].indexes == [2, 9] and ms[2].N == 20
    ref = _models(1, T)[0]
    q, ql = ref.full_pass_weighted(x, y[:, :, None], r2)
    assert torch.equal(q, out[2][0]) and torch.equal(ql, out[2][1])
