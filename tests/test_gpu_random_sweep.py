"""Randomised differential sweep of the pair path against the oracle: basis / segment lengths on both sides of every tile
and kernel boundary (T, T* in 5..256: wave kernels up to 128, cooperative kernels above; T* != T), mixed length-scales
(explicit-operator and solve-based clusters in one batch), iso-diagonal states, `first` inflation, per-segment selection,
grids that are shifted, reversed or coarser than the basis.  Seeds are fixed: the sweep is deterministic."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import hdpgpc_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hdpgpc_amd import ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


CASES = [(5, 5), (16, 16), (17, 31), (33, 20), (48, 64), (65, 65), (90, 45), (96, 97), (112, 128), (128, 128), (129, 129),
         (128, 130), (140, 100), (191, 192), (193, 150), (240, 256), (256, 200)]


@pytest.mark.parametrize("T,Ts", CASES)
def test_pairs_random_sweep(T, Ts):
    rng = np.random.default_rng(1000 * T + Ts)
    N, K = 4, 4
    b = orc.synthetic_batch(N, K, T, seed=T + 7 * Ts)
    ells = rng.choice([0.8, 1.2, 1.2, 2.5, 3.0], size=K)
    b["theta"][:, 1] = ells
    if rng.random() < 0.5:
        b["Sigma"][rng.integers(K)] = rng.uniform(0.5, 3.0) * np.eye(T)          # iso-diagonal state (GPI.py:497-498)
    if Ts == T:
        x = b["x"]
        y = b["y"]
    else:
        x = np.linspace(0, T - 1, Ts)[None, :] + rng.uniform(-0.25, 0.25, (N, Ts)) * (T - 1) / max(Ts - 1, 1)
        y = np.stack([np.interp(x[n], b["xb"], b["mean"][b["labels"][n]]) for n in range(N)]) + rng.normal(0, 3.0, (N, Ts))
    x = x.copy()
    x[1] = x[1][::-1].copy()                                                     # reversed grid
    x[2] = x[2] + 3.7                                                            # shifted grid
    fn = np.where(rng.random((N, K)) < 0.3, rng.uniform(0.01, 0.3, (N, K)), 0.0)
    plan = ops.PairsPlan(T, Ts, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    assert int(plan.info.abs().max()) == 0
    quad, logdet, info = plan.loglik(dev(x), dev(y), first_noise=dev(fn))
    assert int(info.abs().max()) == 0
    _, q_ref, ld_ref = orc.loglik_pairs(x, y, b["xb"], b["theta"], b["mean"], b["Sigma"], first_noise=fn)
    assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8, (ells, plan.solve_based())
    assert rel_err(logdet.cpu().numpy(), ld_ref) < 1e-8
    sel = rng.integers(0, K, N).astype(np.int32)
    q1, l1, _ = plan.loglik(dev(x), dev(y), first_noise=dev(fn[np.arange(N), sel]), sel=sel)
    assert torch.equal(q1, quad[torch.arange(N), torch.as_tensor(sel, device="cuda").long()])
    assert torch.equal(l1, logdet[torch.arange(N), torch.as_tensor(sel, device="cuda").long()])
