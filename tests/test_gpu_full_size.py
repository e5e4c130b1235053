"""BASELINE.json's full sizes, checked through size-independent properties (the oracle would need hours here):
shard invariance (what the multi-GPU path relies on), permutation equivariance in segments and clusters,
per-segment selection, independence of the log-determinant from y, plus an oracle spot check of a random sample."""
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


@pytest.mark.parametrize("N,K,T,n_spot", [(2048, 8, 128, 16), (4096, 16, 256, 4), (32768, 16, 256, 2)])
def test_full_size_batch_properties(N, K, T, n_spot):
    """configs[1] (2 048 x 8, T = 128), the per-GPU shard of configs[3] (4 096 segments x 16 clusters, T = 256) and the WHOLE
    configs[3] batch (32 768 x 16, T = 256: what bench.py --gpus N shards) on irregular grids."""
    b = orc.synthetic_batch(N, K, T, seed=20260703)
    xb, mean, Sig = dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"])
    x, y = dev(b["x"]), dev(b["y"])
    plan = ops.PairsPlan(T, T, b["theta"]).update(xb, mean, Sig)
    quad, logdet, info = plan.loglik(x, y)
    assert int(info.abs().max()) == 0 and bool(torch.isfinite(quad).all()) and bool(torch.isfinite(logdet).all())
    rng = np.random.default_rng(7)

    # 1. shard invariance: the two halves scored separately are the rows of the full result, bit for bit
    h = N // 2
    qa, la, _ = plan.loglik(x[:h].contiguous(), y[:h].contiguous())
    qb, lb, _ = plan.loglik(x[h:].contiguous(), y[h:].contiguous())
    assert torch.equal(torch.cat((qa, qb)), quad) and torch.equal(torch.cat((la, lb)), logdet)

    # 2. permuting the segments permutes the rows
    perm = torch.as_tensor(rng.permutation(N), device="cuda")
    qp, lp, _ = plan.loglik(x[perm].contiguous(), y[perm].contiguous())
    assert torch.equal(qp, quad[perm]) and torch.equal(lp, logdet[perm])

    # 3. per-segment selection returns the selected column
    sel = torch.as_tensor(rng.integers(0, K, N), dtype=torch.int32, device="cuda")
    qs, ls, _ = plan.loglik(x, y, sel=sel)
    idx = sel.long().unsqueeze(1)
    assert torch.equal(qs, quad.gather(1, idx)[:, 0]) and torch.equal(ls, logdet.gather(1, idx)[:, 0])

    # 4. permuting the clusters permutes the columns (a new plan: per-cluster operators are rebuilt)
    cp = rng.permutation(K)
    plan2 = ops.PairsPlan(T, T, b["theta"][cp]).update(xb, dev(b["mean"][cp]), dev(b["Sigma"][cp]))
    q2, l2, _ = plan2.loglik(x, y)
    cpt = torch.as_tensor(cp, device="cuda")
    assert torch.equal(q2, quad[:, cpt]) and torch.equal(l2, logdet[:, cpt])

    # 5. the covariance (hence its log-determinant) does not depend on y
    _, l3, _ = plan.loglik(x, dev(rng.normal(size=(N, T))))
    assert torch.equal(l3, logdet)

    # 6. oracle spot check on a random sample of segments
    rows = rng.choice(N, n_spot, replace=False)
    _, q_ref, ld_ref = orc.loglik_pairs(b["x"][rows], b["y"][rows], b["xb"], b["theta"], b["mean"], b["Sigma"])
    assert rel_err(quad[rows].cpu().numpy(), q_ref) < 1e-8
    assert rel_err(logdet[rows].cpu().numpy(), ld_ref) < 1e-8
    # hard assignments of the sample: arg-max of the reference's score (no log-determinant) is identical
    assert np.array_equal(np.argmin(quad[rows].cpu().numpy(), axis=1), np.argmin(q_ref, axis=1))


def test_full_size_shared_grid_member_path_properties():
    """The reference's member dataflow at record-100 scale (2 272 segments, one Sigma_i each, T = 90): shard invariance
    and symmetric-fast-path equality, plus an oracle spot check."""
    S, T = 2272, 90
    rng = np.random.default_rng(3)
    Q = rng.normal(size=(64, T, T))
    A = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    Sig = dev(np.tile(A, (S // 64 + 1, 1, 1))[:S])
    Y, mean = dev(rng.normal(size=(S, T))), dev(rng.normal(size=(S, T)))
    sm = torch.arange(S, dtype=torch.int32, device="cuda")
    quad, logdet, info = ops.score_each(Y, mean, Sig, sm, want_logdet=True)
    assert int(info.abs().max()) == 0
    q_sym, l_sym, _ = ops.score_each(Y, mean, Sig, sm, symmetric=True, want_logdet=True)
    assert torch.equal(q_sym, quad) and torch.equal(l_sym, logdet)           # A is symmetric bit for bit
    h = S // 2
    qa, _, _ = ops.score_each(Y[:h].contiguous(), mean[:h].contiguous(), Sig[:h].contiguous(), sm[:h].contiguous())
    assert torch.equal(qa, quad[:h])
    for n in rng.choice(S, 8, replace=False):
        Sn = Sig[n].cpu().numpy()
        d = (Y[n] - mean[n]).cpu().numpy()
        Sn = 0.5 * (Sn + Sn.T)
        Sn = Sn + 1e-8 * max(np.mean(np.abs(np.diag(Sn))), np.finfo(float).eps) * np.eye(T)
        assert abs(float(quad[n]) - d @ np.linalg.solve(Sn, d)) <= 1e-9 * abs(float(quad[n]))
