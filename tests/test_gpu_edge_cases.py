"""Edge cases of the C-ABI / host layer on the GPU: empty batches, tiny and ragged sizes, one cluster,
non-positive-definite inputs (LAPACK-style info -> torch.linalg.LinAlgError), argument validation."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import hdpgpc_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hdpgpc_amd import _ffi, ops
    from hdpgpc_amd.GPI import RBFWhiteKernel
    from hdpgpc_amd.GPI_model import GPI_model


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def test_empty_batches_are_noops():
    T = 24
    b = orc.synthetic_batch(1, 2, T, seed=1)
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(torch.empty((0, T), dtype=torch.float64, device="cuda"),
                                     torch.empty((0, T), dtype=torch.float64, device="cuda"))
    assert quad.shape == (0, 2) and logdet.shape == (0, 2)
    L, info = ops.potrf_batched(torch.empty((0, T, T), dtype=torch.float64, device="cuda"))
    assert L.shape == (0, T, T)
    q, _, _ = ops.score_each(torch.empty((0, T), dtype=torch.float64, device="cuda"), None, dev(np.eye(T)[None]),
                             np.zeros(0, np.int32))
    assert q.numel() == 0


@pytest.mark.parametrize("T", [1, 2, 15, 16, 17, 31, 100, 127])
def test_sizes_that_are_not_multiples_of_the_tile(T):
    rng = np.random.default_rng(T)
    Q = rng.normal(size=(2, T, T))
    A = Q @ Q.transpose(0, 2, 1) + T * np.eye(T)
    L, info, logdet = ops.potrf_batched(dev(A), 0.0, 0.0, want_logdet=True)
    assert int(info.abs().max()) == 0
    for k in range(2):
        ref = np.linalg.cholesky(A[k])
        assert np.allclose(L[k].cpu().numpy(), ref, rtol=1e-11, atol=1e-12)
        assert abs(float(logdet[k]) - 2 * np.log(np.diag(ref)).sum()) <= 1e-10 * max(1.0, T)
    if T >= 2:
        b = orc.synthetic_batch(3, 2, T, seed=T)
        plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
        quad, logdet2, info2 = plan.loglik(dev(b["x"]), dev(b["y"]))
        _, q_ref, ld_ref = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
        assert int(info2.abs().max()) == 0
        assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8 and rel_err(logdet2.cpu().numpy(), ld_ref) < 1e-8


def test_segment_grid_longer_and_shorter_than_the_basis():
    """T* != T: the plan is padded to the larger of the two (e.g. scoring on a resampled grid)."""
    T, K = 40, 2
    b = orc.synthetic_batch(4, K, T, seed=3)
    for Ts in (25, 64):
        rng = np.random.default_rng(Ts)
        x = np.sort(rng.uniform(0, T - 1, size=(4, Ts)), axis=1)
        y = rng.normal(size=(4, Ts)) * 3
        plan = ops.PairsPlan(T, Ts, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
        quad, logdet, info = plan.loglik(dev(x), dev(y))
        _, q_ref, ld_ref = orc.loglik_pairs(x, y, b["xb"], b["theta"], b["mean"], b["Sigma"])
        assert int(info.abs().max()) == 0
        assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8 and rel_err(logdet.cpu().numpy(), ld_ref) < 1e-8


def test_single_cluster_single_segment():
    b = orc.synthetic_batch(1, 1, 20, seed=4)
    plan = ops.PairsPlan(20, 20, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, _, _ = plan.loglik(dev(b["x"]), dev(b["y"]))
    _, q_ref, _ = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8


def test_not_positive_definite_raises_like_torch():
    T = 20
    kern = RBFWhiteKernel(5.0, 1.2, 0.5)
    m = GPI_model(kern, np.arange(float(T))[:, None])
    bad = np.eye(T)
    bad[7, 7] = -3.0
    with pytest.raises(torch.linalg.LinAlgError):
        m._chol_spd(bad)
    with pytest.raises(torch.linalg.LinAlgError):
        m._gaussian_score_shared_cov(np.zeros((2, T)), np.zeros(T), bad)
    # the info value is LAPACK's: index (1-based) of the first non-positive pivot
    _, info = ops.potrf_batched(dev(bad[None]), 0.0, 0.0)
    assert int(info[0]) == 8
    # NaN input is reported, not silently propagated as "ok"
    bad2 = np.eye(T)
    bad2[3, 3] = np.nan
    _, info = ops.potrf_batched(dev(bad2[None]), 0.0, 0.0)
    assert int(info[0]) > 0


def test_argument_validation_returns_status_codes():
    lib = _ffi.lib
    z = ctypes.c_void_p(0)
    assert lib.hgp_potrf_batched_f64(z, 8, 1, 0.0, 0.0, z, z, z, z) == -1
    assert lib.hgp_potrf_batched_f64(ctypes.c_void_p(16), 4096, 1, 0.0, 0.0, z, z, z, z) == -2     # T beyond every kernel
    assert lib.hgp_chol_rank1_f64(ctypes.c_void_p(16), ctypes.c_void_p(16), z, z, 300, 1, z, z) == -2
    with pytest.raises(NotImplementedError):
        ops.PairsPlan(300, 300, np.array([[1.0, 1.0, 0.1]]))
    with pytest.raises(ValueError):
        ops.PairsPlan(16, 16, np.array([[1.0, -1.0, 0.1]]))                                       # negative length-scale
    with pytest.raises(TypeError):
        ops.gram_rbf(torch.zeros(4), None, 1.0, 1.0)                                               # CPU tensor: no fallback


def test_many_right_hand_sides_one_factor():
    """a4 with B >> 64: the group is split into work items that each refactor once (GPI_model.py:531 with B = N)."""
    rng = np.random.default_rng(11)
    T, B = 90, 1000
    Q = rng.normal(size=(T, T))
    cov = Q @ Q.T / T + np.eye(T)
    mean = rng.normal(size=T)
    Y = rng.normal(size=(B, T)) * 2
    items = ops.build_items([0], [0.0], [B])
    quad, _, info = ops.score_groups(dev(Y), dev(mean), dev(cov), *items)
    assert int(info.abs().max()) == 0
    ref = -2.0 * (orc.gaussian_score_shared_cov(Y, mean, cov) + 0.5 * T * orc.LOG2PI)
    assert rel_err(quad.cpu().numpy(), ref) < 1e-10


@pytest.mark.parametrize("T", [129, 177, 255])
def test_large_sizes_that_are_not_multiples_of_the_tile(T):
    """128 < T <= 256 goes through the cooperative kernels (one workgroup per matrix / per pair), padded to 192 / 256."""
    b = orc.synthetic_batch(3, 2, T, seed=T)
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(b["x"]), dev(b["y"]))
    _, q_ref, ld_ref = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    assert int(info.abs().max()) == 0
    assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8 and rel_err(logdet.cpu().numpy(), ld_ref) < 1e-8


def test_large_T_segment_grid_longer_and_shorter_than_the_basis():
    """T* != T above 128 points: basis of 150 scored on 200-point segments (plan padded to 256) and on 100-point ones;
    basis of 100 scored on 140-point segments (the segment side alone forces the cooperative kernel)."""
    for T, Ts in ((150, 200), (150, 100), (100, 140)):
        b = orc.synthetic_batch(3, 2, T, seed=T + Ts)
        rng = np.random.default_rng(Ts)
        x = np.sort(rng.uniform(0, T - 1, size=(3, Ts)), axis=1)
        y = rng.normal(size=(3, Ts)) * 3
        plan = ops.PairsPlan(T, Ts, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
        quad, logdet, info = plan.loglik(dev(x), dev(y))
        _, q_ref, ld_ref = orc.loglik_pairs(x, y, b["xb"], b["theta"], b["mean"], b["Sigma"])
        assert int(info.abs().max()) == 0
        assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8 and rel_err(logdet.cpu().numpy(), ld_ref) < 1e-8


def test_large_T_mixed_lengthscales_iso_first_and_selection():
    """Cooperative kernel with everything at once: two length-scale groups (one launch each), an iso-diagonal cluster
    (GPI.py:497 short cut), per-pair `first` inflation and per-segment cluster selection."""
    T, N, K = 144, 6, 4
    b = orc.synthetic_batch(N, K, T, seed=12)
    b["theta"][1, 1] = 0.9                                 # second length-scale group
    b["theta"][3, 1] = 0.9
    b["Sigma"][2] = 1.7 * np.eye(T)                        # iso branch
    rng = np.random.default_rng(1)
    fn = rng.uniform(0.0, 0.05, size=(N, K)) * (rng.uniform(size=(N, K)) < 0.5)
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(b["x"]), dev(b["y"]), first_noise=dev(fn))
    _, q_ref, ld_ref = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"], first_noise=fn)
    assert int(info.abs().max()) == 0
    assert rel_err(quad.cpu().numpy(), q_ref) < 1e-8 and rel_err(logdet.cpu().numpy(), ld_ref) < 1e-8
    sel = np.array([0, 1, 2, 3, 1, 2])
    q1, l1, i1 = plan.loglik(dev(b["x"]), dev(b["y"]), first_noise=dev(fn[np.arange(N), sel]),
                             sel=torch.as_tensor(sel, dtype=torch.int32, device="cuda"))
    assert int(i1.abs().max()) == 0
    assert rel_err(q1.cpu().numpy(), q_ref[np.arange(N), sel]) < 1e-8
    assert rel_err(l1.cpu().numpy(), ld_ref[np.arange(N), sel]) < 1e-8


def test_large_T_not_positive_definite_is_reported_per_pair():
    """A cluster whose Sigma makes cov_f indefinite: info > 0 for its pairs only, the other cluster's results are exact."""
    T, N = 160, 3
    b = orc.synthetic_batch(N, 2, T, seed=5)
    Sig = b["Sigma"].copy()
    Sig[1] = -50.0 * np.eye(T) + 0.1 * np.ones((T, T))      # not iso (off-diagonal), strongly negative
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(Sig))
    quad, logdet, info = plan.loglik(dev(b["x"]), dev(b["y"]))
    info = info.cpu().numpy()
    assert (info[:, 0] == 0).all() and (info[:, 1] > 0).all()
    _, q_ref, _ = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"][:1], b["mean"][:1], b["Sigma"][:1])
    assert rel_err(quad[:, 0].cpu().numpy(), q_ref[:, 0]) < 1e-8


@pytest.mark.parametrize("N,K", [(1, 1), (1, 5), (2, 1), (17, 64), (0, 4)])
def test_hmm_messages_edge_sizes(N, K):
    """hgp_hmm_messages_f64: single step, single state, the 64-state limit, the empty batch; 65 states is refused (-2)."""
    rng = np.random.default_rng(N * 100 + K)
    q = rng.normal(size=(N, K)) * 5 - 20
    lt = np.log(rng.dirichlet(np.ones(K), size=K)) if K > 1 else np.zeros((1, 1))
    lp = np.log(rng.dirichlet(np.ones(K))) if K > 1 else np.zeros(1)
    f, m, b, c = ops.hmm_messages(dev(q), dev(lp), dev(lt))
    if N == 0:
        assert f.shape == (0, K)
        return
    fr, mr = orc.hmm_forward(q, lp, lt)
    with np.errstate(divide="ignore", invalid="ignore"):
        br = orc.hmm_backward(q, lt)
    assert np.allclose(f.cpu().numpy(), fr, rtol=1e-10, atol=1e-300) and np.allclose(m.cpu().numpy(), mr, rtol=1e-10)
    assert np.allclose(b.cpu().numpy(), br, rtol=1e-10, atol=1e-300, equal_nan=True)
    z = ctypes.c_void_p(16)
    assert _ffi.lib.hgp_hmm_messages_f64(z, z, z, 3, 65, z, z, z, None, None) == -2


def test_pair_plans_are_cached_in_a_bounded_lru():
    """plans used to be cached per model (up to 8 each, 37-410 MiB apiece).  Scoring many (segment length, cluster
    count) shapes through several models must keep the device memory of the cached plans under ops.PLAN_CACHE_BYTES."""
    import torch
    from hdpgpc_amd import ops
    from hdpgpc_amd.GPI import RBFWhiteKernel
    from hdpgpc_amd.GPI_model import GPI_model
    T = 48
    models = []
    for i in range(3):
        m = GPI_model(RBFWhiteKernel(300.0, 1.2 + 0.1 * i, 0.5), np.arange(float(T))[:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
        cond = m.GPR_dynamic(2.0, 1.0)
        m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
        models.append(m)
    rng = np.random.default_rng(0)
    old = ops.PLAN_CACHE_BYTES
    ops.PLAN_CACHE_BYTES = 64 << 20
    try:
        peak = 0
        for Ts in range(20, 48, 2):
            x = np.sort(rng.uniform(0, T - 1, Ts))[:, None]
            y = rng.normal(size=(Ts, 1))
            for m in models:
                assert np.isfinite(float(m.log_sq_error(x, y, i=-1)))
            held = sum(p._buf.numel() for p in ops._PLANS.values())
            peak = max(peak, held)
            assert held <= max(ops.PLAN_CACHE_BYTES, max(p._buf.numel() for p in ops._PLANS.values()))
        assert len(ops._PLANS) < 3 * 14 and peak > 0
    finally:
        ops.PLAN_CACHE_BYTES = old


def test_assignment_tail_treats_nan_like_torch():
    """torch.max / torch.argmax of the reference's LogLik / _safe_exp treat NaN as the maximum: a NaN score in a
    non-first column must poison the row maximum and win the arg-max, as it does there."""
    import torch
    from hdpgpc_amd import ops
    q = torch.tensor([[-3.0, float("nan"), -1.0], [-2.0, -5.0, -4.0], [float("nan"), 1.0, 2.0]], dtype=torch.float64, device="cuda")
    out, rowmax = ops.loglik_rows(q.contiguous())
    ref = torch.max(q.cpu(), dim=1)[0]
    assert torch.equal(torch.isnan(rowmax.cpu()), torch.isnan(ref)) and float(rowmax[1]) == -2.0
    f = torch.tensor([[0.2, float("nan"), 0.5], [0.1, 0.7, 0.2], [0.3, 0.3, float("nan")]], dtype=torch.float64, device="cuda")
    b = torch.ones_like(f)
    labels = ops.assign(f.contiguous(), b.contiguous()).cpu()
    assert torch.equal(labels, torch.argmax(torch.log(f.cpu() * b.cpu()), dim=1))


@pytest.mark.parametrize("T, Ts", [(128, 128), (90, 90), (120, 128), (128, 100), (96, 96)])
def test_static_band_sweeps_equal_the_mask_driven_ones_bit_for_bit(T, Ts, monkeypatch):
    """k_pairs takes straight-line sweeps when the workgroup's E is block-tridiagonal (band_sweeps, hgp_pairs.hip) and the
    mask-driven ones otherwise: same operations in the same order per accumulator, so the two covariances agree to the last bit -
    on jittered unit grids (band everywhere), and on a batch where some segments are NOT banded (stretched / reversed grids take
    the generic path inside the same launch).  HGP_PAIRS_GENERIC=1 forces the generic sweeps."""
    N, K = 96, 5
    b = orc.synthetic_batch(N, K, T, seed=7 + T + Ts)
    rng = np.random.default_rng(Ts)
    x = np.arange(Ts, dtype=np.float64)[None, :] * (T - 1.0) / max(Ts - 1.0, 1.0) + rng.uniform(-0.3, 0.3, size=(N, Ts))
    x[5] = x[5][::-1].copy()                       # reversed grid: anti-diagonal E
    x[11] = 0.5 * x[11]                            # compressed grid: wider band
    y = rng.normal(0.0, 30.0, size=(N, Ts))
    plan = ops.PairsPlan(T, Ts, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    fn = dev(rng.uniform(0.0, 2.0, size=(N, K)))
    monkeypatch.delenv("HGP_PAIRS_GENERIC", raising=False)
    q1, l1, i1 = plan.loglik(dev(x), dev(y), first_noise=fn)
    monkeypatch.setenv("HGP_PAIRS_GENERIC", "1")
    q0, l0, i0 = plan.loglik(dev(x), dev(y), first_noise=fn)
    monkeypatch.delenv("HGP_PAIRS_GENERIC", raising=False)
    assert int(i0.abs().max()) == 0 and int(i1.abs().max()) == 0
    # identical covariances, hence identical factors and log-determinants; the quadratic form only to rounding since the band
    # kernel sums the right-hand side's updates over the four 16-lane rows once per block instead of once per tile (NB = 8)
    assert torch.equal(l0, l1)
    assert float(((q0 - q1).abs() / q0.abs()).max()) <= 1e-12
    # and both against the oracle on a few pairs
    for n in (0, 5, 11, N - 1):
        for k in (0, K - 1):
            _, qq, ll = orc.log_sq_error_state(x[n], y[n], b["xb"], b["mean"][k], b["Sigma"][k], tuple(b["theta"][k]), float(fn[n, k]))
            assert abs(float(q1[n, k]) - qq) <= 1e-8 * abs(qq) and abs(float(l1[n, k]) - ll) <= 1e-8 * abs(ll)


def test_score_output_written_by_the_kernels():
    """hgp_pairs_plan_set_score_output: out_quad receives -0.5 quad - 0.5 Ts log(2 pi) (GPI_model.py:285) from the pair kernels
    themselves - fast, cooperative (T > 128) and solve-based clusters, with and without a per-segment selection; the uncleared
    output buffers are written everywhere."""
    import math
    for (N, K, T, ells) in [(96, 4, 90, (1.2, 1.2, 3.0, 1.2)), (40, 3, 144, (1.2, 3.0, 1.2))]:
        b = orc.synthetic_batch(N, K, T, seed=11)
        theta = b["theta"].copy()
        theta[:, 1] = ells
        plan = ops.PairsPlan(T, T, theta).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
        x, y = dev(b["x"]), dev(b["y"])
        quad, _, info = plan.loglik(x, y, want_logdet=False)
        sc, _, info2 = plan.loglik(x, y, want_logdet=False, score=True)
        ref = -0.5 * quad - 0.5 * T * math.log(2.0 * math.pi)
        assert int(info.abs().max()) == 0 and torch.equal(info, info2)
        assert torch.allclose(sc, ref, rtol=1e-15, atol=0.0)
        s2, _ = plan.score(x, y)
        assert torch.equal(s2, sc)
        sel = torch.arange(N, device="cuda", dtype=torch.int32) % K
        q1, _, _ = plan.loglik(x, y, want_logdet=False, sel=sel)
        s1, _, _ = plan.loglik(x, y, want_logdet=False, sel=sel, score=True)
        assert torch.allclose(s1, -0.5 * q1 - 0.5 * T * math.log(2.0 * math.pi), rtol=1e-15, atol=0.0)
        assert torch.equal(q1, quad[torch.arange(N, device="cuda"), sel.long()])


def test_more_segments_than_the_fall_back_list_holds():
    """hgp_loglik_pairs_f64 launches the band / generic kernel pair in chunks of 65 536 segments (the capacity of the plan's
    fall-back list): a batch beyond that - with non-banded segments on both sides of the chunk boundary, a per-pair
    first_noise and a per-segment selection - must equal its pieces scored separately, bit for bit."""
    N, K, T = 70000, 2, 90
    b = orc.synthetic_batch(256, K, T, seed=5)
    rng = np.random.default_rng(5)
    x = np.tile(b["x"], (N // 256 + 1, 1))[:N].copy()
    y = np.tile(b["y"], (N // 256 + 1, 1))[:N].copy()
    for n in (3, 65535, 65536, 65537, 69999):
        x[n] = x[n][::-1].copy()                   # reversed grid: not block-tridiagonal -> generic kernel through the list
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    fn = dev(rng.uniform(0.0, 1.0, size=(N, K)))
    X, Y = dev(x), dev(y)
    q, l, i = plan.loglik(X, Y, first_noise=fn)
    assert int(i.abs().max()) == 0
    h = 40000
    qa, la, _ = plan.loglik(X[:h].contiguous(), Y[:h].contiguous(), first_noise=fn[:h].contiguous())
    qb, lb, _ = plan.loglik(X[h:].contiguous(), Y[h:].contiguous(), first_noise=fn[h:].contiguous())
    assert torch.equal(torch.cat((qa, qb)), q) and torch.equal(torch.cat((la, lb)), l)
    sel = torch.as_tensor(rng.integers(0, K, N), dtype=torch.int32, device="cuda")
    qs, ls, _ = plan.loglik(X, Y, first_noise=fn.gather(1, sel.long().unsqueeze(1))[:, 0].contiguous(), sel=sel)
    assert torch.equal(qs, q.gather(1, sel.long().unsqueeze(1))[:, 0]) and torch.equal(ls, l.gather(1, sel.long().unsqueeze(1))[:, 0])
