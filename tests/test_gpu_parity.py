"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the reference's golden vectors.

Tolerances (fp64): wave kernels reproduce the reference's operation order up to summation order
(rtol 1e-10); the per-pair path uses the algebraically equivalent per-cluster operator
M = c^2 (K~^-1 Sigma K~^-1 - K~^-1) instead of a triangular solve per pair, which moves results at the
1e-11 level at T = 128 (measured against the oracle) - asserted at rtol 1e-8.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from oracle import hdpgpc_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hdpgpc_amd import _ffi, ops

DEV = "cuda"
RT_WAVE = 1e-10
RT_PAIR = 1e-8


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)


def test_mfma_lane_maps():
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, size=(16, 16)).astype(np.float64)      # asymmetric, exact in fp64
    B = rng.integers(-8, 9, size=(16, 16)).astype(np.float64)
    dA, dB = dev(A), dev(B)
    dC = torch.zeros(16, 16, dtype=torch.float64, device=DEV)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = _ffi.lib.hgp_debug_mfma_f64(ctypes.c_void_p(dA.data_ptr()), ctypes.c_void_p(dB.data_ptr()),
                                     ctypes.c_void_p(dC.data_ptr()), s)
    assert rc == 0
    assert np.array_equal(dC.cpu().numpy(), A @ B)


def test_exp_neg4_against_libm():
    """The per-pair kernels build E and K** with their own exp(-h) (tile_f64.hpp, exp_neg4): held to the platform's exp()."""
    rng = np.random.default_rng(1)
    h = np.concatenate([rng.uniform(0.0, 60.0, 40000), rng.uniform(0.0, 1e-3, 2000), rng.uniform(60.0, 700.0, 2000),
                        np.array([0.0, 0.6931471805599453, 55.45177444479562, 82.9, 745.0, 799.0, 801.0, 1e300])])
    dh = dev(h)
    out = torch.empty_like(dh)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _ffi.lib.hgp_debug_exp_neg_f64(ctypes.c_void_p(dh.data_ptr()), h.size, ctypes.c_void_p(out.data_ptr()), s) == 0
    got, ref = out.cpu().numpy(), np.exp(-h)
    normal = ref > 1e-300
    rel = np.abs(got[normal] - ref[normal]) / ref[normal]
    assert rel.max() <= 4e-16, rel.max()
    assert np.all(got[~normal] <= 1e-300) and np.all(got[~normal] >= 0.0)
    assert got[h.size - 8] == 1.0


def test_gram_a1():
    g = golden("gram.npz")
    for i in range(int(g["n_cases"])):
        c, ell, noise = (float(v) for v in g[f"c{i}_theta"])
        X, Y = dev(g[f"c{i}_X"]), dev(g[f"c{i}_Y"])
        assert np.allclose(ops.gram_rbf(X, None, c, ell, noise).cpu().numpy(), g[f"c{i}_K_one"], rtol=1e-13, atol=1e-300)
        assert np.allclose(ops.gram_rbf(X, Y, c, ell).cpu().numpy(), g[f"c{i}_K_two"], rtol=1e-13, atol=1e-300)


def test_potrf_a3_golden():
    g = golden("score_shared.npz")
    for i in range(int(g["n_cases"])):
        cov = g[f"c{i}_cov"]
        L, info, Linv, logdet = ops.potrf_batched(dev(cov), 1e-8, 0.0, want_inv=True, want_logdet=True)
        assert int(info[0]) == 0
        Lh = L[0].cpu().numpy()
        ref = g[f"c{i}_L"]
        assert np.allclose(Lh, ref, rtol=RT_WAVE, atol=RT_WAVE * np.abs(ref).max())
        assert np.array_equal(np.triu(Lh, 1), np.zeros_like(Lh))
        T = cov.shape[0]
        assert np.allclose(Linv[0].cpu().numpy() @ ref, np.eye(T), atol=1e-9)
        assert abs(float(logdet[0]) - 2 * np.log(np.diag(ref)).sum()) <= 1e-10 * max(1.0, abs(2 * np.log(np.diag(ref)).sum()))


def test_potrf_batch_and_failure_info():
    rng = np.random.default_rng(4)
    T, b = 40, 9
    Q = rng.normal(size=(b, T, T))
    A = Q @ Q.transpose(0, 2, 1) + 0.5 * np.eye(T)
    A[3, 17, 17] = -1.0                                   # not positive definite -> LAPACK-style info
    L, info = ops.potrf_batched(dev(A), 0.0, 0.0)
    info = info.cpu().numpy()
    assert info[3] > 0 and np.all(np.delete(info, 3) == 0)
    for k in (0, 8):
        assert np.allclose(L[k].cpu().numpy(), np.linalg.cholesky(0.5 * (A[k] + A[k].T)), rtol=1e-10, atol=1e-10)
    with pytest.raises(torch.linalg.LinAlgError):
        ops.raise_on_info(torch.as_tensor(info), "potrf")


def test_score_shared_cov_a4_golden():
    g = golden("score_shared.npz")
    for i in range(int(g["n_cases"])):
        cov, mean, Y = g[f"c{i}_cov"], g[f"c{i}_mean"], g[f"c{i}_Y"]
        B, T = Y.shape
        items = ops.build_items([0], [0.0], [B])
        quad, logdet, info = ops.score_groups(dev(Y), dev(mean), dev(cov), *items, want_logdet=True)
        assert int(info.abs().max()) == 0
        score = -0.5 * quad.cpu().numpy() - 0.5 * T * orc.LOG2PI
        assert rel_err(score, g[f"c{i}_score"]) < RT_WAVE
        assert np.allclose(logdet.cpu().numpy(), orc.quad_logdet(Y[0] - mean, cov)[1], rtol=1e-10)


@pytest.mark.parametrize("tag", ["t30", "t45", "t90", "t45l3"])
def test_grouped_scoring_a6_state(tag):
    """compute_sq_err_all on the shared grid: groups from the host logic, Sigma_i stack scored on the GPU."""
    g = golden(f"state_{tag}.npz")
    y = g["y"]
    n, T = y.shape
    idx = g["st_indexes"]
    C, f_star, Sig = g["st_C"], g["st_f_star"], g["st_Sigma"]
    for no_first, key in ((False, "q_shared"), (True, "q_shared_nofirst")):
        i_vals, first = orc.step_of_segments(idx, n, no_first)
        # oracle's state selection (a7) gives the (C, f_star) indices; means are C_i f_i
        st = orc.ClusterState(g["st_x_basis"], g["st_theta"], list(f_star), list(Sig), list(C), list(idx))
        code = i_vals * 2 + first
        order = np.argsort(code, kind="stable")
        codes, counts = np.unique(code, return_counts=True)
        means, mats, adds = [], [], []
        ini = 1e-2 * float(np.mean(np.diag(Sig[0])))
        for cd in codes:
            m, _, (ci, fi) = orc.observe_select(st, int(cd // 2))
            means.append(m)
            mats.append(ci)
            adds.append(ini if cd % 2 else 0.0)
        # one mean per group: pass a per-group mean table and per-group Sigma index through item_mat
        Sg = np.stack([Sig[ci] for ci in mats])
        items = ops.build_items(list(range(len(codes))), adds, counts.tolist())
        quad, _, info = ops.score_groups(dev(y), dev(np.stack(means)), dev(Sg), *items, seg_ids=order.astype(np.int32))
        assert int(info.abs().max()) == 0
        score = -0.5 * quad.cpu().numpy() - 0.5 * T * orc.LOG2PI
        assert rel_err(score, g[key]) < RT_WAVE


def _pairs_case(N, K, T, seed, irregular=True):
    b = orc.synthetic_batch(N, K, T, seed=seed, irregular=irregular)
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(b["x"]), dev(b["y"]))
    assert int(plan.info.abs().max()) == 0 and int(info.abs().max()) == 0
    sc, q_ref, ld_ref = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    return quad.cpu().numpy(), logdet.cpu().numpy(), q_ref, ld_ref, sc


@pytest.mark.parametrize("T", [20, 33, 64, 90, 128])
def test_pairs_a2_a5_synthetic(T):
    quad, logdet, q_ref, ld_ref, sc = _pairs_case(5, 3, T, seed=100 + T)
    assert rel_err(quad, q_ref) < RT_PAIR
    assert rel_err(logdet, ld_ref) < RT_PAIR
    # hard assignments (arg-max over clusters of the reference's score) are identical
    score = -0.5 * quad - 0.5 * T * orc.LOG2PI
    assert np.array_equal(np.argmax(score, axis=1), np.argmax(sc, axis=1))


def test_pairs_iso_and_first_and_mixed_lengthscales():
    rng = np.random.default_rng(9)
    T, N, K = 48, 6, 5
    b = orc.synthetic_batch(N, K, T, seed=77)
    b["Sigma"][1] = 1.7 * np.eye(T)                       # iso-diagonal state: GPI.py:497-498 branch
    b["theta"][3, 1] = 0.9                                # a second length-scale group
    b["theta"][4, 1] = 0.9
    fn = np.zeros((N, K))
    fn[::2, 0] = 0.03
    fn[1, 1] = 0.5
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(b["x"]), dev(b["y"]), first_noise=dev(fn))
    sc, q_ref, ld_ref = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"], first_noise=fn)
    assert rel_err(quad.cpu().numpy(), q_ref) < RT_PAIR
    assert rel_err(logdet.cpu().numpy(), ld_ref) < RT_PAIR


@pytest.mark.parametrize("tag", ["t30", "t45", "t90", "t45l3"])
def test_pairs_against_reference_irregular_grids(tag):
    """q_irr of the fixtures = the reference's compute_sq_err_all on irregular grids (per-segment state)."""
    g = golden(f"state_{tag}.npz")
    y, x = g["y"], g["x_irr"]
    n, T = y.shape
    idx = g["st_indexes"]
    st = orc.ClusterState(g["st_x_basis"], g["st_theta"], list(g["st_f_star"]), list(g["st_Sigma"]), list(g["st_C"]), list(idx))
    i_vals, first = orc.step_of_segments(idx, n)
    # every distinct LDS step is one "cluster" of the batch; each segment reads its own column
    steps = np.unique(i_vals)
    means = np.stack([orc.observe_select(st, int(s))[0] for s in steps])
    Sig = np.stack([orc.observe_select(st, int(s))[1] for s in steps])
    theta = np.repeat(np.asarray(st.theta)[None, :], len(steps), axis=0)
    col = np.searchsorted(steps, i_vals)
    fn = np.zeros((n, len(steps)))
    fn[np.arange(n), col] = np.where(first, 1e-2 * float(np.mean(np.diag(st.Sigma[0]))), 0.0)
    plan = ops.PairsPlan(T, T, theta).update(dev(st.x_basis), dev(means), dev(Sig))
    score, info = plan.score(dev(x), dev(y), first_noise=dev(fn))
    got = score.cpu().numpy()[np.arange(n), col]
    assert rel_err(got, g["q_irr"]) < RT_PAIR
    # and the online-style call against the last state (i = -1)
    mean_last = (st.C[-1] @ st.f_star[-1])[None]
    plan1 = ops.PairsPlan(T, T, np.asarray(st.theta)[None]).update(dev(st.x_basis), dev(mean_last), dev(st.Sigma[-1][None]))
    s1, _ = plan1.score(dev(x), dev(y))
    assert rel_err(s1.cpu().numpy()[:, 0], g["lse_last"]) < RT_PAIR


def test_pairs_accuracy_bound_tracks_conditioning():
    """The explicit per-cluster operator loses digits when K~ is ill-conditioned; the plan reports a bound and routes
    the clusters above its tolerance to the solve-based kernel (tests/test_gpu_pairs_acc.py)."""
    T, N, K = 48, 4, 2
    b = orc.synthetic_batch(N, K, T, seed=5)
    b["theta"][1, 1] = 2.5                                # smooth kernel: cond(K~) ~ 1e7
    _, q_ref, _ = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    # explicit operator everywhere (acc_tol < 0): the error of the ill-conditioned cluster follows the indicator
    plan = ops.PairsPlan(T, T, b["theta"], acc_tol=-1.0).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    bound = plan.accuracy_bound()
    assert bound[0] < 1e-8 and bound[1] > 1e-6
    assert not plan.solve_based().any()
    quad, _, _ = plan.loglik(dev(b["x"]), dev(b["y"]))
    err = np.abs(quad.cpu().numpy() - q_ref) / np.abs(q_ref)
    assert err[:, 0].max() < RT_PAIR
    assert err[:, 1].max() < 10 * bound[1]                # the indicator is of the right order
    # default tolerance: the ill-conditioned cluster (only) is scored by the solve-based kernel, at full accuracy
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    assert list(plan.solve_based()) == [False, True]
    quad, _, info = plan.loglik(dev(b["x"]), dev(b["y"]))
    assert int(info.abs().max()) == 0
    assert rel_err(quad.cpu().numpy(), q_ref) < RT_PAIR


@pytest.mark.parametrize("T", [64, 90, 128])
def test_pairs_block_skipping_on_arbitrary_grids(T):
    """The kernel drops 16x16 blocks of the cross-Gram whose entries are all < 1e-36, and whole half-sweeps whose rows of
    M'E no active block reads; both decisions are taken from the data, so shifted, reversed, permuted and far-away grids
    must all agree with the oracle."""
    K = 3
    b = orc.synthetic_batch(6, K, T, seed=31)
    rng = np.random.default_rng(3)
    x = b["x"].copy()
    x[1] = x[1] + 37.5                      # shifted: most blocks inactive, band off the diagonal
    x[2] = x[2][::-1].copy()                # reversed: anti-diagonal band
    x[3] = rng.permutation(x[3])            # permuted: no band structure at all
    x[4] = x[4] + 500.0                     # far away: every block of E inactive (cov = K** only)
    x[5] = np.linspace(10.0, 20.0, T)       # dense cluster of points: few wide active blocks
    x[0] = np.concatenate((x[0][T // 2:], x[0][:T // 2]))     # halves swapped: the early panels need the LOWER rows of M'E
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(x), dev(b["y"]))
    assert int(info.abs().max()) == 0
    _, q_ref, ld_ref = orc.loglik_pairs(x, b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    assert rel_err(quad.cpu().numpy(), q_ref) < RT_PAIR
    assert rel_err(logdet.cpu().numpy(), ld_ref) < RT_PAIR


# ------------------------------------------------------------------ 128 < T <= 256: cooperative kernels
@pytest.mark.parametrize("T", [144, 192, 256])
def test_large_T_potrf_and_score(T):
    rng = np.random.default_rng(T)
    b = 3
    Q = rng.normal(size=(b, T, T))
    A = Q @ Q.transpose(0, 2, 1) / T + np.diag(rng.uniform(0.5, 2.0, T)) + 1e-3 * rng.normal(size=(b, T, T))
    L, info, Linv, logdet = ops.potrf_batched(dev(A), 1e-8, 0.25, want_inv=True, want_logdet=True)
    assert int(info.abs().max()) == 0
    for k in range(b):
        S = 0.5 * (A[k] + A[k].T) + 0.25 * np.eye(T)
        S = S + 1e-8 * np.mean(np.abs(np.diag(S))) * np.eye(T)
        ref = np.linalg.cholesky(S)
        Lh = L[k].cpu().numpy()
        assert np.allclose(Lh, ref, rtol=1e-10, atol=1e-11)
        assert np.array_equal(np.triu(Lh, 1), np.zeros_like(Lh))
        assert np.allclose(Linv[k].cpu().numpy() @ ref, np.eye(T), atol=1e-9)
        assert abs(float(logdet[k]) - 2 * np.log(np.diag(ref)).sum()) < 1e-9 * T
    mean = rng.normal(size=(b, T))
    Y = rng.normal(size=(40, T)) * 2
    grp = rng.integers(0, b, size=40)
    order = np.argsort(grp, kind="stable")
    counts = np.bincount(grp, minlength=b)
    im, ia, io, ic = ops.build_items(list(range(b)), [0.0, 0.3, 0.0], counts.tolist())
    # large T: at most 16 segments ride along one factorisation
    quad, logdet2, info2 = ops.score_groups(dev(Y), dev(mean), dev(A), im, ia, io, ic, seg_ids=order.astype(np.int32), want_logdet=True)
    assert int(info2.abs().max()) == 0
    for n in range(40):
        k = grp[n]
        cov = A[k] + (0.3 if k == 1 else 0.0) * np.eye(T)
        q_ref, ld_ref = orc.quad_logdet(Y[n] - mean[k], cov)
        assert abs(float(quad[n]) - q_ref) <= 1e-9 * abs(q_ref)
        assert abs(float(logdet2[n]) - ld_ref) <= 1e-9 * abs(ld_ref)


@pytest.mark.parametrize("T", [160, 256])
def test_large_T_pairs(T):
    N, K = 3, 2
    b = orc.synthetic_batch(N, K, T, seed=900 + T)
    if T == 160:
        b["Sigma"][1] = 2.2 * np.eye(T)                   # iso branch of the cooperative kernel
    fn = np.zeros((N, K))
    fn[0, 0] = 0.04
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    assert int(plan.info.abs().max()) == 0
    quad, logdet, info = plan.loglik(dev(b["x"]), dev(b["y"]), first_noise=dev(fn))
    assert int(info.abs().max()) == 0
    _, q_ref, ld_ref = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"], first_noise=fn)
    assert rel_err(quad.cpu().numpy(), q_ref) < RT_PAIR
    assert rel_err(logdet.cpu().numpy(), ld_ref) < RT_PAIR


@pytest.mark.parametrize("T,N", [(150, 720), (256, 40)])
def test_large_T_pairs_arbitrary_grids_and_selection(T, N):
    """Cooperative kernel (one workgroup per pair): E lives in LDS as a list of active 16x16 blocks; grids without
    band structure (permuted, reversed) activate more blocks than fit and spill to the workgroup's global scratch
    area.  More pairs than areas can be in flight at once exercises the hand-out / release of those areas.  `sel`
    scores every segment against one cluster only."""
    K = 3
    b = orc.synthetic_batch(N, K, T, seed=77 + T)
    rng = np.random.default_rng(5)
    x = b["x"].copy()
    for n in range(0, N, 4):
        x[n] = rng.permutation(x[n])           # no band structure: every block of E is active
    x[1] = x[1][::-1].copy()                    # anti-diagonal band
    x[2] = x[2] + 61.0                          # shifted band
    x[3] = x[3] + 5000.0                        # nothing active
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(x), dev(b["y"]))
    assert int(info.abs().max()) == 0
    _, q_ref, ld_ref = orc.loglik_pairs(x, b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    assert rel_err(quad.cpu().numpy(), q_ref) < RT_PAIR
    assert rel_err(logdet.cpu().numpy(), ld_ref) < RT_PAIR
    sel = torch.as_tensor(rng.integers(0, K, N), dtype=torch.int32, device="cuda")
    q1, l1, i1 = plan.loglik(dev(x), dev(b["y"]), sel=sel)
    idx = sel.cpu().numpy()
    assert int(i1.abs().max()) == 0
    assert rel_err(q1.cpu().numpy(), q_ref[np.arange(N), idx]) < RT_PAIR
    assert rel_err(l1.cpu().numpy(), ld_ref[np.arange(N), idx]) < RT_PAIR
    # a second call on the same plan: every scratch area must have been released
    q2, _, _ = plan.loglik(dev(x), dev(b["y"]))
    assert torch.equal(q2, quad)


@pytest.mark.parametrize("T", [8, 33, 90, 128])
def test_score_each_own_state_per_segment(T):
    rng = np.random.default_rng(40 + T)
    n, S = 37, 9
    Q = rng.normal(size=(S, T, T))
    Sig = Q @ Q.transpose(0, 2, 1) / T + np.eye(T) + 1e-3 * rng.normal(size=(S, T, T))
    mean = rng.normal(size=(5, T))
    Y = rng.normal(size=(n, T)) * 2
    sm, sme = rng.integers(0, S, n), rng.integers(0, 5, n)
    add = np.where(rng.random(n) < 0.3, 0.07, 0.0)
    quad, logdet, info = ops.score_each(dev(Y), dev(mean), dev(Sig), sm.astype(np.int32), sme.astype(np.int32), add, want_logdet=True)
    assert int(info.abs().max()) == 0
    for i in range(n):
        q_ref, ld_ref = orc.quad_logdet(Y[i] - mean[sme[i]], Sig[sm[i]] + add[i] * np.eye(T))
        assert abs(float(quad[i]) - q_ref) <= RT_WAVE * abs(q_ref)
        assert abs(float(logdet[i]) - ld_ref) <= 1e-10 * max(1.0, abs(ld_ref))


def test_score_each_symmetric_fast_path_matches():
    rng = np.random.default_rng(77)
    n, T = 50, 90
    Q = rng.normal(size=(n, T, T))
    Sig = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    Sig = 0.5 * (Sig + Sig.transpose(0, 2, 1))
    Y, mean = rng.normal(size=(n, T)), rng.normal(size=(n, T))
    sm = np.arange(n, dtype=np.int32)
    q0, ld0, _ = ops.score_each(dev(Y), dev(mean), dev(Sig), sm, want_logdet=True)
    q1, ld1, i1 = ops.score_each(dev(Y), dev(mean), dev(Sig), sm, want_logdet=True, symmetric=True)
    assert int(i1.abs().max()) == 0
    assert torch.equal(q0, q1) and torch.equal(ld0, ld1)     # same arithmetic: 0.5 (a + a) == a exactly
