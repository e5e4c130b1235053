"""a8 (log_lat_error) and a9 (log_likelihood_MNIW): the fused one-wave-per-item kernels (T <= 128) and the composed path
(128 < T <= 256), batched, against the oracle's restatement of GPI_model.py:288-323,1346-1362 (itself pinned to the
reference's outputs by tests/test_oracle_golden.py; the reference's own values are checked through the mirror API in
tests/test_gpu_mirror_api.py).  Also: score_each / chol_inverse beyond T = 128."""
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


def spd(rng, b, T, scale=1.0):
    Q = rng.normal(size=(b, T, T))
    return scale * (Q @ Q.transpose(0, 2, 1) / T + np.eye(T) * rng.uniform(0.2, 1.0, (b, 1, 1)))


@pytest.mark.parametrize("T,b", [(8, 3), (33, 5), (64, 4), (90, 37), (128, 6), (129, 2), (144, 2), (200, 5), (256, 2), (256, 9)])
def test_lat_error_a8(T, b):
    rng = np.random.default_rng(T)
    A = np.eye(T)[None] + 0.05 * rng.normal(size=(b, T, T))
    Gam, P = spd(rng, b, T, 0.3), spd(rng, b, T, 2.0)
    Gam = Gam + 1e-3 * rng.normal(size=Gam.shape)           # slightly non-symmetric: _chol_spd symmetrises
    fc, fp = rng.normal(size=(b, T)) * 5, rng.normal(size=(b, T)) * 5
    out, info = ops.lat_error(dev(fc), dev(fp), dev(A), dev(Gam), dev(P))
    assert int(info.abs().max()) == 0
    ref = np.array([orc.lat_error_terms(fc[i], fp[i], A[i], Gam[i], P[i]) for i in range(b)])
    assert rel_err(out.cpu().numpy() - 0.5 * T * orc.LOG2PI, ref) < 1e-9


@pytest.mark.parametrize("T,b", [(8, 3), (33, 5), (64, 4), (90, 37), (128, 6), (129, 2), (144, 2), (200, 5), (256, 2), (256, 9)])
@pytest.mark.parametrize("prior", ["shared_identity", "per_item_diagonal", "per_item_dense"])
def test_mniw_loglik_a9(T, b, prior):
    rng = np.random.default_rng(1000 + T)
    M = np.eye(T)[None] + 0.05 * rng.normal(size=(b, T, T))
    Sig = spd(rng, b, T, 0.7)
    if prior == "shared_identity":     # the hot path: prior = (C_def, I, Sigma_def) shared by all items (GPI_model.py:481-484)
        mm, R, S = np.eye(T) * 0.9, None, np.diag(rng.uniform(0.5, 3.0, T))
        out, info = ops.mniw_loglik(dev(M), dev(Sig), dev(mm), None, dev(S))
        ref = np.array([orc.mniw_log_likelihood(M[i], Sig[i], mm, np.eye(T), S) for i in range(b)])
    elif prior == "per_item_diagonal":  # every item its own prior mean and diagonal scale (the online step's batched candidates)
        mm = 0.1 * rng.normal(size=(b, T, T))
        S = np.stack([np.diag(rng.uniform(0.5, 3.0, T)) for _ in range(b)])
        out, info = ops.mniw_loglik(dev(M), dev(Sig), dev(mm), None, dev(S), scale_is_diagonal=True)
        ref = np.array([orc.mniw_log_likelihood(M[i], Sig[i], mm[i], np.eye(T), S[i]) for i in range(b)])
    else:
        mm, R, S = 0.1 * rng.normal(size=(b, T, T)), spd(rng, b, T), spd(rng, b, T, 3.0)
        out, info = ops.mniw_loglik(dev(M), dev(Sig), dev(mm), dev(R), dev(S))
        ref = np.array([orc.mniw_log_likelihood(M[i], Sig[i], mm[i], R[i], S[i]) for i in range(b)])
    assert int(info.abs().max()) == 0
    assert rel_err(out.cpu().numpy(), ref) < 1e-9


def test_matlik_coop_reports_non_spd_items():
    """The fused cooperative kernels (128 < T <= 256): LAPACK-style info per item, NaN in the output of a failed item only."""
    rng = np.random.default_rng(6)
    T, b = 200, 4
    Gam = spd(rng, b, T)
    Gam[1, 150, 150] = -3.0
    A = np.tile(np.eye(T), (b, 1, 1))
    out, info = ops.lat_error(dev(rng.normal(size=(b, T))), dev(rng.normal(size=(b, T))), dev(A), dev(Gam), dev(spd(rng, b, T)))
    info, out = info.cpu().numpy(), out.cpu().numpy()
    assert info[1] > 0 and np.all(np.delete(info, 1) == 0) and np.isnan(out[1]) and np.all(np.isfinite(np.delete(out, 1)))
    out, info = ops.mniw_loglik(dev(A), dev(Gam), dev(np.eye(T)), None, dev(2.0 * np.eye(T)), scale_is_diagonal=True)
    info, out = info.cpu().numpy(), out.cpu().numpy()
    assert info[1] > 0 and np.all(np.delete(info, 1) == 0) and np.isnan(out[1]) and np.all(np.isfinite(np.delete(out, 1)))


def test_matlik_reports_non_spd_items():
    rng = np.random.default_rng(5)
    T, b = 40, 6
    Gam = spd(rng, b, T)
    Gam[2, 11, 11] = -3.0
    out, info = ops.lat_error(dev(rng.normal(size=(b, T))), dev(rng.normal(size=(b, T))), dev(np.tile(np.eye(T), (b, 1, 1))),
                              dev(Gam), dev(spd(rng, b, T)))
    info = info.cpu().numpy()
    assert info[2] > 0 and np.all(np.delete(info, 2) == 0)
    with pytest.raises(torch.linalg.LinAlgError):
        ops.raise_on_info(torch.as_tensor(info), "log_lat_error")


@pytest.mark.parametrize("T", [144, 200, 256])
def test_score_each_and_inverse_beyond_128(T):
    """configs[4] is 'T = 256 online': the member dataflow (one Sigma_i, one right-hand side per segment) and the
    Cholesky inverse run on the cooperative kernels there."""
    rng = np.random.default_rng(T)
    n = 5
    Sig = spd(rng, n, T)
    Y, mean = rng.normal(size=(n, T)), rng.normal(size=(n, T))
    add = np.array([0.0, 0.02, 0.0, 0.3, 0.0])
    quad, logdet, info = ops.score_each(dev(Y), dev(mean), dev(Sig), np.arange(n, dtype=np.int32), seg_add=add, want_logdet=True)
    assert int(info.abs().max()) == 0
    for i in range(n):
        q, ld = orc.quad_logdet(Y[i] - mean[i], Sig[i] + add[i] * np.eye(T))
        assert abs(float(quad[i]) - q) <= 1e-10 * abs(q) and abs(float(logdet[i]) - ld) <= 1e-10 * abs(ld)
    Z, info = ops.chol_inverse(dev(Sig))
    assert int(info.abs().max()) == 0
    for i in range(n):
        L = np.linalg.cholesky(0.5 * (Sig[i] + Sig[i].T))
        assert np.allclose(Z[i].cpu().numpy() @ L, np.eye(T), atol=1e-10)
