"""Direct tests of the member-step entry points of hgp_chain.hip through the C-ABI: the device-resident
product lists and the Cholesky inverse with riding right-hand sides, against NumPy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def test_gemm_list_heterogeneous_items():
    from hdpgpc_amd import ops
    rng = np.random.default_rng(5)
    gl = ops.GemmList("cuda")
    cases = []
    shapes = [(90, 90, 90), (16, 16, 16), (17, 33, 5), (128, 128, 128), (1, 7, 100), (45, 1, 45), (100, 90, 3), (64, 1, 1),
              (256, 256, 256), (200, 1, 256), (144, 192, 130)]
    for idx, (M, N, K) in enumerate(shapes):
        tA, tB = bool(idx & 1), bool(idx & 2)
        A = rng.normal(size=(K, M) if tA else (M, K))
        B = rng.normal(size=(N, K) if tB else (K, N))
        D = rng.normal(size=(M, N)) if idx % 3 != 0 else None
        alpha, beta = [1.0, -1.0, 0.5][idx % 3], [1.0, -1.0, 2.0][(idx + 1) % 3]
        eye = 1.0 if (M == N and idx % 2 == 0) else 0.0
        dA, dB, dD = dev(A), dev(B), (dev(D) if D is not None else None)
        out = torch.full((M, N), np.nan, dtype=torch.float64, device="cuda")
        out2 = torch.full((M, N), np.nan, dtype=torch.float64, device="cuda") if idx % 4 == 1 else None
        if N == 1:          # vectors are passed 1-D (read as columns)
            dB1 = dB.reshape(-1) if not tB else dB
            gl.add(dA, dB1 if not tB else dB, out.reshape(-1), D=None if dD is None else dD.reshape(-1), transA=tA, transB=tB,
                   alpha=alpha, beta=beta, add_eye=0.0, out2=None if out2 is None else out2.reshape(-1))
            eye = 0.0
        else:
            gl.add(dA, dB, out, D=dD, transA=tA, transB=tB, alpha=alpha, beta=beta, add_eye=eye, out2=out2)
        ref = alpha * ((A.T if tA else A) @ (B.T if tB else B))
        if D is not None:
            ref = ref + beta * D
        if eye:
            ref = ref + eye * np.eye(M)
        cases.append((out, out2, ref))
    gl.run()
    gl.run()                                            # a list is replayed (hipGraph): same result every time
    gl.run_range(2, 3)                                  # a sub-range of the list is one launch too
    torch.cuda.synchronize()
    for out, out2, ref in cases:
        assert np.allclose(out.cpu().numpy(), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        if out2 is not None:
            assert np.array_equal(out2.cpu().numpy(), out.cpu().numpy())
    with pytest.raises(ValueError):
        gl.add(dev(np.zeros((4, 5))), dev(np.zeros((6, 4))), torch.zeros((4, 4), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        gl.add(dev(np.zeros((257, 4))), dev(np.zeros((4, 4))), torch.zeros((257, 4), dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("T", [8, 33, 90, 128])
@pytest.mark.parametrize("rhs_trans", [False, True])
def test_chol_inverse_with_riding_right_hand_sides(T, rhs_trans):
    from hdpgpc_amd import ops
    rng = np.random.default_rng(T)
    b = 3
    Q = rng.normal(size=(b, T, T))
    A = Q @ Q.transpose(0, 2, 1) / T + np.eye(T) + 1e-3 * rng.normal(size=(b, T, T))     # slightly non-symmetric: symmetrised on load
    R = rng.normal(size=(b, T, T))
    on = np.array([1, 0, 1], dtype=np.int32)
    Z = torch.full((b, T, T), np.nan, dtype=torch.float64, device="cuda")
    Y = torch.zeros((b, T, T), dtype=torch.float64, device="cuda")
    info = torch.full((b,), -7, dtype=torch.int32, device="cuda")
    ops.chol_inverse_rhs(dev(A), Z, dev(R), Y, info, rhs_on=torch.as_tensor(on, device="cuda"), rhs_trans=rhs_trans, add_diag=1e-8)
    torch.cuda.synchronize()
    assert info.tolist() == [0, 0, 0]
    for m in range(b):
        L = np.linalg.cholesky(0.5 * (A[m] + A[m].T) + 1e-8 * np.eye(T))
        Zr = np.linalg.inv(L)
        assert np.allclose(Z[m].cpu().numpy(), Zr, rtol=1e-10, atol=1e-12 * np.abs(Zr).max())
        if on[m]:
            Yr = Zr @ (R[m].T if rhs_trans else R[m])
            assert np.allclose(Y[m].cpu().numpy(), Yr, rtol=1e-10, atol=1e-11 * np.abs(Yr).max())
        else:
            assert not Y[m].any()                      # untouched
    # solves only (no inverse wanted), and a matrix that is not positive definite
    Abad = A.copy()
    Abad[1] = -Abad[1]
    info2 = torch.zeros((b,), dtype=torch.int32, device="cuda")
    Y2 = torch.zeros((b, T, T), dtype=torch.float64, device="cuda")
    ops.chol_inverse_rhs(dev(Abad), None, dev(R), Y2, info2, rhs_trans=rhs_trans, add_diag=1e-8)
    torch.cuda.synchronize()
    got = info2.tolist()
    assert got[0] == 0 and got[2] == 0 and got[1] == 1            # LAPACK-style: first non-positive pivot, 1-based
    assert np.allclose(Y2[0].cpu().numpy(), Y[0].cpu().numpy(), rtol=1e-13, atol=0.0)


@pytest.mark.parametrize("T", [17, 40, 50, 64, 70, 90, 100, 113, 128])
def test_chol_inverse_rhs_small_and_large_batches_agree_bit_for_bit(T):
    """Few matrices take one workgroup per (matrix, panel) (k_coop_inv_rhs), many take one wave per panel (k_wave_inv_rhs): the same
    tile arithmetic in the same order, so a matrix must come out identical whichever batch it rides in - including its info."""
    from hdpgpc_amd import ops
    rng = np.random.default_rng(100 + T)
    nblk = (T + 15) // 16
    big = 256 // (2 * nblk) + 5                        # beyond the cooperative kernel's one round of workgroups
    Q = rng.normal(size=(big, T, T))
    A = Q @ Q.transpose(0, 2, 1) / T + 0.5 * np.eye(T)
    A[1] = -A[1]                                       # not positive definite
    R = rng.normal(size=(big, T, T))
    on = (np.arange(big) % 3 != 2).astype(np.int32)

    def run(n, trans):
        Z = torch.zeros((n, T, T), dtype=torch.float64, device="cuda")
        Y = torch.zeros((n, T, T), dtype=torch.float64, device="cuda")
        info = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        ops.chol_inverse_rhs(dev(A[:n]), Z, dev(R[:n]), Y, info, rhs_on=torch.as_tensor(on[:n], device="cuda"), rhs_trans=trans, add_diag=1e-8)
        torch.cuda.synchronize()
        return Z.cpu().numpy(), Y.cpu().numpy(), info.cpu().numpy()

    for trans in (False, True):
        Zs, Ys, Is = run(4, trans)
        Zb, Yb, Ib = run(big, trans)
        assert Is.tolist() == [0, 1, 0, 0] and np.array_equal(Is, Ib[:4]) and not Ib[4:].any()
        for m in (0, 2, 3):
            assert np.array_equal(Zs[m], Zb[m]) and np.array_equal(Ys[m], Yb[m])
        Lr = np.linalg.inv(np.linalg.cholesky(A[big - 1] + 1e-8 * np.eye(T)))
        assert np.allclose(Zb[big - 1], Lr, rtol=1e-10, atol=1e-12 * np.abs(Lr).max())


@pytest.mark.parametrize("T", [90, 144, 192, 256])
def test_chol_inverse_only_up_to_256_with_caller_buffers(T):
    """hgp_chol_inverse_batched_f64: Z = L^-1 without touching A, caller-allocated outputs (the graphed member step), LAPACK-style
    info also from the cooperative kernels (128 < T <= 256)."""
    from hdpgpc_amd import ops
    rng = np.random.default_rng(T + 1)
    b = 3
    Q = rng.normal(size=(b, T, T))
    A = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    A[2] = -A[2]                                            # not positive definite: pivot 1 fails
    dA = dev(A)
    keep = dA.clone()
    Z = torch.full((b, T, T), np.nan, dtype=torch.float64, device="cuda")
    info = torch.full((b,), -3, dtype=torch.int32, device="cuda")
    Zr, ir = ops.chol_inverse(dA, 0.0, 1e-8, out=Z, info=info)
    torch.cuda.synchronize()
    assert Zr.data_ptr() == Z.data_ptr() and ir.data_ptr() == info.data_ptr()
    assert torch.equal(dA, keep)                            # input untouched
    assert info.tolist() == [0, 0, 1]
    for m in range(2):
        ref = np.linalg.inv(np.linalg.cholesky(0.5 * (A[m] + A[m].T) + 1e-8 * np.eye(T)))
        assert np.allclose(Z[m].cpu().numpy(), ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
