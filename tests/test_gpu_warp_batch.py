"""Warping_system.compute_warp_batch as one HIP launch (hgp_warp_batch_f64) against the reference's own
outputs (tests/golden/warp_batch.npz: torch autograd + torch.optim.Adam on the CPU): the warps, the warped observations,
the GP-prior scores and the per-iteration batch-mean loss, for scalar / tuple / absent theta, weights, a second output
dimension, a non-unit grid and the warm start across calls."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hdpgpc_amd.amtgp_warping_system import Warping_system


def test_compute_warp_batch_golden():
    g = golden("warp_batch.npz")
    for i in range(int(g["n_cases"])):
        iters, recursive, reps, n_ctrl, lr, ls, la, nw, lo, hi = g[f"c{i}_meta"]
        th = g[f"c{i}_theta"]
        theta = None if np.isnan(th[0]) else (float(th[0]) if th.size == 1 else tuple(float(v) for v in th))
        x = g[f"c{i}_x"]
        ws = Warping_system(x[:, None], noise_warp=float(nw), bound_noise_warp=(float(lo), float(hi)), recursive=bool(recursive),
                            bayesian=True, mode="rough")
        assert ws.n_ctrl == int(n_ctrl) and ws.lr == lr
        w = g[f"c{i}_w"] if f"c{i}_w" in g.files else None
        for rep in range(int(reps)):
            xw, yw, lik, tr = ws.compute_warp_batch(x, g[f"c{i}_Yt"], g[f"c{i}_Ym"], theta=theta, noise=g[f"c{i}_noise"],
                                                    weights=w, train_iter=int(iters))
            ref_xw, ref_yw, ref_lik, ref_loss = g[f"c{i}_r{rep}_xw"], g[f"c{i}_r{rep}_yw"], g[f"c{i}_r{rep}_lik"], g[f"c{i}_r{rep}_loss"]
            assert xw.shape == (ref_xw.shape[0], x.size, 1) and yw.shape == ref_yw.shape
            assert np.allclose(tr["loss"], ref_loss, rtol=1e-7, atol=1e-9), (i, rep)
            assert np.allclose(xw.cpu().numpy()[:, :, 0], ref_xw, rtol=1e-6, atol=1e-7 * np.abs(ref_xw).max()), (i, rep)
            assert np.allclose(yw.cpu().numpy(), ref_yw, rtol=1e-6, atol=1e-7 * np.abs(ref_yw).max()), (i, rep)
            assert np.allclose(lik.cpu().numpy(), ref_lik, rtol=1e-6), (i, rep)
            # monotone warps that keep the end points (amtgp_warping_system.py:333-351)
            gq = xw.cpu().numpy()[:, :, 0] + x[None, :]
            assert np.all(np.diff(gq, axis=1) > 0) and np.allclose(gq[:, 0], x[0]) and np.allclose(gq[:, -1], x[-1])
