"""the kernel hyper-parameter fit of a new cluster (hdpgpc_amd.kernel_fit), Adam on the exact marginal
log-likelihood through the a10 value / gradient kernels.  gpytorch is absent here, so there is no reference vector
(parity unpinned); checked instead: (1) the GPU trajectory equals a NumPy restatement of the same optimiser
on the oracle's log-likelihood, (2) the loss settles, bounds hold, the gradient vanishes at the end, (3) the producer
uses it when no theta is injected and ends with the reference's post-fit conventions (length-scale 1.2, GPI.py:711)."""
import math

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import hdpgpc_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from hdpgpc_amd.GPI import RBFWhiteKernel
    from hdpgpc_amd.GPI_model import GPI_model
    from hdpgpc_amd.kernel_fit import fit_kernel_adam


from kernel_fit_ref import numpy_adam  # noqa: E402  (the restatement pinned by the reference's notebook output)


def _beat():
    g = golden("mitbih100_lead0.npz")
    key = "y" if "y" in g.files else g.files[0]
    y = np.asarray(g[key], dtype=np.float64)
    return y[0].reshape(-1)


def test_gpu_adam_matches_numpy_restatement():
    y = _beat()
    x = np.arange(float(y.size))
    bounds = (1e-3, 20.0)
    th_g, tr_g = fit_kernel_adam(x, y, bounds, max_iter=150, min_iter=10 ** 9, return_trace=True)
    th_n, tr_n = numpy_adam(x, y, bounds, 150)
    assert np.allclose(tr_g, tr_n, rtol=1e-7, atol=1e-9)
    assert np.allclose(th_g, th_n, rtol=1e-6)
    assert tr_g[-1] < tr_g[0]


@pytest.mark.parametrize("i", [0, 2])
def test_gpu_fit_reproduces_the_notebook_output(i):
    """The reference's own printed fits (hdpgpc/tests/test_step.ipynb cells 22 / 33, tests/golden/kernel_fit_notebook.npz): the
    GPU fit prints the same losses (three decimals) and ends at the same raw gpytorch parameters."""
    g = golden("kernel_fit_notebook.npz")
    y = g["y"][i]
    x = np.arange(float(y.size))
    s = float(g["std_cell8"])
    (c, ell, noise, mean), tr = fit_kernel_adam(x, y, (0.1 * s, 0.2 * s), min_iter=10 ** 9, return_trace=True)
    assert len(tr) == 4000
    assert np.max(np.abs(tr[g["iters"] - 1] - g["losses"][i])) <= 6e-4
    raw = g["raw_final"][i]
    sp = lambda v: math.log1p(math.exp(-abs(v))) + max(v, 0.0)  # noqa: E731
    want = (sp(raw[2]), sp(raw[3]), 0.1 * s + 0.1 * s / (1.0 + math.exp(-raw[0])), raw[1])
    assert np.allclose((c, ell, noise, mean), want, rtol=1e-5)


def test_fit_runs_the_reference_schedule_inside_bounds():
    """4000 Adam steps at most, early stop not before 1000 (GPI.py:660,689-693); like the reference's own run on a beat
    (tests/test_step.ipynb cell 22: loss 788.8 -> 16.6 after all 4000 iterations) the budget usually ends the fit."""
    y = _beat()
    x = np.arange(float(y.size))
    bounds = (1e-3, 20.0)
    (c, ell, noise, mean), tr = fit_kernel_adam(x, y, bounds, return_trace=True)
    assert bounds[0] < noise < bounds[1] and c > 0 and ell > 0
    assert 1000 < len(tr) <= 4000
    assert tr[-1] < 0.1 * tr[0] and np.all(np.isfinite(tr))
    assert np.max(np.diff(tr[200:])) < 1e-2 * abs(tr[200])            # no blow-ups once the step size has adapted
    # the last recorded loss is the oracle's log-likelihood at the parameters of the step before the final update
    (c2, l2, n2, m2), tr2 = fit_kernel_adam(x, y, bounds, max_iter=len(tr) - 1, min_iter=10 ** 9, return_trace=True)
    ref = -orc.log_marginal_likelihood(x, y - m2, (c2, l2, n2), faithful=False) / y.size
    assert abs(tr[-1] - ref) <= 1e-8 * abs(ref)


def test_producer_fits_when_no_theta_is_injected():
    g = golden("state_t45.npz")
    y = g["y"]
    n, T = y.shape
    sigma, gamma = float(g["st_Sigma"][0][0, 0]), float(g["st_Gamma"][0][0, 0])
    m = GPI_model(RBFWhiteKernel(300.0, 3.0, sigma * 1e-5), g["st_x_basis"][:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
    m.noise_bounds = (sigma * 1e-5, sigma * 2.0)
    cond = m.GPR_dynamic(gamma, sigma)
    m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
    assert m.fixed_theta is None
    xs = np.repeat(g["st_x_basis"][None, :, None], n, axis=0)
    resp = np.zeros(n)
    resp[[0, 1, 2, 3, 4]] = 1.0
    q, q_lat = m.full_pass_weighted(xs, y[:, :, None], resp)
    c, ell, noise = m.gp.kernel.params()
    assert ell == 1.2 and c > 0 and m.noise_bounds[0] <= noise <= m.noise_bounds[1]       # GPI.py:708-714
    assert m.fitted and m.indexes == [0, 1, 2, 3, 4] and bool(torch.isfinite(q).all())
    assert float(m.Sigma[0][0, 0]) == sigma                                            # Sigma is reset to the INITIAL sigma
