"""TEST INFRASTRUCTURE: gpytorch's fit of IterativeGaussianProcess.fit_torch (GPI.py:610-770) restated with NumPy on the oracle's
log-marginal-likelihood - constant mean, ScaleKernel(RBFKernel) + Gaussian likelihood in gpytorch's default parameterisation
(softplus for output-scale and length-scale, a sigmoid interval between the white-kernel bounds for the noise, every raw
parameter 0 at the start), loss = -log p(y) / T, torch.optim.Adam(lr = 0.1) in its operation order.
Pinned by the reference's own printed output (tests/golden/kernel_fit_notebook.npz, tests/test_kernel_fit_notebook.py)."""
import math

import numpy as np

from oracle import hdpgpc_oracle as orc


def numpy_adam(x, y, bounds, iters, lr=0.1, return_raw=False):
    sp = lambda v: math.log1p(math.exp(-abs(v))) + max(v, 0.0)  # noqa: E731
    sg = lambda v: 1.0 / (1.0 + math.exp(-v))  # noqa: E731
    lo, hi = bounds
    T = x.size
    p, m1, m2 = np.zeros(4), np.zeros(4), np.zeros(4)
    losses = []
    for it in range(1, iters + 1):
        noise, c, ell = lo + (hi - lo) * sg(p[0]), sp(p[2]), sp(p[3])
        r = y - p[1]
        val, glog = orc.log_marginal_likelihood(x, r, (c, ell, noise), faithful=False, eval_gradient=True)
        K = orc.gram_rbf(x, None, c, ell, noise)
        sa = float(np.sum(np.linalg.solve(K, r)))
        losses.append(-val / T)
        g = np.array([glog[2] / noise * (hi - lo) * sg(p[0]) * (1 - sg(p[0])), sa, glog[0] / c * sg(p[2]), glog[1] / ell * sg(p[3])]) * (-1.0 / T)
        m1 = 0.9 * m1 + 0.1 * g
        m2 = 0.999 * m2 + 0.001 * g * g
        p = p - lr * (m1 / (1 - 0.9 ** it)) / (np.sqrt(m2 / (1 - 0.999 ** it)) + 1e-8)
    theta = (sp(p[2]), sp(p[3]), lo + (hi - lo) * sg(p[0]), p[1])
    return (theta, np.array(losses), p) if return_raw else (theta, np.array(losses))
