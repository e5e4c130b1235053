"""Synthetic beat batches of SURVEY.md 8(d) - the workload generator shared by bench.py, the tests and the oracle.
Plain NumPy, no dependency on the oracle or on the HIP library (bench.py's timed path must not import oracle/)."""
import numpy as np


def synthetic_batch(N, K, T, seed=20260703, irregular=True):
    """Synthetic beat batch of SURVEY.md 8(d): returns dict of fp64 arrays."""
    rng = np.random.default_rng(seed)
    xb = np.arange(T, dtype=np.float64)
    t = xb[None, :]
    mu = np.zeros((K, T))
    for k in range(K):
        a = rng.uniform(50, 300)
        for _ in range(3):
            w = rng.uniform(-1, 1)
            cc = rng.uniform(0.2 * T, 0.8 * T)
            s = rng.uniform(0.02 * T, 0.1 * T)
            mu[k] += a * w * np.exp(-0.5 * ((t[0] - cc) / s) ** 2)
    z = rng.integers(0, K, size=N)
    y = mu[z] + rng.normal(0.0, 3.0, size=(N, T))
    theta = np.stack([np.array([341.0 * (1 + 0.05 * k), 1.2, 0.9]) for k in range(K)])
    Sigma = np.zeros((K, T, T))
    for k in range(K):
        v = rng.normal(size=T)
        s = rng.uniform(0.5, 5.0)
        Sigma[k] = s * (np.eye(T) + 0.1 * np.outer(v, v))
    if irregular:
        x = xb[None, :] + rng.uniform(-0.3, 0.3, size=(N, T))
    else:
        x = np.repeat(xb[None, :], N, axis=0)
    return dict(xb=xb, x=x, y=y, theta=theta, mean=mu, Sigma=Sigma, labels=z)
