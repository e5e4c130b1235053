"""Timing of the fused a8 / a9 kernels (hgp_matlik.hip) on one GPU: evals/s and algorithmic HBM bytes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hdpgpc_amd import ops


def dev(a):
    return torch.as_tensor(a, dtype=torch.float64, device="cuda")


def timeit(fn, n=5, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for (b, T) in [(16384, 90), (2272, 90), (256, 90), (16, 90), (4096, 128), (8192, 64), (256, 256)]:
    rng = np.random.default_rng(1)
    Q = rng.normal(size=(8, T, T))
    G = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    Gam = dev(np.tile(G, (b // 8 + 1, 1, 1))[:b])
    A = dev(rng.normal(size=(b, T, T)) * 0.1)
    cp = dev(np.tile(G, (b // 8 + 1, 1, 1))[:b])
    fc, fp = dev(rng.normal(size=(b, T))), dev(rng.normal(size=(b, T)))
    t = timeit(lambda: ops.lat_error(fc, fp, A, Gam, cp), n=3, w=1)
    print(f"lat_error (a8)   b={b} T={T}: {t*1e3:.3f} ms -> {b/t:.3e} evals/s, {b*(3*T*T+2*T)*8/t/1e9:.0f} GB/s algorithmic", flush=True)
    M = dev(rng.normal(size=(b, T, T)))
    m0 = dev(np.eye(T))
    sc = dev(1.7 * np.eye(T))          # the hot path's prior scale: sigma I (GPI_model.py:481-484)
    t = timeit(lambda: ops.mniw_loglik(M, Gam, m0, None, sc, scale_is_diagonal=True), n=3, w=1)
    print(f"mniw_loglik (a9) b={b} T={T}: {t*1e3:.3f} ms -> {b/t:.3e} evals/s, {b*2*T*T*8/t/1e9:.0f} GB/s algorithmic (diagonal prior scale)", flush=True)
    sc = dev(G[0])
    t = timeit(lambda: ops.mniw_loglik(M, Gam, m0, None, sc, scale_is_diagonal=False), n=3, w=1)
    print(f"mniw_loglik (a9) b={b} T={T}: {t*1e3:.3f} ms -> {b/t:.3e} evals/s (dense prior scale)", flush=True)
