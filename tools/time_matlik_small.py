"""a8 / a9 at T = 256 for the SMALL batches of the online step (a few items per call: latency, not throughput):
    python tools/time_matlik_small.py        (HGP_MATLIK_COOP4=1: the four-wave kernels)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hdpgpc_amd import ops  # noqa: E402

T = 256
rng = np.random.default_rng(3)
d = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")  # noqa: E731
for b in (2, 8, 24, 70, 256):
    Q = rng.normal(size=(b, T, T))
    G = d(Q @ Q.transpose(0, 2, 1) / T + np.eye(T))
    A = d(rng.normal(size=(b, T, T)) * 0.1)
    P = d(Q @ Q.transpose(0, 2, 1) / T * 0.3 + 0.1 * np.eye(T))
    fc, fp = d(rng.normal(size=(b, T))), d(rng.normal(size=(b, T)))
    M, mean, scale = d(rng.normal(size=(b, T, T))), d(np.eye(T)), d(0.7 * np.eye(T))
    res = {}
    for name, fn in (("a8", lambda: ops.lat_error(fc, fp, A, G, P)), ("a9", lambda: ops.mniw_loglik(M, G, mean, None, scale, scale_is_diagonal=True))):
        for _ in range(3):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 50 * 1e3, float(out[0].sum()))
    print(f"T=256 b={b:3d}: a8 {res['a8'][0]:7.1f} us per call (sum {res['a8'][1]:.10e})   a9 {res['a9'][0]:7.1f} us (sum {res['a9'][1]:.10e})")
