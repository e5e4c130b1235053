"""A/B timing of the per-pair kernel at one shape:  python tools/time_pairs.py T N K [lib.so]
Events around hgp_loglik_pairs_f64 only; prints ms per launch, evals/s and the fraction of the fp64 MFMA peak on T^3/3 + 3T^2."""
import os
import sys

if len(sys.argv) > 4:
    os.environ["HGP_LIB"] = sys.argv[4]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import synthetic_workload as synth
from hdpgpc_amd import ops

T, N, K = (int(v) for v in sys.argv[1:4])
b = synth.synthetic_batch(N, K, T, seed=20260703)
d = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")  # noqa: E731
plan = ops.PairsPlan(T, T, b["theta"])
xb, mean, Sig, x, y = d(b["xb"]), d(b["mean"]), d(b["Sigma"]), d(b["x"]), d(b["y"])
plan.update(xb, mean, Sig)
for _ in range(3):
    quad, _, info = plan.loglik(x, y, want_logdet=False)
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    quad, _, info = plan.loglik(x, y, want_logdet=False)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = T ** 3 / 3.0 + 3.0 * T ** 2
print(f"T={T} N={N} K={K} lib={os.environ.get('HGP_LIB', 'default')}: {ms:.4f} ms/launch, {N * K / ms * 1e3:.4e} evals/s, "
      f"frac {N * K * fl / (ms * 1e-3) / 78.6e12:.4f}, checksum {float(quad.sum()):.10e}, info {int(info.abs().max())}")
