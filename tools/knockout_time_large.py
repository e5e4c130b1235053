"""Timing of the cooperative pair kernel (T = 256, 1024 x 16) for experiment builds; HGP_LIB selects the build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdpgpc_amd import ops, _ffi
import synthetic_workload as synth
dev = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")
N, K, T = 1024, 16, 256
b = synth.synthetic_batch(N, K, T, seed=20260703)
plan = ops.PairsPlan(T, T, b["theta"])
xb, mean, Sig, x, y = dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]), dev(b["x"]), dev(b["y"])
plan.update(xb, mean, Sig)
plan.loglik(x, y, want_logdet=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    q, _, info = plan.loglik(x, y, want_logdet=False)
e1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(_ffi.LIB_PATH)}: k_pairs_cooph<16> {e0.elapsed_time(e1) / 3:.3f} ms per {N * K} pairs; checksum {float(q.sum()):.12e}")
