"""In-situ knock-out timing of k_pairs<8> (diagnostic builds made with -DHGP_EXP_*; results of those builds are wrong by
construction, only the time matters).  HGP_LIB selects the build."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdpgpc_amd import ops, _ffi
import synthetic_workload as synth
dev = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")
N, K, T = 2048, 8, 128
b = synth.synthetic_batch(N, K, T, seed=20260703)
plan = ops.PairsPlan(T, T, b["theta"])
xb, mean, Sig, x, y = dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]), dev(b["x"]), dev(b["y"])
plan.update(xb, mean, Sig)
for _ in range(3):
    plan.loglik(x, y, want_logdet=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan.loglik(x, y, want_logdet=False)
e1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(_ffi.LIB_PATH)}: k_pairs<8> {e0.elapsed_time(e1) / 20:.4f} ms per 16384 pairs")
