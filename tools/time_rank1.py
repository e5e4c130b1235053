"""k_chol_rank1 (rank-1 Cholesky update, BASELINE configs[4]'s kernel): ms per launch and HBM fraction at b factors of size T, and a
checksum of the updated factors.   python tools/time_rank1.py [b] [T]      (HGP_RANK1_DIRECT=1: the uncoalesced row-per-thread form)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from hdpgpc_amd import ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
r = bench.secondary_rank1("cuda", ops, b=b, T=T, reps=20)
g = torch.Generator().manual_seed(5)
Q = torch.randn(4, T, T, generator=g, dtype=torch.float64)
L = torch.linalg.cholesky(Q @ Q.transpose(1, 2) / T + torch.eye(T, dtype=torch.float64)).cuda().contiguous()
v = torch.randn(4, T, generator=g, dtype=torch.float64).cuda().contiguous()
Ln, info = ops.chol_rank1(L, v, alpha=[0.9, 1.0, 0.98, 0.7], beta=[0.5, 1.0, 0.1, 2.0])
ref = torch.linalg.cholesky(torch.tensor([0.9, 1.0, 0.98, 0.7], device="cuda", dtype=torch.float64)[:, None, None] * (L @ L.transpose(1, 2))
                            + torch.tensor([0.5, 1.0, 0.1, 2.0], device="cuda", dtype=torch.float64)[:, None, None] * v[:, :, None] * v[:, None, :])
err = float((Ln - ref).abs().max())
print(f"b={b} T={T}: {r['kernel_ms']:.4f} ms per launch, {r['roofline']['achieved']:.0f} GB/s = {r['roofline']['frac']:.3f} of HBM; "
      f"sha1 {hashlib.sha1(Ln.cpu().numpy().tobytes()).hexdigest()[:12]}  |L' - chol| {err:.2e}  info {int(info.abs().max())}")
