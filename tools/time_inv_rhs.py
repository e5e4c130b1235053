"""Latency of ops.chol_inverse_rhs (the two inversions of the LDS member step) at small batches:
    python tools/time_inv_rhs.py [T]        (HGP_INV_COOP_MAX_WG=0 in the environment selects the one-wave-per-panel kernel)
Prints us per call for b = 2, 4, 8, 20, 40, 80 and a checksum of the outputs (the two kernels must agree bit for bit)."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hdpgpc_amd import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 90
rng = np.random.default_rng(7)
dev = "cuda"
for b in (2, 4, 8, 20, 40, 80):
    G = rng.standard_normal((b, T, T))
    A = torch.as_tensor(G @ G.transpose(0, 2, 1) / T + 0.5 * np.eye(T), device=dev)
    B = torch.as_tensor(rng.standard_normal((b, T, T)), device=dev)
    Z, Y = torch.zeros_like(A), torch.zeros_like(A)
    info = torch.zeros(b, dtype=torch.int32, device=dev)
    on = torch.tensor([1, 1, 0, 0] * (b // 4) + [1] * (b % 4), dtype=torch.int32, device=dev)
    for _ in range(5):
        ops.chol_inverse_rhs(A, Z, B, Y, info, rhs_on=on, add_diag=1e-8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        ops.chol_inverse_rhs(A, Z, B, Y, info, rhs_on=on, add_diag=1e-8)
    e1.record()
    torch.cuda.synchronize()
    keep = on.bool().cpu().numpy()
    h = hashlib.sha1(Z.cpu().numpy().tobytes() + Y.cpu().numpy()[keep].tobytes() + info.cpu().numpy().tobytes()).hexdigest()[:12]
    err = float(torch.linalg.norm(Z @ A @ Z.transpose(1, 2) - torch.eye(T, device=dev, dtype=torch.float64)) / np.sqrt(b * T))
    print(f"T={T} b={b:3d}: {1e3 * e0.elapsed_time(e1) / reps:7.2f} us per call (back to back)   sha1 {h}   |Z A Z^T - I| {err:.2e}  info {int(info.abs().max())}")
