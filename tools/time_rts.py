"""k_rts_chain (the sequential part of the RTS smoother, one workgroup walking the chain): us per step and a checksum.
    python tools/time_rts.py [n_steps] [T]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hdpgpc_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
T = int(sys.argv[2]) if len(sys.argv) > 2 else 90
rng = np.random.default_rng(5)
d = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")  # noqa: E731
J = d(rng.normal(size=(n, T, T)) * (0.5 / np.sqrt(T)))
Q = rng.normal(size=(n, T, T)) / np.sqrt(T)
P = d(Q @ Q.transpose(0, 2, 1) + np.eye(T))
C0 = d(0.8 * (Q @ Q.transpose(0, 2, 1)) + 0.5 * np.eye(T))
AM, M0 = d(rng.normal(size=(n, T))), d(rng.normal(size=(n, T)))
for rep in range(3):
    M, Cv = M0.clone(), C0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.rts_chain(J, P, AM, M, Cv)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
# reference of the recursion on the host (float64): m_t += J_t (m_{t+1} - A m_t), C_t += J_t (C_{t+1} - P_t) J_t^T
Mh, Ch = M0.cpu().numpy().copy(), C0.cpu().numpy().copy()
Jh, Ph, AMh = J.cpu().numpy(), P.cpu().numpy(), AM.cpu().numpy()
for t in range(n - 2, max(n - 40, -1), -1):
    Mh[t] = Mh[t] + Jh[t] @ (Mh[t + 1] - AMh[t])
    Ch[t] = Ch[t] + Jh[t] @ (Ch[t + 1] - Ph[t]) @ Jh[t].T
lo = max(n - 39, 0)
err = max(float(np.abs(M.cpu().numpy()[lo:] - Mh[lo:]).max()), float(np.abs(Cv.cpu().numpy()[lo:] - Ch[lo:]).max()))
print(f"n={n} T={T}: {ms:.3f} ms = {1e3 * ms / (n - 1):.2f} us per step; sha1 {hashlib.sha1(Cv.cpu().numpy().tobytes() + M.cpu().numpy().tobytes()).hexdigest()[:12]}; "
      f"|err| vs host recursion (last 39 steps) {err:.2e}")
