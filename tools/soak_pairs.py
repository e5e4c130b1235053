"""Soak test of the per-pair path against the oracle: random sizes, cluster counts, length-scales and grid shapes (sorted jitter,
shifted, reversed, permuted, clustered, partially out of range).  Not part of the test-suite (minutes); run on the GPU box:
    python tools/soak_pairs.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdpgpc_amd import ops
from oracle import hdpgpc_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
worst, bad = 0.0, 0
for case in range(n_cases):
    T = int(rng.choice([8, 17, 32, 45, 64, 90, 96, 100, 128, 144, 200, 256]))
    Ts = T if rng.random() < 0.6 else int(np.clip(T + rng.integers(-T // 3, T // 3 + 1), 4, 256))
    K = int(rng.integers(1, 9))
    N = int(rng.integers(1, 7))
    b = orc.synthetic_batch(N, K, T, seed=int(rng.integers(1 << 30)))
    x = np.empty((N, Ts)); y = rng.normal(size=(N, Ts)) * 30.0
    for n in range(N):
        base = np.linspace(b["xb"][0], b["xb"][-1], Ts) + rng.uniform(-0.3, 0.3, Ts)
        mode = rng.integers(0, 6)
        if mode == 1: base = base + rng.uniform(-60.0, 60.0)
        elif mode == 2: base = base[::-1].copy()
        elif mode == 3: base = rng.permutation(base)
        elif mode == 4: base = np.sort(rng.uniform(b["xb"][0] + 5.0, b["xb"][0] + 25.0, Ts))
        elif mode == 5: base = np.concatenate((base[Ts // 2:], base[:Ts // 2]))
        x[n] = base
    plan = ops.PairsPlan(T, Ts, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    quad, logdet, info = plan.loglik(dev(x), dev(y))
    _, q_ref, ld_ref = orc.loglik_pairs(x, y, b["xb"], b["theta"], b["mean"], b["Sigma"])
    ok = int(info.abs().max()) == 0
    e = max(float(np.max(np.abs(quad.cpu().numpy() - q_ref) / np.abs(q_ref))), float(np.max(np.abs(logdet.cpu().numpy() - ld_ref) / np.maximum(np.abs(ld_ref), 1.0))))
    worst = max(worst, e)
    if not ok or not e < 1e-8:
        bad += 1
        print(f"case {case}: T={T} Ts={Ts} K={K} N={N} info_ok={ok} err={e:.2e}", flush=True)
print(f"{n_cases} cases, worst relative error {worst:.2e}, failures {bad}")
