"""Where does GPI_HDP.include_sample (BASELINE configs[4]: online path, T = 256) spend its wall-clock?  Run on the GPU box:
    python tools/time_online.py [fixture] [--profile]
Prints ms per beat of the second (warm) run, beat by beat, and with --profile the host-side cProfile table."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from online_trace import run_online  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "include_sample_r102_t256_n24.npz"
g = np.load(os.path.join(ROOT, "tests", "golden", name))
run_online(g)                        # warm-up (library load, plan caches)
torch.cuda.synchronize()
pr = cProfile.Profile() if "--profile" in sys.argv else None
if "--nogc" in sys.argv:
    import gc
    gc.disable()
if os.environ.get("HGP_ONLINE_TIMING"):
    from hdpgpc_amd import online_loop
    online_loop.TIMING.clear()
    online_loop._t_last[0] = time.perf_counter()
t0 = time.perf_counter()
if pr:
    pr.enable()
sw, tr = run_online(g)
torch.cuda.synchronize()
if pr:
    pr.disable()
wall = time.perf_counter() - t0
n = len(tr)
print(f"include_sample on {n} beats of {name} (T = {g['y'].shape[1]}): {wall:.3f} s = {1e3 * wall / n:.2f} ms per beat; "
      f"clusters at the end {sw.M}; reference {float(g['secs'].sum()):.1f} s")
if os.environ.get("HGP_ONLINE_TIMING"):
    from hdpgpc_amd import online_loop
    print({k: round(1e3 * v / n, 2) for k, v in online_loop.TIMING.items()}, "ms per beat")
if pr:
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
