"""Where does GPI_HDP.include_batch on MIT-BIH record 100 spend its wall-clock?  (host profile; run on the GPU box)
    python tools/time_offline.py [n_beats]"""
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
from offline_trace import run_model  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "include_batch_r100.npz"))
y = np.load(os.path.join(ROOT, "tests", "golden", "mitbih100_lead0.npz"))["y"]
if len(sys.argv) > 1:
    y = y[:int(sys.argv[1])]
run_model(g, y)                      # warm-up (library load, first graph captures)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
sw = run_model(g, y)
torch.cuda.synchronize()
pr.disable()
print(f"include_batch on {y.shape[0]} beats: {time.perf_counter() - t0:.3f} s, final counts {[len(m.indexes) for m in sw.gpmodels[0]]}")
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
