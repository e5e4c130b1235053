"""SURVEY.md 8b Face 1 / BASELINE configs[4]: GPI_HDP.include_sample - the online variational step - on the HIP kernels, driven as
hdpgpc/tests/test_online.py:41-83 drives the reference, against the traces of the reference's own runs
(tests/golden/include_sample_*.npz, make_golden.py online90 / "online256 trace"): after EVERY beat the chosen cluster, the number
of clusters, the hard assignments of the whole history and the cluster sizes are identical, the score matrix within 1e-8."""
import time

import pytest
import torch

from conftest import golden
from online_trace import compare_online, run_online

pytestmark = pytest.mark.gpu


def test_include_sample_r102_t90():
    g = golden("include_sample_r102_n40.npz")
    torch.cuda.synchronize()
    t0 = time.time()
    _, tr = run_online(g)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_online(g, tr, 1e-8)
    print(f"include_sample, 40 beats T=90: {wall:.2f} s (reference {float(g['secs'].sum()):.1f} s), worst {worst:.2e}")


def test_include_sample_r102_t256():
    """configs[4]: beats resampled to T = 256 (linear interpolation), same loop."""
    g = golden("include_sample_r102_t256_n24.npz")
    torch.cuda.synchronize()
    t0 = time.time()
    _, tr = run_online(g)
    torch.cuda.synchronize()
    wall = time.time() - t0
    worst = compare_online(g, tr, 1e-8)
    print(f"include_sample, 24 beats T=256: {wall:.2f} s (reference {float(g['secs'].sum()):.1f} s), worst {worst:.2e}")
