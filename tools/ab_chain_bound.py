"""The kernels whose critical path is the diag16 chain itself (few waves per SIMD): shared-grid member path, a8 / a9 at T = 90, the
member step's inversion - for A/B runs of diag16_acc variants:   HGP_LIB=... python tools/ab_chain_bound.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from hdpgpc_amd import ops  # noqa: E402

for rep in range(2):
    s = bench.secondary_shared_grid("cuda", ops)
    m = bench.secondary_matrix_terms("cuda", ops)
    print(f"shared grid T=90: {s['kernel_ms']:.4f} ms ({s['value'] / 1e6:.2f} M evals/s)   a8 {m['a8']['kernel_ms']:.4f} ms   a9 {m['a9']['kernel_ms']:.4f} ms")
