"""a8 / a9 at T = 256, b = 256 exactly as bench.py measures them (for A/B runs with HGP_LIB / HGP_MATLIK_COOP4)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from hdpgpc_amd import ops  # noqa: E402

for rep in range(2):
    m = bench.secondary_matrix_terms("cuda", ops, b=256, T=256, reps=10)
    print(f"T=256 b=256: a8 {m['a8']['kernel_ms']:.4f} ms   a9 {m['a9']['kernel_ms']:.4f} ms")
