"""Root-cause probe for the 'wrong rows in L^-1' build of round 1: the 4-wave cooperative factor with 16
right-hand sides (coop_factor<NB, 1>, T > 128) dealt the right-hand-side row tiles to the waves round-robin, while the
wave that OWNS block column I reads R_I at the top of step I without a barrier in between.  For columns whose snake
owner differs from I & 3 that is a cross-wave write/read race, normally hidden by the ~4 k cycles of diag16 in front of
the read.  `make raceprobe` builds the library twice with a delay injected in front of the row update: with the old
dealing (libhgp_race_old.so) and with the owner dealing (libhgp_race_new.so).  Usage:
    HGP_LIB=build/probe/libhgp_race_old.so python tools/probe_coop_race.py     # expected: wrong rows
    HGP_LIB=build/probe/libhgp_race_new.so python tools/probe_coop_race.py     # expected: exact
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdpgpc_amd import ops, _ffi

print("library:", _ffi.LIB_PATH)
rng = np.random.default_rng(0)
for T in (144, 192, 256):
    B = rng.standard_normal((4, T, T))
    A = B @ B.transpose(0, 2, 1) + T * np.eye(T)
    Z, info = ops.chol_inverse(torch.tensor(A, device="cuda"))
    Z = Z.cpu().numpy()
    ref = np.stack([np.linalg.inv(np.linalg.cholesky(a)) for a in A])
    err = np.abs(Z - ref).max(axis=(0, 2)) / np.abs(ref).max()
    bad = np.nonzero(err > 1e-9)[0]
    blocks = sorted(set((bad // 16).tolist()))
    print(f"T={T}: max rel err {err.max():.2e}; rows off by > 1e-9: {bad.size} (row blocks {blocks})")
