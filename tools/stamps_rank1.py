"""Diagnostic (HGP_STAMPS build): cycles of workgroup 0 of k_chol_rank1_pipe per wave and phase.  python tools/stamps_rank1.py [b] [T]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HGP_LIB", os.path.join(ROOT, "hdpgpc_amd", "lib", "ab", "libhgp_stamps.so"))
import numpy as np, torch
import hdpgpc_amd._ffi as ffi
from hdpgpc_amd import ops
ffi.lib.hgp_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator().manual_seed(5)
Q = torch.randn(b, T, T, generator=g, dtype=torch.float64)
L = torch.linalg.cholesky(Q @ Q.transpose(1, 2) / T + torch.eye(T, dtype=torch.float64)).cuda().contiguous()
v = torch.randn(b, T, generator=g, dtype=torch.float64).cuda().contiguous()
buf = (ctypes.c_ulonglong * 16)()
ffi.lib.hgp_debug_stamps(buf)
ops.chol_rank1(L, v); torch.cuda.synchronize()
ffi.lib.hgp_debug_stamps(buf)          # reset after the warm-up
reps = 10
for _ in range(reps):
    ops.chol_rank1(L, v)
torch.cuda.synchronize()
ffi.lib.hgp_debug_stamps(buf)
x = np.array(list(buf), dtype=np.float64).reshape(4, 4) / reps
print(f"b={b} T={T}: cycles of workgroup 0 per launch (s_memtime), {T // 16 + 1} iterations")
for w in range(4):
    print(f"  wave {w}: part A {x[w,0]:9.0f} | chain {x[w,1]:9.0f} | rotations to LDS {x[w,2]:8.0f} | barrier wait {x[w,3]:9.0f} | total {x[w].sum():9.0f}")
