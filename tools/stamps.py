"""Diagnostic: where does a (segment, cluster) pair spend its cycles?  Uses the HGP_STAMPS build (make stamps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HGP_LIB", os.path.join(ROOT, "hdpgpc_amd", "lib", "ab", "libhgp_stamps.so"))
import hdpgpc_amd._ffi as ffi
import numpy as np, torch
from hdpgpc_amd import ops
from oracle import hdpgpc_oracle as orc
ffi.lib.hgp_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")
for (N, K, T) in [(2048, 8, 128), (2048, 8, 96)]:
    b = orc.synthetic_batch(N, K, T)
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    x, y = dev(b["x"]), dev(b["y"])
    plan.loglik(x, y); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    ffi.lib.hgp_debug_stamps(buf)           # reset
    plan.loglik(x, y); torch.cuda.synchronize()
    ffi.lib.hgp_debug_stamps(buf)
    v = np.array(list(buf), dtype=np.float64) / (N * K)
    names = ["d=y-E^T a'", "sweep1 (M'E)", "K** init", "sweep2 (E^T B)", "regularise", "factor+solve", "prologue (x4 waves / K pairs)", "  of which diag16"]
    print(f"T={T}: cycles per pair (s_memtime ticks, 100 MHz-independent shader clock)")
    for nme, val in zip(names, v):
        print(f"   {nme:18s} {val:10.0f}")
    print(f"   total              {v[:6].sum():10.0f}")
    print("   prologue, cumulative per WORKGROUP (cycles): loads+barrier %.0f | + E build, K** tests, barrier %.0f | + band check %.0f | + K** cache = all %.0f"
          % tuple(v[[8, 9, 10, 6]] * K / 4))

# cooperative kernel (one workgroup per pair), last wave's view
for (N, K, T) in [(256, 16, 256), (256, 16, 192)]:
    b = orc.synthetic_batch(N, K, T)
    plan = ops.PairsPlan(T, T, b["theta"]).update(dev(b["xb"]), dev(b["mean"]), dev(b["Sigma"]))
    x, y = dev(b["x"]), dev(b["y"])
    plan.loglik(x, y); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    ffi.lib.hgp_debug_stamps(buf)
    plan.loglik(x, y); torch.cuda.synchronize()
    ffi.lib.hgp_debug_stamps(buf)
    v = np.array(list(buf), dtype=np.float64) / (N * K)
    names = ["setup + E build", "d=y-E^T a'", "K** init", "sweep1 (M'E)", "sweep2 (E^T B)", "regularise", "factor+solve", "-"]
    print(f"coop T={T}: cycles per pair (wave 3 of the workgroup)")
    for nme, val in zip(names, v):
        print(f"   {nme:18s} {val:10.0f}")
    print(f"   total              {v[:7].sum():10.0f}")
    print("   factor split (same wave): diag16+rhs %.0f | wait W %.0f | panel %.0f | wait row+WAR %.0f | trailing MFMA %.0f" % tuple(v[8:13]))
