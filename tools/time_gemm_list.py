"""One level of the member step as a product list, for a kernel trace:  FORM=walk|mapped python tools/time_gemm_list.py [T] [items]
(run under rocprofv3 --kernel-trace --stats: the average duration of k_gemm_list is the number; back-to-back eager launches only show
the launch rate).  walk = every wave finds its item by walking the device-resident list, mapped = per-tile map."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hdpgpc_amd import _ffi, ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 90
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
form = os.environ.get("FORM", "walk")
g = torch.Generator().manual_seed(3)
mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).cuda()      # noqa: E731
A, B, D = [mk(T, T) for _ in range(n)], [mk(T, T) for _ in range(n)], [mk(T, T) for _ in range(n)]
C = [torch.zeros(T, T, dtype=torch.float64, device="cuda") for _ in range(n)]
gl = ops.GemmList("cuda")
for i in range(n):
    gl.add(A[i], B[i], C[i], D=D[i] if i % 2 == 0 else None, transA=bool(i & 1), transB=bool(i & 2), alpha=0.5 + i, beta=-1.0)
gl.finalize()
m = np.concatenate([(i << 16) | np.arange(t, dtype=np.uint32) for i, t in enumerate(gl._item_tiles)]).astype(np.uint32)
tmap = torch.from_numpy(m.view(np.int32)).cuda()
for _ in range(1000):
    if form == "walk":
        _ffi.check(_ffi.lib.hgp_gemm_list_f64(ops._ptr(gl._dev), n, gl.tiles, ops._stream()), "walk")
    else:
        _ffi.check(_ffi.lib.hgp_gemm_list_mapped_f64(ops._ptr(gl._dev), n, ops._ptr(tmap), gl.tiles, ops._stream()), "mapped")
torch.cuda.synchronize()
print(form, "done")
