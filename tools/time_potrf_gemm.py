import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from hdpgpc_amd import ops
def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for b, T in [(256, 256), (1024, 256), (256, 192)]:
    rng = np.random.default_rng(1)
    Q = rng.normal(size=(8, T, T)); G = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    A = torch.as_tensor(np.tile(G, (b // 8, 1, 1)), dtype=torch.float64, device="cuda")
    B = torch.as_tensor(rng.normal(size=(b, T, T)), dtype=torch.float64, device="cuda")
    t1 = timeit(lambda: ops.potrf_batched(A, 1e-8, 0.0))
    t2 = timeit(lambda: ops.potrf_batched(A, 1e-8, 0.0, want_inv=True))
    t3 = timeit(lambda: ops.gemm_batched(A, B))
    t4 = timeit(lambda: ops.chol_inverse(A))
    print(f"b={b} T={T}: potrf {t1*1e3:.3f} ms, potrf+inv {t2*1e3:.3f} ms, gemm {t3*1e3:.3f} ms ({2*b*T**3/t3/1e12:.2f} TFLOP/s), chol_inverse {t4*1e3:.3f} ms")
