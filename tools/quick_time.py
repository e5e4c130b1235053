"""Scratch timing of the main kernels on one GPU (not the bench contract - see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdpgpc_amd import ops
from oracle import hdpgpc_oracle as orc

def dev(a): return torch.as_tensor(a, dtype=torch.float64, device="cuda")

def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n

for (N,K,T) in [(2048,8,128),(2048,8,96),(256,8,64)]:
    b = orc.synthetic_batch(N,K,T)
    plan = ops.PairsPlan(T,T,b["theta"])
    xb,mean,Sig,x,y = dev(b["xb"]),dev(b["mean"]),dev(b["Sigma"]),dev(b["x"]),dev(b["y"])
    tu = timeit(lambda: plan.update(xb,mean,Sig))
    tp = timeit(lambda: plan.loglik(x,y))
    print(f"pairs N={N} K={K} T={T}: update {tu*1e3:.3f} ms, pairs {tp*1e3:.3f} ms -> {N*K/(tu+tp):.3e} evals/s", flush=True)

for (S,T) in [(4096,128),(4096,90),(4096,64)]:
    rng=np.random.default_rng(0)
    Q=rng.normal(size=(64,T,T)); A=Q@Q.transpose(0,2,1)/T+np.eye(T)
    Sig=dev(np.tile(A,(S//64,1,1))); Y=dev(rng.normal(size=(S,T))); mean=dev(rng.normal(size=(S,T)))
    items=ops.build_items(list(range(S)),[0.0]*S,[1]*S)
    items=[dev(i).to(torch.int32) if i.dtype!=np.float64 else dev(i) for i in items]
    t=timeit(lambda: ops.score_groups(Y,mean,Sig,*items))
    print(f"score_groups S={S} T={T}: {t*1e3:.3f} ms -> {S/t:.3e} evals/s, {S*T*T*8/t/1e9:.1f} GB/s", flush=True)

for (N,K,T) in [(256,16,256),(1024,16,256),(256,16,192)]:
    b = orc.synthetic_batch(N,K,T)
    plan = ops.PairsPlan(T,T,b["theta"])
    xb,mean,Sig,x,y = dev(b["xb"]),dev(b["mean"]),dev(b["Sigma"]),dev(b["x"]),dev(b["y"])
    tu = timeit(lambda: plan.update(xb,mean,Sig), n=2, w=1)
    tp = timeit(lambda: plan.loglik(x,y), n=2, w=1)
    print(f"pairs (cooperative, large T) N={N} K={K} T={T}: update {tu*1e3:.3f} ms, pairs {tp*1e3:.3f} ms -> {N*K/(tu+tp):.3e} evals/s", flush=True)

for (S,T) in [(4096,128),(4096,90),(4096,64),(16384,90)]:
    rng=np.random.default_rng(0)
    Q=rng.normal(size=(64,T,T)); A=Q@Q.transpose(0,2,1)/T+np.eye(T)
    Sig=dev(np.tile(A,(S//64,1,1))); Y=dev(rng.normal(size=(S,T))); mean=dev(rng.normal(size=(S,T)))
    sm=torch.arange(S,dtype=torch.int32,device="cuda")
    t=timeit(lambda: ops.score_each(Y,mean,Sig,sm))
    print(f"score_each   S={S} T={T}: {t*1e3:.3f} ms -> {S/t:.3e} evals/s, {S*T*T*8/t/1e9:.1f} GB/s", flush=True)
    Ss=(0.5*(Sig+Sig.transpose(1,2))).contiguous()
    t=timeit(lambda: ops.score_each(Y,mean,Ss,sm,symmetric=True))
    print(f"score_each(sym) S={S} T={T}: {t*1e3:.3f} ms -> {S/t:.3e} evals/s, {S*T*T*8/t/1e9:.1f} GB/s (algorithmic bytes)", flush=True)

for (b, T) in [(16, 256), (1024, 256), (4096, 128), (4096, 90)]:
    rng = np.random.default_rng(0)
    Q = rng.normal(size=(8, T, T)); A = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    L = dev(np.tile(np.linalg.cholesky(A), (b // 8, 1, 1))); v = dev(rng.normal(size=(b, T)))
    al = dev(np.full(b, 0.98)); be = dev(np.full(b, 0.5))
    info = torch.zeros(b, dtype=torch.int32, device="cuda")
    from hdpgpc_amd import _ffi
    Lw = L.clone()
    f = lambda: _ffi.check(_ffi.lib.hgp_chol_rank1_f64(ops._ptr(Lw), ops._ptr(v), ops._ptr(al), ops._ptr(be), T, b, ops._ptr(info), ops._stream()), "r1")
    t = timeit(f, n=3, w=1)
    print(f"chol_rank1 b={b} T={T}: {t*1e3:.3f} ms -> {b/t:.3e} updates/s, {b*T*T*8/t/1e9:.1f} GB/s (algorithmic: lower triangle in and out)", flush=True)

# a8 / a9: matrix-valued likelihood terms, batched
for (b, T) in [(2272, 90), (256, 90), (16, 90), (1024, 128)]:
    rng = np.random.default_rng(1)
    Q = rng.normal(size=(8, T, T)); G = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    Gam = dev(np.tile(G, (b // 8 + 1, 1, 1))[:b]); A = dev(rng.normal(size=(b, T, T)) * 0.1); cp = dev(np.tile(G, (b // 8 + 1, 1, 1))[:b])
    fc, fp = dev(rng.normal(size=(b, T))), dev(rng.normal(size=(b, T)))
    t = timeit(lambda: ops.lat_error(fc, fp, A, Gam, cp), n=3, w=1)
    print(f"lat_error (a8)   b={b} T={T}: {t*1e3:.3f} ms -> {b/t:.3e} evals/s", flush=True)
    M = dev(rng.normal(size=(b, T, T))); m0 = dev(np.eye(T)); sc = dev(G[0])
    t = timeit(lambda: ops.mniw_loglik(M, Gam, m0, None, sc), n=3, w=1)
    print(f"mniw_loglik (a9) b={b} T={T}: {t*1e3:.3f} ms -> {b/t:.3e} evals/s", flush=True)

# switching-variable messages (forward + backward + pair responsibilities), sequential in N
for (N, K) in [(2272, 3), (2272, 9), (32768, 16), (32768, 64)]:
    rng = np.random.default_rng(2)
    q = dev(rng.normal(size=(N, K)) * 30 - 100); lt = dev(np.log(rng.dirichlet(np.ones(K), size=K))); lp = dev(np.log(rng.dirichlet(np.ones(K))))
    t = timeit(lambda: ops.hmm_messages(q, lp, lt), n=3, w=1)
    print(f"hmm_messages N={N} K={K}: {t*1e3:.3f} ms ({t/N*1e9:.0f} ns per step)", flush=True)
