"""chain_batch.run vs sequential full_pass_weighted: bit-identical?  and timing."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from offline_trace import build_model
from hdpgpc_amd import chain_batch
g = np.load("/root/repo/tests/golden/include_batch_r100.npz")
y = np.load("/root/repo/tests/golden/mitbih100_lead0.npz")["y"]
sw, x_trains, data = build_model(g, y)
x, yy = sw.cond_to_torch(x_trains), sw.cond_to_torch(data)
sw.redefine_default(x, yy)
N = yy.shape[0]
rng = np.random.default_rng(0)
splits = [2126, 145, 2083, 188, 2010, 261, 1414, 857, 1613, 658]
resps = []
for k in splits:
    r = torch.zeros(N); r[torch.as_tensor(np.sort(rng.choice(N, k, replace=False)))] = 1.0
    resps.append(r)
def seq():
    outs = []
    for r in resps:
        gp = sw.create_gp_default()
        outs.append(gp.full_pass_weighted(x, yy[:, :, [0]], r))
    return outs
def bat():
    jobs = [chain_batch.Job(sw.create_gp_default(), x, yy[:, :, [0]], r) for r in resps]
    return chain_batch.run(jobs), jobs
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = seq(); torch.cuda.synchronize(); t1 = time.perf_counter()
    (b, jobs) = bat(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"sequential {t1 - t0:.3f} s   batched {t2 - t1:.3f} s")
for (qa, la), (qb, lb) in zip(a, b):
    print(bool(torch.equal(qa, qb)), bool(torch.equal(la, lb)), float((qa - qb).abs().max()), float((la - lb).abs().max()))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
(b, jobs) = bat(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
