"""Time GPI_model.full_pass_weighted on MIT-BIH record 100, lead 0 (data fixture), one cluster over the
first N beats - the call the reference spends 91 % of its offline wall-clock in.  Reference on this container's 8 vCPU:
10.3 ms per member at N = 600 (tests/golden/make_golden.py: build_model)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdpgpc_amd.GPI import RBFWhiteKernel
from hdpgpc_amd.GPI_model import GPI_model

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mitbih100_lead0.npz"))["y"]
for N in (int(a) for a in (sys.argv[1:] or ["600"])):
    y = d[:N]
    T = y.shape[1]
    samples_, samples__ = y[:N - 2].T, y[1:N - 1].T           # get_data.compute_estimators_LDS (host-side, 30 lines)
    std = np.mean(np.diag((samples_ - samples_.mean(1, keepdims=True)) @ (samples_ - samples_.mean(1, keepdims=True)).T) / (N - 2))
    std_dif = np.mean(np.diag((samples__ - samples_) @ (samples__ - samples_).T) / (N - 2))
    if std > 1:
        std, std_dif = std * 0.02, std_dif * 0.02
    std_dif = min(max(std, std_dif), std * 1.5)
    xb = np.arange(float(T))
    m = GPI_model(RBFWhiteKernel(300.0, 3.0, std * 1e-5), xb[:, None], annealing=True, bayesian=True, free_deg_MNIV=5)
    cond = m.GPR_dynamic(std_dif, std)
    m.initial_conditions(ini_A=cond[0], ini_Gamma=cond[1], ini_C=cond[2], ini_Sigma=cond[3])
    m.fixed_theta = (341.0, 1.2, min(4.66, std * 2.0))
    xs = np.repeat(xb[None, :, None], N, axis=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q, ql = m.full_pass_weighted(xs, y[:, :, None], np.ones(N))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"full_pass_weighted N={N} T={T}: {dt:.2f} s ({dt / N * 1e3:.3f} ms per member), q[:3]={q[:3].cpu().numpy()}", flush=True)
    # the consumers on the finished lists (the reference: 5.3 k evals/s and 2.3 k evals/s on 8 vCPUs)
    xs_d, y_d = m.cond_to_torch(xs), m.cond_to_torch(y[:, :, None])
    for name, fn in (("compute_sq_err_all", lambda: m.compute_sq_err_all(xs_d, y_d)), ("compute_q_lat_all", lambda: m.compute_q_lat_all(xs_d))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"   {name}: {dt * 1e3:.2f} ms for {N} evals -> {N / dt:.3e} evals/s (host logic + kernels)", flush=True)
