#!/usr/bin/env python3
"""bench.py - GP log-lik evals/sec of the GP-emission hot path on N MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic segments:
    per-cluster operators from the current cluster state (hgp_pairs_plan_update)
  + the N x K per-(segment, cluster) evaluation: Gram build -> Cholesky factor/solve -> Gaussian score
    (hgp_loglik_pairs_f64), inputs resident in HBM
  + for N_gpus > 1: one RCCL all-gather of the [N, K] score rows back to every rank (the sampler's view).
Workload at 1 GPU = BASELINE.json configs[1]: 2 048 segments x 8 clusters, T = 128, fp64, irregular
segment grids (the general path of pred_dist).  At N > 1 GPUs the workload is configs[3], as north_star states it:
32 768 segments x 16 clusters, T = 256, rows [r N/G, (r+1) N/G) on rank r (STRONG scaling: the batch is fixed,
4 096 rows per GPU at 8 GPUs), cluster state broadcast once from rank 0, scores returned by one all-gather
(hdpgpc_amd.batch.emission_scores - the function the tests exercise is the function timed here).

Prints ONE JSON line (rank 0).  Run: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either launch it under
python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
or just run `python bench.py --gpus N`: without WORLD_SIZE in the environment the process starts that launcher itself as a
CHILD process (before it has made any GPU call - nothing that touched the GPU is ever re-executed), relays rank 0's line
and exits with the launcher's code.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix (= vector) peak, AMD datasheet; SURVEY.md 8(d)
N_SEG, K_CL, T_LEN = 2048, 8, 128


def algorithmic_flops_per_eval(T):
    """SURVEY.md 8(d), primary path: Cholesky + two triangular solves + one exp per Gram entry."""
    return T ** 3 / 3.0 + 3.0 * T ** 2


def cpu_baseline(budget_s=20.0):
    """The oracle (NumPy/SciPy restatement of the reference's per-pair path) on the host cores, bounded sample: one
    single-threaded worker process per core of this process's share (at most 16 - the one-GPU box's share), each
    scoring its own slice of the SAME workload (at most `budget_s` seconds each).  The rate is the sum of the workers'
    rates.  Runs BEFORE the first GPU call of this process (the workers are separate processes either way)."""
    import subprocess
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    per = N_SEG // cores
    t_wall = time.perf_counter()
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_bench", str(w * per), str((w + 1) * per), str(N_SEG),
                               str(K_CL), str(T_LEN), "20260703", str(budget_s)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for w in range(cores)]
    done, rate, cpu_s, slow = 0, 0.0, 0.0, 0.0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s + 120)
            r = json.loads(out.strip().splitlines()[-1])
            done += r["done"]
            rate += r["done"] / r["dt"]
            cpu_s += r["dt"]
            slow = max(slow, r["dt"])
        except Exception:                 # a worker that failed or overran contributes nothing
            p.kill()
    t_wall = time.perf_counter() - t_wall
    return {"value": rate, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{done} evals of the same workload (T={T_LEN}, {K_CL} clusters): {cpu_s:.1f} CPU-seconds of scoring on "
                      f"{cores} worker processes, one BLAS thread each (slowest worker {slow:.2f} s, {t_wall:.1f} s wall with "
                      f"start-up); NumPy/SciPy oracle"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: run torch.distributed.run as a child process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def secondary_offline(dev, rec, survey_s, cfg):
    """GPI_HDP.include_batch on a whole MIT-BIH record, lead 0 (T = 90), driven as hdpgpc/tests/test_offline.py drives the
    reference, kernel hyper-parameters injected as in the golden run (the gpytorch fit is not part of either number).  The beats are
    the committed fixture tests/golden/mitbih<rec>_lead0.npz; the reference's own wall-clock for the same call (traced, this build
    container's 8 vCPUs) is stored in tests/golden/include_batch_r<rec>.npz; SURVEY.md section 6 measured `survey_s` untraced."""
    gdir = os.path.join(ROOT, "tests", "golden")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from offline_trace import run_model
    g = np.load(os.path.join(gdir, f"include_batch_r{rec}.npz"))
    y = np.load(os.path.join(gdir, f"mitbih{rec}_lead0.npz"))["y"]
    times = []
    for _ in range(2):                      # first run includes library / graph warm-up; report both
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sw = run_model(g, y)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    counts = [len(m.indexes) for m in sw.gpmodels[0]]
    ok = counts == [int(c) for c in g["counts_final"]] and bool(np.array_equal(sw.resp_assigned[-1].numpy(), g["resp_assigned"][-1]))
    ref = float(g["wall_s"])
    return {"workload": f"BASELINE {cfg}: GPI_HDP.include_batch, MIT-BIH record {rec} lead 0 ({y.shape[0]} beats, T=90), settings of "
                        "hdpgpc/tests/test_offline.py, theta injected",
            f"offline_r{rec}_s": times[1], "first_run_s": times[0], "reference_cpu_s_survey": survey_s, "reference_cpu_s_traced_run": ref,
            "speedup_vs_survey": survey_s / times[1], "final_counts": counts,
            "assignments_identical_to_reference": ok, "higher_is_better": False}


def secondary_online_t256(dev):
    """BASELINE configs[4]: the online path, GPI_HDP.include_sample beat by beat at T = 256 (beats resampled), driven as
    hdpgpc/tests/test_online.py drives the reference, on the two committed traces of the reference's own runs (24 beats of record
    102; 16 + 16 beats of records 100 / 102 concatenated).  Reports ms per beat of the second (warm) run, the C-ABI calls per
    beat counted live, and whether every beat took the reference's decision.  (The rank-1 update kernel the config names is
    opt-in: it needs annealing off, which no driver sets - see secondary_rank1 and DESIGN.md section 0.)"""
    gdir = os.path.join(ROOT, "tests", "golden")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from online_trace import run_online
    from hdpgpc_amd import _ffi
    out = {"workload": "BASELINE configs[4]: GPI_HDP.include_sample, beats resampled to T=256, settings of hdpgpc/tests/test_online.py, "
                       "theta injected", "higher_is_better": False}
    calls = [0]

    class Counting:                          # every call into libhdpgpc_hip.so made by the host layer
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            fn = getattr(self._lib, name)

            def wrapped(*a):
                calls[0] += 1
                return fn(*a)
            return wrapped

    for tag, name in (("r102_n24", "include_sample_r102_t256_n24.npz"), ("r100_r102_n32", "include_sample_r100_r102_t256_n32.npz")):
        g = np.load(os.path.join(gdir, name))
        run_online(g)                        # warm-up
        torch.cuda.synchronize()
        real = _ffi.lib
        _ffi.lib = Counting(real)
        calls[0] = 0
        try:
            t0 = time.perf_counter()
            sw, tr = run_online(g)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            _ffi.lib = real
        n = len(tr)
        same = all(tr[i][0] == int(g["state"][i]) and tr[i][1] == int(g["M"][i]) and np.array_equal(tr[i][2], g[f"b{i}_labels"])
                   for i in range(n))
        out[tag] = {"beats": n, "ms_per_beat": 1e3 * dt / n, "c_abi_calls_per_beat": calls[0] / n, "clusters_at_end": int(sw.M),
                    "reference_cpu_s": float(g["secs"].sum()), "reference_ms_per_beat": 1e3 * float(g["secs"].sum()) / n,
                    "decisions_identical_to_reference": bool(same)}
    out["ms_per_beat"] = out["r102_n24"]["ms_per_beat"]
    kt = os.path.join(ROOT, "profiles", "r04_online_kernel_stats.json")
    if os.path.exists(kt):                   # rocprofv3 kernel trace of tools/time_online.py (launch counts cannot be taken live)
        out["kernel_trace"] = json.load(open(kt))
    return out


def secondary_shared_grid(dev, ops, S=16384, T=90, reps=10):
    """The reference's shared-grid member dataflow (records 100/102: T = 90, one Sigma_i per segment, one right-hand
    side each; SURVEY.md 8d 'secondary'): hgp_score_each_f64, HBM-bound, 8 T^2 + 16 T + 8 algorithmic bytes per eval.
    Measured outside the timed region of the headline metric."""
    rng = np.random.default_rng(1)
    Q = rng.normal(size=(64, T, T))
    A = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    Sig = torch.as_tensor(np.tile(A, (S // 64, 1, 1)), dtype=torch.float64, device=dev)
    Y = torch.as_tensor(rng.normal(size=(S, T)), dtype=torch.float64, device=dev)
    mean = torch.as_tensor(rng.normal(size=(S, T)), dtype=torch.float64, device=dev)
    sm = torch.arange(S, dtype=torch.int32, device=dev)
    Sig = 0.5 * (Sig + Sig.transpose(1, 2)).contiguous()     # exactly symmetric, like the MNIW scale recursion's output
    for _ in range(2):
        ops.score_each(Y, mean, Sig, sm, symmetric=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.score_each(Y, mean, Sig, sm, symmetric=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = S * (8.0 * T * T + 16 * T + 8)
    gbs = byts / (ms * 1e-3) / 1e9
    return {"workload": f"shared-grid member path: {S} segments, one Sigma_i each, T={T} (k_wave_score1)",
            "value": S / (ms * 1e-3), "unit": "evals/s", "kernel_ms": ms,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}}


def secondary_large_T(dev, ops, synth, N=4096, K=16, T=256, reps=3):
    """BASELINE configs[3]'s per-GPU shard on ONE GPU (4 096 of its 32 768 segments = the 8-GPU share): 16 clusters,
    T = 256, irregular grids - the cooperative pairs kernel (one workgroup per pair).  Same accounting as the headline:
    algorithmic FLOPs T^3/3 + 3T^2 per eval against the fp64 MFMA peak."""
    b = synth.synthetic_batch(N, K, T, seed=20260703)
    d = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)  # noqa: E731
    plan = ops.PairsPlan(T, T, b["theta"], device=dev)
    xb, mean, Sig, x, y = d(b["xb"]), d(b["mean"]), d(b["Sigma"]), d(b["x"]), d(b["y"])
    plan.update(xb, mean, Sig)
    plan.loglik(x, y, want_logdet=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.update(xb, mean, Sig)
    e0.record()
    for _ in range(reps):
        quad, _, info = plan.loglik(x, y, want_logdet=False)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert int(info.abs().max()) == 0 and bool(torch.isfinite(quad).all())
    ms = e0.elapsed_time(e1) / reps
    tf = N * K * algorithmic_flops_per_eval(T) / (ms * 1e-3) / 1e12
    return {"workload": f"configs[3] per-GPU shard on one GPU: {N} segments x {K} clusters, T={T}, irregular grids (k_pairs_cooph<16>)",
            "value": N * K / dt, "unit": "evals/s", "ms_per_step": dt * 1e3, "kernel_ms": ms,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_eval": algorithmic_flops_per_eval(T)}}


def secondary_pairs_t90(dev, ops, synth, N=2048, K=8, T=90, reps=10):
    """The per-pair path at the REAL data's size (MIT-BIH beats: T = 90, k_pairs<6>), irregular grids, same accounting as the
    headline (T^3/3 + 3 T^2 algorithmic FLOPs per eval against the fp64 MFMA peak)."""
    b = synth.synthetic_batch(N, K, T, seed=20260703)
    d = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)  # noqa: E731
    plan = ops.PairsPlan(T, T, b["theta"], device=dev)
    xb, mean, Sig, x, y = d(b["xb"]), d(b["mean"]), d(b["Sigma"]), d(b["x"]), d(b["y"])
    plan.update(xb, mean, Sig)
    for _ in range(2):
        plan.loglik(x, y, want_logdet=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        quad, _, info = plan.loglik(x, y, want_logdet=False)
    e1.record()
    torch.cuda.synchronize()
    assert int(info.abs().max()) == 0 and bool(torch.isfinite(quad).all())
    ms = e0.elapsed_time(e1) / reps
    tf = N * K * algorithmic_flops_per_eval(T) / (ms * 1e-3) / 1e12
    return {"workload": f"per-pair path at the records' size: {N} segments x {K} clusters, T={T}, irregular grids (k_pairs<6, true>, eight waves per workgroup = two per SIMD)",
            "value": N * K / (ms * 1e-3), "unit": "evals/s", "kernel_ms": ms,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_eval": algorithmic_flops_per_eval(T)}}


def secondary_pairs_generic(dev, ops, synth, N=2048, K=8, T=128, reps=10):
    """The mask-driven pair kernel k_pairs<8, false> - what scores every segment whose E is not the block-tridiagonal pattern of
    the static sweeps (other length-scales, other spacings).  (a) the headline batch with the static schedule switched off
    (HGP_PAIRS_GENERIC=1: identical pairs, identical MFMA count, kernel against kernel); (b) a batch whose segment grids are
    jittered by +-0.45 points instead of +-0.3 (neighbours stay >= 0.1 apart): entries 13 points off the diagonal come alive, k-steps
    outside the static schedule, and the band kernel hands those segments over by itself.  Same accounting as the headline (T^3/3 + 3 T^2 FLOPs per eval against the fp64 MFMA peak)."""
    b = synth.synthetic_batch(N, K, T, seed=20260703)
    d = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)  # noqa: E731
    plan = ops.PairsPlan(T, T, b["theta"], device=dev)
    xb, mean, Sig, y = d(b["xb"]), d(b["mean"]), d(b["Sigma"]), d(b["y"])
    plan.update(xb, mean, Sig)
    rng = np.random.default_rng(7)
    xs = {"band": d(b["x"]), "forced": d(b["x"]), "wide_jitter": d(b["xb"][None, :] + rng.uniform(-0.45, 0.45, size=(N, T)))}
    out = {}
    for tag, x in xs.items():
        if tag == "forced":
            os.environ["HGP_PAIRS_GENERIC"] = "1"
        try:
            for _ in range(2):
                plan.loglik(x, y, want_logdet=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                quad, _, info = plan.loglik(x, y, want_logdet=False)
            e1.record()
            torch.cuda.synchronize()
        finally:
            os.environ.pop("HGP_PAIRS_GENERIC", None)
        assert int(info.abs().max()) == 0 and bool(torch.isfinite(quad).all())
        ms = e0.elapsed_time(e1) / reps
        tf = N * K * algorithmic_flops_per_eval(T) / (ms * 1e-3) / 1e12
        out[tag] = {"kernel_ms": ms, "value": N * K / (ms * 1e-3), "achieved": tf, "frac": tf / FP64_MFMA_PEAK_TFLOPS}
    return {"workload": f"{N} segments x {K} clusters, T={T}: k_pairs<8, true> (band) vs k_pairs<8, false> on the same pairs (forced) and on "
                        "segment grids jittered by +-0.45 points (wide_jitter: live k-steps outside the static schedule)",
            "value": out["forced"]["value"], "unit": "evals/s", "kernel_ms": out["forced"]["kernel_ms"],
            "generic_over_band_same_pairs": out["band"]["kernel_ms"] / out["forced"]["kernel_ms"],
            "runs": out,
            "roofline": {"bound": "mfma", "achieved": out["forced"]["achieved"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": out["forced"]["frac"], "algorithmic_flops_per_eval": algorithmic_flops_per_eval(T)}}


def secondary_rank1(dev, ops, b=1024, T=256, reps=5):
    """BASELINE configs[4]'s kernel: L <- chol(alpha L L^T + beta v v^T) by a rank-1 update, batch of b factors, T = 256.
    HBM-bound: 8 T^2 algorithmic bytes per update (the lower triangle in and out)."""
    from hdpgpc_amd import _ffi
    rng = np.random.default_rng(2)
    Q = rng.normal(size=(8, T, T))
    L0 = np.linalg.cholesky(Q @ Q.transpose(0, 2, 1) / T + np.eye(T))
    L = torch.as_tensor(np.tile(L0, (b // 8, 1, 1)), dtype=torch.float64, device=dev)
    v = torch.as_tensor(rng.normal(size=(b, T)), dtype=torch.float64, device=dev)
    al = torch.full((b,), 0.98, dtype=torch.float64, device=dev)
    be = torch.full((b,), 0.5, dtype=torch.float64, device=dev)
    info = torch.zeros(b, dtype=torch.int32, device=dev)

    def run():
        _ffi.check(_ffi.lib.hgp_chol_rank1_f64(ops._ptr(L), ops._ptr(v), ops._ptr(al), ops._ptr(be), T, b, ops._ptr(info),
                                               ops._stream()), "chol_rank1")
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    assert int(info.abs().max()) == 0
    ms = e0.elapsed_time(e1) / reps
    gbs = b * 8.0 * T * T / (ms * 1e-3) / 1e9
    return {"workload": f"configs[4] kernel: rank-1 Cholesky update of {b} factors, T={T} (k_chol_rank1_pipe; opt-in kernel, see DESIGN.md section 0)",
            "value": b / (ms * 1e-3), "unit": "updates/s", "kernel_ms": ms,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}}


def secondary_matrix_terms(dev, ops, b=16384, T=90, reps=5):
    """a8 (latent-transition score) and a9 (MNIW log-likelihood of the LDS parameters, the online hot spot) at the
    records' size T = 90: one fused kernel each, one wavefront per item.  Algorithmic HBM bytes per item:
    a8 3 T^2 + 2 T doubles (A, Gamma, P, two vectors), a9 2 T^2 (M, Sigma; the prior is shared)."""
    rng = np.random.default_rng(3)
    Q = rng.normal(size=(8, T, T))
    G = Q @ Q.transpose(0, 2, 1) / T + np.eye(T)
    d = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)  # noqa: E731
    Gam = d(np.tile(G, (b // 8, 1, 1)))
    A = d(rng.normal(size=(b, T, T)) * 0.1)
    fc, fp = d(rng.normal(size=(b, T))), d(rng.normal(size=(b, T)))
    M, m0, sc = d(rng.normal(size=(b, T, T))), d(np.eye(T)), d(1.7 * np.eye(T))

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out[0] if isinstance(out, tuple) else out).all())
        return e0.elapsed_time(e1) / reps

    ms8 = timed(lambda: ops.lat_error(fc, fp, A, Gam, Gam))
    ms9 = timed(lambda: ops.mniw_loglik(M, Gam, m0, None, sc, scale_is_diagonal=True))
    gb8 = b * (3.0 * T * T + 2 * T) * 8 / (ms8 * 1e-3) / 1e9
    gb9 = b * 2.0 * T * T * 8 / (ms9 * 1e-3) / 1e9
    if T > 128:
        # these sizes are bound by the matrix core, not by HBM.  Yardstick (the reference's operation count, kept from round 3): a8
        # factor T^3/3 + solve A T^3 + solve A P T^3 + the product A P 2 T^3 ... = 11/3 T^3 flops for 3 T^2 doubles (39 flop/byte at
        # T = 256 against a ridge of 9.8), a9 with the diagonal prior scale 5/3 T^3.  The fused cooperative kernels of round 4
        # (hgp_matlik_coop.hip) EXECUTE 5/3 T^3 for a9 and 7/3 T^3 for a8 (Gram form: factor, one solve, Y^T Y by symmetry).
        tf8 = b * (11.0 / 3.0) * T ** 3 / (ms8 * 1e-3) / 1e12
        tf9 = b * (5.0 / 3.0) * T ** 3 / (ms9 * 1e-3) / 1e12
        return {"workload": f"{b} items, T={T}: a8 / a9 as ONE cooperative launch each (k_coop_lat / k_coop_mniw: factor once, packed factor "
                            "in the workspace, forward solves by panel pairs, reductions fused; a8 in Gram form: 7/3 T^3 executed)",
                "a8": {"value": b / (ms8 * 1e-3), "unit": "evals/s", "kernel_ms": ms8,
                       "roofline": {"bound": "mfma", "achieved": tf8, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf8 / FP64_MFMA_PEAK_TFLOPS, "flops_per_eval": (11.0 / 3.0) * T ** 3, "hbm_GBps": gb8}},
                "a9": {"value": b / (ms9 * 1e-3), "unit": "evals/s", "kernel_ms": ms9,
                       "roofline": {"bound": "mfma", "achieved": tf9, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf9 / FP64_MFMA_PEAK_TFLOPS, "flops_per_eval": (5.0 / 3.0) * T ** 3, "hbm_GBps": gb9}}}
    return {"workload": f"{b} items, T={T}: a8 hgp_lat_error_f64 (k_wave_lat_gram), a9 hgp_mniw_loglik_f64 (k_wave_mniw, diagonal prior scale)",
            "a8": {"value": b / (ms8 * 1e-3), "unit": "evals/s", "kernel_ms": ms8,
                   "roofline": {"bound": "hbm", "achieved": gb8, "peak": 8000.0, "unit": "GB/s", "frac": gb8 / 8000.0}},
            "a9": {"value": b / (ms9 * 1e-3), "unit": "evals/s", "kernel_ms": ms9,
                   "roofline": {"bound": "hbm", "achieved": gb9, "peak": 8000.0, "unit": "GB/s", "frac": gb9 / 8000.0}}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline workload only (used for the PMC passes)")
    ap.add_argument("--no-offline", action="store_true", help="skip the driver-level secondaries (include_batch on records 100 / 102, include_sample at T = 256: ~10^5 launches)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal on a one-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--segments", type=int, default=0, help="override the batch size (rehearsals only; 0 = the named config)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))       # no GPU call has been made by this process
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # the CPU baseline (rank 0, N = 1 only) runs before this process touches the GPU
    cpu = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None

    import torch.distributed as dist
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import synthetic_workload as synth        # workload generator: plain NumPy, independent of oracle/
    from hdpgpc_amd import batch, ops

    if world == 1:   # configs[1]
        n_seg, k_cl, t_len, kern = N_SEG, K_CL, T_LEN, "k_pairs<8, true>"
        name = "BASELINE configs[1]: synthetic 2048 segments x 8 clusters, T=128, fp64, irregular segment grids"
    else:            # configs[3], strong scaling: the whole batch is fixed, rank r scores rows [r N/G, (r+1) N/G)
        n_seg, k_cl, t_len, kern = 32768, 16, 256, "k_pairs_cooph<16>"
        name = "BASELINE configs[3]: synthetic 32768 segments x 16 clusters, T=256, fp64, irregular segment grids"
    if args.segments:
        n_seg = args.segments
    b = synth.synthetic_batch(n_seg, k_cl, t_len, seed=20260703)      # segments: replicated on every rank (inputs)
    d = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)  # noqa: E731
    x, y = d(b["x"]), d(b["y"])
    # cluster state: owned by rank 0, replicated with ONE broadcast; the other ranks pass placeholders
    if rank == 0:
        theta, xb, mean, Sig = batch.broadcast_cluster_state(b["theta"], b["xb"], b["mean"], b["Sigma"], dev)
    else:
        theta, xb, mean, Sig = batch.broadcast_cluster_state(np.zeros_like(b["theta"]), np.zeros_like(b["xb"]),
                                                             np.zeros_like(b["mean"]), np.zeros_like(b["Sigma"]), dev)
    plan = ops.PairsPlan(t_len, t_len, theta, device=dev)
    lo, hi = batch.shard_bounds(n_seg, world, rank)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        plan.update(xb, mean, Sig)
        return batch.emission_scores(plan, x, y, events=None if i is None else (ev0[i], ev1[i]), want_info=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out, info = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert tuple(out.shape) == (n_seg, k_cl)
    assert int(info.abs().max()) == 0 and bool(torch.isfinite(out).all())

    if rank == 0:
        evals = n_seg * k_cl * args.steps
        kern_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in zip(ev0, ev1)]))
        flops = (hi - lo) * k_cl * algorithmic_flops_per_eval(t_len)          # per launch on this rank
        achieved = flops / (kern_ms * 1e-3) / 1e12
        traffic = mfma_per_pair = traffic_src = None
        tf = os.path.join(ROOT, "profiles", "pairs_traffic.json")
        if world == 1 and os.path.exists(tf):
            tfj = json.load(open(tf))
            traffic = tfj.get("hbm_bytes_per_launch")
            mfma_per_pair = tfj.get("mfma_f64_instructions_per_pair")
            traffic_src = {"file": "profiles/pairs_traffic.json", "measured_commit": tfj.get("measured_commit"),
                           "measured_date": tfj.get("measured_date"),
                           "note": "PMC counters of a separate rocprofv3 --pmc run of this bench (tools/profile_bench.sh), not of this run"}
        res = {
            "metric": "GP log-lik evals/sec (NxK batch, T-point segments)",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": name + " (per-pair Gram + Cholesky + score)",
                       "segments": n_seg, "segments_per_gpu": hi - lo, "clusters": k_cl, "T": t_len,
                       "sharding": (f"rows [r N/{world}, (r+1) N/{world}) per rank, cluster state broadcast from rank 0, "
                                    f"1 RCCL all-gather of the score rows") if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": kern, "kernel_ms": kern_ms, "pairs_per_launch": (hi - lo) * k_cl,
                         "algorithmic_flops_per_eval": algorithmic_flops_per_eval(t_len)},
        }
        if traffic_src:
            res["roofline"]["traffic_source"] = traffic_src
        if mfma_per_pair:   # what the kernel EXECUTES (PMC count): cov_f = K** + E^T M' E is formed per pair before the one
            # Cholesky the yardstick counts, so `frac` cannot exceed algorithmic / executed even at 100 % MFMA-busy
            ex = mfma_per_pair * 2048.0
            res["roofline"]["executed_flops_per_eval"] = ex
            res["roofline"]["executed_frac"] = achieved / FP64_MFMA_PEAK_TFLOPS * ex / algorithmic_flops_per_eval(t_len)
            res["roofline"]["frac_ceiling"] = algorithmic_flops_per_eval(t_len) / ex
        if world == 1 and not args.no_secondary:
            res["secondary"] = secondary_shared_grid(dev, ops)
            res["secondary_pairs_T90"] = secondary_pairs_t90(dev, ops, synth)
            res["secondary_pairs_generic"] = secondary_pairs_generic(dev, ops, synth)
            res["secondary_large_T"] = secondary_large_T(dev, ops, synth)
            # the WHOLE batch of configs[3] on this one GPU: the N = 1 point of the strong-scaling curve `--gpus N > 1` measures
            # (there the headline line itself is configs[3], 32 768 / N rows per rank)
            base = secondary_large_T(dev, ops, synth, N=32768, reps=2)
            base["workload"] = "configs[3] whole batch on ONE GPU (strong-scaling base of --gpus N): " + base["workload"].split(": ", 1)[1]
            res["strong_scaling_base"] = base
            res["secondary_rank1"] = secondary_rank1(dev, ops)
            res["secondary_matrix_terms"] = secondary_matrix_terms(dev, ops)
            big = secondary_matrix_terms(dev, ops, b=256, T=256, reps=3)       # configs[4]'s size: composition of batched kernels
            res["secondary_matrix_terms_T256"] = big
            if not args.no_offline:
                res["offline_r100"] = secondary_offline(dev, "100", 119.5, "configs[0]")
                res["offline_r102"] = secondary_offline(dev, "102", 348.0, "configs[2]")
                res["online_t256"] = secondary_online_t256(dev)
        if cpu is not None:
            res["cpu_baseline"] = cpu
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
