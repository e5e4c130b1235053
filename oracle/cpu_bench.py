"""Worker of bench.py's `cpu_baseline` leg (TEST/BENCH INFRASTRUCTURE, never imported by the product path):
`python -m oracle.cpu_bench LO HI N K T SEED BUDGET_S` scores segments [LO, HI) of the synthetic workload with the
NumPy/SciPy oracle, BLAS pinned to ONE thread, for a bounded wall-clock budget, and prints `{"done": .., "dt": ..}`.
A separate process per worker: nothing of the parent's GPU state is inherited."""
import json
import sys
import time


def run_slice(seg_lo, seg_hi, N, K, T, seed, budget_s):
    from threadpoolctl import threadpool_limits

    from oracle import hdpgpc_oracle as orc
    b = orc.synthetic_batch(N, K, T, seed=seed)
    done = 0
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for n in range(seg_lo, seg_hi):
            orc.loglik_pairs(b["x"][n:n + 1], b["y"][n:n + 1], b["xb"], b["theta"], b["mean"], b["Sigma"])
            done += K
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    return done, dt


if __name__ == "__main__":
    a = sys.argv[1:]
    d, t = run_slice(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6]))
    print(json.dumps({"done": d, "dt": t}), flush=True)
