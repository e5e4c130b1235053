"""Worker of tests/test_gpu_sharding.py: one rank of a 2-rank run of hdpgpc_amd.batch.emission_scores whose compute leg is
the HIP kernel.  Both ranks share cuda:0 (the box has one GPU), so the collective runs on gloo (host-staged); rank 0 also
scores the whole batch alone and compares bit for bit.  Launched with torch.distributed.run; exit code 0 = equal."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rank = int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import synthetic_workload as synth
    from hdpgpc_amd import batch, ops
    b = synth.synthetic_batch(n, K, T, seed=11)
    b["theta"][K - 1, 1] = 3.0                      # one ill-conditioned cluster: both kernels take part
    d = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)  # noqa: E731
    x, y = d(b["x"]), d(b["y"])
    if rank == 0:
        st = batch.broadcast_cluster_state(b["theta"], b["xb"], b["mean"], b["Sigma"], dev)
    else:                                            # placeholders: the state arrives by broadcast
        st = batch.broadcast_cluster_state(np.zeros_like(b["theta"]), np.zeros_like(b["xb"]), np.zeros_like(b["mean"]),
                                           np.zeros_like(b["Sigma"]), dev)
    theta, xb, mean, Sig = st
    ok = np.array_equal(theta, b["theta"]) and torch.equal(mean, d(b["mean"])) and torch.equal(Sig, d(b["Sigma"]))
    plan = ops.PairsPlan(T, T, theta, device=dev).update(xb, mean, Sig)
    fn = torch.rand((n, K), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 0.05
    q, info = batch.emission_scores(plan, x, y, first_noise=fn, want_info=True)
    lo, hi = batch.shard_bounds(n, 2, rank)
    ok = ok and tuple(q.shape) == (n, K) and tuple(info.shape) == (hi - lo, K) and int(info.abs().max()) == 0
    # single-rank result on the same device (no collective): rows of q must be identical bit for bit
    full, _ = plan.score(x, y, first_noise=fn)
    ok = ok and bool(torch.equal(q, full))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag[0]) == 1 else 1)


if __name__ == "__main__":
    main()
