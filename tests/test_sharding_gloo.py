"""N > 1 path on CPU: world_size-2 gloo run of the shard -> evaluate -> all-gather plumbing (hdpgpc_amd.batch).
The compute leg here is the CPU oracle (tests may call it); on the GPU it is the HIP kernel (bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hdpgpc_amd import batch
    from oracle import hdpgpc_oracle as orc
    b = orc.synthetic_batch(n, 3, 16, seed=5)

    def score(xs, ys):
        sc, _, _ = orc.loglik_pairs(xs.numpy(), ys.numpy(), b["xb"], b["theta"], b["mean"], b["Sigma"])
        return torch.from_numpy(sc)

    q = batch.sharded_scores(score, torch.from_numpy(b["x"]), torch.from_numpy(b["y"]))
    full, _, _ = orc.loglik_pairs(b["x"], b["y"], b["xb"], b["theta"], b["mean"], b["Sigma"])
    ok = q.shape == (n, 3) and np.array_equal(q.numpy(), full)
    lo, hi = batch.shard_bounds(n, world, rank)
    ret[rank] = (bool(ok), lo, hi)
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8])
def test_shard_and_allgather_world2(n):
    port = 29500 + (os.getpid() % 400) + n
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, n, ret), nprocs=2, join=True)
        ret = dict(ret)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == 0 and ret[0][2] == ret[1][1] and ret[1][2] == n        # contiguous, complete partition


def test_shard_bounds_cover():
    from hdpgpc_amd.batch import shard_bounds
    for n in (0, 1, 5, 32768):
        for w in (1, 2, 3, 8):
            segs = [shard_bounds(n, w, r) for r in range(w)]
            assert segs[0][0] == 0 and segs[-1][1] == n
            assert all(segs[i][1] == segs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in segs) - min(h - l for l, h in segs) <= 1
