"""The N > 1 path with the HIP kernel as its compute leg: two ranks (gloo, both on cuda:0 - the box has one GPU) run
hdpgpc_amd.batch.emission_scores - the function bench.py times - and must reproduce the single-rank scores bit for bit,
with the cluster state arriving by broadcast from rank 0.  (RCCL itself needs one GPU per rank: the driver's 8-GPU run.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,K,T", [(37, 3, 40), (64, 4, 144)])
def test_two_ranks_on_one_gpu_match_single_rank(n, K, T):
    port = 29600 + (os.getpid() % 300) + n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_shard_worker.py"), str(n), str(K), str(T)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
