"""The N x K batch of GP emission scores, sharded over the GPUs of one node.

Segments (rows of q) are partitioned across ranks; the per-cluster state (theta, mean, Sigma) is replicated - one
broadcast from rank 0 whenever the clusters change; every rank evaluates its rows with the HIP per-pair
kernels and ONE all-gather (RCCL over xGMI with the nccl backend) returns the full [N, K] score matrix to every rank,
where the forward-backward of the sampler runs redundantly.  There is no other data-path collective.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

LOG2PI = math.log(2.0 * math.pi)


def _dist_on(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def shard_bounds(n, world, rank):
    """Contiguous, balanced partition of n rows: rank r owns [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged(group):
    """gloo has no device collectives: tensors are staged through the host (CPU rehearsals of the N > 1 path, and the
    2-rank test that runs both ranks on one GPU).  nccl (= RCCL) moves device buffers directly."""
    return dist.get_backend(group) == "gloo"


def gather_rows(local, n_total, group=None):
    """All-gather row blocks of unequal size back into an [n_total, ...] tensor on every rank (one collective)."""
    if not _dist_on(group):
        return local
    world = dist.get_world_size(group)
    cap = -(-n_total // world)
    dev = local.device
    stage = local.is_cuda and _host_staged(group)
    src = local.cpu() if stage else local
    if n_total % world == 0:
        pad = src.contiguous()                    # equal blocks: the local rows are the send buffer as they are
    else:
        pad = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
    out = torch.empty((world * cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if n_total % world != 0:
        pieces = []
        for r in range(world):
            lo, hi = shard_bounds(n_total, world, r)
            pieces.append(out[r * cap: r * cap + (hi - lo)])
        out = torch.cat(pieces, dim=0)
    return out.to(dev) if stage else out


def broadcast_cluster_state(theta, x_basis, mean, Sigma, device, src=0, group=None):
    """Replicate the per-cluster state from rank `src`: theta [K,3] (host array on return: the plan keeps it host-side),
    x_basis [T], mean [K,T], Sigma [K,T,T] (device tensors on return).  Non-source ranks pass arrays of the right shape
    (their contents are overwritten).  One broadcast of one packed buffer - K (T^2 + T + 3) + T doubles, 8.4 MB at
    K = 16, T = 256."""
    theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).reshape(-1, 3))
    K = theta.shape[0]
    T = int(np.asarray(x_basis).reshape(-1).shape[0])
    parts = [torch.as_tensor(theta, dtype=torch.float64).reshape(-1),
             torch.as_tensor(np.asarray(x_basis), dtype=torch.float64).reshape(-1),
             torch.as_tensor(np.asarray(mean), dtype=torch.float64).reshape(-1),
             torch.as_tensor(np.asarray(Sigma), dtype=torch.float64).reshape(-1)]
    assert parts[2].numel() == K * T and parts[3].numel() == K * T * T
    buf = torch.cat(parts)
    if _dist_on(group):
        if _host_staged(group):
            dist.broadcast(buf, src=src, group=group)
            buf = buf.to(device)
        else:
            buf = buf.to(device)
            dist.broadcast(buf, src=src, group=group)
    else:
        buf = buf.to(device)
    o = np.cumsum([0, 3 * K, T, K * T, K * T * T])
    theta_out = buf[o[0]:o[1]].cpu().numpy().reshape(K, 3)
    return theta_out, buf[o[1]:o[2]].contiguous(), buf[o[2]:o[3]].reshape(K, T).contiguous(), buf[o[3]:o[4]].reshape(K, T, T).contiguous()


def sharded_scores(score_fn, x, y, group=None, events=None):
    """q[N, K] = score_fn(x_rows, y_rows) evaluated on this rank's rows only, then gathered.

    score_fn maps ([n_loc, T], [n_loc, T]) -> [n_loc, K]; on the GPU it is PairsPlan.score (the HIP kernels).
    events: optional (start, end) torch.cuda.Event pair recorded around the local evaluation (bench.py's kernel timing)."""
    n = x.shape[0]
    if _dist_on(group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(n, world, rank)
    if events is not None:
        events[0].record()
    local = score_fn(x[lo:hi], y[lo:hi])
    if events is not None:
        events[1].record()
    return gather_rows(local, n, group)


def emission_scores(plan, x, y, first_noise=None, group=None, events=None, want_info=False):
    """Reference scores (no log-determinant) of every (segment, cluster) pair; x, y [N, Ts] replicated on all ranks.
    Returns q [N, K] on every rank (and, with want_info, this rank's LAPACK info block [n_loc, K])."""
    infos = []
    if _dist_on(group):
        lo, hi = shard_bounds(x.shape[0], dist.get_world_size(group), dist.get_rank(group))
    else:
        lo, hi = 0, x.shape[0]

    def fn(xs, ys):
        fnl = None if first_noise is None else first_noise[lo:hi].contiguous()
        if events is not None:
            events[0].record()
        q, _, info = plan.loglik(xs.contiguous(), ys.contiguous(), first_noise=fnl, want_logdet=False, score=True)
        if events is not None:                   # the bracket holds the pair kernels only
            events[1].record()
        infos.append(info)
        return q                                 # GPI_model.py:285 (no log-determinant), written by the kernels

    q = sharded_scores(fn, x, y, group)
    return (q, infos[0]) if want_info else q
