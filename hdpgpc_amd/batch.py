"""The N x K batch of GP emission scores, sharded over the GPUs of one node.

Segments (rows of q) are partitioned across ranks; the per-cluster state is replicated; every rank evaluates
its rows with the HIP per-pair kernel and ONE all-gather (RCCL over xGMI with the nccl backend) returns the full
[N, K] score matrix to every rank, where the (host-side) forward-backward of the sampler runs redundantly
(SURVEY.md 8e).  There is no other data-path collective.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous, balanced partition of n rows: rank r owns [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, n_total, group=None):
    """All-gather row blocks of unequal size back into an [n_total, ...] tensor on every rank (one collective)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    cap = -(-n_total // world)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        pieces.append(out[r * cap: r * cap + (hi - lo)])
    return torch.cat(pieces, dim=0)


def sharded_scores(score_fn, x, y, group=None):
    """q[N, K] = score_fn(x_rows, y_rows) evaluated on this rank's rows only, then gathered.

    score_fn maps ([n_loc, T], [n_loc, T]) -> [n_loc, K]; on the GPU it is PairsPlan.score (the HIP kernel)."""
    n = x.shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(n, world, rank)
    return gather_rows(score_fn(x[lo:hi], y[lo:hi]), n, group)


def emission_scores(plan, x, y, first_noise=None, group=None):
    """Reference scores (no log-determinant) of every (segment, cluster) pair; x, y [N, Ts] replicated on all ranks."""
    def fn(xs, ys):
        return plan.score(xs.contiguous(), ys.contiguous())[0]
    return sharded_scores(fn, x, y, group)
