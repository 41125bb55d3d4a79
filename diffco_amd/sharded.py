"""Multi-GPU evaluation of the score(+grad) path: one process per GPU, the configuration batch
sharded across ranks, the model (supports + weights, <= a few hundred KB) replicated.

The reference has no distributed code at all.  Evaluations of different
configurations are independent, so the only exchange is an all-gather of the per-rank results
(`torch.distributed` backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).  The
messages are tiny (config #3: 65536 x 5 floats = 1.3 MB in total) so the collective is latency-
bound; when the consumer is itself sharded (fused trajectory optimiser) skip the gather and keep
the results local.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """contiguous [lo, hi) slice of n items for `rank`: the first n % world ranks get one extra"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local, n_total, group=None):
    """Concatenate per-rank row blocks (shard_bounds order) into [n_total, ...] on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo (the CPU tests, and the two-processes-on-one-GPU test) moves host memory: stage through it
        return all_gather_rows(local.cpu(), n_total, group).to(local.device)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    tail = tuple(local.shape[1:])
    if len(set(counts)) == 1:
        out = local.new_empty((n_total,) + tail)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged: pad every block to the largest, gather, then strip the padding
    m = max(counts)
    pad = local.new_zeros((m,) + tail)
    pad[:local.shape[0]] = local
    buf = local.new_empty((world * m,) + tail)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * m:r * m + counts[r]] for r in range(world)], dim=0)


class ShardedScorer:
    """Wraps any `fn(q_local) -> tensor or tuple of tensors with leading dim len(q_local)` (e.g.
    `ScoreModel.score_and_grad`, `DiffCo.poly_score`) so that a call with the FULL batch on every rank
    evaluates only this rank's slice and returns the gathered full result on every rank."""

    def __init__(self, fn, group=None, gather=True):
        self.fn, self.group, self.gather = fn, group, gather

    def __call__(self, q_full, *args):
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        n = len(q_full)
        lo, hi = shard_bounds(n, rank, world)
        sliced = [a[lo:hi] if torch.is_tensor(a) and a.ndim > 0 and len(a) == n else a for a in args]
        out = self.fn(q_full[lo:hi], *sliced)
        if not self.gather or world == 1:
            return out
        if isinstance(out, (tuple, list)):
            return type(out)(all_gather_rows(o, n, self.group) for o in out)
        return all_gather_rows(out, n, self.group)
