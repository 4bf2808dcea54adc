"""Multi-GPU: one process per GPU, the batch of independent series sharded across ranks.

Series never interact (every reference op is batched over the leading dims), so the
solve needs NO data-path collective.  The only exchanges are
  * ``allreduce_gradients``: sum of the vector field's parameter gradients (8,448 floats for H=32, C=8) --
    one small all-reduce over RCCL/xGMI per backward (latency-bound; ring bandwidth is irrelevant at 33 KB);
  * ``shard`` / ``gather_batch``: optional helpers when the data is not born sharded.
Works with any ``torch.distributed`` backend ("nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
import contextlib
import threading

import torch
import torch.distributed as dist

# ---------------------------------------------------------------------------------------------- adaptive solves
# torchdiffeq's dopri5 has ONE step controller for the whole batch (the error norm is an RMS over all series).  When
# the batch is sharded over ranks, every rank running its own controller gives valid but different step sequences.
# Inside `shared_step_control()` the fused adaptive solves (forward K4 and backward K4a) instead all-reduce the two
# (four) error sums of every attempted step, so that all shards take the same step sequence (the unsharded batch's, up to
# the summation order of float32 partial sums).
_local = threading.local()    # .control = (reduce(tensor) -> None, global number of series); read when a solve is planned


@contextlib.contextmanager
def shared_step_control(global_batch, group=None, reduce=None):
    """``with shared_step_control(B_total): z = cdeint(X_shard, func, z0_shard, t)`` (and its ``backward()`` inside the
    same block: the plan made by the forward call keeps the setting).  ``reduce`` defaults to a sum all-reduce over ``group``; every rank must make the same calls."""
    if reduce is None:
        def reduce(sums):
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    previous = getattr(_local, "control", None)
    _local.control = (reduce, int(global_batch))
    try:
        yield
    finally:
        _local.control = previous


def step_control():
    """The active (reduce, global_batch) pair, or None.  cdeint reads it when it plans an adaptive solve; the plan carries
    it into the backward pass (which autograd runs on its own thread)."""
    return getattr(_local, "control", None)


def shard_bounds(n, rank=None, world=None):
    """Contiguous, balanced [lo, hi) slice of ``n`` series owned by ``rank``."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(tensor, rank=None, world=None):
    """This rank's slice along the leading (series) dimension."""
    lo, hi = shard_bounds(tensor.size(0), rank, world)
    return tensor[lo:hi]


def allreduce_gradients(params, group=None):
    """Sum ``.grad`` of every parameter across ranks with ONE flat all-reduce."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized():
        return
    # fast path: the fused adjoint hands out weight/bias gradients as adjacent views of one flat buffer
    base = grads[0]._base
    if (base is not None and all(g._base is base for g in grads) and base.is_contiguous()
            and sum(g.numel() for g in grads) == base.numel()):
        dist.all_reduce(base, op=dist.ReduceOp.SUM, group=group)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    offset = 0
    for g in grads:
        g.copy_(flat[offset:offset + g.numel()].view_as(g))
        offset += g.numel()


def gather_batch(local, total, group=None):
    """All-gather per-rank results of a sharded batch back into series order (ragged shards allowed)."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    biggest = max(sizes)
    padded = local.new_zeros((biggest,) + tuple(local.shape[1:]))
    padded[:local.size(0)] = local
    pieces = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(pieces, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(pieces, sizes)], dim=0)
