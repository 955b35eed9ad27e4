"""Probe-parallel helpers: one process per GPU, torch.distributed (backend
"nccl" is RCCL on ROCm; "gloo" in CPU tests).

The reference's only parallel axis is the N+1 independent solves mapped over a
process pool (runlmc/lmc/stochastic_deriv.py:39-52).  Here the N probes are
dealt round-robin to the ranks, every rank holds a replica of the operator and
solves for alpha next to its probes; rank 0's alpha is broadcast after the solve
(n doubles) and ONE all-reduce of the summed gradient partials (a few KB, rank
0's alpha terms among them) closes the step, so that every rank holds the same
bits whatever its shard size.  There is no collective inside the solve.  A
plain block of right-hand sides is split the same way and put together again
with one all-gather.

`force_collectives(True)` makes every helper issue its collective even in a
world of ONE rank: on a one-GPU box that is the only way to push the step's
broadcast / all-reduce / all-gather through RCCL itself (bench.py --force-dist,
tests/test_gpu_multi_rank.py)."""
import torch
import torch.distributed as dist

_FORCE = False


def force_collectives(on=True):
    """Issue collectives even when the world has one rank (needs an
    initialised process group)."""
    global _FORCE
    _FORCE = bool(on)


def _skip(world):
    return world == 1 and not (_FORCE and dist.is_available() and dist.is_initialized())


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_rows(count, group=None):
    """Indices of the probes this rank owns: rank, rank + world, ..."""
    rank, world = rank_world(group)
    return list(range(rank, count, world))


def broadcast_(t, src=0, group=None):
    """In-place broadcast from rank `src`; no-op for one rank."""
    rank, world = rank_world(group)
    if _skip(world):
        return t
    backend = dist.get_backend(group)
    if backend == 'gloo' and t.device.type != 'cpu':
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def all_reduce_sum_(flat, group=None):
    """In-place sum over ranks of a 1-D float64 tensor; no-op for one rank.
    Moves through host memory when the backend cannot take the tensor's
    device (gloo with a GPU tensor)."""
    rank, world = rank_world(group)
    if _skip(world):
        return flat
    backend = dist.get_backend(group)
    if backend == 'gloo' and flat.device.type != 'cpu':
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    elif backend == 'nccl' and flat.device.type == 'cpu':
        dev = flat.cuda()
        dist.all_reduce(dev, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(dev)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def all_gather_rows(local, counts, group=None):
    """Concatenation over ranks of (rows_r, width) float64 blocks, rank order;
    `counts[r]` = rows of rank r (known to everyone).  One all_gather of
    equal-sized buffers (each rank pads its block to max(counts) rows): a rank
    sends its own rows once instead of reducing a world-sized zero-padded
    block.  Returns a (sum(counts), width) tensor on `local`'s device."""
    rank, world = rank_world(group)
    if _skip(world):
        return local
    width = local.shape[1]
    most = max(counts)
    backend = dist.get_backend(group)
    on_host = backend == 'gloo' and local.device.type != 'cpu'
    send = torch.zeros((most, width), dtype=local.dtype,
                       device='cpu' if on_host else local.device)
    send[:local.shape[0]] = local.cpu() if on_host else local
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    out = torch.cat([recv[r][:counts[r]] for r in range(world)], dim=0)
    return out.to(local.device)
