"""Probe-parallel helpers: one process per GPU, torch.distributed (backend
"nccl" is RCCL on ROCm; "gloo" in CPU tests).

The reference's only parallel axis is the N+1 independent solves mapped over a
process pool (runlmc/lmc/stochastic_deriv.py:39-52).  Here the N probes are
dealt round-robin to the ranks, every rank holds a replica of the operator and
solves for alpha itself (rank 0's alpha is then broadcast, n doubles, so that
all ranks assemble bit-identical gradients), and ONE all-reduce of the summed
gradient partials (a few KB) closes the step.  There is no collective inside
the solve."""
import torch
import torch.distributed as dist


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
    if world == 1:
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
    if world == 1:
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
