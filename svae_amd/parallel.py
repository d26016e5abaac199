"""Multi-GPU exchange step of the structured E-step.

Sequences (LDS) and points (GMM) are conditionally independent given the replicated global natural
parameters, so the batch is sharded across ranks with no data-path communication; the ONLY
collective is one all-reduce (sum, fp64) per natural-gradient step of the packed global expected
statistics (4n^2+n+2 doubles for the LDS: [sum E_init | sum E_pair | sum lognorm | count]) that feed
svae/svae.py:33-34 in the reference.  At ~3 KB it is latency-bound on xGMI: a single RCCL call on
one packed buffer (backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).
"""
import torch
import torch.distributed as dist


def allreduce_global_stats(packed, group=None):
    """In-place sum over ranks of a packed statistics buffer; no-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


def shard_bounds(num_items, rank=None, world=None):
    """Contiguous shard [lo, hi) of `num_items` sequences/points owned by `rank`."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
