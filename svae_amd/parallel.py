"""Multi-GPU exchange step of the structured E-step.

Sequences (LDS) and points (GMM) are conditionally independent given the replicated global natural
parameters, so the batch is sharded across ranks with no data-path communication; the ONLY
collective is one all-reduce (sum, fp64) per natural-gradient step of the packed global expected
statistics (4n^2+n+2 doubles for the LDS: [sum E_init | sum E_pair | sum lognorm | count]) that feed
svae/svae.py:33-34 in the reference.  At ~3 KB it is latency-bound on xGMI: a single RCCL call on
one packed buffer (backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests) -- or, opt-in, the one-shot
mailbox kernel of svae_amd/ipc.py (use_mailbox_allreduce).
"""
import torch
import torch.distributed as dist


_mailbox = {}      # group -> svae_amd.ipc.MailboxAllReduce (opt-in, use_mailbox_allreduce)


def use_mailbox_allreduce(n_doubles, group=None):
    """Opt in: route allreduce_global_stats on `group` through the one-shot IPC mailbox kernel (svae_amd/ipc.py: one
    launch, one xGMI hop, rank-order sum fused in) for buffers of up to `n_doubles` float64 CUDA elements; larger or
    non-CUDA buffers keep the collective backend.  One node only.  Returns the MailboxAllReduce: a peer that never
    publishes within the spin limit turns the affected elements into NaN (never a plausible-looking sum) and raises
    the status word, which .check() -- called automatically every `check_every` calls -- reports.  Validated with several processes on ONE device (tests/test_distributed_hip.py); RCCL
    stays the default until a multi-GPU run has measured both."""
    from .ipc import MailboxAllReduce
    ar = MailboxAllReduce(n_doubles, group)
    _mailbox[group] = ar
    return ar


def allreduce_global_stats(packed, group=None):
    """In-place sum over ranks of a packed statistics buffer; no-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        ar = _mailbox.get(group)
        if ar is not None and packed.is_cuda and packed.dtype == torch.float64 and packed.is_contiguous() \
                and packed.numel() <= ar.n:
            ar(packed)
        else:
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


def allreduce_nested(struct, scalar=None, group=None):
    """ONE all-reduce (sum) of every tensor leaf of a nested tuple/list `struct` plus an optional scalar tensor:
    the leaves are packed into one buffer, reduced, and unpacked into the same nesting (the exchange step of the
    SLDS model: svae/models/slds_svae.py:229-243 statistics are sums over sequences).  Returns (struct, scalar);
    the scalar carries the global VALUE and, if it is on the autograd tape, this rank's gradient.  No-op without
    a process group."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return struct, scalar
    leaves = []

    def walk(s):
        if isinstance(s, (tuple, list)):
            for x in s:
                walk(x)
        else:
            leaves.append(s)
    walk(struct)
    parts = [x.detach().reshape(-1).to(torch.float64) for x in leaves]
    if scalar is not None:
        parts.append(scalar.detach().reshape(1).to(torch.float64))
    packed = torch.cat(parts)
    allreduce_global_stats(packed, group)
    pos = [0]

    def build(s):
        if isinstance(s, (tuple, list)):
            return tuple(build(x) for x in s)
        k = s.numel()
        out = packed[pos[0]:pos[0] + k].reshape(s.shape)
        pos[0] += k
        return out
    out = build(struct)
    if scalar is not None:
        scalar = scalar + (packed[-1] - scalar.detach())
    return out, scalar


def allreduce_lds_stats(reduced, local_kl, n, T, group=None, return_packed=False):
    """The exchange step of the LDS model (svae/models/lds.py:35-52 with B sequences per rank).

    `reduced` is this rank's buffer from svae_lds_reduce_stats_f64, [sum E_init (n^2+n) | sum E_pair (3n^2)
    | sum lognorm | count]; `local_kl` this rank's sum of <nn_potentials, E_node> - lognorm.  ONE all-reduce
    ships both (the sum-lognorm slot carries the local KL).  Returns (niw_stats dense-packed (n+2,n+2),
    mniw_stats (4-tuple), local_kl) with the statistics and the VALUE of local_kl summed over all ranks;
    if local_kl is on the autograd tape its gradient stays this rank's (each rank differentiates its own
    shard, gradients are averaged by the caller's DDP)."""
    from .distributions import expfam
    packed = reduced.clone()
    packed[-2] = local_kl.detach()
    allreduce_global_stats(packed, group)
    nn_ = n * n
    o = nn_ + n
    cnt = packed[-1]
    niw_stats = expfam.pack_dense(packed[:nn_].reshape(n, n), packed[nn_:o], cnt, cnt)
    mniw_stats = (packed[o:o + nn_].reshape(n, n), packed[o + nn_:o + 2 * nn_].reshape(n, n),
                  packed[o + 2 * nn_:o + 3 * nn_].reshape(n, n), cnt * (T - 1))
    kl = local_kl + (packed[-2] - local_kl.detach())
    return (niw_stats, mniw_stats, kl, packed) if return_packed else (niw_stats, mniw_stats, kl)
