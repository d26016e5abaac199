"""Multi-process CPU tests (gloo, world_size 2) of the N > 1 path: contiguous sharding of sequences
and the ONE all-reduce of the packed global statistics (svae_amd/parallel.py).  The per-rank
statistics are produced by the oracle here (no GPU in this container); the collective and the
packing are the product code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lds_numpy
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
from svae_amd.parallel import allreduce_global_stats, shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack(stats_list, lognorms, n):
    """[sum E_init (n^2+n) | sum E_pair (3 n^2) | sum lognorm | count] -- the layout written by
    svae_lds_reduce_stats_f64 (include/svae_hip.h)."""
    out = np.zeros(4 * n * n + n + 2)
    for (Ei, Ep, _), ln in zip(stats_list, lognorms):
        out[:n * n] += Ei[0].ravel()
        out[n * n:n * n + n] += Ei[1]
        o = n * n + n
        for i in range(3):
            out[o + i * n * n:o + (i + 1) * n * n] += np.asarray(Ep[i]).ravel()
        out[o + 3 * n * n] += ln
    out[-1] = len(lognorms)
    return out


def _worker(rank, world, port, B, T, n, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        natparam = rand_lds_natparam(n, np.random.default_rng(0))          # replicated globals
        node = rand_node_potentials((B, T, n), np.random.default_rng(1))    # the whole job's data
        lo, hi = shard_bounds(B)                                           # this rank's sequences
        res = [lds_numpy.natural_lds_estep_general(natparam, (node[0][b], node[1][b])) for b in range(lo, hi)]
        packed = torch.from_numpy(_pack([r[1] for r in res], [r[0] for r in res], n))
        allreduce_global_stats(packed)
        q.put((rank, lo, hi, packed.numpy()))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_the_batch():
    for B in (0, 1, 7, 512, 4097):
        for world in (1, 2, 3, 8):
            segs = [shard_bounds(B, r, world) for r in range(world)]
            assert segs[0][0] == 0 and segs[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(segs[:-1], segs[1:]))
            assert max(h - l for l, h in segs) - min(h - l for l, h in segs) <= 1


def test_allreduce_is_a_noop_without_a_process_group():
    x = torch.arange(5, dtype=torch.float64)
    assert torch.equal(allreduce_global_stats(x.clone()), x)


@pytest.mark.timeout(120)
def test_two_rank_stat_allreduce_matches_single_process():
    B, T, n, world = 7, 6, 3, 2          # uneven shards: 4 + 3 sequences
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, T, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    natparam = rand_lds_natparam(n, np.random.default_rng(0))
    node = rand_node_potentials((B, T, n), np.random.default_rng(1))
    res = [lds_numpy.natural_lds_estep_general(natparam, (node[0][b], node[1][b])) for b in range(B)]
    want = _pack([r[1] for r in res], [r[0] for r in res], n)
    covered = sorted((lo, hi) for _, lo, hi, _ in got)
    assert covered == [(0, 4), (4, 7)]
    for _, _, _, packed in got:
        np.testing.assert_allclose(packed, want, rtol=1e-12, atol=1e-12)   # every rank holds the sum
    assert got[0][3][-1] == B
