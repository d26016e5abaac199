"""Multi-process CPU tests (gloo, world_size 2) of the N > 1 path: contiguous sharding of sequences
and the ONE all-reduce of the packed global statistics + local KL (svae_amd/parallel.py:
allreduce_lds_stats, the function models.lds.run_inference / run_inference_differentiable and
bench.py end with; models.gmm._allreduce_stats_and_kl for the GMM).  The per-rank kernel outputs are
produced by the oracle here (no GPU in this container), in the layout svae_lds_reduce_stats_f64
writes; the collective, the packing and the unpacking are the product code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lds_numpy
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
from svae_amd.parallel import allreduce_global_stats, allreduce_lds_stats, shard_bounds


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
        # local KL of this shard, on the autograd tape like run_inference_differentiable's
        w = torch.ones((), dtype=torch.float64, requires_grad=True)
        kl_local = sum(float((node[0][b] * r[1][2][0]).sum() + (node[1][b] * r[1][2][1]).sum() - r[0])
                       for b, r in zip(range(lo, hi), res))
        niw, mniw, kl = allreduce_lds_stats(packed.clone(), w * kl_local, n, T)
        kl.backward()
        allreduce_global_stats(packed)
        # GMM flavour of the same exchange
        from svae_amd.models.gmm import _allreduce_stats_and_kl
        (ds, ns), gkl = _allreduce_stats_and_kl((torch.full((3,), 1.0 + rank, dtype=torch.float64),
                                                 torch.full((3, 4, 4), 2.0 * (rank + 1), dtype=torch.float64)),
                                                torch.tensor(0.5 + rank, dtype=torch.float64), None)
        q.put((rank, lo, hi, packed.numpy(), niw.numpy(), [np.asarray(x) for x in mniw[:3]] + [float(mniw[3])],
               float(kl.detach()), float(w.grad), kl_local, ds.numpy(), ns.numpy(), float(gkl)))
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
    covered = sorted((g[1], g[2]) for g in got)
    assert covered == [(0, 4), (4, 7)]
    from oracle import expfam_numpy as ef
    kl_all = sum(float((node[0][b] * r[1][2][0]).sum() + (node[1][b] * r[1][2][1]).sum() - r[0]) for b, r in enumerate(res))
    o = n * n + n
    for g in got:
        packed, niw, mniw, kl, wgrad, kl_local = g[3], g[4], g[5], g[6], g[7], g[8]
        np.testing.assert_allclose(packed, want, rtol=1e-12, atol=1e-12)   # every rank holds the sum
        # unpacked exactly like models.lds.run_inference: NIW dense-packed with the counts, MNIW 4-tuple
        np.testing.assert_allclose(niw, ef.pack_dense(want[:n * n].reshape(n, n), want[n * n:o], np.array(float(B)),
                                                      np.array(float(B))), rtol=1e-12, atol=1e-12)
        for i in range(3):
            np.testing.assert_allclose(mniw[i], want[o + i * n * n:o + (i + 1) * n * n].reshape(n, n), rtol=1e-12, atol=1e-12)
        assert mniw[3] == B * (T - 1)
        assert kl == pytest.approx(kl_all, rel=1e-12)          # global value ...
        assert wgrad == pytest.approx(kl_local, rel=1e-12)     # ... this rank's gradient
        np.testing.assert_allclose(g[9], 3.0)                  # GMM: 1 + 2
        np.testing.assert_allclose(g[10], 6.0)                 #      2 + 4
        assert g[11] == pytest.approx(2.0)                     #      0.5 + 1.5
    assert got[0][3][-1] == B
