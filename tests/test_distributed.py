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


# ---- GMM fixed point with the points sharded over ranks: the sweep protocol (models.gmm.run_sweeps) -------

class _NumpySweeps(object):
    """run_sweeps backend with the HIP kernels' semantics (include/svae_hip.h, svae_gmm_mw_*) restated on the
    oracle's per-point functions: sweep i is a no-op once an earlier sweep met the stopping rule on kl_hist."""

    def __init__(self, lg, gg, node_dense, init, tol, max_iter):
        from oracle import gmm_numpy
        self.g, self.lg, self.gg, self.node, self.tol, self.max_iter = gmm_numpy, lg, gg, node_dense, tol, max_iter
        self.kl_hist = torch.zeros(max_iter + 1, dtype=torch.float64)
        self.r = init

    def _converged_at(self, upto):
        prev = np.inf
        for j in range(upto):
            if abs(float(self.kl_hist[j]) - prev) < self.tol:
                return j
            prev = float(self.kl_hist[j])
        return -1

    def begin(self):
        self.kl_hist.zero_()

    def sweep(self, i):
        if self._converged_at(i) >= 0:
            return
        g = self.g
        natparam, stats, gkl = g.gaussian_meanfield(self.gg, self.node, self.r)
        _, r_new, lkl = g.label_meanfield(self.lg, self.gg, stats)
        lin = natparam - np.tensordot(r_new, self.gg, [1, 0]) - self.node
        self.kl_hist[i] = float(lkl + gkl + np.tensordot(lin, stats, 3))       # this rank's points only
        self.r = r_new

    def final(self):
        g = self.g
        conv = self._converged_at(self.max_iter)
        self.iters = conv + 1 if conv >= 0 else self.max_iter
        _, self.gstats, gkl = g.gaussian_meanfield(self.gg, self.node, self.r)
        _, self.r, lkl = g.label_meanfield(self.lg, self.gg, self.gstats)
        self.kl = float(lkl + gkl)

    def stats(self):
        self.dirichlet_stats = self.r.sum(0)
        self.niw_stats = np.tensordot(self.r, self.gstats, [0, 0])


def _gmm_problem(T, N, K, seed):
    from oracle import expfam_numpy as ef
    rng = np.random.default_rng(seed)
    niw = np.stack([ef.niw_standard_to_natural((N + 10.) * np.eye(N), 2 * rng.standard_normal(N), np.array(10.),
                                               np.array(N + 10.)) for _ in range(K)])
    lg, gg = ef.dirichlet_expectedstats(rng.random(K) + 0.5), ef.niw_expectedstats(niw)
    node = rand_node_potentials((T, N), rng)
    init = rng.random((T, K))
    init /= init.sum(-1, keepdims=True)
    return lg, gg, ef.pack_dense(*node), node, init


def _gmm_worker(rank, world, port, T, N, K, tol, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svae_amd.models.gmm import run_sweeps, _allreduce_stats_and_kl
        lg, gg, dense, _, init = _gmm_problem(T, N, K, 3)
        lo, hi = shard_bounds(T)
        be = _NumpySweeps(lg, gg, dense[lo:hi], init[lo:hi], tol, 100)
        run_sweeps(be, 100)
        (ds, ns), kl = _allreduce_stats_and_kl((torch.from_numpy(be.dirichlet_stats), torch.from_numpy(be.niw_stats)),
                                               torch.tensor(be.kl, dtype=torch.float64), None)
        q.put((rank, lo, hi, be.iters, be.r, ds.numpy(), ns.numpy(), float(kl)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gmm_fixed_point_stops_on_the_batch_total_kl():
    """Points sharded over 2 ranks, kl_hist[i] all-reduced after every sweep: same iteration count, same
    responsibilities and the same summed statistics as ONE process on all the points (the reference's rule is on
    the minibatch total, gmm.py:104-105); each shard ALONE would stop at a different sweep."""
    from oracle import gmm_numpy
    T, N, K, tol, world = 301, 2, 5, 1e-3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gmm_worker, args=(r, world, port, T, N, K, tol, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=150) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    lg, gg, dense, node, init = _gmm_problem(T, N, K, 3)
    (ls, gs), (ds, ns), _, kl, iters = gmm_numpy.local_meanfield(lg, gg, node, init, tol)
    alone = [gmm_numpy.meanfield_fixed_point(lg, gg, dense[g[1]:g[2]], init[g[1]:g[2]], tol, return_iters=True)[1]
             for g in got]
    assert any(a != iters for a in alone), "pick a problem whose shards stop at different sweeps on their own"
    for g in got:
        assert g[3] == iters
        np.testing.assert_allclose(g[4], ls[g[1]:g[2]], rtol=1e-10, atol=1e-13)
        assert np.array_equal(g[4].argmax(1), ls[g[1]:g[2]].argmax(1))
        np.testing.assert_allclose(g[5], ds, rtol=1e-11)
        np.testing.assert_allclose(g[6], ns, rtol=1e-10, atol=1e-10)
        assert g[7] == pytest.approx(kl, rel=1e-10)


def test_sweep_protocol_single_process_matches_the_fixed_point():
    """run_sweeps without a process group = the plain fixed point (iteration count and responsibilities)."""
    from oracle import gmm_numpy
    from svae_amd.models.gmm import run_sweeps
    for T, N, K, mi in ((150, 2, 4, 100), (40, 3, 3, 2), (10, 1, 2, 0)):
        lg, gg, dense, node, init = _gmm_problem(T, N, K, T)
        be = _NumpySweeps(lg, gg, dense, init, 1e-3, mi)
        run_sweeps(be, mi)
        (ls, _), (ds, ns), _, kl, iters = gmm_numpy.local_meanfield(lg, gg, node, init, 1e-3, mi)
        assert be.iters == iters
        np.testing.assert_allclose(be.r, ls, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(be.niw_stats, ns, rtol=1e-12, atol=1e-12)
        assert be.kl == pytest.approx(kl, rel=1e-12)


def _nested_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svae_amd.parallel import allreduce_nested
        K, n = 3, 2
        f = lambda *shape: torch.full(shape, 1.0 + rank, dtype=torch.float64)
        stats = ((f(K), f(K, K)), ((f(K, n, n), f(K, n), f(K), f(K)), (f(K, n, n), f(K, n, n), f(K, n, n), f(K))))   # get_global_stats nesting
        w = torch.ones((), dtype=torch.float64, requires_grad=True)
        out, vlb = allreduce_nested(stats, w * (10.0 + rank))
        vlb.backward()
        leaves = []
        def walk(s):
            for x in s:
                walk(x) if isinstance(x, tuple) else leaves.append(x)
        walk(out)
        q.put((rank, [float(x.min()) for x in leaves], [tuple(x.shape) for x in leaves], float(vlb.detach()), float(w.grad)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_nested_allreduce_of_the_slds_statistics():
    """The SLDS exchange step (models.slds_svae.run_inference(group=...)): every leaf of the nested statistics
    and the local bound summed over ranks in ONE collective, nesting and shapes preserved, gradient local."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nested_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, mins, shapes, vlb, grad in got:
        assert all(m == 3.0 for m in mins)                       # 1 + 2 in every entry
        assert shapes == [(3,), (3, 3), (3, 2, 2), (3, 2), (3,), (3,), (3, 2, 2), (3, 2, 2), (3, 2, 2), (3,)]
        assert vlb == 21.0 and grad == 10.0 + rank               # global value, this rank's gradient


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the form of the driver's N = 1 command) must run two ranks -- or
    fail -- never print an N = 1 number under an N = 2 label: bench.py re-executes itself under torch.distributed.run
    and reports the observed world size and the devices of the ranks.  CPU rehearsal with gloo (`--launch-check` stops
    after rendezvous + barrier + all-gather; the measurement itself needs a GPU: tests/test_distributed_hip.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVAE_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    got = json.loads(line)
    assert got["n_gpus"] == 2 and got["world_size_observed"] == 2 and len(got["rank_devices"]) == 2
    assert got["backend"] == "gloo" and got["launcher"] == "torch.distributed.run"
    # RCCL with fewer visible devices than ranks: refuse (non-zero), do not fall back to one rank
    env["SVAE_BENCH_BACKEND"] = "nccl"
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)
    # a WORLD_SIZE that contradicts --gpus (a launcher mis-invocation) is refused as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", SVAE_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env2, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
