"""GPU tests of the N > 1 path with REAL kernel outputs in the collective: two ranks (gloo) share the one MI355X of
the test box -- RCCL refuses two ranks on one device, and the collective's backend is not what is under test: the
sharding, the per-rank HIP kernels, the packing of their outputs into the ONE all-reduce and the batch-total stopping
rule of the GMM are.  Each case runs models.*.run_inference on every rank's contiguous shard and compares with the
single-process result on all the sequences / points (svae/svae.py:33-34 consumes the summed statistics;
gmm.py:104-105 stops on the minibatch total; the SLDS ascent is per sequence, slds_svae.py:159-175)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _np(x):
    return x.detach().cpu().numpy()


def _leaves(s, out=None):
    out = [] if out is None else out
    if isinstance(s, (tuple, list)):
        for x in s:
            _leaves(x, out)
    else:
        out.append(_np(s) if hasattr(s, "detach") else np.asarray(s, float))
    return out


# ---- problems (seeded: every process rebuilds the same one) ---------------------------------------------------------

def _lds_problem():
    from oracle import expfam_numpy as ef
    from svae_amd.lds.synthetic_data import rand_node_potentials
    n, T, B, S = 6, 15, 11, 2
    rng = np.random.default_rng(21)

    def glob(scale):
        S_ = scale * (n + 2.) * np.eye(n)
        return (ef.niw_standard_to_natural(S_, 0.2 * rng.standard_normal(n), np.array(0.7), np.array(n + 2.5)),
                ef.mniw_standard_to_natural(n + 3., S_, 0.9 * np.eye(n) + 0.05 * rng.standard_normal((n, n)), 0.5 * np.eye(n)))
    prior, g = glob(1.0), glob(0.8)
    node = rand_node_potentials((B, T, n), rng, with_logZ=True)
    eps = rng.standard_normal((B, T, S, n))
    return prior, g, node, eps, S


def _gmm_problem():
    from svae_amd.lds.synthetic_data import rand_node_potentials
    from svae_amd.models import gmm
    T, N, K, S = 700, 2, 5, 1
    gen = torch.Generator().manual_seed(3)
    prior = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, generator=gen)
    glob = gmm.init_pgm_param(K, N, alpha=1.0, niw_conc=1.0, random_scale=3.0, generator=gen)
    rng = np.random.default_rng(3)
    node = rand_node_potentials((T, N), rng)
    init = rng.random((T, K))
    init /= init.sum(-1, keepdims=True)
    eps = rng.standard_normal((T, S, N))
    return prior, glob, node, init, eps, S


def _slds_problem():
    from svae_amd.lds.synthetic_data import rand_slds_global_natparam
    K, n, T, B, S = 3, 4, 14, 7, 1
    rng = np.random.default_rng(8)
    glob, prior = rand_slds_global_natparam(K, n, rng), rand_slds_global_natparam(K, n, rng)
    node = (-0.5 * (0.5 + rng.random((B, T, n))), 2. * rng.standard_normal((B, T, n)))
    return prior, glob, node, rng.standard_normal((B, T, 1, n)), rng.standard_normal((B, T, S, n)), S


def _run_all(lo_hi=None, group_on=False):
    """Every model's run_inference on the sequences / points [lo, hi) (None: all) -> dict of numpy results."""
    from svae_amd.models import gmm, lds, slds_svae
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    out = {}
    prior, g, node, eps, S = _lds_problem()
    sl = slice(None) if lo_hi is None else slice(*lo_hi(node[1].shape[0]))
    samples, stats, gkl, lkl = lds.run_inference(prior, g, tuple(t(x[sl]) for x in node), S, eps=t(eps[sl]))
    out["lds"] = dict(samples=_np(samples), stats=_leaves(stats), global_kl=float(gkl), local_kl=float(lkl))
    nJ, nh, nz = (t(x[sl]).requires_grad_(True) for x in node)
    samples, stats, gkl, lkl = lds.run_inference_differentiable(prior, g, (nJ, nh, nz), S, eps=t(eps[sl]))
    (lkl + (samples ** 2).sum()).backward()
    out["lds_diff"] = dict(stats=_leaves(stats), local_kl=float(lkl.detach()), gJ=_np(nJ.grad), gh=_np(nh.grad))

    prior, g, node, init, eps, S = _gmm_problem()
    sl = slice(None) if lo_hi is None else slice(*lo_hi(node[1].shape[0]))
    samples, stats, gkl, lkl = gmm.run_inference(prior, g, tuple(t(x[sl]) for x in node), S, label_init=t(init[sl]),
                                                 eps=t(eps[sl]))
    out["gmm"] = dict(samples=_np(samples), stats=_leaves(stats), local_kl=float(lkl))
    # the fixed point itself (labels, number of sweeps) on this shard with the batch-total rule
    (ls, _), _, _, _ = gmm.local_meanfield(g, tuple(t(x[sl]) for x in node), label_init=t(init[sl]),
                                           multi_wg=True if group_on else None)
    out["gmm"]["labels"] = _np(ls).argmax(1)

    prior, g, node, init_eps, eps, S = _slds_problem()
    sl = slice(None) if lo_hi is None else slice(*lo_hi(node[1].shape[0]))
    samples, stats, gvlb, lvlb = slds_svae.run_inference(prior, g, tuple(t(x[sl]) for x in node), S,
                                                         init_eps=t(init_eps[sl]), eps=t(eps[sl]))
    out["slds"] = dict(samples=_np(samples), stats=_leaves(stats), local_vlb=float(lvlb))
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svae_amd.parallel import shard_bounds
        res = _run_all(lambda count: shard_bounds(count, rank, world), group_on=True)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_reproduce_the_single_process_results():
    import torch.multiprocessing as mp
    from svae_amd.parallel import shard_bounds
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = _run_all()
    close = lambda a, b, tol=1e-9: np.max(np.abs(np.asarray(a) - np.asarray(b))) <= tol * max(1e-300, np.max(np.abs(np.asarray(b))))
    for rank in range(world):
        r = got[rank]
        for model, count in (("lds", 11), ("gmm", 700), ("slds", 7)):
            lo, hi = shard_bounds(count, rank, world)
            # per-sequence / per-point outputs: this rank's shard of the single-process result
            assert close(r[model]["samples"], want[model]["samples"][lo:hi]), model
            # statistics and local KL / bound: the single-process totals, on EVERY rank
            for a, b in zip(r[model]["stats"], want[model]["stats"]):
                assert close(a, b), model
            key = "local_vlb" if model == "slds" else "local_kl"
            assert abs(r[model][key] - want[model][key]) <= 1e-9 * abs(want[model][key]), model
        lo, hi = shard_bounds(11, rank, world)
        d, w = r["lds_diff"], want["lds_diff"]
        for a, b in zip(d["stats"], w["stats"]):
            assert close(a, b)
        assert abs(d["local_kl"] - w["local_kl"]) <= 1e-9 * abs(w["local_kl"])       # global value ...
        assert close(d["gJ"], w["gJ"][lo:hi], 1e-8) and close(d["gh"], w["gh"][lo:hi], 1e-8)   # ... this rank's gradient
        assert abs(r["lds"]["global_kl"] - want["lds"]["global_kl"]) <= 1e-12 * abs(want["lds"]["global_kl"])
        lo, hi = shard_bounds(700, rank, world)
        assert np.array_equal(r["gmm"]["labels"], want["gmm"]["labels"][lo:hi])      # bit-exact assignments


def test_two_streams_with_different_kernel_selections_are_independent():
    """The library is re-entrant (ABI 6: the kernel selection travels with each call; helper stream and events are
    keyed on (device, caller stream)): two training-path passes -- E-step keeping the hand-off (two kernels forked /
    joined inside the call), sampler, VJP -- on two streams with DIFFERENT selections, launched interleaved so that
    they overlap on the device, give bit for bit what each gives alone."""
    from svae_amd import _lib
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    dev = torch.device("cuda:0")
    B, T, n, S = 96, 60, 10, 2
    rng = np.random.default_rng(4)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    data = []
    for _ in range(2):
        nJ, nh = rand_node_potentials((B, T, n), rng)
        data.append(dict(args=[t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)], eps=t(rng.standard_normal((B, T, S, n))),
                         g=[t(rng.standard_normal(B)), t(rng.standard_normal((B, T, n))), t(rng.standard_normal((B, T, n))),
                            t(rng.standard_normal((B, T, S, n)))]))
    opts = [_lib.OPT_DEFAULT, _lib.OPT_TWOEND_OFF | _lib.OPT_LAYOUT_PACKED | _lib.OPT_PRODUCERS_OFF]

    def one_pass(plan, d, keep):
        plan.launch(*d["args"], None, False, keep, keep)
        outs = [plan.lognorm.clone(), plan.E_init.clone(), plan.E_pair.clone(), plan.E_node_x.clone()]
        if keep:
            smp = plan.sample(d["eps"])
            outs += [smp] + list(plan.vjp(d["g"][0], d["g"][1], d["g"][2], d["g"][3], d["eps"], smp))
        return outs

    for keep in (True, False):
        alone = []
        for i in range(2):
            alone.append(one_pass(LDSEStepPlan(B, T, n, dev, options=opts[i]), data[i], keep))
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        plans = [LDSEStepPlan(B, T, n, dev, options=opts[i]) for i in range(2)]
        torch.cuda.synchronize()
        both = [None, None]
        for rep in range(3):                                   # interleaved: the two streams' work overlaps
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    both[i] = one_pass(plans[i], data[i], keep)
        torch.cuda.synchronize()
        for i in range(2):
            for a, b in zip(both[i], alone[i]):
                assert torch.equal(a, b)


# ---- one-shot mailbox all-reduce over IPC-mapped memory (svae_amd/ipc.py, csrc/ipc_allreduce.hip) ---------------------

def _ipc_worker(rank, world, port, q, n, rounds):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svae_amd.ipc import MailboxAllReduce
        dev = torch.device("cuda:0")
        ar = MailboxAllReduce(n, device=dev)
        gen = torch.Generator(device=dev).manual_seed(100 + rank)
        worst, same = 0.0, True
        for it in range(rounds):
            x = torch.randn(n, dtype=torch.float64, device=dev, generator=gen) * (10.0 ** (it % 7 - 3))
            want = x.clone().cpu()
            dist.all_reduce(want)                                  # gloo on the host: the reference sum
            got = ar(x.clone())
            if it % 5 == 0:                                        # skew the ranks: one runs ahead of the other
                torch.cuda.synchronize()
            worst = max(worst, float((got.cpu() - want).abs().max() / want.abs().max()))
            # every rank must hold the SAME bits
            bits = [None] * world
            dist.all_gather_object(bits, got.cpu().numpy().tobytes())
            same = same and all(b == bits[0] for b in bits)
        # a shorter buffer through the same mailbox, and the status word
        y = torch.full((7,), float(rank + 1), dtype=torch.float64, device=dev)
        ar(y)
        ar.check()
        # calls of ALTERNATING lengths back to back, no host collective in between, one rank held back before every
        # call in turn: a rank runs a call ahead of a peer that has not read the previous one yet (round 4 laid the
        # parity sets out with the call's length: they overlapped).  Inputs are a function of (call, rank), so every rank
        # knows the exact rank-order sum.
        import time
        lengths = [7, n, 33, max(1, n // 2), n, 1]
        mk = lambda it, r, L: torch.randn(L, dtype=torch.float64, generator=torch.Generator().manual_seed(7919 * it + r))
        outs = []
        for it in range(36):
            L = lengths[it % len(lengths)]
            if it % world == rank:
                time.sleep(0.003)
            outs.append(ar(mk(it, rank, L).to(dev)))
        torch.cuda.synchronize()
        ar.check()
        alt_ok = True
        for it, got in enumerate(outs):
            want = torch.zeros(got.numel(), dtype=torch.float64)
            for r in range(world):
                want = want + mk(it, r, got.numel())
            alt_ok = alt_ok and torch.equal(got.cpu(), want)
        # a peer that never shows up: NaN, not a plausible sum, and the status word says so
        dist.barrier()
        timed_out = None
        if rank == 0:
            ar.spin_limit = 3000
            z = ar(torch.ones(5, dtype=torch.float64, device=dev))
            torch.cuda.synchronize()
            try:
                ar.check()
                raised = False
            except RuntimeError:
                raised = True
            timed_out = bool(torch.isnan(z).all()) and raised
        dist.barrier()
        q.put((rank, worst, same, y.cpu().tolist(), alt_ok, timed_out))
        ar.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_mailbox_allreduce_between_processes_on_one_gpu(world):
    """The one-shot all-reduce of the packed statistics (4 n^2 + n + 2 = 412 doubles at n = 10; here 415 and a ragged
    1030) over fine-grained, IPC-mapped mailboxes: `world` processes share the one MI355X of the test box -- the IPC
    mapping, the tagged-word protocol with its parity double-buffering (ranks deliberately skewed) and the rank-order sum
    are what is under test; xGMI is not.  Against gloo's all-reduce on the host: equal to rounding, and the SAME bits on
    every rank; then calls of alternating lengths under rank skew (bit-exact rank-order sums), and a call a peer never
    joins (NaN + status word, never a plausible sum)."""
    import torch.multiprocessing as mp
    for n, rounds in ((415, 60), (1030, 12)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, q, n, rounds)) for r in range(world)]
        for p_ in procs:
            p_.start()
        res, waited = [], 0.0
        while len(res) < world:                      # fail fast if a worker dies (never sit out the queue's timeout)
            try:
                res.append(q.get(timeout=1.0))
            except Exception:
                waited += 1.0
                dead = [p_.exitcode for p_ in procs if p_.exitcode not in (None, 0)]
                if dead or waited > 240:
                    for p_ in procs:
                        p_.kill()
                    pytest.fail("mailbox all-reduce worker failed (exit codes %r, waited %.0f s)" % (dead, waited))
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
        for rank, worst, same, y, alt_ok, timed_out in res:
            assert worst < 1e-15 * world, (rank, worst)
            assert same
            assert y == [float(world * (world + 1) // 2)] * 7
            assert alt_ok, "alternating lengths under rank skew: wrong sum on rank %d" % rank
            assert timed_out in (None, True), "a missing peer must give NaN and a raised status word"


# ---- RCCL on the device path, single rank (the only RCCL configuration a 1-GPU box can run) ------------------------------

def _rccl_single_rank_worker(port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from svae_amd.lds.lds_inference import LDSEStepPlan
        from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
        from svae_amd.parallel import allreduce_global_stats, allreduce_lds_stats
        B, T, n = 37, 20, 10
        rng = np.random.default_rng(0)
        init, pair = rand_lds_natparam(n, rng)
        nJ, nh = rand_node_potentials((B, T, n), rng)
        t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
        plan = LDSEStepPlan(B, T, n, dev)
        plan.launch(t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1),
                    t(nJ), t(nh), None)
        reduced = plan.reduce().clone()
        # the product's exchange step with the product's backend: dist.all_reduce on the packed DEVICE buffer through RCCL
        # (world size 1: the sum is the identity -- what is exercised is the backend's device path, launch and stream
        # ordering behind the library's kernels, not a reduction)
        packed = reduced.clone()
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        out = dist.all_reduce(packed)                      # the collective itself, directly (allreduce_global_stats
        torch.cuda.synchronize()                           #  short-circuits at world size 1)
        same = bool(torch.equal(packed, reduced))
        allreduce_global_stats(packed)
        lkl = torch.tensor(3.25, dtype=torch.float64, device=dev)
        niw_stats, mniw_stats, kl = allreduce_lds_stats(reduced, lkl, n, T)
        ok = same and float(kl) == 3.25 and bool(torch.equal(mniw_stats[1], reduced[n * n + n + n * n:n * n + n + 2 * n * n].reshape(n, n)))
        q.put(("ok" if ok else "mismatch", dist.get_backend()))
    except Exception as e:          # pragma: no cover
        q.put(("error: %r" % (e,), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_allreduce_of_the_packed_statistics_single_rank_on_the_device():
    """`dist.all_reduce` of the packed statistics buffer through RCCL (backend "nccl") on the GPU, one rank: the
    configuration of the product's collective that a one-GPU box can execute (tools/rccl_single_rank_check.py promoted
    into the suite; RCCL refuses two ranks on one device, so the multi-rank GPU tests above use gloo)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), q))
    p_.start()
    try:
        status, backend = q.get(timeout=240)
    finally:
        p_.join(timeout=60)
        if p_.is_alive():
            p_.kill()
    assert status == "ok" and backend == "nccl", status


def test_bench_gpus_2_without_a_launcher_runs_two_ranks():
    """`python bench.py --gpus 2` (no torchrun): bench.py launches its own two ranks and the line says so -- here both
    on the one GPU of the box through gloo (SVAE_BENCH_BACKEND; RCCL refuses ranks that share a device, and bench.py
    refuses RCCL with fewer devices than ranks instead of running one rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVAE_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert got["n_gpus"] == 2 and got["world_size_observed"] == 2 and len(got["rank_devices"]) == 2
    assert got["config"]["global_sequences"] == 2 * got["config"]["sequences_per_gpu"]
    assert got["parity"]["ok"]
    if torch.cuda.device_count() < 2:
        env["SVAE_BENCH_BACKEND"] = "nccl"
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--no-extra"],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)
