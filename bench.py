#!/usr/bin/env python
"""bench.py -- E-step sequences/sec of the LDS forward-backward smoother (T=200, n=10) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One JSON line on rank 0.

A "step" is one pass of the hot path over one batch of synthetic sequences resident in HBM:
  svae_lds_estep_f64 (filter + smoother + expected statistics + log-normaliser, per sequence)
  -> svae_lds_reduce_stats_f64 (deterministic batch sum of the global statistics)
  -> [N > 1] one RCCL all-reduce (sum, fp64) of the packed 4n^2+n+2 global statistics.
Workload = BASELINE.json configs[1]: 512 sequences x T=200, n=10 per GPU (weak scaling: N=8 is
configs[2], 4096 sequences sharded 8 x 512 with the stat all-reduce).

Extra JSON objects: `roofline` (dominant kernel vs HBM peak, duration measured live with events
on the launch stream), at N=1 on rank 0 `cpu_baseline` (the reference's own compiled E-step,
oracle/_ref, timed on this box's host cores on a bounded sample of the same workload, plus the
NumPy restatement of its Python path), and at N=1 `extra`: the same measurement for north_star's
single-GPU target (4096 sequences on ONE GPU) and for BASELINE configs[4]'s shape (latent dim 64,
T=1000), each with its own roofline, plus the training path at the headline shape (E-step keeping the
hand-off + sampler + VJP), BASELINE configs[3] (SLDS local mean field) and configs[0] (GMM fixed point)
-- reported beside `value`, never instead of it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x (16x16x4x2 flop / 64 clk, tools/ubench/mfma_f64.hip) x 2.4 GHz
# --workload: the headline (BASELINE configs[1]/[2]) or BASELINE configs[4] (latent dim 64, T=1000; the
# batch is not specified there: 512 sequences per GPU = two workgroups per CU)
WORKLOADS = {"lds10": (200, 10, 512), "lds64": (1000, 64, 512)}
# committed rocprofv3 PMC passes (profiles/run_profile.sh), by (kernel family, sequences per GPU)
PMC_PROFILES = {("twoend", 512): "r6_twoend", ("twoend_rpc", 4096): "r6_twoend_b4096",
                ("split", 512): "r1_final", ("packed", 4096): "r1_final_b4096", ("tile", 512): "r6_tile_n64_b512"}


def algorithmic_bytes_per_seq(T, n):
    """read node_J,node_h,node_logZ + write E_node (diag, x) + per-sequence
    E_init, E_pair sums, lognorm."""
    return 8 * (T * (2 * n + 1) + 2 * T * n + (n * n + n) + 3 * n * n + 1)


def algorithmic_flops_per_seq(T, n):
    """T (35/3) n^3 (filter: potrf + trsm + gemm; RTS: potrf, trsm, potri, gemms)."""
    return T * (35.0 / 3.0) * n ** 3


def cpu_baseline(B, T, n, budget_s=10.0):
    """Reference CPU path on a bounded sample of the same workload, in a SEPARATE process (a forked
    pool must not inherit an initialised HIP runtime): see oracle/cpu_baseline.py."""
    import subprocess
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--B", str(B), "--T", str(T),
                        "--n", str(n), "--budget", str(budget_s)], cwd=ROOT, capture_output=True,
                       text=True, timeout=30 * budget_s + 120)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-400:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def kernel_name(options, B, T, n):
    """Which E-step kernel the library dispatches to for this `options` word (include/svae_hip.h: SVAE_OPT_*;
    the dispatch is a pure function of the call's arguments) -> (profile family, demangled name as rocprofv3
    prints it)."""
    from svae_amd import _lib
    if n > _lib.LDS_MAX_N:
        # (two register budgets: up to one workgroup per CU the instance without spills, csrc/lds_estep_tile.hip)
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        nb = (n + 15) // 16
        if B <= cus:
            return "tile", "svae::lds_estep_tile_kernel<%d,false,1,0>" % nb
        # beyond one workgroup per CU: two launches of single-half instances (forward half, backward half)
        return "tile", "svae::lds_estep_tile_kernel<%d,false,2,1> + svae::lds_estep_tile_kernel<%d,false,2,2>" % (nb, nb)
    twoend = 0 if options & _lib.OPT_TWOEND_OFF else (2 if options & _lib.OPT_TWOEND_FULL else 1)
    split_max = (1 << 30) if options & _lib.OPT_LAYOUT_SPLIT else (0 if options & _lib.OPT_LAYOUT_PACKED else 1023)
    if twoend and n <= 10 and T >= 4:
        layout = 1 if options & _lib.OPT_LAYOUT_SPLIT else (2 if options & _lib.OPT_LAYOUT_PACKED else 0)
        if twoend == 1 and (layout == 2 or (layout == 0 and B >= 1025)):    # TE_RPC_MIN_B (csrc/lds_args.hpp)
            return "twoend_rpc", "svae::lds_estep_twoend_rpc_kernel<%d>" % n
        if twoend == 1 and B <= 512:     # TE_S4_MAX_B (csrc/lds_args.hpp): one chain per wavefront in the smoother phase
            return "twoend", "svae::lds_estep_twoend_kernel<%d,false,true,false,false,true>" % n
        return "twoend", "svae::lds_estep_twoend_kernel<%d,false,%s>" % (n, "true" if twoend == 1 else "false")
    if B <= split_max:
        return "split", "svae::lds_estep_split_kernel<%d,false,false>" % n
    return "packed", "svae::lds_estep_kernel<%d,false,false>" % n


PARITY_TOL = 1e-6        # the gate: bench.py exits non-zero if a timed output is further than this from the reference
PARITY_TOL_ELEMENTWISE = 1e-5   # north_star's "1e-5 relative", as a plain element-wise |a-b|/|b| (no absolute floor; entries above 1e-12 of the array maximum)
PARITY_SEQUENCES = 8


def snapshot_for_parity(plan, natparam, d_J, d_h, count=PARITY_SEQUENCES):
    """`count` sequences spread over the batch: the global parameters, their node potentials as they sit in HBM and
    what the timed plan's LAST launch left for them -> dict of NumPy arrays for oracle/bench_parity.py."""
    init, pair = natparam
    idx = np.unique(np.linspace(0, plan.B - 1, min(count, plan.B)).astype(int))
    ix = torch.as_tensor(idx, device=plan.device)
    g = lambda x: x.index_select(0, ix).cpu().numpy()
    return dict(index=idx, init_J=np.asarray(init[0], float), init_h=np.asarray(init[1], float),
                init_logZ=np.asarray(init[2], float), J11=np.asarray(pair[0], float), J12=np.asarray(pair[1], float),
                J22=np.asarray(pair[2], float), logZ_pair=np.asarray(pair[3], float),
                node_J=g(d_J), node_h=g(d_h), lognorm=g(plan.lognorm), E_init=g(plan.E_init), E_pair=g(plan.E_pair),
                E_node_diagxx=g(plan.E_node_diagxx), E_node_x=g(plan.E_node_x))


def parity_gate(snapshot, tol=PARITY_TOL):
    """the timed plan's outputs against the reference's compiled E-step, in a separate
    process (oracle/bench_parity.py: the checker, never the thing measured) -> {"max_rel", ..., "tol", "ok"}."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(prefix="svae_bench_parity_") as tmp:
        path = os.path.join(tmp, "snapshot.npz")
        np.savez(path, **snapshot)
        r = subprocess.run([sys.executable, "-m", "oracle.bench_parity", path], cwd=ROOT, capture_output=True,
                           text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("oracle.bench_parity failed: " + r.stderr[-400:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["tol"] = tol
    out["tol_elementwise"] = PARITY_TOL_ELEMENTWISE
    out["ok"] = bool(out["max_rel"] < tol and out.get("max_rel_elementwise", 0.0) < PARITY_TOL_ELEMENTWISE)
    return out


def measure(dev, rank, world, dist, options, T, n, B, steps, warmup, parity=0):
    """W untimed + K timed steps of the hot path on B sequences per GPU -> (elapsed s [max over ranks],
    mean kernel ms from events on the launch stream[, parity snapshot of the timed plan's outputs])."""
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_node_potentials
    from svae_amd.parallel import allreduce_global_stats
    init, pair = bench_natparam(n)                                    # replicated global params
    node_J, node_h = rand_node_potentials((B, T, n), np.random.default_rng(1000 + rank))  # this rank's shard
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    d_init = [t(init[0]), t(init[1]), t(init[2]).reshape(1)]
    d_pair = [t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1)]
    d_J, d_h = t(node_J), t(node_h)
    del node_J, node_h
    plan = LDSEStepPlan(B, T, n, dev, options=options)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        plan.launch(d_init[0], d_init[1], d_init[2], d_pair[0], d_pair[1], d_pair[2], d_pair[3],
                    d_J, d_h, None)
        if i is not None:
            ev[i][1].record()
        packed = plan.reduce()
        if world > 1:
            allreduce_global_stats(packed)

    # Clock spin-up, untimed and outside the driver's warm-up count: a 20-step run of a 0.15 ms kernel is over in 3 ms,
    # before the shader clock has left its idle state (measured: 148.6 us per launch in such a run, 142 us in a 200-step
    # one) -- >= 60 ms of the same launches first.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.06:
        for _ in range(8):
            plan.launch(d_init[0], d_init[1], d_init[2], d_pair[0], d_pair[1], d_pair[2], d_pair[3], d_J, d_h, None)
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    plan.check_info()
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, steps)
    if parity:
        return elapsed, kern_ms, snapshot_for_parity(plan, (init, pair), d_J, d_h, parity)
    return elapsed, kern_ms


def tile_parity_on_rand_lds(dev, T=1000, n=64, B=2):
    """Latent dim 64 is TIMED on the well-conditioned rotation model (bench_natparam); this is the parity distance of
    the same kernels on the reference's own generator `rand_lds` (svae/lds/synthetic_data.py:8-28), whose state noise
    has cond ~ n^2 -- there the reference's compiled fp64 path is itself 2.5e-6 .. 3e-5 away from extended precision
    (oracle/lds_longdouble.py; tests/test_lds_tile_hip.py pins kernel-vs-reference <= reference-vs-arbiter + 5e-6)."""
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(0)
    init, pair = rand_lds_natparam(n, rng)
    node_J, node_h = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    plan = LDSEStepPlan(B, T, n, dev)
    d_J, d_h = t(node_J), t(node_h)
    plan.launch(t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1),
                d_J, d_h, None)
    torch.cuda.synchronize()
    plan.check_info()
    out = parity_gate(snapshot_for_parity(plan, (init, pair), d_J, d_h, B), tol=1e-5)
    out["model"] = "rand_lds(n=%d), T=%d, %d sequences (NOT the timed model)" % (n, T, B)
    return out


def bench_natparam(n):
    """Global parameters of the bench workloads (also what oracle/cpu_baseline.py times).  n <= 15: the reference's
    own generator `rand_lds` (svae/lds/synthetic_data.py:8-28).  Latent dim 64: the well-conditioned rotation model
    -- `rand_lds`'s state noise B B' has cond ~ n^2, where the reference's own fp64 path is only good to ~1e-5 at
    T = 1000; on this one the tile kernel is pinned against the reference's compiled path at 1e-8 at full size
    (tests/test_lds_tile_hip.py::test_tile_estep_full_size_well_conditioned).  Kernel time does not depend on it."""
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rotation_lds_natparam
    rng = np.random.default_rng(0)
    return rotation_lds_natparam(n, rng) if n > 15 else rand_lds_natparam(n, rng)


def roofline(options, T, n, B, kern_ms):
    from svae_amd import _lib
    family, kernel = kernel_name(options, B, T, n)
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (separate runs of this same
    # command, profiles/run_profile.sh); only quoted when kernel and workload match the profile.
    traffic, traffic_src, p = None, None, {}
    tag = PMC_PROFILES.get((family, B))
    prof = os.path.join(ROOT, "profiles", tag, "pmc_hbm.json") if tag else None
    if prof and os.path.isfile(prof) and (T, n) in ((200, 10), (1000, 64)):
        p = json.load(open(prof))
        if p.get("csrc_sha16") != _lib.source_hash():
            # the counters were collected on other kernel sources than the ones timed here: not quoted (round 5)
            traffic_src = "%s is STALE (collected on csrc %s, this build is %s): not quoted" % (
                os.path.relpath(prof, ROOT), p.get("csrc_sha16"), _lib.source_hash())
            p = {}
        elif all(k.replace(" ", "") in p.get("kernel", "").replace(" ", "") for k in kernel.split(" + ")):
            traffic, traffic_src = p["hbm_bytes_per_launch_corrected"], os.path.relpath(prof, ROOT)
    bytes_launch = B * algorithmic_bytes_per_seq(T, n)
    achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
    # The honest bound at n <= 15 is fp64 vector issue (docs/DESIGN_HISTORY.md 3.5), reported beside the HBM figure north_star asks
    # for: algorithmic flops (T (35/3) n^3 per sequence) over the kernel time against the vector fp64 FMA
    # peak (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz = the MFMA fp64 peak), and how many flops the kernel ISSUES
    # for each algorithmic one (SQ_INSTS_VALU of the committed PMC pass x 64 lanes x 2; an upper bound: not every
    # VALU instruction is an FMA).
    flops_launch = B * algorithmic_flops_per_seq(T, n)
    tf = flops_launch / (kern_ms * 1e-3) / 1e12
    valu_insts = p.get("SQ_INSTS_VALU_per_launch_mean") if traffic is not None else None
    valu = {"bound": "fp64 valu issue", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops_launch,
            "issued_over_algorithmic": (valu_insts * 128.0 / flops_launch) if valu_insts else None}
    hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
           "traffic_unit": "bytes per launch (rocprofv3 PMC, 2*FETCH_SIZE+WRITE_SIZE; separate passes of this same "
                           "command, quoted only if the profile's source hash equals this build's)",
           "traffic_source": traffic_src,
           "traffic_over_algorithmic": (traffic / bytes_launch) if traffic else None,
           "kernel": kernel, "kernel_ms": kern_ms,
           "algorithmic_bytes_per_launch": bytes_launch,
           "kernel_sequences_per_s": B / (kern_ms * 1e-3)}
    if n > _lib.LDS_MAX_N:      # dense contraction: priced against the fp64 MFMA peak
        return dict(hbm, bound="mfma", achieved=tf, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=tf / FP64_MFMA_PEAK_TFLOPS, algorithmic_flops_per_launch=flops_launch,
                    hbm_algorithmic_GBps=achieved)
    hbm["valu"] = valu
    return hbm


def measure_training_path(dev, T, n, B, S=1, reps=5, options=None):
    """E-step keeping what the VJP needs + backward sampler (ONE call: svae_lds_inference_f64, the reference's
    cython_natural_lds_inference_general) + VJP at the headline shape (what one SVAE training step adds around the
    recognition / decoder networks): ms per pass, events on the launch stream.  Above 2048 sequences the library keeps
    lean per-step records (csrc/lds_lean_estep.hpp); `record_format` says which format the timed calls used."""
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(0)
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
    eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)
    g = [torch.randn(B, dtype=torch.float64, device=dev), torch.randn(B, T, n, dtype=torch.float64, device=dev),
         torch.randn(B, T, n, dtype=torch.float64, device=dev), torch.randn(B, T, S, n, dtype=torch.float64, device=dev)]
    plan = LDSEStepPlan(B, T, n, dev, options=options)
    smp = torch.empty_like(eps)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    acc = [0.0, 0.0]
    for rep in range(reps + 1):
        ev[0].record(); plan.infer(*args, None, False, eps, smp)
        ev[1].record(); plan.vjp(g[0], g[1], g[2], g[3], eps, smp)
        ev[2].record(); torch.cuda.synchronize()
        if rep:
            for i in range(2):
                acc[i] += ev[i].elapsed_time(ev[i + 1]) / reps
    plan.check_info()
    return {"workload": "training path at the headline shape: E-step keeping the VJP's records + backward sampler (one "
                        "call) + VJP, %d sequences x T=%d, n=%d, %d sample" % (B, T, n, S),
            "record_format": "lean" if plan.lean else "full",
            "estep_and_sampler_ms": acc[0], "vjp_ms": acc[1], "ms_per_pass": sum(acc),
            "value": B / sum(acc) * 1e3, "unit": "sequences/s"}


def measure_tile_training(dev, B=64, T=1000, n=64, S=1):
    """BASELINE configs[4] shape through a whole training-path pass (tile-kernel E-step, sampler, VJP kernels)."""
    from svae_amd.lds.lds_inference import LDSEStepPlan, lds_inference_differentiable
    from svae_amd.lds.synthetic_data import rand_node_potentials
    rng = np.random.default_rng(0)
    init, pair = bench_natparam(n)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    nJ, nh = t(nJ).requires_grad_(True), t(nh).requires_grad_(True)
    eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)
    gs = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)
    plan = LDSEStepPlan(B, T, n, dev)

    def it():
        lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(natparam, (nJ, nh), eps=eps, plan=plan)
        loss = lognorm.sum() + (dxx * 0.3).sum() + ex.sum() + (samples * gs).sum()
        return torch.autograd.grad(loss, [nJ, nh])
    it(); it(); torch.cuda.synchronize()     # (two: the pass alternates between two workspace blocks of the caching allocator)
    times = []
    for _ in range(5):                       # each pass timed on its own, the median reported (a pass is ~30 launches on
        t0 = time.perf_counter()             #  three streams: one allocator or scheduling hiccup would double a 2-pass mean)
        it()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = sorted(times)[len(times) // 2]
    return {"workload": "training path at BASELINE configs[4] shape: tile-kernel E-step + sampler + VJP kernels, "
                        "%d sequences x T=%d, n=%d, %d sample" % (B, T, n, S),
            "ms_per_pass": ms, "ms_per_pass_all": [round(x, 2) for x in times], "value": B / ms * 1e3,
            "unit": "sequences/s"}


def measure_slds(dev, B=2048, T=500, n=10, K=8):
    """BASELINE configs[3]: SLDS-SVAE local mean field (coordinate ascent between the HMM kernel and the fused LDS
    mean-field kernel), wall clock of the whole ascent."""
    from svae_amd.lds.synthetic_data import rand_slds_global_natparam
    from svae_amd.models import slds_svae
    rng = np.random.default_rng(0)
    glob = rand_slds_global_natparam(K, n, rng)
    # (the global parameters live on the device, as in a training loop: the global -> local maps then run as kernels)
    _d = lambda x: tuple(_d(y) for y in x) if isinstance(x, (tuple, list)) else torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    glob = _d(glob)
    node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
            torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
    eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    best = None
    for rep in range(5):        # (best of five: the first repetitions behind an empty_cache() also pay for the allocator's warm-up)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, _, _, iters = slds_svae.optimize_local_meanfield(glob, node, eps, pair_stats=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # ... and the whole local step of the model around it (run_inference, slds_svae.py:289-310: ascent + final LDS E-step on
    # the per-step parameters + sampler + HMM bound + global statistics), forward values
    prior = _d(rand_slds_global_natparam(K, n, rng))
    eps2 = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    best_ri = None
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = slds_svae.run_inference(prior, glob, node, 1, init_eps=eps, eps=eps2)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best_ri = dt if best_ri is None else min(best_ri, dt)
    del out
    return {"workload": "BASELINE configs[3]: SLDS-SVAE local mean field, K=%d, n=%d, %d sequences x T=%d" % (K, n, B, T),
            "ms_per_ascent": 1e3 * best, "ms_per_run_inference": 1e3 * best_ri,
            "sweeps_max": int(iters.max()), "sweeps_mean": float(iters.double().mean()),
            "value": B / best, "unit": "sequences/s",
            "kernel": "svae::slds_meanfield_rpc_kernel<10,false> (row-per-chain consumers + MFMA producer wavefronts) + svae::hmm_estep2_kernel<8>"}


def measure_gmm(dev, K=5, N=2, T=1000, what="BASELINE configs[0]"):
    """GMM mean-field fixed point: us per fixed point.  BASELINE configs[0] says K = 5, 2-D, 1000 points; the script the
    reference ships (experiments/gmm_svae_synth.py:29-33) runs K = 15 on 500 points -- SURVEY 8d: both, labelled."""
    from svae_amd.distributions import expfam
    from svae_amd.models import gmm
    gen = torch.Generator().manual_seed(K)
    d, niws = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen)
    lg, gg = expfam.dirichlet_expectedstats(d).to(dev), expfam.niw_expectedstats(niws).to(dev)
    rng = np.random.default_rng(0)
    node = (torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev),
            torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev))
    init = gmm.initialize_meanfield(T, K, dev, torch.Generator(device=dev).manual_seed(1))
    o = gmm.meanfield_from_globals(lg, gg, node, init)
    torch.cuda.synchronize()
    if T <= gmm.GMM_PERSISTENT_MAX_T and o["path"] != "persistent":
        raise RuntimeError("GMM fixed point ran on path %r, expected the persistent kernel" % o["path"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        o = gmm.meanfield_from_globals(lg, gg, node, init, check=False)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    return {"workload": "%s: GMM mean-field fixed point, K=%d, %d-D, %d points" % (what, K, N, T),
            "us_per_fixed_point": us, "sweeps": int(o["iters"]), "path": o["path"], "value": T / us * 1e6, "unit": "points/s"}


def measure_gmm_training(dev, K=5, N=2, T=1000, S=1):
    """BASELINE configs[0] through the whole differentiable local step (gmm.run_inference_differentiable + backward):
    global maps, fixed point + final pass, sampler, and the derived adjoint kernel -- us per step, wall clock."""
    from svae_amd.models import gmm
    gen = torch.Generator().manual_seed(K)
    prior = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, generator=gen)
    glob = tuple(x.to(dev) for x in gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen))
    prior = tuple(x.to(dev) for x in prior)
    rng = np.random.default_rng(0)
    nJ = torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev).requires_grad_(True)
    nh = torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev).requires_grad_(True)
    init = gmm.initialize_meanfield(T, K, dev, torch.Generator(device=dev).manual_seed(1))
    eps = torch.randn(T, S, N, dtype=torch.float64, device=dev)
    gs = torch.randn(T, S, N, dtype=torch.float64, device=dev)

    def it(check=True):
        samples, stats, gkl, lkl = gmm.run_inference_differentiable(prior, glob, (nJ, nh), S, label_init=init, eps=eps,
                                                                     check=check)
        return torch.autograd.grad(lkl + (samples * gs).sum(), [nJ, nh])
    it(); it(); torch.cuda.synchronize()

    def median_us(fn):
        times = []
        for _ in range(9):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e6)
        return sorted(times)[len(times) // 2]
    us = median_us(it)
    out = {"workload": "BASELINE configs[0], training step of the local model: global maps + fixed point + final pass + "
                       "sampler + adjoint kernel, K=%d, %d-D, %d points, %d sample" % (K, N, T, S),
           "us_per_step": us, "value": T / us * 1e6, "unit": "points/s"}
    # the same step without the host's status read (check=False; gmm.check_info() afterwards), replayed as ONE hipGraph
    try:
        ref = [x.clone() for x in it()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            it(False); it(False)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            res = it(False)
        graph.replay(); torch.cuda.synchronize()
        gmm.check_info()
        out["graph_matches_eager"] = bool(all(torch.equal(a, b) for a, b in zip(res, ref)))
        out["us_per_step_graph"] = median_us(graph.replay)
    except Exception as e:                                   # pragma: no cover - reported, not fatal
        out["graph_error"] = repr(e)[:200]
    return out


def measure_gradfun_step(dev, B=512, T=200, n=10, p=20, hidden=32, reps=9):
    """One END-TO-END training step of the LDS-SVAE at the headline shape (BASELINE configs[1]: latent dim 10, obs dim 20,
    512 sequences x T=200), driven through svae_amd.svae.make_gradfun (/root/reference/svae/svae.py:10-39): recognition
    MLPs (stock torch) -> global-step kernel -> E-step keeping the hand-off -> sampler -> decoder MLP + ELBO -> autograd
    backward through the VJP kernels -> natural-gradient kernel.  Wall clock per step, eager and -- if the whole step
    captures -- replayed as ONE hipGraph (torch.cuda.CUDAGraph: the library's launches, incl. its helper-stream
    fork / joins, are recorded from the capturing stream)."""
    import functools
    from svae_amd import svae
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.models import lds
    gen = torch.Generator(device=dev).manual_seed(0)
    data = torch.randn(B, T, p, dtype=torch.float64, device=dev, generator=gen)
    prior = tuple(x.to(dev) if isinstance(x, torch.Tensor) else tuple(y.to(dev) for y in x) for x in lds.make_prior_natparam(n))
    pgm = tuple(x.clone() if isinstance(x, torch.Tensor) else tuple(y.clone() for y in x) for x in prior)

    # (svae_amd.nnet.linear: x @ w whose weight gradient avoids rocBLAS's fp64 path for a 10^5-row reduction axis, 11 ms
    #  per layer at this shape; the networks themselves are stock torch, out of the library's scope)
    #  The layer form is the reference's: nonlin(x W + b) stacks (nnet.py:23), the recognition network ONE MLP whose
    #  2 n outputs are split into (J_input, h) (nnet.py:43-47).
    from svae_amd.nnet import gaussian_info, init_mlp, tanh_mlp
    recogn, decoder = init_mlp([p, hidden, 2 * n], device=dev, generator=gen), init_mlp([n, hidden, p], device=dev, generator=gen)
    recognize = gaussian_info
    loglike = lambda params, samples, batch: -0.5 * ((batch.unsqueeze(2) - tanh_mlp(params, samples)) ** 2).sum() / samples.shape[2]
    plan = LDSEStepPlan(B, T, n, dev)
    eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=gen)
    run = functools.partial(lds.run_inference_differentiable, plan=plan, eps=eps)
    gradfun = svae.make_gradfun(run, recognize, loglike, prior, data, B, 1, natgrad_scale=1e2, callback=None, permute=False)
    params = (pgm, decoder, recogn)
    for _ in range(3):
        gradfun(params, 0)
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter(); gradfun(params, 0); torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    eager = sorted(times)[len(times) // 2]
    out = {"workload": "end-to-end LDS-SVAE training step through make_gradfun: recognition MLP (layers with biases, one network split into (J, h) as svae/nnet.py) + global step + E-step + "
                       "sampler + decoder + backward (VJP kernels) + natural gradient, %d sequences x T=%d, n=%d, obs dim %d"
                       % (B, T, n, p),
           "eager_ms_per_step": eager, "graph_ms_per_step": None, "graph_error": None,
           "value": B / eager * 1e3, "unit": "sequences/s"}
    try:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                gradfun(params, 0)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = gradfun(params, 0)
        graph.replay(); torch.cuda.synchronize()
        eager_out = gradfun(params, 0)
        torch.cuda.synchronize()
        same = all(torch.allclose(a, b, rtol=1e-12, atol=0) for a, b in zip(svae._leaves(captured), svae._leaves(eager_out)))
        times = []
        for _ in range(reps):
            t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        out["graph_ms_per_step"] = sorted(times)[len(times) // 2]
        out["graph_matches_eager"] = bool(same)
        out["value"] = B / out["graph_ms_per_step"] * 1e3
    except Exception as e:      # the eager number stands on its own
        out["graph_error"] = repr(e)[:300]
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
    return out


def socket_hostname():
    import socket
    return socket.gethostname()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="lds10")
    ap.add_argument("--seqs-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra single-GPU configurations")
    ap.add_argument("--kernel", choices=["auto", "twoend", "twoend_full", "split", "packed"], default="auto",
                    help="A/B measurements: force one of the E-step kernels (n <= 15)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring up the ranks (rendezvous, barrier, all-gather of the device ids), print the launch "
                         "facts as one JSON line and exit: what tests/test_distributed.py runs on CPU with gloo")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    backend = os.environ.get("SVAE_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, as the contract's
        # torch.distributed.run line) instead of printing an N = 1 number under an N > 1 label.
        if backend == "nccl" and torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d but only %d GPU(s) visible (RCCL needs one device per rank; "
                             "SVAE_BENCH_BACKEND=gloo rehearses the control flow on fewer)" % (args.gpus, torch.cuda.device_count()))
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not args.launch_check:
        raise SystemExit("bench.py needs a GPU (only --launch-check runs without one)")
    if backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
        raise SystemExit("%d ranks on %d visible GPU(s): RCCL needs one device per rank" % (world, torch.cuda.device_count()))
    # (modulo: a rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks, see SVAE_BENCH_BACKEND)
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count())) if have_gpu else torch.device("cpu")
    if have_gpu:
        torch.cuda.set_device(dev)
    dist = None
    launch = {"world_size_observed": 1, "rank_devices": [str(dev)], "backend": None,
              "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else ("env" if "WORLD_SIZE" in os.environ else "single process")}
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl") is the product path.  SVAE_BENCH_BACKEND=gloo only exists to rehearse the multi-rank
        # control flow (barriers, max-over-ranks timing, the packed all-reduce) on ONE GPU shared by the ranks,
        # where RCCL refuses duplicate devices; such a run is not a scaling measurement.
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # launch facts, checked: the line below can only ever carry the number of ranks that really ran
        if dist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
        ids = [None] * world
        dist.all_gather_object(ids, "%s:%s" % (socket_hostname(), dev))
        launch.update(world_size_observed=dist.get_world_size(), rank_devices=ids, backend=dist.get_backend())
        if backend == "nccl" and len(set(ids)) != world:
            raise SystemExit("RCCL ranks share a device: %r" % (ids,))
    if args.launch_check:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            print(json.dumps(dict(launch, launch_check=True, n_gpus=world, gpus_flag=args.gpus)), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    from svae_amd import _lib
    _lib.load()                                   # no library, no bench: there is no fallback path
    if world > 1 and os.environ.get("SVAE_BENCH_ALLREDUCE", "") == "mailbox":
        # opt-in A/B: the exchange step through the one-shot IPC mailbox kernel instead of the collective backend
        from svae_amd.parallel import use_mailbox_allreduce
        use_mailbox_allreduce(4 * 64 * 64 + 64 + 2)
    options = _lib.KERNEL_OPTIONS[args.kernel]    # per-call selection word (A/B measurements); "auto" = 0
    T, n, B = WORKLOADS[args.workload]
    B = args.seqs_per_gpu or B

    elapsed, kern_ms, snap = measure(dev, rank, world, dist, options, T, n, B, args.steps, args.warmup, parity=PARITY_SEQUENCES)
    parity_failed = False

    if rank == 0:
        total_seqs = B * world * args.steps
        out = {
            "metric": "E-step sequences/sec (LDS fwd-bwd smoother, T=%d n=%d)" % (T, n),
            "value": total_seqs / elapsed, "unit": "sequences/s",
            "n_gpus": world, "world_size_observed": launch["world_size_observed"], "rank_devices": launch["rank_devices"],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "LDS-SVAE E-step, latent dim %d, T=%d, %d sequences per GPU "
                                   "(%s)" % (n, T, B, "BASELINE configs[1]; x8 GPUs = configs[2]" if n == 10
                                                  else "BASELINE configs[4] shape; batch chosen here"),
                       "sequences_per_gpu": B, "T": T, "n": n, "global_sequences": B * world,
                       "parallelism": "dp%d" % world, "collective_backend": (dist.get_backend() if world > 1 else None),
                       "step": "estep kernel + batch stat reduce" + (" + %s all-reduce" % ("IPC mailbox" if os.environ.get("SVAE_BENCH_ALLREDUCE", "") == "mailbox" else ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend())) if world > 1 else "")},
            "roofline": roofline(options, T, n, B, kern_ms),
            "csrc_sha16": _lib.source_hash(),
        }
        # parity gate: PARITY_SEQUENCES sequences of the batch just timed (rank 0's shard), as the
        # timed plan left them, against the reference's compiled E-step -- the line carries the distance, and the
        # process exits non-zero beyond PARITY_TOL
        try:
            out["parity"] = parity_gate(snap)
            parity_failed = not out["parity"]["ok"]
        except Exception as e:
            out["parity"] = {"error": repr(e)[:300], "ok": False}
            parity_failed = True
        if world == 1 and not args.no_extra and args.workload == "lds10" and args.seqs_per_gpu is None:
            # the other single-GPU configurations BASELINE.json names, same measurement, fewer steps
            extra = []
            for (eT, en, eB, esteps, what) in ((200, 10, 4096, max(5, args.steps // 2),
                                                "north_star single-GPU target: 4096 sequences x T=200, n=10 on ONE GPU"),
                                               (1000, 64, 512, max(3, args.steps // 10),
                                                "BASELINE configs[4] shape: latent dim 64, T=1000, 512 sequences per GPU")):
                try:
                    # (0.16 s per sequence in the reference at n = 64, T = 1000: four of them)
                    el, km, esnap = measure(dev, 0, 1, None, options, eT, en, eB, esteps, 2,
                                            parity=PARITY_SEQUENCES if en <= 15 else 4)
                    extra.append({"workload": what, "value": eB * esteps / el, "unit": "sequences/s",
                                  "steps": esteps, "warmup": 2, "ms_per_step": 1e3 * el / esteps,
                                  "roofline": roofline(options, eT, en, eB, km)})
                    if en > 15:
                        extra[-1]["model"] = "rotation_lds_natparam (well-conditioned; the timed model)"
                    extra[-1]["parity"] = parity_gate(esnap)
                    parity_failed = parity_failed or not extra[-1]["parity"]["ok"]
                    if en > 15:
                        extra[-1]["parity_rand_lds"] = tile_parity_on_rand_lds(dev, eT, en)
                except Exception as e:  # the headline line must survive
                    extra.append({"workload": what, "error": repr(e)})
                torch.cuda.empty_cache()
            # the other BASELINE configurations and the training path: measured beside `value`, never instead of it
            # (order kept across rounds: [2] training path, [3] tile training, [4] SLDS, [5] GMM, [6] training path at 4096,
            #  [7] tile training at one workgroup per CU, [8] GMM training step, [9] end-to-end make_gradfun step, eager and as one hipGraph,
            #  [10] the GMM fixed point at the shipped script's shape K = 15 / 500 points)
            for fn in (lambda: measure_training_path(dev, T, n, B), lambda: measure_tile_training(dev),
                       lambda: measure_slds(dev), lambda: measure_gmm(dev),
                       lambda: measure_training_path(dev, T, n, 4096), lambda: measure_tile_training(dev, B=256),
                       lambda: measure_gmm_training(dev), lambda: measure_gradfun_step(dev),
                       lambda: measure_gmm(dev, K=15, T=500, what="the reference's shipped script shape "
                                                                   "(experiments/gmm_svae_synth.py:29-33)")):
                try:
                    extra.append(fn())
                except Exception as e:
                    extra.append({"error": repr(e)})
                torch.cuda.empty_cache()
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(B, T, n)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
                if n == 10 and "extra" in out and "value" in out["extra"][0]:
                    # north_star: >= 50x the reference CPU E-step on 4096 sequences at 1 GPU (quoted against
                    # the FASTER CPU figure: the compiled reference on all host cores)
                    out["extra"][0]["speedup_vs_cpu_baseline"] = out["extra"][0]["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # never lose the GPU line to a baseline hiccup
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: parity gate failed (see \"parity\" in the line above)")


if __name__ == "__main__":
    main()
