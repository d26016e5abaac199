"""Timing of the training-step path of the LDS model at the headline shape: E-step (keeping the sampler /
VJP hand-off), sampler, VJP as separate calls, then as the one-call inference + VJP the model layer uses.  Usage: python tools/bench_train_path.py [B T n S] [--options NAME]   (NAME: a key of
svae_amd._lib.KERNEL_OPTIONS, e.g. twoend_seq)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


def main():
    B, T, n, S = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (512, 200, 10, 1)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
    eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)
    g = [torch.randn(B, dtype=torch.float64, device=dev), torch.randn(B, T, n, dtype=torch.float64, device=dev),
         torch.randn(B, T, n, dtype=torch.float64, device=dev), torch.randn(B, T, S, n, dtype=torch.float64, device=dev)]
    from svae_amd import _lib
    options = _lib.KERNEL_OPTIONS[sys.argv[sys.argv.index("--options") + 1]] if "--options" in sys.argv else None
    plan = LDSEStepPlan(B, T, n, dev, options=options)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for rep in range(3):
        ev[0].record(); plan.launch(*args)
        ev[1].record(); plan.launch(*args, None, False, True, True)
        ev[2].record(); smp = plan.sample(eps)
        ev[3].record(); plan.vjp(g[0], g[1], g[2], g[3], eps, smp)
        ev[4].record(); plan.vjp(g[0], g[1], g[2])
        ev[5].record(); torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    print("B=%d T=%d n=%d S=%d: E-step %.3f ms | E-step keeping factor+cross %.3f | sampler %.3f | VJP %.3f  "
          "=> training path %.3f ms (%.0f seq/s)   [VJP without sample cotangents: %.3f ms]"
          % (B, T, n, S, ms[0], ms[1], ms[2], ms[3], sum(ms[1:4]), B / sum(ms[1:4]) * 1e3, ms[4]))
    # the training path as the model layer runs it since round 6: E-step + sampler in ONE call (svae_lds_inference_f64;
    # lean per-step records above 1024 sequences), then the VJP
    smp2 = torch.empty_like(eps)
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for rep in range(4):
        ev2[0].record(); plan.infer(*args, None, False, eps, smp2)
        ev2[1].record(); plan.vjp(g[0], g[1], g[2], g[3], eps, smp2)
        ev2[2].record(); torch.cuda.synchronize()
    m2 = [ev2[i].elapsed_time(ev2[i + 1]) for i in range(2)]
    print("  one-call inference (%s records) %.3f ms | VJP %.3f  => training path %.3f ms (%.0f seq/s)"
          % ("lean" if plan.lean else "full", m2[0], m2[1], sum(m2), B / sum(m2) * 1e3))
    if options is not None:
        return
    # the same path through the model layer (models.lds.run_inference_differentiable + backward): host wall
    # clock per iteration vs the kernels' sum above = launch / allocation / glue overhead
    import time
    from svae_amd.models import lds as lds_model
    prior = lds_model.make_prior_natparam(n, device=dev)
    glob = lds_model.make_prior_natparam(n, device=dev)
    nodeJ, nodeh = args[7].clone().requires_grad_(True), args[8].clone().requires_grad_(True)
    mplan = LDSEStepPlan(B, T, n, dev)

    def it():
        samples, stats, gkl, lkl = lds_model.run_inference_differentiable(prior, glob, (nodeJ, nodeh), S, eps=eps, plan=mplan)
        loss = (samples * g[3]).sum() + lkl
        gJ, gh = torch.autograd.grad(loss, [nodeJ, nodeh])
        return gJ
    for _ in range(3):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        it()
    torch.cuda.synchronize()
    print("  model layer (run_inference_differentiable + autograd backward): %.3f ms per iteration (host wall clock)"
          % ((time.perf_counter() - t0) / reps * 1e3))


if __name__ == "__main__":
    main()
