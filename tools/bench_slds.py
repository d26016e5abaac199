"""Timing of the SLDS-SVAE local mean field (BASELINE configs[3]: K=8 discrete states, latent dim 10,
2048 sequences x T=500).  Usage: python tools/bench_slds.py [B T n K]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.distributions import expfam
from svae_amd.models import slds_svae
from svae_amd.hmm.hmm_inference import hmm_estep
from svae_amd.lds.lds_inference import LDSEStepPlan


def globals_(K, n, rng):
    from svae_amd.lds.synthetic_data import rand_slds_global_natparam
    return rand_slds_global_natparam(K, n, rng)


def main():
    B, T, n, K = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 500, 10, 8)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    glob = globals_(K, n, rng)
    node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
            torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
    eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev)
    modes = [("fused", True)] if slds_svae.SLDSMeanfieldPlan.supported(n, T, K) else []
    if "--fused-only" not in sys.argv:
        modes.append(("materialised", False))
    for name, fused in modes:
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            (hmm_stats, lds_stats), _, (hv, lv), iters = slds_svae.optimize_local_meanfield(
                glob, node, eps, fused=fused, pair_stats=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        it = int(iters.max())
        print("SLDS local mean field [%s] B=%d T=%d n=%d K=%d: %.1f ms for %d sweeps (%.1f ms/sweep, %.0f sequence-sweeps/s); "
              "iterations per sequence min/mean/max %d/%.1f/%d"
              % (name, B, T, n, K, dt * 1e3, it, dt * 1e3 / it, B * it / dt, int(iters.min()),
                 float(iters.double().mean()), it))
        del hmm_stats, lds_stats
    if modes and modes[0][1]:
        # the fused LDS mean-field kernel alone (all sequences active)
        _, _, di, dp = slds_svae.global_to_local_maps(glob, dev)
        w = torch.softmax(2. * torch.randn(B, T, K, dtype=torch.float64, device=dev), -1)
        plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev)
        plan.launch(di, dp, w, node); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): plan.launch(di, dp, w, node)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("  fused LDS mean-field kernel alone: %.3f ms per call (%.2f ns per sequence-step)" % (ms, ms * 1e6 / (B * T)))
    if "--run-inference" in sys.argv:
        prior = globals_(K, n, rng)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = slds_svae.run_inference(prior, glob, node, 1, init_eps=eps)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("  run_inference (ascent + final pass + sampler + global statistics): %.1f ms" % (dt * 1e3))
        del out
    # the two kernels alone, same shapes
    node_hmm = torch.randn(B, T, K, dtype=torch.float64, device=dev)
    init = torch.zeros(K, dtype=torch.float64, device=dev); pair = torch.log_softmax(torch.randn(K, K, dtype=torch.float64, device=dev), -1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    hmm_estep((init, pair, node_hmm)); torch.cuda.synchronize()
    ev[0].record()
    for _ in range(5): hmm_estep((init, pair, node_hmm))
    ev[1].record(); torch.cuda.synchronize()
    print("  HMM E-step kernel alone: %.3f ms per call" % (ev[0].elapsed_time(ev[1]) / 5))


if __name__ == "__main__":
    main()
