"""Timing of the training-step path at latent dimension 16..64 (tile-kernel E-step + sampler + VJP through
svae_amd/lds/lds_large.py).  Usage: python tools/bench_tile_train.py [B T n S]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.lds.lds_inference import LDSEStepPlan, lds_inference_differentiable
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


def main():
    B, T, n, S = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 1000, 64, 1)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    init, pair = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    nJ, nh = t(nJ).requires_grad_(True), t(nh).requires_grad_(True)
    eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)
    plan = LDSEStepPlan(B, T, n, dev)
    gs = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)

    def sync_time(fn, reps=2):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    def fwd():
        return lds_inference_differentiable(natparam, (nJ, nh), eps=eps, plan=plan)
    ms_f, out = sync_time(fwd)
    lognorm, (dxx, ex), samples, _ = out

    def fwd_bwd():
        lognorm, (dxx, ex), samples, _ = fwd()
        loss = lognorm.sum() + (dxx * 0.3).sum() + ex.sum() + (samples * gs).sum()
        return torch.autograd.grad(loss, [nJ, nh])
    ms_fb, _ = sync_time(fwd_bwd)
    print("B=%d T=%d n=%d S=%d: E-step + sampler %.1f ms | + VJP (forward + backward) %.1f ms  => VJP alone %.1f ms"
          % (B, T, n, S, ms_f, ms_fb, ms_fb - ms_f))


if __name__ == "__main__":
    main()
