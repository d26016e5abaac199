"""Training-path kernels with / without producer wavefronts at several batch sizes (tuning of
the 1024-sequence default behind SVAE_OPT_PRODUCERS_ON / _OFF).  Usage: python tools/prod_threshold.py [B ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import _lib
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


def run(B, T, n, S, prod):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
    eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev)
    g = [torch.randn(B, dtype=torch.float64, device=dev), torch.randn(B, T, n, dtype=torch.float64, device=dev),
         torch.randn(B, T, n, dtype=torch.float64, device=dev), torch.randn(B, T, S, n, dtype=torch.float64, device=dev)]
    plan = LDSEStepPlan(B, T, n, dev, options=_lib.OPT_PRODUCERS_ON if prod else _lib.OPT_PRODUCERS_OFF)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for rep in range(3):
        plan.launch(*args, None, False, True, True)
        ev[0].record(); smp = plan.sample(eps)
        ev[1].record(); plan.vjp(g[0], g[1], g[2], g[3], eps, smp)
        ev[2].record(); plan.vjp(g[0], g[1], g[2])
        ev[3].record(); torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]


if __name__ == "__main__":
    Bs = [int(x) for x in sys.argv[1:]] or [512, 1024, 1536, 2048, 2304, 4096]
    for B in Bs:
        a = run(B, 200, 10, 1, True)
        b = run(B, 200, 10, 1, False)
        print("B=%5d  producers: sampler %.3f VJP %.3f VJP(no samples) %.3f | without: %.3f %.3f %.3f" % (B, *a, *b))
