"""E-step kernel variants (svae_amd._lib.KERNEL_OPTIONS: per-call selection words) at several batch sizes:
ms per launch from events on the launch stream.  Usage: python tools/bench_estep_variants.py [B ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import _lib
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


def run(B, T, n, name, reps=20):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
    plan = LDSEStepPlan(B, T, n, dev, options=_lib.KERNEL_OPTIONS[name])
    for _ in range(3):
        plan.launch(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.launch(*args)
    e1.record(); torch.cuda.synchronize()
    plan.check_info()
    return e0.elapsed_time(e1) / reps, plan.lognorm.clone(), plan.E_pair.clone()


if __name__ == "__main__":
    Bs = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192]
    T, n = 200, 10
    for B in Bs:
        res = {k: run(B, T, n, k) for k in ("twoend_seq", "twoend_rpc", "packed")}
        ref = res["twoend_seq"]
        d = max(float((res["twoend_rpc"][1] - ref[1]).abs().max() / ref[1].abs().max()),
                float((res["twoend_rpc"][2] - ref[2]).abs().max() / ref[2].abs().max()))
        print("B=%5d  one seq/wavefront %.3f ms (%.2f M seq/s) | row-per-chain %.3f ms (%.2f M seq/s) | packed one-directional "
              "%.3f ms   [rpc vs seq max rel diff %.1e]" % (B, ref[0], B / ref[0] / 1e3, res["twoend_rpc"][0],
                                                         B / res["twoend_rpc"][0] / 1e3, res["packed"][0], d))
