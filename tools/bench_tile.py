"""Timing of the LDS-tiled MFMA E-step (16 <= n <= 64).  Usage: python tools/bench_tile.py n T B [reps]"""
import sys
import time

import numpy as np
import torch

from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


def main():
    n, T, B = (int(x) for x in sys.argv[1:4])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
    plan = LDSEStepPlan(B, T, n, dev)
    plan.launch(*args)
    torch.cuda.synchronize()
    plan.check_info()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        plan.launch(*args)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    best = min(ms)
    NP = 16 * ((n + 15) // 16)
    nb = NP // 16
    prods = nb * ((2 * (2 * nb)) + (nb - 1) * 2 * nb + 2 * (nb - 1) + 1) + nb * (nb + 1) * nb + 2 * nb * nb * nb
    flops = prods * 2 * 16 ** 3 * T * B
    print("n=%d T=%d B=%d: %.3f ms  (%.1f seq/s, %.2f us/step/seq-slot, %.2f TFLOP/s on %d tile products/step)"
          % (n, T, B, best, B / best * 1e3, best * 1e3 / T, flops / best / 1e9, prods), ms)


if __name__ == "__main__":
    main()
