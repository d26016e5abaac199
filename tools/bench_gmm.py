"""Timing of the GMM mean-field kernel (BASELINE configs[0]: K = 5, 2-D latents, 1000 points; and the
shapes the reference's gmm_svae_synth.py ships: K = 15, 500 points).  One launch runs the whole fixed
point; reported per call with the number of sweeps it took.  Usage: python tools/bench_gmm.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.distributions import expfam
from svae_amd.models import gmm


def main():
    dev = torch.device("cuda:0")
    for K, N, T in [(5, 2, 1000), (15, 2, 500), (5, 2, 16384)]:
        gen = torch.Generator().manual_seed(K)
        d, niws = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen)
        label_global = expfam.dirichlet_expectedstats(d).to(dev)
        gaussian_globals = expfam.niw_expectedstats(niws).to(dev)
        rng = np.random.default_rng(0)
        node = (torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev),
                torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev))
        init = gmm.initialize_meanfield(T, K, dev)
        o = gmm.meanfield_from_globals(label_global, gaussian_globals, node, init)
        torch.cuda.synchronize()
        reps = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            o = gmm.meanfield_from_globals(label_global, gaussian_globals, node, init, check=False)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        print("GMM mean field K=%d N=%d T=%d: %.1f us per call (host wrapper included), %d sweeps, kl %.4f"
              % (K, N, T, 1e3 * ms, int(o["iters"]), float(o["kl"])))


if __name__ == "__main__":
    main()
