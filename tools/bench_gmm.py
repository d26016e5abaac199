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
    for K, N, T, mw in [(5, 2, 1000, False), (15, 2, 500, False), (15, 2, 500, True), (5, 2, 8192, True), (5, 2, 16384, False),
                        (5, 2, 16384, True), (5, 2, 1000, True), (5, 2, 1000, "sweeps"), (15, 2, 262144, True),
                        (5, 2, 1048576, True)]:
        gen = torch.Generator().manual_seed(K)
        d, niws = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen)
        label_global = expfam.dirichlet_expectedstats(d).to(dev)
        gaussian_globals = expfam.niw_expectedstats(niws).to(dev)
        rng = np.random.default_rng(0)
        node = (torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev),
                torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev))
        init = gmm.initialize_meanfield(T, K, dev)
        kw = dict(multi_wg=bool(mw), persistent=(mw is True))
        o = gmm.meanfield_from_globals(label_global, gaussian_globals, node, init, **kw)
        torch.cuda.synchronize()
        reps = 20 if T < 100000 else 5
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            o = gmm.meanfield_from_globals(label_global, gaussian_globals, node, init, check=False, **kw)
        ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        it = int(o["iters"])
        print("GMM mean field [%s] K=%d N=%d T=%d: %.1f us per call (host wrapper included), %d sweeps, %.1f us per sweep, "
              "%.1f M point-sweeps/s, kl %.4f" % (o["path"], K, N, T, 1e3 * ms, it, 1e3 * ms / max(it, 1), T * it / ms / 1e3, float(o["kl"])))


if __name__ == "__main__":
    main()
