"""Per-phase cycle counts of the persistent GMM kernel (variant library built with -DSVAE_GMM_TIMING):
tools/build_unit_variant.sh tests/_variants/gmm_timing.so gmm_meanfield -DSVAE_GMM_TIMING
SVAE_AMD_LIB=tests/_variants/gmm_timing.so python tools/gmm_phase_timing.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.distributions import expfam
from svae_amd.models import gmm

dev = torch.device("cuda:0")
for K, N, T in [(5, 2, 1000), (15, 2, 500), (5, 2, 8192)]:
    gen = torch.Generator().manual_seed(K)
    d, niws = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen)
    lg, gg = expfam.dirichlet_expectedstats(d).to(dev), expfam.niw_expectedstats(niws).to(dev)
    rng = np.random.default_rng(0)
    node = (torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev),
            torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev))
    init = gmm.initialize_meanfield(T, K, dev, torch.Generator(device=dev).manual_seed(1))
    print("K=%d N=%d T=%d" % (K, N, T), flush=True)
    for _ in range(3):
        o = gmm.meanfield_from_globals(lg, gg, node, init)
        torch.cuda.synchronize()
