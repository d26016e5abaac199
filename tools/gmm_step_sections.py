"""Wall-clock sections of the GMM training step (BASELINE configs[0] shape): where the host layer spends a step.
Usage: python tools/gmm_step_sections.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.distributions import expfam
from svae_amd.models import gmm

dev = torch.device("cuda:0")
K, N, T, S = 5, 2, 1000, 1
gen = torch.Generator().manual_seed(K)
prior = tuple(x.to(dev) for x in gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, generator=gen))
glob = tuple(x.to(dev) for x in gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen))
rng = np.random.default_rng(0)
nJ = torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev).requires_grad_(True)
nh = torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev).requires_grad_(True)
init = gmm.initialize_meanfield(T, K, dev, torch.Generator(device=dev).manual_seed(1))
eps = torch.randn(T, S, N, dtype=torch.float64, device=dev)


def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


lg, gg = expfam.dirichlet_expectedstats(glob[0]), expfam.niw_expectedstats(glob[1])
print("dirichlet + niw expectedstats      %8.1f us" % t(lambda: (expfam.dirichlet_expectedstats(glob[0]), expfam.niw_expectedstats(glob[1]))))
print("prior_kl                           %8.1f us" % t(lambda: gmm.prior_kl(glob, prior)))
print("meanfield_from_globals (no check)  %8.1f us" % t(lambda: gmm.meanfield_from_globals(lg, gg, (nJ.detach(), nh.detach()), init, check=False)))
print("meanfield_from_globals (check)     %8.1f us" % t(lambda: gmm.meanfield_from_globals(lg, gg, (nJ.detach(), nh.detach()), init)))
o = gmm.meanfield_from_globals(lg, gg, (nJ.detach(), nh.detach()), init)
print("sampler kernel                     %8.1f us" % t(lambda: gmm.gaussian_sample(o["gaussian_natparam"], eps)))


def step():
    samples, stats, gkl, lkl = gmm.run_inference_differentiable(prior, glob, (nJ, nh), S, label_init=init, eps=eps)
    return torch.autograd.grad(lkl + samples.sum(), [nJ, nh])


print("run_inference_differentiable + bwd %8.1f us" % t(step))
