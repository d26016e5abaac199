import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svae_amd.models import gmm
dev = torch.device("cuda:0")
K, N, T, S = 5, 2, 1000, 1
gen = torch.Generator().manual_seed(K)
prior = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, generator=gen)
glob = tuple(x.to(dev) for x in gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, random_scale=3., generator=gen))
prior = tuple(x.to(dev) for x in prior)
rng = np.random.default_rng(0)
nJ = torch.as_tensor(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N)))), device=dev).requires_grad_(True)
nh = torch.as_tensor(3. * rng.standard_normal((T, N)), device=dev).requires_grad_(True)
init = gmm.initialize_meanfield(T, K, dev, torch.Generator(device=dev).manual_seed(1))
eps = torch.randn(T, S, N, dtype=torch.float64, device=dev)
gs = torch.randn(T, S, N, dtype=torch.float64, device=dev)
CHECK = True
def it():
    global CHECK
    samples, stats, gkl, lkl = gmm.run_inference_differentiable(prior, glob, (nJ, nh), S, label_init=init, eps=eps, check=CHECK)
    return torch.autograd.grad(lkl + (samples * gs).sum(), [nJ, nh])
it(); it(); torch.cuda.synchronize()
ts = []
for _ in range(9):
    t0 = time.perf_counter(); it(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print("eager us", sorted(ts)[4])
ref = [x.clone() for x in it()]
CHECK = False
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        it(); it()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = it()
    g.replay(); torch.cuda.synchronize()
    print("graph matches eager:", all(torch.equal(a, b) for a, b in zip(out, ref)))
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    print("graph us", sorted(ts)[4])
except Exception as e:
    print("capture failed:", repr(e)[:300])
