"""run_inference of the SLDS at configs[3], wall clock per repetition, with host-resident and device-resident global
parameters (what bench.py's extra[4] times: device-resident, best of 3 after an empty_cache)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.models import slds_svae as S
from svae_amd.lds.synthetic_data import rand_slds_global_natparam

B, T, n, K = 2048, 500, 10, 8
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
glob_h = rand_slds_global_natparam(K, n, rng)
prior_h = rand_slds_global_natparam(K, n, rng)
_d = lambda x: tuple(_d(y) for y in x) if isinstance(x, (tuple, list)) else torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev), torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
g = torch.Generator(device=dev).manual_seed(1)
init_eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=g)
eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=g)
for name, glob, prior in (("host params", glob_h, prior_h), ("device params", _d(glob_h), _d(prior_h))):
    torch.cuda.empty_cache()
    ts = []
    for rep in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = S.run_inference(prior, glob, node, 1, init_eps=init_eps, eps=eps)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(name, " ".join("%.2f" % t for t in ts))
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S.optimize_local_meanfield(glob, node, init_eps, pair_stats=False)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("   ascent alone", " ".join("%.2f" % t for t in ts))
