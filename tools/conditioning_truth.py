"""The smoothed means of the HIP E-step, of its sampler at eps = 0 (the same posterior mean through the one-directional
records) and of the reference's compiled E-step against a 60-digit block-tridiagonal solve (mpmath), on the draws of the
reference's `rand_lds` generator with the worst-conditioned pair blocks.  The arbiter behind DESIGN section 2, "Conditioning":
the smoother kernels keep P^-1 in their per-step records and rebuild P^-1 J12 from it -- cond^2 eps; the sampler's records
carry P^-1 J12 from the elimination itself -- cond eps, like the reference's factor-and-solve.
usage: python tools/conditioning_truth.py [n] [T] [seeds]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref
from svae_amd.lds.lds_inference import natural_lds_estep_general
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
T = int(sys.argv[2]) if len(sys.argv) > 2 else 45
NSEEDS = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
from oracle.lds_mp import smoothed_means_mp                   # (checker only)
truth = lambda init, pair, node: smoothed_means_mp(init, pair, node[0][0], node[1][0])
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
found = 0
bad_seeds = []
for seed in range(NSEEDS):
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((1, T, n), rng, with_logZ=True)
    cond = float(np.linalg.cond(np.asarray(pair[2])))
    if cond < 3e6 and seed not in (0, 1):
        continue
    bad_seeds.append(seed)
    want = ref.estep((init, pair), tuple(x[0] for x in node))
    with torch.no_grad():
        ln, (Ei, Ep, En) = natural_lds_estep_general((tuple(t(x) for x in init), tuple(t(x) for x in pair)), tuple(t(x) for x in node))
    ex = En[1][0].cpu().numpy(); rx = np.asarray(want[1][2][1])
    tx = truth(init, pair, node)
    print("seed %3d cond(J22) %.1e: HIP vs 60-digit truth %.1e | reference vs truth %.1e | HIP vs reference %.1e" % (seed, cond, rel(ex, tx), rel(rx, tx), rel(ex, rx)), flush=True)

print("--- sampler at eps = 0 (one-directional filter + backward recursion) against the two-ended smoother's E[x]")
from svae_amd.lds.lds_inference import lds_inference_differentiable
for seed in bad_seeds:
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((1, T, n), rng, with_logZ=True)
    tx = truth(init, pair, node)
    with torch.no_grad():
        lognorm, (dxx, ex), samples, _ = lds_inference_differentiable((tuple(t(x) for x in init), tuple(t(x) for x in pair)), tuple(t(x) for x in node), eps=torch.zeros((1, T, 1, n), dtype=torch.float64, device=dev))
    sm = samples[0, :, 0].cpu().numpy(); e2 = ex[0].cpu().numpy()
    err = np.abs(e2 - tx).max(axis=1) / np.abs(tx).max()
    print("seed %3d: sampler(eps=0) vs truth %.1e | E[x] of the same call vs truth %.1e | worst nodes of E[x]: %s" % (seed, rel(sm, tx), rel(e2, tx), np.argsort(-err)[:6].tolist()), flush=True)
