"""How far the smoothed moments of the HIP E-step are from the reference's compiled E-step as the model gets ill-conditioned:
the reference's `rand_lds` generator over many seeds (its random SPD blocks have a heavy-tailed condition number), one line
per decade of cond(J22).  Both sides are fp64; the kernels form P^-1 explicitly (Gauss-Jordan) and multiply, the reference
factors and solves -- the smoothed MEANS then differ by ~cond^2 eps against ~cond eps ("Conditioning").
usage: python tools/conditioning_sweep.py [n] [T] [seeds]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref                                        # (checker only)
from svae_amd.lds.lds_inference import natural_lds_estep_general
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
T = int(sys.argv[2]) if len(sys.argv) > 2 else 45
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (np.max(np.abs(np.asarray(b))) + 1e-300))
rows = []
for seed in range(seeds):
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((1, T, n), rng, with_logZ=True)
    cond = float(np.linalg.cond(np.asarray(pair[2])))
    want = ref.estep((init, pair), tuple(x[0] for x in node))
    noise = tuple(x * (1 + 1e-13 * rng.standard_normal(x.shape)) for x in node)
    want2 = ref.estep((init, pair), tuple(x[0] for x in noise))
    with torch.no_grad():
        ln, (Ei, Ep, En) = natural_lds_estep_general((tuple(t(x) for x in init), tuple(t(x) for x in pair)), tuple(t(x) for x in node))
    ex, dxx = En[1][0].cpu().numpy(), En[0][0].cpu().numpy()
    wln, (wEi, wEp, wEn) = want
    rows.append((cond, rel(ln.cpu().numpy()[0], wln), rel(ex, wEn[1]), rel(dxx, wEn[0]), rel(want2[1][2][1], wEn[1])))
rows = np.array(rows)
print("n = %d, T = %d, %d seeds of rand_lds: max over the seeds of a decade, relative to max|reference|" % (n, T, seeds))
print("%-22s %6s %10s %10s %10s %28s" % ("cond(J22)", "seeds", "lognorm", "E[x]", "diag E[xx']", "reference E[x] under 1e-13 input noise"))
for lo in range(0, 10):
    m = (rows[:, 0] >= 10.0 ** lo) & (rows[:, 0] < 10.0 ** (lo + 1))
    if m.any():
        r = rows[m]
        print("1e%d .. 1e%d %14s %6d %10.1e %10.1e %10.1e %28.1e" % (lo, lo + 1, "", m.sum(), r[:, 1].max(), r[:, 2].max(), r[:, 3].max(), r[:, 4].max()))
