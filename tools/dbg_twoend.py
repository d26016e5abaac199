"""Development check: the two-ended E-step kernel against the packed one-directional kernel (same
library), output by output, on a grid of (T, n) incl. the edge cases T = 4, 5 (even / odd)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import _lib
from svae_amd.lds.lds_inference import natural_lds_estep_general
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials

lib = _lib.load()
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)


def run(init, pair, node, twoend):
    lib.svae_lds_set_twoend(int(twoend)); lib.svae_lds_set_split_max_b(0)
    ln, (Ei, Ep, En) = natural_lds_estep_general((tuple(t(x) for x in init), tuple(t(x) for x in pair)),
                                                 tuple(t(x) for x in node), check=False)
    torch.cuda.synchronize()
    return [x.clone() for x in (ln, Ei[0], Ei[1], Ep[0], Ep[1], Ep[2], En[0], En[1])]


names = ["lognorm", "ExxT0", "Ex0", "Ep0", "Ep1", "Ep2", "En_dxx", "En_x"]
worst = 0.0
MODE = int(os.environ.get("TE_MODE", "1"))
for inhomog in (False, True):
    for n in (1, 2, 3, 5, 8, 9, 10):
        for T in (4, 5, 6, 7, 12, 33, 200):
            rng = np.random.default_rng(17 * n + T)
            B = 3
            init, pair = rand_lds_natparam(n, rng)
            if inhomog:
                ps = [rand_lds_natparam(n, rng)[1] for _ in range(T - 1)]
                pair = tuple(np.stack([p[i] for p in ps]) for i in range(4))
            node = rand_node_potentials((B, T, n), rng, with_logZ=True)
            a = run(init, pair, node, MODE)
            r = run(init, pair, node, False)
            errs = []
            for nm, x, y in zip(names, a, r):
                sc = float(y.abs().max()) + 1e-300
                d = (x - y).abs()
                errs.append(float(d.max()) / sc)
            w = max(errs)
            worst = max(worst, w)
            flag = "" if w < 1e-9 else "   <-- " + ", ".join("%s %.1e" % (nm, e) for nm, e in zip(names, errs) if e >= 1e-9)
            if flag and T <= 12:
                d = (a[7] - r[7]).abs().amax(dim=(0, 2))
                flag += "\n      En_x err by t: " + " ".join("%.0e" % v for v in d.tolist())
            print("inhomog=%d n=%2d T=%3d  max rel diff %.2e%s" % (inhomog, n, T, w, flag))
print("WORST", worst)
