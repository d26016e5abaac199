"""which nodes of E[x_t] the default SLDS mean-field dispatch gets wrong against the table kernel, for a few (n, T), under the
library SVAE_AMD_LIB names (one line: the round-6 bisection of the n = 4 ring-consumer failure ran this per variant)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/isa_patch/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import slds_rpc_debug as D
out = []
for (K, n, T) in ((3, 4, 9), (3, 4, 12), (3, 4, 13), (8, 4, 20), (3, 5, 12), (3, 3, 12)):
    B = 5
    args = D.setup(K, n, T, B, seed=K + T + n)
    base = D.run("tables", K, n, T, B, args)
    got = D.run("default", K, n, T, B, args)
    d = (base["x"] - got["x"]).abs().amax(dim=(0, 2))
    sc = float(base["x"].abs().max())
    out.append("n=%d T=%d: x nodes off %s" % (n, T, (d > 1e-9 * sc).nonzero().flatten().tolist()))
print(os.environ.get("SVAE_AMD_LIB", "default"), " | ".join(out))
