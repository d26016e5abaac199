"""x = E[x_t] of sequence 0 at a few nodes from the default SLDS mean-field dispatch (K = 3, n = 4, T = 9) under the library
SVAE_AMD_LIB names: with a dump_reg.py-patched library the rows hold the dumped register (lanes 0 .. n-1 of both chains)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/isa_patch/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import slds_rpc_debug as D
K, n, T, B = 3, 4, 9, 5
args = D.setup(K, n, T, B, seed=K + T + n)
got = D.run("default", K, n, T, B, args)
x = got["x"][0].cpu().numpy()
np.set_printoptions(precision=6, suppress=True, linewidth=200)
print(os.path.basename(os.environ["SVAE_AMD_LIB"]), "node1", x[1], "node7", x[7], "node2", x[2], "node6", x[6])
