"""Per-phase cycle counts of the tiled E-step's halves at two workgroups per CU (timing build of tools/build_tile_variant.sh
<so> -DSVAE_TILE_TIMING, selected with SVAE_AMD_LIB).  Usage: n T B"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
n, T, B = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda:0")
init, pair = rand_lds_natparam(n, np.random.default_rng(0))
node = rand_node_potentials((B, T, n), np.random.default_rng(1))
t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
plan = LDSEStepPlan(B, T, n, dev)
args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1), t(node[0]), t(node[1]), None]
names = ["step start", "first factor (w0)", "barrier before P1", "P1 pivot row + barrier", "P2 (w0: look-ahead+factor; w1: row)",
         "barrier after GJ", "hand-off + barrier", "schur + barrier", "reload + barrier",
         "B2 sigma/stats + barrier", "B0 stage + emit", "B1 W, matvec + barrier"]
for half in (1, 2):
    for _ in range(2):
        plan.launch(*args, half=half)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); plan.launch(*args, half=half); ev1.record(); torch.cuda.synchronize()
    tm = plan.E_init[:, :24].cpu().numpy().mean(0).reshape(2, 12) / T
    print("half %d  n=%d T=%d B=%d  %.2f ms; cycles/step (mean over sequences); totals: wave0 %.0f  wave1 %.0f"
          % (half, n, T, B, ev0.elapsed_time(ev1), tm[0].sum(), tm[1].sum()))
    for i, nm in enumerate(names):
        if tm[0, i] or tm[1, i]:
            print("  %-40s w0 %8.0f   w1 %8.0f" % (nm, tm[0, i], tm[1, i]))
