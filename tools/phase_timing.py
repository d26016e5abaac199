"""Per-phase cycle counts of the forward half (variants/timing.so built with -DSVAE_PHASE_TIMING)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
B, T, n = int(os.environ.get("B", 512)), 200, 10
dev = torch.device("cuda:0")
init, pair = rand_lds_natparam(n, np.random.default_rng(0))
node = rand_node_potentials((B, T, n), np.random.default_rng(1))
t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
plan = LDSEStepPlan(B, T, n, dev)
args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1), t(node[0]), t(node[1]), None]
for _ in range(3):
    plan.launch(*args)
torch.cuda.synchronize()
tm = plan.E_init[:, :4].cpu().numpy() / T
print("B=%d cycles/step (mean over waves): setup %.0f  gauss_jordan %.0f  stores %.0f  schur %.0f  | total %.0f" % (B, *tm.mean(0), tm.mean(0).sum()))
