"""Per-phase cycle counts of the two-ended kernel (variants/te_timing.so, built with
-DSVAE_PHASE_TIMING): SVAE_AMD_LIB=variants/te_timing.so python tools/te_phase_timing.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
T, n = 200, 10
dev = torch.device("cuda:0")
for B in (512, 1024, 4096):
    init, pair = rand_lds_natparam(n, np.random.default_rng(0))
    node = rand_node_potentials((B, T, n), np.random.default_rng(1))
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    plan = LDSEStepPlan(B, T, n, dev)
    args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1), t(node[0]), t(node[1]), None]
    for _ in range(3):
        plan.launch(*args)
    torch.cuda.synchronize()
    tm = plan.E_init[:, :6].cpu().numpy()
    e = T // 2
    m = tm.mean(0)
    print("B=%d cycles per iteration (mean over waves): setup %.0f  gauss_jordan %.0f  schur %.0f  scale+store+gather %.0f | fwd %.0f   meeting+lognorm %.0f (once)   smoother %.0f per step | total %.0f cycles"
          % (B, m[0] / e, m[1] / e, m[2] / e, m[3] / e, m[:4].sum() / e, m[4], m[5] / (e + 1), m.sum()))
