import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
B, T, n = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 200, 10)))
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
(J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
nJ, nh = rand_node_potentials((B, T, n), rng)
t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
plan = LDSEStepPlan(B, T, n, dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for rep in range(3):
    ev[0].record(); plan.filter(*args)
    ev[1].record(); plan.launch(*args)
    ev[2].record(); plan.launch(*args, None, False, True, True)
    ev[3].record(); torch.cuda.synchronize()
print("filter-only %.3f ms | two-ended E-step %.3f ms | one-directional E-step keeping both hand-offs %.3f ms"
      % tuple(ev[i].elapsed_time(ev[i + 1]) for i in range(3)))
