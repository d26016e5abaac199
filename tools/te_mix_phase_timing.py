"""Per-phase cycle counts of the fused SLDS LDS mean-field kernel (two-ended kernel, MIX mode) from a variant
library built with -DSVAE_PHASE_TIMING:
  tools/build_variant.sh variants/te_timing.so 10 -DSVAE_PHASE_TIMING
  SVAE_AMD_LIB=variants/te_timing.so python tools/te_mix_phase_timing.py [B T K]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.models import slds_svae
from bench_slds import globals_
B, T, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 500, 8)
n = 10
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
glob = globals_(K, n, rng)
node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
        torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
_, _, di, dp = slds_svae.global_to_local_maps(glob, dev)
w = torch.softmax(2. * torch.randn(B, T, K, dtype=torch.float64, device=dev), -1)
plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev)
for _ in range(3):
    plan.launch(di, dp, w, node)
torch.cuda.synchronize()
tm = plan.E_init[:, :6].cpu().numpy()
e = T // 2
m = tm.mean(0)
print("B=%d T=%d K=%d cycles per iteration (mean over waves): condition+mix %.0f  gauss_jordan %.0f  mix+schur %.0f  "
      "scale+store+gather %.0f | elimination step %.0f   meeting+lognorm %.0f (once)   smoother step (with contraction) %.0f "
      "| total %.0f cycles" % (B, T, K, m[0] / e, m[1] / e, m[2] / e, m[3] / e, m[:4].sum() / e, m[4], m[5] / (e + 1), m.sum()))
