"""Per-section cycle counters of tile VJP phase 2 (a timing build: tools/build_vjp_tile_variant.sh <out.so> -DSVAE_TV_TIMING,
run with SVAE_AMD_LIB=<out.so>).  Usage: python tools/tile_vjp_timing.py [n T B]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds import lds_large
from svae_amd.lds.lds_large import vjp_from_handoff_hip
lds_large._keep_ws = True
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials

NAMES = ["stage Pinv, J12", "matvec J12 h", "J12 J_bar -> X_bar", "stage G, -Pinv_bar, requests", "X_bar G'",
         "- Pinv_bar Pinv", "store Y", "Pinv Y", "(unused)", "matvec Pinv c", "epilogue", "store + symmetrise",
         "outputs"]
NAMES1 = ["direct cotangents, x_bar", "records (stores)", "stage G, Sigma, requests", "Sigma_bar G", "SG Sigma",
          "G_bar epilogue (stores)", "matvecs", "G' SG", "store + symmetrise"]


def main():
    n, T, B = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1000, 64)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    (J0, h0, z0), (J11, J12, J22, zp) = rand_lds_natparam(n, rng)
    nJ, nh = rand_node_potentials((B, T, n), rng)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(x) for x in (J0, h0, z0, J11, J12, J22, zp, nJ, nh)]
    plan = LDSEStepPlan(B, T, n, dev)
    plan.launch(*args)
    g = [torch.randn(B, dtype=torch.float64, device=dev), torch.randn(B, T, n, dtype=torch.float64, device=dev),
         torch.randn(B, T, n, dtype=torch.float64, device=dev)]
    gJ, gh = vjp_from_handoff_hip(plan, args[4], False, plan.E_node_x.clone(), g[0], g[1], g[2])
    torch.cuda.synchronize()
    tm = gh[:, 0, :16].cpu().numpy().mean(0)
    tot = tm[:16].sum()
    print("phase 2, n=%d T=%d B=%d: %.0f cycles per step (counter ticks)" % (n, T, B, tot / T))
    for k, name in enumerate(NAMES):
        print("  %-28s %8.0f  %5.1f %%" % (name, tm[k] / T, 100 * tm[k] / tot))
    print("  (of 'stage G, -Pinv_bar, requests': stage G %.0f, stage -Pinv_bar %.0f, requests %.0f, barrier %.0f)"
          % (tm[13] / T, tm[14] / T, tm[15] / T, tm[3] / T))
    # phase 1 leaves its counters in c_bar[b, 0, :16] of the VJP workspace (kept by the timing build of lds_large)
    ws = getattr(lds_large, "_last_ws", None)
    if ws is not None:
        off = 2 * B * T * n * n + B * (T - 1) * n * n
        tm1 = ws[off:off + B * T * n].view(B, T * n)[:, :16].cpu().numpy().mean(0)
        tot1 = tm1[:9].sum()
        print("phase 1: %.0f cycles per step" % (tot1 / T))
        for k, name in enumerate(NAMES1):
            print("  %-28s %8.0f  %5.1f %%" % (name, tm1[k] / T, 100 * tm1[k] / tot1))


if __name__ == "__main__":
    main()
