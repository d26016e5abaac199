"""Per-phase cycle counts of the row-per-chain two-ended kernel (a variant library built with -DSVAE_PHASE_TIMING:
tools/build_variant.sh variants/rpc_timing.so 10 -DSVAE_PHASE_TIMING):
SVAE_AMD_LIB=variants/rpc_timing.so python tools/rpc_phase_timing.py [B ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import _lib
from svae_amd.lds.lds_inference import LDSEStepPlan
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
T, n = 200, 10
dev = torch.device("cuda:0")
for B in [int(x) for x in sys.argv[1:]] or (512, 2048, 4096):
    init, pair = rand_lds_natparam(n, np.random.default_rng(0))
    node = rand_node_potentials((B, T, n), np.random.default_rng(1))
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    plan = LDSEStepPlan(B, T, n, dev, options=_lib.OPT_LAYOUT_PACKED)
    args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1), t(node[0]), t(node[1]), None]
    for _ in range(3):
        plan.launch(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.launch(*args); e1.record(); torch.cuda.synchronize()
    tm = plan.E_init[::2, :10].cpu().numpy()
    wall = plan.E_init[::2, 10:12].cpu().numpy() / 100.0          # us (s_memrealtime: 100 MHz)
    w0 = wall[:, 0].min()
    st, en = wall[:, 0] - w0, wall[:, 1] - w0
    e = T // 2
    tot = tm.sum(1)
    order = np.argsort(en - st)
    pick = order[[0, len(order) // 4, len(order) // 2, 3 * len(order) // 4, len(order) - 1]]
    print("   busy us / cycles / GHz at quantiles 0,25,50,75,100 %: " + "  ".join(
        "%.0f/%.0fk/%.2f" % (en[i] - st[i], tot[i] / 1e3, tot[i] / (en[i] - st[i]) / 1e3) for i in pick))
    elim = tm[:, :4].sum(1) / e
    smo = tm[:, 5:].sum(1) / (e + 1)
    print("   elimination cycles/step at those wavefronts: " + " ".join("%.0f" % elim[i] for i in pick) +
          " | smoother: " + " ".join("%.0f" % smo[i] for i in pick) + " | workgroup ids: " + " ".join(str(int(i)) for i in pick))
    print("   wavefront start (us after the first): median %.1f  p90 %.1f  max %.1f | end: min %.1f median %.1f max %.1f | "
          "busy per wavefront: median %.1f us" % (np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en),
                                                en.max(), np.median(en - st)))
    e = T // 2
    m = tm.mean(0)
    print("B=%d (%.3f ms) cycles per step, mean over wavefronts: ELIM cond %.0f  gauss_jordan %.0f  schur %.0f  hand-off %.0f = %.0f | "
          "meeting+lognorm %.0f (once) | SMOOTH loads+G %.0f  transpose %.0f  W %.0f  sums(W) + S %.0f  diag+sums+stores %.0f = %.0f | total %.0f cycles"
          % (B, e0.elapsed_time(e1), m[0] / e, m[1] / e, m[2] / e, m[3] / e, m[:4].sum() / e, m[4], m[5] / (e + 1), m[6] / (e + 1), m[7] / (e + 1),
             m[8] / (e + 1), m[9] / (e + 1), m[5:].sum() / (e + 1), m.sum()))
