"""Section timing of the fused SLDS local mean field (host clock around synchronised sections).
Usage: python tools/slds_sections.py [B T n K]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.models import slds_svae
from svae_amd.hmm.hmm_inference import hmm_estep
from bench_slds import globals_


def timed(name, fn, reps=3):
    out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print("  %-58s %8.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))
    return out


def main():
    B, T, n, K = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 500, 10, 8)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    glob = globals_(K, n, rng)
    node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
            torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
    eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev)
    d = lambda x: slds_svae._dev64(x, dev)
    hmm_init, hmm_pair, di, dp = timed("global_to_local_maps (host)", lambda: slds_svae.global_to_local_maps(glob, dev))
    x = timed("_initial_sample_path", lambda: slds_svae._initial_sample_path(node, eps))
    node_hmm = timed("_arhmm_nodeparams_from_path", lambda: slds_svae._arhmm_nodeparams_from_path(di, dp, x))
    hv, (Ei, Et, Es) = timed("hmm_estep", lambda: hmm_estep((hmm_init, hmm_pair, node_hmm)))
    plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev)
    active = torch.ones(B, dtype=torch.bool, device=dev)
    timed("fused LDS mean-field launch (all sequences)", lambda: plan.launch(di, dp, Es, node))
    half = torch.arange(0, B, 2, dtype=torch.int32, device=dev)
    timed("fused LDS mean-field launch (every other sequence)", lambda: plan.launch(di, dp, Es, node, half))
    timed("lds_vlb", lambda: plan.lds_vlb(di, dp, Es))
    timed("hmm_nodeparams", lambda: plan.hmm_nodeparams(di, dp))
    sel = lambda new, old: torch.where(active.reshape((B,) + (1,) * (new.dim() - 1)), new, old)
    timed("5 x torch.where commit", lambda: [sel(a, a) for a in (Ei, Et, Es, hv, node_hmm)])
    timed("convergence test (host sync)", lambda: bool(((hv - 1.).abs() < 1e-2).any()))
    timed("whole ascent (fused)", lambda: slds_svae.optimize_local_meanfield(glob, node, eps, fused=True, pair_stats=False), reps=2)
    prior = globals_(K, n, rng)
    timed("run_inference", lambda: slds_svae.run_inference(prior, glob, node, 1, init_eps=eps), reps=2)
    st = slds_svae.optimize_local_meanfield(glob, node, eps, fused=True, pair_stats=False)
    (hmm_stats, _), (hmm_nat, (lds_init, lds_pair)), _, _ = st
    from svae_amd.lds.lds_inference import LDSEStepPlan
    mplan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
    lognorm, (Ei_, Ep, En) = timed("final pass: materialised E-step keeping the factor",
                                   lambda: slds_svae._lds_estep_batched_init(mplan, lds_init, lds_pair, node, keep_factor=True))
    timed("final pass: sampler", lambda: mplan.sample(eps))
    timed("final pass: get_arhmm_local_nodeparams", lambda: slds_svae.get_arhmm_local_nodeparams(di, dp, (Ei_[0], Ei_[1]), (Ep[0], Ep[1], Ep[2])))
    timed("final pass: get_global_stats", lambda: slds_svae.get_global_stats(hmm_stats, (Ei_[0], Ei_[1]), (Ep[0], Ep[1], Ep[2])))
    timed("final pass: get_var_lds_local_natparam", lambda: slds_svae.get_var_lds_local_natparam(di, dp, hmm_stats[2]))
    timed("slds_prior_vlb", lambda: slds_svae.slds_prior_vlb(glob, prior, dev))


if __name__ == "__main__":
    main()
