"""Randomised cross-check of the small-batch kernels (second wavefront per sequence in the two-ended E-step; producer /
helper / one-sequence-per-wavefront sampler and VJP kernels) against the packed schedules of the same arithmetic
(SVAE_OPT_TWOEND_OFF | SVAE_OPT_PRODUCERS_OFF).  Usage: python tools/stress_small_batch.py [cases]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import _lib
from svae_amd.lds.lds_inference import lds_inference_differentiable, natural_lds_estep_general, set_default_options
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


def rel(x, y):
    return float((x - y).abs().max()) / (float(y.abs().max()) + 1e-300)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2026)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    worst = 0.0
    for case in range(cases):
        n = int(rng.integers(1, 16)); T = int(rng.integers(1, 45)); B = int(rng.integers(1, 11)); S = int(rng.integers(1, 7))
        init, pair = rand_lds_natparam(n, rng)
        node = rand_node_potentials((B, T, n), rng)
        natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
        g = [t(rng.standard_normal(B)), t(rng.standard_normal((B, T, n))), t(rng.standard_normal((B, T, n))),
             t(rng.standard_normal((B, T, S, n)))]
        eps = t(rng.standard_normal((B, T, S, n)))

        def estep():
            ln, (Ei, Ep, En) = natural_lds_estep_general(natparam, (t(node[0]), t(node[1])))
            return [ln, Ei[0], Ei[1], Ep[0], Ep[1], Ep[2], En[0], En[1]]

        def train(with_samples):
            nJ, nh = t(node[0]).requires_grad_(True), t(node[1]).requires_grad_(True)
            ln, (dxx, ex), smp, _ = lds_inference_differentiable(natparam, (nJ, nh), eps=eps if with_samples else None)
            loss = (g[0] * ln).sum() + (g[1] * dxx).sum() + (g[2] * ex).sum()
            if with_samples:
                loss = loss + (g[3] * smp).sum()
            loss.backward()
            return [nJ.grad, nh.grad] + ([smp.detach()] if with_samples else [])

        new = estep() + train(False) + train(True)
        o1 = set_default_options(_lib.OPT_TWOEND_OFF | _lib.OPT_PRODUCERS_OFF)
        try:
            old = estep() + train(False) + train(True)
        finally:
            set_default_options(o1)
        errs = [rel(x, y) for x, y in zip(new, old)]
        err = max(errs)
        if err >= 1e-9:      # outputs: 8 of the E-step, 2 gradients without samples, 2 gradients + samples with
            print("   per output:", " ".join("%.1e" % e for e in errs))
        worst = max(worst, err)
        flag = "" if err < 1e-9 else "   <-- conditioning? (two-ended vs one-directional elimination on an ill-conditioned model)"
        print("case %3d n=%2d T=%2d B=%2d S=%d: max rel diff %.2e%s" % (case, n, T, B, S, err, flag))
    # (different elimination orders agree to ~cond * eps: 1e-6 on the worst random models here; the second-wavefront
    #  smoother itself is BITWISE equal to the two-row one -- checked with a TE_S4_MAX_B = 0 build)
    print("worst", worst)
    assert worst < 1e-5


if __name__ == "__main__":
    main()
