"""Randomised cross-check of the HMM E-step kernels against the log-space NumPy restatement (oracle/hmm_numpy.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svae_amd.hmm.hmm_inference import hmm_estep
from oracle import hmm_numpy
rng = np.random.default_rng(7)
worst = 0.0
for trial in range(60):
    K = int(rng.integers(1, 17)); T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 257])); B = int(rng.integers(1, 14))
    scale = float(rng.choice([0.3, 1.0, 5.0, 30.0]))
    init = np.log(rng.dirichlet(np.ones(K)))
    pair = np.log(rng.dirichlet(np.ones(K), size=K)) + 0.3 * rng.standard_normal((K, K))
    node = scale * rng.standard_normal((B, T, K))
    logZ, (Ei, Et, Es) = hmm_estep((init, pair, node))
    for b in range(B):
        lz, (oi, ot, os_) = hmm_numpy.hmm_estep((init, pair, node[b]))
        e = max(abs(float(logZ[b]) - lz) / max(1.0, abs(lz)), float(np.abs(Ei[b].cpu().numpy() - oi).max()),
                float(np.abs(Et[b].cpu().numpy() - ot).max()) / max(1, T), float(np.abs(Es[b].cpu().numpy() - os_).max()))
        worst = max(worst, e)
        assert e < 1e-9, (K, T, B, scale, b, e)
print("60 random HMM problems (K 1..16, T 1..257, B 1..13): worst deviation from the log-space oracle %.2e" % worst)
