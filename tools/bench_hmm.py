"""HMM E-step launch time by batch size (K = 8, T = 500: the SLDS configuration): python tools/bench_hmm.py [B ...]
   python tools/bench_hmm.py --K 64 [B ...]: another number of states (17 .. 64: the one-wavefront-per-sequence kernel)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.hmm.hmm_inference import hmm_estep
from svae_amd import _lib

dev = torch.device("cuda:0")
K, T = 8, 500
argv = sys.argv[1:]
if "--K" in argv:
    K = int(argv[argv.index("--K") + 1])
    del argv[argv.index("--K"):argv.index("--K") + 2]
rng = np.random.default_rng(0)
for B in [int(x) for x in argv] or [8, 128, 512, 2048]:
    init = torch.as_tensor(rng.standard_normal(K), device=dev)
    pair = torch.as_tensor(rng.standard_normal((K, K)), device=dev)
    node = torch.as_tensor(3.0 * rng.standard_normal((B, T, K)), device=dev)
    ws = torch.empty(int(_lib.load().svae_hmm_workspace_bytes(B, T, K)) // 8, dtype=torch.float64, device=dev)
    for _ in range(3):
        hmm_estep((init, pair, node), ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = hmm_estep((init, pair, node), ws)
    e1.record(); torch.cuda.synchronize()
    print("HMM E-step K=%d T=%d B=%4d: %.3f ms per call (logZ[0] %.6f)" % (K, T, B, e0.elapsed_time(e1) / 10, float(out[0][0])))
