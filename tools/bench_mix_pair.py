"""svae_slds_mix_pair_natparam_f64 alone (the per-step pair parameters of the converged SLDS mean field: 2.45 GB written at
configs[3]).  Usage: python tools/bench_mix_pair.py [B T n K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd.models import slds_svae

B, T, n, K = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 500, 10, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
w = torch.softmax(torch.randn(B, T, K, dtype=torch.float64, device=dev, generator=g), -1)
pair = tuple(torch.randn(K, n, n, dtype=torch.float64, device=dev, generator=g) for _ in range(3)) + \
    (torch.randn(K, dtype=torch.float64, device=dev, generator=g),)
init = (torch.randn(K, n, n, dtype=torch.float64, device=dev), torch.randn(K, n, dtype=torch.float64, device=dev))
for _ in range(2):
    _, out = slds_svae.get_var_lds_local_natparam(init, pair, w)
torch.cuda.synchronize()
ref = torch.tensordot(w[:4, 1:], pair[1], dims=1)
print("max abs err vs tensordot:", float((out[1][:4] - ref).abs().max()))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
for i in range(5):
    ev[i].record()
    _, out = slds_svae.get_var_lds_local_natparam(init, pair, w)
ev[5].record(); torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
gb = B * (T - 1) * (3 * n * n + 1) * 8 / 1e9
print("B=%d T=%d n=%d K=%d: %.3f ms best (%.2f TB/s written)" % (B, T, n, K, min(ms), gb / min(ms)), [round(x, 3) for x in ms])
