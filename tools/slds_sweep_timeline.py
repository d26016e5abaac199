"""Per-sweep timeline of the fused SLDS ascent at BASELINE configs[3]: sequences still iterating and the device time of
the three launches of each sweep (events around the library calls).  Usage: python tools/slds_sweep_timeline.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import _lib
from svae_amd.models import slds_svae
from svae_amd.lds.synthetic_data import rand_slds_global_natparam

B, T, n, K = 2048, 500, 10, 8
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
glob = rand_slds_global_natparam(K, n, rng)
# (the global parameters live on the device, as in a training loop: the global -> local maps then run as kernels)
_d = lambda x: tuple(_d(y) for y in x) if isinstance(x, (tuple, list)) else torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
glob = _d(glob)
node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
        torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
lib = _lib.load()
marks = []


def wrap(obj, name, tag):
    fn = getattr(obj, name)

    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        marks.append((tag, a[0] if tag != "lds" else (a[5] if len(a) > 5 else None), e0, e1))
        return r
    setattr(obj, name, inner)


class LibProxy(object):
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        return getattr(self._lib, k)


proxy = LibProxy(lib)
wrap(proxy, "svae_slds_hmm_meanfield_f64", "hmm")
wrap(proxy, "svae_slds_sweep_glue_f64", "glue")
_lib.load = lambda: proxy
orig_launch = slds_svae.SLDSMeanfieldPlan.launch


def launch(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_launch(self, *a, **k); e1.record()
    marks.append(("lds", a[5] if len(a) > 5 else None, e0, e1))
    return r


slds_svae.SLDSMeanfieldPlan.launch = launch
for rep in range(2):
    marks.clear()
    torch.cuda.synchronize()
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    out = slds_svae.optimize_local_meanfield(glob, node, eps, pair_stats=False)
    s1 = torch.cuda.Event(enable_timing=True); s1.record()
    torch.cuda.synchronize()
print("whole ascent %.2f ms (device time between first and last event)" % s0.elapsed_time(s1))
sweep = 0
row = {}
for tag, cnt, e0, e1 in marks:
    row[tag] = e0.elapsed_time(e1)
    if tag == "hmm":
        row["n"] = cnt
        row["t0"] = s0.elapsed_time(e0)
    if tag == "glue":
        print("sweep %2d: %4d sequences  start %6.2f ms | hmm %.3f  lds %.3f  glue %.3f ms" % (
            sweep, row.get("n", -1), row.get("t0", 0), row.get("hmm", 0), row.get("lds", 0), row.get("glue", 0)))
        sweep += 1
        row = {}
