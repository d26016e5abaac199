"""Round-5 bring-up of the producer-wavefront SLDS mean-field kernel (csrc/lds_estep_twoend_rpcmix.hpp): the same LDS
mean-field step through the table kernel (rounds 2 - 4), the new kernel with reference producers and with MFMA producers;
prints per-output differences and, with --time, ms per launch at BASELINE configs[3]'s shape.
usage: python tools/slds_rpc_debug.py [--time] [--only rpc_ref|rpc_mfma]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svae_amd import _lib                                    # noqa: E402
from svae_amd.lds.synthetic_data import rand_slds_global_natparam  # noqa: E402
from svae_amd.models import slds_svae                        # noqa: E402

OPTS = {"tables": _lib.OPT_LAYOUT_SPLIT, "rpc_ref": _lib.OPT_LAYOUT_PACKED | _lib.OPT_PRODUCERS_OFF,
        "rpc_mfma": _lib.OPT_LAYOUT_PACKED, "default": 0}      # default: one-sequence consumer up to one sequence per CU
dev = torch.device("cuda:0")


def setup(K, n, T, B, seed=0):
    rng = np.random.default_rng(seed)
    glob = rand_slds_global_natparam(K, n, rng)
    d = lambda x: tuple(d(y) for y in x) if isinstance(x, (tuple, list)) else torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    _, _, dense_init, dense_pair = slds_svae.global_to_local_maps(d(glob), dev)
    node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
            torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
    w = rng.random((B, T, K)) ** 3 + 1e-3
    w = torch.as_tensor(w / w.sum(-1, keepdims=True), device=dev)
    return tuple(x.contiguous() for x in dense_init), tuple(x.contiguous() for x in dense_pair), node, w


def run(kernel, K, n, T, B, args, rows=None):
    dense_init, dense_pair, node, w = args
    plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev, options=OPTS[kernel])
    for buf in (plan.lognorm, plan.E_init, plan.E_node_diagxx, plan.E_node_x, plan.pair_contr):
        buf.fill_(-7.25)
    plan.launch(dense_init, dense_pair, w, node, rows)
    torch.cuda.synchronize()
    nodep = plan.hmm_nodeparams(dense_init, dense_pair)
    return dict(lognorm=plan.lognorm.clone(), E_init=plan.E_init.clone(), dxx=plan.E_node_diagxx.clone(),
                x=plan.E_node_x.clone(), pc=plan.pair_contr.clone(), nodep=nodep.clone(), info=int(plan.info.item()))


def compare(a, b, tag):
    worst = 0.0
    for k in ("lognorm", "E_init", "dxx", "x", "nodep"):
        d = (a[k] - b[k]).abs()
        scale = float(a[k].abs().max()) + 1e-300
        rel = float(d.max()) / scale
        worst = max(worst, rel)
        flag = "" if rel < 1e-9 else "   <-- MISMATCH"
        print("   %-8s %-8s max|diff|/max|ref| = %.2e%s" % (tag, k, rel, flag))
        if rel >= 1e-9:
            bad = (d.reshape(d.shape[0], -1).max(1).values > 1e-9 * scale).nonzero().flatten().tolist()
            print("            sequences off:", bad[:24])
            if k == "nodep":
                tb = (d[bad[0]].max(-1).values > 1e-9 * scale).nonzero().flatten().tolist()
                print("            sequence %d: steps off %s" % (bad[0], tb[:40]))
    return worst


def sweep():
    """which shapes the default dispatch (one-sequence consumer for small launches) gets right"""
    for K in (3, 8):
        for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            for T in (9, 12):
                args = setup(K, n, T, 5, seed=K + T + n)
                base = run("tables", K, n, T, 5, args)
                got = run("default", K, n, T, 5, args)
                rel = max(float((base[k] - got[k]).abs().max()) / (float(base[k].abs().max()) + 1e-300) for k in ("lognorm", "x", "dxx", "nodep", "E_init"))
                d = (base["nodep"] - got["nodep"]).abs().amax(dim=(0, 2))
                print("K=%d n=%2d T=%2d  %s  %.1e  nodep steps off: %s" % (K, n, T, "ok      " if rel < 1e-9 else "MISMATCH", rel,
                      (d > 1e-9 * float(base["nodep"].abs().max())).nonzero().flatten().tolist()), flush=True)


def main():
    if "--sweep" in sys.argv:
        return sweep()
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    kernels = [k for k in ("rpc_ref", "rpc_mfma", "default") if only in (None, k)]
    ok = True
    for (K, n, T, B, rows) in ((8, 10, 12, 8, None), (8, 10, 13, 8, None), (3, 4, 9, 5, None), (8, 10, 40, 21, "idx"),
                               (5, 7, 4, 16, None), (8, 10, 101, 70, "idx"), (8, 10, 500, 16, None)):
        args = setup(K, n, T, B, seed=K + T)
        r = None
        if rows == "idx":
            live = [b for b in range(B) if b % 5 != 2]
            live = live[1::2] + live[0::2]
            live = live[:4] + [-1] + live[4:] + [-1, -1, -1]
            r = torch.tensor(live, dtype=torch.int32, device=dev)
        base = run("tables", K, n, T, B, args, r)
        print("K=%d n=%d T=%d B=%d rows=%s  (table kernel info %d)" % (K, n, T, B, rows, base["info"]), flush=True)
        for kern in kernels:
            got = run(kern, K, n, T, B, args, r)
            w = compare(base, got, kern)
            ok = ok and w < 1e-9 and got["info"] == base["info"]
            sys.stdout.flush()
    print("ALL OK" if ok else "MISMATCHES ABOVE")
    if "--time" in sys.argv and ok:
        K, n, T = 8, 10, 500
        for B in (1, 8, 64, 256, 512, 1024, 2048):
            dense_init, dense_pair, node, w = setup(K, n, T, B, seed=1)
            for kern in ("tables", "rpc_mfma", "default"):
                plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev, options=OPTS[kern])
                for _ in range(3):
                    plan.launch(dense_init, dense_pair, w, node)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    plan.launch(dense_init, dense_pair, w, node)
                e1.record(); torch.cuda.synchronize()
                print("B=%5d T=%d %-9s %.3f ms per launch" % (B, T, kern, e0.elapsed_time(e1) / reps), flush=True)


if __name__ == "__main__":
    main()
