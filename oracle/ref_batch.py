"""The reference's compiled E-step / VJPs on EVERY sequence of a batch, spread over the host cores.
TEST / BENCH INFRASTRUCTURE ONLY (the checker of the full-size parity tests and of bench.py's parity gate).

`estep_all` / `estep_vjp_all` are called from a process that may have initialised the HIP runtime (a forked pool
must not inherit it), so they hand the inputs to a SEPARATE interpreter through an .npz file:

    python -m oracle.ref_batch <in.npz> <out.npz>

which forks one worker per core the box grants and runs, per sequence, exactly what oracle/ref.py wraps:
`cython_natural_lds_estep_general` (svae/lds/lds_inference.py:232-237 = cython_lds_inference.pyx:28-90 + 149-210)
and the composite VJP wired as lds_inference.py:26-39 (cython_lds_inference.pyx:92-145, 236-306, 357-409).
About 0.8 ms (E-step) / 2.5 ms (VJP with one sample) per sequence and core at T = 200, n = 10: all 4096
sequences of north_star's batch cost a few seconds.
"""
import math
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_G = {}


def _procs():
    from oracle.cpu_baseline import affinity_cores, quota_cores
    q = quota_cores()
    n = affinity_cores() if q is None else min(affinity_cores(), int(math.ceil(q)))
    return max(1, min(n, 32))


def _natparam(d):
    init = (d["init_J"], d["init_h"], float(d["init_logZ"]))
    pair = (d["J11"], d["J12"], d["J22"], d["logZ_pair"] if d["logZ_pair"].ndim else float(d["logZ_pair"]))
    return init, pair


def _estep_chunk(idx):
    from oracle import ref
    d = _G["d"]
    natparam = _natparam(d)
    T, n = d["node_h"].shape[1:]
    out = dict(lognorm=np.empty(len(idx)), ExxT0=np.empty((len(idx), n, n)), Ex0=np.empty((len(idx), n)),
               Epair=np.empty((len(idx), 3, n, n)), Enode_diagxx=np.empty((len(idx), T, n)),
               Enode_x=np.empty((len(idx), T, n)))
    z = np.zeros(T)
    for j, b in enumerate(idx):
        nz = d["node_logZ"][b] if "node_logZ" in d else z
        ln, (Ei, Ep, En) = ref.estep(natparam, (d["node_J"][b], d["node_h"][b], nz))
        out["lognorm"][j] = ln
        out["ExxT0"][j], out["Ex0"][j] = Ei[0], Ei[1]
        for i in range(3):
            out["Epair"][j, i] = np.asarray(Ep[i])
        out["Enode_diagxx"][j], out["Enode_x"][j] = En[0], En[1]
    return idx, out


def _vjp_chunk(idx):
    from oracle import ref
    d = _G["d"]
    natparam = _natparam(d)
    T, n = d["node_h"].shape[1:]
    S = d["g_s"].shape[2] if "g_s" in d else 0
    out = dict(gJ=np.empty((len(idx), T, n)), gh=np.empty((len(idx), T, n)), gz=np.empty((len(idx), T)))
    if S:
        out["eps"] = np.empty((len(idx), T, S, n))
    z = np.zeros(T)
    for j, b in enumerate(idx):
        nz = d["node_logZ"][b] if "node_logZ" in d else z
        (gJ, gh, gz), e = ref.estep_vjp(natparam, (d["node_J"][b], d["node_h"][b], nz), d["g_ln"][b],
                                        (d["g_dxx"][b], d["g_x"][b]), d["g_s"][b] if S else None,
                                        seed=int(d["seeds"][b]))
        out["gJ"][j], out["gh"][j], out["gz"][j] = gJ, gh, gz
        if S:
            out["eps"][j] = e
    return idx, out


def _main(src, dst):
    import multiprocessing as mp
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    d = dict(np.load(src))
    _G["d"] = d
    B = d["node_h"].shape[0]
    sel = d["select"].astype(int) if "select" in d else np.arange(B)
    fn = _vjp_chunk if "g_ln" in d else _estep_chunk
    procs = _procs()
    chunks = [c for c in np.array_split(sel, max(1, min(len(sel), 4 * procs))) if len(c)]
    if procs > 1 and len(chunks) > 1:
        with mp.get_context("fork").Pool(procs) as pool:          # no GPU runtime in this process
            parts = pool.map(fn, chunks, chunksize=1)
    else:
        parts = [fn(c) for c in chunks]
    out = {"index": np.concatenate([p[0] for p in parts])}
    for k in parts[0][1]:
        out[k] = np.concatenate([p[1][k] for p in parts])
    np.savez(dst, **out)


def _call(payload):
    payload = {k: np.asarray(v) for k, v in payload.items() if v is not None}
    with tempfile.TemporaryDirectory(prefix="svae_refbatch_") as tmp:
        src, dst = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        np.savez(src, **payload)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        r = subprocess.run([sys.executable, "-m", "oracle.ref_batch", src, dst], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=1800)
        if r.returncode != 0:
            raise RuntimeError("oracle.ref_batch failed: " + r.stderr[-800:])
        return dict(np.load(dst))


def _pack(natparam, node):
    init, pair = natparam
    d = dict(init_J=init[0], init_h=init[1], init_logZ=np.sum([np.asarray(z, float) for z in init[2:]]),
             J11=pair[0], J12=pair[1], J22=pair[2], logZ_pair=pair[3], node_J=node[0], node_h=node[1])
    if len(node) > 2 and node[2] is not None:
        d["node_logZ"] = node[2]
    return d


def estep_all(natparam, node, select=None):
    """ref.estep on every sequence (or on `select`) of node = (J (B,T,n), h (B,T,n)[, logZ (B,T)]) ->
    dict(index, lognorm (m), ExxT0 (m,n,n), Ex0 (m,n), Epair (m,3,n,n), Enode_diagxx (m,T,n), Enode_x (m,T,n))."""
    d = _pack(natparam, node)
    d["select"] = select
    return _call(d)


def estep_vjp_all(natparam, node, g_ln, g_dxx, g_x, g_s=None, seeds=None, select=None):
    """ref.estep_vjp on every sequence (or on `select`); seeds[b] seeds the reference sampler's global RNG for
    sequence b -> dict(index, gJ (m,T,n), gh (m,T,n), gz (m,T)[, eps (m,T,S,n) = the noise the sampler drew])."""
    d = _pack(natparam, node)
    B = np.asarray(node[1]).shape[0]
    d.update(g_ln=g_ln, g_dxx=g_dxx, g_x=g_x, g_s=g_s, select=select,
             seeds=np.arange(B) if seeds is None else np.asarray(seeds))
    return _call(d)


def _slds_one(b):
    from oracle import slds_numpy
    glob, J, h, eps, kw = _G["slds"]
    r = slds_numpy.optimize_local_meanfield(glob, (J[b], h[b]), eps[b], **kw)
    return int(b), {k: r[k] for k in ("hmm_stats", "init_stats", "pair_stats", "node_stats", "hmm_vlb", "lds_vlb",
                                      "iters", "node_hmm")}


def _slds_main(src, dst):
    import multiprocessing as mp
    import pickle
    with open(src, "rb") as f:
        glob, J, h, eps, select, kw = pickle.load(f)
    _G["slds"] = (glob, J, h, eps, kw)
    procs = min(_procs(), len(select))
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_slds_one, list(select), chunksize=1)
    else:
        res = [_slds_one(b) for b in select]
    with open(dst, "wb") as f:
        pickle.dump(dict(res), f)


def slds_ascent_select(global_natparam, J, h, eps, select, **kw):
    """oracle/slds_numpy.optimize_local_meanfield (slds_svae.py:159-175 restated, pinned to the reference's own
    module through tests/golden/slds_*.npz) on the sequences `select` of a batch, one process per core ->
    {b: result dict}.  Only the selected rows travel."""
    import pickle
    select = [int(b) for b in select]
    J, h, eps = np.asarray(J), np.asarray(h), np.asarray(eps)
    rows = {b: i for i, b in enumerate(select)}
    payload = (global_natparam, J[select], h[select], eps[select], list(range(len(select))), kw)
    with tempfile.TemporaryDirectory(prefix="svae_refbatch_") as tmp:
        src, dst = os.path.join(tmp, "in.pkl"), os.path.join(tmp, "out.pkl")
        with open(src, "wb") as f:
            pickle.dump(payload, f)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        r = subprocess.run([sys.executable, "-m", "oracle.ref_batch", "--slds", src, dst], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=3600)
        if r.returncode != 0:
            raise RuntimeError("oracle.ref_batch --slds failed: " + r.stderr[-800:])
        with open(dst, "rb") as f:
            out = pickle.load(f)
    return {b: out[i] for b, i in rows.items()}


if __name__ == "__main__":
    if sys.argv[1] == "--slds":
        _slds_main(sys.argv[2], sys.argv[3])
    else:
        _main(sys.argv[1], sys.argv[2])
