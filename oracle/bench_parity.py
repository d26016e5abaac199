"""bench.py's parity gate ("parity gate before any timing is reported").
TEST / BENCH INFRASTRUCTURE ONLY -- the checker, never the thing measured.

    python -m oracle.bench_parity <file.npz>

The file holds, for a handful of sequences of the batch bench.py has just TIMED: the global natural parameters,
those sequences' node potentials as they sat in HBM, and the outputs the timed plan left for them
(lognorm, E_init, E_pair, E_node).  This process recomputes the same sequences with the reference's own compiled
E-step (oracle/_ref, `cython_natural_lds_estep_general`, svae/lds/lds_inference.py:232-237; kind "reference") or,
when that build is absent, the NumPy restatement (kind "port"), and prints one JSON object with the largest
relative deviation per quantity (relative to max(|want_ij|, 1e-3 max|want|), as the GPU tests measure it).
"""
import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = np.maximum(np.abs(b), 1e-3 * max(np.max(np.abs(b)), 1e-300))
    return float(np.max(np.abs(a - b) / scale)) if b.size else 0.0


GUARD = 1e-12


def _rel_plain(a, b):
    """plain element-wise |a - b| / |b| (no absolute floor) over the entries with |b| > GUARD * max|b| -- the guard only
    removes exact zeros and values at the level of rounding noise of the array (north_star: "1e-5 relative")"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if not b.size:
        return 0.0
    m = np.abs(b) > GUARD * max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


def check(path):
    from oracle import lds_numpy, ref
    d = np.load(path)
    kind, est = ("reference", ref.estep) if ref.available() else ("port", lds_numpy.natural_lds_estep_general)
    natparam = ((d["init_J"], d["init_h"], float(d["init_logZ"])),
                (d["J11"], d["J12"], d["J22"], float(d["logZ_pair"])))
    m, T, n = d["node_h"].shape
    worst = dict(lognorm=0.0, E_init=0.0, E_pair=0.0, E_node=0.0)
    plain = 0.0
    z = np.zeros(T)
    for j in range(m):
        ln, (Ei, Ep, En) = est(natparam, (d["node_J"][j], d["node_h"][j], z))
        worst["lognorm"] = max(worst["lognorm"], _rel(d["lognorm"][j], ln))
        worst["E_init"] = max(worst["E_init"], _rel(d["E_init"][j, :n * n].reshape(n, n), Ei[0]),
                              _rel(d["E_init"][j, n * n:], Ei[1]))
        worst["E_pair"] = max([worst["E_pair"]] + [_rel(d["E_pair"][j, i], np.asarray(Ep[i])) for i in range(3)])
        worst["E_node"] = max(worst["E_node"], _rel(d["E_node_diagxx"][j], En[0]), _rel(d["E_node_x"][j], En[1]))
        plain = max([plain, _rel_plain(d["lognorm"][j], ln), _rel_plain(d["E_init"][j, :n * n].reshape(n, n), Ei[0]),
                     _rel_plain(d["E_init"][j, n * n:], Ei[1]), _rel_plain(d["E_node_diagxx"][j], En[0]),
                     _rel_plain(d["E_node_x"][j], En[1])] + [_rel_plain(d["E_pair"][j, i], np.asarray(Ep[i])) for i in range(3)])
    return {"max_rel": max(worst.values()), "max_rel_note": "|a-b| / max(|b|, 1e-3 max|b|) per array",
            "max_rel_elementwise": plain, "max_rel_elementwise_note": "|a-b| / |b| over entries with |b| > %g max|b|" % GUARD,
            "per_quantity": worst, "sequences": int(m),
            "sequence_index": [int(i) for i in d["index"]], "checker": kind,
            "against": "oracle/_ref: the reference's compiled cython_natural_lds_estep_general" if kind == "reference"
                       else "oracle/lds_numpy.py (NumPy restatement; oracle/_ref not built)"}


if __name__ == "__main__":
    print(json.dumps(check(sys.argv[1])))
