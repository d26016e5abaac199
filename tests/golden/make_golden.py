"""Generate the committed golden fixtures from the REFERENCE ITSELF (run where /root/reference
exists):  python tests/golden/make_golden.py

  lds_*.npz  : inputs + outputs of the reference's compiled E-step
               (cython_natural_lds_estep_general, svae/lds/lds_inference.py:232-237, built from
               svae/lds/cython_lds_inference.pyx by oracle/build_ref.py), one entry per sequence.
  gmm_*.npz  : inputs + outputs of the reference's own svae/models/gmm.py:local_meanfield, run
               through oracle/ref_py2.py (lib2to3 in memory + autograd stand-in), with the
               global NumPy RNG seeded so that `initialize_meanfield` (gmm.py:126-128) is
               reproducible; the drawn initial responsibilities are stored too.
  expfam.npz : niw / mniw / dirichlet expectedstats and logZ from svae/distributions/*.py.

Fixtures are small (< 1 MB total) and are what the GPU box checks against (it has no
/root/reference).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref, ref_py2                      # noqa: E402
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials  # noqa: E402


def lds_case(name, B, T, n, seed, inhomog=False, with_logZ=True):
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    if inhomog:
        pairs = [rand_lds_natparam(n, rng)[1] for _ in range(T - 1)]
        pair = tuple(np.stack([p[i] for p in pairs]) for i in range(4))
    node = rand_node_potentials((B, T, n), rng, with_logZ=with_logZ)
    out = dict(init_J=init[0], init_h=init[1], init_logZ=init[2], J11=pair[0], J12=pair[1],
               J22=pair[2], logZ_pair=np.asarray(pair[3]), node_J=node[0], node_h=node[1])
    if with_logZ:
        out["node_logZ"] = node[2]
    res = []
    for b in range(B):
        nb = tuple(x[b] for x in node) if with_logZ else (node[0][b], node[1][b], np.zeros(T))
        res.append(ref.estep((init, pair), nb))
    out["lognorm"] = np.array([r[0] for r in res])
    out["ExxT0"] = np.stack([r[1][0][0] for r in res])
    out["Ex0"] = np.stack([r[1][0][1] for r in res])
    for i, k in enumerate(("Epair_xx", "Epair_xxn", "Epair_xnxn")):
        out[k] = np.stack([np.asarray(r[1][1][i]) for r in res])
    out["Enode_diagxx"] = np.stack([r[1][2][0] for r in res])
    out["Enode_x"] = np.stack([r[1][2][1] for r in res])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gmm_case(name, K, N, T, seed, alpha=1.0, random_scale=1.0):
    ref_py2.load_reference()
    from svae.models import gmm
    from svae.distributions import dirichlet, niw
    np.random.seed(seed)
    g = gmm.init_pgm_param(K, N, alpha=alpha, niw_conc=10., random_scale=random_scale)
    rng = np.random.default_rng(seed)
    node = rand_node_potentials((T, N), rng)
    st = np.random.get_state()
    label_init = np.random.rand(T, K)
    label_init = label_init / np.sum(label_init, axis=-1, keepdims=True)   # util.normalize
    np.random.set_state(st)
    (label_stats, gaussian_stats), (dir_stats, niw_stats), (label_nat, gauss_nat), kl = \
        gmm.local_meanfield(g, node)
    out = dict(dirichlet_natparam=g[0], niw_natparam=g[1],
               label_global=dirichlet.expectedstats(g[0]), gaussian_globals=niw.expectedstats(g[1]),
               node_J=node[0], node_h=node[1], label_init=label_init,
               label_stats=label_stats, gaussian_stats=gaussian_stats, dirichlet_stats=dir_stats,
               niw_stats=niw_stats, label_natparam=label_nat, gaussian_natparam=gauss_nat,
               kl=np.asarray(kl))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "kl", kl)


def gmm_run_case(name, K, N, T, S, seed):
    """gmm.run_inference (svae/models/gmm.py:12-16) with the global RNG seeded; the draws it makes
    (rand(T,K) in initialize_meanfield, then randn(T,N,S) in gaussian.natural_sample,
    distributions/gaussian.py:27-33) are replayed and stored."""
    ref_py2.load_reference()
    from svae.models import gmm
    np.random.seed(seed)
    prior = gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5)
    glob = gmm.init_pgm_param(K, N, alpha=1.0, niw_conc=1.0, random_scale=3.0)
    node = rand_node_potentials((T, N), np.random.default_rng(seed))
    st = np.random.get_state()
    label_init = np.random.rand(T, K)
    label_init = label_init / np.sum(label_init, axis=-1, keepdims=True)
    draws = np.random.randn(T, N, S)
    np.random.set_state(st)
    samples, stats, global_kl, local_kl = gmm.run_inference(prior, glob, node, S)
    out = dict(prior_dir=prior[0], prior_niw=prior[1], glob_dir=glob[0], glob_niw=glob[1],
               node_J=node[0], node_h=node[1], label_init=label_init,
               eps=np.transpose(draws, (0, 2, 1)), samples=samples, dirichlet_stats=stats[0],
               niw_stats=stats[1], global_kl=np.asarray(global_kl), local_kl=np.asarray(local_kl))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "global_kl", global_kl, "local_kl", local_kl)


def expfam_case():
    ref_py2.load_reference()
    from svae.distributions import dirichlet, niw, mniw
    rng = np.random.default_rng(7)
    out = {}
    d = rng.random((3, 4)) * 3
    out["dir_nat"], out["dir_es"], out["dir_logZ"] = d, dirichlet.expectedstats(d), dirichlet.logZ(d)
    n = 3
    nats = []
    for _ in range(4):
        A = rng.standard_normal((n, n)); S = A @ A.T + n * np.eye(n)
        nats.append(niw.standard_to_natural(S, rng.standard_normal(n), np.array(rng.random() + .5),
                                            np.array(n + 2 + rng.random() * 3)))
    nats = np.stack(nats)
    out["niw_nat"], out["niw_es"], out["niw_logZ"] = nats, niw.expectedstats(nats), niw.logZ(nats)
    A = rng.standard_normal((n, n)); S = A @ A.T + n * np.eye(n)
    Kk = rng.standard_normal((n, n)); Kk = Kk @ Kk.T + np.eye(n)
    M = rng.standard_normal((n, n))
    nat = mniw.standard_to_natural(n + 3.5, S, M, Kk)
    es = mniw.expectedstats(nat)
    for i, x in enumerate(nat):
        out["mniw_nat%d" % i] = np.asarray(x)
    for i, x in enumerate(es):
        out["mniw_es%d" % i] = np.asarray(x)
    out["mniw_logZ"] = np.asarray(mniw.logZ(nat))
    np.savez_compressed(os.path.join(HERE, "expfam.npz"), **out)
    print("expfam ok")


if __name__ == "__main__":
    assert build_ref.build(), "reference build failed"
    lds_case("lds_T5_n3", B=2, T=5, n=3, seed=0)
    lds_case("lds_T20_n10", B=3, T=20, n=10, seed=1)
    lds_case("lds_T200_n10", B=2, T=200, n=10, seed=2, with_logZ=False)
    lds_case("lds_T1_n4", B=2, T=1, n=4, seed=3)
    lds_case("lds_T2_n15", B=1, T=2, n=15, seed=4)
    lds_case("lds_T12_n4_inhomog", B=2, T=12, n=4, seed=5, inhomog=True)
    gmm_case("gmm_K5_N2_T100", K=5, N=2, T=100, seed=0)
    gmm_case("gmm_K15_N2_T50", K=15, N=2, T=50, seed=1)
    gmm_case("gmm_K4_N3_T33", K=4, N=3, T=33, seed=2)
    gmm_run_case("gmm_run_K5_N2_T60", K=5, N=2, T=60, S=3, seed=4)
    expfam_case()
