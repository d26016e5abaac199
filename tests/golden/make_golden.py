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
  slds_*.npz : the SLDS glue of svae/models/slds_svae.py executed as shipped (oracle/ref_py2.py:
               load_reference_slds -- four dead import lines aliased, nothing else touched) on the
               reference's compiled LDS / HMM kernels: get_var_lds_local_natparam (:92-103),
               hmm_prior_expectedstats (:120-128), get_arhmm_local_nodeparams (:131-147),
               get_global_stats (:229-243), optimize_local_meanfield (:159-175, with the number of
               sweeps it took) and run_inference (:289-310, forward values), RNG draws replayed.

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


def lds_dense_case(name, B, T, n, seed, inhomog=False):
    """DENSE node potentials (T,n,n) through the reference's Python path -- natural_lds_estep_general,
    svae/lds/lds_inference.py:223-229 (natural_condition_on_general, gaussian.py:46-49; dense node statistics,
    lds_inference.py:163-166) -- which the compiled path does not take (cython_lds_inference.pyx:43)."""
    ref_py2.load_reference()
    import svae.lds.lds_inference as li
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    if inhomog:
        pairs = [rand_lds_natparam(n, rng)[1] for _ in range(T - 1)]
        pair = tuple(np.stack([p[i] for p in pairs]) for i in range(4))
    nJ, nh, nz = rand_node_potentials((B, T, n), rng, with_logZ=True)
    A = 0.3 * np.abs(nJ).mean() * rng.standard_normal((B, T, n, n)) / n
    A = A + np.swapaxes(A, -1, -2)
    idx = np.arange(n)
    A[..., idx, idx] = nJ                      # diagonal as drawn (negative), small symmetric off-diagonal part
    out = dict(init_J=init[0], init_h=init[1], init_logZ=init[2], J11=pair[0], J12=pair[1], J22=pair[2],
               logZ_pair=np.asarray(pair[3]), node_J=A, node_h=nh, node_logZ=nz)
    res = [li.natural_lds_estep_general((init, pair), (A[b], nh[b], nz[b])) for b in range(B)]
    out["lognorm"] = np.array([float(r[0]) for r in res])
    out["ExxT0"] = np.stack([np.asarray(r[1][0][0]) for r in res])
    out["Ex0"] = np.stack([np.asarray(r[1][0][1]) for r in res])
    if inhomog:
        for i, k in enumerate(("Epair_xx", "Epair_xxn", "Epair_xnxn")):
            out[k] = np.stack([np.stack([np.asarray(st[i]) for st in r[1][1]]) for r in res])
    else:
        for i, k in enumerate(("Epair_xx", "Epair_xxn", "Epair_xnxn")):
            out[k] = np.stack([np.asarray(r[1][1][i]) for r in res])
    out["Enode_xx"] = np.stack([np.asarray(r[1][2][0]) for r in res])
    out["Enode_x"] = np.stack([np.asarray(r[1][2][1]) for r in res])
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


def _slds_globals(K, n, rng):
    """Global natural parameters of an SLDS with K rotating/decaying dynamics (NIW / MNIW natural
    parameters through the reference's own standard_to_natural, distributions/niw.py:39-42,
    mniw.py:25-31)."""
    from svae.distributions import niw, mniw
    dir_nat = rng.random(K) * 2.
    mdir_nat = rng.random((K, K)) * 2. + 3. * np.eye(K)
    lds = []
    for k in range(K):
        nu, S = n + 1. + rng.random(), 2. * (n + 1) * np.eye(n)
        mu, kappa = 0.3 * rng.standard_normal(n), 0.5
        th = 0.4 * (k + 1)
        M = 0.95 * np.eye(n)
        M[:2, :2] = 0.95 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        lds.append((niw.standard_to_natural(S, mu, np.array(kappa), np.array(nu)),
                    tuple(np.asarray(x, float) for x in mniw.standard_to_natural(nu, S, M, 0.2 * np.eye(n)))))
    return (dir_nat, mdir_nat), lds


def slds_case(name, K, n, T, B, S, seed, compact=False):
    # compact (BASELINE configs[3]'s T = 500): without the per-step (T-1,n,n) arrays -- the glue functions' fixed-input
    # cases and the per-step pair statistics of the ascent (2.4 MB per sequence); everything else as the small cases
    slds = ref_py2.load_reference_slds()
    rng = np.random.default_rng(seed)
    glob = _slds_globals(K, n, rng)
    prior = _slds_globals(K, n, rng)
    (dir_nat, mdir_nat), lds = glob
    out = dict(dir_nat=dir_nat, mdir_nat=mdir_nat, niw_nat=np.stack([a for a, _ in lds]),
               prior_dir_nat=prior[0][0], prior_mdir_nat=prior[0][1],
               prior_niw_nat=np.stack([a for a, _ in prior[1]]))
    for i in range(4):
        out["mniw_nat%d" % i] = np.stack([np.asarray(m[i], float) for _, m in lds])
        out["prior_mniw_nat%d" % i] = np.stack([np.asarray(m[i], float) for _, m in prior[1]])
    node_J = -0.5 * (0.5 + rng.random((B, T, n)))
    node_h = rng.standard_normal((B, T, n)) * 2.
    out["node_J"], out["node_h"] = node_J, node_h

    # --- the pure glue functions on fixed random inputs ----------------------------------------
    hmm_init, hmm_pair = slds.hmm_prior_expectedstats(glob[0])
    out["hmm_init"], out["hmm_pair"] = hmm_init, hmm_pair
    all_init, all_pair = slds.get_all_lds_local_natparams(lds)
    for i in range(4):
        out["dense_init%d" % i] = np.stack([np.asarray(p[i], float) for p in all_init])
        out["dense_pair%d" % i] = np.stack([np.asarray(p[i], float) for p in all_pair])
    w = rng.random((T, K)); w /= w.sum(1, keepdims=True)
    if not compact:
      out["glue_states"] = w
      gi, gp = slds.get_var_lds_local_natparam(lds, (None, None, w))
      for i in range(4):
          out["glue_init%d" % i], out["glue_pair%d" % i] = np.asarray(gi[i], float), np.asarray(gp[i], float)
      x = rng.standard_normal((T, n))
      o = lambda a, b: a[..., :, None] * b[..., None, :]
      A = rng.standard_normal((T - 1, n, n)) * 0.1
      init_stats = (o(x[0], x[0]) + 0.3 * np.eye(n), x[0], 1., 1.)
      pair_stats = [o(x[:-1], x[:-1]) + A @ A.transpose(0, 2, 1), o(x[:-1], x[1:]) + A,
                    o(x[1:], x[1:]) + A.transpose(0, 2, 1) @ A, np.ones(T - 1)]
      out["glue_ExxT0"], out["glue_Ex0"] = init_stats[0], init_stats[1]
      for i in range(3):
          out["glue_pairstat%d" % i] = pair_stats[i]
      out["glue_node_hmm"] = slds.get_arhmm_local_nodeparams(lds, (init_stats, pair_stats))
      Etrans = rng.random((K, K))
      (ghi, ght), glds = slds.get_global_stats((w[0], Etrans, w), (init_stats, pair_stats))
      glds = list(glds)
      out["glue_gstat_hmm_init"], out["glue_gstat_hmm_trans"] = ghi, ght
      out["glue_gstat_init_xx"] = np.stack([g[0][0] for g in glds])
      out["glue_gstat_init_x"] = np.stack([g[0][1] for g in glds])
      out["glue_gstat_init_1"] = np.array([[g[0][2], g[0][3]] for g in glds])
      for i in range(4):
          out["glue_gstat_pair%d" % i] = np.stack([np.asarray(list(g[1])[i], float) for g in glds])

    # --- the coordinate ascent, as shipped ---------------------------------------------------
    # The compiled filter reads init_params[2] only (cython_lds_inference.pyx:32): with the
    # 4-tuple init potential (J, h, a, b) the SLDS builds, the `b` = 1/2 E log|J| term is NOT in
    # lds_vlb (the Python twin sums the tail, lds_inference.py:62-63).  Goldens hold the shipped
    # value; `*_init_b` is the dropped term sum_k E[z_0 = k] b_k for the other convention.
    calls = [0]
    estep0 = slds.hmm_estep

    def counting(natparam):
        calls[0] += 1
        return estep0(natparam)
    slds.hmm_estep = counting
    keys = ("iters", "hmm_vlb", "lds_vlb", "init_b", "E_hmm_init", "E_hmm_trans", "E_states", "ExxT0", "Ex0",
            "Epair0", "Epair1", "Epair2", "Enode_diagxx", "Enode_x", "init_eps", "node_hmm")
    acc = {k: [] for k in keys}
    bvec = out["dense_init3"]
    for b in range(B):
        node = (node_J[b], node_h[b], np.zeros(T))
        np.random.seed(seed * 1000 + b)
        acc["init_eps"].append(np.random.randn(T, 1, n)[::-1].copy())    # as oracle/ref.py:sample_backward
        np.random.seed(seed * 1000 + b)
        calls[0] = 0
        (hmm_stats, lds_stats), (hmm_nat, lds_nat), (hv, lv) = slds.optimize_local_meanfield(glob, node)
        acc["iters"].append(calls[0])
        acc["hmm_vlb"].append(hv); acc["lds_vlb"].append(lv)
        acc["init_b"].append(float(hmm_stats[2][0] @ bvec))
        acc["E_hmm_init"].append(hmm_stats[0]); acc["E_hmm_trans"].append(hmm_stats[1])
        acc["E_states"].append(hmm_stats[2])
        E_init, E_pair, E_node = lds_stats
        E_pair = list(E_pair)
        acc["ExxT0"].append(E_init[0]); acc["Ex0"].append(E_init[1])
        for i in range(3):
            acc["Epair%d" % i].append(np.asarray(E_pair[i]))
        acc["Enode_diagxx"].append(E_node[0]); acc["Enode_x"].append(E_node[1])
        acc["node_hmm"].append(hmm_nat[2])
    slds.hmm_estep = estep0
    for k in keys:
        if compact and k.startswith("Epair"):
            continue
        out["opt_" + k] = np.stack([np.asarray(v, float) for v in acc[k]])

    # --- run_inference (:289-310), forward values, one sequence --------------------------------
    node = (node_J[0], node_h[0], np.zeros(T))
    np.random.seed(seed + 77)
    out["run_init_eps"] = np.random.randn(T, 1, n)[::-1].copy()
    out["run_eps"] = np.random.randn(T, S, n)[::-1].copy()
    np.random.seed(seed + 77)
    samples, stats, gvlb, lvlb = slds.run_inference(prior, glob, node, S)
    (shi, sht), slds_g = stats
    slds_g = list(slds_g)
    out["run_samples"] = np.asarray(samples)
    out["run_global_vlb"], out["run_local_vlb"] = np.asarray(gvlb), np.asarray(lvlb)
    out["run_stat_hmm_init"], out["run_stat_hmm_trans"] = shi, sht
    out["run_stat_init_xx"] = np.stack([g[0][0] for g in slds_g])
    out["run_stat_init_x"] = np.stack([g[0][1] for g in slds_g])
    for i in range(4):
        out["run_stat_pair%d" % i] = np.stack([np.asarray(list(g[1])[i], float) for g in slds_g])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "iters", out["opt_iters"], "vlb", out["opt_hmm_vlb"] + out["opt_lds_vlb"], "local_vlb", lvlb)


CASES = [
    (lds_dense_case, "lds_dense_T7_n4", dict(B=2, T=7, n=4, seed=11)),
    (lds_dense_case, "lds_dense_T6_n3_inhomog", dict(B=2, T=6, n=3, seed=12, inhomog=True)),
    (lds_case, "lds_T5_n3", dict(B=2, T=5, n=3, seed=0)),
    (lds_case, "lds_T20_n10", dict(B=3, T=20, n=10, seed=1)),
    (lds_case, "lds_T200_n10", dict(B=2, T=200, n=10, seed=2, with_logZ=False)),
    (lds_case, "lds_T1_n4", dict(B=2, T=1, n=4, seed=3)),
    (lds_case, "lds_T2_n15", dict(B=1, T=2, n=15, seed=4)),
    (lds_case, "lds_T12_n4_inhomog", dict(B=2, T=12, n=4, seed=5, inhomog=True)),
    (gmm_case, "gmm_K5_N2_T100", dict(K=5, N=2, T=100, seed=0)),
    (gmm_case, "gmm_K15_N2_T50", dict(K=15, N=2, T=50, seed=1)),
    (gmm_case, "gmm_K4_N3_T33", dict(K=4, N=3, T=33, seed=2)),
    # BASELINE configs[0] at its stated size (K = 5, 2-D, 1000 points) and the shipped script's shape
    # (experiments/gmm_svae_synth.py:29-33: K = 15, 500 points), from the reference's own gmm.py (round 5)
    (gmm_case, "gmm_K5_N2_T1000", dict(K=5, N=2, T=1000, seed=5)),
    (gmm_case, "gmm_K15_N2_T500", dict(K=15, N=2, T=500, seed=6)),
    (gmm_run_case, "gmm_run_K5_N2_T60", dict(K=5, N=2, T=60, S=3, seed=4)),
    (slds_case, "slds_K3_n4_T12", dict(K=3, n=4, T=12, B=3, S=2, seed=11)),
    (slds_case, "slds_K8_n10_T40", dict(K=8, n=10, T=40, B=2, S=1, seed=12)),
    # BASELINE configs[3]'s shape per sequence (K = 8, latent dim 10, T = 500), from the reference's own slds_svae.py (round 6)
    (slds_case, "slds_K8_n10_T500", dict(K=8, n=10, T=500, B=4, S=1, seed=13, compact=True)),
]


if __name__ == "__main__":
    # python tests/golden/make_golden.py [case-name ...]   (no names: every fixture, incl. expfam.npz)
    assert build_ref.build(), "reference build failed"
    only = set(sys.argv[1:])
    for fn, name, kw in CASES:
        if not only or name in only:
            fn(name, **kw)
    if not only or "expfam" in only:
        expfam_case()
