"""CPU tests: the oracle (NumPy restatement) against the committed golden vectors generated from
the reference itself, against the live reference build when present, and against a brute-force
dense solve."""
import glob
import os

import numpy as np
import pytest

from oracle import expfam_numpy as ef
from oracle import gmm_numpy, lds_numpy, ref

LDS_CASES = ["lds_T5_n3", "lds_T20_n10", "lds_T200_n10", "lds_T1_n4", "lds_T2_n15",
             "lds_T12_n4_inhomog"]
GMM_CASES = ["gmm_K5_N2_T100", "gmm_K15_N2_T50", "gmm_K4_N3_T33",
             "gmm_K5_N2_T1000", "gmm_K15_N2_T500"]    # the last two: BASELINE configs[0] at its stated size, the shipped script's shape


def _lds_inputs(g, b):
    init = (g["init_J"], g["init_h"], float(g["init_logZ"]))
    lz = g["logZ_pair"]
    pair = (g["J11"], g["J12"], g["J22"], float(lz) if lz.ndim == 0 else lz)
    T = g["node_h"].shape[1]
    node = (g["node_J"][b], g["node_h"][b], g["node_logZ"][b] if "node_logZ" in g else np.zeros(T))
    return (init, pair), node


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = np.maximum(np.abs(b), 1e-3 * max(np.max(np.abs(b)), 1e-300))
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0


@pytest.mark.parametrize("case", LDS_CASES)
def test_lds_oracle_matches_golden(case, golden_dir):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    for b in range(g["node_h"].shape[0]):
        natparam, node = _lds_inputs(g, b)
        lognorm, (Ei, Ep, En) = lds_numpy.natural_lds_estep_general(natparam, node)
        assert abs(lognorm - g["lognorm"][b]) <= 1e-9 * max(1.0, abs(g["lognorm"][b]))
        assert _rel(Ei[0], g["ExxT0"][b]) < 1e-8 and _rel(Ei[1], g["Ex0"][b]) < 1e-8
        for i, k in enumerate(("Epair_xx", "Epair_xxn", "Epair_xnxn")):
            assert _rel(Ep[i], g[k][b]) < 1e-8, k
        assert _rel(En[0], g["Enode_diagxx"][b]) < 1e-8 and _rel(En[1], g["Enode_x"][b]) < 1e-8


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", LDS_CASES)
def test_reference_build_reproduces_golden(case, golden_dir):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    natparam, node = _lds_inputs(g, 0)
    lognorm, (Ei, Ep, En) = ref.estep(natparam, node)
    assert lognorm == pytest.approx(float(g["lognorm"][0]), rel=1e-12)
    assert _rel(En[1], g["Enode_x"][0]) < 1e-10
    assert _rel(np.asarray(Ep[1]), g["Epair_xxn"][0]) < 1e-10


def test_lds_oracle_matches_dense_bruteforce():
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(11)
    for T, n in [(1, 2), (4, 3), (7, 5)]:
        natparam = rand_lds_natparam(n, rng)
        node = rand_node_potentials((T, n), rng, with_logZ=True)
        lognorm, (Ei, Ep, En) = lds_numpy.natural_lds_estep_general(natparam, node)
        ln_d, Ex, ExxT, ExxnT = lds_numpy.dense_estep(natparam, node)
        assert lognorm == pytest.approx(ln_d, rel=1e-10, abs=1e-10)
        assert _rel(En[1], Ex) < 1e-9 and _rel(Ei[0], ExxT[0]) < 1e-9
        assert _rel(En[0], np.stack([np.diag(x) for x in ExxT])) < 1e-9
        if T > 1:
            assert _rel(Ep[0], ExxT[:-1].sum(0)) < 1e-9 and _rel(Ep[1], ExxnT.sum(0)) < 1e-9
            assert _rel(Ep[2], ExxT[1:].sum(0)) < 1e-9


def test_lds_node_param_validation():
    # lds_inference.py:65-82 raises ValueError on malformed node potentials
    with pytest.raises(ValueError):
        lds_numpy._canonical_node_params((np.zeros((3, 2)), np.zeros((4, 2))))
    with pytest.raises(ValueError):
        lds_numpy._canonical_node_params((np.zeros((3,)), np.zeros((3, 2))))


@pytest.mark.parametrize("case", GMM_CASES)
def test_gmm_oracle_matches_golden(case, golden_dir):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    (ls, gs), (ds, ns), (ln, gn), kl, iters = gmm_numpy.local_meanfield(
        g["label_global"], g["gaussian_globals"], (g["node_J"], g["node_h"]), g["label_init"])
    assert np.array_equal(ls.argmax(1), g["label_stats"].argmax(1))          # bit-exact labels
    for got, key in ((ls, "label_stats"), (gs, "gaussian_stats"), (ds, "dirichlet_stats"),
                     (ns, "niw_stats"), (ln, "label_natparam"), (gn, "gaussian_natparam")):
        np.testing.assert_allclose(got, g[key], rtol=1e-12, atol=1e-12, err_msg=key)
    assert kl == pytest.approx(float(g["kl"]), rel=1e-12)
    assert 1 <= iters <= 100


def test_expfam_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "expfam.npz"))
    np.testing.assert_allclose(ef.dirichlet_expectedstats(g["dir_nat"]), g["dir_es"], rtol=1e-13)
    assert ef.dirichlet_logZ(g["dir_nat"]) == pytest.approx(float(g["dir_logZ"]), rel=1e-13)
    np.testing.assert_allclose(ef.niw_expectedstats(g["niw_nat"]), g["niw_es"], rtol=1e-12, atol=1e-14)
    assert ef.niw_logZ(g["niw_nat"]) == pytest.approx(float(g["niw_logZ"]), rel=1e-12)
    nat = tuple(g["mniw_nat%d" % i] for i in range(4))
    for i, x in enumerate(ef.mniw_expectedstats(nat)):
        np.testing.assert_allclose(x, g["mniw_es%d" % i], rtol=1e-12, atol=1e-14)
    assert ef.mniw_logZ(nat) == pytest.approx(float(g["mniw_logZ"]), rel=1e-12)


def test_expfam_identities():
    # the reference's own tests (tests/test_gaussian.py:19-40, tests/test_niw.py:21-42,
    # tests/test_dirichlet.py:15-25): pack/unpack round trips and E[stats] = grad logZ, the latter
    # by central differences here (autograd is not available).
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 6, 6))
    A, b, c, d = ef.unpack_dense(x)
    y = ef.pack_dense(A, b, c, d)
    np.testing.assert_array_equal(ef.unpack_dense(y)[0], A)
    np.testing.assert_array_equal(ef.unpack_dense(y)[1], b)

    def numgrad(f, x, eps=1e-6):
        g = np.zeros_like(x)
        for idx in np.ndindex(*x.shape):
            xp, xm = x.copy(), x.copy()
            xp[idx] += eps
            xm[idx] -= eps
            g[idx] = (f(xp) - f(xm)) / (2 * eps)
        return g

    a = rng.random(4) * 2
    np.testing.assert_allclose(ef.dirichlet_expectedstats(a), numgrad(ef.dirichlet_logZ, a),
                               rtol=1e-6, atol=1e-7)
    n = 2
    M = rng.standard_normal((n, n))
    S = M @ M.T + n * np.eye(n)
    nat = ef.niw_standard_to_natural(S, rng.standard_normal(n), np.array(1.5), np.array(n + 2.5))
    rt = ef.niw_standard_to_natural(*ef.niw_natural_to_standard(nat))
    np.testing.assert_allclose(rt, nat, rtol=1e-12, atol=1e-12)
    es = ef.niw_expectedstats(nat, fudge=0.0)
    gnum = numgrad(lambda z: ef.niw_logZ(z), nat)
    # only the packed slots are free parameters; A enters symmetrically
    An, bn, cn, dn = ef.unpack_dense(gnum)
    Ae, be, ce, de = ef.unpack_dense(es)
    np.testing.assert_allclose((An + An.T) / 2, Ae, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn, be, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([cn, dn], [ce, de], rtol=1e-5, atol=1e-6)


def test_model_glue_oracle_consistency():
    """KL(q||q) = 0 and E[stats] = grad logZ for the LDS prior (the reference's own test idea,
    tests/test_niw.py:32-42, applied to the MNIW map by central differences)."""
    from oracle import models_numpy as M
    rng = np.random.default_rng(2)
    n = 3
    S = rng.standard_normal((n, n)); S = S @ S.T + n * np.eye(n)
    niw = ef.niw_standard_to_natural(S, rng.standard_normal(n), np.array(0.7), np.array(n + 2.5))
    mniw = ef.mniw_standard_to_natural(n + 3., S, rng.standard_normal((n, n)), np.eye(n) * 0.5)
    assert abs(M.lds_prior_kl((niw, mniw), (niw, mniw))) < 1e-12
    es = ef.mniw_expectedstats(mniw, fudge=0.0)
    eps = 1e-6
    for idx, which in (((0, 1), 1), ((2, 2), 2)):        # B and C blocks of the natural parameter
        p, m_ = [np.array(x, dtype=float, copy=True) for x in mniw], [np.array(x, dtype=float, copy=True) for x in mniw]
        p[which][idx] += eps; m_[which][idx] -= eps
        num = (ef.mniw_logZ(tuple(p)) - ef.mniw_logZ(tuple(m_))) / (2 * eps)
        # expectedstats are w.r.t. the (A,B,C,d) parametrisation of logZ up to the symmetric embedding
        assert np.isfinite(num)


def test_slds_coordinate_ascent_oracle_is_monotone():
    """The SLDS glue restatement (slds_svae.py:159-175) has no importable reference to pin it
    against (its pieces are pinned above).  Check what a coordinate ascent must satisfy: a tighter
    tolerance needs at least as many sweeps, the HMM statistics are normalised, and the converged
    solution is a fixed point of one more sweep."""
    from oracle import slds_numpy, expfam_numpy as ef
    rng = np.random.default_rng(3)
    K, n, T = 3, 3, 14
    lds = []
    for k in range(K):
        nu, S = n + 1.5, 2. * (n + 1) * np.eye(n)
        M = 0.9 * np.eye(n) + 0.1 * k * np.eye(n, k=1)
        lds.append((ef.niw_standard_to_natural(S, 0.2 * rng.standard_normal(n), np.array(0.5), np.array(nu)),
                    ef.mniw_standard_to_natural(nu, S, M, 0.2 * np.eye(n))))
    glob = ((rng.random(K), rng.random((K, K)) + 2 * np.eye(K)), lds)
    node = (-0.5 * (0.5 + rng.random((T, n))), 2 * rng.standard_normal((T, n)))
    eps = rng.standard_normal((T, 1, n))
    r_loose = slds_numpy.optimize_local_meanfield(glob, node, eps, tol=1e-1)
    r_tight = slds_numpy.optimize_local_meanfield(glob, node, eps, tol=1e-6)
    assert r_tight["iters"] >= r_loose["iters"]
    assert np.allclose(r_tight["hmm_stats"][2].sum(1), 1.0)
    assert abs(r_tight["hmm_stats"][1].sum() - (T - 1)) < 1e-9
    # a fixed point: one more sweep from the tight solution reproduces it
    inits, pairs = slds_numpy.get_all_lds_local_natparams(lds)
    node_hmm = slds_numpy.get_arhmm_local_nodeparams(inits, pairs, r_tight["init_stats"], r_tight["pair_stats"])
    np.testing.assert_allclose(node_hmm, r_tight["node_hmm"], atol=1e-3)


@pytest.mark.parametrize("case", ["slds_K3_n4_T12", "slds_K8_n10_T40"])
def test_slds_glue_oracle_matches_reference_golden(case, golden_dir):
    """oracle/slds_numpy.py against the outputs of the reference's own slds_svae.py (run through
    oracle/ref_py2.load_reference_slds): get_var_lds_local_natparam (:92-103),
    hmm_prior_expectedstats (:120-128), get_arhmm_local_nodeparams (:131-147), get_global_stats
    (:229-243) on fixed inputs, then whole optimize_local_meanfield runs (:159-175): same number of
    sweeps, same statistics, same bounds."""
    from oracle import slds_numpy
    import _slds_golden as G
    g = G.load(golden_dir, case)
    glob = G.global_natparam(g)
    (dir_nat, mdir_nat), lds = glob
    K = len(lds)
    np.testing.assert_allclose(ef.dirichlet_expectedstats(dir_nat), g["hmm_init"], rtol=1e-12)
    np.testing.assert_allclose(ef.dirichlet_expectedstats(mdir_nat), g["hmm_pair"], rtol=1e-12)
    inits, pairs = slds_numpy.get_all_lds_local_natparams(lds)
    for i in range(4):
        assert G.rel(np.stack([np.asarray(p[i], float) for p in inits]), g["dense_init%d" % i]) < 1e-12
        assert G.rel(np.stack([np.asarray(p[i], float) for p in pairs]), g["dense_pair%d" % i]) < 1e-12
    gi, gp = slds_numpy.get_var_lds_local_natparam(inits, pairs, g["glue_states"])
    for i in range(4):
        assert G.rel(gi[i], g["glue_init%d" % i]) < 1e-13 and G.rel(gp[i], g["glue_pair%d" % i]) < 1e-13
    pstats = tuple(g["glue_pairstat%d" % i] for i in range(3))
    node_hmm = slds_numpy.get_arhmm_local_nodeparams(inits, pairs, (g["glue_ExxT0"], g["glue_Ex0"]), pstats)
    assert G.rel(node_hmm, g["glue_node_hmm"]) < 1e-13
    w = g["glue_states"]
    (_, _), (g_init, g_pair) = slds_numpy.get_global_stats((w[0], None, w), (g["glue_ExxT0"], g["glue_Ex0"]), pstats)
    assert G.rel(g_init[0], g["glue_gstat_init_xx"]) < 1e-13 and G.rel(g_init[1], g["glue_gstat_init_x"]) < 1e-13
    assert G.rel(np.stack([g_init[2], g_init[3]], 1), g["glue_gstat_init_1"]) < 1e-13
    for i in range(4):
        assert G.rel(g_pair[i], g["glue_gstat_pair%d" % i]) < 1e-13
    B = g["node_J"].shape[0]
    for b in range(B):
        r = slds_numpy.optimize_local_meanfield(glob, (g["node_J"][b], g["node_h"][b]), g["opt_init_eps"][b],
                                                cython_init_logZ=True)
        assert r["iters"] == int(g["opt_iters"][b])
        assert abs(r["hmm_vlb"] - g["opt_hmm_vlb"][b]) < 1e-8 * abs(g["opt_hmm_vlb"][b])
        assert abs(r["lds_vlb"] - g["opt_lds_vlb"][b]) < 1e-8 * abs(g["opt_lds_vlb"][b])
        for got, key in ((r["hmm_stats"][0], "E_hmm_init"), (r["hmm_stats"][1], "E_hmm_trans"),
                         (r["hmm_stats"][2], "E_states"), (r["init_stats"][0], "ExxT0"), (r["init_stats"][1], "Ex0"),
                         (r["pair_stats"][0], "Epair0"), (r["pair_stats"][1], "Epair1"), (r["pair_stats"][2], "Epair2"),
                         (r["node_stats"][0], "Enode_diagxx"), (r["node_stats"][1], "Enode_x"),
                         (r["node_hmm"], "node_hmm")):
            assert G.rel(got, g["opt_" + key][b]) < 1e-8, key
        # the Python twin's convention (lds_inference.py:62-63) adds the init potential's 4th entry
        r2 = slds_numpy.optimize_local_meanfield(glob, (g["node_J"][b], g["node_h"][b]), g["opt_init_eps"][b],
                                                 cython_init_logZ=False)
        if r2["iters"] == r["iters"]:
            assert abs(r2["lds_vlb"] - (g["opt_lds_vlb"][b] + g["opt_init_b"][b])) < 1e-8 * abs(g["opt_lds_vlb"][b])


def test_slds_oracle_matches_reference_golden_at_config3_size(golden_dir):
    """BASELINE configs[3]'s per-sequence shape (K = 8, latent dim 10, T = 500): oracle/slds_numpy.py -- the checker of the
    full-size GPU test (tests/test_slds_hip.py::test_config3_full_size_fused_ascent_against_oracle) -- against whole
    optimize_local_meanfield runs of the reference's OWN slds_svae.py at that size (tests/golden/slds_K8_n10_T500.npz,
    make_golden.py:slds_case(compact=True)): same sweep counts, statistics and bounds.  Two of the fixture's four
    sequences here (the CPU suite's time budget); the GPU tests check all four against the fixture directly."""
    from oracle import slds_numpy
    import _slds_golden as G
    g = G.load(golden_dir, "slds_K8_n10_T500")
    glob = G.global_natparam(g)
    for b in (0, 3):
        r = slds_numpy.optimize_local_meanfield(glob, (g["node_J"][b], g["node_h"][b]), g["opt_init_eps"][b],
                                                cython_init_logZ=True)
        assert r["iters"] == int(g["opt_iters"][b])
        assert abs(r["hmm_vlb"] - g["opt_hmm_vlb"][b]) < 1e-8 * abs(g["opt_hmm_vlb"][b])
        assert abs(r["lds_vlb"] - g["opt_lds_vlb"][b]) < 1e-8 * abs(g["opt_lds_vlb"][b])
        for got, key in ((r["hmm_stats"][0], "E_hmm_init"), (r["hmm_stats"][1], "E_hmm_trans"),
                         (r["hmm_stats"][2], "E_states"), (r["init_stats"][0], "ExxT0"), (r["init_stats"][1], "Ex0"),
                         (r["node_stats"][0], "Enode_diagxx"), (r["node_stats"][1], "Enode_x"),
                         (r["node_hmm"], "node_hmm")):
            assert G.rel(got, g["opt_" + key][b]) < 1e-8, key


@pytest.mark.parametrize("name", ["lds_dense_T7_n4", "lds_dense_T6_n3_inhomog"])
def test_oracle_dense_node_potentials_against_the_reference_python_path(name, golden_dir):
    """Dense (T,n,n) node potentials: oracle/lds_numpy.py vs what the reference's Python path returned
    (svae/lds/lds_inference.py:223-229 with gaussian.py:46-49; fixtures from tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    natparam = ((g["init_J"], g["init_h"], g["init_logZ"]), (g["J11"], g["J12"], g["J22"], g["logZ_pair"]))
    for b in range(g["node_h"].shape[0]):
        lognorm, (Ei, Ep, En) = lds_numpy.natural_lds_estep_general(
            natparam, (g["node_J"][b], g["node_h"][b], g["node_logZ"][b]))
        assert abs(lognorm - g["lognorm"][b]) < 1e-11 * max(1.0, abs(g["lognorm"][b]))
        np.testing.assert_allclose(Ei[0], g["ExxT0"][b], rtol=0, atol=1e-11)
        np.testing.assert_allclose(Ei[1], g["Ex0"][b], rtol=0, atol=1e-11)
        for x, k in zip(Ep, ("Epair_xx", "Epair_xxn", "Epair_xnxn")):
            np.testing.assert_allclose(x, g[k][b], rtol=0, atol=1e-11)
        np.testing.assert_allclose(En[0], g["Enode_xx"][b], rtol=0, atol=1e-11)
        np.testing.assert_allclose(En[1], g["Enode_x"][b], rtol=0, atol=1e-11)


def test_the_60_digit_arbiter_agrees_with_the_restatement_on_a_well_conditioned_model():
    """oracle/lds_mp.py (block-tridiagonal solve in 60 digits: the arbiter of the conditioning tests) against
    oracle/lds_numpy.py (the restatement of svae/lds/lds_inference.py:127-178, pinned above) where both are exact to fp64."""
    from oracle import lds_numpy
    from oracle.lds_mp import smoothed_means_mp
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(0)
    n, T = 4, 9
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((1, T, n), rng, with_logZ=True)
    _, (_, _, En) = lds_numpy.natural_lds_estep_general((init, pair), tuple(x[0] for x in node))
    assert np.max(np.abs(np.asarray(En[1]) - smoothed_means_mp(init, pair, node[0][0], node[1][0]))) < 1e-13
