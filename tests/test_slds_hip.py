"""GPU tests of the SLDS-SVAE local inference (svae_amd/models/slds_svae.py): the coordinate ascent
between the HMM and LDS E-step kernels against the NumPy restatement of svae/models/slds_svae.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import expfam_numpy as ef, slds_numpy  # noqa: E402  (checker only)


def _np(x):
    return x.detach().cpu().numpy()


def _err(got, want):
    """relative to max(|want_ij|, 1e-3 max|want|): entries that cancel to ~0 are judged on the array's scale"""
    got, want = _np(got), np.asarray(want, float)
    scale = np.maximum(np.abs(want), 1e-3 * max(np.max(np.abs(want)), 1e-300))
    return float(np.max(np.abs(got - want) / scale))


def _close(got, want, tol=1e-6):
    err = _err(got, want)
    assert err < tol, err


def _globals(K, n, rng):
    dir_nat = rng.random(K) * 2.
    mdir_nat = rng.random((K, K)) * 2. + 3. * np.eye(K)
    lds = []
    for k in range(K):
        nu, S = n + 1. + rng.random(), 2. * (n + 1) * np.eye(n)
        mu, kappa = 0.3 * rng.standard_normal(n), 0.5
        th = 0.4 * (k + 1)
        M = 0.95 * np.eye(n)
        if n >= 2:
            M[:2, :2] = 0.95 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        else:
            M[0, 0] = 0.95 * np.cos(th)
        Kmat = 0.2 * np.eye(n)
        lds.append((ef.niw_standard_to_natural(S, mu, np.array(kappa), np.array(nu)),
                    ef.mniw_standard_to_natural(nu, S, M, Kmat)))
    return (dir_nat, mdir_nat), lds


def _nodes(B, T, n, rng):
    J = -0.5 * (0.5 + rng.random((B, T, n)))
    h = rng.standard_normal((B, T, n)) * 2.
    return J, h


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("fused", [False, True])
# (K = 20, 40: more discrete states than a DPP row holds -- the HMM step runs the one-wavefront-per-sequence kernel of round 6,
#  csrc/hmm_estep_wide.hip; the fused LDS mean-field kernel covers K <= 16, so fused=True falls back to the materialised path)
@pytest.mark.parametrize("K,n,T,B", [(3, 2, 12, 3), (4, 5, 30, 6), (2, 10, 16, 2), (5, 9, 7, 9), (1, 3, 5, 2),
                                     (20, 3, 10, 3), (40, 2, 8, 2)])
def test_optimize_local_meanfield_matches_oracle(K, n, T, B, fused, compat):
    """compat = True: the reference as shipped (its compiled filter drops the init potential's 4th entry; the default
    of both the library and the restatement since round 5); False: the convention of the reference's Python twin."""
    from svae_amd.models import slds_svae
    rng = np.random.default_rng(100 * K + n)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    eps = rng.standard_normal((B, T, 1, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    (hmm_stats, lds_stats), _, (hmm_vlb, lds_vlb), iters = slds_svae.optimize_local_meanfield(
        glob, node, eps, fused=fused, reference_compat=compat)
    for b in range(B):
        ref = slds_numpy.optimize_local_meanfield(glob, (J[b], h[b]), eps[b], cython_init_logZ=compat)
        assert int(iters[b]) == ref["iters"]
        assert float(hmm_vlb[b]) == pytest.approx(ref["hmm_vlb"], rel=1e-7, abs=1e-7)
        assert float(lds_vlb[b]) == pytest.approx(ref["lds_vlb"], rel=1e-7, abs=1e-7)
        _close(hmm_stats[2][b], ref["hmm_stats"][2])
        _close(hmm_stats[1][b], ref["hmm_stats"][1])
        for got, want in zip(lds_stats[0], ref["init_stats"]):
            _close(got[b], want)
        for got, want in zip(lds_stats[1], ref["pair_stats"]):
            _close(got[b], want)


@pytest.mark.parametrize("max_iter", [1, 2, 3, 5])
def test_fused_ascent_with_a_sweep_budget_matches_the_materialised_one(max_iter):
    """The fused ascent queues sweep i + 1 before the host has read the list length after sweep i (slots behind the
    list are -1 and skipped): stopping at the sweep budget, with sequences still iterating, must leave the same
    per-sequence sweep counts, bounds and statistics as the materialised loop (one blocking read per sweep)."""
    from svae_amd.models import slds_svae
    K, n, T, B = 4, 5, 30, 11
    rng = np.random.default_rng(max_iter)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    eps = rng.standard_normal((B, T, 1, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    out = [slds_svae.optimize_local_meanfield(glob, node, eps, fused=f, max_iter=max_iter, tol=1e-6) for f in (True, False)]
    (sf, _, (hv_f, lv_f), it_f), (sm, _, (hv_m, lv_m), it_m) = out
    assert torch.equal(it_f.cpu(), it_m.cpu()) and int(it_f.max()) == max_iter
    assert float((hv_f - hv_m).abs().max()) < 1e-8 and float((lv_f - lv_m).abs().max()) < 1e-7
    _close(sf[0][2], _np(sm[0][2]))


def test_optimize_local_meanfield_at_latent_dim_16():
    """Latent dimension beyond the DPP-row kernels (16 <= n <= 64): the initial sample path comes from the tile
    E-step + sampler (`natural_lds_sample` has no filter-only form there), the ascent from the materialised path."""
    from svae_amd.models import slds_svae
    K, n, T, B = 3, 16, 9, 3
    rng = np.random.default_rng(1600)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    eps = rng.standard_normal((B, T, 1, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    (hmm_stats, lds_stats), _, (hmm_vlb, lds_vlb), iters = slds_svae.optimize_local_meanfield(glob, node, eps)
    for b in range(B):
        ref = slds_numpy.optimize_local_meanfield(glob, (J[b], h[b]), eps[b])
        assert int(iters[b]) == ref["iters"]
        assert float(lds_vlb[b]) == pytest.approx(ref["lds_vlb"], rel=1e-7, abs=1e-7)
        _close(hmm_stats[2][b], ref["hmm_stats"][2])
        for got, want in zip(lds_stats[0], ref["init_stats"]):
            _close(got[b], want)


def test_run_inference_and_global_stats():
    from svae_amd.models import slds_svae
    K, n, T, B, S = 3, 4, 20, 4, 2
    rng = np.random.default_rng(7)
    glob, prior = _globals(K, n, rng), _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    init_eps, eps = rng.standard_normal((B, T, 1, n)), rng.standard_normal((B, T, S, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    samples, (hmm_g, (g_init, g_pair)), global_vlb, local_vlb = slds_svae.run_inference(
        prior, glob, node, S, init_eps=init_eps, eps=eps)
    assert tuple(samples.shape) == (B, T, S, n) and torch.isfinite(samples).all()
    want_hmm = [0., 0.]
    want_init, want_pair = None, None
    tot = 0.
    for b in range(B):
        ref = slds_numpy.optimize_local_meanfield(glob, (J[b], h[b]), init_eps[b])
        (Ei, Et), (gi, gp) = slds_numpy.get_global_stats(ref["hmm_stats"], ref["init_stats"], ref["pair_stats"])
        want_hmm = [want_hmm[0] + Ei, want_hmm[1] + Et]
        want_init = gi if want_init is None else tuple(x + y for x, y in zip(want_init, gi))
        want_pair = gp if want_pair is None else tuple(x + y for x, y in zip(want_pair, gp))
    np.testing.assert_allclose(_np(hmm_g[0]), want_hmm[0], rtol=1e-6)
    _close(hmm_g[1], want_hmm[1])
    for got, want in zip(g_init, want_init):
        _close(got, want)
    for got, want in zip(g_pair, want_pair):
        _close(got, want)
    assert np.isfinite(float(global_vlb)) and np.isfinite(float(local_vlb))


def test_withlabels_matches_oracle():
    from svae_amd.models import slds_svae
    K, n, T, B = 3, 3, 15, 3
    rng = np.random.default_rng(11)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    labels = rng.integers(0, K, (B, T))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    (hmm_stats, lds_stats), _, (_, lds_vlb) = slds_svae.optimize_local_meanfield_withlabels(glob, node, labels)
    for b in range(B):
        ref = slds_numpy.optimize_local_meanfield_withlabels(glob, (J[b], h[b]), labels[b])
        assert float(lds_vlb[b]) == pytest.approx(ref["lds_vlb"], rel=1e-8)
        np.testing.assert_allclose(_np(hmm_stats[1][b]), ref["hmm_stats"][1])
        np.testing.assert_allclose(_np(hmm_stats[2][b]), ref["hmm_stats"][2], rtol=1e-12)
        for got, want in zip(lds_stats[1], ref["pair_stats"]):
            _close(got[b], want)
        for got, want in zip(lds_stats[2], ref["node_stats"]):
            _close(got[b], want)


def _fused_options(kernel):
    """kernel-selection word of svae_slds_lds_meanfield_f64: the library's choice (K <= 8: up to one sequence per CU the
    one-sequence consumer next to three MFMA producers, above the row-per-chain consumers with four); the table kernel of
    rounds 2 - 4 (one sequence per wavefront); the row-per-chain kernel whatever the batch, with reference producers
    (plain loops) and with the MFMA producers"""
    from svae_amd import _lib
    return {"default": 0, "tables": _lib.OPT_LAYOUT_SPLIT, "rpc_ref": _lib.OPT_LAYOUT_PACKED | _lib.OPT_PRODUCERS_OFF,
            "rpc_mfma": _lib.OPT_LAYOUT_PACKED}[kernel]


@pytest.mark.parametrize("kernel", ["tables", "rpc_ref", "rpc_mfma", "default"])
@pytest.mark.parametrize("K,n,T,B", [(3, 4, 9, 5), (8, 10, 40, 11), (7, 10, 5, 3), (16, 6, 12, 4), (2, 2, 4, 2),
                                     (5, 7, 13, 21), (8, 10, 6, 9), (8, 9, 4, 17), (1, 3, 7, 8), (8, 10, 31, 40),
                                     (8, 4, 16, 6), (3, 2, 8, 3), (8, 5, 10, 4), (8, 8, 11, 7), (6, 4, 8, 300),
                                     (4, 1, 6, 3), (8, 1, 9, 20), (8, 3, 12, 5), (8, 6, 12, 5), (8, 7, 12, 5)])
def test_fused_lds_meanfield_step_matches_materialised(K, n, T, B, kernel):
    """One LDS mean-field step through svae_slds_lds_meanfield_f64 (K parameter sets in LDS, mixed per step
    by the HMM marginals, pair statistics contracted in the kernel) against the path that materialises the
    per-step pair parameters and statistics (get_var_lds_local_natparam / get_arhmm_local_nodeparams,
    slds_svae.py:92-103, 131-147, themselves pinned against the NumPy restatement above); sequences not
    listed in `seq_index` (frozen ones) keep their buffers."""
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.models import slds_svae
    rng = np.random.default_rng(1000 * K + 10 * n + T)
    (_, _), lds = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    node = (t(J), t(h), t(rng.standard_normal((B, T))))
    lds_d = [(t(a), tuple(t(y) for y in m)) for a, m in lds]
    dense_init, dense_pair = slds_svae.get_all_lds_local_natparams(lds_d)
    dense_init = tuple(x.to(dev) for x in dense_init)
    dense_pair = tuple(x.to(dev) for x in dense_pair)
    w = rng.random((B, T, K)) ** 3 + 1e-3
    w = t(w / w.sum(-1, keepdims=True))
    assert slds_svae.SLDSMeanfieldPlan.supported(n, T, K)
    if K > 8 and kernel.startswith("rpc"):
        pytest.skip("the producer-wavefront kernel covers K <= 8")
    plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev, options=_fused_options(kernel))
    sentinel = -7.25
    for buf in (plan.lognorm, plan.E_init, plan.E_node_diagxx, plan.E_node_x, plan.pair_contr):
        buf.fill_(sentinel)
    rows = [b for b in range(B) if b != B // 2]
    rows = rows[1::2] + rows[0::2]                                                   # any order
    if B >= 8:                       # an unused slot (negative entry) in the middle of the list
        rows = rows[:3] + [-1] + rows[3:]
    rows = torch.tensor(rows, dtype=torch.int32, device=dev)
    plan.launch(dense_init, dense_pair, w, node, rows)
    torch.cuda.synchronize()
    assert int(plan.info.item()) == 0
    # materialised path
    lds_init, lds_pair = slds_svae.get_var_lds_local_natparam(dense_init, dense_pair, w)
    mplan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
    lognorm, (Ei, Ep, En) = slds_svae._lds_estep_batched_init(mplan, lds_init, lds_pair, node)
    node_hmm = slds_svae.get_arhmm_local_nodeparams(dense_init, dense_pair, (Ei[0], Ei[1]), (Ep[0], Ep[1], Ep[2]))
    got_vlb = plan.lds_vlb(dense_init, dense_pair, w)
    got_hmm = plan.hmm_nodeparams(dense_init, dense_pair)
    for b in range(B):
        if b == B // 2:
            for buf in (plan.lognorm, plan.E_init, plan.E_node_diagxx, plan.E_node_x, plan.pair_contr):
                assert bool((buf[b] == sentinel).all()), "frozen sequence was rewritten"
            continue
        assert float(got_vlb[b]) == pytest.approx(float(lognorm[b]), rel=1e-10, abs=1e-9)
        _close(plan.E_init[b, :n * n].reshape(n, n), _np(Ei[0][b]), 1e-8)
        _close(plan.E_init[b, n * n:], _np(Ei[1][b]), 1e-8)
        _close(plan.E_node_diagxx[b], _np(En[0][b]), 1e-8)
        _close(plan.E_node_x[b], _np(En[1][b]), 1e-8)
        _close(got_hmm[b], _np(node_hmm[b]), 1e-8)


@pytest.mark.parametrize("fused", [False, True])
def test_config3_shape_properties(fused):
    """BASELINE configs[3] shape (K=8 states, latent dim 10, T=500), a slice of the batch:
    size-independent properties of the converged local mean field plus parity of one sequence."""
    from svae_amd.models import slds_svae
    K, n, T, B = 8, 10, 500, 24
    rng = np.random.default_rng(3)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    eps = rng.standard_normal((B, T, 1, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    (hmm_stats, lds_stats), _, (hmm_vlb, lds_vlb), iters = slds_svae.optimize_local_meanfield(glob, node, eps,
                                                                                             fused=fused)
    Ei, Et, Es = hmm_stats
    assert int(iters.max()) < 100
    assert torch.allclose(Es.sum(-1), torch.ones_like(Es.sum(-1)), atol=1e-12)
    assert torch.allclose(Et.sum((-1, -2)), torch.full((B,), T - 1., dtype=torch.float64, device=dev), atol=1e-9)
    assert torch.allclose(Ei, Es[:, 0], atol=1e-12)
    var = lds_stats[2][0] - lds_stats[2][1] ** 2
    assert float(var.min()) > 0 and torch.isfinite(hmm_vlb + lds_vlb).all()
    b = 5
    ref = slds_numpy.optimize_local_meanfield(glob, (J[b], h[b]), eps[b])
    assert int(iters[b]) == ref["iters"]
    _close(Es[b], ref["hmm_stats"][2], 1e-5)
    for got, want in zip(lds_stats[1], ref["pair_stats"]):
        _close(got[b], want, 1e-5)


def test_config3_full_size_fused_ascent_against_oracle():
    """BASELINE configs[3] at FULL size (K = 8, latent dim 10, 2048 sequences x T = 500) through the fused LDS
    mean-field kernel: size-independent properties over the whole batch, and parity of the converged mean field
    (iteration counts, HMM marginals, bounds, node statistics) for sequences spread over the batch against the
    NumPy restatement (32 sequences at 1e-8: observed 4e-10; north_star asks 1e-5)."""
    from svae_amd.models import slds_svae
    K, n, T, B = 8, 10, 500, 2048
    rng = np.random.default_rng(3)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    eps = rng.standard_normal((B, T, 1, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    assert slds_svae.SLDSMeanfieldPlan.supported(n, T, K)
    (hmm_stats, lds_stats), _, (hmm_vlb, lds_vlb), iters = slds_svae.optimize_local_meanfield(
        glob, node, torch.as_tensor(eps, device=dev), fused=True, pair_stats=False)
    Ei, Et, Es = hmm_stats
    assert int(iters.max()) < 100 and int(iters.min()) >= 1
    assert torch.allclose(Es.sum(-1), torch.ones_like(Es.sum(-1)), atol=1e-12)
    assert torch.allclose(Et.sum((-1, -2)), torch.full((B,), T - 1., dtype=torch.float64, device=dev), atol=1e-9)
    assert torch.allclose(Ei, Es[:, 0], atol=1e-12)
    var = lds_stats[2][0] - lds_stats[2][1] ** 2
    assert float(var.min()) > 0 and torch.isfinite(hmm_vlb + lds_vlb).all()
    # 32 sequences spread over the batch (round 5; rounds 2 - 4: four), the restatement on all host cores
    # (oracle/ref_batch.py)
    from oracle import ref_batch
    picks = sorted(set(np.linspace(0, B - 1, 30).astype(int).tolist()) | {1, B - 2})
    refs = ref_batch.slds_ascent_select(glob, J, h, eps, picks)
    worst = 0.0
    for b in picks:
        ref = refs[b]
        assert int(iters[b]) == ref["iters"], b
        assert float(hmm_vlb[b]) == pytest.approx(ref["hmm_vlb"], rel=1e-8, abs=1e-7)
        assert float(lds_vlb[b]) == pytest.approx(ref["lds_vlb"], rel=1e-8, abs=1e-7)
        pairs = [(Es[b], ref["hmm_stats"][2]), (lds_stats[2][0][b], ref["node_stats"][0]),
                 (lds_stats[2][1][b], ref["node_stats"][1])]
        pairs += [(got[b], want) for got, want in zip(lds_stats[0], ref["init_stats"])]
        worst = max(worst, max(_err(got, want) for got, want in pairs))
    print("configs[3] full size, %d sequences vs the restatement: worst rel err %.2e" % (len(picks), worst))
    assert worst < 1e-8, worst               # observed 3.8e-10 (round 5); north_star asks 1e-5


@pytest.mark.parametrize("fused_contraction", [False, True])
def test_final_pass_gradient_against_finite_differences(fused_contraction):
    """d(local_vlb + <g, samples>)/d(nn_potentials) through the VJP kernels (per-step pair parameters,
    cotangents of E_init / E_pair from the HMM bound) against central differences of the forward.
    fused_contraction: the HMM node potentials through svae_slds_pair_contract_f64 with its hand-written backward
    (models.slds_svae._PairContract: what run_inference_differentiable runs) instead of the library GEMM under autograd."""
    from svae_amd.models import slds_svae
    K, n, T, B, S = 3, 3, 6, 2, 2
    rng = np.random.default_rng(21)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    node = (t(J), t(h))
    (hmm_stats, _), (hmm_nat, lds_nat), _, _ = slds_svae.optimize_local_meanfield(glob, node, t(rng.standard_normal((B, T, 1, n))))
    eps = t(rng.standard_normal((B, T, S, n)))
    gs = t(rng.standard_normal((B, T, S, n)))

    def f(nJ, nh):
        if fused_contraction:
            samples, _, local_vlb, sums = slds_svae.final_pass_differentiable(glob, hmm_nat, lds_nat, (nJ, nh), eps,
                                                                             expected_states=hmm_stats[2])
            assert sums is not None and tuple(sums.shape) == (K, 3, n, n)
        else:
            samples, _, local_vlb = slds_svae.final_pass_differentiable(glob, hmm_nat, lds_nat, (nJ, nh), eps)
        return local_vlb + (gs * samples).sum()

    nJ, nh = node[0].clone().requires_grad_(True), node[1].clone().requires_grad_(True)
    f(nJ, nh).backward()
    d = 1e-6
    for name, x, grad in (("J", node[0], nJ.grad), ("h", node[1], nh.grad)):
        fd = torch.zeros_like(x)
        for idx in np.ndindex(*x.shape):
            xp, xm = x.clone(), x.clone()
            xp[idx] += d
            xm[idx] -= d
            args_p = (xp, node[1]) if name == "J" else (node[0], xp)
            args_m = (xm, node[1]) if name == "J" else (node[0], xm)
            fd[idx] = (f(*args_p) - f(*args_m)).detach() / (2 * d)
        err = float((grad - fd).abs().max() / fd.abs().max())
        assert err < 1e-6, (name, err)


def test_run_inference_differentiable_matches_forward_values():
    from svae_amd.models import slds_svae
    K, n, T, B, S = 3, 4, 10, 3, 2
    rng = np.random.default_rng(9)
    glob, prior = _globals(K, n, rng), _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    init_eps, eps = rng.standard_normal((B, T, 1, n)), rng.standard_normal((B, T, S, n))
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    s0, st0, g0, l0 = slds_svae.run_inference(prior, glob, node, S, init_eps=init_eps, eps=eps)
    nd = tuple(x.clone().requires_grad_(True) for x in node)
    s1, st1, g1, l1 = slds_svae.run_inference_differentiable(prior, glob, nd, S, init_eps=init_eps, eps=eps)
    assert torch.allclose(s0, s1, rtol=1e-10, atol=1e-12)
    assert float(l1.detach()) == pytest.approx(float(l0), rel=1e-10) and float(g1) == pytest.approx(float(g0), rel=1e-12)
    for a, b in zip(st0[1][1], st1[1][1]):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-10)
    (l1 + s1.sum()).backward()
    assert torch.isfinite(nd[0].grad).all() and torch.isfinite(nd[1].grad).all() and float(nd[1].grad.abs().max()) > 0


def test_make_gradfun_with_the_slds_model():
    """One training-step gradient of an SLDS-SVAE through svae.make_gradfun: recognition net ->
    run_inference_differentiable -> decoder; the PGM natural gradient lines up with the global
    natural parameters' structure."""
    from functools import partial
    from svae_amd import svae
    from svae_amd.models import slds_svae
    K, n, T, B, S, p = 2, 3, 8, 4, 1, 5
    rng = np.random.default_rng(4)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    to_t = lambda s: tuple(to_t(x) for x in s) if isinstance(s, (tuple, list)) else t(s)
    glob, prior = to_t(_globals(K, n, rng)), to_t(_globals(K, n, rng))
    data = t(rng.standard_normal((2 * B, T, p)))
    W_r = (t(0.3 * rng.standard_normal((p, n))).requires_grad_(True), t(0.3 * rng.standard_normal((p, n))).requires_grad_(True))
    W_d = (t(0.3 * rng.standard_normal((n, p))).requires_grad_(True),)

    def recognize(params, batch):            # (J diag <= 0, h)
        return -0.5 * torch.nn.functional.softplus(batch @ params[0]) - 0.1, batch @ params[1]

    def loglike(params, samples, batch):     # unit-variance Gaussian decoder, mean over samples
        mean = samples @ params[0]
        return -0.5 * ((batch[:, :, None, :] - mean) ** 2).sum() / samples.shape[2]

    gen = torch.Generator(device=dev).manual_seed(0)
    def run(prior_, glob_, pots, S_):
        samples, stats, gv, lv = slds_svae.run_inference_differentiable(prior_, glob_, pots, S_, generator=gen)
        return samples, slds_svae.global_stats_as_natparam(stats), -gv, -lv      # vlb -> kl sign of svae.py

    gradfun = svae.make_gradfun(run, recognize, loglike, prior, data, B, S, callback=None, permute=False)
    natgrad, g_dec, g_rec = gradfun((glob, W_d, W_r), 0)
    assert svae.flat(natgrad).shape == svae.flat(glob).shape
    assert torch.isfinite(svae.flat(natgrad)).all()
    assert all(torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in g_rec + g_dec)


# ---- against the REFERENCE's own slds_svae.py (tests/golden/slds_*.npz, see tests/_slds_golden.py) ----

def _t(x, dev):
    return torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)


@pytest.mark.parametrize("case", ["slds_K3_n4_T12", "slds_K8_n10_T40"])
def test_glue_functions_against_reference_golden(case, golden_dir):
    """get_all_lds_local_natparams / hmm_prior_expectedstats / get_var_lds_local_natparam /
    get_arhmm_local_nodeparams / get_global_stats of the product (torch, device) against the outputs of the
    reference's slds_svae.py:86-147, 229-243 on the same inputs."""
    from svae_amd.models import slds_svae
    import _slds_golden as G
    g = G.load(golden_dir, case)
    glob = G.global_natparam(g)
    dev = torch.device("cuda:0")
    hmm_init, hmm_pair, dense_init, dense_pair = slds_svae.global_to_local_maps(glob, dev)
    assert G.rel(_np(hmm_init), g["hmm_init"]) < 1e-12 and G.rel(_np(hmm_pair), g["hmm_pair"]) < 1e-12
    for i in range(4):
        assert G.rel(_np(dense_init[i]), g["dense_init%d" % i]) < 1e-10
        assert G.rel(_np(dense_pair[i]), g["dense_pair%d" % i]) < 1e-10
    w = _t(g["glue_states"], dev)[None]
    gi, gp = slds_svae.get_var_lds_local_natparam(dense_init, dense_pair, w)
    for i in range(4):
        assert G.rel(_np(gi[i][0]), g["glue_init%d" % i]) < 1e-10
        assert G.rel(_np(gp[i][0]), g["glue_pair%d" % i]) < 1e-10
    init_stats = (_t(g["glue_ExxT0"], dev)[None], _t(g["glue_Ex0"], dev)[None])
    pstats = tuple(_t(g["glue_pairstat%d" % i], dev)[None] for i in range(3))
    node_hmm = slds_svae.get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, pstats)
    assert G.rel(_np(node_hmm[0]), g["glue_node_hmm"]) < 1e-10
    Et = _t(np.zeros((1,) + g["hmm_pair"].shape), dev)
    _, (g_init, g_pair) = slds_svae.get_global_stats((w[:, 0], Et, w), init_stats, pstats)
    assert G.rel(_np(g_init[0]), g["glue_gstat_init_xx"]) < 1e-12
    assert G.rel(_np(g_init[1]), g["glue_gstat_init_x"]) < 1e-12
    assert G.rel(np.stack([_np(g_init[2]), _np(g_init[3])], 1), g["glue_gstat_init_1"]) < 1e-12
    for i in range(4):
        assert G.rel(_np(g_pair[i]), g["glue_gstat_pair%d" % i]) < 1e-12


@pytest.mark.parametrize("fused", [False, True])
# (slds_K8_n10_T500: BASELINE configs[3]'s shape per sequence -- K = 8, latent dim 10, T = 500 -- from the reference's OWN
#  slds_svae.py, round 6; the fixture leaves out the per-step (T-1,n,n) pair statistics)
@pytest.mark.parametrize("case", ["slds_K3_n4_T12", "slds_K8_n10_T40", "slds_K8_n10_T500"])
def test_optimize_local_meanfield_against_reference_golden(case, fused, golden_dir):
    """The whole coordinate ascent (HMM kernel <-> LDS kernel; fused = svae_slds_lds_meanfield_f64) against the
    reference's own optimize_local_meanfield run (slds_svae.py:159-175 on its compiled kernels): the same
    number of sweeps per sequence, the same statistics, the same bounds (reference_compat: the compiled filter
    drops the init potential's 4th entry, cython_lds_inference.pyx:32 -- the library's default); and the Python twin's
    convention (reference_compat=False) differs from it by exactly that term."""
    from svae_amd.models import slds_svae
    import _slds_golden as G
    g = G.load(golden_dir, case)
    glob = G.global_natparam(g)
    dev = torch.device("cuda:0")
    node = (_t(g["node_J"], dev), _t(g["node_h"], dev))
    B = node[0].shape[0]
    (hmm_stats, lds_stats), (hmm_nat, _), (hmm_vlb, lds_vlb), iters = slds_svae.optimize_local_meanfield(
        glob, node, g["opt_init_eps"], fused=fused)       # the default IS the reference as shipped (round 5)
    assert [int(i) for i in iters] == [int(i) for i in g["opt_iters"]]
    assert G.rel(_np(hmm_vlb), g["opt_hmm_vlb"]) < 1e-7 and G.rel(_np(lds_vlb), g["opt_lds_vlb"]) < 1e-7
    worst = 0.
    for got, key in ((hmm_stats[0], "E_hmm_init"), (hmm_stats[1], "E_hmm_trans"), (hmm_stats[2], "E_states"),
                     (lds_stats[0][0], "ExxT0"), (lds_stats[0][1], "Ex0"), (lds_stats[1][0], "Epair0"),
                     (lds_stats[1][1], "Epair1"), (lds_stats[1][2], "Epair2"), (lds_stats[2][0], "Enode_diagxx"),
                     (lds_stats[2][1], "Enode_x"), (hmm_nat[2], "node_hmm")):
        if "opt_" + key not in g:            # (compact fixture: no per-step pair statistics)
            continue
        e = G.rel(_np(got), g["opt_" + key])
        worst = max(worst, e)
        assert e < 1e-6, (key, e)
    print("SLDS ascent vs reference golden (%s, fused=%s): worst rel err %.2e" % (case, fused, worst))
    _, _, (_, lds_vlb2), iters2 = slds_svae.optimize_local_meanfield(glob, node, g["opt_init_eps"], fused=fused,
                                                                     reference_compat=False)
    same = (iters2 == iters).cpu().numpy()
    d = _np(lds_vlb2) - (g["opt_lds_vlb"] + g["opt_init_b"])
    assert np.all(np.abs(d[same]) < 1e-6 * np.abs(g["opt_lds_vlb"][same]))


@pytest.mark.parametrize("case", ["slds_K3_n4_T12", "slds_K8_n10_T40", "slds_K8_n10_T500"])
def test_run_inference_against_reference_golden(case, golden_dir):
    """run_inference (slds_svae.py:289-310) forward values: samples (the reference's RNG draws replayed),
    the global statistics, global_vlb (slds_prior_vlb :248-286) and local_vlb of the reference's own run."""
    from svae_amd.models import slds_svae
    import _slds_golden as G
    g = G.load(golden_dir, case)
    glob, prior = G.global_natparam(g), G.global_natparam(g, "prior_")
    dev = torch.device("cuda:0")
    node = (_t(g["node_J"][:1], dev), _t(g["node_h"][:1], dev))
    S = g["run_eps"].shape[1]
    samples, (hmm_g, (g_init, g_pair)), global_vlb, local_vlb = slds_svae.run_inference(
        prior, glob, node, S, init_eps=g["run_init_eps"][None], eps=_t(g["run_eps"][None], dev))   # default: as shipped
    assert G.rel(_np(samples[0]), g["run_samples"]) < 1e-6
    assert abs(float(global_vlb) - float(g["run_global_vlb"])) < 1e-8 * abs(float(g["run_global_vlb"]))
    assert abs(float(local_vlb) - float(g["run_local_vlb"])) < 1e-7 * abs(float(g["run_local_vlb"]))
    assert G.rel(_np(hmm_g[0]), g["run_stat_hmm_init"]) < 1e-6 and G.rel(_np(hmm_g[1]), g["run_stat_hmm_trans"]) < 1e-6
    assert G.rel(_np(g_init[0]), g["run_stat_init_xx"]) < 1e-6 and G.rel(_np(g_init[1]), g["run_stat_init_x"]) < 1e-6
    for i in range(4):
        assert G.rel(_np(g_pair[i]), g["run_stat_pair%d" % i]) < 1e-6


@pytest.mark.parametrize("K,n,T,B", [(3, 4, 9, 5), (8, 10, 40, 11), (16, 15, 6, 3), (1, 1, 2, 2), (5, 7, 2, 4)])
def test_contraction_kernels_against_the_dense_forms(K, n, T, B):
    """svae_slds_path_nodeparams_f64 / svae_slds_mix_pair_natparam_f64 (the two ends of the ascent) against the dense
    contractions they replace, evaluated on the CPU: get_arhmm_local_nodeparams on the outer products of a path
    (slds_svae.py:203-226, 131-147) and the tensordot form of get_var_lds_local_natparam (:92-103)."""
    from svae_amd.models import slds_svae as S
    g = torch.Generator().manual_seed(K * 100 + n)
    r = lambda *s: torch.randn(*s, dtype=torch.float64, generator=g)
    dense_init = (r(K, n, n), r(K, n), r(K), r(K))
    dense_pair = (r(K, n, n), r(K, n, n), r(K, n, n), r(K))
    x = r(B, T, n)
    w = torch.softmax(r(B, T, K), -1)
    dev = torch.device("cuda:0")
    to = lambda t: tuple(v.to(dev) for v in t)
    # path -> node potentials
    got = S._arhmm_nodeparams_from_path(to(dense_init), to(dense_pair), x.to(dev))
    out = lambda a, b: a.unsqueeze(-1) * b.unsqueeze(-2)
    init_stats = (out(x[:, 0], x[:, 0]), x[:, 0])
    pair_stats = (out(x[:, :-1], x[:, :-1]), out(x[:, :-1], x[:, 1:]), out(x[:, 1:], x[:, 1:]))
    want = S.get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, pair_stats)
    assert got.shape == (B, T, K)
    _close(got, _np(want), 1e-11)
    # marginals -> per-step natural parameters
    gi, gp = S.get_var_lds_local_natparam(to(dense_init), to(dense_pair), w.to(dev))
    wi, wp = S.get_var_lds_local_natparam(dense_init, dense_pair, w)
    for a, b in zip(tuple(gi) + tuple(gp), tuple(wi) + tuple(wp)):
        assert tuple(a.shape) == tuple(b.shape)
        if b.numel():
            _close(a, _np(b), 1e-12)


@pytest.mark.parametrize("B,T,n", [(5, 9, 3), (37, 70, 10), (3, 1, 4), (2, 33, 1), (130, 40, 6)])
def test_initial_sample_path_diagonal_kernel_equals_the_dense_filter_and_sampler(B, T, n):
    """initialize_local_meanfield (slds_svae.py:203-226) samples ONE path of a random-walk LDS whose matrices are all
    diagonal: svae_lds_diag_sample_f64 (n scalar recursions per sequence) must give the samples of the dense path --
    svae_lds_filter_f64 + svae_lds_sample_f64 on the same model, i.e. cython_natural_lds_sample -- for the same eps."""
    from svae_amd.lds.lds_inference import natural_lds_sample
    from svae_amd.models import slds_svae
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(B + T + n)
    node = (torch.as_tensor(-0.5 * (0.2 + 3 * rng.random((B, T, n))), device=dev),
            torch.as_tensor(2.0 * rng.standard_normal((B, T, n)), device=dev))
    eps = torch.as_tensor(rng.standard_normal((B, T, 1, n)), device=dev)
    got = slds_svae._initial_sample_path(node, eps)
    want = natural_lds_sample(slds_svae._random_walk_natparam(n, dev), node, num_samples=1, eps=eps)[:, :, 0]
    assert tuple(got.shape) == (B, T, n)
    assert float((got - want).abs().max() / want.abs().max()) < 1e-12
    assert int(slds_svae._initial_sample_path.last_info.item()) == 0


def test_check_info_reports_invalid_global_parameters_and_stays_silent_otherwise():
    """The device paths of the ascent leave their status words on the device; slds_svae.check_info() is the one host
    read that reports them (round 5: they used to sit in function attributes nothing read)."""
    from svae_amd.models import slds_svae
    K, n, T, B = 3, 4, 9, 2
    rng = np.random.default_rng(5)
    glob = _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    eps = rng.standard_normal((B, T, 1, n))
    slds_svae.optimize_local_meanfield(glob, node, eps, fused=True)
    slds_svae.check_info()                                            # valid parameters: nothing to report
    (hmm_g, lds) = glob
    niw_bad = np.array(lds[1][0], copy=True)
    niw_bad[:n, :n] = -niw_bad[:n, :n]                                # NIW scale matrix no longer positive definite
    bad = (hmm_g, [lds[0], (niw_bad, lds[1][1])] + list(lds[2:]))
    slds_svae.global_to_local_maps(bad, dev)
    with pytest.raises(FloatingPointError, match="global -> local maps"):
        slds_svae.check_info()
    slds_svae.check_info()                                            # the word is cleared once reported


@pytest.mark.parametrize("K,n,T,B", [(8, 10, 40, 11), (3, 4, 9, 5), (1, 1, 2, 1), (5, 7, 2, 4), (8, 3, 130, 700), (7, 10, 6, 3),
                                     (8, 9, 5, 1200)])
def test_final_pass_contractions_against_the_two_library_forms(K, n, T, B):
    """svae_slds_pair_contract_f64 (ONE pass over the per-step pair statistics of the final LDS E-step) against
    get_arhmm_local_nodeparams and get_global_stats (slds_svae.py:131-147, 229-243; themselves pinned on the reference's
    goldens above): HMM node potentials and the weighted sums of the pair statistics."""
    from svae_amd.models import slds_svae
    rng = np.random.default_rng(77 * K + 5 * n + T + B)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    (_, _), lds = _globals(K, n, rng)
    dense_init, dense_pair = slds_svae.get_all_lds_local_natparams([(t(a), tuple(t(y) for y in m)) for a, m in lds])
    dense_init = tuple(x.to(dev) for x in dense_init)
    dense_pair = tuple(x.to(dev) for x in dense_pair)
    E_pair = t(rng.standard_normal((B, T - 1, 3, n, n)))
    init_stats = (t(rng.standard_normal((B, n, n))), t(rng.standard_normal((B, n))))
    w = rng.random((B, T, K)) ** 2 + 1e-3
    Es = t(w / w.sum(-1, keepdims=True))
    hmm_stats = (Es[:, 0].clone(), t(rng.random((B, K, K))), Es)
    got = slds_svae.final_pass_contractions(dense_init, dense_pair, init_stats, E_pair, Es)
    assert got is not None
    node, pair_sums = got
    want_node = slds_svae.get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, E_pair)
    assert tuple(node.shape) == (B, T, K)
    _close(node, _np(want_node), 1e-11)
    want = slds_svae.get_global_stats(hmm_stats, init_stats, E_pair)
    have = slds_svae.get_global_stats(hmm_stats, init_stats, E_pair, pair_sums)
    for a, b in zip(have[1][1], want[1][1]):
        _close(a, _np(b), 1e-11)
    # shapes the kernel does not cover fall back to the library forms
    assert slds_svae.final_pass_contractions(dense_init, dense_pair, init_stats, E_pair.cpu(), Es) is None


def _consumer_sweep(K, ns=range(1, 11), Ts=(9, 12, 4)):
    """default dispatch (one-sequence ring consumer + producer wavefronts for small launches) against the table kernel (a
    different consumer: no ring, no producers); returns the worst relative distance and its case"""
    from svae_amd import _lib
    from svae_amd.models import slds_svae
    from svae_amd.lds.synthetic_data import rand_slds_global_natparam
    dev = torch.device("cuda:0")
    B = 5
    d = lambda x: tuple(d(y) for y in x) if isinstance(x, (tuple, list)) else \
        torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    worst = (0.0, None)
    for n in ns:
        for T in Ts:
            rng = np.random.default_rng(100 * K + 10 * n + T)
            glob = rand_slds_global_natparam(K, n, rng)
            _, _, dense_init, dense_pair = slds_svae.global_to_local_maps(d(glob), dev)
            dense_init = tuple(x.contiguous() for x in dense_init)
            dense_pair = tuple(x.contiguous() for x in dense_pair)
            node = (torch.as_tensor(-0.5 * (0.5 + rng.random((B, T, n))), device=dev),
                    torch.as_tensor(2. * rng.standard_normal((B, T, n)), device=dev))
            w = rng.random((B, T, K)) ** 3 + 1e-3
            w = torch.as_tensor(w / w.sum(-1, keepdims=True), device=dev)
            out = {}
            for name, opt in (("tables", _lib.OPT_LAYOUT_SPLIT), ("default", 0)):
                plan = slds_svae.SLDSMeanfieldPlan(B, T, n, K, dev, options=opt)
                plan.launch(dense_init, dense_pair, w, node, None)
                torch.cuda.synchronize()
                assert int(plan.info.item()) == 0
                out[name] = [x.clone() for x in (plan.lognorm, plan.E_init, plan.E_node_diagxx, plan.E_node_x,
                                                 plan.hmm_nodeparams(dense_init, dense_pair))]
            for a, b in zip(out["default"], out["tables"]):
                rel = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-300)
                if rel > worst[0]:
                    worst = (rel, "K=%d n=%d T=%d" % (K, n, T))
    return worst


@pytest.mark.parametrize("K", [1, 3, 8])
def test_one_sequence_consumer_sweep_every_latent_dim_against_the_table_kernel(K):
    """The default SLDS mean-field step for small launches (slds_meanfield_seq_kernel: one-sequence consumer + three
    producer wavefronts): EVERY latent dimension 1..10, odd and even T, against the table kernel.  Round 5 shipped the
    consumer with a loop form chosen because the natural one was wrong at n = 4 only, cause unknown; round 6 found the
    cause (order-dependent transposition-tile stores, csrc/lds_estep_twoend.hpp) and fixed it -- the sweep stays, and
    test_ring_consumer_two_steps_per_trip_build_at_n4 runs the loop form that exposed it."""
    rel, case = _consumer_sweep(K)
    assert rel < 1e-10, "%s: %.2e" % (case, rel)


def test_ring_consumer_two_steps_per_trip_build_at_n4():
    """Regression build for the round-5 "wrong at n = 4 only" finding (docs/experiments/
    r6_ring_two_per_trip_reproducer.md): tests/_variants/libsvae_hip_ring2.so (`make ring2`, made by __graft_entry__.build())
    is the library with the n = 4 unit's ring consumer in its two-steps-per-trip loop form.  With the transposition-tile
    stores as they were (every lane at its own column, lanes >= the row stride landing in the rows behind) hipcc
    scheduled the last rows' store first and the mean was lost from the loop's second step on: x wrong by O(1).  With the
    stores confined to their own rows the build must agree with the table kernel like the shipped one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "tests", "_variants", "libsvae_hip_ring2.so")
    if not os.path.exists(variant):
        pytest.skip("tests/_variants/libsvae_hip_ring2.so not built (python __graft_entry__.py)")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_slds_hip as m; "
            "from svae_amd import _lib; assert _lib.LIB_PATH.endswith('libsvae_hip_ring2.so'), _lib.LIB_PATH; "
            "w = max(m._consumer_sweep(K, ns=(4,), Ts=(9, 12, 13, 20, 4)) for K in (1, 3, 8)); print('WORST', w[0], w[1]); "
            "assert w[0] < 1e-10, w" % (root, os.path.join(root, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SVAE_AMD_LIB=variant), timeout=900,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_run_inference_final_pass_on_lean_records_equals_the_full_record_path():
    """Above 1024 sequences the final pass of run_inference (slds_svae.py:289-310: LDS E-step on the per-step parameters of the
    converged mean field + backward sampler) runs as ONE launch on lean per-step records (csrc/lds_lean_estep.hpp, INH);
    SVAE_OPT_LEAN_OFF sends it through the full-record E-step + sampler kernels: same samples, statistics and bounds."""
    from svae_amd import _lib
    from svae_amd.lds.lds_inference import set_default_options
    from svae_amd.models import slds_svae
    K, n, T, B = 4, 6, 12, 1100
    rng = np.random.default_rng(77)
    glob, prior = _globals(K, n, rng), _globals(K, n, rng)
    J, h = _nodes(B, T, n, rng)
    dev = torch.device("cuda:0")
    node = (torch.as_tensor(J, device=dev), torch.as_tensor(h, device=dev))
    init_eps = torch.as_tensor(rng.standard_normal((B, T, 1, n)), device=dev)
    eps = torch.as_tensor(rng.standard_normal((B, T, 1, n)), device=dev)
    outs = []
    for opt in (_lib.OPT_DEFAULT, _lib.OPT_LEAN_OFF):
        old = set_default_options(opt)
        try:
            samples, (hmm_g, (g_init, g_pair)), gv, lv = slds_svae.run_inference(prior, glob, node, 1, init_eps=init_eps, eps=eps)
        finally:
            set_default_options(old)
        outs.append([samples, hmm_g[0], hmm_g[1], g_init[0], g_init[1], g_pair[0], g_pair[1], g_pair[2], lv.reshape(1), gv.reshape(1)])
    for a, b in zip(*outs):
        assert _err(a, _np(b)) < 1e-9
