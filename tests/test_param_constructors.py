"""CPU tests of the parameter constructors that sit on the data-format side of the hot path
(models/lds.py:57-67, models/gmm.py:33-52, models/slds_svae.py:27-75 of the reference), against the
NumPy restatement of the exponential-family maps."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import expfam_numpy as ef


def test_lds_make_prior_natparam_matches_the_reference_values():
    from svae_amd.models.lds import make_prior_natparam, lds_prior_expectedstats
    n = 4
    niw, mniw = make_prior_natparam(n)
    nu, S, mu, kappa = n + 1., 2. * (n + 1) * np.eye(n), np.zeros(n), 1. / (2. * n)
    np.testing.assert_allclose(niw.numpy(), ef.niw_standard_to_natural(S, mu, np.array(kappa), np.array(nu)), rtol=1e-14)
    want = ef.mniw_standard_to_natural(nu, S, np.eye(n), kappa * np.eye(n))
    for a, b in zip(mniw, want):
        np.testing.assert_allclose(np.asarray(a), b, rtol=1e-14)
    # usable as a global parameter: its expected statistics are a valid LDS local natparam
    init, pair = lds_prior_expectedstats((niw, mniw))
    assert np.all(np.linalg.eigvalsh(-2 * np.asarray(pair[2])) > 0)


def test_gmm_init_pgm_param_and_prior_terms():
    from svae_amd.models import gmm
    K, N = 3, 2
    d, niws = gmm.init_pgm_param(K, N, alpha=0.5)
    assert tuple(d.shape) == (K,) and tuple(niws.shape) == (K, N + 2, N + 2) and float(d[0]) == 0.5
    want = ef.niw_standard_to_natural((N + 10.) * np.eye(N), np.zeros(N), np.array(10.), np.array(N + 10.))
    np.testing.assert_allclose(niws[1].numpy(), want, rtol=1e-14)
    es = gmm.prior_expectedstats((d, niws))
    np.testing.assert_allclose(es[0].numpy(), ef.dirichlet_expectedstats(d.numpy()), rtol=1e-12)
    np.testing.assert_allclose(es[1].numpy(), ef.niw_expectedstats(niws.numpy()), rtol=1e-10)
    lz = ef.dirichlet_logZ(d.numpy()) + np.sum(ef.niw_logZ(niws.numpy()))
    assert float(gmm.prior_logZ((d, niws))) == pytest.approx(float(lz), rel=1e-12)
    g = torch.Generator().manual_seed(0)
    d2, niws2 = gmm.init_pgm_param(K, N, alpha=0.5, random_scale=1.0, generator=g)
    assert not torch.equal(niws2[0], niws2[1]) and float(d2.max()) <= 0.5


def test_slds_global_natparam_constructors():
    from svae_amd.models import slds_svae
    K, n = 4, 3
    (d, md), lds = slds_svae.make_slds_global_natparam(K, n, alpha=5., sticky_bias=2.)
    assert tuple(d.shape) == (K,) and tuple(md.shape) == (K, K) and float(md[1, 1]) == 7. and float(md[0, 1]) == 5.
    assert len(lds) == K
    want_niw = ef.niw_standard_to_natural((n + 10.) * np.eye(n), np.zeros(n), np.array(10.), np.array(n + 10.))
    np.testing.assert_allclose(lds[2][0].numpy(), want_niw, rtol=1e-14)
    want = ef.mniw_standard_to_natural(n + 10., (n + 10.) * np.eye(n), np.zeros((n, n)), 10. * np.eye(n))
    for a, b in zip(lds[0][1], want):
        np.testing.assert_allclose(np.asarray(a), b, rtol=1e-14)
    g = torch.Generator().manual_seed(1)
    (_, _), lds_r = slds_svae.make_slds_global_natparam(K, n, random=True, generator=g)
    init, pair = slds_svae.get_all_lds_local_natparams(lds_r)
    assert tuple(pair[0].shape) == (K, n, n) and torch.isfinite(pair[0]).all()


def test_nnet_linear_matches_plain_matmul_and_its_gradients():
    """svae_amd.nnet.linear = x @ w; its blocked weight gradient equals autograd's (CPU, float64)."""
    import torch
    from svae_amd.nnet import gaussian_info, gaussian_info_two_heads, init_mlp, linear, tanh_mlp
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(3, 301, 5, dtype=torch.float64, generator=gen, requires_grad=True)     # 903 rows: 3 blocks + a tail
    w = torch.randn(5, 4, dtype=torch.float64, generator=gen, requires_grad=True)
    g = torch.randn(3, 301, 4, dtype=torch.float64, generator=gen)
    gx, gw = torch.autograd.grad((linear(x, w) * g).sum(), [x, w])
    hx, hw = torch.autograd.grad(((x @ w) * g).sum(), [x, w])
    assert torch.allclose(gx, hx, rtol=1e-13, atol=1e-13) and torch.allclose(gw, hw, rtol=1e-12, atol=1e-12)
    J, h = gaussian_info_two_heads(([w], [w]), x)
    assert bool((J <= 0).all()) and J.shape == h.shape == (3, 301, 4)
    # the reference's form (nnet.py:23, 43-47): layers nonlin(x W + b), ONE network split into (J_input, h)
    layers = init_mlp([5, 7, 8], generator=gen)
    J, h = gaussian_info(layers, x)
    (W1, b1), (W2, b2) = layers
    out = torch.tanh(x @ W1 + b1) @ W2 + b2
    assert torch.allclose(J, -0.5 * torch.log1p(torch.exp(out[..., :4])), rtol=1e-13, atol=1e-14)
    assert torch.allclose(h, out[..., 4:], rtol=1e-13, atol=1e-14) and torch.allclose(tanh_mlp(layers, x), out)
    assert all(g is not None for g in torch.autograd.grad((J.sum() + (h * h).sum()), [W1, b1, W2, b2]))


def test_accurate_smoother_switch_sets_the_full_record_option():
    """set_accurate_smoother: the documented way to the cond * eps kernels (full records for the E-step, the chol(P)^-T
    records for inference + VJP at every batch size); host-side word only."""
    from svae_amd import _lib
    from svae_amd.lds import lds_inference as li
    old = li.set_accurate_smoother(True)
    try:
        assert li._default_options & _lib.OPT_TWOEND_FULL and li._default_options & _lib.OPT_LEAN_ON
        assert li.set_accurate_smoother(False) & _lib.OPT_TWOEND_FULL
        assert not (li._default_options & (_lib.OPT_TWOEND_FULL | _lib.OPT_LEAN_ON))
    finally:
        li.set_default_options(old)


def test_host_condition_guard_picks_the_accurate_kernels_for_ill_conditioned_host_parameters():
    """_host_condition_options: homogeneous pair blocks that arrive as host data (what a caller of the reference passes) and
    are ill-conditioned -> the option bits of the cond * eps kernels for a plan made on the spot; well-conditioned blocks,
    per-step parameters and anything that is not plain host data -> 0 (device tensors are never inspected)."""
    import numpy as np
    import torch
    from svae_amd import _lib
    from svae_amd.lds import lds_inference as li
    from svae_amd.lds.synthetic_data import rand_lds_natparam
    acc = _lib.OPT_TWOEND_FULL | _lib.OPT_LEAN_ON
    good = rand_lds_natparam(7, np.random.default_rng(0))[1]       # cond(J22) 1e3
    bad = rand_lds_natparam(7, np.random.default_rng(262))[1]      # cond(J22) 7.8e7
    assert li._host_condition_options(good) == 0
    assert li._host_condition_options(bad) == acc
    assert li._host_condition_options(tuple(torch.as_tensor(np.asarray(x)) for x in bad)) == acc
    assert li._host_condition_options(tuple(np.stack([np.asarray(x)] * 3) for x in bad[:3]) + (np.zeros(3),)) == 0
    old, li.CONDITION_GUARD_THRESHOLD = li.CONDITION_GUARD_THRESHOLD, None
    try:
        assert li._host_condition_options(bad) == 0
    finally:
        li.CONDITION_GUARD_THRESHOLD = old
