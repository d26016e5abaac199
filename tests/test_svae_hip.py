"""GPU tests of the training-step contract (svae_amd.svae.make_gradfun, the torch counterpart of
/root/reference/svae/svae.py:10-39) and of the differentiable run_inference call surface."""
import functools
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import expfam_numpy as ef, models_numpy  # noqa: E402  (checker only)
from tests.test_models_hip import _lds_globals, _np  # noqa: E402

DEV = "cuda:0"
t64 = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=DEV)


def test_gmm_differentiable_path_equals_kernel_path(golden_dir):
    from svae_amd.models import gmm
    g = np.load(os.path.join(golden_dir, "gmm_run_K5_N2_T60.npz"))
    prior, glob = (g["prior_dir"], g["prior_niw"]), (g["glob_dir"], g["glob_niw"])
    nJ, nh = t64(g["node_J"]).requires_grad_(True), t64(g["node_h"]).requires_grad_(True)
    samples, (ds, ns), global_kl, local_kl = gmm.run_inference_differentiable(
        prior, glob, (nJ, nh), 3, label_init=g["label_init"], eps=g["eps"])
    np.testing.assert_allclose(_np(samples), g["samples"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(_np(ns), g["niw_stats"], rtol=1e-8, atol=1e-10)
    assert float(local_kl) == pytest.approx(float(g["local_kl"]), rel=1e-9)
    assert float(global_kl) == pytest.approx(float(g["global_kl"]), rel=1e-10)     # the default = the reference as shipped
    (local_kl + samples.sum()).backward()
    assert torch.isfinite(nJ.grad).all() and torch.isfinite(nh.grad).all() and float(nh.grad.abs().sum()) > 0


def test_gmm_final_pass_reference_gradcheck():
    """the torch reference of the final pass itself against finite differences"""
    from _gmm_torch import final_pass_torch
    from svae_amd.distributions import expfam
    rng = np.random.default_rng(0)
    K, N, T = 3, 2, 4
    niw = np.stack([ef.niw_standard_to_natural(12. * np.eye(N), rng.standard_normal(N), np.array(10.), np.array(12.))
                    for _ in range(K)])
    lg, gg = t64(ef.dirichlet_expectedstats(np.ones(K))), t64(ef.niw_expectedstats(niw))
    r = t64(rng.random((T, K))); r = r / r.sum(-1, keepdim=True)
    nJ = t64(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N))))).requires_grad_(True)
    nh = t64(rng.standard_normal((T, N))).requires_grad_(True)
    f = lambda a, b: final_pass_torch(lg, gg, expfam.pack_dense(a, b), r)[2]
    assert torch.autograd.gradcheck(f, (nJ, nh), eps=1e-6, atol=1e-6)


def test_gmm_step_without_host_sync_captures_into_one_graph():
    """check=False leaves the fixed point's status word on the device (gmm.last_info / gmm.check_info()): the whole
    differentiable step -- global maps, fixed point + final pass, sampler, adjoint -- then records into ONE hipGraph whose
    replay reproduces the eager results bit for bit; a non-PD point still surfaces through check_info()."""
    from svae_amd.models import gmm
    K, N, T, S = 5, 2, 400, 1
    rng = np.random.default_rng(3)
    gen = torch.Generator().manual_seed(1)
    prior = tuple(x.to(DEV) for x in gmm.init_pgm_param(K, N, alpha=0.5, niw_conc=1.0, generator=gen))
    glob = tuple(x.to(DEV) for x in gmm.init_pgm_param(K, N, alpha=1.0, niw_conc=2.0, random_scale=2.0, generator=gen))
    nJ = t64(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N))))).requires_grad_(True)
    nh = t64(2 * rng.standard_normal((T, N))).requires_grad_(True)
    init = t64(rng.random((T, K))); init = init / init.sum(-1, keepdim=True)
    eps, gs = t64(rng.standard_normal((T, S, N))), t64(rng.standard_normal((T, S, N)))

    def step(check):
        samples, stats, gkl, lkl = gmm.run_inference_differentiable(prior, glob, (nJ, nh), S, label_init=init, eps=eps,
                                                                     check=check)
        return torch.autograd.grad(lkl + (samples * gs).sum(), [nJ, nh]) + (samples.detach(), lkl.detach())
    ref = [x.clone() for x in step(True)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(False)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step(False)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    gmm.check_info()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    # a point whose Gaussian factor is not positive definite: silent with check=False, raised by check_info()
    bad = nJ.detach().clone(); bad[7] = 50.0
    gmm.run_inference(prior, glob, (bad, nh.detach()), S, label_init=init, eps=eps, check=False)
    with pytest.raises(FloatingPointError):
        gmm.check_info()


@pytest.mark.parametrize("K,N,T,S", [(5, 2, 300, 1), (3, 1, 7, 2), (15, 2, 50, 3), (4, 3, 33, 2), (6, 5, 20, 1), (2, 8, 9, 2),
                                     (20, 2, 40, 0)])
def test_gmm_local_step_kernels_against_torch_autograd(K, N, T, S):
    """run_inference_differentiable (fixed point + final pass, svae_gmm_sample_f64, and in backward()
    svae_gmm_local_vjp_f64) against the same tail written in torch on the autograd tape (tests/_gmm_torch.py:
    gmm.py:74-86 + gaussian.py:27-33): samples, local KL, and the gradients of a random functional of both w.r.t. the
    node potentials, for N = 1 .. 8, any K, with and without sample cotangents."""
    from _gmm_torch import final_pass_torch, sample_torch
    from svae_amd.distributions import expfam
    from svae_amd.models import gmm
    rng = np.random.default_rng(100 * K + 10 * N + S)
    gen = torch.Generator().manual_seed(K + N)
    prior = gmm.init_pgm_param(K, N, alpha=0.5, niw_conc=1.0, generator=gen)
    glob = gmm.init_pgm_param(K, N, alpha=1.0, niw_conc=2.0, random_scale=2.0, generator=gen)
    glob = tuple(x.to(DEV) for x in glob)
    nJ = t64(-0.5 * np.log1p(np.exp(rng.standard_normal((T, N))))).requires_grad_(True)
    nh = t64(2 * rng.standard_normal((T, N))).requires_grad_(True)
    init = t64(rng.random((T, K))); init = init / init.sum(-1, keepdim=True)
    eps = t64(rng.standard_normal((T, max(S, 1), N)))
    wS, wk = t64(rng.standard_normal((T, max(S, 1), N))), 0.7
    samples, stats, gkl, lkl = gmm.run_inference_differentiable(prior, glob, (nJ, nh), max(S, 1), label_init=init, eps=eps)
    loss = wk * lkl + ((wS * samples).sum() if S else 0.0)
    gJ, gh = torch.autograd.grad(loss, [nJ, nh])
    # the same tail on the tape, from the same fixed point
    lg, gg = expfam.dirichlet_expectedstats(glob[0]), expfam.niw_expectedstats(glob[1])
    o = gmm.meanfield_from_globals(lg, gg, (nJ.detach(), nh.detach()), init)
    a, b = nJ.detach().clone().requires_grad_(True), nh.detach().clone().requires_grad_(True)
    _, (_, natp), kl_t = final_pass_torch(lg, gg, expfam.pack_dense(a, b), o["label_fixed"])
    smp_t = sample_torch(natp, eps)
    loss_t = wk * kl_t + ((wS * smp_t).sum() if S else 0.0)
    wJ, wh = torch.autograd.grad(loss_t, [a, b])
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-300))
    assert rel(samples, smp_t) < 1e-10 and abs(float(lkl) - float(kl_t)) < 1e-9 * max(1.0, abs(float(kl_t)))
    assert rel(gJ, wJ) < 1e-8, rel(gJ, wJ)
    assert rel(gh, wh) < 1e-8, rel(gh, wh)


def _lds_problem(n=3, p=4, T=6, Bn=2, seed=0):
    rng = np.random.default_rng(seed)
    prior, pgm = _lds_globals(n, rng), _lds_globals(n, rng, scale=0.8)
    data = t64(rng.standard_normal((2 * Bn, T, p)))            # 2 minibatches of Bn sequences
    recogn = (t64(0.3 * rng.standard_normal((p, n))).requires_grad_(True),
              t64(0.3 * rng.standard_normal((p, n))).requires_grad_(True))
    loglike_p = (t64(0.5 * rng.standard_normal((n, p))).requires_grad_(True),)
    eps = t64(rng.standard_normal((Bn, T, 1, n)))

    def recognize(params, batch):                              # nnet.gaussian_info-like head (nnet.py:43-47)
        WJ, Wh = params
        return -0.5 * torch.nn.functional.softplus(batch @ WJ), batch @ Wh

    def loglike(params, samples, batch):                       # samples (B,T,S,n)
        (C,) = params
        pred = samples @ C                                      # (B,T,S,p)
        return -0.5 * ((batch.unsqueeze(2) - pred) ** 2).sum() / samples.shape[2]
    return prior, pgm, data, recogn, loglike_p, eps, recognize, loglike, Bn


def test_make_gradfun_lds_against_finite_differences_and_natgrad_formula():
    from svae_amd.models.lds import run_inference_differentiable
    from svae_amd.svae import make_gradfun, flat
    prior, pgm, data, recogn, loglike_p, eps, recognize, loglike, Bn = _lds_problem()
    run = functools.partial(run_inference_differentiable, eps=eps)
    seen = []
    gradfun = make_gradfun(run, recognize, loglike, tuple(t64(x) if not isinstance(x, tuple) else tuple(t64(y) for y in x) for x in prior),
                           data, Bn, 1, natgrad_scale=10., callback=lambda i, v, p_, g: seen.append(v), permute=False)
    pgm_t = tuple(t64(x) if not isinstance(x, tuple) else tuple(t64(y) for y in x) for x in pgm)
    params = (pgm_t, loglike_p, recogn)
    pgm_natgrad, loglike_grad, recogn_grad = gradfun(params, 0)
    assert len(seen) == 1 and np.isfinite(seen[0])
    # (a) recognition / decoder gradients vs central differences of the same objective
    h = 1e-6
    for (tensor, grad) in ((recogn[0], recogn_grad[0]), (recogn[1], recogn_grad[1]), (loglike_p[0], loglike_grad[0])):
        for idx in [(0, 0), (1, 2), (3, 1)]:
            if idx[0] >= tensor.shape[0] or idx[1] >= tensor.shape[1]:
                continue
            with torch.no_grad():
                old = float(tensor[idx]); tensor[idx] = old + h
                fp = float(-gradfun.mc_elbo(pgm_t, loglike_p, recogn, 0)); tensor[idx] = old - h
                fm = float(-gradfun.mc_elbo(pgm_t, loglike_p, recogn, 0)); tensor[idx] = old
            num = (fp - fm) / (2 * h)
            assert abs(float(grad[idx]) - num) < 1e-5 * max(1.0, abs(num)), (idx, float(grad[idx]), num)
    # (b) natural gradient = -scale/N (prior + num_batches * stats - params), svae.py:33-34, with the
    #     statistics of minibatch 0 from the oracle
    nn = [_np(x) for x in recognize(recogn, data[:Bn])]
    want = [models_numpy.lds_run_inference(prior, pgm, (nn[0][b], nn[1][b]), _np(eps)[b]) for b in range(Bn)]
    E_init = sum(ef.pack_dense(w[1][0][0], w[1][0][1], np.array(1.), np.array(1.)) for w in want)
    E_pair = [sum(np.asarray(w[1][1][i]) for w in want) for i in range(3)] + [Bn * (data.shape[1] - 1.)]
    flatnp = lambda s: np.concatenate([np.ravel(np.asarray(x, float)) for x in ([s[0]] + list(s[1]))])
    stats = flatnp((E_init, E_pair))
    # (number of data points = time steps, get_num_datapoints of the reference, svae.py:13)
    expect = -10. / (data.shape[0] * data.shape[1]) * (flatnp(prior) + 2 * stats - flatnp(pgm))
    got = _np(flat(pgm_natgrad))
    assert np.max(np.abs(got - expect)) < 1e-6 * np.max(np.abs(expect))


def test_example_training_loop_runs_and_improves():
    """examples/lds_svae_synth.py: a few natural-gradient / SGD iterations through the HIP E-step,
    sampler and VJP kernels; the Monte-Carlo ELBO estimate must stay finite and go up on average."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "lds_svae_synth.py")
    spec = importlib.util.spec_from_file_location("lds_svae_synth", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    vals = mod.main(["--iters", "24", "--seqs", "64", "--T", "40", "--n", "4", "--p", "8", "--batch", "32", "--quiet"])
    assert len(vals) == 24 and np.all(np.isfinite(vals))
    assert np.mean(vals[-6:]) > np.mean(vals[:6])


def test_gmm_example_training_loop_runs_and_improves():
    """examples/gmm_svae_synth.py (the loop of the reference's experiments/gmm_svae_synth.py: pinwheel data, K = 15,
    minibatches of 50): natural-gradient / SGD iterations through the GMM kernels -- global step, persistent fixed
    point, sampler, derived adjoint; the Monte-Carlo ELBO estimate stays finite and goes up on average."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "gmm_svae_synth.py")
    spec = importlib.util.spec_from_file_location("gmm_svae_synth", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    vals, used = mod.main(["--iters", "60", "--quiet"])
    assert len(vals) == 60 and np.all(np.isfinite(vals))
    assert np.mean(vals[-10:]) > np.mean(vals[:10]), (vals[:10], vals[-10:])
    assert 1 <= used <= 15
