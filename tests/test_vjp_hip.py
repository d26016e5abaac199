"""GPU parity tests of the reverse-mode sweeps (svae_lds_estep_vjp_f64) against the reference's own
compiled VJPs (natural_filter_grad / natural_smoother_general_grad / natural_sample_backward_grad,
composed as lds_inference.py:26-39 wires them; oracle/ref.estep_vjp) and against central finite
differences of the HIP forward itself."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ref  # noqa: E402  (checker only)


def _rel(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, float)
    b = np.asarray(b, float)
    scale = np.maximum(np.abs(b), 1e-3 * max(np.max(np.abs(b)), 1e-300))
    return float(np.max(np.abs(a - b) / scale))


def _rel_rows(a, b):
    """_rel per leading index (one value per sequence)"""
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, float)
    b = np.asarray(b, float)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    scale = np.maximum(np.abs(b), 1e-3 * np.maximum(np.max(np.abs(b), axis=1, keepdims=True), 1e-300))
    return np.max(np.abs(a - b) / scale, axis=1)


def _setup(n, T, B, S, seed):
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((B, T, n), rng, with_logZ=True)
    g = dict(ln=rng.standard_normal(B), dxx=rng.standard_normal((B, T, n)), x=rng.standard_normal((B, T, n)),
             s=rng.standard_normal((B, T, S, n)))
    return init, pair, node, g


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
# (small batches run the sweeps with producer / helper wavefronts for S <= 4, sweep 2 for n <= 12: the cases cover
#  T = 1, 2, 3 -- every wavefront of a workgroup must execute the same T+1 barriers --, S > 4 and n > 12)
@pytest.mark.parametrize("n,T,B,S", [(3, 6, 2, 2), (10, 25, 5, 1), (10, 40, 3, 4), (15, 5, 1, 3), (1, 4, 2, 1),
                                     (5, 1, 3, 1), (7, 2, 5, 2), (10, 3, 2, 4), (6, 9, 3, 6), (13, 7, 2, 2),
                                     (12, 8, 6, 3)])
@pytest.mark.parametrize("with_samples", [False, True])
def test_vjp_against_reference_compiled_vjps(n, T, B, S, with_samples):
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    init, pair, node, g = _setup(n, T, B, S, 31 * n + T)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    want, eps = [], np.zeros((B, T, S, n))
    for b in range(B):
        (gJ, gh, gz), e = ref.estep_vjp((init, pair), tuple(x[b] for x in node), g["ln"][b],
                                        (g["dxx"][b], g["x"][b]), g["s"][b] if with_samples else None,
                                        seed=100 + b)
        want.append((gJ, gh, gz))
        if with_samples:
            eps[b] = e
    nJ, nh, nz = (t(x).requires_grad_(True) for x in node)
    lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(
        (tuple(t(x) for x in init), tuple(t(x) for x in pair)), (nJ, nh, nz),
        eps=t(eps) if with_samples else None)
    loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum()
    if with_samples:
        loss = loss + (t(g["s"]) * samples).sum()
    loss.backward()
    for b in range(B):
        assert _rel(nJ.grad[b], want[b][0]) < 1e-6, "g_node_J"
        assert _rel(nh.grad[b], want[b][1]) < 1e-6, "g_node_h"
        assert _rel(nz.grad[b], want[b][2]) < 1e-12, "g_node_logZ"


def test_vjp_against_finite_differences_of_the_hip_forward():
    """Independent of any reference: d/d(node) of a random linear functional of the HIP outputs."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    n, T, B, S = 4, 6, 2, 2
    init, pair, node, g = _setup(n, T, B, S, 5)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    eps = t(np.random.default_rng(9).standard_normal((B, T, S, n)))

    def f(nJ, nh):
        lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(natparam, (nJ, nh), eps=eps)
        return (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum() \
            + (t(g["s"]) * samples).sum()

    nJ, nh = t(node[0]).requires_grad_(True), t(node[1]).requires_grad_(True)
    f(nJ, nh).backward()
    h = 1e-6
    rng = np.random.default_rng(1)
    for _ in range(12):
        b, tt, i = rng.integers(B), rng.integers(T), rng.integers(n)
        for which, grad in ((0, nJ.grad), (1, nh.grad)):
            d = torch.zeros(B, T, n, dtype=torch.float64, device=dev)
            d[b, tt, i] = h
            args = [t(node[0]), t(node[1])]
            args[which] = args[which] + d
            fp = float(f(*args))
            args[which] = args[which] - 2 * d
            fm = float(f(*args))
            num = (fp - fm) / (2 * h)
            assert abs(float(grad[b, tt, i]) - num) < 2e-5 * max(1.0, abs(num))


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("n,T,B,S", [(3, 6, 2, 2), (10, 20, 3, 1), (8, 5, 2, 3), (4, 2, 2, 1)])
@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("with_samples", [False, True])
def test_vjp_inhomogeneous_with_statistics_cotangents(n, T, B, S, batched, with_samples):
    """Per-step (and per-sequence) pair parameters plus cotangents of E_init and the per-step E_pair --
    what the SLDS-SVAE differentiates (slds_svae.py:295-300) -- against the reference's compiled
    VJPs (its _compute_stats_grad, cython_lds_inference.pyx:212-234, takes those cotangents)."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(7 * n + T + 100 * batched)
    init = rand_lds_natparam(n, rng)[0]
    sets = B if batched else 1
    pairs = [[rand_lds_natparam(n, rng)[1] for _ in range(T - 1)] for _ in range(sets)]
    stack = lambda i: np.stack([np.stack([p[i] for p in row]) for row in pairs])
    pair_b = tuple(stack(i) for i in range(4))                      # (sets,T-1,n,n) x3, (sets,T-1)
    pair = pair_b if batched else tuple(x[0] for x in pair_b)
    node = rand_node_potentials((B, T, n), rng, with_logZ=True)
    g = dict(ln=rng.standard_normal(B), dxx=rng.standard_normal((B, T, n)), x=rng.standard_normal((B, T, n)),
             s=rng.standard_normal((B, T, S, n)), i=rng.standard_normal((B, n * n + n)),
             p=rng.standard_normal((B, T - 1, 3, n, n)))
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    want, eps = [], np.zeros((B, T, S, n))
    for b in range(B):
        pb = tuple(x[b if batched else 0] for x in pair_b)
        (gJ, gh, gz), e = ref.estep_vjp(
            (init, pb), tuple(x[b] for x in node), g["ln"][b], (g["dxx"][b], g["x"][b]),
            g["s"][b] if with_samples else None, seed=200 + b,
            g_E_init=(g["i"][b, :n * n].reshape(n, n), g["i"][b, n * n:]),
            g_E_pair=(g["p"][b, :, 0], g["p"][b, :, 1], g["p"][b, :, 2]))
        want.append((gJ, gh, gz))
        if with_samples:
            eps[b] = e
    nJ, nh, nz = (t(x).requires_grad_(True) for x in node)
    lognorm, (dxx, ex), samples, (E_init, E_pair) = lds_inference_differentiable(
        (tuple(t(x) for x in init), tuple(t(x) for x in pair)), (nJ, nh, nz),
        eps=t(eps) if with_samples else None)
    loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum() \
        + (t(g["i"]) * E_init).sum() + (t(g["p"]) * E_pair).sum()
    if with_samples:
        loss = loss + (t(g["s"]) * samples).sum()
    loss.backward()
    for b in range(B):
        assert _rel(nJ.grad[b], want[b][0]) < 1e-6, "g_node_J"
        assert _rel(nh.grad[b], want[b][1]) < 1e-6, "g_node_h"
        assert _rel(nz.grad[b], want[b][2]) < 1e-12, "g_node_logZ"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("n,T,B,S", [(3, 6, 2, 2), (10, 30, 3, 1), (6, 2, 2, 1)])
@pytest.mark.parametrize("with_samples", [False, True])
def test_vjp_homogeneous_summed_pair_statistics_cotangents(n, T, B, S, with_samples):
    """Homogeneous pair parameters, cotangents of E_init and of the SUMMED E_pair (B,3,n,n): the homogeneous
    branch of the reference's _compute_stats_grad (cython_lds_inference.pyx:229-231), through
    lds_inference_differentiable(..., pair_stats_grad=True)."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(31 * n + T)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((B, T, n), rng, with_logZ=True)
    g = dict(ln=rng.standard_normal(B), dxx=rng.standard_normal((B, T, n)), x=rng.standard_normal((B, T, n)),
             s=rng.standard_normal((B, T, S, n)), i=rng.standard_normal((B, n * n + n)),
             p=rng.standard_normal((B, 3, n, n)))
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    want, eps, fwd = [], np.zeros((B, T, S, n)), []
    for b in range(B):
        (gJ, gh, gz), e = ref.estep_vjp(
            (init, pair), tuple(x[b] for x in node), g["ln"][b], (g["dxx"][b], g["x"][b]),
            g["s"][b] if with_samples else None, seed=300 + b,
            g_E_init=(g["i"][b, :n * n].reshape(n, n), g["i"][b, n * n:]),
            g_E_pair=(g["p"][b, 0], g["p"][b, 1], g["p"][b, 2]))
        want.append((gJ, gh, gz))
        fwd.append(ref.estep((init, pair), tuple(x[b] for x in node)))
        if with_samples:
            eps[b] = e
    nJ, nh, nz = (t(x).requires_grad_(True) for x in node)
    lognorm, (dxx, ex), samples, (E_init, E_pair) = lds_inference_differentiable(
        (tuple(t(x) for x in init), tuple(t(x) for x in pair)), (nJ, nh, nz),
        eps=t(eps) if with_samples else None, pair_stats_grad=True)
    assert tuple(E_pair.shape) == (B, 3, n, n)
    for b in range(B):          # the summed statistics themselves
        _, (Ei, Ep, En) = fwd[b]
        for i in range(3):
            assert _rel(E_pair[b, i], np.asarray(Ep[i])) < 1e-8
    loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum() \
        + (t(g["i"]) * E_init).sum() + (t(g["p"]) * E_pair).sum()
    if with_samples:
        loss = loss + (t(g["s"]) * samples).sum()
    loss.backward()
    for b in range(B):
        assert _rel(nJ.grad[b], want[b][0]) < 1e-6, "g_node_J"
        assert _rel(nh.grad[b], want[b][1]) < 1e-6, "g_node_h"
        assert _rel(nz.grad[b], want[b][2]) < 1e-12, "g_node_logZ"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("B", [512, 700, 1500, 2304, 3200, 4096])
def test_vjp_full_size_against_reference(B):
    """BASELINE configs[1] shape (T=200, n=10): E-step + sampler + VJP of the whole batch, EVERY sequence checked
    against the reference's compiled VJPs (round 5: oracle/ref_batch.py spreads them over the host cores; rounds
    1 - 4 spot-checked 16).  B = 512 runs the role-split sweeps (two
    workgroups per four sequences, B <= 2048) with one sequence per consumer, 700 with producer wavefronts (<= 1024),
    1500 without, B = 2304 the fused sweeps, 3200 also the forward pass without the one-directional filter (> 3072);
    4096 = north_star's batch on one GPU, the shape `bench.py` times as extra[6]."""
    from oracle import ref_batch
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    n, T, S = 10, 200, 1
    init, pair, node, g = _setup(n, T, B, S, 4242)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    want = ref_batch.estep_vjp_all((init, pair), node, g["ln"], g["dxx"], g["x"], g["s"], seeds=1000 + np.arange(B))
    assert np.array_equal(want["index"], np.arange(B))
    eps = want["eps"]                       # the noise the reference's sampler drew, per sequence
    nJ, nh, nz = (t(x).requires_grad_(True) for x in node)
    lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(
        (tuple(t(x) for x in init), tuple(t(x) for x in pair)), (nJ, nh, nz), eps=t(eps))
    loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum() \
        + (t(g["s"]) * samples).sum()
    loss.backward()
    eJ, eh = _rel_rows(nJ.grad, want["gJ"]), _rel_rows(nh.grad, want["gh"])
    worst = float(max(eJ.max(), eh.max()))

    def plain(a, b, guard=1e-9):     # element-wise |a-b|/|b| without an absolute floor (entries above 1e-9 of the maximum:
        a, b = a.detach().cpu().numpy(), np.asarray(b, float)       # a gradient entry is a sum of T x n^2 terms)
        m = np.abs(b) > guard * np.max(np.abs(b))
        return float(np.max(np.abs(a - b)[m] / np.abs(b)[m]))
    pl = max(plain(nJ.grad, want["gJ"]), plain(nh.grad, want["gh"]))
    print("B=%d: all %d sequences vs the compiled reference's VJPs, worst rel err %.2e (sequence %d); plain "
          "element-wise |a-b|/|b| (no floor, entries > 1e-9 max): %.2e" % (B, B, worst, int(np.maximum(eJ, eh).argmax()), pl))
    assert pl < 1e-3, pl             # (reported; cancellation in near-zero gradient entries: the bound is loose)
    assert float(_rel_rows(nz.grad, want["gz"]).max()) < 1e-12
    assert worst < 1e-6, worst


def test_forward_outputs_do_not_depend_on_the_hand_off_kept():
    """keep = 0 (the E-step proper; the two-ended kernel for n <= 10) and keep = 3 (factor + cross
    moments kept for the sampler / VJP: one-directional kernels) are different schedules of the same
    arithmetic: their outputs agree to rounding at the headline shape."""
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    B, T, n = 512, 200, 10
    rng = np.random.default_rng(77)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((B, T, n), rng)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]),
            t(pair[3]).reshape(1), t(node[0]), t(node[1]), None]
    plan = LDSEStepPlan(B, T, n, dev)
    plan.launch(*args)
    a = [x.clone() for x in (plan.lognorm, plan.E_init, plan.E_pair, plan.E_node_diagxx, plan.E_node_x)]
    plan.launch(*args, False, True, True)
    b = [plan.lognorm, plan.E_init, plan.E_pair, plan.E_node_diagxx, plan.E_node_x]
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) <= 1e-12 * float(y.abs().max()) + 1e-300, float((x - y).abs().max())


def test_plan_reuse_before_backward_is_refused():
    """The VJP reads the workspace of the plan's last launch: a second forward on the same plan before
    backward must raise instead of returning gradients of the wrong pass."""
    from svae_amd.lds.lds_inference import LDSEStepPlan, lds_inference_differentiable
    init, pair, node, g = _setup(4, 6, 2, 1, 3)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    plan = LDSEStepPlan(2, 6, 4, dev)
    nJ, nh = t(node[0]).requires_grad_(True), t(node[1]).requires_grad_(True)
    lognorm, _, _, _ = lds_inference_differentiable(natparam, (nJ, nh), plan=plan)
    lds_inference_differentiable(natparam, (t(node[0]), t(node[1])), plan=plan)      # e.g. an evaluation pass
    with pytest.raises(RuntimeError, match="launched again"):
        lognorm.sum().backward()
    lognorm2, _, _, _ = lds_inference_differentiable(natparam, (nJ, nh), plan=plan)
    lognorm2.sum().backward()
    assert torch.isfinite(nJ.grad).all()


@pytest.mark.parametrize("n,T,B,S", [(10, 37, 7, 1), (6, 12, 3, 3), (13, 9, 5, 2)])
def test_producer_wavefront_kernels_agree_with_the_packed_ones(n, T, B, S):
    """SVAE_OPT_PRODUCERS_OFF sends the sampler and the sweeps through the packed kernels (one wavefront per
    four sequences, register prefetch); the default runs them with producer / helper wavefronts and, up to 512
    sequences, with one sequence per wavefront.  Different schedules of the same arithmetic: gradients and samples
    agree to rounding."""
    from svae_amd import _lib
    from svae_amd.lds.lds_inference import lds_inference_differentiable, set_default_options
    init, pair, node, g = _setup(n, T, B, S, 11 * n + T)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    eps = t(np.random.default_rng(3).standard_normal((B, T, S, n)))

    def run(with_samples):
        nJ, nh = t(node[0]).requires_grad_(True), t(node[1]).requires_grad_(True)
        lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(natparam, (nJ, nh),
                                                                    eps=eps if with_samples else None)
        loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum()
        if with_samples:
            loss = loss + (t(g["s"]) * samples).sum()
        loss.backward()
        return [nJ.grad.clone(), nh.grad.clone()] + ([samples.detach().clone()] if with_samples else [])

    for with_samples in (False, True):
        old = set_default_options(_lib.OPT_PRODUCERS_OFF)
        try:
            packed = run(with_samples)
        finally:
            assert set_default_options(old) == _lib.OPT_PRODUCERS_OFF
        default = run(with_samples)
        for x, y in zip(default, packed):
            assert float((x - y).abs().max()) <= 1e-11 * float(y.abs().max()) + 1e-300, float((x - y).abs().max())
