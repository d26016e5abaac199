"""GPU parity tests of the LEAN-record training path (round 6): svae_lds_inference_f64 -- E-step + backward sampler in
one launch, per-step records [chol(P)^-T | c] -- and the two VJP sweeps that read those records
(csrc/lds_lean_estep.hpp, lds_lean_vjp.hpp), against the reference's own compiled code (oracle/_ref:
cython_lds_inference.pyx forward, sampler and *_grad functions) and against the full-record kernels of this library.
The default dispatch takes the lean path above 2048 sequences (tests/test_vjp_hip.py checks EVERY sequence of B = 2304,
3200 and 4096 against the reference); here SVAE_OPT_LEAN_ON forces it at sizes the reference finishes in seconds."""
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


def _rel_plain(a, b, guard=1e-12):
    """element-wise |a - b| / |b| over the entries with |b| > guard * max|b| (no absolute floor)"""
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, float)
    b = np.asarray(b, float)
    m = np.abs(b) > guard * max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


def _setup(n, T, B, S, seed):
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(seed)
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((B, T, n), rng, with_logZ=True)
    g = dict(ln=rng.standard_normal(B), dxx=rng.standard_normal((B, T, n)), x=rng.standard_normal((B, T, n)),
             s=rng.standard_normal((B, T, S, n)))
    return init, pair, node, g


def _plan(B, T, n, lean):
    from svae_amd import _lib
    from svae_amd.lds.lds_inference import LDSEStepPlan
    return LDSEStepPlan(B, T, n, "cuda:0", options=_lib.OPT_LEAN_ON if lean else _lib.OPT_LEAN_OFF)


CASES = [(10, 25, 5, 1), (10, 40, 3, 2), (3, 6, 2, 2), (1, 4, 2, 1), (7, 2, 5, 2), (10, 3, 9, 3), (5, 17, 13, 1),
         (9, 8, 4, 2), (2, 5, 1, 1), (8, 31, 6, 2), (4, 11, 7, 1), (6, 200, 4, 1)]


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("n,T,B,S", CASES)
@pytest.mark.parametrize("with_samples", [False, True])
def test_lean_path_against_reference_compiled_code(n, T, B, S, with_samples):
    """Forward outputs (lognorm, node statistics, global statistics, samples under the reference's own noise) and the
    gradients w.r.t. the node potentials, lean records forced, against the compiled reference, every sequence."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    init, pair, node, g = _setup(n, T, B, S, 17 * n + T)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    want, eps, want_s = [], np.zeros((B, T, S, n)), np.zeros((B, T, S, n))
    for b in range(B):
        nb = tuple(x[b] for x in node)
        (gJ, gh, gz), e = ref.estep_vjp((init, pair), nb, g["ln"][b], (g["dxx"][b], g["x"][b]),
                                        g["s"][b] if with_samples else None, seed=100 + b)
        want.append((gJ, gh, gz, ref.estep((init, pair), nb)))
        if with_samples:
            eps[b] = e
            smp, e2 = ref.sample_backward((init, pair), nb, S, seed=100 + b)
            assert np.array_equal(e, e2)
            want_s[b] = smp
    plan = _plan(B, T, n, lean=True)
    lean = (not with_samples) or S <= 2            # at most two samples are drawn inside the smoother's loop
    assert plan.lib.svae_lds_inference_is_lean(B, T, n, S if with_samples else 0, 0, 1, plan.options) == int(lean)
    nJ, nh, nz = (t(x).requires_grad_(True) for x in node)
    lognorm, (dxx, ex), samples, (E_init, E_pair) = lds_inference_differentiable(
        (tuple(t(x) for x in init), tuple(t(x) for x in pair)), (nJ, nh, nz),
        eps=t(eps) if with_samples else None, plan=plan)
    assert plan.lean == lean
    loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum()
    if with_samples:
        loss = loss + (t(g["s"]) * samples).sum()
    loss.backward()
    plan.check_info()
    for b in range(B):
        ln, (oi, op, on) = want[b][3]
        assert _rel(lognorm[b], ln) < 1e-8
        assert _rel(E_init[b, :n * n].reshape(n, n), oi[0]) < 1e-8 and _rel(E_init[b, n * n:], oi[1]) < 1e-8
        for i in range(3):
            assert _rel(E_pair[b, i], np.asarray(op[i])) < 1e-8, "E_pair[%d]" % i
        assert _rel(dxx[b], on[0]) < 1e-8 and _rel(ex[b], on[1]) < 1e-8
        if with_samples:
            assert _rel(samples[b], want_s[b]) < 1e-8, "samples"
        assert _rel(nJ.grad[b], want[b][0]) < 1e-6, "g_node_J"
        assert _rel(nh.grad[b], want[b][1]) < 1e-6, "g_node_h"
        assert _rel(nz.grad[b], want[b][2]) < 1e-12, "g_node_logZ"


@pytest.mark.parametrize("n,T,B,S", [(10, 60, 37, 1), (10, 33, 9, 2), (6, 20, 21, 2), (4, 2, 3, 1), (10, 200, 16, 1)])
@pytest.mark.parametrize("with_samples", [False, True])
def test_lean_and_full_records_agree(n, T, B, S, with_samples):
    """The lean path is a different storage format of the same recursions: outputs, samples and gradients agree with
    the full-record kernels far inside the parity tolerance (floor-free relative error reported as well)."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    init, pair, node, g = _setup(n, T, B, S, 5 * n + T)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    natparam = (tuple(t(x) for x in init), tuple(t(x) for x in pair))
    eps = t(np.random.default_rng(3).standard_normal((B, T, S, n)))

    def run(lean):
        plan = _plan(B, T, n, lean)
        nJ, nh = t(node[0]).requires_grad_(True), t(node[1]).requires_grad_(True)
        lognorm, (dxx, ex), samples, (Ei, Ep) = lds_inference_differentiable(
            natparam, (nJ, nh), eps=eps if with_samples else None, plan=plan)
        assert plan.lean == (lean and (not with_samples or S <= 2))
        loss = (t(g["ln"]) * lognorm).sum() + (t(g["dxx"]) * dxx).sum() + (t(g["x"]) * ex).sum()
        if with_samples:
            loss = loss + (t(g["s"]) * samples).sum()
        loss.backward()
        out = [lognorm, dxx, ex, Ei, Ep, nJ.grad, nh.grad] + ([samples] if with_samples else [])
        return [x.detach().clone() for x in out]

    names = ["lognorm", "diagxx", "x", "E_init", "E_pair", "g_node_J", "g_node_h", "samples"]
    for name, a, b in zip(names, run(True), run(False)):
        assert _rel(a, b.cpu().numpy()) < 1e-9, name
        assert _rel_plain(a, b.cpu().numpy(), guard=1e-9) < 1e-6, name


def test_lean_plan_refuses_a_separate_sampler_call_and_statistics_cotangents():
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    n, T, B, S = 4, 6, 3, 1
    init, pair, node, g = _setup(n, T, B, S, 3)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    plan = _plan(B, T, n, lean=True)
    eps = t(np.random.default_rng(0).standard_normal((B, T, S, n)))
    lds_inference_differentiable((tuple(t(x) for x in init), tuple(t(x) for x in pair)),
                                 (t(node[0]), t(node[1])), eps=eps, plan=plan)
    assert plan.lean
    with pytest.raises(RuntimeError, match="lean"):
        plan.sample(eps)
    with pytest.raises(ValueError, match="lean"):
        plan.vjp(t(g["ln"]), g_E_init=torch.zeros(B, n * n + n, dtype=torch.float64, device=dev))


def test_inference_entry_point_equals_estep_plus_sampler_where_lean_records_do_not_apply():
    """n > 10 / per-step pair parameters / S > 4: svae_lds_inference_f64 is svae_lds_estep_f64 (keep = 3) followed by
    svae_lds_sample_f64, bit for bit."""
    from svae_amd import _lib
    from svae_amd.lds.lds_inference import LDSEStepPlan
    for n, T, B, S in ((12, 7, 5, 2), (6, 9, 3, 6)):
        init, pair, node, _ = _setup(n, T, B, S, 41)
        dev = torch.device("cuda:0")
        t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
        args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]),
                t(pair[3]).reshape(1), t(node[0]), t(node[1]), t(node[2])]
        eps = t(np.random.default_rng(1).standard_normal((B, T, S, n)))
        p1 = LDSEStepPlan(B, T, n, dev, options=_lib.OPT_LEAN_ON)
        s1 = p1.infer(*args, False, eps)
        assert not p1.lean
        p2 = LDSEStepPlan(B, T, n, dev)
        p2.launch(*args, False, True, True)
        s2 = p2.sample(eps)
        assert torch.equal(s1, s2) and torch.equal(p1.lognorm, p2.lognorm) and torch.equal(p1.E_pair, p2.E_pair)


def test_lean_path_reports_a_non_positive_definite_sequence():
    """The reference ignores LAPACK info (cython_gaussian_grads.pxd:54-76); the lean kernels raise the plan's status word
    like the others: first offending sequence + 1."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable
    n, T, B, S = 4, 7, 6, 1
    init, pair, node, g = _setup(n, T, B, S, 9)
    node = [np.array(x) for x in node]
    node[0][3, 2, :] = +1e6          # -1/2 J > 0  => indefinite filtered precision in sequence 3
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev)
    plan = _plan(B, T, n, lean=True)
    eps = t(np.random.default_rng(0).standard_normal((B, T, S, n)))
    lds_inference_differentiable((tuple(t(x) for x in init), tuple(t(x) for x in pair)),
                                 (t(node[0]), t(node[1])), eps=eps, plan=plan)
    assert plan.lean
    with pytest.raises(FloatingPointError, match="sequence 3"):
        plan.check_info()


def test_lean_training_step_replays_as_one_hip_graph():
    """No host synchronisation, no allocation outside the graph's pool: the one-call inference + VJP on lean records
    capture into a hipGraph (what a training loop does with the step) and replay to the same bits."""
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd import _lib
    n, T, B, S = 10, 30, 12, 1
    init, pair, node, g = _setup(n, T, B, S, 21)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1),
            t(node[0]), t(node[1]), t(node[2])]
    eps = t(np.random.default_rng(1).standard_normal((B, T, S, n)))
    gs = [t(g["ln"]), t(g["dxx"]), t(g["x"]), t(g["s"])]
    plan = LDSEStepPlan(B, T, n, dev, options=_lib.OPT_LEAN_ON)
    smp = torch.empty_like(eps)
    plan.infer(*args, False, eps, smp)
    gJ0, gh0 = plan.vjp(gs[0], gs[1], gs[2], gs[3], eps, smp)          # eager (also sizes the VJP workspace)
    ref_out = [smp.clone(), gJ0.clone(), gh0.clone(), plan.lognorm.clone()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            plan.infer(*args, False, eps, smp)
            gJ, gh = plan.vjp(gs[0], gs[1], gs[2], gs[3], eps, smp)
    smp.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip([smp, gJ, gh, plan.lognorm], ref_out):
        assert torch.equal(a, b)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("n,T,B,S,batched", [(10, 30, 5, 1, True), (4, 9, 7, 2, False), (7, 2, 3, 1, True), (10, 3, 6, 0, True),
                                             (3, 17, 9, 1, True)])
def test_lean_forward_only_with_per_step_pair_parameters(n, T, B, S, batched):
    """keep_vjp = 0 (forward values only): lean records also serve per-step (T-1,n,n) and per-sequence (B,T-1,n,n) pair
    parameters -- the final pass of the SLDS's run_inference (slds_svae.py:289-310).  Against the reference's compiled
    E-step and sampler on every sequence (per-step pair parameters: cython_lds_inference.pyx:17-26, 73; 1e-6, north_star
    asks 1e-5) and against the full-record kernels of this library."""
    from svae_amd import _lib
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(13 * n + T + B)
    init = rand_lds_natparam(n, rng)[0]
    sets = B if batched else 1
    pairs = [[rand_lds_natparam(n, rng)[1] for _ in range(T - 1)] for _ in range(sets)]
    pk = [np.stack([np.stack([np.asarray(p[i], float) for p in seq]) for seq in pairs]) for i in range(4)]   # (sets,T-1,..)
    node = rand_node_potentials((B, T, n), rng, with_logZ=True)
    dev = torch.device("cuda:0")
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    pd = [t(x if batched else x[0]) for x in pk]
    args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), pd[0], pd[1], pd[2], pd[3].reshape(-1),
            t(node[0]), t(node[1]), t(node[2])]
    eps_np = np.zeros((B, T, max(S, 1), n))
    want = []
    for b in range(B):
        pb = tuple(x[b if batched else 0] for x in pk)
        nb = tuple(x[b] for x in node)
        ln, stats = ref.estep((init, pb), nb)
        smp = None
        if S:
            smp, e = ref.sample_backward((init, pb), nb, S, seed=50 + b)
            eps_np[b] = e
        want.append((ln, stats, smp))
    eps = t(eps_np) if S else None
    out = {}
    for name, opt in (("lean", _lib.OPT_LEAN_ON), ("full", _lib.OPT_LEAN_OFF)):
        plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=batched, options=opt)
        smp = plan.infer(*args, batched, eps, keep_vjp=False)
        assert plan.lean == (name == "lean")
        plan.check_info()
        out[name] = [x.clone() for x in (plan.lognorm, plan.E_init, plan.E_pair, plan.E_node_diagxx, plan.E_node_x)] \
            + ([smp.clone()] if S else [])
        if name == "lean":
            with pytest.raises(RuntimeError):
                plan.vjp(torch.zeros(B, dtype=torch.float64, device=dev))
    for a, b_ in zip(out["lean"], out["full"]):
        assert _rel(a, b_.cpu().numpy()) < 1e-7      # (a fresh random pair per step: conditioning noise ~1e-8, as between the reference and either)
    lognorm, E_init, E_pair, dxx, ex = out["lean"][:5]
    for b in range(B):
        ln, (oi, op, on), smp = want[b]
        assert _rel(lognorm[b], ln) < 1e-6
        assert _rel(E_init[b, :n * n].reshape(n, n), oi[0]) < 1e-6 and _rel(E_init[b, n * n:], oi[1]) < 1e-6
        for i in range(3):
            assert _rel(E_pair[b, :, i], np.asarray(op[i])) < 1e-6, "E_pair[%d]" % i
        assert _rel(dxx[b], on[0]) < 1e-6 and _rel(ex[b], on[1]) < 1e-6
        if S:
            assert _rel(out["lean"][5][b], smp) < 1e-6, "samples"
