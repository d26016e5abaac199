"""Dense (T,n,n) node potentials -- the reference's Python path (`natural_condition_on_general`,
/root/reference/svae/lds/gaussian.py:46-49; `_canonical_node_params` and the dense node statistics,
svae/lds/lds_inference.py:65-82, 163-166) -- run on the kernels by folding the off-diagonal part of each node
potential into per-step pair parameters (svae_amd/lds/lds_inference.py:_fold_dense_nodes).

Checked against the golden fixtures the reference's own Python path produced (tests/golden/lds_dense_*.npz, made by
tests/golden/make_golden.py) and against the NumPy restatement (oracle/lds_numpy.py), which is itself pinned on the same
fixtures in tests/test_oracle.py."""
import os

import numpy as np
import pytest
import torch

from oracle import lds_numpy
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(x):
    return torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device="cuda:0")


def _rel(got, want):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, float)
    want = np.asarray(want, float)
    return float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-300))


def _dense_nodes(shape, rng, scale=0.3):
    nJ, nh, nz = rand_node_potentials(shape, rng, with_logZ=True)
    n = shape[-1]
    A = scale * np.abs(nJ).mean() * rng.standard_normal(shape + (n,)) / n
    A = A + np.swapaxes(A, -1, -2)
    idx = np.arange(n)
    A[..., idx, idx] = nJ
    return A, nh, nz


def _check_stats(got, want, tol):
    (gi, gp, gn), (wi, wp, wn) = got, want
    for g, w in zip(gi[:2], wi[:2]):
        assert _rel(g, w) < tol
    for g, w in zip(gp[:3], wp[:3]):
        assert _rel(g, w) < tol
    assert _rel(gp[3], wp[3]) == 0.0
    for g, w in zip(gn, wn):
        assert _rel(g, w) < tol


@pytest.mark.parametrize("name", ["lds_dense_T7_n4", "lds_dense_T6_n3_inhomog"])
@pytest.mark.parametrize("batched", [True, False])
def test_dense_node_potentials_against_the_reference_python_path(name, batched):
    from svae_amd.lds.lds_inference import natural_lds_estep_general
    g = np.load(os.path.join(GOLD, name + ".npz"))
    natparam = ((_t(g["init_J"]), _t(g["init_h"]), _t(g["init_logZ"])),
                (_t(g["J11"]), _t(g["J12"]), _t(g["J22"]), _t(g["logZ_pair"])))
    B = g["node_h"].shape[0]
    if batched:
        lognorm, (Ei, Ep, En) = natural_lds_estep_general(natparam, (_t(g["node_J"]), _t(g["node_h"]), _t(g["node_logZ"])))
        outs = [(lognorm[b], tuple(x[b] for x in Ei), tuple(x[b] for x in Ep), tuple(x[b] for x in En)) for b in range(B)]
    else:
        outs = []
        for b in range(B):
            lognorm, (Ei, Ep, En) = natural_lds_estep_general(
                natparam, (_t(g["node_J"][b]), _t(g["node_h"][b]), _t(g["node_logZ"][b])))
            outs.append((lognorm, tuple(x.clone() for x in Ei), tuple(x.clone() for x in Ep), tuple(x.clone() for x in En)))
    for b, (lognorm, Ei, Ep, En) in enumerate(outs):
        assert _rel(lognorm, g["lognorm"][b]) < 1e-11
        assert _rel(Ei[0], g["ExxT0"][b]) < 1e-10 and _rel(Ei[1], g["Ex0"][b]) < 1e-10
        for x, k in zip(Ep, ("Epair_xx", "Epair_xxn", "Epair_xnxn")):
            assert tuple(x.shape) == g[k][b].shape
            assert _rel(x, g[k][b]) < 1e-10
        assert tuple(En[0].shape) == g["Enode_xx"][b].shape          # (T,n,n): E[x x'] per step, not its diagonal
        assert _rel(En[0], g["Enode_xx"][b]) < 1e-10 and _rel(En[1], g["Enode_x"][b]) < 1e-10


@pytest.mark.parametrize("n,T,B,inhomog", [(10, 30, 5, False), (10, 17, 3, True), (1, 6, 2, False), (15, 4, 2, False),
                                           (20, 6, 3, False), (40, 5, 2, True), (7, 2, 4, False)])
def test_dense_estep_filter_and_sampler_against_the_oracle(n, T, B, inhomog):
    """E-step, forward messages and the sampler (same eps) on dense node potentials vs oracle/lds_numpy.py; the register
    kernels (n <= 15) and the tile kernels (n >= 16)."""
    from svae_amd.lds.lds_inference import (natural_filter_forward_general, natural_lds_estep_general,
                                            natural_lds_inference_general, natural_lds_sample)
    rng = np.random.default_rng(100 * n + T)
    init, pair = rand_lds_natparam(n, rng)
    if inhomog:
        pairs = [rand_lds_natparam(n, rng)[1] for _ in range(T - 1)]
        pair = tuple(np.stack([p[i] for p in pairs]) for i in range(4))
    node = _dense_nodes((B, T, n), rng)
    S = 3
    eps = rng.standard_normal((B, T, S, n))
    natparam = (tuple(_t(x) for x in init), tuple(_t(x) for x in pair))
    nodes = tuple(_t(x) for x in node)
    tol = 1e-8 if n <= 15 else 1e-6
    lognorm, stats = natural_lds_estep_general(natparam, nodes)
    stats = tuple(tuple(x.clone() for x in grp) for grp in stats)
    lognorm = lognorm.clone()
    samples, stats2, lognorm2 = natural_lds_inference_general(natparam, nodes, num_samples=S, eps=_t(eps))
    samples = samples.clone()
    only = natural_lds_sample(natparam, nodes, num_samples=S, eps=_t(eps))
    assert _rel(only, samples.cpu().numpy()) < 1e-9
    if n <= 15:
        ((Jp, hp), (Jf, hf)), ln3 = natural_filter_forward_general(natparam[0], natparam[1], nodes)
    for b in range(B):
        nb = tuple(x[b] for x in node)
        msgs, ln = lds_numpy.natural_filter_forward_general(init, pair, nb)
        want = lds_numpy.natural_lds_estep_general((init, pair), nb)
        assert _rel(lognorm[b], want[0]) < tol and _rel(lognorm2[b], want[0]) < tol
        for st in (stats, stats2):
            _check_stats(tuple(tuple(x[b] for x in grp) for grp in st), want[1], tol)
        ws = lds_numpy.natural_sample_backward_general(msgs, pair, eps[b])
        assert _rel(samples[b], ws) < 100 * tol
        if n <= 15:
            assert _rel(ln3[b], ln) < tol
            for got, w in ((Jp[b], msgs[0][0]), (hp[b], msgs[0][1]), (Jf[b], msgs[1][0]), (hf[b], msgs[1][1])):
                assert _rel(got, w) < tol


def test_dense_single_step_and_diagonal_limit():
    """T = 1 (the potential goes into the initial block) and a dense potential that IS diagonal (same numbers as the
    diagonal entry point)."""
    from svae_amd.lds.lds_inference import natural_lds_estep_general
    rng = np.random.default_rng(5)
    n = 6
    init, pair = rand_lds_natparam(n, rng)
    natparam = (tuple(_t(x) for x in init), tuple(_t(x) for x in pair))
    node = _dense_nodes((1, n), rng)
    lognorm, stats = natural_lds_estep_general(natparam, tuple(_t(x) for x in node))
    want = lds_numpy.natural_lds_estep_general((init, pair), node)
    assert _rel(lognorm, want[0]) < 1e-10
    assert _rel(stats[0][0], want[1][0][0]) < 1e-10 and _rel(stats[2][0], want[1][2][0]) < 1e-10
    assert _rel(stats[2][1], want[1][2][1]) < 1e-10
    with pytest.raises(ValueError):
        natural_lds_estep_general(natparam, tuple(_t(np.stack([x, x])) for x in node))     # T = 1, two sequences
    T = 9
    nJ, nh, nz = rand_node_potentials((T, n), rng, with_logZ=True)
    dense = np.zeros((T, n, n))
    dense[:, np.arange(n), np.arange(n)] = nJ
    l1, s1 = natural_lds_estep_general(natparam, (_t(dense), _t(nh), _t(nz)))
    l1, s1 = l1.clone(), tuple(tuple(x.clone() for x in grp) for grp in s1)
    l2, s2 = natural_lds_estep_general(natparam, (_t(nJ), _t(nh), _t(nz)))
    assert _rel(l1, l2.cpu().numpy()) < 1e-12
    for a, b in zip(s1[1][:3], s2[1][:3]):
        assert _rel(a, b.cpu().numpy()) < 1e-11
    assert _rel(torch.diagonal(s1[2][0], dim1=-2, dim2=-1), s2[2][0].cpu().numpy()) < 1e-11


def test_dense_rejections():
    from svae_amd.lds.lds_inference import LDSEStepPlan, lds_inference_differentiable, natural_lds_estep_general
    rng = np.random.default_rng(6)
    n, T = 4, 5
    init, pair = rand_lds_natparam(n, rng)
    natparam = (tuple(_t(x) for x in init), tuple(_t(x) for x in pair))
    node = tuple(_t(x) for x in _dense_nodes((2, T, n), rng))
    with pytest.raises(ValueError):
        natural_lds_estep_general(natparam, node, plan=LDSEStepPlan(2, T, n, "cuda:0"))
    with pytest.raises(ValueError):          # (dense nodes are differentiable since round 6; a caller's plan is still refused)
        lds_inference_differentiable(natparam, node, plan=LDSEStepPlan(2, T, n, "cuda:0"))
    with pytest.raises(ValueError):
        natural_lds_estep_general(natparam, (node[0][:, :, :, :3], node[1]))


@pytest.mark.parametrize("n,T,B,S,inhomog", [(3, 5, 2, 2, False), (4, 7, 3, 1, True), (10, 12, 2, 1, False), (2, 2, 1, 1, False)])
def test_gradients_through_dense_node_potentials(n, T, B, S, inhomog):
    """The reference's Python path is differentiable end to end w.r.t. dense (T,n,n) node potentials
    (lds_inference.py:65-82, 205-218; autograd is not installed here, so no reference gradient exists to compare with).
    Checked: (1) forward values of the differentiable entry point equal the non-differentiable dense path (itself pinned to
    the reference's goldens above); (2) the gradient w.r.t. J (B,T,n,n), h, logZ of a random linear functional of ALL
    outputs -- lognorm, E[x x'] (B,T,n,n), E[x], samples, E_init, pair statistics -- against central finite differences
    in random directions (non-symmetric ones included: the gradient is the symmetric matrix); (3) on a diagonal J the
    diagonal of the dense gradient equals the diagonal kernels' gradient."""
    from svae_amd.lds.lds_inference import lds_inference_differentiable, natural_lds_inference_general
    rng = np.random.default_rng(11 * n + T)
    init, pair = rand_lds_natparam(n, rng)
    if inhomog:
        ps = [rand_lds_natparam(n, rng)[1] for _ in range(T - 1)]
        pair = tuple(np.stack([q[i] for q in ps]) for i in range(4))
    natparam = (tuple(_t(x) for x in init), tuple(_t(x) for x in pair))
    nJ, nh, nz = (_t(x) for x in _dense_nodes((B, T, n), rng))
    eps = _t(rng.standard_normal((B, T, S, n)))
    w = [_t(rng.standard_normal(s)) for s in ((B,), (B, T, n, n), (B, T, n), (B, T, S, n), (B, n * n + n))]
    wp = _t(rng.standard_normal((B, 3, n, n) if not inhomog else (B, T - 1, 3, n, n)))

    def f(J, h, z):
        lognorm, (ExxT, ex), samples, (E_init, E_pair) = lds_inference_differentiable(natparam, (J, h, z), eps=eps)
        return (w[0] * lognorm).sum() + (w[1] * ExxT).sum() + (w[2] * ex).sum() + (w[3] * samples).sum() \
            + (w[4] * E_init).sum() + (wp * E_pair).sum(), (lognorm, ExxT, ex, samples)

    J, h, z = nJ.clone().requires_grad_(True), nh.clone().requires_grad_(True), nz.clone().requires_grad_(True)
    val, (lognorm, ExxT, ex, samples) = f(J, h, z)
    # (1) forward values
    s2, (Ei2, Ep2, En2), ln2 = natural_lds_inference_general(natparam, (nJ, nh, nz), num_samples=S, eps=eps)
    assert _rel(lognorm, ln2.cpu().numpy()) < 1e-12 and _rel(ExxT, En2[0].cpu().numpy()) < 1e-12
    assert _rel(ex, En2[1].cpu().numpy()) < 1e-12 and _rel(samples, s2.cpu().numpy()) < 1e-12
    val.backward()
    assert float((J.grad - J.grad.transpose(-1, -2)).abs().max()) == 0.0
    # (2) finite differences (step 1e-5: at 1e-6 the kernels' own rounding noise, ~1e-11 of the outputs on these random
    #     per-step models, already shows at the 1e-5 level of the quotient)
    step = 1e-5
    for trial in range(8):
        dJ = _t(rng.standard_normal((B, T, n, n))) * (0.3 if trial % 2 else 1.0)
        if trial % 3 == 0:
            dJ = 0.5 * (dJ + dJ.transpose(-1, -2))
        dh, dz = _t(rng.standard_normal((B, T, n))), _t(rng.standard_normal((B, T)))
        with torch.no_grad():
            fp = float(f(nJ + step * dJ, nh + step * dh, nz + step * dz)[0])
            fm = float(f(nJ - step * dJ, nh - step * dh, nz - step * dz)[0])
        num = (fp - fm) / (2 * step)
        ana = float((J.grad * dJ).sum() + (h.grad * dh).sum() + (z.grad * dz).sum())
        # (the directional derivative is a sum of terms that largely cancel: the bound is relative to their absolute sum)
        scale = float((J.grad * dJ).abs().sum() + (h.grad * dh).abs().sum() + (z.grad * dz).abs().sum())
        assert abs(ana - num) < 1e-6 * scale + 1e-8, (trial, ana, num, scale)
    # (3) the diagonal limit
    Jd = torch.diag_embed(torch.diagonal(nJ, dim1=-2, dim2=-1)).clone().requires_grad_(True)
    jd = torch.diagonal(nJ, dim1=-2, dim2=-1).clone().requires_grad_(True)
    ln_a, _, smp_a, _ = lds_inference_differentiable(natparam, (Jd, nh), eps=eps)
    ((w[0] * ln_a).sum() + (w[3] * smp_a).sum()).backward()
    ln_b, _, smp_b, _ = lds_inference_differentiable(natparam, (jd, nh), eps=eps)
    ((w[0] * ln_b).sum() + (w[3] * smp_b).sum()).backward()
    assert _rel(torch.diagonal(Jd.grad, dim1=-2, dim2=-1), jd.grad.cpu().numpy()) < 1e-9
