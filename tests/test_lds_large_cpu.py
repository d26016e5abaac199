"""CPU test (no kernel launch) of the hand-written adjoint the tile VJP kernels implement for latent dimension 16..64
(tests/_lds_large_torch.py: vjp_from_handoff, the form csrc/lds_vjp_tile.hip is held to on the GPU) against torch
autograd through the restatement of the recursion (torch_estep) -- homogeneous, per-step and per-sequence pair
parameters, with and without sample cotangents, with the cotangents of E_init and of the per-step pair statistics."""
import numpy as np
import pytest
import torch

import _lds_large_torch as lds_large
from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials


@pytest.mark.parametrize("n,T,B,S,mode,with_init,with_pair", [
    (3, 5, 2, 2, "homog", False, False), (4, 1, 2, 1, "homog", True, False), (6, 7, 3, 0, "homog", True, False),
    (5, 6, 2, 3, "inhomog", True, False), (5, 4, 3, 1, "batched", False, False), (17, 9, 2, 2, "homog", True, False),
    (2, 2, 1, 1, "homog", True, False), (5, 6, 2, 2, "inhomog", True, True), (4, 5, 3, 0, "batched", False, True),
    (6, 2, 2, 1, "inhomog", True, True), (18, 4, 2, 1, "batched", True, True)])
def test_manual_adjoint_matches_autograd(n, T, B, S, mode, with_init, with_pair):
    rng = np.random.default_rng(n * 100 + T + B)
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64)
    init = rand_lds_natparam(n, rng)[0]
    if mode == "homog":
        J11, J12, J22, lz = (t(x) for x in rand_lds_natparam(n, rng)[1])
    else:
        sets = B if mode == "batched" else 1
        pairs = [[rand_lds_natparam(n, rng)[1] for _ in range(T - 1)] for _ in range(sets)]
        st = lambda i: t(np.stack([np.stack([p[i] for p in row]) for row in pairs]))
        J11, J12, J22, lz = (st(i) for i in range(4))
        if mode != "batched":
            J11, J12, J22, lz = J11[0], J12[0], J22[0], lz[0]
    params = (t(init[0]), t(init[1]), t(init[2]), J11, J12, J22, lz)
    nJ, nh = (t(x) for x in rand_node_potentials((B, T, n), rng))
    eps = t(rng.standard_normal((B, T, S, n))) if S else None
    g = dict(ln=t(rng.standard_normal(B)), dxx=t(rng.standard_normal((B, T, n))), x=t(rng.standard_normal((B, T, n))),
             s=t(rng.standard_normal((B, T, max(S, 1), n))), i=t(rng.standard_normal((B, n * n + n))),
             p=t(rng.standard_normal((B, max(T - 1, 0), 3, n, n))))
    a, b = nJ.clone().requires_grad_(True), nh.clone().requires_grad_(True)
    out, (G, Pinv, c) = lds_large.torch_estep(params, a, b, eps, per_step_stats=with_pair, return_handoff=True)
    loss = (g["ln"] * out[0]).sum() + (g["dxx"] * out[1]).sum() + (g["x"] * out[2]).sum()
    if S:
        loss = loss + (g["s"] * out[3]).sum()
    if with_init:
        loss = loss + (g["i"] * out[4]).sum()
    if with_pair:
        loss = loss + (g["p"] * out[5]).sum()
    wJ, wh = torch.autograd.grad(loss, [a, b])
    gJ, gh = lds_large.vjp_from_handoff(G.detach(), Pinv.detach(), c.detach(), out[2].detach(), J12, g["ln"], g["dxx"],
                                        g["x"], out[3].detach() if S else None, eps, g["s"] if S else None,
                                        g["i"] if with_init else None, g_E_pair=g["p"] if with_pair else None)
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    assert rel(gJ, wJ) < 1e-8 and rel(gh, wh) < 1e-8
