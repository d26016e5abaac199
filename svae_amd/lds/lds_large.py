"""Sampler and reverse-mode derivative of the LDS E-step for latent dimension 16 <= n <= 64.

The E-step itself runs in the LDS-tiled MFMA kernel (csrc/lds_estep_tile.hip).  What follows it in a
training step -- `natural_sample_backward` (cython_lds_inference.pyx:310-355) and the three VJPs
(`natural_filter_grad` :92-145, `natural_smoother_general_grad` :236-306, `natural_sample_backward_grad`
:357-409) -- has no hand-written kernel at these sizes yet.  This module provides them on the device with
batched dense linear algebra (rocBLAS / rocSOLVER through torch, one launch per operation and time step:
launch-bound, a few hundred ms at T = 1000), so that BASELINE configs[4] can take a training step:

  * sampler: from the tile kernel's hand-off (G_t = -P_t^-1 J12, c_t, P_t^-1 per step).  The reference's
    noise map is chol(P_t)^-T eps_t; chol(P)^-T is the unique upper-triangular M with P^-1 = M M', i.e.
    the Cholesky factor of the index-reversed P^-1, index-reversed -- computed for all (sequence, step)
    at once; only the recursion x_t = c_t + G_t x_{t+1} + noise_t is serial.
  * VJP: backward() re-runs the forward recursion (information-form filter by Cholesky, moment-form
    smoother, sampler: the same algebra as the kernels) in torch with autograd enabled and differentiates
    it; the VALUES the model uses always come from the HIP kernel.

Everything here is float64 on the GPU; there is no CPU path.
"""
import torch

from .. import _lib


def _np16(n):
    return 16 * ((n + 15) // 16)


def handoff_views(plan):
    """(G (B,T,n,n), Pinv (B,T,n,n), c (B,T,n)) views of the tile kernel's workspace (TileCfg::WSTEP:
    X, P^-1 row-major NP x NP, then c)."""
    B, T, n = plan.B, plan.T, plan.n
    NP = _np16(n)
    W = 2 * NP * NP + NP
    ws = plan.ws[:B * T * W].view(B, T, W)
    G = ws[..., :NP * NP].view(B, T, NP, NP)[..., :n, :n]
    Pinv = ws[..., NP * NP:2 * NP * NP].view(B, T, NP, NP)[..., :n, :n]
    c = ws[..., 2 * NP * NP:2 * NP * NP + n]
    return G, Pinv, c


def sample_from_handoff(plan, eps, chunk_bytes=2 << 30):
    """Backward sampling after a tile-kernel E-step: eps (B,T,S,n) -> samples (B,T,S,n), same
    eps -> sample map as the reference (noise_t = chol(P_t)^-T eps_t)."""
    B, T, n = plan.B, plan.T, plan.n
    G, Pinv, c = handoff_views(plan)
    S = eps.shape[2]
    noise = torch.empty(B, T, S, n, dtype=torch.float64, device=plan.device)
    per_seq = T * n * n * 8 * 3
    step = max(1, int(chunk_bytes // per_seq))
    for b0 in range(0, B, step):
        P = Pinv[b0:b0 + step]
        Lf = torch.linalg.cholesky(P.flip(-1, -2))
        M = Lf.flip(-1, -2)                                    # upper triangular, P^-1 = M M'
        noise[b0:b0 + step] = torch.matmul(eps[b0:b0 + step], M.transpose(-1, -2))
    out = torch.empty_like(noise)
    x = c[:, T - 1, None, :] + noise[:, T - 1]
    out[:, T - 1] = x
    for t in range(T - 2, -1, -1):
        x = c[:, t, None, :] + noise[:, t] + torch.matmul(x, G[:, t].transpose(-1, -2))
        out[:, t] = x
    return out


def _pair_at(M, t):
    """pair parameter (n,n) | (T-1,n,n) | (B,T-1,n,n) at step t, broadcastable against (B,n,n)."""
    if M.dim() == 2:
        return M
    if M.dim() == 3:
        return M[t]
    return M[:, t]


def torch_estep(params, node_J, node_h, eps=None, per_step_stats=False):
    """Differentiable restatement of the E-step (+ sampler) on batched torch tensors: the algebra of the
    kernels (filter: P = J_pred + J11 + diag(J_node), Schur complement; smoother in moment form;
    sampler x_t = c_t + G_t x_{t+1} + chol(P_t)^-T eps_t).  Returns (lognorm (B), E_node_diagxx, E_node_x
    (B,T,n), samples | None, E_init (B, n*n+n), E_pair (B,3,n,n) or (B,T-1,3,n,n))."""
    init_J, init_h, init_logZ, J11, J12, J22, logZ_pair = params
    B, T, n = node_h.shape
    Jp = (-2.0 * init_J).expand(B, n, n)
    hp = init_h.expand(B, n)
    lognorm = init_logZ.reshape(()).expand(B).clone()
    Gs, cs, Pis, Ls = [], [], [], []
    eye = torch.eye(n, dtype=node_h.dtype, device=node_h.device)
    for t in range(T):
        last = t == T - 1
        P = Jp + torch.diag_embed(-2.0 * node_J[:, t])
        hf = hp + node_h[:, t]
        if not last:
            P = P + (-2.0) * _pair_at(J11, t)
        L = torch.linalg.cholesky(P)
        Lt = L.transpose(-1, -2)
        # P^-1 [h | I | J12] by two triangular solves (torch.cholesky_solve is unreliable on this ROCm build:
        # wrong results in 26 of 40 calls at n = 32, 37 of 40 at n = 48, right-hand side contiguous or not)
        rhs = [hf.unsqueeze(-1), eye.expand(B, n, n)]
        if not last:
            R = -_pair_at(J12, t)                               # info-form off-diagonal block
            rhs.append(R.expand(B, n, n))
        sol = torch.linalg.solve_triangular(Lt, torch.linalg.solve_triangular(L, torch.cat(rhs, -1), upper=False),
                                            upper=True)
        c = sol[..., 0]
        lognorm = lognorm + 0.5 * (hf * c).sum(-1) - torch.log(torch.diagonal(L, dim1=-1, dim2=-2)).sum(-1)
        Pis.append(sol[..., 1:n + 1])
        cs.append(c)
        Ls.append(L)
        if not last:
            X = sol[..., n + 1:]                                # P^-1 J12
            Gs.append(-X)
            Jp = -2.0 * _pair_at(J22, t) - torch.matmul(R.transpose(-1, -2), X)
            hp = -torch.matmul(R.transpose(-1, -2), c.unsqueeze(-1))[..., 0]
            if J11.dim() == 2:
                lognorm = lognorm + logZ_pair.reshape(-1)[0]
            elif J11.dim() == 3:
                lognorm = lognorm + logZ_pair.reshape(-1)[t]
            else:
                lognorm = lognorm + logZ_pair.reshape(B, T - 1)[:, t]
    Sig, m = Pis[T - 1], cs[T - 1]
    Exx = [None] * T
    Ex = [None] * T
    Ecr = [None] * (T - 1)
    Exx[T - 1] = Sig + m.unsqueeze(-1) * m.unsqueeze(-2)
    Ex[T - 1] = m
    for t in range(T - 2, -1, -1):
        G = Gs[t]
        W = torch.matmul(Sig, G.transpose(-1, -2))              # Cov(x_{t+1}, x_t)
        mn = cs[t] + torch.matmul(G, m.unsqueeze(-1))[..., 0]
        Sig = Pis[t] + torch.matmul(G, W)
        Sig = 0.5 * (Sig + Sig.transpose(-1, -2))
        Ecr[t] = W.transpose(-1, -2) + mn.unsqueeze(-1) * m.unsqueeze(-2)      # E[x_t x_{t+1}']
        m = mn
        Exx[t] = Sig + m.unsqueeze(-1) * m.unsqueeze(-2)
        Ex[t] = m
    Exs = torch.stack(Ex, 1)
    dxx = torch.stack([torch.diagonal(e, dim1=-1, dim2=-2) for e in Exx], 1)
    E_init = torch.cat([Exx[0].reshape(B, n * n), Ex[0]], -1)
    if T > 1:
        if per_step_stats:
            E_pair = torch.stack([torch.stack(Exx[:-1], 1), torch.stack(Ecr, 1), torch.stack(Exx[1:], 1)], 2)
        else:
            E_pair = torch.stack([sum(Exx[:-1]), sum(Ecr), sum(Exx[1:])], 1)
    else:
        E_pair = torch.zeros(B, 0, 3, n, n, dtype=node_h.dtype, device=node_h.device) if per_step_stats \
            else torch.zeros(B, 3, n, n, dtype=node_h.dtype, device=node_h.device)
    samples = None
    if eps is not None:
        out = [None] * T
        for t in range(T - 1, -1, -1):
            noise = torch.linalg.solve_triangular(Ls[t].transpose(-1, -2), eps[:, t].transpose(-1, -2),
                                                  upper=True).transpose(-1, -2)        # (B,S,n)
            x = cs[t].unsqueeze(1) + noise
            if t < T - 1:
                x = x + torch.matmul(out[t + 1], Gs[t].transpose(-1, -2))
            out[t] = x
        samples = torch.stack(out, 1)
    return lognorm, dxx, Exs, samples, E_init, E_pair


class LDSInferenceLarge(torch.autograd.Function):
    """Differentiable (w.r.t. the node potentials) E-step + sampler for 16 <= n <= 64: forward = the
    tile kernel (+ sample_from_handoff), backward = autograd through torch_estep re-run on the same
    inputs."""

    @staticmethod
    def forward(ctx, node_J, node_h, node_logZ, eps, plan, params, pair_batched):
        init_J, init_h, init_logZ, J11, J12, J22, logZ_pair = params
        plan.launch(init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ,
                    pair_batched, False, False)
        samples = sample_from_handoff(plan, eps) if eps is not None else \
            torch.zeros(0, dtype=torch.float64, device=plan.device)
        ctx.params, ctx.inhomog, ctx.has_logZ, ctx.has_eps = params, plan.inhomog, node_logZ is not None, eps is not None
        ctx.save_for_backward(node_J, node_h, eps if eps is not None else samples)
        E_init, E_pair = plan.E_init.clone(), plan.E_pair.clone()
        if not plan.inhomog:
            ctx.mark_non_differentiable(E_init, E_pair)
        return (plan.lognorm.clone(), plan.E_node_diagxx.clone(), plan.E_node_x.clone(), samples, E_init, E_pair)

    @staticmethod
    def backward(ctx, g_lognorm, g_dxx, g_x, g_samples, g_init, g_pair):
        node_J, node_h, eps = ctx.saved_tensors
        with torch.enable_grad():
            nJ = node_J.detach().requires_grad_(True)
            nh = node_h.detach().requires_grad_(True)
            out = torch_estep(ctx.params, nJ, nh, eps if ctx.has_eps else None, per_step_stats=ctx.inhomog)
            pairs = [(out[0], g_lognorm), (out[1], g_dxx), (out[2], g_x)]
            if ctx.has_eps:
                pairs.append((out[3], g_samples))
            if ctx.inhomog:
                pairs += [(out[4], g_init), (out[5], g_pair)]
            ys = [y for y, g in pairs if g is not None]
            gs = [g for y, g in pairs if g is not None]
            gJ, gh = torch.autograd.grad(ys, [nJ, nh], gs, allow_unused=True)
        B, T = node_h.shape[:2]
        gz = g_lognorm[:, None].expand(B, T).clone() if (ctx.has_logZ and g_lognorm is not None) else None
        zero = lambda g, like: torch.zeros_like(like) if g is None else g
        return zero(gJ, node_J), zero(gh, node_h), gz, None, None, None, None
