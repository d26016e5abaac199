"""Sampler and reverse-mode derivative of the LDS E-step for latent dimension 16 <= n <= 64.

The E-step itself runs in the LDS-tiled MFMA kernel (csrc/lds_estep_tile.hip).  What follows it in a
training step -- `natural_sample_backward` (cython_lds_inference.pyx:310-355) and the three VJPs
(`natural_filter_grad` :92-145, `natural_smoother_general_grad` :236-306, `natural_sample_backward_grad`
:357-409) -- has no hand-written kernel at these sizes yet.  This module provides them on the device with
batched dense linear algebra (rocBLAS / rocSOLVER through torch, one launch per operation and time step:
launch-bound, a few hundred ms at T = 1000), so that BASELINE configs[4] can take a training step:

  * sampler: from the tile kernel's hand-off (G_t = -P_t^-1 J12, c_t, P_t^-1 per step).  The reference's
    noise map is chol(P_t)^-T eps_t; chol(P)^-T is the unique upper-triangular M with P^-1 = M M' ("UL"
    Cholesky), computed per (sequence, step) by svae_lds_tile_noise_f64 (one workgroup each: it does not depend
    on the recursion); the serial recursion x_t = c_t + G_t x_{t+1} + noise_t is svae_lds_tile_sample_f64.
  * VJP: `vjp_from_handoff` -- the adjoint of the kernels' recursion written out by hand on the tile kernel's
    hand-off (three passes over time of batched matrix products; the Cholesky adjoint of the noise factor
    batched over all (sequence, step) pairs), checked against autograd through `torch_estep` (the torch
    restatement of the recursion, which remains the fallback for per-step pair-statistic cotangents and was
    50x slower: 4.6 s at B = 64, T = 1000, n = 64).

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
    if S <= 16:      # noise factor per (sequence, step) and the serial recursion: two launches
        lib, p = _lib.load(), _lib.ptr
        eps = eps.contiguous()
        out = torch.empty_like(noise)
        rc = lib.svae_lds_tile_noise_f64(0, B, T, n, S, p(eps), p(noise), p(plan.ws), None, p(plan.info),
                                         _lib.current_stream(plan.device))
        _lib.check(rc, "svae_lds_tile_noise_f64")
        rc = lib.svae_lds_tile_sample_f64(B, T, n, S, p(noise), p(out), p(plan.ws), _lib.current_stream(plan.device))
        _lib.check(rc, "svae_lds_tile_sample_f64")
        return out
    per_seq = T * n * n * 8 * 3
    step = max(1, int(chunk_bytes // per_seq))
    for b0 in range(0, B, step):
        P = Pinv[b0:b0 + step]
        Lf = torch.linalg.cholesky(P.flip(-1, -2))
        M = Lf.flip(-1, -2)                                    # upper triangular, P^-1 = M M'
        noise[b0:b0 + step] = torch.matmul(eps[b0:b0 + step], M.transpose(-1, -2))
    # (more than 16 samples per sequence: batched library calls for the factor, a torch loop for the recursion)
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


def torch_estep(params, node_J, node_h, eps=None, per_step_stats=False, return_handoff=False):
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
    if return_handoff:      # (G (B,T,n,n) with G_{T-1} = 0, Pinv (B,T,n,n), c (B,T,n)) as the tile kernel hands them off
        Gall = torch.stack(Gs + [torch.zeros_like(Pis[0])], 1)
        return (lognorm, dxx, Exs, samples, E_init, E_pair), (Gall, torch.stack(Pis, 1), torch.stack(cs, 1))
    return lognorm, dxx, Exs, samples, E_init, E_pair


def _upper_factor(Pinv):
    """Upper-triangular M with Pinv = M M' (= chol(P)^-T: the reference's noise map), batched: the Cholesky
    factor of the index-reversed matrix, index-reversed."""
    return torch.linalg.cholesky(Pinv.flip(-1, -2)).flip(-1, -2)


def _upper_factor_adjoint(M, Mbar):
    """Cotangent of Pinv under Pinv -> M (upper, Pinv = M M'), given Mbar (upper): the Cholesky adjoint
    A_bar = sym(L^-T Phi(L' L_bar) L^-1) (Phi: lower triangle, diagonal halved) on the index-reversed problem."""
    L, Lbar = M.flip(-1, -2), Mbar.flip(-1, -2)
    K = torch.matmul(L.transpose(-1, -2), Lbar)
    Phi = torch.tril(K)
    Phi = Phi - 0.5 * torch.diag_embed(torch.diagonal(Phi, dim1=-1, dim2=-2))
    Q = torch.linalg.solve_triangular(L.transpose(-1, -2), Phi, upper=True)              # L^-T Phi
    Q = torch.linalg.solve_triangular(L.transpose(-1, -2), Q.transpose(-1, -2), upper=True).transpose(-1, -2)   # (...) L^-1
    return (0.5 * (Q + Q.transpose(-1, -2))).flip(-1, -2)


def vjp_from_handoff(G, Pinv, c, m, J12, g_lognorm, g_dxx, g_x, samples=None, eps=None, g_samples=None,
                     g_E_init=None, chunk_bytes=4 << 30):
    """Reverse-mode derivative of the E-step (+ sampler) w.r.t. the node potentials from the forward pass's own
    quantities -- the adjoint of the recursion the kernels run, written out by hand as batched matrix products
    (what natural_filter_grad / natural_smoother_general_grad / natural_sample_backward_grad compute,
    cython_lds_inference.pyx:92-145, 236-306, 357-409):
      G (B,T,n,n) = -P_t^-1 J12 (info form), Pinv (B,T,n,n) = P_t^-1, c (B,T,n) = P_t^-1 h_filt  -- the hand-off;
      m (B,T,n) = E[x_t]; J12: natural pair parameter (n,n) | (T-1,n,n) | (B,T-1,n,n);
      cotangents g_lognorm (B), g_dxx / g_x (B,T,n) of diag E[x x'] / E[x], g_samples (B,T,S,n) of the samples
      drawn with eps (B,T,S,n), g_E_init (B, n*n+n) of (E[x_0 x_0'], E[x_0]).
    Three passes over time: (0) smoothed covariances Sigma_t (backward in time, stored); (1) adjoint of the
    smoother / sampler recursions (forward in time): Sigma_bar, m_bar, x_bar -> per-step cotangents of G_t,
    c_t, P_t^-1; the cotangent through the noise factor chol(P_t)^-T is a Cholesky adjoint batched over ALL
    (sequence, step) pairs at once; (2) adjoint of the filter (backward in time).  -> (g_node_J, g_node_h)."""
    B, T, n = c.shape
    f64 = dict(dtype=c.dtype, device=c.device)
    zeros = lambda *shape: torch.zeros(*shape, **f64)
    R_at = lambda t: -_pair_at(J12, t)                      # info-form off-diagonal block of pair t
    tr = lambda A: A.transpose(-1, -2)
    mv = lambda A, v: torch.matmul(A, v.unsqueeze(-1))[..., 0]
    has_s = g_samples is not None
    g_dxx = zeros(B, T, n) if g_dxx is None else g_dxx
    g_x = zeros(B, T, n) if g_x is None else g_x
    # ---- pass 0: Sigma_t = Pinv_t + G_t Sigma_{t+1} G_t'
    Sig = torch.empty(B, T, n, n, **f64)
    Sig[:, T - 1] = Pinv[:, T - 1]
    for t in range(T - 2, -1, -1):
        S = Pinv[:, t] + torch.matmul(torch.matmul(G[:, t], Sig[:, t + 1]), tr(G[:, t]))
        Sig[:, t] = 0.5 * (S + tr(S))
    # ---- pass 1: adjoint of the smoother / sampler recursions
    Sb, mb = zeros(B, n, n), zeros(B, n)
    xb = zeros(B, samples.shape[2], n) if has_s else None
    Pinv_bar = torch.empty(B, T, n, n, **f64)
    c_bar = torch.empty(B, T, n, **f64)
    G_bar = torch.empty(B, max(T - 1, 0), n, n, **f64)
    xb_all = torch.empty(B, T, samples.shape[2], n, **f64) if has_s else None
    for t in range(T):
        Sb = Sb + torch.diag_embed(g_dxx[:, t])
        mb = mb + g_x[:, t] + 2.0 * g_dxx[:, t] * m[:, t]
        if t == 0 and g_E_init is not None:
            gS = g_E_init[:, :n * n].reshape(B, n, n)
            Sb = Sb + 0.5 * (gS + tr(gS))
            mb = mb + g_E_init[:, n * n:] + mv(gS + tr(gS), m[:, 0])
        Pinv_bar[:, t] = Sb
        cb = mb
        if has_s:
            xb = xb + g_samples[:, t]
            xb_all[:, t] = xb
            cb = cb + xb.sum(1)
        c_bar[:, t] = cb
        if t < T - 1:
            Gt = G[:, t]
            SG = torch.matmul(Sb, Gt)
            Gb = 2.0 * torch.matmul(SG, Sig[:, t + 1]) + mb.unsqueeze(-1) * m[:, t + 1].unsqueeze(-2)
            if has_s:
                Gb = Gb + torch.matmul(tr(xb), samples[:, t + 1])
                xb = torch.matmul(xb, Gt)
            G_bar[:, t] = Gb
            Sb = torch.matmul(tr(Gt), SG)
            Sb = 0.5 * (Sb + tr(Sb))
            mb = mv(tr(Gt), mb)
    del Sig
    if has_s:
        # noise_t = M_t eps_t, M_t = upper factor of Pinv_t: M_bar = triu(sum_s x_bar_s eps_s'), all (b,t) at once
        per_seq = T * n * n * 8 * 6
        step = max(1, int(chunk_bytes // per_seq))
        for b0 in range(0, B, step):
            sl = slice(b0, b0 + step)
            M = _upper_factor(Pinv[sl])
            Mbar = torch.triu(torch.matmul(tr(xb_all[sl]), eps[sl]))
            Pinv_bar[sl] += _upper_factor_adjoint(M, Mbar)
    # ---- pass 2: adjoint of the filter
    gJ, gh = torch.empty(B, T, n, **f64), torch.empty(B, T, n, **f64)
    Jb, hb = zeros(B, n, n), zeros(B, n)
    gl = g_lognorm.reshape(B, 1, 1)
    for t in range(T - 1, -1, -1):
        Pi, ct, cb = Pinv[:, t], c[:, t], c_bar[:, t]
        Pb = -torch.matmul(torch.matmul(Pi, Pinv_bar[:, t]), Pi)
        if t < T - 1:
            R = R_at(t)
            Xb = -torch.matmul(R.expand(B, n, n), Jb) - G_bar[:, t]
            cb = cb - mv(R.expand(B, n, n), hb)
            Pb = Pb + torch.matmul(torch.matmul(Pi, Xb), tr(G[:, t]))
        Pc = mv(Pi, cb)
        Pb = Pb - Pc.unsqueeze(-1) * ct.unsqueeze(-2) - 0.5 * gl * (ct.unsqueeze(-1) * ct.unsqueeze(-2)) - 0.5 * gl * Pi
        Pb = 0.5 * (Pb + tr(Pb))
        hfb = Pc + g_lognorm.reshape(B, 1) * ct
        gJ[:, t] = -2.0 * torch.diagonal(Pb, dim1=-1, dim2=-2)
        gh[:, t] = hfb
        Jb, hb = Pb, hfb
    return gJ, gh


def vjp_from_handoff_hip(plan, J12, pair_batched, ex, g_lognorm, g_dxx, g_x, samples=None, eps=None, g_samples=None,
                         g_E_init=None):
    """vjp_from_handoff on the device kernels (svae_lds_tile_vjp_f64: three phases, one workgroup per sequence,
    n x n state in LDS); the Cholesky adjoint of the sampler's noise factor -- parallel over all (sequence, step)
    pairs -- stays a batched library call between phases 1 and 2.  Reads the hand-off of the plan's last launch."""
    lib = _lib.load()
    B, T, n = plan.B, plan.T, plan.n
    dev = plan.device
    f64 = dict(dtype=torch.float64, device=dev)
    cont = lambda x: None if x is None else x.to(**f64).contiguous()
    g_lognorm, g_dxx, g_x, g_E_init = cont(g_lognorm), cont(g_dxx), cont(g_x), cont(g_E_init)
    has_s = g_samples is not None
    samples, eps, g_samples = (cont(samples), cont(eps), cont(g_samples)) if has_s else (None, None, None)
    S = samples.shape[2] if has_s else 0
    if S > 16:
        raise ValueError("at most 16 samples per sequence are differentiable")
    nws = int(lib.svae_lds_tile_vjp_workspace_doubles(max(B, 1), T, n, S))
    ws = torch.empty(nws, **f64)
    gJ, gh = torch.empty(B, T, n, **f64), torch.empty(B, T, n, **f64)
    J12 = cont(J12)
    inhomog = J12.dim() >= 3
    p = _lib.ptr

    def phase(k):
        rc = lib.svae_lds_tile_vjp_f64(k, B, T, n, S, int(inhomog), int(bool(pair_batched)), p(J12), p(g_lognorm),
                                       p(g_dxx), p(g_x), p(g_E_init), p(g_samples), p(samples), p(ex), p(gJ), p(gh),
                                       p(plan.ws), p(ws), nws, _lib.current_stream(dev))
        _lib.check(rc, "svae_lds_tile_vjp_f64")
    phase(0)
    phase(1)
    if has_s:    # Cholesky adjoint of the noise factor, one workgroup per (sequence, step), into pinv_bar
        rc = lib.svae_lds_tile_noise_f64(1, B, T, n, S, p(eps), None, p(plan.ws), p(ws), p(plan.info),
                                         _lib.current_stream(dev))
        _lib.check(rc, "svae_lds_tile_noise_f64")
    phase(2)
    return gJ, gh


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
        ctx.plan, ctx.epoch, ctx.pair_batched = plan, plan.epoch, pair_batched
        ctx.ex = plan.E_node_x.clone()
        ctx.samples = samples if eps is not None else None
        ctx.save_for_backward(node_J, node_h, eps if eps is not None else samples)
        E_init, E_pair = plan.E_init.clone(), plan.E_pair.clone()
        if not plan.inhomog:
            ctx.mark_non_differentiable(E_init, E_pair)
        return (plan.lognorm.clone(), plan.E_node_diagxx.clone(), plan.E_node_x.clone(), samples, E_init, E_pair)

    @staticmethod
    def backward(ctx, g_lognorm, g_dxx, g_x, g_samples, g_init, g_pair):
        node_J, node_h, eps = ctx.saved_tensors
        B, T = node_h.shape[:2]
        if g_pair is None or not ctx.inhomog:
            # the adjoint of the kernels' recursion, from the hand-off of THIS forward pass (vjp_from_handoff)
            plan = ctx.plan
            if plan.epoch != ctx.epoch:
                raise RuntimeError("LDSEStepPlan was launched again before backward(): the hand-off workspace of "
                                   "this forward pass is gone (use one plan per live autograd graph)")
            zero = lambda g, like: torch.zeros_like(like) if g is None else g
            gs = g_samples if (ctx.has_eps and g_samples is not None) else None
            gJ, gh = vjp_from_handoff_hip(plan, ctx.params[4], ctx.pair_batched, ctx.ex, zero(g_lognorm, plan.lognorm),
                                          g_dxx, g_x, ctx.samples if gs is not None else None,
                                          eps if gs is not None else None, gs, g_init if ctx.inhomog else None)
            gz = g_lognorm[:, None].expand(B, T).clone() if (ctx.has_logZ and g_lognorm is not None) else None
            return gJ, gh, gz, None, None, None, None
        # per-step pair-statistic cotangents (per-step pair parameters at n > 15): autograd through the torch
        # restatement of the recursion
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
