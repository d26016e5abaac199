"""Torch restatements of the n > 15 LDS path.  TEST INFRASTRUCTURE (moved out of svae_amd/ in round 3: the product
path at 16 <= n <= 64 is HIP kernels only, svae_amd/lds/lds_large.py).

  torch_estep       differentiable restatement of the E-step (+ sampler) recursion the tile kernel runs
                    (filter: P = J_pred + J11 + diag(J_node), Schur complement; smoother in moment form; sampler
                    x_t = c_t + G_t x_{t+1} + chol(P_t)^-T eps_t) -- autograd through it is the CPU check of
  vjp_from_handoff  the adjoint of that recursion written out by hand as batched matrix products on the tile
                    kernel's hand-off (what natural_filter_grad / natural_smoother_general_grad /
                    natural_sample_backward_grad / _compute_stats_grad compute, cython_lds_inference.pyx:92-145,
                    212-306, 357-409) -- the form the kernels of csrc/lds_vjp_tile.hip are held to on the GPU.
"""
import torch


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
                     g_E_init=None, chunk_bytes=4 << 30, g_E_pair=None):
    """Reverse-mode derivative of the E-step (+ sampler) w.r.t. the node potentials from the forward pass's own
    quantities -- the adjoint of the recursion the kernels run, written out by hand as batched matrix products
    (what natural_filter_grad / natural_smoother_general_grad / natural_sample_backward_grad compute,
    cython_lds_inference.pyx:92-145, 236-306, 357-409):
      G (B,T,n,n) = -P_t^-1 J12 (info form), Pinv (B,T,n,n) = P_t^-1, c (B,T,n) = P_t^-1 h_filt  -- the hand-off;
      m (B,T,n) = E[x_t]; J12: natural pair parameter (n,n) | (T-1,n,n) | (B,T-1,n,n);
      cotangents g_lognorm (B), g_dxx / g_x (B,T,n) of diag E[x x'] / E[x], g_samples (B,T,S,n) of the samples
      drawn with eps (B,T,S,n), g_E_init (B, n*n+n) of (E[x_0 x_0'], E[x_0]), g_E_pair (B,T-1,3,n,n) of the
      per-step pair statistics (E x_t x_t', E x_t x_{t+1}', E x_{t+1} x_{t+1}').
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
        if g_E_pair is not None:
            sym2 = lambda A: A + tr(A)
            if t < T - 1:
                A0, A1 = g_E_pair[:, t, 0], g_E_pair[:, t, 1]
                Sb = Sb + 0.5 * sym2(A0)
                mb = mb + mv(sym2(A0), m[:, t]) + mv(A1, m[:, t + 1])
            if t > 0:
                A2, A1p = g_E_pair[:, t - 1, 2], g_E_pair[:, t - 1, 1]
                Sb = Sb + 0.5 * sym2(A2)
                mb = mb + mv(sym2(A2), m[:, t]) + mv(tr(A1p), m[:, t - 1])
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
            Sb = torch.matmul(tr(Gt), SG)
            if g_E_pair is not None and T > 1:      # E x_t x_{t+1}' = G_t Sigma_{t+1} + m_t m_{t+1}'
                Gb = Gb + torch.matmul(g_E_pair[:, t, 1], Sig[:, t + 1])
                Sb = Sb + torch.matmul(tr(Gt), g_E_pair[:, t, 1])
            G_bar[:, t] = Gb
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
