"""NumPy prototype of the reverse-mode (VJP) sweeps for the HIP E-step algorithm.

Development aid (not product, not oracle): states the adjoint algebra the HIP VJP kernel implements
and checks it against central finite differences and against the reference's compiled VJPs.
Forward (per step t):  P = A_t + diag(Jo_t);  (Pinv, X, cv) = (P^-1, P^-1 J12, P^-1 hf_t);
  A_{t+1} = C - J12' X,  hp_{t+1} = -J12' cv,  lognorm += 1/2 hf'cv - 1/2 log|P|
Smoother:  S~_t = G~_t S~_{t+1} G~_t' + diag(Pinv,0),  W~_t = S~_{t+1} G~_t',  G~_t = [[-X, cv],[0,1]]
Sampler:   x_t = cv_t - X_t x_{t+1} + chol(P_t)^-T eps_t
"""
import numpy as np


def forward(init, pair, node, eps=None):
    J0, h0 = -2 * init[0], init[1]
    J11, J12, J22 = -2 * pair[0], -pair[1], -2 * pair[2]
    Jo, ho = -2 * node[0], node[1]
    T, n = ho.shape
    A = J0 + (J11 if T > 1 else 0)
    hp = h0.copy()
    ln = init[2] + (T - 1) * pair[3] + (node[2].sum() if len(node) > 2 else 0.0)
    st = dict(Pinv=[], X=[], cv=[], G=[], S=[None] * (T + 1), W=[None] * T, U=[])
    for t in range(T):
        last = t == T - 1
        P = A + np.diag(Jo[t])
        hf = hp + ho[t]
        Pinv = np.linalg.inv(P)
        X = Pinv @ J12 if not last else np.zeros((n, n))
        cv = Pinv @ hf
        ln += 0.5 * hf @ cv - 0.5 * np.linalg.slogdet(P)[1]
        st["Pinv"].append(Pinv); st["X"].append(X); st["cv"].append(cv)
        st["U"].append(np.linalg.inv(np.linalg.cholesky(P)).T)
        G = np.zeros((n + 1, n + 1)); G[:n, :n] = -X; G[:n, n] = cv; G[n, n] = 1
        st["G"].append(G)
        if not last:
            A = J22 + (J11 if t + 1 < T - 1 else 0) - J12.T @ X
            hp = -J12.T @ cv
    S = np.zeros((n + 1, n + 1)); S[n, n] = 1
    st["S"][T] = S
    for t in range(T - 1, -1, -1):
        G = st["G"][t]
        W = S @ G.T
        Pi = np.zeros((n + 1, n + 1)); Pi[:n, :n] = st["Pinv"][t]
        S = G @ W + Pi
        st["S"][t] = S; st["W"][t] = W
    Ex = np.stack([st["S"][t][n, :n] for t in range(T)])
    dxx = np.stack([np.diag(st["S"][t])[:n] for t in range(T)])
    out = dict(lognorm=ln, Ex=Ex, dxx=dxx)
    if eps is not None:                      # eps (T,S,n)
        xs = [None] * T
        xn = np.zeros((eps.shape[1], n))
        for t in range(T - 1, -1, -1):
            x = st["cv"][t][None] - xn @ st["X"][t].T + eps[t] @ st["U"][t].T
            xs[t] = x; xn = x
        out["samples"] = np.stack(xs)
    return out, st


def vjp(init, pair, node, st, out, g_ln, g_Ex, g_dxx, g_samples=None):
    """-> (g_node_J, g_node_h) w.r.t. the NATURAL node parameters (J = -1/2 precision diag, h)."""
    J12 = -pair[1]
    T, n = node[1].shape
    Xb = [np.zeros((n, n)) for _ in range(T)]       # adjoint of X_t
    cb = [np.zeros(n) for _ in range(T)]            # adjoint of cv_t
    Pib = [np.zeros((n, n)) for _ in range(T)]      # adjoint of Pinv_t
    Pb_extra = [np.zeros((n, n)) for _ in range(T)] # direct adjoint of P_t (sampler noise)
    # sweep 1: smoother adjoint, forward in time
    Sh = np.zeros((n + 1, n + 1))
    for t in range(T):
        G, Sn = st["G"][t], st["S"][t + 1]
        Sh = Sh.copy()
        Sh[n, :n] += g_Ex[t]
        Sh[np.arange(n), np.arange(n)] += g_dxx[t]
        Gb = Sh @ G @ Sn.T + Sh.T @ G @ Sn
        Xb[t] += -Gb[:n, :n]; cb[t] += Gb[:n, n]
        Pib[t] += Sh[:n, :n]
        Sh = G.T @ Sh @ G
    # sampler adjoint, forward in time
    if g_samples is not None:
        xs = out["samples"]
        xh = np.zeros_like(g_samples[0])
        for t in range(T):
            xh = g_samples[t] - (xh @ st["X"][t - 1] if t > 0 else 0)     # x^_t (S,n)
            cb[t] += xh.sum(0)
            if t < T - 1:
                Xb[t] += -xh.T @ xs[t + 1]
            # noise y = U e,  U = chol(P)^-T:  Pbar = -U Psi U',  Psi = sym-lower(e z'),  z = U' ybar
            U = st["U"][t]
            z = xh @ U                                   # (S,n): z_s = U' xh_s
            E = np.einsum('si,sj->ij', eps_cache[t], z)      # sum_s e_s z_s'
            Psi = np.tril(E, -1) + 0.5 * np.diag(np.diag(E))
            Psi = Psi + Psi.T
            Pb_extra[t] += -0.5 * U @ Psi @ U.T
    # sweep 2: filter adjoint, backward in time
    gJ = np.zeros((T, n)); gh = np.zeros((T, n))
    Ab = np.zeros((n, n)); hpb = np.zeros(n)
    for t in range(T - 1, -1, -1):
        Pinv, X, cv = st["Pinv"][t], st["X"][t], st["cv"][t]
        if t < T - 1:                                  # A_{t+1} = C - J12' X_t ; hp_{t+1} = -J12' cv_t
            Xb[t] += -J12 @ Ab
            cb[t] += -J12 @ hpb
        Bb = Pinv @ np.column_stack([Xb[t], cb[t]])    # adjoint of [J12 | hf]
        hfb = Bb[:, n]
        Pb = -Pinv @ Pib[t] @ Pinv - Bb[:, :n] @ X.T - np.outer(hfb, cv) + Pb_extra[t]
        # lognorm_t = 1/2 hf' P^-1 hf - 1/2 log|P|
        hfb = hfb + g_ln * cv
        Pb = Pb - 0.5 * g_ln * (np.outer(cv, cv) + Pinv)
        gJ[t] = -2 * np.diag(Pb)
        gh[t] = hfb
        Ab, hpb = Pb, hfb
    return gJ, gh


eps_cache = None

if __name__ == "__main__":
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    rng = np.random.default_rng(0)
    n, T, S = 3, 5, 2
    init, pair = rand_lds_natparam(n, rng)
    node = rand_node_potentials((T, n), rng, with_logZ=True)
    eps = rng.standard_normal((T, S, n))
    eps_cache = eps
    g_ln, g_Ex, g_dxx, g_x = rng.standard_normal(), rng.standard_normal((T, n)), rng.standard_normal((T, n)), rng.standard_normal((T, S, n))

    def obj(nJ, nh):
        o, _ = forward(init, pair, (nJ, nh, node[2]), eps)
        return g_ln * o["lognorm"] + (g_Ex * o["Ex"]).sum() + (g_dxx * o["dxx"]).sum() + (g_x * o["samples"]).sum()

    out, st = forward(init, pair, node, eps)
    gJ, gh = vjp(init, pair, node, st, out, g_ln, g_Ex, g_dxx, g_x)
    h = 1e-6
    nJ, nh = node[0], node[1]
    numJ, numh = np.zeros_like(nJ), np.zeros_like(nh)
    for idx in np.ndindex(*nJ.shape):
        d = np.zeros_like(nJ); d[idx] = h
        numJ[idx] = (obj(nJ + d, nh) - obj(nJ - d, nh)) / (2 * h)
        numh[idx] = (obj(nJ, nh + d) - obj(nJ, nh - d)) / (2 * h)
    print("gJ err", np.abs(gJ - numJ).max() / np.abs(numJ).max(), "gh err", np.abs(gh - numh).max() / np.abs(numh).max())
