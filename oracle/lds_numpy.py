"""Plain-NumPy restatement of the reference's natural-parameter LDS E-step.  TEST INFRASTRUCTURE.

Follows, function by function, the reference's pure-Python path (which cannot be imported here:
Python 2 + autograd) and agrees with its compiled Cython twin:

  natural_condition_on_general   svae/lds/gaussian.py:46-49     (cython_gaussian_grads.pxd:125-144)
  natural_predict                svae/lds/gaussian.py:52-66     (cython_gaussian_grads.pxd:38-82)
  natural_lognorm                svae/lds/gaussian.py:80-84     (cython_gaussian_grads.pxd:167-188)
  natural_rts_backward_step      svae/lds/gaussian.py:87-106    (cython_gaussian_grads.pxd:240-296)
  info_to_mean / natural_to_mean svae/lds/gaussian.py:117-127   (cython_gaussian_grads.pxd:208-220)
  natural_filter_forward_general svae/lds/lds_inference.py:86-106 (cython_lds_inference.pyx:28-90)
  natural_smoother_general       svae/lds/lds_inference.py:127-178 (cython_lds_inference.pyx:149-210)
  natural_lds_estep_general      svae/lds/lds_inference.py:223-229
  natural_sample_backward_general svae/lds/lds_inference.py:109-124 (cython_lds_inference.pyx:310-355)
  mean_to_natural / pair_mean_to_natural  svae/lds/gaussian.py:110-114,130-143

All API-level arrays are NATURAL parameters (-1/2 J, h), C-order, time-major, float64.
"""
import numpy as np
from scipy.linalg import solve_triangular as _st


def _solve_tri(L, x, trans="N"):
    return _st(L, x, lower=True, trans=trans)


# --- per-step kernels ---------------------------------------------------------------------------

def natural_condition_on_general(J, h, Jo, ho, logZo):
    """gaussian.py:46-49 -- add a node potential (diagonal (n,) or dense (n,n))."""
    Jo = Jo if Jo.ndim == 2 else np.diag(Jo)
    return (J + Jo, h + ho), logZo


def natural_predict(J, h, J11, J12, J22, logZ):
    """gaussian.py:52-66 -- marginalise x_t out of the 2x2-block precision via Cholesky."""
    J, J11, J12, J22 = -2 * J, -2 * J11, -J12, -2 * J22
    L = np.linalg.cholesky(J + J11)
    v = _solve_tri(L, h)
    lognorm = 0.5 * np.dot(v, v) - np.sum(np.log(np.diag(L)))
    h_predict = -np.dot(J12.T, _solve_tri(L, v, trans="T"))
    temp = _solve_tri(L, J12)
    J_predict = J22 - np.dot(temp.T, temp)
    return (-0.5 * J_predict, h_predict), lognorm + logZ


def natural_lognorm(J, h):
    """gaussian.py:80-84.  NB: no (n/2) log 2pi term, as in the reference."""
    J = -2 * J
    L = np.linalg.cholesky(J)
    v = _solve_tri(L, h)
    return 0.5 * np.dot(v, v) - np.sum(np.log(np.diag(L)))


def info_to_mean(J, h):
    """gaussian.py:117-121."""
    Sigma = np.linalg.inv(J)
    return np.dot(Sigma, h), Sigma


def natural_rts_backward_step(next_smooth, next_pred, filtered, pair_param):
    """gaussian.py:87-106."""
    (Jns, hns, mun), (Jnp, hnp), (Jf, hf) = next_smooth, next_pred, filtered
    J11, J12, J22 = pair_param[:3]
    Jns, Jnp, Jf, J11, J12, J22 = -2 * Jns, -2 * Jnp, -2 * Jf, -2 * J11, -J12, -2 * J22

    J11, J12, J22 = Jf + J11, J12, Jns - Jnp + J22
    L = np.linalg.cholesky(J22)
    temp = _solve_tri(L, J12.T)
    Js = J11 - np.dot(temp.T, temp)
    hs = hf - np.dot(temp.T, _solve_tri(L, hns - hnp))

    mu, sigma = info_to_mean(Js, hs)
    ExnxT = -_solve_tri(L, _solve_tri(L, np.dot(J12.T, sigma)), trans="T") + np.outer(mun, mu)
    ExxT = sigma + np.outer(mu, mu)
    return -0.5 * Js, hs, (mu, ExxT, ExnxT)


# --- parameter bookkeeping ----------------------------------------------------------------------

def _canonical_init_params(init_params):
    """lds_inference.py:62-63."""
    return init_params[0], init_params[1], sum(init_params[2:])


def _canonical_node_params(node_params):
    """lds_inference.py:65-82 (shape checks; adds a zero logZ if missing)."""
    node_params = tuple(np.asarray(a, dtype=np.float64) for a in node_params)
    ndims = tuple(a.ndim for a in node_params)
    if ndims not in [(3, 2, 1), (3, 2), (2, 2, 1), (2, 2)]:
        raise ValueError
    T, N = node_params[1].shape
    if len(node_params) == 2:
        node_params = node_params + (np.zeros(T),)
    shapes = tuple(a.shape for a in node_params)
    if shapes not in [((T, N, N), (T, N), (T,)), ((T, N), (T, N), (T,))]:
        raise ValueError
    return node_params


def _pair_at(pair_params, t):
    """_repeat_param, lds_inference.py:49-60: homogeneous (J11.ndim==2) or per-step (T-1,n,n)."""
    J11, J12, J22, logZ = pair_params
    if np.ndim(J11) == 2:
        return J11, J12, J22, logZ
    return J11[t], J12[t], J22[t], np.asarray(logZ)[t]


# --- filter / smoother / E-step -----------------------------------------------------------------

def natural_filter_forward_general(init_params, pair_params, node_params):
    """lds_inference.py:86-106.  Returns ((J_pred,h_pred),(J_filt,h_filt)) stacked over time (the
    compiled path's output format, cython_lds_inference.pyx:84-90) and lognorm."""
    J, h, lognorm = _canonical_init_params(init_params)
    Jn, hn, zn = _canonical_node_params(node_params)
    T = hn.shape[0]
    Jp, hp, Jf, hf = [], [], [], []
    for t in range(T):
        Jp.append(J), hp.append(h)
        (J, h), term = natural_condition_on_general(J, h, Jn[t], hn[t], zn[t])
        lognorm = lognorm + term
        Jf.append(J), hf.append(h)
        if t < T - 1:
            (J, h), term = natural_predict(J, h, *_pair_at(pair_params, t))
            lognorm = lognorm + term
    lognorm = lognorm + natural_lognorm(Jf[-1], hf[-1])
    stack = lambda xs: np.stack(xs, 0)
    return ((stack(Jp), stack(hp)), (stack(Jf), stack(hf))), lognorm


def natural_smoother_general(forward_messages, pair_params, diagonal_nodes=True):
    """lds_inference.py:127-178 / cython_lds_inference.pyx:149-210."""
    (Jp, hp), (Jf, hf) = forward_messages
    T = hf.shape[0]
    inhomog = np.ndim(pair_params[0]) == 3

    mu, Sigma = info_to_mean(-2 * Jf[-1], hf[-1])
    stats = [(mu, Sigma + np.outer(mu, mu), None)]
    smooth = (Jf[-1], hf[-1], mu)
    for t in range(T - 2, -1, -1):
        Js, hs, (mu, ExxT, ExnxT) = natural_rts_backward_step(
            smooth, (Jp[t + 1], hp[t + 1]), (Jf[t], hf[t]), _pair_at(pair_params, t))
        smooth = (Js, hs, mu)
        stats.insert(0, (mu, ExxT, ExnxT))

    E_init = (stats[0][1], stats[0][0], 1., 1.)
    pair = [(a[1], a[2].T, b[1], 1.) for a, b in zip(stats[:-1], stats[1:])]
    if inhomog:
        E_pair = tuple(np.stack([p[i] for p in pair], 0) for i in range(3)) + (np.ones(T - 1),)
    else:
        n = hf.shape[1]
        E_pair = tuple(sum((p[i] for p in pair), np.zeros((n, n))) for i in range(3)) \
            + (float(T - 1),)
    if diagonal_nodes:
        E_node = (np.stack([np.diag(s[1]) for s in stats]), np.stack([s[0] for s in stats]),
                  np.ones(T))
    else:
        E_node = (np.stack([s[1] for s in stats]), np.stack([s[0] for s in stats]), np.ones(T))
    return E_init, E_pair, E_node


def natural_lds_estep_general(natparam, node_params):
    """lds_inference.py:223-229 -> (lognorm, (E_init, E_pair, E_node))."""
    init_params, pair_params = natparam
    node_params = _canonical_node_params(node_params)
    messages, lognorm = natural_filter_forward_general(init_params, pair_params, node_params)
    stats = natural_smoother_general(messages, pair_params, node_params[0].ndim == 2)
    return lognorm, stats


def natural_sample_backward_general(forward_messages, pair_params, eps):
    """lds_inference.py:109-124 / cython_lds_inference.pyx:310-355, with the standard-normal draws
    passed in: ``eps[t]`` (S, n) is the noise used for x_t (the compiled path draws
    ``flipud(randn(T,S,N))`` so that eps[T-1] is drawn first; callers pass what they drew)."""
    (_, _), (Jf, hf) = forward_messages
    T, n = hf.shape
    S = eps.shape[1]
    out = np.zeros((T, S, n))

    def sample(J_nat, h, e):     # gaussian.py:69-77 (natural_sample), h: (S,n)
        L = np.linalg.cholesky(-2 * J_nat)
        noise = _solve_tri(L, e.T, trans="T")
        return _solve_tri(L, _solve_tri(L, h.T), trans="T").T + noise.T

    out[T - 1] = sample(Jf[T - 1], np.tile(hf[T - 1], (S, 1)), eps[T - 1])
    for t in range(T - 2, -1, -1):
        J11, J12 = _pair_at(pair_params, t)[:2]
        # natural_condition_on, gaussian.py:37-43: J + Jxx, h + Jxy y
        out[t] = sample(Jf[t] + J11, hf[t] + np.dot(J12, out[t + 1].T).T, eps[t])
    return out


# --- converting standard (mean) parameters to natural ones, gaussian.py:110-143 -----------------

def mean_to_natural(mu, sigma):
    neghalfJ = -0.5 * np.linalg.inv(sigma)
    h = np.linalg.solve(sigma, mu)
    logZ = -0.5 * np.dot(mu, h) - 0.5 * np.linalg.slogdet(sigma)[1]
    return neghalfJ, h, logZ


def pair_mean_to_natural(A, sigma):
    assert 2 <= A.ndim == sigma.ndim <= 3
    ein = 'tji,tjk->tik' if A.ndim == 3 else 'ji,jk->ik'
    trans = (0, 2, 1) if A.ndim == 3 else (1, 0)
    temp = np.linalg.solve(sigma, A)
    Jxx = -0.5 * np.einsum(ein, A, temp)
    Jxy = np.transpose(temp, trans)
    Jyy = -0.5 * np.linalg.inv(sigma)
    logZ = -0.5 * np.linalg.slogdet(sigma)[1]
    return Jxx, Jxy, Jyy, logZ


# --- brute-force dense check ---------------------------------------------

def dense_estep(natparam, node_params):
    """Assemble the (Tn x Tn) joint precision and solve densely.  Tiny T only.  Homogeneous or
    inhomogeneous pair params, diagonal node potentials."""
    (Ji, hi, zi), pair = (_canonical_init_params(natparam[0]), natparam[1])
    Jn, hn, zn = _canonical_node_params(node_params)
    T, n = hn.shape
    J = np.zeros((T * n, T * n))
    h = np.zeros(T * n)
    lz = zi + np.sum(zn)
    sl = lambda t: slice(t * n, (t + 1) * n)
    J[sl(0), sl(0)] += -2 * Ji
    h[sl(0)] += hi
    for t in range(T):
        J[sl(t), sl(t)] += -2 * (np.diag(Jn[t]) if Jn.ndim == 2 else Jn[t])
        h[sl(t)] += hn[t]
        if t < T - 1:
            J11, J12, J22, z = _pair_at(pair, t)
            J[sl(t), sl(t)] += -2 * J11
            J[sl(t + 1), sl(t + 1)] += -2 * J22
            J[sl(t), sl(t + 1)] += -J12
            J[sl(t + 1), sl(t)] += -J12.T
            lz += z
    Sigma = np.linalg.inv(J)
    mu = Sigma @ h
    lognorm = 0.5 * h @ mu - 0.5 * np.linalg.slogdet(J)[1] + lz
    M = Sigma + np.outer(mu, mu)
    Ex = mu.reshape(T, n)
    ExxT = np.stack([M[sl(t), sl(t)] for t in range(T)])
    ExxnT = np.stack([M[sl(t), sl(t + 1)] for t in range(T - 1)]) if T > 1 else np.zeros((0, n, n))
    return lognorm, Ex, ExxT, ExxnT
