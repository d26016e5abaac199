"""Extended-precision (x87 80-bit long double, eps ~1e-19) LDS E-step.  TEST INFRASTRUCTURE.

An arbiter for ill-conditioned models where the fp64 oracle (oracle/lds_numpy.py, a restatement of
svae/lds/lds_inference.py:86-178) and the fp64 kernels differ by more than rounding of the outputs:
the same model evaluated with ~3 more decimal digits tells which of the two carries the error.
Plain NumPy on np.longdouble arrays (no LAPACK: Gauss-Jordan inverse with vectorised row updates),
moment-form smoother; homogeneous or per-step pair parameters, diagonal node potentials.
"""
import numpy as np

LD = np.longdouble


def _inv_logdet(A):
    """SPD inverse + log det by Gauss-Jordan without pivoting, in long double."""
    A = np.array(A, dtype=LD)
    n = A.shape[0]
    inv = np.eye(n, dtype=LD)
    logdet = LD(0)
    for k in range(n):
        p = A[k, k]
        logdet += np.log(p)
        A[k] /= p
        inv[k] /= p
        f = A[:, k].copy()
        f[k] = 0
        A -= np.outer(f, A[k])
        inv -= np.outer(f, inv[k])
    return (inv + inv.T) / 2, logdet


def estep(natparam, node_params):
    (J0, h0, *z0), (J11, J12, J22, zp) = natparam
    nJ, nh = (np.asarray(x, dtype=LD) for x in node_params[:2])
    nz = np.asarray(node_params[2], dtype=LD) if len(node_params) > 2 else np.zeros(nh.shape[0], LD)
    T, n = nh.shape
    inhomog = np.ndim(J11) == 3
    pair = lambda M, t: np.asarray(M[t] if inhomog else M, dtype=LD)
    Lp = -2 * np.asarray(J0, LD)
    hp = np.asarray(h0, LD)
    lognorm = LD(sum(np.sum(z) for z in z0)) + nz.sum()
    lognorm += np.sum(np.asarray(zp, LD)) if inhomog else (T - 1) * LD(zp)
    Xs, cs, Pinvs = [], [], []
    for t in range(T):
        Lf = Lp + np.diag(-2 * nJ[t])
        hf = hp + nh[t]
        if t < T - 1:
            P = Lf + (-2 * pair(J11, t))
            R = pair(J12, t)
        else:
            P, R = Lf, np.zeros((n, n), LD)
        Pinv, logdet = _inv_logdet(P)
        X, c = Pinv.dot(R), Pinv.dot(hf)
        lognorm += LD(0.5) * hf.dot(c) - LD(0.5) * logdet
        Xs.append(X), cs.append(c), Pinvs.append(Pinv)
        if t < T - 1:
            Lp = -2 * pair(J22, t) - R.T.dot(X)
            hp = R.T.dot(c)
    Sig, m = np.zeros((n, n), LD), np.zeros(n, LD)
    Exx, Ex, Ecross = [None] * T, [None] * T, [None] * T
    for t in range(T - 1, -1, -1):
        W = Sig.dot(Xs[t].T)                         # Cov(x_{t+1}, x_t)
        mn = cs[t] + Xs[t].dot(m)
        Sig = Pinvs[t] + Xs[t].dot(W)
        Sig = (Sig + Sig.T) / 2
        Ecross[t] = (W + np.outer(m, mn)).T          # E[x_t x_{t+1}']
        m = mn
        Exx[t], Ex[t] = Sig + np.outer(m, m), m
    f = lambda x: np.asarray(x, dtype=np.float64)
    E_init = (f(Exx[0]), f(Ex[0]), 1., 1.)
    if inhomog:
        E_pair = (f(np.stack(Exx[:-1])), f(np.stack(Ecross[:-1])), f(np.stack(Exx[1:])), np.ones(T - 1))
    else:
        z = np.zeros((n, n), LD)
        E_pair = (f(sum(Exx[:-1], z)), f(sum(Ecross[:-1], z)), f(sum(Exx[1:], z)), float(T - 1))
    E_node = (f(np.stack([np.diag(e) for e in Exx])), f(np.stack(Ex)), np.ones(T))
    return float(lognorm), (E_init, E_pair, E_node)
