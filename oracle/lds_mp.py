"""TEST INFRASTRUCTURE: the posterior means of an LDS in natural parameters by a 60-digit block-tridiagonal solve (mpmath).

Not a restatement of the reference's algorithm (that is oracle/lds_numpy.py, svae/lds/lds_inference.py:127-178) but the
arbiter between implementations of it where fp64 conditioning makes them disagree: the joint precision of x_0 .. x_{T-1}
under svae/lds/gaussian.py:52-66's conventions (precision blocks -2 J11, -J12, -2 J22 of a pair potential; node and init
potentials on the diagonal) is block tridiagonal; block LDL' forward, back substitution backward, everything in `dps`
digits.  Only tests/ and tools/ import this."""
import numpy as np


def smoothed_means_mp(init, pair, node_J, node_h, dps=60):
    """init = (J (n,n), h (n,)[, ..]), pair = (J11, J12, J22[, ..]) homogeneous, node_J / node_h (T, n) diagonal node
    potentials: E[x_t] (T, n) as float64 of the `dps`-digit solution"""
    import mpmath as mp
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        M = lambda a: mp.matrix(np.asarray(a, float).tolist())
        iJ, ih = np.asarray(init[0], float), np.asarray(init[1], float)
        J11, J12, J22 = [np.asarray(x, float) for x in pair[:3]]
        node_J, node_h = np.asarray(node_J, float), np.asarray(node_h, float)
        T, n = node_h.shape
        A, h = [], []
        for s in range(T):
            a = np.diag(node_J[s]) + (iJ if s == 0 else 0) + (J11 if s < T - 1 else 0) + (J22 if s > 0 else 0)
            A.append(M(-2 * a))
            h.append(M((node_h[s] + (ih if s == 0 else 0)).reshape(-1, 1)))
        Bm = M(-J12)                                           # block (t, t + 1) of the joint precision
        D, y = [A[0]], [h[0]]
        for s in range(1, T):
            Di = mp.inverse(D[s - 1])
            D.append(A[s] - Bm.T * Di * Bm)
            y.append(h[s] - Bm.T * Di * y[s - 1])
        x = [None] * T
        x[T - 1] = mp.lu_solve(D[T - 1], y[T - 1])
        for s in range(T - 2, -1, -1):
            x[s] = mp.lu_solve(D[s], y[s] - Bm * x[s + 1])
        return np.array([[float(x[s][i]) for i in range(n)] for s in range(T)])
    finally:
        mp.mp.dps = old
