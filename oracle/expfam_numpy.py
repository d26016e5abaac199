"""Plain-NumPy restatement of the reference's exponential-family building blocks.  TEST INFRASTRUCTURE.

  gaussian.pack_dense / unpack_dense      svae/distributions/gaussian.py:39-57
  gaussian.expectedstats / logZ           svae/distributions/gaussian.py:11-25
  categorical.expectedstats / logZ        svae/distributions/categorical.py:6-9, svae/util.py:23-24
  dirichlet.expectedstats / logZ          svae/distributions/dirichlet.py:5-11
  niw.expectedstats / logZ / conversions  svae/distributions/niw.py:15-42
  mniw.expectedstats / logZ / conversions svae/distributions/mniw.py:13-55
"""
import numpy as np
from scipy.special import digamma, gammaln, multigammaln, logsumexp

T_ = lambda X: np.swapaxes(X, -1, -2)
symmetrize = lambda X: (X + T_(X)) / 2.          # util.py:57
outer = lambda x, y: x[..., :, None] * y[..., None, :]   # util.py:46


# --- gaussian (dense-packed natural parameters) -------------------------------------------------

def pack_dense(A, b, *args):
    """gaussian.py:39-53."""
    leading_dim, N = b.shape[:-1], b.shape[-1]
    z1, z2 = np.zeros(leading_dim + (N, 1)), np.zeros(leading_dim + (1, 1))
    c, d = args if args else (z2, z2)
    A = A[..., None] * np.eye(N)[None, ...] if A.ndim == b.ndim else A
    b = b[..., None]
    c, d = np.reshape(c, leading_dim + (1, 1)), np.reshape(d, leading_dim + (1, 1))
    vs, hs = (lambda x: np.concatenate(x, axis=-2)), (lambda x: np.concatenate(x, axis=-1))
    return vs((hs((A, b, z1)), hs((T_(z1), c, z2)), hs((T_(z1), z2, d))))


def unpack_dense(arr):
    """gaussian.py:55-57."""
    N = arr.shape[-1] - 2
    return arr[..., :N, :N], arr[..., :N, N], arr[..., N, N], arr[..., N + 1, N + 1]


def gaussian_expectedstats(natparam):
    """gaussian.py:11-17."""
    neghalfJ, h, _, _ = unpack_dense(natparam)
    J = -2 * neghalfJ
    Ex = np.linalg.solve(J, h[..., None])[..., 0]
    ExxT = np.linalg.inv(J) + Ex[..., None] * Ex[..., None, :]
    En = np.ones(J.shape[0]) if J.ndim == 3 else 1.
    return pack_dense(ExxT, Ex, En, En)


def gaussian_logZ(natparam):
    """gaussian.py:19-25."""
    neghalfJ, h, a, b = unpack_dense(natparam)
    J = -2 * neghalfJ
    L = np.linalg.cholesky(J)
    return 0.5 * np.sum(h * np.linalg.solve(J, h[..., None])[..., 0]) \
        - np.sum(np.log(np.diagonal(L, axis1=-1, axis2=-2))) + np.sum(a + b)


# --- categorical / dirichlet --------------------------------------------------------------------

def softmax(x):
    """util.py:23-24."""
    e = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return e / np.sum(e, axis=-1, keepdims=True)

categorical_expectedstats = softmax


def categorical_logZ(natparam):
    """categorical.py:8-9."""
    return np.sum(logsumexp(natparam, axis=-1))


def dirichlet_expectedstats(natparam):
    """dirichlet.py:5-7."""
    alpha = natparam + 1
    return digamma(alpha) - digamma(np.sum(alpha, -1, keepdims=True))


def dirichlet_logZ(natparam):
    """dirichlet.py:9-11."""
    alpha = natparam + 1
    return np.sum(np.sum(gammaln(alpha), -1) - gammaln(np.sum(alpha, -1)))


# --- NIW (dense-packed) -------------------------------------------------------------------------

def niw_natural_to_standard(natparam):
    """niw.py:33-37."""
    A, b, kappa, nu = unpack_dense(natparam)
    m = b / np.expand_dims(kappa, -1)
    S = A - outer(b, m)
    return S, m, kappa, nu


def niw_standard_to_natural(S, m, kappa, nu):
    """niw.py:39-42."""
    b = np.expand_dims(kappa, -1) * m
    A = S + outer(b, m)
    return pack_dense(A, b, kappa, nu)


def niw_expectedstats(natparam, fudge=1e-8):
    """niw.py:15-25."""
    S, m, kappa, nu = niw_natural_to_standard(natparam)
    d = m.shape[-1]
    E_J = nu[..., None, None] * symmetrize(np.linalg.inv(S)) + fudge * np.eye(d)
    E_h = np.matmul(E_J, m[..., None])[..., 0]
    E_hTJinvh = d / kappa + np.matmul(m[..., None, :], E_h[..., None])[..., 0, 0]
    E_logdetJ = (np.sum(digamma((nu[..., None] - np.arange(d)[None, ...]) / 2.), -1)
                 + d * np.log(2.)) - np.linalg.slogdet(S)[1]
    return pack_dense(-0.5 * E_J, E_h, -0.5 * E_hTJinvh, 0.5 * E_logdetJ)


def niw_logZ(natparam):
    """niw.py:27-31."""
    S, m, kappa, nu = niw_natural_to_standard(natparam)
    d = m.shape[-1]
    return np.sum(d * nu / 2. * np.log(2.) + multigammaln_vec(nu / 2., d)
                  - nu / 2. * np.linalg.slogdet(S)[1] - d / 2. * np.log(kappa))


def multigammaln_vec(a, d):
    a = np.asarray(a, dtype=float)
    return np.vectorize(lambda x: multigammaln(x, d))(a)


# --- MNIW (tuple natural parameters) ------------------------------------------------------------

def mniw_standard_to_natural(nu, S, M, K):
    """mniw.py:23-29."""
    Kinv = np.linalg.inv(K)
    A = Kinv
    B = np.dot(Kinv, M.T)
    C = S + np.dot(M, B)
    return (A, B, C, nu)


def mniw_natural_to_standard(natparam):
    """mniw.py:31-38."""
    A, B, C, d = natparam
    nu = d
    K = symmetrize(np.linalg.inv(A))
    M = np.dot(K, B).T
    S = C - np.dot(M, B)
    return nu, S, M, K


def mniw_expectedstats(natparam, fudge=1e-8):
    """mniw.py:19-20,40-55 -> (-1/2 E[A' Sinv A], E[Sinv A]', -1/2 E[Sinv], 1/2 E[log|Sinv|])."""
    nu, S, M, K = mniw_natural_to_standard(natparam)
    m = M.shape[0]
    E_Sigmainv = nu * symmetrize(np.linalg.inv(S)) + fudge * np.eye(S.shape[0])
    E_Sigmainv_A = nu * np.linalg.solve(S, M)
    E_AT_Sigmainv_A = m * K + nu * symmetrize(np.dot(M.T, np.linalg.solve(S, M))) \
        + fudge * np.eye(K.shape[0])
    E_logdetSigmainv = digamma((nu - np.arange(m)) / 2.).sum() + m * np.log(2) \
        - np.linalg.slogdet(S)[1]
    return (-0.5 * E_AT_Sigmainv_A, E_Sigmainv_A.T, -0.5 * E_Sigmainv, 0.5 * E_logdetSigmainv)


def mniw_logZ(natparam):
    """mniw.py:13-17."""
    nu, S, _, K = mniw_natural_to_standard(natparam)
    n = S.shape[0]
    return n * nu / 2. * np.log(2) + multigammaln(nu / 2., n) \
        - nu / 2. * np.linalg.slogdet(S)[1] + n / 2. * np.linalg.slogdet(K)[1]
