"""Global -> local maps of the exponential-family priors, in float64 torch (device-agnostic).

Once-per-step O(K N^2) / O(n^3) host-or-device work that feeds the E-step kernels; mirrors
  dirichlet.expectedstats   /root/reference/svae/distributions/dirichlet.py:5-7
  niw.expectedstats         /root/reference/svae/distributions/niw.py:15-25   (dense-packed)
  niw.natural_to_standard   /root/reference/svae/distributions/niw.py:33-37
  mniw.expectedstats        /root/reference/svae/distributions/mniw.py:19-20,31-55
  gaussian.pack_dense / unpack_dense  /root/reference/svae/distributions/gaussian.py:39-57
These are not on the per-sequence hot path ("keep on host/torch").
"""
import math

import torch


def _sym(X):
    return (X + X.transpose(-1, -2)) / 2.


def pack_dense(A, b, c=None, d=None):
    """gaussian.py:39-53.  A: (...,N,N) or diagonal (...,N); b: (...,N); c,d: (...) or None (=0)."""
    lead, N = b.shape[:-1], b.shape[-1]
    out = torch.zeros(lead + (N + 2, N + 2), dtype=b.dtype, device=b.device)
    if A.dim() == b.dim():
        out[..., :N, :N] = torch.diag_embed(A)
    else:
        out[..., :N, :N] = A
    out[..., :N, N] = b
    if c is not None:
        out[..., N, N] = c
    if d is not None:
        out[..., N + 1, N + 1] = d
    return out


def unpack_dense(arr):
    """gaussian.py:55-57."""
    N = arr.shape[-1] - 2
    return arr[..., :N, :N], arr[..., :N, N], arr[..., N, N], arr[..., N + 1, N + 1]


def dirichlet_expectedstats(natparam):
    alpha = natparam + 1
    return torch.digamma(alpha) - torch.digamma(alpha.sum(-1, keepdim=True))


def niw_natural_to_standard(natparam):
    A, b, kappa, nu = unpack_dense(natparam)
    m = b / kappa.unsqueeze(-1)
    S = A - b.unsqueeze(-1) * m.unsqueeze(-2)
    return S, m, kappa, nu


def niw_standard_to_natural(S, m, kappa, nu):
    b = kappa.unsqueeze(-1) * m
    A = S + b.unsqueeze(-1) * m.unsqueeze(-2)
    return pack_dense(A, b, kappa, nu)


def niw_expectedstats(natparam, fudge=1e-8):
    S, m, kappa, nu = niw_natural_to_standard(natparam)
    d = m.shape[-1]
    eye = torch.eye(d, dtype=S.dtype, device=S.device)
    E_J = nu[..., None, None] * _sym(torch.linalg.inv(S)) + fudge * eye
    E_h = torch.matmul(E_J, m.unsqueeze(-1))[..., 0]
    E_hTJinvh = d / kappa + (m * E_h).sum(-1)
    ar = torch.arange(d, dtype=S.dtype, device=S.device)
    E_logdetJ = torch.digamma((nu.unsqueeze(-1) - ar) / 2.).sum(-1) + d * math.log(2.) \
        - torch.linalg.slogdet(S)[1]
    return pack_dense(-0.5 * E_J, E_h, -0.5 * E_hTJinvh, 0.5 * E_logdetJ)


def mniw_natural_to_standard(natparam):
    A, B, C, d = natparam
    K = _sym(torch.linalg.inv(A))
    M = torch.matmul(K, B).transpose(-1, -2)
    S = C - torch.matmul(M, B)
    return d, S, M, K


def mniw_standard_to_natural(nu, S, M, K):
    Kinv = torch.linalg.inv(K)
    B = torch.matmul(Kinv, M.transpose(-1, -2))
    return (Kinv, B, S + torch.matmul(M, B), nu)


def mniw_expectedstats(natparam, fudge=1e-8):
    """-> (-1/2 E[A' Sinv A], E[Sinv A]', -1/2 E[Sinv], 1/2 E[log|Sinv|]) = LDS pair natparams."""
    nu, S, M, K = mniw_natural_to_standard(natparam)
    m = M.shape[0]
    eyeS = torch.eye(S.shape[0], dtype=S.dtype, device=S.device)
    eyeK = torch.eye(K.shape[0], dtype=S.dtype, device=S.device)
    E_Sigmainv = nu * _sym(torch.linalg.inv(S)) + fudge * eyeS
    SinvM = torch.linalg.solve(S, M)
    E_Sigmainv_A = nu * SinvM
    E_AT_Sigmainv_A = m * K + nu * _sym(torch.matmul(M.transpose(-1, -2), SinvM)) + fudge * eyeK
    ar = torch.arange(m, dtype=S.dtype, device=S.device)
    E_logdet = torch.digamma((nu - ar) / 2.).sum() + m * math.log(2.) - torch.linalg.slogdet(S)[1]
    return (-0.5 * E_AT_Sigmainv_A, E_Sigmainv_A.transpose(-1, -2), -0.5 * E_Sigmainv, 0.5 * E_logdet)


# --- log-normalisers (for the global KL terms; svae/models/gmm.py:44-58, svae/models/lds.py:16-30) ---

def dirichlet_logZ(natparam):
    """dirichlet.py:9-11."""
    alpha = natparam + 1
    return (torch.lgamma(alpha).sum(-1) - torch.lgamma(alpha.sum(-1))).sum()


def niw_logZ(natparam):
    """niw.py:27-31 (dense-packed natparam, possibly stacked)."""
    S, m, kappa, nu = niw_natural_to_standard(natparam)
    d = m.shape[-1]
    return (d * nu / 2. * math.log(2.) + torch.special.multigammaln(nu / 2., d)
            - nu / 2. * torch.linalg.slogdet(S)[1] - d / 2. * torch.log(kappa)).sum()


def mniw_logZ(natparam):
    """mniw.py:13-17."""
    nu, S, _, K = mniw_natural_to_standard(natparam)
    n = S.shape[0]
    nu = torch.as_tensor(nu, dtype=S.dtype, device=S.device)
    return n * nu / 2. * math.log(2.) + torch.special.multigammaln(nu / 2., n) \
        - nu / 2. * torch.linalg.slogdet(S)[1] + n / 2. * torch.linalg.slogdet(K)[1]


def gaussian_natural_sample(natparam, eps):
    """gaussian.py:27-33: x = J^-1 h + L^-T eps, dense-packed natparam (...,N+2,N+2), eps (...,S,N)."""
    neghalfJ, h, _, _ = unpack_dense(natparam)
    J = -2 * neghalfJ
    L = torch.linalg.cholesky(J)
    noise = torch.linalg.solve_triangular(L.transpose(-1, -2), eps.transpose(-1, -2), upper=True)
    mean = torch.linalg.solve(J, h.unsqueeze(-1))[..., 0]
    return mean.unsqueeze(-2) + noise.transpose(-1, -2)
