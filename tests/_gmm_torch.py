"""Plain torch fp64 reference of the differentiable tail of the GMM local step (test infrastructure): the one pass
after the fixed point that the reference keeps on the autograd tape, /root/reference/svae/models/gmm.py:74-86, and
gaussian.natural_sample (svae/distributions/gaussian.py:27-33) -- what svae_gmm_local_vjp_f64 / svae_gmm_sample_f64
(csrc/gmm_train.hip) are checked against through torch.autograd."""
import torch

from svae_amd.distributions import expfam


def final_pass_torch(label_global, gaussian_globals, node_dense, label_stats):
    """gmm.py:74-86 in torch: the ONE pass after the fixed point that the reference keeps on the
    autograd tape (`gaussian_meanfield` + `label_meanfield` on the boxed node potentials).  The fixed
    point itself (gmm.py:71, <= 100 sweeps, not differentiated: `getval`) runs in the HIP kernel."""
    N = node_dense.shape[-1] - 2
    gaussian_natparam = node_dense + torch.tensordot(label_stats, gaussian_globals, dims=([1], [0]))
    neghalfJ, h = gaussian_natparam[..., :N, :N], gaussian_natparam[..., :N, N]
    J = -2 * neghalfJ
    L = torch.linalg.cholesky(J)
    # J^-1 [h | I] by two triangular solves (torch.cholesky_solve / cholesky_inverse return wrong results
    # intermittently on this ROCm build for some sizes, see svae_amd/lds/lds_large.py)
    eye = torch.eye(N, dtype=J.dtype, device=J.device).expand(J.shape[0], N, N)
    sol = torch.linalg.solve_triangular(
        L.transpose(-1, -2), torch.linalg.solve_triangular(L, torch.cat([h.unsqueeze(-1), eye], -1), upper=False),
        upper=True)
    Ex = sol[..., 0]
    ExxT = sol[..., 1:] + Ex.unsqueeze(-1) * Ex.unsqueeze(-2)
    ones = torch.ones(Ex.shape[0], dtype=Ex.dtype, device=Ex.device)
    gaussian_stats = expfam.pack_dense(ExxT, Ex, ones, ones)
    logZ = 0.5 * (h * Ex).sum() - torch.log(torch.diagonal(L, dim1=-1, dim2=-2)).sum() \
        + (gaussian_natparam[..., N, N] + gaussian_natparam[..., N + 1, N + 1]).sum()
    gaussian_kl = (node_dense * gaussian_stats).sum() - logZ
    node_l = torch.tensordot(gaussian_stats, gaussian_globals, dims=([1, 2], [1, 2]))
    label_natparam = node_l + label_global
    label_stats_new = torch.softmax(label_natparam, dim=-1)
    label_kl = (label_stats_new * node_l).sum() - torch.logsumexp(label_natparam, dim=-1).sum()
    return (label_stats_new, gaussian_stats), (label_natparam, gaussian_natparam), label_kl + gaussian_kl


def sample_torch(gaussian_natparam, eps):
    return expfam.gaussian_natural_sample(gaussian_natparam, eps)
