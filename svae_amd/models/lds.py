"""LDS-SVAE inference glue on MI355X; mirrors /root/reference/svae/models/lds.py:16-52.

  lds_prior_expectedstats(natparam)             (lds.py:23-25)
  lds_prior_logZ(natparam)                      (lds.py:28-30)
  lds_prior_kl(global, prior, expected_stats)   (lds.py:16-20)
  run_inference(prior, global, nn_potentials, num_samples) -> (samples, global_stats, global_kl,
                                                               local_kl)      (lds.py:35-52)

The reference module is stale as shipped (it imports `svae.lds.niw` / `svae.lds.mniw`, which do not
exist); the formulas are taken from it with the exponential families of
`svae/distributions/{niw,mniw}.py`: the global natural parameter is
(NIW dense-packed (n+2,n+2), MNIW tuple (A,B,C,d)); the LDS init potential is the unpacked NIW
expected statistic, the pair potential the MNIW expected statistic.

New relative to the reference: `nn_potentials` may hold B sequences (B,T,n); statistics are then
summed over the batch (the reference's one-sequence-per-minibatch semantics with B = 1) and, under
torch.distributed, all-reduced across ranks by `svae_amd.parallel.allreduce_global_stats`.
"""
import torch

from ..distributions import expfam
from ..lds.lds_inference import LDSEStepPlan, natural_lds_estep_general, reduce_stats
from ..parallel import allreduce_lds_stats


def _dev64(x, device):
    # dtype given up front: torch.as_tensor(python_float) alone would round to float32
    t = x.detach() if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)
    return t.to(device=device, dtype=torch.float64).contiguous()


def lds_prior_expectedstats(natparam):
    niw_natparam, mniw_natparam = natparam
    return expfam.niw_expectedstats(niw_natparam), expfam.mniw_expectedstats(mniw_natparam)


def lds_prior_logZ(natparam):
    niw_natparam, mniw_natparam = natparam
    return expfam.niw_logZ(niw_natparam) + expfam.mniw_logZ(mniw_natparam)


def _contract(a, b):
    if isinstance(a, (tuple, list)):
        return sum(_contract(x, y) for x, y in zip(a, b))
    return (torch.as_tensor(a, dtype=torch.float64, device=b.device if isinstance(b, torch.Tensor) else None)
            * b).sum()


def lds_prior_kl(global_natparam, prior_natparam, expected_stats=None):
    if expected_stats is None:
        expected_stats = lds_prior_expectedstats(global_natparam)
    sub = lambda p, g: tuple(sub(x, y) for x, y in zip(p, g)) if isinstance(p, (tuple, list)) else p - g
    return -_contract(sub(prior_natparam, global_natparam), expected_stats) \
        + (lds_prior_logZ(prior_natparam) - lds_prior_logZ(global_natparam))


def local_natparam_from_global(global_natparam):
    """(init_params, pair_params) of the LDS from the global factors (lds.py:23-25, mniw.py:54-55),
    plus the expected statistics themselves (needed again by the global KL)."""
    es = lds_prior_expectedstats(global_natparam)
    init_params = expfam.unpack_dense(es[0])           # (-1/2 E[J], E[h], -1/2 E[h'J^-1 h], 1/2 E[log|J|])
    return (init_params, es[1]), es


def run_inference(prior_natparam, global_natparam, nn_potentials, num_samples, eps=None,
                  plan=None, generator=None, group=None):
    """lds.py:35-52.  Returns (samples, global_expected_stats, global_kl, local_kl); with B sequences,
    samples is (B,T,S,n) and the statistics / local_kl are sums over the (global) batch."""
    dev = torch.device("cuda", torch.cuda.current_device())
    g = (_dev64(global_natparam[0], dev), tuple(_dev64(x, dev) for x in global_natparam[1]))
    p = (_dev64(prior_natparam[0], dev), tuple(_dev64(x, dev) for x in prior_natparam[1]))
    local_natparam, global_es = local_natparam_from_global(g)
    node = tuple(_dev64(x, dev) for x in nn_potentials)
    batched = node[1].dim() == 3
    nodeb = node if batched else tuple(x[None] for x in node)
    B, T, n = nodeb[1].shape
    if plan is None:
        plan = LDSEStepPlan(B, T, n, dev)
    lognorm, (Ei, Ep, En) = natural_lds_estep_general(local_natparam, nodeb, plan=plan, keep_factor=True)
    S = 1 if num_samples is None else int(num_samples)
    if eps is None:
        eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev, generator=generator)
    else:
        eps = _dev64(eps, dev)
        eps = eps if batched else eps[None]
    samples = plan.sample(eps)
    # local KL: <nn_potentials, E_node> - lognorm  (lds.py:40), summed over the batch
    local_kl = (nodeb[0] * En[0]).sum() + (nodeb[1] * En[1]).sum() - lognorm.sum()
    if len(nodeb) == 3:
        local_kl = local_kl + nodeb[2].sum()
    # global statistics: deterministic device reduction, then the one collective (statistics + local KL)
    niw_stats, mniw_stats, local_kl = allreduce_lds_stats(plan.reduce(), local_kl, n, T, group)
    global_kl = lds_prior_kl(g, p, global_es)
    if not batched:
        samples = samples[0]
    return samples, (niw_stats, mniw_stats), global_kl, local_kl


def run_inference_differentiable(prior_natparam, global_natparam, nn_potentials, num_samples,
                                 eps=None, plan=None, generator=None, group=None):
    """run_inference with gradients w.r.t. nn_potentials = (J (B,T,n), h (B,T,n)[, logZ (B,T)]) flowing
    into `samples` and `local_kl` through the HIP VJP kernels (the reference differentiates exactly
    these two, svae.py:21-24; the statistics go to `saved.stats` undifferentiated)."""
    from ..lds.lds_inference import lds_inference_differentiable
    dev = nn_potentials[1].device
    g = (_dev64(global_natparam[0], dev), tuple(_dev64(x, dev) for x in global_natparam[1]))
    p = (_dev64(prior_natparam[0], dev), tuple(_dev64(x, dev) for x in prior_natparam[1]))
    local_natparam, global_es = local_natparam_from_global(g)
    node = tuple(x.to(torch.float64) for x in nn_potentials)
    batched = node[1].dim() == 3
    nodeb = node if batched else tuple(x[None] for x in node)
    B, T, n = nodeb[1].shape
    S = 1 if num_samples is None else int(num_samples)
    if eps is None:
        eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev, generator=generator)
    else:
        eps = _dev64(eps, dev)
        eps = eps if batched else eps[None]
    if plan is None:
        plan = LDSEStepPlan(B, T, n, dev)
    lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(local_natparam, nodeb, eps=eps, plan=plan)
    local_kl = (nodeb[0] * dxx).sum() + (nodeb[1] * ex).sum() - lognorm.sum()
    if len(nodeb) == 3:
        local_kl = local_kl + nodeb[2].sum()
    # one collective for statistics AND the local KL (as run_inference): the returned local_kl has the
    # global value and this rank's gradient
    niw_stats, mniw_stats, local_kl = allreduce_lds_stats(plan.reduce(), local_kl, n, T, group)
    global_kl = lds_prior_kl(g, p, global_es)
    if not batched:
        samples = samples[0]
    return samples, (niw_stats, mniw_stats), global_kl, local_kl


def make_prior_natparam(n, random=False, scaling=1., dtype=torch.float64, device="cpu"):
    """(/root/reference/svae/models/lds.py:57-67) -> (NIW natparam of the initial state, MNIW natparam
    of the dynamics) for an n-dimensional LDS; the values of the reference (nu = n+1,
    S = 2 scaling (n+1) I, mu = 0, kappa = 1/(2 scaling n), M = I, K = kappa I)."""
    if random:
        raise NotImplementedError
    eye = torch.eye(n, dtype=dtype, device=device)
    nu = torch.tensor(n + 1., dtype=dtype, device=device)
    S = 2. * scaling * (n + 1) * eye
    mu = torch.zeros(n, dtype=dtype, device=device)
    kappa = torch.tensor(1. / (2. * scaling * n), dtype=dtype, device=device)
    M, K = eye.clone(), 1. / (2. * scaling * n) * eye
    return expfam.niw_standard_to_natural(S, mu, kappa, nu), expfam.mniw_standard_to_natural(nu, S, M, K)
