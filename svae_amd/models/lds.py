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

from .. import _lib
from ..distributions import expfam
from ..lds.lds_inference import LDSEStepPlan, natural_lds_estep_general, natural_lds_inference_general, reduce_stats
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


GLOBAL_STEP_MAX_N = 64


def global_step(global_natparam, prior_natparam=None, info=None):
    """The once-per-step global side in ONE kernel launch (svae_lds_global_step_f64): the LDS potentials from
    the global factors (lds.py:23-25 = niw.expectedstats niw.py:15-25 + mniw.expectedstats mniw.py:33-55) and,
    if `prior_natparam` is given, the prior KL of lds.py:16-20.  Inputs are device tensors
    (NIW dense (n+2,n+2), (A, B, C, d)).  Returns ((init_params, pair_params), global_kl | None, niw_expectedstats)
    with init_params = (-1/2 E[J], E[h], logZ) and pair_params = (J11, J12, J22, logZ) as the E-step takes them.
    Same values as local_natparam_from_global / lds_prior_kl (the torch path, ~150 small launches).
    `info`: (1,) int32 device status word; the kernel raises it to 1 on a non-positive Gauss-Jordan pivot, i.e. global
    natural parameters that are not valid (the reference asserts is_posdef in mniw.expectedstats, mniw.py:49-50).
    Without one a fresh word is allocated and kept as `global_step.last_info` -- never read here: checking costs a
    host synchronisation; run_inference passes the plan's word, so that plan.check_info() reports it."""
    niw, (A, B, C, d) = global_natparam
    dev = niw.device
    n = niw.shape[-1] - 2
    if not (1 <= n <= GLOBAL_STEP_MAX_N):
        raise ValueError("global_step: latent dimension %d outside 1..%d" % (n, GLOBAL_STEP_MAX_N))
    f64 = dict(dtype=torch.float64, device=dev)
    c = lambda x: torch.as_tensor(x, **f64).contiguous()
    niw, A, B, C, d = c(niw), c(A), c(B), c(C), c(d).reshape(1)
    pr = [None] * 5
    if prior_natparam is not None:
        pn, (pA, pB, pC, pd) = prior_natparam
        pr = [c(pn), c(pA), c(pB), c(pC), c(pd).reshape(1)]
    out = torch.empty(4 * n * n + n + 3 + (n + 2) * (n + 2), **f64)      # one allocation for every output
    o = [0]

    def take(k, shape):
        v = out[o[0]:o[0] + k].view(shape)
        o[0] += k
        return v
    init_J, init_h, init_logZ = take(n * n, (n, n)), take(n, (n,)), take(1, (1,))
    J11, J12, J22, lz = take(n * n, (n, n)), take(n * n, (n, n)), take(n * n, (n, n)), take(1, (1,))
    kl, es = take(1, (1,)), take((n + 2) * (n + 2), (n + 2, n + 2))
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=dev)
    global_step.last_info = info
    p = _lib.ptr
    rc = _lib.load().svae_lds_global_step_f64(
        n, p(niw), p(A), p(B), p(C), p(d), p(pr[0]), p(pr[1]), p(pr[2]), p(pr[3]), p(pr[4]),
        p(init_J), p(init_h), p(init_logZ), p(J11), p(J12), p(J22), p(lz), p(es),
        p(kl) if prior_natparam is not None else None, p(info), _lib.current_stream(dev))
    _lib.check(rc, "svae_lds_global_step_f64")
    return ((init_J, init_h, init_logZ), (J11, J12, J22, lz)), (kl[0] if prior_natparam is not None else None), es


class PackedLDSStats(tuple):
    """(niw_stats, mniw_stats) as the reference returns them, carrying the packed all-reduced buffer of
    svae_lds_reduce_stats_f64 they were unpacked from (`.packed`, `.T`): the natural-gradient kernel reads it."""
    packed = None
    T = None


def natural_gradient(prior_natparam, global_natparam, stats, num_batches, scale):
    """svae.py:33-34 for the LDS global parameter in one launch (svae_lds_natgrad_f64):
    -scale * (prior + num_batches * stats - params), nested like the parameters.  `stats` must be the
    PackedLDSStats run_inference returns."""
    dev = stats.packed.device
    flatten = lambda q: torch.cat([torch.as_tensor(x, dtype=torch.float64, device=dev).reshape(-1)
                                   for x in (q[0],) + tuple(q[1])])
    prior, params = flatten(prior_natparam), flatten(global_natparam)
    n = stats[0].shape[-1] - 2
    out = torch.empty_like(params)
    p = _lib.ptr
    rc = _lib.load().svae_lds_natgrad_f64(n, int(stats.T), p(stats.packed), p(prior), p(params), float(num_batches),
                                          float(scale), p(out), _lib.current_stream(dev))
    _lib.check(rc, "svae_lds_natgrad_f64")
    D2, nn = (n + 2) * (n + 2), n * n
    return out[:D2].view(n + 2, n + 2), (out[D2:D2 + nn].view(n, n), out[D2 + nn:D2 + 2 * nn].view(n, n),
                                          out[D2 + 2 * nn:D2 + 3 * nn].view(n, n), out[D2 + 3 * nn])


def _globals_on_device(prior_natparam, global_natparam, dev, info=None):
    """(local_natparam, global_kl) through the global-step kernel (n <= 64), else the torch maps."""
    g = (_dev64(global_natparam[0], dev), tuple(_dev64(x, dev) for x in global_natparam[1]))
    p = (_dev64(prior_natparam[0], dev), tuple(_dev64(x, dev) for x in prior_natparam[1]))
    if g[0].shape[-1] - 2 <= GLOBAL_STEP_MAX_N:
        local_natparam, global_kl, _ = global_step(g, p, info)
        return local_natparam, global_kl
    local_natparam, global_es = local_natparam_from_global(g)
    return local_natparam, lds_prior_kl(g, p, global_es)


def run_inference(prior_natparam, global_natparam, nn_potentials, num_samples, eps=None,
                  plan=None, generator=None, group=None):
    """lds.py:35-52.  Returns (samples, global_expected_stats, global_kl, local_kl); with B sequences,
    samples is (B,T,S,n) and the statistics / local_kl are sums over the (global) batch."""
    dev = torch.device("cuda", torch.cuda.current_device())
    node = tuple(_dev64(x, dev) for x in nn_potentials)
    batched = node[1].dim() == 3
    nodeb = node if batched else tuple(x[None] for x in node)
    B, T, n = nodeb[1].shape
    if plan is None:
        plan = LDSEStepPlan(B, T, n, dev)
    # (invalid global parameters raise the PLAN's status word: plan.check_info() / check=True report them)
    local_natparam, global_kl = _globals_on_device(prior_natparam, global_natparam, dev, plan.info)
    S = 1 if num_samples is None else int(num_samples)
    if eps is None:
        eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev, generator=generator)
    else:
        eps = _dev64(eps, dev)
        eps = eps if batched else eps[None]
    # E-step + sampler as the reference's composite (cython_natural_lds_inference_general, lds_inference.py:196-202): one
    # library call for n <= 15 (lean per-step records above 1024 sequences)
    samples, (Ei, Ep, En), lognorm = natural_lds_inference_general(local_natparam, nodeb, num_samples=S, eps=eps, plan=plan)
    # local KL: <nn_potentials, E_node> - lognorm  (lds.py:40), summed over the batch
    local_kl = (nodeb[0] * En[0]).sum() + (nodeb[1] * En[1]).sum() - lognorm.sum()
    if len(nodeb) == 3:
        local_kl = local_kl + nodeb[2].sum()
    # global statistics: deterministic device reduction, then the one collective (statistics + local KL)
    stats, local_kl = _exchange(plan, local_kl, n, T, group)
    if not batched:
        samples = samples[0]
    return samples, stats, global_kl, local_kl


def _exchange(plan, local_kl, n, T, group):
    """Batch reduction on the device, the ONE collective, and the statistics as the reference nests them
    (carrying the packed buffer for natural_gradient)."""
    niw_stats, mniw_stats, local_kl, packed = allreduce_lds_stats(plan.reduce(), local_kl, n, T, group, return_packed=True)
    stats = PackedLDSStats((niw_stats, mniw_stats))
    stats.packed, stats.T = packed, T
    return stats, local_kl


def run_inference_differentiable(prior_natparam, global_natparam, nn_potentials, num_samples,
                                 eps=None, plan=None, generator=None, group=None):
    """run_inference with gradients w.r.t. nn_potentials = (J (B,T,n), h (B,T,n)[, logZ (B,T)]) flowing
    into `samples` and `local_kl` through the HIP VJP kernels (the reference differentiates exactly
    these two, svae.py:21-24; the statistics go to `saved.stats` undifferentiated)."""
    from ..lds.lds_inference import lds_inference_differentiable
    dev = nn_potentials[1].device
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
    local_natparam, global_kl = _globals_on_device(prior_natparam, global_natparam, dev, plan.info)
    lognorm, (dxx, ex), samples, _ = lds_inference_differentiable(local_natparam, nodeb, eps=eps, plan=plan)
    local_kl = (nodeb[0] * dxx).sum() + (nodeb[1] * ex).sum() - lognorm.sum()
    if len(nodeb) == 3:
        local_kl = local_kl + nodeb[2].sum()
    # one collective for statistics AND the local KL (as run_inference): the returned local_kl has the
    # global value and this rank's gradient
    stats, local_kl = _exchange(plan, local_kl, n, T, group)
    if not batched:
        samples = samples[0]
    return samples, stats, global_kl, local_kl


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
