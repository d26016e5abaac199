"""GMM-SVAE local inference on MI355X; mirrors /root/reference/svae/models/gmm.py.

  local_meanfield(global_natparam, node_potentials) -> (local_stats, prior_stats, natparam, kl)
                                                                        (gmm.py:62-88)
  run_inference(prior_natparam, global_natparam, nn_potentials, num_samples)   (gmm.py:12-16)

The fixed point (gmm.py:90-110) and the final pass run in ONE kernel launch of libsvae_hip.so
(svae_gmm_meanfield_f64); the two global->local maps (dirichlet / niw expectedstats, gmm.py:67-68)
are O(K N^2) torch ops.  Difference to the reference, by design: `initialize_meanfield`
(gmm.py:126-128) draws from the global NumPy RNG inside the loop; here the initial
responsibilities are an argument (`label_init`, default: drawn from a torch generator), so results
are reproducible and parity-testable.  No CPU fallback.
"""
import torch

from .. import _lib
from ..distributions import expfam

GMM_MAX_N, GMM_MAX_K = 8, 64


def _dev64(x, device):
    # dtype given up front: torch.as_tensor(python_float) alone would round to float32
    t = x.detach() if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)
    return t.to(device=device, dtype=torch.float64).contiguous()


def initialize_meanfield(T, K, device, generator=None):
    """normalize(rand(T, K)), gmm.py:126-128."""
    r = torch.rand(T, K, dtype=torch.float64, device=device, generator=generator)
    return r / r.sum(-1, keepdim=True)


# Dispatch by minibatch size (measured, tools/bench_gmm.py): one workgroup / one launch up to 512 points; several
# workgroups in ONE launch that exchange tagged KL partials per sweep up to 8192 points (the persistent kernel);
# beyond, one launch per sweep (16 us at 16 k points, 146 us at 1 M) -- which is also the form that shards over GPUs.
# `out["path"]` of meanfield_from_globals says which one ran ("single_wg" / "persistent" / "sweeps").
GMM_SINGLE_WG_MAX_T = 0      # (the one-workgroup kernel is kept for multi_wg=False: it no longer wins at any size)
GMM_PERSISTENT_MAX_T = 8192


def run_sweeps(backend, max_iter, group=None):
    """The sweep protocol of the multi-workgroup / multi-GPU fixed point (include/svae_hip.h, svae_gmm_mw_*):
    `backend` enqueues begin / sweep(i) / final / stats and exposes `kl_hist` (tensor, one total per sweep).
    Under torch.distributed (points sharded over the ranks of `group`) kl_hist[i] is all-reduced in place
    after sweep i, so that every rank applies the reference's stopping rule to the SAME batch-total KL
    (gmm.py:104-105).  No host synchronisation: sweeps after convergence are device-side no-ops."""
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    backend.begin()
    for i in range(max_iter):
        backend.sweep(i)
        if sharded:
            dist.all_reduce(backend.kl_hist[i:i + 1], group=group)
    backend.final()
    backend.stats()


class _HipSweeps(object):
    """run_sweeps backend on libsvae_hip.so (svae_gmm_mw_begin / svae_gmm_mw_step_f64)."""

    def __init__(self, lib, dims, tensors, out, tol, max_iter, dev):
        self.lib, self.dims, self.t, self.out = lib, dims, tensors, out
        self.tol, self.max_iter, self.dev = float(tol), int(max_iter), dev
        T, N, K = dims
        self.ws_bytes = int(lib.svae_gmm_mw_workspace_bytes(T, N, K, self.max_iter))
        self.ws = torch.empty((self.ws_bytes + 7) // 8, dtype=torch.float64, device=dev)
        self.kl_hist = self.ws[:self.max_iter + 1]

    def begin(self):
        T, N, K = self.dims
        _lib.check(self.lib.svae_gmm_mw_begin(T, N, K, self.max_iter, _lib.ptr(self.ws), self.ws_bytes,
                                              _lib.current_stream(self.dev)), "svae_gmm_mw_begin")

    def _step(self, phase, sweep):
        T, N, K = self.dims
        p, o = _lib.ptr, self.out
        lg, gg, nJ, nh, li = self.t
        rc = self.lib.svae_gmm_mw_step_f64(
            phase, sweep, T, N, K, p(lg), p(gg), p(nJ), p(nh), p(li), self.tol, self.max_iter,
            p(o["label_stats"]), p(o["label_fixed"]), p(o["gaussian_stats"]), p(o["label_natparam"]),
            p(o["gaussian_natparam"]), p(o["dirichlet_stats"]), p(o["niw_stats"]),
            p(o["kl"]), p(o["iters"]), p(o["assign"]), p(o["info"]), p(self.ws), self.ws_bytes,
            _lib.current_stream(self.dev))
        _lib.check(rc, "svae_gmm_mw_step_f64")

    def sweep(self, i):
        self._step(0, i)

    def final(self):
        self._step(1, 0)

    def stats(self):
        self._step(2, 0)


def meanfield_from_globals(label_global, gaussian_globals, node_potentials, label_init,
                           tol=1e-3, max_iter=100, check=True, multi_wg=None, group=None, persistent=True):
    """The kernel call: everything after gmm.py:68.  Returns a dict of device tensors.

    multi_wg=None picks the single-workgroup, single-launch kernel for small minibatches and the
    multi-workgroup sweeps (svae_gmm_mw_*) above GMM_SINGLE_WG_MAX_T points or when the points are sharded
    over the ranks of `group` (then T is this rank's share, the stopping rule runs on the all-reduced KL and
    `kl` / the statistics returned are this rank's: run_inference sums them)."""
    import torch.distributed as dist
    lib = _lib.load()
    dev = gaussian_globals.device if isinstance(gaussian_globals, torch.Tensor) and \
        gaussian_globals.is_cuda else torch.device("cuda", torch.cuda.current_device())
    lg, gg = _dev64(label_global, dev), _dev64(gaussian_globals, dev)
    nJ, nh = _dev64(node_potentials[0], dev), _dev64(node_potentials[1], dev)
    if nJ.dim() != 2 or nJ.shape != nh.shape:
        raise ValueError("node potentials must be (J diag (T,N), h (T,N))")
    T, N = nh.shape
    K = lg.shape[0]
    D = N + 2
    if tuple(gg.shape) != (K, D, D):
        raise ValueError("gaussian_globals must be (K, N+2, N+2)")
    if not (1 <= N <= GMM_MAX_N and 1 <= K <= GMM_MAX_K):
        raise ValueError("GMM kernel limits: N <= %d, K <= %d" % (GMM_MAX_N, GMM_MAX_K))
    li = _dev64(label_init, dev)
    if tuple(li.shape) != (T, K):
        raise ValueError("label_init must be (T, K)")
    f64 = dict(dtype=torch.float64, device=dev)
    out = dict(label_stats=torch.empty(T, K, **f64), label_fixed=torch.empty(T, K, **f64),
               gaussian_stats=torch.empty(T, D, D, **f64),
               label_natparam=torch.empty(T, K, **f64), gaussian_natparam=torch.empty(T, D, D, **f64),
               dirichlet_stats=torch.empty(K, **f64), niw_stats=torch.empty(K, D, D, **f64),
               kl=torch.empty(1, **f64), iters=torch.zeros(1, dtype=torch.int32, device=dev),
               assign=torch.empty(T, dtype=torch.int32, device=dev),
               info=torch.zeros(1, dtype=torch.int32, device=dev))
    p = _lib.ptr
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if multi_wg is None:
        multi_wg = sharded or T > GMM_SINGLE_WG_MAX_T
    if sharded and not multi_wg:
        raise ValueError("sharded points need the multi-workgroup sweeps (batch-total stopping rule)")
    if multi_wg:
        be = _HipSweeps(lib, (T, N, K), (lg, gg, nJ, nh, li), out, tol, max_iter, dev)
        want_persistent = not sharded and persistent and T <= GMM_PERSISTENT_MAX_T
        out["path"] = "sweeps"
        if want_persistent:
            # single GPU: ONE launch for the whole fixed point, the workgroups exchange tagged KL partials per sweep
            # (svae_gmm_mw_fixed_point_f64)
            rc = lib.svae_gmm_mw_fixed_point_f64(
                T, N, K, p(lg), p(gg), p(nJ), p(nh), p(li), float(tol), int(max_iter),
                p(out["label_stats"]), p(out["label_fixed"]), p(out["gaussian_stats"]), p(out["label_natparam"]),
                p(out["gaussian_natparam"]), p(out["dirichlet_stats"]), p(out["niw_stats"]),
                p(out["kl"]), p(out["iters"]), p(out["assign"]), p(out["info"]), p(be.ws), be.ws_bytes,
                _lib.current_stream(dev))
            if rc == -50:
                # never silently: the per-sweep launches are 1.5x slower at 1000 points (bench.py fails its GMM leg on it)
                import warnings
                warnings.warn("svae_gmm_mw_fixed_point_f64: device could not be queried (rc -50); "
                              "falling back to one launch per sweep", RuntimeWarning)
            else:
                _lib.check(rc, "svae_gmm_mw_fixed_point_f64")
                out["path"] = "persistent"
        if out["path"] == "sweeps":     # sharded points, large minibatches (or the fallback above): one launch per sweep
            run_sweeps(be, int(max_iter), group)
    else:
        out["path"] = "single_wg"
        rc = lib.svae_gmm_meanfield_f64(
            T, N, K, p(lg), p(gg), p(nJ), p(nh), p(li), float(tol), int(max_iter),
            p(out["label_stats"]), p(out["label_fixed"]), p(out["gaussian_stats"]), p(out["label_natparam"]),
            p(out["gaussian_natparam"]), p(out["dirichlet_stats"]), p(out["niw_stats"]),
            p(out["kl"]), p(out["iters"]), p(out["assign"]), p(out["info"]), _lib.current_stream(dev))
        _lib.check(rc, "svae_gmm_meanfield_f64")
    global last_info
    last_info = out["info"]
    if check:
        check_info(out["info"])
    return out


last_info = None     # status word of the most recent fixed point (device tensor; see check_info)


def check_info(info=None):
    """Read a fixed point's status word (default: the most recent call's) -- a host synchronisation -- and raise what
    it reports."""
    info = last_info if info is None else info
    if info is None:
        return
    v = int(info.item())
    if v < 0:
        raise RuntimeError("GMM mean field: a workgroup never received a partner's KL partial (info %d): "
                           "the grid of the persistent kernel was not co-resident" % v)
    if v != 0:
        raise FloatingPointError("GMM mean field: point %d has a non positive definite "
                                 "Gaussian factor" % (v - 1))


def global_step(global_natparam, prior_natparam=None, info=None, reference_compat=True):
    """The global side of a step in ONE launch (svae_gmm_global_step_f64): (label_global (K), gaussian_globals
    (K,N+2,N+2)) = (dirichlet.expectedstats, niw.expectedstats) of gmm.py:67-68 and, with `prior_natparam`, the prior KL
    of gmm.py:54-58 -- `reference_compat=True` (default): the value the reference AS SHIPPED returns (its `flat`,
    util.py:39 after util.py:166, keeps the first scalar: the contraction is its first term); False: the full
    contraction the code spells.  The kernel writes both.  -> (label_global, gaussian_globals, kl | None).  `info`:
    (1,) int32 status word, raised to 1 on an invalid NIW scale matrix (kept as global_step.last_info, never read here)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    for x in global_natparam:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            dev = x.device
    dn, nn_ = _dev64(global_natparam[0], dev), _dev64(global_natparam[1], dev)
    K, N = dn.shape[0], nn_.shape[-1] - 2
    if tuple(nn_.shape) != (K, N + 2, N + 2) or not (1 <= N <= GMM_MAX_N and 1 <= K <= GMM_MAX_K):
        raise ValueError("GMM global parameters: dirichlet (K), NIW (K, N+2, N+2) with N <= %d, K <= %d" % (GMM_MAX_N, GMM_MAX_K))
    f64 = dict(dtype=torch.float64, device=dev)
    lg, gg = torch.empty(K, **f64), torch.empty(K, N + 2, N + 2, **f64)
    kl, pd, pn = None, None, None
    if prior_natparam is not None:
        pd, pn = _dev64(prior_natparam[0], dev), _dev64(prior_natparam[1], dev)
        kl = torch.empty(2, **f64)
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=dev)
    global_step.last_info = info
    p = _lib.ptr
    _lib.check(_lib.load().svae_gmm_global_step_f64(K, N, p(dn), p(nn_), p(pd), p(pn), p(lg), p(gg), p(kl), p(info),
                                                    _lib.current_stream(dev)), "svae_gmm_global_step_f64")
    return lg, gg, (kl[1 if reference_compat else 0] if kl is not None else None)


def local_meanfield(global_natparam, node_potentials, label_init=None, tol=1e-3, max_iter=100,
                    generator=None, group=None, multi_wg=None):
    """gmm.py:62-88 -> (local_stats, prior_stats, natparam, kl).  With the points sharded over the ranks of
    `group`: the fixed point stops on the all-reduced batch-total KL like the reference's single process;
    prior_stats and kl are this rank's share."""
    dirichlet_natparam, niw_natparams = global_natparam
    dev = torch.device("cuda", torch.cuda.current_device())
    for x in (niw_natparams, node_potentials[0]):
        if isinstance(x, torch.Tensor) and x.is_cuda:
            dev = x.device
    dn, nn_ = _dev64(dirichlet_natparam, dev), _dev64(niw_natparams, dev)
    label_global, gaussian_globals, _ = global_step((dn, nn_))  # gmm.py:67-68
    T = node_potentials[1].shape[0]
    if label_init is None:
        label_init = initialize_meanfield(T, dn.shape[0], dev, generator)
    o = meanfield_from_globals(label_global, gaussian_globals, node_potentials, label_init,
                               tol, max_iter, multi_wg=multi_wg, group=group)
    local_stats = o["label_stats"], o["gaussian_stats"]
    prior_stats = o["dirichlet_stats"], o["niw_stats"]
    natparam = o["label_natparam"], o["gaussian_natparam"]
    return local_stats, prior_stats, natparam, o["kl"][0]


def prior_kl(global_natparam, prior_natparam, reference_compat=True):
    """gmm.py:54-58: KL(q(theta) || p(theta)) of the Dirichlet x NIW^K global factors.

    `reference_compat=False`: the full contraction <eta_q - eta_p, E_q[t(theta)]> - (logZ(q) - logZ(p)) the code spells.
    `reference_compat=True` (the default since round 5) returns what the reference AS SHIPPED computes: svae/util.py:169 rebinds
    `flatten`, so `flat` (util.py:42) keeps only the FIRST scalar of each nested structure and the
    contraction reduces to its first element (checked against the reference run in memory,
    tests/golden/gmm_run_K5_N2_T60.npz `global_kl`)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    g = [_dev64(x, dev) for x in global_natparam]
    p = [_dev64(x, dev) for x in prior_natparam]
    es = (expfam.dirichlet_expectedstats(g[0]), expfam.niw_expectedstats(g[1]))
    if reference_compat:
        diff = (g[0].reshape(-1)[0] - p[0].reshape(-1)[0]) * es[0].reshape(-1)[0]
    else:
        diff = sum(((a - b) * e).sum() for a, b, e in zip(g, p, es))
    logZ = lambda q: expfam.dirichlet_logZ(q[0]) + expfam.niw_logZ(q[1])
    return diff - (logZ(g) - logZ(p))


def _allreduce_stats_and_kl(stats, local_kl, group):
    """Sharded data points: ONE all-reduce of [dirichlet_stats | niw_stats | local_kl] (as the LDS model
    does, svae_amd/parallel.py); the returned local_kl carries the global value and, if it is on the
    autograd tape, this rank's gradient."""
    from ..parallel import allreduce_global_stats
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return stats, local_kl
    ds, ns = stats
    packed = torch.cat([ds.reshape(-1), ns.reshape(-1), local_kl.detach().reshape(1)])
    allreduce_global_stats(packed, group)
    k = ds.numel()
    return (packed[:k].reshape(ds.shape), packed[k:-1].reshape(ns.shape)), local_kl + (packed[-1] - local_kl.detach())


def run_inference(prior_natparam, global_natparam, nn_potentials, num_samples, label_init=None,
                  eps=None, generator=None, group=None, reference_compat=True, check=True):
    """gmm.py:12-16 -> (samples (T,S,N), (dirichlet_stats, niw_stats), global_kl, local_kl).  Under
    torch.distributed the points are this rank's shard; statistics and local_kl are summed over ranks.
    check=False: no host synchronisation (hipGraph capture, asynchronous pipelines) -- the fixed point's status word
    stays on the device as `gmm.last_info`; `gmm.check_info()` reads and raises it when the caller chooses to."""
    dev = torch.device("cuda", torch.cuda.current_device())
    for x in (global_natparam[1], nn_potentials[0]):
        if isinstance(x, torch.Tensor) and x.is_cuda:
            dev = x.device
    label_global, gaussian_globals, global_kl = global_step(global_natparam, prior_natparam,
                                                            reference_compat=reference_compat)
    Tn = nn_potentials[1].shape[0]
    if label_init is None:
        label_init = initialize_meanfield(Tn, label_global.shape[0], dev, generator)
    o = meanfield_from_globals(label_global, gaussian_globals, nn_potentials, label_init, group=group, check=check)
    stats, local_natparam, local_kl = (o["dirichlet_stats"], o["niw_stats"]), (o["label_natparam"], o["gaussian_natparam"]), o["kl"][0]
    gn = local_natparam[1]
    T, N = gn.shape[0], gn.shape[-1] - 2
    if eps is None:
        eps = torch.randn(T, num_samples, N, dtype=torch.float64, device=gn.device, generator=generator)
    samples = gaussian_sample(gn, eps)
    stats, local_kl = _allreduce_stats_and_kl(stats, local_kl, group)
    return samples, stats, global_kl, local_kl


# --- differentiable call surface (what make_gradfun drives) ------------------------------------------

def gaussian_sample(gaussian_natparam, eps):
    """gaussian.py:27-33 on the device: x = J^-1 h + chol(J)^-T eps for the (T, N+2, N+2) dense-packed factors the
    fixed-point kernels return; eps (T,S,N) -> samples (T,S,N)  (svae_gmm_sample_f64)."""
    lib = _lib.load()
    gn = gaussian_natparam.contiguous()
    T, N = gn.shape[0], gn.shape[-1] - 2
    eps = _dev64(eps, gn.device)
    S = eps.shape[1]
    if tuple(eps.shape) != (T, S, N):
        raise ValueError("eps must be (T, S, N)")
    out = torch.empty(T, S, N, dtype=torch.float64, device=gn.device)
    _lib.check(lib.svae_gmm_sample_f64(T, N, S, _lib.ptr(gn), _lib.ptr(eps), _lib.ptr(out),
                                       _lib.current_stream(gn.device)), "svae_gmm_sample_f64")
    return out


class _LocalTail(torch.autograd.Function):
    """(node_J, node_h) -> (samples, local_kl): the one pass of gmm.py:74-86 the reference keeps on the autograd tape
    + gaussian.natural_sample.  Forward values are the fixed-point kernel's final pass (`o`, on detached potentials)
    and svae_gmm_sample_f64; backward is svae_gmm_local_vjp_f64 (the adjoint derived in csrc/gmm_train.hip)."""

    @staticmethod
    def forward(ctx, node_J, node_h, eps, label_global, gaussian_globals, o):
        samples = gaussian_sample(o["gaussian_natparam"], eps)
        ctx.save_for_backward(node_J.detach().contiguous(), node_h.detach().contiguous(), eps, label_global,
                              gaussian_globals, o["gaussian_natparam"], o["label_natparam"])
        return samples, o["kl"][0].clone()

    @staticmethod
    def backward(ctx, g_samples, g_kl):
        nJ, nh, eps, lg, gg, gn, ln = ctx.saved_tensors
        lib = _lib.load()
        T, N = nh.shape
        K, S = lg.shape[0], eps.shape[1]
        gJ, gh = torch.empty_like(nJ), torch.empty_like(nh)
        p = _lib.ptr
        gs = None if g_samples is None else g_samples.to(torch.float64).contiguous()
        gk = None if g_kl is None else g_kl.to(torch.float64).reshape(1).contiguous()
        _lib.check(lib.svae_gmm_local_vjp_f64(T, N, K, S, p(lg), p(gg), p(nJ), p(nh), p(gn), p(ln), p(gk), p(eps),
                                              p(gs), p(gJ), p(gh), _lib.current_stream(nh.device)),
                   "svae_gmm_local_vjp_f64")
        return gJ, gh, None, None, None, None


def run_inference_differentiable(prior_natparam, global_natparam, nn_potentials, num_samples,
                                 label_init=None, eps=None, generator=None, group=None,
                                 reference_compat=True, check=True):
    """run_inference (gmm.py:12-16) with gradients w.r.t. nn_potentials flowing into `samples` and
    `local_kl`, exactly the two quantities the reference differentiates (svae.py:21-24); statistics
    are returned detached (`unbox(stats)`, gmm.py:16).  Kernel launches only: fixed point + final pass, sampler,
    and -- in backward() -- the derived adjoint of the final pass and the sampler.  check=False: as in run_inference
    (the whole step then captures into ONE hipGraph: bench.py extra[8])."""
    dev = nn_potentials[1].device
    g = [_dev64(x, dev) for x in global_natparam]
    label_global, gaussian_globals, global_kl = global_step(g, prior_natparam, reference_compat=reference_compat)
    nJ, nh = nn_potentials[0].to(torch.float64), nn_potentials[1].to(torch.float64)
    T, N = nh.shape
    if label_init is None:
        label_init = initialize_meanfield(T, g[0].shape[0], dev, generator)
    o = meanfield_from_globals(label_global, gaussian_globals, (nJ.detach(), nh.detach()), label_init, group=group,
                               check=check)
    if eps is None:
        eps = torch.randn(T, num_samples, N, dtype=torch.float64, device=dev, generator=generator)
    samples, local_kl = _LocalTail.apply(nJ, nh, _dev64(eps, dev), label_global.contiguous(),
                                         gaussian_globals.contiguous(), o)
    stats, local_kl = _allreduce_stats_and_kl((o["dirichlet_stats"], o["niw_stats"]), local_kl, group)
    return samples, stats, global_kl, local_kl


def init_pgm_param(K, N, alpha, niw_conc=10., random_scale=0., generator=None, dtype=torch.float64, device="cpu"):
    """(/root/reference/svae/models/gmm.py:33-42) -> (dirichlet natparam (K), NIW natparams (K, N+2, N+2)).
    The reference draws from the global NumPy RNG; here from `generator`."""
    kw = dict(dtype=dtype, device=device)
    nu = torch.tensor(N + niw_conc, **kw)
    S = (N + niw_conc) * torch.eye(N, **kw)
    kappa = torch.tensor(float(niw_conc), **kw)
    niws = []
    for _ in range(K):
        m = torch.zeros(N, **kw)
        if random_scale:
            m = m + random_scale * torch.randn(N, generator=generator, **kw)
        niws.append(expfam.niw_standard_to_natural(S, m, kappa, nu))
    dirichlet = alpha * (torch.rand(K, generator=generator, **kw) if random_scale else torch.ones(K, **kw))
    return dirichlet, torch.stack(niws)


def prior_logZ(gmm_natparam):
    """(gmm.py:44-46)"""
    dirichlet_natparam, niw_natparams = gmm_natparam
    return expfam.dirichlet_logZ(dirichlet_natparam) + expfam.niw_logZ(niw_natparams).sum()


def prior_expectedstats(gmm_natparam):
    """(gmm.py:48-52)"""
    dirichlet_natparam, niw_natparams = gmm_natparam
    return expfam.dirichlet_expectedstats(dirichlet_natparam), expfam.niw_expectedstats(niw_natparams)
