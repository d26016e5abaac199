"""Host-side mirror of the reference's LDS E-step wrappers, running on MI355X.

Mirrors /root/reference/svae/lds/lds_inference.py:
  natural_lds_estep_general(natparam, node_params) -> (lognorm, expected_stats)   (:223-229)
  cython_natural_lds_estep_general                   (same contract, :232-237)
with the same argument meaning (NATURAL parameters, time-major, float64) and the same shape
checks / ValueError behaviour as `_canonical_node_params` (:65-82) and `_canonical_init_params`
(:62-63).  New relative to the reference: node potentials may carry a leading batch axis
(B, T, n) -- B conditionally independent sequences sharing (init, pair) parameters -- in which case
every output gains a leading B axis ("sum of per-sequence stats" semantics
are obtained with `reduce_stats`).

All arithmetic happens in libsvae_hip.so (svae_lds_estep_f64); this file only validates, lays out
buffers and launches.  No CPU fallback.
"""
import torch

from .. import _lib

__all__ = ["LDSEStepPlan", "natural_lds_estep_general", "cython_natural_lds_estep_general",
           "natural_lds_inference_general", "cython_natural_lds_inference_general", "reduce_stats"]


def _as_dev(x, device):
    if isinstance(x, torch.Tensor):
        t = x.detach()
    else:
        t = torch.as_tensor(x)
    return t.to(device=device, dtype=torch.float64).contiguous()


def _canonical_init_params(init_params, device):
    """lds_inference.py:62-63: (J, h, sum of the remaining log-normaliser terms)."""
    J, h = _as_dev(init_params[0], device), _as_dev(init_params[1], device)
    logZ = sum(_as_dev(z, device).reshape(()) for z in init_params[2:]) \
        if len(init_params) > 2 else torch.zeros((), dtype=torch.float64, device=device)
    return J, h, logZ.reshape(1).contiguous()


class LDSEStepPlan(object):
    """Pre-allocated buffers for repeated E-steps of one shape (B, T, n): the launch itself does no
    allocation, no host<->device copy and no synchronisation."""

    def __init__(self, B, T, n, device="cuda", inhomog=False):
        if not (1 <= n <= _lib.LDS_MAX_N):
            raise ValueError("latent dimension n=%d outside the register path (1..%d)"
                             % (n, _lib.LDS_MAX_N))
        if T < 1 or B < 0:
            raise ValueError("need T >= 1 and B >= 0")
        self.lib = _lib.load()
        self.B, self.T, self.n, self.inhomog = B, T, n, bool(inhomog)
        self.device = torch.device(device)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.ws_bytes = int(self.lib.svae_lds_workspace_bytes(max(B, 1), T, n))
        self.ws = torch.empty(self.ws_bytes // 8, **f64)
        self.lognorm = torch.empty(B, **f64)
        self.E_init = torch.empty(B, n * n + n, **f64)
        if self.inhomog:
            self.E_pair = torch.empty(B, max(T - 1, 0), 3, n, n, **f64)
        else:
            self.E_pair = torch.empty(B, 3, n, n, **f64)
        self.E_node_diagxx = torch.empty(B, T, n, **f64)
        self.E_node_x = torch.empty(B, T, n, **f64)
        self.info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.reduced = torch.empty(4 * n * n + n + 2, **f64)

    def launch(self, init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h,
               node_logZ=None, pair_batched=False, keep_factor=False):
        """Raw launch on the current stream.  All arguments: contiguous float64 device tensors."""
        p = _lib.ptr
        rc = self.lib.svae_lds_estep_f64(
            self.B, self.T, self.n, int(self.inhomog), int(pair_batched), int(keep_factor),
            p(init_J), p(init_h), p(init_logZ), p(J11), p(J12), p(J22), p(logZ_pair),
            p(node_J), p(node_h), p(node_logZ),
            p(self.lognorm), p(self.E_init), p(self.E_pair), p(self.E_node_diagxx),
            p(self.E_node_x), p(self.info), p(self.ws), self.ws_bytes,
            _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_estep_f64")
        self.has_factor = bool(keep_factor)

    def sample(self, eps, out=None):
        """Backward sampling from the messages of the last `launch(..., keep_factor=True)`.
        eps: (B,T,S,n) standard-normal draws -> samples (B,T,S,n)
        [natural_sample_backward, cython_lds_inference.pyx:310-355]."""
        if not getattr(self, "has_factor", False):
            raise RuntimeError("sample() needs a preceding launch(..., keep_factor=True)")
        if eps.dim() != 4 or eps.shape[0] != self.B or eps.shape[1] != self.T or eps.shape[3] != self.n:
            raise ValueError("eps must be (B,T,S,n)")
        eps = eps.to(device=self.device, dtype=torch.float64).contiguous()
        S = eps.shape[2]
        if out is None:
            out = torch.empty_like(eps)
        p = _lib.ptr
        rc = self.lib.svae_lds_sample_f64(self.B, self.T, self.n, S, p(eps), p(out), p(self.ws),
                                          self.ws_bytes, _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_sample_f64")
        return out

    def reduce(self):
        """Deterministic batch sums [sum E_init | sum E_pair | sum lognorm | B] (homogeneous)."""
        p = _lib.ptr
        rc = self.lib.svae_lds_reduce_stats_f64(
            self.B, self.n, p(self.E_init), p(self.E_pair), p(self.lognorm), p(self.reduced),
            _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_reduce_stats_f64")
        return self.reduced

    def check_info(self):
        """Synchronising check of the device-side status word (the reference never checks LAPACK
        `info`, cython_gaussian_grads.pxd:54-76; we do, on request)."""
        v = int(self.info.item())
        if v != 0:
            self.info.zero_()
            raise FloatingPointError("LDS E-step: sequence %d hit a non-positive pivot "
                                     "(potentials not positive definite)" % (v - 1))


def natural_lds_estep_general(natparam, node_params, plan=None, check=True, keep_factor=False):
    """E-step = filter + smoother (lds_inference.py:223-237).

    natparam = (init_params, pair_params); init_params = (-1/2 J0, h0, logZ...) and
    pair_params = (J11, J12, J22, logZ) homogeneous (n,n) or per-step (T-1,n,n)
    [or (B,T-1,n,n) with batched nodes]; node_params = (J, h[, logZ]) with diagonal J of shape
    (T,n) or (B,T,n).

    Returns (lognorm, (E_init_stats, E_pair_stats, E_node_stats)) shaped like the reference's
    (cython_lds_inference.pyx:197-210); with a batch axis first when the nodes are batched.
    """
    init_params, pair_params = natparam
    if not isinstance(node_params, (tuple, list)) or len(node_params) not in (2, 3):
        raise ValueError("node_params must be (J, h) or (J, h, logZ)")
    dev = None
    for x in list(node_params) + list(init_params[:2]):
        if isinstance(x, torch.Tensor) and x.is_cuda:
            dev = x.device
            break
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    node_J, node_h = _as_dev(node_params[0], dev), _as_dev(node_params[1], dev)
    node_logZ = _as_dev(node_params[2], dev) if len(node_params) == 3 else None
    if node_J.dim() == 3 and node_h.dim() == 2:
        raise ValueError("dense (T,n,n) node potentials are not supported by the compiled path "
                         "(as in the reference, cython_lds_inference.pyx:43)")
    batched = node_h.dim() == 3
    if node_J.shape != node_h.shape or node_h.dim() not in (2, 3):
        raise ValueError("node potentials must both be (T,n) or (B,T,n)")
    if not batched:
        node_J, node_h = node_J[None], node_h[None]
        node_logZ = None if node_logZ is None else node_logZ[None]
    B, T, n = node_h.shape
    if node_logZ is not None and tuple(node_logZ.shape) != (B, T):
        raise ValueError("node logZ must be (T,) / (B,T)")

    init_J, init_h, init_logZ = _canonical_init_params(init_params, dev)
    if tuple(init_J.shape) != (n, n) or tuple(init_h.shape) != (n,):
        raise ValueError("init_params shapes do not match the node potentials")
    J11, J12, J22 = (_as_dev(x, dev) for x in pair_params[:3])
    logZ_pair = _as_dev(pair_params[3], dev).reshape(-1)
    inhomog = J11.dim() >= 3
    pair_batched = J11.dim() == 4
    want = {2: (n, n), 3: (T - 1, n, n), 4: (B, T - 1, n, n)}.get(J11.dim())
    if want is None or any(tuple(x.shape) != want for x in (J11, J12, J22)):
        raise ValueError("pair_params must be (n,n), (T-1,n,n) or (B,T-1,n,n)")
    if inhomog and logZ_pair.numel() != (B * (T - 1) if pair_batched else T - 1):
        raise ValueError("pair logZ must have one entry per step")

    if plan is None:
        plan = LDSEStepPlan(B, T, n, dev, inhomog)
    elif (plan.B, plan.T, plan.n, plan.inhomog) != (B, T, n, inhomog):
        raise ValueError("plan shape mismatch")
    plan.launch(init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ,
                pair_batched, keep_factor)
    if check:
        plan.check_info()

    one = torch.ones((), dtype=torch.float64, device=dev)
    ExxT0 = plan.E_init[:, :n * n].reshape(B, n, n)
    Ex0 = plan.E_init[:, n * n:]
    if inhomog:
        Ep = (plan.E_pair[:, :, 0], plan.E_pair[:, :, 1], plan.E_pair[:, :, 2],
              torch.ones(B, T - 1, dtype=torch.float64, device=dev))
    else:
        Ep = (plan.E_pair[:, 0], plan.E_pair[:, 1], plan.E_pair[:, 2],
              torch.full((B,), float(T - 1), dtype=torch.float64, device=dev))
    En = (plan.E_node_diagxx, plan.E_node_x, torch.ones(B, T, dtype=torch.float64, device=dev))
    Ei = (ExxT0, Ex0, one.expand(B), one.expand(B))
    lognorm = plan.lognorm
    if not batched:
        sq = lambda tup: tuple(x[0] for x in tup)
        return lognorm[0], (sq(Ei), sq(Ep), sq(En))
    return lognorm, (Ei, Ep, En)


cython_natural_lds_estep_general = natural_lds_estep_general


def natural_lds_inference_general(natparam, node_params, num_samples=None, eps=None, plan=None,
                                  generator=None):
    """E-step + backward sampling: (samples, expected_stats, lognorm), mirroring
    `cython_natural_lds_inference_general` (lds_inference.py:196-202).  samples: (T,S,n), or (T,n) when
    num_samples is None as in the Python path (:109-124); batched nodes add a leading B axis.
    The reference draws its noise from the global NumPy RNG inside the sampler
    (cython_lds_inference.pyx:333); here `eps` (B,T,S,n) / (T,S,n) may be passed in, else it is drawn
    from `generator` on the device."""
    batched = (node_params[1].ndim if hasattr(node_params[1], "ndim") else torch.as_tensor(node_params[1]).dim()) == 3
    S = 1 if num_samples is None else int(num_samples)
    if plan is None:
        nh = torch.as_tensor(node_params[1])
        B, T, n = (nh.shape if batched else (1,) + tuple(nh.shape))
        inhomog = torch.as_tensor(natparam[1][0]).dim() >= 3
        plan = LDSEStepPlan(B, T, n, "cuda", inhomog)
    lognorm, stats = natural_lds_estep_general(natparam, node_params, plan=plan, keep_factor=True)
    if eps is None:
        eps = torch.randn(plan.B, plan.T, S, plan.n, dtype=torch.float64, device=plan.device,
                          generator=generator)
    else:
        eps = torch.as_tensor(eps, dtype=torch.float64)
        eps = eps if batched else eps[None]
    samples = plan.sample(eps)
    if num_samples is None:
        samples = samples[:, :, 0]
    if not batched:
        samples = samples[0]
    return samples, stats, lognorm


cython_natural_lds_inference_general = natural_lds_inference_general


def reduce_stats(plan):
    """Sum over the batch of the global statistics, unpacked like the reference tuples:
    ((sum ExxT0, sum Ex0, B, B), (sum E_pair[0..2], B*(T-1)), sum lognorm).  This is the buffer
    all-reduced across GPUs before the natural-gradient step (svae.py:33-34)."""
    n, B, T = plan.n, plan.B, plan.T
    r = plan.reduce()
    nn = n * n
    Ei = (r[:nn].reshape(n, n), r[nn:nn + n], float(B), float(B))
    o = nn + n
    Ep = (r[o:o + nn].reshape(n, n), r[o + nn:o + 2 * nn].reshape(n, n),
          r[o + 2 * nn:o + 3 * nn].reshape(n, n), float(B * (T - 1)))
    return Ei, Ep, r[o + 3 * nn]
