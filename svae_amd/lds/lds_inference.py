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
import numpy as np
import torch

from .. import _lib

__all__ = ["LDSEStepPlan", "natural_lds_estep_general", "cython_natural_lds_estep_general",
           "natural_filter_forward_general", "natural_smoother_general", "natural_sample_backward",
           "natural_lds_sample", "cython_natural_lds_sample",
           "natural_lds_inference_general", "cython_natural_lds_inference_general", "reduce_stats",
           "lds_inference_differentiable"]


def _as_dev(x, device):
    if isinstance(x, torch.Tensor):
        t = x.detach()
    else:
        t = torch.as_tensor(x, dtype=torch.float64)   # (a bare python float would become float32)
    return t.to(device=device, dtype=torch.float64).contiguous()


def _canonical_init_params(init_params, device):
    """lds_inference.py:62-63: (J, h, sum of the remaining log-normaliser terms)."""
    J, h = _as_dev(init_params[0], device), _as_dev(init_params[1], device)
    logZ = sum(_as_dev(z, device).reshape(()) for z in init_params[2:]) \
        if len(init_params) > 2 else torch.zeros((), dtype=torch.float64, device=device)
    return J, h, logZ.reshape(1).contiguous()


_default_options = _lib.OPT_DEFAULT


def set_default_options(options):
    """Kernel-selection word (svae_amd._lib.OPT_*) that plans created WITHOUT an explicit `options` take; returns
    the previous one.  Host-side convenience for the tests and A/B tools that run a whole suite through one
    kernel family -- the library itself holds no selection state: each plan passes its word with every call."""
    global _default_options
    old, _default_options = _default_options, int(options)
    return old


def set_accurate_smoother(on=True):
    """Plans created without an explicit `options` word take the kernels whose smoothed moments (and the gradients through
    them) are accurate to cond * eps, like the reference's factor-and-solve, when `on`:
      * E-step: FULL per-step hand-off records (SVAE_OPT_TWOEND_FULL) instead of the two-ended kernels' lean `[P^-1 | c]`
        records, from which P^-1 J12 is rebuilt by multiplying with the explicit inverse every step (cond^2 * eps);
        + 19 % per step at 512 sequences of n = 10, + 79 % at 4096;
      * inference + VJP (n <= 10, <= 2 samples): the one-call kernels on `[chol(P)^-T | c]` records at EVERY batch size
        (SVAE_OPT_LEAN_ON; the default from 1025 sequences) -- factors of balanced magnitude: cond * eps; 1.7 instead of
        0.7 ms per forward + backward at 512 sequences.
    On a model with cond(J22) = 7.8e7 (the worst of 400 draws of the reference's rand_lds): E[x] 1.4e-9 / 1.7e-9 from a
    60-digit solve instead of 2.6e-4 / 7e-4, node gradients 3.5e-9 from the reference's instead of 1.2e-4; on well-conditioned
    models 1e-12 either way ("Conditioning").  Returns the previous default word."""
    acc = _lib.OPT_TWOEND_FULL | _lib.OPT_LEAN_ON
    return set_default_options(((_default_options & ~_lib.OPT_LEAN_OFF) | acc) if on else (_default_options & ~acc))


class LDSEStepPlan(object):
    """Pre-allocated buffers for repeated E-steps of one shape (B, T, n): the launch itself does no
    allocation, no host<->device copy and no synchronisation.  The sampler and the VJP read the
    workspace of the plan's LAST launch: a plan may be reused for a new forward pass only after the
    backward pass of the previous one (checked through `epoch`).  `options`: kernel-selection word passed with
    every call of this plan (svae_amd._lib.OPT_*, include/svae_hip.h SVAE_OPT_*; None = set_default_options)."""

    def __init__(self, B, T, n, device="cuda", inhomog=False, pair_batched=False, options=None):
        if not (1 <= n <= _lib.LDS_TILE_MAX_N):
            raise ValueError("latent dimension n=%d outside the supported range (1..%d)"
                             % (n, _lib.LDS_TILE_MAX_N))
        if T < 1 or B < 0:
            raise ValueError("need T >= 1 and B >= 0")
        self.lib = _lib.load()
        self.B, self.T, self.n, self.inhomog = B, T, n, bool(inhomog)
        self.options = _default_options if options is None else int(options)
        self.device = torch.device(device)
        f64 = dict(dtype=torch.float64, device=self.device)
        # (n > 15: the workspace also holds the re-packed pair parameters, one set per sequence if batched)
        self.ws_bytes = int(self.lib.svae_lds_workspace_bytes_ex(max(B, 1), T, n, int(self.inhomog),
                                                                 int(bool(pair_batched))))
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
        # constants of the reference's statistic tuples (built once: no per-call allocation)
        self.ones_B = torch.ones(B, **f64)
        self.ones_BT = torch.ones(B, T, **f64)
        self.ones_pair = torch.ones(B, max(T - 1, 0), **f64) if self.inhomog \
            else torch.full((B,), float(T - 1), **f64)
        # launch counter: the sampler / VJP read the workspace of the LAST launch, so an autograd node
        # remembers the launch it belongs to and refuses to run after the plan has been reused
        self.epoch = 0

    def launch(self, init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h,
               node_logZ=None, pair_batched=False, keep_factor=False, keep_cross=False, half=0, keep_sigma=False):
        """Raw launch on the current stream.  All arguments: contiguous float64 device tensors.
        half (16 <= n <= 64 only): 1 = the forward half of the E-step (filter, hand-off, lognorm), 2 = the backward half
        (smoother + statistics from the hand-off of a preceding half=1 launch); 0 = both.
        keep_sigma (16 <= n <= 64 only, after `vjp_tail`): the backward half leaves the smoothed covariances in the
        first section of the VJP workspace behind the hand-off (SVAE_KEEP_SIGMA), which saves the VJP its phase 0."""
        p = _lib.ptr
        ev = getattr(self, "_side_event", None)
        if ev is not None:       # work on a helper stream still reads the hand-off this launch overwrites (lds_large.py)
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._side_event = None
        keep = int(bool(keep_factor)) | (2 if keep_cross else 0)
        if self.n > _lib.LDS_MAX_N:
            # tile kernel: its hand-off always serves the sampler / VJP kernels (lds_large.py)
            keep = _lib.KEEP_SIGMA if (keep_sigma and half != 1) else 0
        options = self.options
        if half:
            if self.n <= _lib.LDS_MAX_N:
                raise ValueError("E-step halves: latent dimension > %d only" % _lib.LDS_MAX_N)
            options |= _lib.OPT_TILE_FORWARD if half == 1 else _lib.OPT_TILE_BACKWARD
        rc = self.lib.svae_lds_estep_f64(
            self.B, self.T, self.n, int(self.inhomog), int(pair_batched), keep, options,
            p(init_J), p(init_h), p(init_logZ), p(J11), p(J12), p(J22), p(logZ_pair),
            p(node_J), p(node_h), p(node_logZ),
            p(self.lognorm), p(self.E_init), p(self.E_pair), p(self.E_node_diagxx),
            p(self.E_node_x), p(self.info), p(self.ws), self.ws_bytes,
            _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_estep_f64")
        self.epoch += 1
        self.has_factor = bool(keep_factor) and self.n <= _lib.LDS_MAX_N
        self.has_cross = bool(keep_cross) and self.n <= _lib.LDS_MAX_N
        self.lean, self._infer_S = False, None
        self._J12 = J12
        self._pair_batched = bool(pair_batched)

    def fresh_outputs(self):
        """New output tensors for the next launch (lognorm, E_init, E_pair, E_node_*): a caller that hands the outputs on
        -- the autograd node below -- returns them as they are instead of copying them out of buffers the next launch
        would overwrite (two of them are (B,T,n): 2 x 65 MB at 4096 x 200 x 10)."""
        f64 = dict(dtype=torch.float64, device=self.device)
        B, T, n = self.B, self.T, self.n
        self.lognorm = torch.empty(B, **f64)
        self.E_init = torch.empty(B, n * n + n, **f64)
        self.E_pair = torch.empty(B, max(T - 1, 0), 3, n, n, **f64) if self.inhomog else torch.empty(B, 3, n, n, **f64)
        self.E_node_diagxx = torch.empty(B, T, n, **f64)
        self.E_node_x = torch.empty(B, T, n, **f64)

    def infer(self, init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ=None,
              pair_batched=False, eps=None, out=None, keep_vjp=True):
        """E-step + backward sampler in ONE call (svae_lds_inference_f64 = cython_natural_lds_inference_general,
        lds_inference.py:196-202), keeping what `vjp()` needs.  eps (B,T,S,n) or None (no sampling) -> samples or None.
        For large homogeneous batches (n <= 10, S <= 2, B > 2048, or OPT_LEAN_ON) the library keeps LEAN per-step
        records (csrc/lds_lean_estep.hpp): the same results with a fifth of the hand-off traffic; `sample()` cannot
        follow such a launch (`self.lean`).  keep_vjp=False: forward values only -- no cross-moment record, and lean
        records then also serve per-step / per-sequence pair parameters; `vjp()` cannot follow."""
        if self.n > _lib.LDS_MAX_N:
            raise ValueError("infer(): latent dimension <= %d (the tile path runs its stages separately)" % _lib.LDS_MAX_N)
        p = _lib.ptr
        S = 0
        if eps is not None:
            if eps.dim() != 4 or eps.shape[0] != self.B or eps.shape[1] != self.T or eps.shape[3] != self.n \
                    or eps.shape[2] < 1:
                raise ValueError("eps must be (B,T,S,n) with S >= 1")
            eps = eps.to(device=self.device, dtype=torch.float64).contiguous()
            S = eps.shape[2]
            if out is None:
                out = torch.empty_like(eps)
        rc = self.lib.svae_lds_inference_f64(
            self.B, self.T, self.n, S, int(self.inhomog), int(pair_batched), int(bool(keep_vjp)), self.options,
            p(init_J), p(init_h), p(init_logZ), p(J11), p(J12), p(J22), p(logZ_pair),
            p(node_J), p(node_h), p(node_logZ), p(eps), p(out),
            p(self.lognorm), p(self.E_init), p(self.E_pair), p(self.E_node_diagxx), p(self.E_node_x),
            p(self.info), p(self.ws), self.ws_bytes, _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_inference_f64")
        self.epoch += 1
        self.lean = bool(self.lib.svae_lds_inference_is_lean(self.B, self.T, self.n, S, int(self.inhomog),
                                                             int(bool(keep_vjp)), self.options))
        self.has_factor = not self.lean and (bool(keep_vjp) or S > 0)
        self.has_cross = bool(keep_vjp)
        self._infer_S = S
        self._J12 = J12
        self._pair_batched = bool(pair_batched)
        return out if eps is not None else None

    def vjp_tail(self, S, pair_batched=False):
        """16 <= n <= 64: the workspace of svae_lds_tile_vjp_f64 for S sample cotangents as a view BEHIND the hand-off in
        the plan's own buffer (grown if necessary -- call it before the launch whose hand-off the VJP will read), at
        svae_lds_tile_sigma_offset_bytes: where a launch with keep_sigma leaves the smoothed covariances."""
        B, T, n = max(self.B, 1), self.T, self.n
        off = int(self.lib.svae_lds_tile_sigma_offset_bytes(B, T, n, int(self.inhomog), int(bool(pair_batched)))) // 8
        nws = int(self.lib.svae_lds_tile_vjp_workspace_doubles(B, T, n, S))
        if off == 0:
            raise ValueError("vjp_tail: latent dimension > %d only" % _lib.LDS_MAX_N)
        if self.ws.numel() < off + nws:
            self.ws = torch.empty(off + nws, dtype=torch.float64, device=self.device)
            self.ws_bytes = self.ws.numel() * 8
        return self.ws[off:off + nws]

    def filter(self, init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h,
               node_logZ=None, pair_batched=False, J_pred=None, h_pred=None, J_filt=None, h_filt=None):
        """Filter-only launch (svae_lds_filter_f64): lognorm, optional forward messages, and the hand-off
        `sample()` needs."""
        p = _lib.ptr
        rc = self.lib.svae_lds_filter_f64(
            self.B, self.T, self.n, int(self.inhomog), int(pair_batched), self.options,
            p(init_J), p(init_h), p(init_logZ), p(J11), p(J12), p(J22), p(logZ_pair),
            p(node_J), p(node_h), p(node_logZ), p(self.lognorm), p(J_pred), p(h_pred), p(J_filt), p(h_filt),
            p(self.info), p(self.ws), self.ws_bytes, _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_filter_f64")
        self.epoch += 1
        self.has_factor, self.has_cross = True, False
        self.lean, self._infer_S = False, None
        self._J12 = J12
        self._pair_batched = bool(pair_batched)

    def sample(self, eps, out=None):
        """Backward sampling from the messages of the last `launch(..., keep_factor=True)`.
        eps: (B,T,S,n) standard-normal draws -> samples (B,T,S,n)
        [natural_sample_backward, cython_lds_inference.pyx:310-355]."""
        if eps.dim() != 4 or eps.shape[0] != self.B or eps.shape[1] != self.T or eps.shape[3] != self.n \
                or eps.shape[2] < 1:
            raise ValueError("eps must be (B,T,S,n) with S >= 1")
        if self.n > _lib.LDS_MAX_N:
            # 16 <= n <= 64: noise-factor kernel + recursion kernel on the tile kernel's hand-off (lds_large.py)
            from .lds_large import sample_from_handoff
            if self.epoch == 0:
                raise RuntimeError("sample() needs a preceding launch()")
            return sample_from_handoff(self, eps.to(device=self.device, dtype=torch.float64))
        if getattr(self, "lean", False):
            raise RuntimeError("sample(): the last launch was infer() on lean records -- its samples were drawn there")
        if not getattr(self, "has_factor", False):
            raise RuntimeError("sample() needs a preceding launch(..., keep_factor=True)")
        eps = eps.to(device=self.device, dtype=torch.float64).contiguous()
        S = eps.shape[2]
        if out is None:
            out = torch.empty_like(eps)
        p = _lib.ptr
        rc = self.lib.svae_lds_sample_f64(self.B, self.T, self.n, S, self.options, p(eps), p(out), p(self.ws),
                                          self.ws_bytes, _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_sample_f64")
        return out

    def vjp(self, g_lognorm, g_E_node_diagxx=None, g_E_node_x=None, g_samples=None, eps=None,
            samples=None, g_E_init=None, g_E_pair=None, dense_out=None):
        """Vector-Jacobian product w.r.t. the node potentials of the last
        `launch(..., keep_factor=True, keep_cross=True)` [+ `sample`]: returns (g_node_J, g_node_h)
        (B,T,n) each; g_node_logZ[b,t] = g_lognorm[b].  Replaces the reference's natural_filter_grad /
        natural_smoother_general_grad / natural_sample_backward_grad
        (cython_lds_inference.pyx:92-145, 236-306, 357-409).  g_E_init (B, n*n+n) and, with per-step
        pair parameters, g_E_pair (B,T-1,3,n,n) are the cotangents of the remaining statistics
        (_compute_stats_grad, :212-234) -- what the SLDS-SVAE differentiates.  dense_out (B,T,n,n): also receives
        -2 Pbar_t, the (unsymmetrised) cotangent of a DENSE node potential J_t (svae_lds_estep_vjp_dense_f64)."""
        lean = getattr(self, "lean", False)
        if not (getattr(self, "has_cross", False) and (lean or getattr(self, "has_factor", False))):
            raise RuntimeError("vjp() needs a preceding launch(..., keep_factor=True, keep_cross=True) or infer()")
        if lean and (g_E_init is not None or g_E_pair is not None):
            raise ValueError("lean records (infer() on a large homogeneous batch): no cotangents of E_init / E_pair")
        if g_E_pair is not None and not self.inhomog:
            raise ValueError("a homogeneous plan keeps only the SUMMED pair statistics; their cotangents go through "
                             "the per-step layout: lds_inference_differentiable(..., pair_stats_grad=True)")
        f64 = dict(dtype=torch.float64, device=self.device)
        c = lambda x: None if x is None else x.to(**f64).contiguous()
        g_lognorm, g_E_node_diagxx, g_E_node_x = c(g_lognorm), c(g_E_node_diagxx), c(g_E_node_x)
        g_samples, eps, samples = c(g_samples), c(eps), c(samples)
        g_E_init, g_E_pair = c(g_E_init), c(g_E_pair)
        S = 0 if g_samples is None else g_samples.shape[2]
        options = self.options
        infer_S = getattr(self, "_infer_S", None)
        if infer_S is not None:
            # the workspace was written by infer(): the VJP is told so and takes the S of that call (it fixes the format)
            if g_samples is not None and S != infer_S:
                raise ValueError("vjp(): %d sample cotangents for an infer() call that drew %d" % (S, infer_S))
            options |= _lib.OPT_INFER_RECORDS
            S = infer_S
        if S > 16:
            # the kernels take 16 sample cotangents per launch; the VJP is linear in the cotangents: the first chunk
            # travels with all the others, the remaining chunks alone
            gJ, gh = self.vjp(g_lognorm, g_E_node_diagxx, g_E_node_x, g_samples[:, :, :16], eps[:, :, :16],
                              samples[:, :, :16], g_E_init, g_E_pair)
            zero = torch.zeros_like(g_lognorm)
            for s0 in range(16, S, 16):
                aJ, ah = self.vjp(zero, None, None, g_samples[:, :, s0:s0 + 16], eps[:, :, s0:s0 + 16],
                                  samples[:, :, s0:s0 + 16])
                gJ += aJ
                gh += ah
            return gJ, gh
        if not hasattr(self, "vjp_ws"):
            self.vjp_ws_bytes = int(self.lib.svae_lds_vjp_workspace_bytes(max(self.B, 1), self.T, self.n))
            self.vjp_ws = torch.empty(self.vjp_ws_bytes // 8, **f64)
        gJ = torch.empty(self.B, self.T, self.n, **f64)
        gh = torch.empty(self.B, self.T, self.n, **f64)
        p = _lib.ptr
        if dense_out is not None:
            if S > 16 or lean:
                raise ValueError("dense node-potential cotangents: at most 16 sample cotangents, full records")
            rc = self.lib.svae_lds_estep_vjp_dense_f64(
                self.B, self.T, self.n, S, int(self.inhomog), int(self._pair_batched), options,
                p(self._J12), p(g_lognorm), p(g_E_node_diagxx), p(g_E_node_x), p(g_E_init), p(g_E_pair),
                p(g_samples), p(eps), p(samples), p(self.E_pair), p(self.E_node_x), p(gJ), p(gh), p(dense_out),
                p(self.ws), self.ws_bytes, p(self.vjp_ws), self.vjp_ws_bytes, _lib.current_stream(self.device))
            _lib.check(rc, "svae_lds_estep_vjp_dense_f64")
            return gJ, gh
        rc = self.lib.svae_lds_estep_vjp_ex_f64(
            self.B, self.T, self.n, S, int(self.inhomog), int(self._pair_batched), options,
            p(self._J12), p(g_lognorm), p(g_E_node_diagxx), p(g_E_node_x), p(g_E_init), p(g_E_pair),
            p(g_samples), p(eps), p(samples), p(self.E_pair), p(self.E_node_x), p(gJ), p(gh),
            p(self.ws), self.ws_bytes, p(self.vjp_ws), self.vjp_ws_bytes, _lib.current_stream(self.device))
        _lib.check(rc, "svae_lds_estep_vjp_ex_f64")
        return gJ, gh

    def reduce(self):
        """Deterministic batch sums [sum E_init | sum E_pair | sum lognorm | B] (homogeneous)."""
        if self.inhomog:
            raise ValueError("reduce(): per-step pair statistics (B,T-1,3,n,n) have no batch-summed form here")
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
                                     "(potentials not positive definite; through models.lds.run_inference also: the "
                                     "global natural parameters are not valid)" % (v - 1))


def _is_dense_nodes(node_params):
    if not isinstance(node_params, (tuple, list)) or len(node_params) < 2:
        return False
    nd = lambda x: x.dim() if isinstance(x, torch.Tensor) else np.ndim(x)
    return nd(node_params[0]) == nd(node_params[1]) + 1


def _fold_dense_nodes(natparam, node_params):
    """Dense node potentials J (T,n,n) / (B,T,n,n) of the reference's Python path (`natural_condition_on_general`,
    svae/lds/gaussian.py:46-49; `_canonical_node_params`, lds_inference.py:65-82 -- the compiled path takes diagonal ones
    only, cython_lds_inference.pyx:43).  Every step of the filter, the smoother and the sampler sees the node potential of
    step t only in the sum J_pred[t] + Jo[t] (+ J11[t]) -- `natural_predict`, `natural_rts_backward_step`,
    `natural_condition_on(J_filt, ., ., J11, J12)` -- so the off-diagonal part of Jo[t] is exactly a contribution to the
    pair block of x_t: J11[t] += offdiag(Jo[t]) for t < T-1 and J22[T-2] += offdiag(Jo[T-1]) (init_J for T = 1), with
    per-step, per-sequence pair parameters.  The kernels then run with the diagonal of Jo; only the forward MESSAGES differ
    by these terms (put back in `natural_filter_forward_general`).
    -> (natparam', node_params' batched (B,T,n), info)"""
    init_params, pair_params = natparam
    dev = None
    for x in list(node_params) + list(init_params[:2]):
        if isinstance(x, torch.Tensor) and x.is_cuda:
            dev = x.device
            break
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    node_J, node_h = _as_dev(node_params[0], dev), _as_dev(node_params[1], dev)
    node_logZ = _as_dev(node_params[2], dev) if len(node_params) == 3 else None
    batched = node_h.dim() == 3
    if node_h.dim() not in (2, 3) or node_J.shape[:-1] != node_h.shape or node_J.shape[-1] != node_h.shape[-1]:
        raise ValueError("dense node potentials must be J (T,n,n), h (T,n) or J (B,T,n,n), h (B,T,n)")
    if not batched:
        node_J, node_h = node_J[None], node_h[None]
        node_logZ = None if node_logZ is None else node_logZ[None]
    B, T, n = node_h.shape
    diag = torch.diagonal(node_J, dim1=-2, dim2=-1).contiguous()
    off = node_J - torch.diag_embed(diag)
    off = 0.5 * (off + off.transpose(-1, -2))          # (the factorisations read one triangle: symmetric part)
    J11, J12, J22 = (_as_dev(x, dev) for x in pair_params[:3])
    lz = _as_dev(pair_params[3], dev).reshape(-1)
    homog = J11.dim() == 2
    if T == 1:
        if B != 1:
            raise ValueError("dense node potentials with T = 1: one sequence per call (the initial potential is shared)")
        init_params = (_as_dev(init_params[0], dev) + off[0, 0],) + tuple(init_params[1:])
        pair = (J11, J12, J22, lz)
    else:
        def per_seq(x):
            x = x if x.dim() == 4 else (x[None] if x.dim() == 3 else x[None, None])
            return x.expand(B, T - 1, n, n).clone()
        J11, J12, J22 = per_seq(J11), per_seq(J12), per_seq(J22)
        lz = (lz.reshape(1, -1) if lz.numel() in (1, T - 1) else lz.reshape(B, T - 1)).expand(B, T - 1).contiguous().reshape(-1)
        J11 += off[:, :T - 1]
        J22[:, T - 2] += off[:, T - 1]
        pair = (J11, J12, J22, lz)
    nodes = (diag, node_h.contiguous()) + ((node_logZ,) if node_logZ is not None else ())
    return (init_params, pair), nodes, dict(off=off, homog=homog, batched=batched, B=B, T=T, n=n)


def _dense_stats(stats, info):
    """Statistics of a folded launch in the reference's dense form: E_node = (E[x x'] (T,n,n), E[x], 1) --
    `make_node_stats`, lds_inference.py:163-166 -- and, for homogeneous pair parameters, the pair statistics summed over
    time (:172-173)."""
    Ei, Ep, En = stats
    B, T, n = info["B"], info["T"], info["n"]
    if T > 1:
        ExxT = torch.cat([Ep[0], Ep[2][:, -1:]], dim=1)
        if info["homog"]:
            Ep = (Ep[0].sum(1), Ep[1].sum(1), Ep[2].sum(1), Ep[3].sum(1))
    else:
        ExxT = Ei[0][:, None].clone()
    En = (ExxT, En[1], En[2])
    if not info["batched"]:
        sq = lambda tup: tuple(x[0] for x in tup)
        Ei, Ep, En = sq(Ei), sq(Ep), sq(En)
    return Ei, Ep, En


# Homogeneous pair parameters that arrive as HOST data (NumPy arrays / CPU tensors: what a caller coming from the reference
# passes) are looked at before they go to the device: beyond this condition number of J11 / J22 a plan made on the spot takes
# the cond * eps kernels (set_accurate_smoother's option bits) for that call.  Two n x n SVDs on the host, no device
# synchronisation; device-resident parameters -- the fast path -- are never inspected.  None switches the guard off.
CONDITION_GUARD_THRESHOLD = 1e6


def _host_condition_options(pair_params):
    """option bits for a plan created on the spot: the accurate kernels when the pair blocks are host data and ill-conditioned"""
    if CONDITION_GUARD_THRESHOLD is None:
        return 0
    blocks = []
    for x in (pair_params[0], pair_params[2]):
        if isinstance(x, torch.Tensor):
            if x.is_cuda:
                return 0
            x = x.detach().numpy()
        x = np.asarray(x, dtype=float)
        if x.ndim != 2 or x.shape[0] != x.shape[1] or x.shape[0] > 15 or not np.all(np.isfinite(x)):
            return 0
        blocks.append(x)
    try:
        worst = max(float(np.linalg.cond(b)) for b in blocks)
    except np.linalg.LinAlgError:
        return 0
    return (_lib.OPT_TWOEND_FULL | _lib.OPT_LEAN_ON) if worst > CONDITION_GUARD_THRESHOLD else 0


def _prepare(natparam, node_params, plan):
    """Shape checks / canonical device tensors shared by the E-step, filter and sampler wrappers
    (`_canonical_node_params`, `_canonical_init_params`, lds_inference.py:59-82)."""
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
        raise ValueError("dense (T,n,n) node potentials: through natural_lds_estep_general / natural_lds_sample / "
                         "natural_lds_inference_general / natural_filter_forward_general (folded into per-step pair "
                         "parameters there); the kernels and the differentiable path take diagonal ones, like the "
                         "reference's compiled path (cython_lds_inference.pyx:43)")
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
        guard = 0 if inhomog else _host_condition_options(pair_params)
        plan = LDSEStepPlan(B, T, n, dev, inhomog, pair_batched,
                            options=((_default_options & ~_lib.OPT_LEAN_OFF) | guard) if guard else None)
    elif (plan.B, plan.T, plan.n, plan.inhomog) != (B, T, n, inhomog):
        raise ValueError("plan shape mismatch")
    return dict(plan=plan, batched=batched, B=B, T=T, n=n, inhomog=inhomog, pair_batched=pair_batched,
                args=(init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ))


def natural_lds_estep_general(natparam, node_params, plan=None, check=False, keep_factor=False, _infer_eps=None):
    """E-step = filter + smoother (lds_inference.py:223-237).

    natparam = (init_params, pair_params); init_params = (-1/2 J0, h0, logZ...) and
    pair_params = (J11, J12, J22, logZ) homogeneous (n,n) or per-step (T-1,n,n)
    [or (B,T-1,n,n) with batched nodes]; node_params = (J, h[, logZ]) with diagonal J of shape
    (T,n) or (B,T,n).

    Returns (lognorm, (E_init_stats, E_pair_stats, E_node_stats)) shaped like the reference's
    (cython_lds_inference.pyx:197-210); with a batch axis first when the nodes are batched.

    Accuracy: lognorm is at cond * eps; the smoothed moments of the default kernels for n <= 15 at cond^2 * eps (1e-12 on
    well-conditioned models, within 1e-5 of the reference up to cond(J22) ~ 1e7); set_accurate_smoother() (or a plan with
    SVAE_OPT_TWOEND_FULL) selects the cond * eps kernels.

    Asynchronous like the reference is silent: no host synchronisation unless `check=True`, which reads
    the device-side status word and raises FloatingPointError for potentials that are not positive
    definite (the reference ignores LAPACK `info`, cython_gaussian_grads.pxd:54-76); `plan.check_info()`
    does the same later.  The returned tensors are views of the plan's buffers: valid until its next launch.
    """
    if _is_dense_nodes(node_params):
        if plan is not None:
            raise ValueError("dense node potentials: the plan is built internally (per-step, per-sequence pair layout)")
        natparam, node_params, info = _fold_dense_nodes(natparam, node_params)
        lognorm, stats = natural_lds_estep_general(natparam, node_params, check=check, keep_factor=keep_factor)
        return (lognorm if info["batched"] else lognorm[0]), _dense_stats(stats, info)
    q = _prepare(natparam, node_params, plan)
    plan, batched, B, T, n, inhomog = q["plan"], q["batched"], q["B"], q["T"], q["n"], q["inhomog"]
    dev = plan.device
    if _infer_eps is not None:
        # (natural_lds_inference_general: E-step + sampler in one call -- lean records for large homogeneous batches)
        plan._infer_samples = plan.infer(*q["args"], q["pair_batched"], _infer_eps, keep_vjp=False)
    else:
        plan.launch(*q["args"], q["pair_batched"], keep_factor)
    if check:
        plan.check_info()

    ExxT0 = plan.E_init[:, :n * n].reshape(B, n, n)
    Ex0 = plan.E_init[:, n * n:]
    if inhomog:
        Ep = (plan.E_pair[:, :, 0], plan.E_pair[:, :, 1], plan.E_pair[:, :, 2], plan.ones_pair)
    else:
        Ep = (plan.E_pair[:, 0], plan.E_pair[:, 1], plan.E_pair[:, 2], plan.ones_pair)
    En = (plan.E_node_diagxx, plan.E_node_x, plan.ones_BT)
    Ei = (ExxT0, Ex0, plan.ones_B, plan.ones_B)
    lognorm = plan.lognorm
    if not batched:
        sq = lambda tup: tuple(x[0] for x in tup)
        return lognorm[0], (sq(Ei), sq(Ep), sq(En))
    return lognorm, (Ei, Ep, En)


cython_natural_lds_estep_general = natural_lds_estep_general


def natural_filter_forward_general(init_params, pair_params, node_params, plan=None, check=False):
    """The forward filter alone, with its messages: ((J_pred, h_pred), (J_filt, h_filt)), lognorm in the
    reference's scaling (natural parameters: J = -1/2 precision), shapes (T,n,n) / (T,n) [(B,...) when
    the nodes are batched] -- `natural_filter_forward_general` (cython_lds_inference.pyx:28-90, result
    :84-87; Python twin lds_inference.py:86-106).  The plan's workspace afterwards serves `plan.sample`."""
    if _is_dense_nodes(node_params):
        if plan is not None:
            raise ValueError("dense node potentials: the plan is built internally")
        (ip, pp), nodes, info = _fold_dense_nodes((init_params, pair_params), node_params)
        ((Jp, hp), (Jf, hf)), lognorm = natural_filter_forward_general(ip, pp, nodes, check=check)
        off, T = info["off"], info["T"]
        Jf[:, :T - 1] += off[:, :T - 1]        # J_filt[t] = J_pred[t] + Jo[t]: the part that was folded into J11[t]
        Jp[:, T - 1] -= off[:, T - 1]          # ... and the part the last prediction (or the initial potential) carried
        if not info["batched"]:
            return ((Jp[0], hp[0]), (Jf[0], hf[0])), lognorm[0]
        return ((Jp, hp), (Jf, hf)), lognorm
    q = _prepare((init_params, pair_params), node_params, plan)
    plan, B, T, n = q["plan"], q["B"], q["T"], q["n"]
    if n > _lib.LDS_MAX_N:
        raise ValueError("filter messages: latent dimension <= %d" % _lib.LDS_MAX_N)
    f64 = dict(dtype=torch.float64, device=plan.device)
    Jp, Jf = torch.empty(B, T, n, n, **f64), torch.empty(B, T, n, n, **f64)
    hp, hf = torch.empty(B, T, n, **f64), torch.empty(B, T, n, **f64)
    plan.filter(*q["args"], q["pair_batched"], Jp, hp, Jf, hf)
    if check:
        plan.check_info()
    lognorm = plan.lognorm
    if not q["batched"]:
        return ((Jp[0], hp[0]), (Jf[0], hf[0])), lognorm[0]
    return ((Jp, hp), (Jf, hf)), lognorm


def _model_from_messages(forward_messages, pair_params):
    """The LDS whose forward filter reproduces the given messages: init = the first predicted message, node potential of
    step t = filtered - predicted message of step t.  For messages that came out of a forward filter with these pair
    parameters -- every call site of the reference (lds_inference.py:196-202, 232-237, 260-264) -- the smoother / sampler
    of this model IS the reference's smoother / sampler on the messages."""
    (Jp, hp), (Jf, hf) = forward_messages
    dev = hf.device if isinstance(hf, torch.Tensor) and hf.is_cuda else torch.device("cuda")
    Jp, hp, Jf, hf = (_as_dev(x, dev) for x in (Jp, hp, Jf, hf))
    batched = hf.dim() == 3
    if not batched:
        Jp, hp, Jf, hf = Jp[None], hp[None], Jf[None], hf[None]
    if Jp.dim() != 4 or Jp.shape != Jf.shape or hp.shape != hf.shape or Jp.shape[:3] != hp.shape:
        raise ValueError("forward_messages = ((J_pred, h_pred), (J_filt, h_filt)) with J (T,n,n) and h (T,n) [or a leading B axis]")
    dJ, dh = Jf - Jp, hf - hp
    if Jp.shape[0] > 1:             # one initial potential for the batch: differences of the first predictions join the node
        dJ[1:, 0] += Jp[1:, 0] - Jp[0, 0]
        dh[1:, 0] += hp[1:, 0] - hp[0, 0]
    dg = torch.diagonal(dJ, dim1=-2, dim2=-1)
    off = float((dJ - torch.diag_embed(dg)).abs().max())
    dense = off > 1e-12 * max(float(dJ.abs().max()), 1e-300)
    natparam = ((Jp[0, 0].contiguous(), hp[0, 0].contiguous(), torch.zeros((), dtype=torch.float64, device=dev)),
                pair_params)
    nodes = (dJ if dense else dg.contiguous(), dh)
    if not batched:
        nodes = tuple(x[0] for x in nodes)
    return natparam, nodes


def natural_smoother_general(forward_messages, pair_params):
    """RTS smoother + expected statistics on CALLER-SUPPLIED forward messages: (E_init, E_pair, E_node) as the
    reference's tuples -- `natural_smoother_general(forward_messages, pair_params)`, the second of the functions the
    reference imports from its compiled module (lds_inference.py:18-24; cython_lds_inference.pyx:149-210).
    forward_messages = ((J_pred, h_pred), (J_filt, h_filt)) in the reference's scaling (natural parameters), as
    `natural_filter_forward_general` returns them [(B,T,...) batched].  See _model_from_messages for what is assumed of
    the messages."""
    natparam, nodes = _model_from_messages(forward_messages, pair_params)
    return natural_lds_estep_general(natparam, nodes)[1]


def natural_sample_backward(forward_messages, pair_params, num_samples, eps=None, generator=None):
    """Backward sampling on caller-supplied forward messages -> samples (T,S,n) [(B,T,S,n)]:
    `natural_sample_backward(forward_messages, pair_params, num_samples)` (lds_inference.py:18-24;
    cython_lds_inference.pyx:310-355).  `eps` (T,S,n) as in natural_lds_inference_general (the reference draws
    flipud(randn(T,S,n)) inside, :333)."""
    natparam, nodes = _model_from_messages(forward_messages, pair_params)
    return natural_lds_sample(natparam, nodes, num_samples, eps=eps, generator=generator)


def natural_lds_sample(natparam, node_params, num_samples=1, eps=None, plan=None, generator=None):
    """Filter + backward sampling WITHOUT the smoother: `cython_natural_lds_sample`
    (lds_inference.py:260-264) -> samples (T,S,n) [(B,T,S,n) batched].  `eps` as in
    natural_lds_inference_general."""
    if _is_dense_nodes(node_params):
        if plan is not None:
            raise ValueError("dense node potentials: the plan is built internally")
        natparam, node_params, info = _fold_dense_nodes(natparam, node_params)
        if eps is not None and not info["batched"]:
            eps = torch.as_tensor(eps, dtype=torch.float64)[None]
        samples = natural_lds_sample(natparam, node_params, num_samples, eps, None, generator)
        return samples if info["batched"] else samples[0]
    nh = node_params[1]
    if int(nh.shape[-1]) > _lib.LDS_MAX_N:
        # 16 <= n <= 64: the tile kernels have no filter-only form; the sampler works on the hand-off of the tile
        # E-step (same eps -> sample map), so this is the E-step + sampler with the statistics dropped
        return natural_lds_inference_general(natparam, node_params, num_samples=num_samples, eps=eps, plan=plan,
                                             generator=generator)[0]
    q = _prepare(natparam, node_params, plan)
    plan = q["plan"]
    plan.filter(*q["args"], q["pair_batched"])
    S = int(num_samples)
    if eps is None:
        eps = torch.randn(plan.B, plan.T, S, plan.n, dtype=torch.float64, device=plan.device, generator=generator)
    else:
        eps = torch.as_tensor(eps, dtype=torch.float64)
        eps = eps if q["batched"] else eps[None]
    samples = plan.sample(eps)
    return samples if q["batched"] else samples[0]


cython_natural_lds_sample = natural_lds_sample


def natural_lds_inference_general(natparam, node_params, num_samples=None, eps=None, plan=None,
                                  generator=None):
    """E-step + backward sampling: (samples, expected_stats, lognorm), mirroring
    `cython_natural_lds_inference_general` (lds_inference.py:196-202).  samples: (T,S,n), or (T,n) when
    num_samples is None as in the Python path (:109-124); batched nodes add a leading B axis.
    The reference draws its noise from the global NumPy RNG inside the sampler
    (cython_lds_inference.pyx:333); here `eps` (B,T,S,n) / (T,S,n) may be passed in, else it is drawn
    from `generator` on the device."""
    if _is_dense_nodes(node_params):
        if plan is not None:
            raise ValueError("dense node potentials: the plan is built internally")
        natparam, node_params, info = _fold_dense_nodes(natparam, node_params)
        if eps is not None and not info["batched"]:
            eps = torch.as_tensor(eps, dtype=torch.float64)[None]
        samples, stats, lognorm = natural_lds_inference_general(natparam, node_params, num_samples, eps, None, generator)
        if not info["batched"]:
            samples, lognorm = samples[0], lognorm[0]
        return samples, _dense_stats(stats, info), lognorm
    batched = (node_params[1].ndim if hasattr(node_params[1], "ndim") else torch.as_tensor(node_params[1]).dim()) == 3
    S = 1 if num_samples is None else int(num_samples)
    if plan is None:
        nh = torch.as_tensor(node_params[1])
        B, T, n = (nh.shape if batched else (1,) + tuple(nh.shape))
        pdim = torch.as_tensor(natparam[1][0]).dim()
        plan = LDSEStepPlan(B, T, n, "cuda", pdim >= 3, pdim == 4)
    if eps is None:
        eps = torch.randn(plan.B, plan.T, S, plan.n, dtype=torch.float64, device=plan.device,
                          generator=generator)
    else:
        eps = torch.as_tensor(eps, dtype=torch.float64)
        eps = eps if batched else eps[None]
    if plan.n <= _lib.LDS_MAX_N and eps.dim() == 4 and eps.shape[2] <= 16:
        # ONE call (svae_lds_inference_f64 = the reference's composite, lds_inference.py:196-202)
        lognorm, stats = natural_lds_estep_general(natparam, node_params, plan=plan, keep_factor=True, _infer_eps=eps)
        samples = plan._infer_samples
    else:
        lognorm, stats = natural_lds_estep_general(natparam, node_params, plan=plan, keep_factor=True)
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


class _LDSInference(torch.autograd.Function):
    """Differentiable (w.r.t. the node potentials) E-step + sampler, the torch counterpart of the
    reference's three autograd primitives (lds_inference.py:26-39).  Forward: one E-step launch
    (+ sampler); backward: the two VJP sweeps.  With homogeneous pair parameters the global
    statistics (E_init, E_pair sums) are returned non-differentiable, as in the reference's use
    (svae.py:21 `saved.stats`); with per-step pair parameters (the SLDS, slds_svae.py:295-300)
    E_init and the per-step E_pair carry gradients too."""

    @staticmethod
    def forward(ctx, node_J, node_h, node_logZ, eps, plan, params, pair_batched):
        init_J, init_h, init_logZ, J11, J12, J22, logZ_pair = params
        # Large batches, eager: the launch writes into FRESH output tensors that are handed to autograd as they are (the
        # copies out of the plan's buffers are 0.3 ms of a 3 ms step at 4096 x 200 x 10).  Small batches and anything
        # under stream capture keep the copies: a captured step must write into buffers that outlive the capture, and
        # replacing the plan's buffers inside a capture of the whole make_gradfun step crashed hipStreamEndCapture
        # (ROCm 7.2; bench.py extra[9]) -- there the copies are a few microseconds.
        _copy_out = plan.B <= 1024 or torch.cuda.is_current_stream_capturing()
        if not _copy_out:
            plan.fresh_outputs()
        if eps is None or eps.shape[2] <= 16:
            # one call: E-step + sampler (lean per-step records for large homogeneous batches)
            samples = plan.infer(init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ,
                                 pair_batched, eps)
            if samples is None:
                samples = torch.zeros(0, dtype=torch.float64, device=plan.device)
        else:
            plan.launch(init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ,
                        pair_batched, True, True)
            samples = plan.sample(eps)
        ctx.plan, ctx.has_logZ, ctx.has_samples = plan, node_logZ is not None, eps is not None
        ctx.epoch = plan.epoch
        ctx.set_materialize_grads(False)       # an output nobody differentiated arrives as None, not as zeros
        ctx.save_for_backward(eps if eps is not None else samples, samples)
        E_init, E_pair = plan.E_init, plan.E_pair
        if _copy_out:
            E_init, E_pair = E_init.clone(), E_pair.clone()
        if not plan.inhomog:
            ctx.mark_non_differentiable(E_init, E_pair)
        if _copy_out:
            return (plan.lognorm.clone(), plan.E_node_diagxx.clone(), plan.E_node_x.clone(), samples, E_init, E_pair)
        return (plan.lognorm, plan.E_node_diagxx, plan.E_node_x, samples, E_init, E_pair)

    @staticmethod
    def backward(ctx, g_lognorm, g_dxx, g_x, g_samples, g_init, g_pair):
        plan = ctx.plan
        if plan.epoch != ctx.epoch:
            raise RuntimeError("LDSEStepPlan was launched again before backward(): the hand-off workspace of "
                               "this forward pass is gone (use one plan per live autograd graph, or call "
                               "backward before the next forward)")
        eps, samples = ctx.saved_tensors
        zero = lambda g, like: torch.zeros_like(like) if g is None else g
        g_lognorm = zero(g_lognorm, plan.lognorm)
        gs = g_samples if (ctx.has_samples and g_samples is not None) else None
        if not plan.inhomog:
            g_init = g_pair = None
        gJ, gh = plan.vjp(g_lognorm, g_dxx, g_x, gs, eps if gs is not None else None,
                          samples if gs is not None else None, g_init, g_pair)
        gz = g_lognorm[:, None].expand(plan.B, plan.T).clone() if ctx.has_logZ else None
        return gJ, gh, gz, None, None, None, None


class _LDSInferenceDense(torch.autograd.Function):
    """E-step + sampler with DENSE node potentials J (B,T,n,n) -- the reference's Python path, differentiable end to end
    there (lds_inference.py:65-82, 205-218) -- differentiable w.r.t. (J, h[, logZ]).  Forward: the off-diagonal part of
    J_t is folded into per-step, per-sequence pair parameters (_fold_dense_nodes) and the kernels run on its diagonal;
    backward: J_t enters the recursions only through the pivot block P_t, so its cotangent is the whole of P_t's,
    -2 Pbar_t, which the second VJP sweep holds in registers (svae_lds_estep_vjp_dense_f64) -- symmetrised here, because
    the forward pass reads the symmetric part of J_t.  Outputs are the per-step launch's: (lognorm, E[x] (B,T,n),
    samples, E_init, per-step E_pair (B,T-1,3,n,n)); the caller assembles E[x x'] (B,T,n,n) from E_pair with torch ops."""

    @staticmethod
    def forward(ctx, node_J, node_h, node_logZ, eps, natparam):
        nodes_in = (node_J.detach(), node_h.detach()) + ((node_logZ.detach(),) if node_logZ is not None else ())
        (ip, pp), nodes, info = _fold_dense_nodes(natparam, nodes_in)
        B, T, n = info["B"], info["T"], info["n"]
        if T < 2:
            raise ValueError("differentiable dense node potentials: T >= 2")
        dev = nodes[1].device
        plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
        init_J, init_h, init_logZ = _canonical_init_params(ip, dev)
        J11, J12, J22 = (x.contiguous() for x in pp[:3])
        samples = plan.infer(init_J, init_h, init_logZ, J11, J12, J22, pp[3].reshape(-1).contiguous(),
                             nodes[0].contiguous(), nodes[1].contiguous(), nodes[2].contiguous() if len(nodes) == 3 else None,
                             True, eps)
        ctx.plan, ctx.epoch, ctx.has_logZ, ctx.has_samples = plan, plan.epoch, node_logZ is not None, eps is not None
        if samples is None:
            samples = torch.zeros(0, dtype=torch.float64, device=dev)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(eps if eps is not None else samples, samples)
        return plan.lognorm.clone(), plan.E_node_x.clone(), samples, plan.E_init.clone(), plan.E_pair.clone()

    @staticmethod
    def backward(ctx, g_lognorm, g_x, g_samples, g_init, g_pair):
        plan = ctx.plan
        if plan.epoch != ctx.epoch:
            raise RuntimeError("the plan of this forward pass was launched again before backward()")
        eps, samples = ctx.saved_tensors
        g_lognorm = torch.zeros_like(plan.lognorm) if g_lognorm is None else g_lognorm
        gs = g_samples if (ctx.has_samples and g_samples is not None) else None
        dense = torch.empty(plan.B, plan.T, plan.n, plan.n, dtype=torch.float64, device=plan.device)
        _, gh = plan.vjp(g_lognorm, None, g_x, gs, eps if gs is not None else None, samples if gs is not None else None,
                         g_init, g_pair, dense_out=dense)
        gJ = 0.5 * (dense + dense.transpose(-1, -2))
        gz = g_lognorm[:, None].expand(plan.B, plan.T).clone() if ctx.has_logZ else None
        return gJ, gh, gz, None, None


def _dense_inference_differentiable(natparam, node_params, eps):
    node_J, node_h = node_params[0], node_params[1]
    node_logZ = node_params[2] if len(node_params) == 3 else None
    if node_h.dim() != 3 or node_J.dim() != 4:
        raise ValueError("dense node potentials: J (B,T,n,n), h (B,T,n)")
    cont = lambda x: None if x is None else x.to(torch.float64).contiguous()
    homog = torch.as_tensor(natparam[1][0]).dim() == 2
    lognorm, ex, samples, E_init, E_pair = _LDSInferenceDense.apply(cont(node_J), cont(node_h), cont(node_logZ), cont(eps),
                                                                    natparam)
    ExxT = torch.cat([E_pair[:, :, 0], E_pair[:, -1:, 2]], dim=1)          # E[x_t x_t'] (B,T,n,n): make_node_stats, :163-166
    if homog:
        E_pair = E_pair.sum(1)                                           # (B,3,n,n), differentiable (:172-173)
    return lognorm, (ExxT, ex), (samples if eps is not None else None), (E_init, E_pair)


def lds_inference_differentiable(natparam, node_params, eps=None, plan=None, pair_stats_grad=False):
    """(lognorm (B), (E_node_diagxx, E_node_x) (B,T,n), samples (B,T,S,n) | None, (E_init, E_pair)):
    differentiable w.r.t. node_params = (J (B,T,n), h (B,T,n)[, logZ (B,T)]) through torch autograd.
    Dense node potentials J (B,T,n,n) (the reference's Python path) are accepted too: the first statistic is then the full
    E[x x'] (B,T,n,n), and the gradient w.r.t. J is the symmetric (B,T,n,n) matrix (_LDSInferenceDense).
    Pair parameters (n,n), (T-1,n,n) or (B,T-1,n,n); in the per-step cases E_init (B, n*n+n) and
    E_pair (B,T-1,3,n,n) are differentiable too.

    pair_stats_grad=True with HOMOGENEOUS pair parameters makes E_init and the summed E_pair (B,3,n,n)
    differentiable as well -- the homogeneous branch of the reference's `_compute_stats_grad`
    (cython_lds_inference.pyx:229-231: the cotangent of a sum is the same block at every step): the launch
    uses the per-step layout with the (n,n) parameters repeated over time (T-1 copies, L2-resident), and the
    sum over time is a torch reduction whose backward broadcasts the cotangent to the per-step blocks the
    VJP kernel consumes.  Costs the per-step statistics' HBM traffic, so it is opt-in (no model of the
    reference differentiates these: svae.py:21 keeps them in `saved.stats`)."""
    init_params, pair_params = natparam
    node_J, node_h = node_params[0], node_params[1]
    node_logZ = node_params[2] if len(node_params) == 3 else None
    dev = node_h.device
    if node_h.dim() == 3 and node_J.dim() == 4:
        # DENSE node potentials (the reference's Python path, lds_inference.py:65-82): differentiable since round 6 --
        # returns E[x x'] (B,T,n,n) in place of its diagonal, like the reference's make_node_stats (:163-166)
        if plan is not None or pair_stats_grad:
            raise ValueError("dense node potentials: the plan is built internally; the pair statistics are differentiable as they are")
        return _dense_inference_differentiable(natparam, node_params, eps)
    if node_h.dim() != 3 or node_J.shape != node_h.shape:
        raise ValueError("lds_inference_differentiable: node potentials J, h of shape (B,T,n), or dense J (B,T,n,n)")
    B, T, n = node_h.shape
    init_J, init_h, init_logZ = _canonical_init_params(init_params, dev)
    J11, J12, J22 = (_as_dev(x, dev) for x in pair_params[:3])
    logZ_pair = _as_dev(pair_params[3], dev).reshape(-1)
    inhomog, pair_batched = J11.dim() >= 3, J11.dim() == 4
    sum_pairs = bool(pair_stats_grad) and not inhomog
    if sum_pairs:
        J11, J12, J22 = (x.expand(max(T - 1, 0), n, n).contiguous() for x in (J11, J12, J22))
        logZ_pair = logZ_pair.reshape(1).expand(max(T - 1, 0)).contiguous()
        inhomog = True
    if plan is None:
        guard = 0 if np.ndim(pair_params[0]) != 2 else _host_condition_options(pair_params)
        plan = LDSEStepPlan(B, T, n, dev, inhomog, pair_batched,
                            options=((_default_options & ~_lib.OPT_LEAN_OFF) | guard) if guard else None)
    elif plan.inhomog != inhomog:
        raise ValueError("plan layout mismatch (pair_stats_grad=True needs a per-step plan: inhomog=True)")
    params = (init_J, init_h, init_logZ, J11, J12, J22, logZ_pair)
    cont = lambda x: None if x is None else x.to(torch.float64).contiguous()
    fn = _LDSInference
    if n > _lib.LDS_MAX_N:
        from .lds_large import LDSInferenceLarge as fn     # tile-kernel forward, tile VJP kernels backward
    out = fn.apply(cont(node_J), cont(node_h), cont(node_logZ), cont(eps), plan, params, pair_batched)
    lognorm, dxx, ex, samples, E_init, E_pair = out
    if sum_pairs:
        E_pair = E_pair.sum(1)                     # (B,3,n,n), differentiable
    return lognorm, (dxx, ex), (samples if eps is not None else None), (E_init, E_pair)
