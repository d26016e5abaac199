"""Sampler and reverse-mode derivative of the LDS E-step for latent dimension 16 <= n <= 64: kernel launches only.

The E-step itself runs in the LDS-tiled MFMA kernel (csrc/lds_estep_tile.hip).  What follows it in a training step
-- `natural_sample_backward` (cython_lds_inference.pyx:310-355) and the VJPs (`natural_filter_grad` :92-145,
`_compute_stats_grad` :212-234, `natural_smoother_general_grad` :236-306, `natural_sample_backward_grad` :357-409)
-- runs on the tile kernel's hand-off (G_t = -P_t^-1 J12, c_t, P_t^-1 per step) in the kernels of
csrc/lds_vjp_tile.hip:

  * sampler: the reference's noise map is chol(P_t)^-T eps_t; chol(P)^-T is the unique upper-triangular M with
    P^-1 = M M' ("UL" Cholesky), computed per (sequence, step) by svae_lds_tile_noise_f64 (it does not depend on the
    recursion); the serial recursion x_t = c_t + G_t x_{t+1} + noise_t is svae_lds_tile_sample_f64.  The kernels take
    up to 16 samples per sequence per launch; more are drawn 16 at a time (the recursion is linear in the noise and
    the factor does not depend on the sample).
  * VJP: svae_lds_tile_vjp_f64 -- the adjoint of the kernels' recursion in three passes over time (0: smoothed
    covariances; 1: adjoint of the smoother / sampler recursions incl. the cotangents of E_init and of the per-step
    pair statistics; 2: adjoint of the filter), with the Cholesky adjoint of the noise factor
    (svae_lds_tile_noise_f64, mode 1) between passes 1 and 2.  More than 16 sample cotangents: the VJP is linear in
    the cotangents, so the chunks beyond the first run as further VJPs with only their sample cotangents.

Concurrency of a training step (one workgroup per sequence leaves most of the chip idle at the batch sizes these models
run at): once the FORWARD half of the E-step has written the hand-off, its backward half (smoother + statistics) and the
noise factor + sampler recursion run side by side on two streams; when a backward pass will follow, the backward half also
leaves the smoothed covariances where the VJP wants them (LDSEStepPlan.vjp_tail / SVAE_KEEP_SIGMA), so that phase 0 is
only launched by callers that bring a plain E-step launch (start_phase0 / vjp_from_handoff_hip without `phase0`).  In the
backward pass the Cholesky adjoint and phase 2 are split into ranges of steps, the adjoint of the next range running next
to phase 2 of the current one.

Torch restatements of the same algebra (the CPU cross-check of the derivation) live in tests/_lds_large_torch.py.
Everything here is float64 on the GPU; there is no CPU path and no library (rocBLAS / rocSOLVER) call.
"""
import torch

from .. import _lib

MAX_S = 16       # samples per sequence per kernel launch (csrc/lds_vjp_tile.hip: TV_MAX_S)
PHASE2_RANGES = 8        # ranges of steps phase 2 and the Cholesky adjoint are split into (see vjp_from_handoff_hip)
PHASE2_MIN_STEPS = 64    # ... none of them shorter than this


def _np16(n):
    return 16 * ((n + 15) // 16)


def handoff_views(plan):
    """(G (B,T,n,n), Pinv (B,T,n,n), c (B,T,n)) views of the tile kernel's workspace (TileCfg::WSTEP:
    X, P^-1 row-major NP x NP, then c)."""
    B, T, n = plan.B, plan.T, plan.n
    NP = _np16(n)
    W = 2 * NP * NP + NP
    ws = plan.ws[:B * T * W].view(B, T, W)
    G = ws[..., :NP * NP].view(B, T, NP, NP)[..., :n, :n]
    Pinv = ws[..., NP * NP:2 * NP * NP].view(B, T, NP, NP)[..., :n, :n]
    c = ws[..., 2 * NP * NP:2 * NP * NP + n]
    return G, Pinv, c


def sample_from_handoff(plan, eps):
    """Backward sampling after a tile-kernel E-step: eps (B,T,S,n) -> samples (B,T,S,n), same
    eps -> sample map as the reference (noise_t = chol(P_t)^-T eps_t)."""
    B, T, n = plan.B, plan.T, plan.n
    S = eps.shape[2]
    lib, p = _lib.load(), _lib.ptr
    stream = _lib.current_stream(plan.device)
    out = torch.empty(B, T, S, n, dtype=torch.float64, device=plan.device)
    for s0 in range(0, S, MAX_S):        # noise factor per (sequence, step) + the serial recursion: two launches per chunk
        s1 = min(S, s0 + MAX_S)
        whole = s0 == 0 and s1 == S
        e = eps.contiguous() if whole else eps[:, :, s0:s1].contiguous()
        noise = torch.empty_like(e)
        o = out if whole else torch.empty_like(e)
        rc = lib.svae_lds_tile_noise_f64(0, B, T, n, s1 - s0, 0, T, p(e), p(noise), p(plan.ws), None, p(plan.info), stream)
        _lib.check(rc, "svae_lds_tile_noise_f64")
        rc = lib.svae_lds_tile_sample_f64(B, T, n, s1 - s0, p(noise), p(o), p(plan.ws), stream)
        _lib.check(rc, "svae_lds_tile_sample_f64")
        if not whole:
            out[:, :, s0:s1] = o
    return out


_side_streams = {}
_keep_ws, _last_ws = False, None      # set by tools/tile_vjp_timing.py only


def _side_stream(dev, which=0):
    """Helper streams per (device, caller stream): work that only depends on the E-step's hand-off runs there, next to
    the kernels of the caller's stream that leave most of the chip idle (one workgroup per sequence)."""
    main = torch.cuda.current_stream(dev)
    key = (dev.index, main.cuda_stream, which)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return main, _side_streams[key]


def start_phase0(plan, J12, pair_batched, S):
    """Phase 0 of the VJP (the smoothed covariances Sigma_t: a function of the E-step's hand-off alone) launched on the
    helper stream right after the E-step, so that it runs NEXT to the noise factor and the sampler recursion of the
    forward pass instead of in front of the backward pass.  -> (workspace, event) for vjp_from_handoff_hip."""
    lib, p = _lib.load(), _lib.ptr
    B, T, n = plan.B, plan.T, plan.n
    dev = plan.device
    f64 = dict(dtype=torch.float64, device=dev)
    nws = int(lib.svae_lds_tile_vjp_workspace_doubles(max(B, 1), T, n, S))
    # the VJP workspace belongs to the plan (one live autograd graph per plan): no allocation per step, and no second
    # block while the allocator waits for the helper streams of the previous step to release the first
    ws = getattr(plan, "_vjp_ws", None)
    if ws is None or ws.numel() < nws:
        ws = plan._vjp_ws = torch.empty(nws, **f64)
    dummy = plan.lognorm                      # (phase 0 reads none of the cotangents; the entry point wants the pointers)
    gJ = torch.empty(1, **f64)
    main, side = _side_stream(dev)
    side.wait_stream(main)                    # the hand-off of the launch just issued (and everything before it)
    J12c = J12.to(**f64).contiguous()
    # temporaries made on the caller's stream and consumed on the helper stream: tell the caching allocator, or their
    # memory may be handed out again while phase 0 is still running
    J12c.record_stream(side)
    gJ.record_stream(side)
    rc = lib.svae_lds_tile_vjp_f64(0, B, T, n, 0, 0, T, int(J12c.dim() >= 3), int(bool(pair_batched)), p(J12c), p(dummy),
                                   None, None, None, None, None, None, p(plan.E_node_x), p(gJ), p(gJ), p(plan.ws),
                                   p(ws), nws, side.cuda_stream)
    _lib.check(rc, "svae_lds_tile_vjp_f64")
    ev = torch.cuda.Event()
    ev.record(side)
    plan._side_event = ev                     # the next launch on this plan overwrites the hand-off: it waits for this
    return ws, ev


def vjp_from_handoff_hip(plan, J12, pair_batched, ex, g_lognorm, g_dxx, g_x, samples=None, eps=None, g_samples=None,
                         g_E_init=None, g_E_pair=None, phase0=None):
    """The VJP w.r.t. the node potentials on the device kernels (svae_lds_tile_vjp_f64: three phases, one workgroup
    per sequence, n x n state in LDS; the Cholesky adjoint of the sampler's noise factor -- parallel over all
    (sequence, step) pairs -- is svae_lds_tile_noise_f64 between phases 1 and 2).  Reads the hand-off of the plan's
    last launch.  phase0: (workspace, event) -- the smoothed covariances are (after `event`) in the first section of
    `workspace`: from start_phase0, or from a forward pass launched with keep_sigma on LDSEStepPlan.vjp_tail.
    -> (g_node_J, g_node_h) (B,T,n)."""
    lib = _lib.load()
    B, T, n = plan.B, plan.T, plan.n
    dev = plan.device
    f64 = dict(dtype=torch.float64, device=dev)
    cont = lambda x: None if x is None else x.to(**f64).contiguous()
    g_lognorm, g_dxx, g_x, g_E_init = cont(g_lognorm), cont(g_dxx), cont(g_x), cont(g_E_init)
    has_s = g_samples is not None
    samples, eps, g_samples = (cont(samples), cont(eps), cont(g_samples)) if has_s else (None, None, None)
    g_E_pair = cont(g_E_pair)
    S = samples.shape[2] if has_s else 0
    if S > MAX_S:
        # linear in the cotangents: the first chunk travels with all the other cotangents, the rest alone
        gJ, gh = vjp_from_handoff_hip(plan, J12, pair_batched, ex, g_lognorm, g_dxx, g_x, samples[:, :, :MAX_S],
                                      eps[:, :, :MAX_S], g_samples[:, :, :MAX_S], g_E_init, g_E_pair, phase0=phase0)
        zero = torch.zeros_like(g_lognorm)
        for s0 in range(MAX_S, S, MAX_S):
            sl = slice(s0, min(S, s0 + MAX_S))
            aJ, ah = vjp_from_handoff_hip(plan, J12, pair_batched, ex, zero, None, None, samples[:, :, sl],
                                          eps[:, :, sl], g_samples[:, :, sl])
            gJ += aJ
            gh += ah
        return gJ, gh
    nws = int(lib.svae_lds_tile_vjp_workspace_doubles(max(B, 1), T, n, S))
    if phase0 is not None and phase0[0].numel() >= nws:
        ws = phase0[0]
        torch.cuda.current_stream(dev).wait_event(phase0[1])
    else:
        ws, phase0 = getattr(plan, "_vjp_ws", None), None
        if ws is None or ws.numel() < nws:
            ws = plan._vjp_ws = torch.empty(nws, **f64)
    gJ, gh = torch.empty(B, T, n, **f64), torch.empty(B, T, n, **f64)
    if _keep_ws:
        global _last_ws
        _last_ws = ws                          # (tools/tile_vjp_timing.py reads phase 1's counters from it)
    J12 = cont(J12)
    inhomog = J12.dim() >= 3
    p = _lib.ptr

    def phase(k, t0=0, t1=T, stream=None):
        rc = lib.svae_lds_tile_vjp_f64(k, B, T, n, S, t0, t1, int(inhomog), int(bool(pair_batched)), p(J12), p(g_lognorm),
                                       p(g_dxx), p(g_x), p(g_E_init), p(g_E_pair), p(g_samples), p(samples), p(ex), p(gJ),
                                       p(gh),
                                       p(plan.ws), p(ws), nws, stream if stream is not None else _lib.current_stream(dev))
        _lib.check(rc, "svae_lds_tile_vjp_f64")

    def chol_adjoint(t0, t1, stream):
        rc = lib.svae_lds_tile_noise_f64(1, B, T, n, S, t0, t1, p(eps), None, p(plan.ws), p(ws), p(plan.info), stream)
        _lib.check(rc, "svae_lds_tile_noise_f64")
    if phase0 is None:
        phase(0)
    phase(1)
    if not has_s:
        phase(2)
        return gJ, gh
    # Cholesky adjoint of the noise factor into pinv_bar (parallel over all (sequence, step) pairs: the whole chip),
    # then phase 2 (one workgroup per sequence, backward in time).  In ranges of steps, last range first: the adjoint of
    # the next (earlier) range runs on the helper stream NEXT to phase 2 of the current one.
    nr = min(PHASE2_RANGES, max(1, T // PHASE2_MIN_STEPS))
    bounds = [(k * T) // nr for k in range(nr + 1)]
    if nr == 1:
        chol_adjoint(0, T, _lib.current_stream(dev))
        phase(2)
        return gJ, gh
    main, side = _side_stream(dev)
    side.wait_stream(main)                      # phase 1 (xbar, the direct part of pinv_bar)
    for buf in (eps, gJ, gh):
        buf.record_stream(side)
    done = []
    for k in reversed(range(nr)):
        chol_adjoint(bounds[k], bounds[k + 1], side.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(side)
        done.append(ev)
    for k, ev in zip(reversed(range(nr)), done):
        main.wait_event(ev)
        phase(2, bounds[k], bounds[k + 1])
    return gJ, gh


class LDSInferenceLarge(torch.autograd.Function):
    """Differentiable (w.r.t. the node potentials) E-step + sampler for 16 <= n <= 64: forward = the tile kernel
    (+ sample_from_handoff), backward = the VJP kernels on the hand-off of the same launch.  With per-step pair
    parameters E_init and the per-step E_pair carry gradients too (the SLDS, slds_svae.py:295-300)."""

    @staticmethod
    def forward(ctx, node_J, node_h, node_logZ, eps, plan, params, pair_batched):
        init_J, init_h, init_logZ, J11, J12, J22, logZ_pair = params
        args = (init_J, init_h, init_logZ, J11, J12, J22, logZ_pair, node_J, node_h, node_logZ, pair_batched, False, False)
        needs_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        ctx.phase0 = None
        if eps is None and not needs_grad:
            plan.launch(*args)
        else:
            # Three consumers of the hand-off run side by side once the forward half of the E-step has written it: the
            # backward half (smoother + statistics), phase 0 of the VJP (a backward pass will follow), and the noise
            # factor + sampler recursion -- each one workgroup per sequence, or shorter than the smoother.
            # (a backward pass will follow: the VJP workspace lives behind the hand-off in the plan's buffer, and the
            #  backward half leaves the smoothed covariances in its first section -- phase 0 of the VJP, which would
            #  rebuild them from the hand-off next to the backward half, is not launched at all)
            vws = plan.vjp_tail(min(eps.shape[2], MAX_S) if eps is not None else 0, pair_batched) if needs_grad else None
            plan.launch(*args, half=1)
            main, side2 = _side_stream(plan.device, 1)
            side2.wait_stream(main)
            with torch.cuda.stream(side2):
                plan.launch(*args, half=2, keep_sigma=needs_grad)
            smoothed = torch.cuda.Event()
            smoothed.record(side2)
            if needs_grad:
                ctx.phase0 = (vws, smoothed)
        samples = sample_from_handoff(plan, eps) if eps is not None else \
            torch.zeros(0, dtype=torch.float64, device=plan.device)
        if eps is not None or needs_grad:
            torch.cuda.current_stream(plan.device).wait_event(smoothed)      # the statistics, before they are copied
        ctx.set_materialize_grads(False)       # an output nobody differentiated arrives as None, not as zeros
        ctx.J12, ctx.inhomog, ctx.has_logZ, ctx.has_eps = J12, plan.inhomog, node_logZ is not None, eps is not None
        ctx.plan, ctx.epoch, ctx.pair_batched = plan, plan.epoch, pair_batched
        ctx.ex = plan.E_node_x.clone()
        ctx.save_for_backward(eps if eps is not None else samples, samples)
        E_init, E_pair = plan.E_init.clone(), plan.E_pair.clone()
        if not plan.inhomog:
            ctx.mark_non_differentiable(E_init, E_pair)
        return (plan.lognorm.clone(), plan.E_node_diagxx.clone(), plan.E_node_x.clone(), samples, E_init, E_pair)

    @staticmethod
    def backward(ctx, g_lognorm, g_dxx, g_x, g_samples, g_init, g_pair):
        eps, samples = ctx.saved_tensors
        plan = ctx.plan
        B, T = plan.B, plan.T
        if plan.epoch != ctx.epoch:
            raise RuntimeError("LDSEStepPlan was launched again before backward(): the hand-off workspace of "
                               "this forward pass is gone (use one plan per live autograd graph)")
        gl = torch.zeros_like(plan.lognorm) if g_lognorm is None else g_lognorm
        gs = g_samples if (ctx.has_eps and g_samples is not None) else None
        gJ, gh = vjp_from_handoff_hip(plan, ctx.J12, ctx.pair_batched, ctx.ex, gl, g_dxx, g_x,
                                      samples if gs is not None else None, eps if gs is not None else None, gs,
                                      g_init if ctx.inhomog else None, g_pair if ctx.inhomog else None,
                                      phase0=ctx.phase0)
        gz = gl[:, None].expand(B, T).clone() if ctx.has_logZ else None
        return gJ, gh, gz, None, None, None, None
