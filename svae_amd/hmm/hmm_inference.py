"""HMM E-step on MI355X; mirrors /root/reference/svae/hmm/hmm_inference.py.

  hmm_estep(natparam) -> (log_normalizer, (E_init, E_trans, E_states))      (:21-41)
  hmm_logZ(natparam)  -> log_normalizer                                      (:12-17, pyx:93-121)

natparam = (init_params (K), pair_params (K,K), node_params (T,K)) are LOG potentials, as in the
reference.  New: node_params may be (B,T,K) (and pair_params (B,K,K)); outputs then carry a leading
batch axis.  All arithmetic in libsvae_hip.so (svae_hmm_estep_f64); no CPU fallback.
"""
import torch

from .. import _lib

HMM_MAX_K = 64      # (K <= 16: DPP-row kernels; 17 .. 64: one wavefront per sequence, csrc/hmm_estep_wide.hip)


def _dev64(x, device):
    # dtype given up front: torch.as_tensor(python_float) alone would round to float32
    t = x.detach() if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)
    return t.to(device=device, dtype=torch.float64).contiguous()


def hmm_estep(natparam, workspace=None):
    init_params, pair_params, node_params = natparam
    dev = node_params.device if isinstance(node_params, torch.Tensor) and node_params.is_cuda \
        else torch.device("cuda", torch.cuda.current_device())
    init_params, pair_params, node = (_dev64(x, dev) for x in (init_params, pair_params, node_params))
    batched = node.dim() == 3
    if node.dim() not in (2, 3):
        raise ValueError("node_params must be (T,K) or (B,T,K)")
    if not batched:
        node = node[None]
    B, T, K = node.shape
    if not (1 <= K <= HMM_MAX_K):
        raise ValueError("number of states K=%d outside 1..%d" % (K, HMM_MAX_K))
    pair_batched = pair_params.dim() == 3
    if tuple(init_params.shape) != (K,) or tuple(pair_params.shape[-2:]) != (K, K) or \
            (pair_batched and pair_params.shape[0] != B):
        raise ValueError("init/pair parameter shapes do not match the node potentials")
    lib = _lib.load()
    f64 = dict(dtype=torch.float64, device=dev)
    wsb = int(lib.svae_hmm_workspace_bytes(max(B, 1), T, K))
    ws = workspace if workspace is not None else torch.empty(wsb // 8, **f64)
    logZ = torch.empty(B, **f64)
    E_init, E_trans, E_states = torch.empty(B, K, **f64), torch.empty(B, K, K, **f64), torch.empty(B, T, K, **f64)
    p = _lib.ptr
    rc = lib.svae_hmm_estep_f64(B, T, K, int(pair_batched), p(init_params), p(pair_params), p(node),
                                p(logZ), p(E_init), p(E_trans), p(E_states), p(ws), wsb,
                                _lib.current_stream(dev))
    _lib.check(rc, "svae_hmm_estep_f64")
    if not batched:
        return logZ[0], (E_init[0], E_trans[0], E_states[0])
    return logZ, (E_init, E_trans, E_states)


def hmm_logZ(natparam):
    return hmm_estep(natparam)[0]


class _HMMLogZ(torch.autograd.Function):
    """log Z of a batch of HMMs, differentiable w.r.t. the node log-potentials: the gradient is the
    matrix of state marginals the same kernel launch returns (hmm_logZ_grad, cython_hmm_inference.pyx:
    126-166, restricted to the node argument -- all the SLDS-SVAE differentiates, slds_svae.py:150-155)."""

    @staticmethod
    def forward(ctx, node_params, init_params, pair_params):
        logZ, (_, _, E_states) = hmm_estep((init_params, pair_params, node_params))
        ctx.save_for_backward(E_states)
        return logZ

    @staticmethod
    def backward(ctx, g):
        (E_states,) = ctx.saved_tensors
        return g.reshape(g.shape + (1,) * (E_states.dim() - g.dim())) * E_states, None, None


def hmm_logZ_differentiable(natparam):
    """hmm_logZ with gradients flowing to node_params ((T,K) or (B,T,K))."""
    init_params, pair_params, node_params = natparam
    return _HMMLogZ.apply(node_params, init_params, pair_params)
