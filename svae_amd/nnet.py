"""Host-side helpers for the recognition / decoder networks around the structured E-step.

The reference's networks (/root/reference/svae/nnet.py) are out of this library's scope -- they are stock autograd /
PyTorch code -- with ONE exception that an LDS-SVAE training step on MI355X cannot do without: the weight gradient of
a dense layer applied to (sequences x T) rows is a GEMM with a reduction axis of 10^5 rows, and rocBLAS's fp64 path
for that shape takes ~11 ms per layer (measured at 512 x 200 rows, 32 x 10 weights: 400x the time of the same product
as a batched GEMM over row blocks + a sum).  `linear` is `x @ w` with that backward; `tanh_mlp` / `gaussian_info` are
the reference's layer stack `nonlin(matmul(x, W) + b)` (nnet.py:23) and its recognition head -- ONE network whose
output is split into (J_input, h) (nnet.py:43-47) -- written on it.  A layer is `(W, b)` as in the reference
(`init_mlp`), or a bare `W` for a bias-free layer.
"""
import math

import torch


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    @torch.autograd.function.once_differentiable       # (a double backward would silently treat the layer as a constant)
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w.t() if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            x2, g2 = x.reshape(-1, x.shape[-1]), g.reshape(-1, g.shape[-1])
            rows = x2.shape[0]
            blk = 256
            full = (rows // blk) * blk
            gw = torch.zeros_like(w)
            if full:
                gw = gw + torch.bmm(x2[:full].reshape(-1, blk, x2.shape[1]).transpose(1, 2),
                                    g2[:full].reshape(-1, blk, g2.shape[1])).sum(0)
            if full < rows:
                gw = gw + x2[full:].t() @ g2[full:]
        return gx, gw


def linear(x, w):
    """x (..., a) @ w (a, b); the weight gradient is a batched GEMM over blocks of 256 rows + a sum."""
    return _Linear.apply(x, w)


def init_mlp(sizes, scale=None, device=None, generator=None):
    """[(W (a, b), bias (b)), ...] for consecutive sizes, float64, requires_grad: the parameter layout of the reference's
    `init_layer_random` (nnet.py:24-25: scale * randn for W and b; its default scale is 1e-2).  scale=None: W ~ 0.3 /
    sqrt(a) randn, b ~ 0.1 randn."""
    kw = dict(dtype=torch.float64, device=device, generator=generator)
    out = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        sw, sb = (0.3 / math.sqrt(a), 0.1) if scale is None else (scale, scale)
        out.append(((sw * torch.randn(a, b, **kw)).requires_grad_(True), (sb * torch.randn(b, **kw)).requires_grad_(True)))
    return out


def _layer(layer, x):
    if isinstance(layer, (tuple, list)):
        w, b = layer
        return linear(x, w) + b
    return linear(x, layer)


def tanh_mlp(layers, x):
    """tanh layers, linear last layer: the reference's `layer(nonlin, W, b)` stacks (nnet.py:23, 52-58).  A layer is
    (W, b), or a bare W (no bias)."""
    for layer in layers[:-1]:
        x = torch.tanh(_layer(layer, x))
    return _layer(layers[-1], x)


def gaussian_info(layers, x):
    """Recognition network + head as in the reference (nnet.py:43-47): ONE MLP whose last layer has 2 n outputs,
    split into (J_input, h); node potentials (J = -1/2 log1pexp(J_input) <= 0: diagonal of -1/2 precision, h)."""
    out = tanh_mlp(layers, x)
    if out.shape[-1] % 2:
        raise ValueError("gaussian_info: the network's output width must be even (J_input | h)")
    J_input, h = out.split(out.shape[-1] // 2, dim=-1)
    return -0.5 * torch.nn.functional.softplus(J_input), h


def gaussian_info_two_heads(params, x):
    """Variant with two separate networks for J and h (NOT the reference's form; rounds 1 - 4 used it)."""
    layers_J, layers_h = params
    return -0.5 * torch.nn.functional.softplus(tanh_mlp(layers_J, x)), tanh_mlp(layers_h, x)
