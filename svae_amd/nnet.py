"""Host-side helpers for the recognition / decoder networks around the structured E-step.

The reference's networks (/root/reference/svae/nnet.py) are out of this library's scope -- they are stock autograd /
PyTorch code -- with ONE exception that an LDS-SVAE training step on MI355X cannot do without: the weight gradient of
a dense layer applied to (sequences x T) rows is a GEMM with a reduction axis of 10^5 rows, and rocBLAS's fp64 path
for that shape takes ~11 ms per layer (measured at 512 x 200 rows, 32 x 10 weights: 400x the time of the same product
as a batched GEMM over row blocks + a sum).  `linear` is `x @ w` with that backward; `tanh_mlp` / `gaussian_info` are
the reference's layer stack and recognition head (nnet.py:19-47) written on it.
"""
import torch


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
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


def tanh_mlp(weights, x):
    """tanh layers, linear last layer (the reference's `make_layer` stacks, nnet.py:19-40, without biases)."""
    for w in weights[:-1]:
        x = torch.tanh(linear(x, w))
    return linear(x, weights[-1])


def gaussian_info(params, x):
    """Recognition head (nnet.py:43-47): node potentials (J <= 0 diagonal of -1/2 precision, h) from two MLPs."""
    wJ, wh = params
    return -0.5 * torch.nn.functional.softplus(tanh_mlp(wJ, x)), tanh_mlp(wh, x)
