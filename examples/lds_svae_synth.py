#!/usr/bin/env python
"""LDS-SVAE on synthetic data, end to end on one MI355X: the reference's training loop
(svae/svae.py:10-39 `make_gradfun` + an SGD/natural-gradient step, cf. experiments/gmm_svae_synth.py)
with the structured E-step, sampler and VJP running in the HIP kernels.

  python examples/lds_svae_synth.py [--iters 30] [--seqs 256] [--T 100] [--n 6] [--p 12]

Recognition network and decoder are small torch MLPs (the reference's svae/nnet.py is out of scope
of this library: stock PyTorch).  Prints the Monte-Carlo ELBO estimate per iteration.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import svae                                   # noqa: E402
from svae_amd.models import lds                             # noqa: E402


def synth_data(seqs, T, n, p, rng):
    """Noisy observations of rotating latent trajectories."""
    th = 0.2
    A = 0.98 * np.eye(n)
    A[:2, :2] = 0.98 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    C = rng.standard_normal((n, p)) / np.sqrt(n)
    x = rng.standard_normal((seqs, n))
    ys = []
    for _ in range(T):
        x = x @ A.T + 0.1 * rng.standard_normal((seqs, n))
        ys.append(np.tanh(x @ C) + 0.1 * rng.standard_normal((seqs, p)))
    return np.stack(ys, 1)


def mlp(sizes, gen, dev):
    return [(0.3 * torch.randn(a, b, dtype=torch.float64, device=dev, generator=gen) / np.sqrt(a)).requires_grad_(True)
            for a, b in zip(sizes[:-1], sizes[1:])]


def forward(ws, x):
    from svae_amd.nnet import tanh_mlp        # x @ w layers whose weight gradient avoids rocBLAS's long-reduction fp64 path
    return tanh_mlp(ws, x)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--seqs", type=int, default=256)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--n", type=int, default=6)
    ap.add_argument("--p", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lr", type=float, default=3e-3)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    data = torch.as_tensor(synth_data(args.seqs, args.T, args.n, args.p, np.random.default_rng(0)), device=dev)
    n, p = args.n, args.p

    prior = tuple(x.to(dev) if isinstance(x, torch.Tensor) else tuple(y.to(dev) for y in x)
                  for x in lds.make_prior_natparam(n))
    pgm = tuple(x.clone() if isinstance(x, torch.Tensor) else tuple(y.clone() for y in x) for x in prior)
    recogn = (mlp([p, 32, n], gen, dev), mlp([p, 32, n], gen, dev))        # heads for (J, h)
    decoder = mlp([n, 32, p], gen, dev)

    def recognize(params, batch):            # nnet.gaussian_info-like head (svae/nnet.py:43-47): J <= 0 diagonal
        wJ, wh = params
        return -0.5 * torch.nn.functional.softplus(forward(wJ, batch)), forward(wh, batch)

    def loglike(params, samples, batch):     # unit-variance Gaussian decoder, averaged over samples
        mean = forward(params, samples)                              # (B,T,S,p)
        return -0.5 * ((batch.unsqueeze(2) - mean) ** 2).sum() / samples.shape[2]

    vals = []
    gradfun = svae.make_gradfun(lambda *a: lds.run_inference_differentiable(*a, generator=gen), recognize, loglike,
                                prior, data, args.batch, 1, natgrad_scale=1e2,
                                callback=lambda i, v, p_, g: vals.append(-v))
    leaves = lambda s: svae._leaves(s)
    for it in range(args.iters):
        natgrad, g_dec, g_rec = gradfun((pgm, decoder, recogn), it)
        with torch.no_grad():
            for w, g in zip(leaves((decoder, recogn)), leaves((g_dec, g_rec))):
                w -= args.lr * g                                    # plain SGD on the networks
            pgm = svae.unflat_like(svae.flat(pgm) - args.lr * svae.flat(natgrad), prior)   # natural-gradient step
        if not args.quiet:
            print("iter %3d  ELBO estimate per datapoint %.4f" % (it, vals[-1]))
    return vals


if __name__ == "__main__":
    main()
