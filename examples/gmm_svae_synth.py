#!/usr/bin/env python
"""GMM-SVAE on pinwheel data, end to end on one MI355X: the training loop of the reference's
experiments/gmm_svae_synth.py (svae/svae.py:10-39 `make_gradfun` + SGD / natural-gradient steps, K = 15 components,
2-D latents, minibatches of 50 points) with the whole structured side -- global maps, mean-field fixed point, final
pass, sampler and the reverse pass -- running in the HIP kernels (svae_amd/models/gmm.py).

  python examples/gmm_svae_synth.py [--iters 200] [--K 15] [--batch 50]

Recognition network and decoder are small torch MLPs on svae_amd.nnet (the reference's gresnets, svae/nnet.py, are out
of this library's scope).  Prints the Monte-Carlo ELBO estimate per iteration and, at the end, how many components
carry points.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svae_amd import svae                                   # noqa: E402
from svae_amd.models import gmm                             # noqa: E402
from svae_amd.nnet import gaussian_info_two_heads as gaussian_info, tanh_mlp           # noqa: E402


def pinwheel(radial_std, tangential_std, num_classes, num_per_class, rate, rng):
    """Pinwheel clusters (the data of the reference's experiment): points spread along `num_classes` arms that bend with
    the radius; returned shuffled and scaled by 10."""
    arms = np.linspace(0., 2. * np.pi, num_classes, endpoint=False)
    pts = rng.standard_normal((num_classes * num_per_class, 2)) * np.array([radial_std, tangential_std])
    pts[:, 0] += 1.
    which = np.repeat(np.arange(num_classes), num_per_class)
    ang = arms[which] + rate * np.exp(pts[:, 0])
    rot = np.stack([np.stack([np.cos(ang), -np.sin(ang)], -1), np.stack([np.sin(ang), np.cos(ang)], -1)], -2)   # (T,2,2)
    out = np.einsum("ti,tij->tj", pts, rot)
    return 10. * out[rng.permutation(len(out))]


def mlp(sizes, gen, dev):
    return [(torch.randn(a, b, dtype=torch.float64, device=dev, generator=gen) / np.sqrt(a)).requires_grad_(True)
            for a, b in zip(sizes[:-1], sizes[1:])]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--K", type=int, default=15)
    ap.add_argument("--clusters", type=int, default=5)
    ap.add_argument("--per-cluster", type=int, default=100)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    cpu_gen = torch.Generator().manual_seed(1)
    K, N, P = args.K, 2, 2
    data = torch.as_tensor(pinwheel(0.3, 0.05, args.clusters, args.per_cluster, 0.25, np.random.default_rng(1)), device=dev)

    to = lambda s: tuple(x.to(dev) for x in s)
    prior = to(gmm.init_pgm_param(K, N, alpha=0.05 / K, niw_conc=0.5, generator=cpu_gen))       # sparsifying prior
    pgm = to(gmm.init_pgm_param(K, N, alpha=1., niw_conc=1., random_scale=3., generator=cpu_gen))
    recogn = (mlp([P, 40, 40, N], gen, dev), mlp([P, 40, 40, N], gen, dev))                      # heads for (J, h)
    decoder = mlp([N, 40, 40, P], gen, dev)

    def loglike(params, samples, batch):     # unit-variance Gaussian decoder, averaged over samples
        mean = tanh_mlp(params, samples)                              # (T,S,P)
        return -0.5 * ((batch.unsqueeze(1) - mean) ** 2).sum() / samples.shape[1]

    vals = []
    run = lambda *a: gmm.run_inference_differentiable(*a, generator=gen)
    gradfun = svae.make_gradfun(run, gaussian_info, loglike, prior, data, args.batch, 1, natgrad_scale=1e4,
                                callback=lambda i, v, p_, g: vals.append(-v), generator=cpu_gen)
    leaves = svae._leaves
    for it in range(args.iters):
        natgrad, g_dec, g_rec = gradfun((pgm, decoder, recogn), it)
        with torch.no_grad():
            for w, g in zip(leaves((decoder, recogn)), leaves((g_dec, g_rec))):
                w -= args.lr * g                                    # plain SGD on the networks
            pgm = svae.unflat_like(svae.flat(pgm) - args.lr * svae.flat(natgrad), prior)   # natural-gradient step
        if not args.quiet and it % 20 == 0:
            print("iter %4d  ELBO estimate per datapoint %.4f" % (it, vals[-1]))
    # which components carry the data after training
    with torch.no_grad():
        (labels, _), _, _, _ = gmm.local_meanfield(pgm, gaussian_info(recogn, data), generator=gen)
    used = int((torch.bincount(labels.argmax(1), minlength=K) > 0).sum())
    if not args.quiet:
        print("components carrying points: %d of %d" % (used, K))
    return vals, used


if __name__ == "__main__":
    main()
