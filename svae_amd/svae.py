"""make_gradfun: the SVAE training-step contract of the reference, on torch.

Mirrors /root/reference/svae/svae.py:10-39 (same argument names and order, same returned tuple):
the Monte-Carlo ELBO estimator `mc_elbo` (:19-24), its gradient w.r.t. (loglike_params,
recogn_params) by reverse-mode AD (autograd there, torch here; the structured E-step in between is
the HIP kernels with their own VJP kernels), and the closed-form NATURAL gradient of the PGM
parameters from the expected statistics the E-step left in `saved.stats` (:33-34).

  gradfun = make_gradfun(run_inference, recognize, loglike, pgm_prior, data,
                         batch_size, num_samples, natgrad_scale=1., callback=callback)
  grad = gradfun((pgm_params, loglike_params, recogn_params), i)
       = (pgm_natgrad [same nested structure as pgm_prior], loglike_grad, recogn_grad)

`recognize(recogn_params, batch) -> nn_potentials` and `loglike(loglike_params, samples, batch) ->
scalar` are the user's torch functions (the reference's svae/nnet.py is out of scope: stock PyTorch);
`run_inference` is svae_amd.models.{lds,gmm}.run_inference_differentiable or anything with the
same signature.  `make_gradfun` is CURRIED like the reference's (`@curry`, svae.py:10; toolz is not a
dependency: `curry` below): a call that leaves required arguments open returns a function waiting for
the rest, so the shipped training scripts' two-stage form works verbatim
(experiments/gmm_svae_synth.py:57-61):
    gradfun = make_gradfun(run_inference, recognize, loglike, pgm_prior_params, data)
    sgd(gradfun(batch_size=50, num_samples=1, natgrad_scale=1e4, callback=plot), params, ...)
"""
import functools
import inspect

import torch


def curry(fn):
    """toolz.curry for the one use the reference makes of it (svae.py:2,10): calling with some required arguments
    still missing returns a curried function holding the ones given (positional and keyword); a call that binds
    every required parameter runs `fn`.  A call that could never bind (unknown keyword, too many positionals)
    raises TypeError at once, as toolz does."""
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def curried(*args, **kwargs):
        try:
            sig.bind(*args, **kwargs)
        except TypeError:
            sig.bind_partial(*args, **kwargs)          # (raises for calls no further argument can complete)
            return curry(functools.partial(fn, *args, **kwargs))
        return fn(*args, **kwargs)

    return curried

callback = lambda i, val, params, grad: print("{}: {}".format(i, val))


def _leaves(struct):
    if isinstance(struct, (tuple, list)):
        out = []
        for s in struct:
            out += _leaves(s)
        return out
    return [struct]


def flat(struct):
    """autograd.util.flatten(struct)[0] (svae.py:8): concatenation of the raveled leaves."""
    return torch.cat([torch.as_tensor(x, dtype=torch.float64).reshape(-1) for x in _leaves(struct)])


def unflat_like(vec, struct):
    pos = [0]

    def build(s):
        if isinstance(s, (tuple, list)):
            return tuple(build(x) for x in s)
        t = torch.as_tensor(s)
        n = t.numel()
        out = vec[pos[0]:pos[0] + n].reshape(t.shape)
        pos[0] += n
        return out
    return build(struct)


def split_into_batches(data, batch_size, permute=True, generator=None):
    """util.py:120-126: chunks of `batch_size` consecutive rows, in a random order (the reference
    permutes the chunks once with the global NumPy RNG; here `generator` seeds torch.randperm,
    permute=False keeps the data order).  For (sequences, T, p) data a chunk is `batch_size` whole
    sequences."""
    k = data.shape[0] // batch_size
    chunks = [data[i * batch_size:(i + 1) * batch_size] for i in range(k)]
    if permute and k > 1:
        order = torch.randperm(k, generator=generator).tolist()
        chunks = [chunks[i] for i in order]
    return chunks, k


@curry
def make_gradfun(run_inference, recognize, loglike, pgm_prior, data, batch_size, num_samples,
                 natgrad_scale=1., callback=callback, permute=True, generator=None):
    # number of data points = rows (time steps / points), as get_num_datapoints(data) in the reference
    # (svae.py:13): for (sequences, T, p) data every time step counts
    num_datapoints = data.shape[0] * (data.shape[1] if data.dim() == 3 else 1)
    data_batches, num_batches = split_into_batches(data, batch_size, permute, generator)
    get_batch = lambda i: data_batches[i % num_batches]
    saved = lambda: None

    def mc_elbo(pgm_params, loglike_params, recogn_params, i):
        nn_potentials = recognize(recogn_params, get_batch(i))
        samples, saved.stats, global_kl, local_kl = \
            run_inference(pgm_prior, pgm_params, nn_potentials, num_samples)
        return (num_batches * loglike(loglike_params, samples, get_batch(i))
                - global_kl - num_batches * local_kl) / num_datapoints

    def gradfun(params, i):
        pgm_params, loglike_params, recogn_params = params
        leaves = [p for p in _leaves((loglike_params, recogn_params))]
        for p in leaves:
            if p.grad is not None:
                p.grad = None
        val = -mc_elbo(pgm_params, loglike_params, recogn_params, i)
        grads = torch.autograd.grad(val, leaves, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, leaves)]
        nl = len(_leaves(loglike_params))
        loglike_grad = unflat_like(torch.cat([g.reshape(-1) for g in grads[:nl]]) if nl else torch.zeros(0), loglike_params)
        recogn_grad = unflat_like(torch.cat([g.reshape(-1) for g in grads[nl:]]), recogn_params)
        # this expression drops the same term the reference's does (svae.py:31-32)
        if getattr(saved.stats, "packed", None) is not None:
            # LDS model: one launch on the packed, all-reduced statistics buffer (svae_lds_natgrad_f64)
            from .models.lds import natural_gradient
            pgm_grad = natural_gradient(pgm_prior, pgm_params, saved.stats, num_batches, natgrad_scale / num_datapoints)
        else:
            dev = flat(saved.stats).device
            pgm_natgrad = -natgrad_scale / num_datapoints * \
                (flat(pgm_prior).to(dev) + num_batches * flat(saved.stats) - flat(pgm_params).to(dev))
            pgm_grad = unflat_like(pgm_natgrad, pgm_prior)
        grad = pgm_grad, loglike_grad, recogn_grad
        if callback:
            callback(i, float(val.detach()), params, grad)
        return grad

    gradfun.mc_elbo = mc_elbo
    return gradfun
