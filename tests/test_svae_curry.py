"""make_gradfun keeps the reference's CURRIED call surface (/root/reference/svae/svae.py:10 `@curry`): the shipped
training script builds the gradient function in two stages (experiments/gmm_svae_synth.py:57-61).  CPU test: the
training-step contract itself is torch code; `run_inference` here is a toy conjugate model with the reference's
signature and return tuple."""
import pytest

torch = pytest.importorskip("torch")

from svae_amd.svae import curry, make_gradfun  # noqa: E402


def _toy():
    torch.manual_seed(0)
    data = torch.randn(40, 3, dtype=torch.float64)
    pgm_prior = (torch.ones(3, dtype=torch.float64), (torch.zeros(2, 2, dtype=torch.float64), torch.tensor(1.0, dtype=torch.float64)))
    pgm_params = (2 * torch.ones(3, dtype=torch.float64), (torch.eye(2, dtype=torch.float64), torch.tensor(0.5, dtype=torch.float64)))
    loglike_params = (torch.randn(3, 3, dtype=torch.float64, requires_grad=True),)
    recogn_params = (torch.randn(3, 3, dtype=torch.float64, requires_grad=True),)

    def recognize(recogn_params, batch):
        return batch @ recogn_params[0]

    def loglike(loglike_params, samples, batch):
        return -((samples @ loglike_params[0] - batch) ** 2).sum()

    def run_inference(pgm_prior, pgm_params, nn_potentials, num_samples):
        samples = nn_potentials * pgm_params[0]
        stats = (nn_potentials.detach().sum(0), (torch.ones(2, 2, dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64)))
        return samples, stats, torch.tensor(0.25, dtype=torch.float64), (nn_potentials ** 2).sum()

    return data, pgm_prior, (pgm_params, loglike_params, recogn_params), recognize, loglike, run_inference


def test_two_stage_call_of_the_shipped_training_script():
    data, pgm_prior_params, params, recognize, loglike, run_inference = _toy()
    seen = []
    plot = lambda i, val, params, grad: seen.append((i, val))
    # experiments/gmm_svae_synth.py:57 and :60, verbatim
    gradfun = make_gradfun(run_inference, recognize, loglike, pgm_prior_params, data)
    step = gradfun(batch_size=10, num_samples=1, natgrad_scale=1e4, callback=plot)
    grad = step(params, 0)
    assert len(grad) == 3 and seen and seen[0][0] == 0
    # the one-stage call gives the same gradient function
    step1 = make_gradfun(run_inference, recognize, loglike, pgm_prior_params, data, 10, 1, natgrad_scale=1e4,
                         callback=None, permute=False)
    step2 = make_gradfun(run_inference, recognize, loglike, pgm_prior_params, data)(10)(num_samples=1)  # noqa
    assert callable(step2)
    g1 = step1(params, 1)
    g2 = make_gradfun(run_inference)(recognize, loglike)(pgm_prior_params, data, batch_size=10, num_samples=1,
                                                         natgrad_scale=1e4, callback=None, permute=False)(params, 1)
    for a, b in zip(g1[0][0:1] + g1[1] + g1[2], g2[0][0:1] + g2[1] + g2[2]):
        assert torch.equal(a, b)


def test_curry_raises_for_calls_that_can_never_bind():
    f = curry(lambda a, b, c=3: (a, b, c))
    assert f(1)(2) == (1, 2, 3) and f(1, 2) == (1, 2, 3) and f(b=2)(1) == (1, 2, 3) and f(1)(b=5, c=7) == (1, 5, 7)
    with pytest.raises(TypeError):
        f(1, 2, 3, 4)
    with pytest.raises(TypeError):
        f(1, nope=2)
    with pytest.raises(TypeError):
        make_gradfun(None, None, None, None, None, bogus=1)
