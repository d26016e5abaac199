"""NumPy restatement of the model-level glue around the E-step.  TEST INFRASTRUCTURE.

  LDS:  run_inference / lds_prior_kl / lds_prior_expectedstats / lds_prior_logZ
        svae/models/lds.py:16-52  (that module is stale as shipped -- it imports svae.lds.niw/mniw,
        which do not exist -- so it cannot be executed; the formulas are restated with
        svae/distributions/{niw,mniw}.py, as the live svae/models/gmm.py does for the GMM)
  GMM:  prior_kl  svae/models/gmm.py:44-58   (run_inference itself is pinned by golden vectors
        generated from the reference's own gmm.py, tests/golden/gmm_run_*.npz)
"""
import numpy as np

from . import expfam_numpy as ef
from . import lds_numpy


def lds_prior_expectedstats(natparam):
    return ef.niw_expectedstats(natparam[0]), ef.mniw_expectedstats(natparam[1])


def lds_prior_logZ(natparam):
    return ef.niw_logZ(natparam[0]) + ef.mniw_logZ(natparam[1])


def _contract(a, b):
    if isinstance(a, (tuple, list)):
        return sum(_contract(x, y) for x, y in zip(a, b))
    return float(np.sum(np.asarray(a) * np.asarray(b)))


def _sub(a, b):
    if isinstance(a, (tuple, list)):
        return tuple(_sub(x, y) for x, y in zip(a, b))
    return np.asarray(a) - np.asarray(b)


def lds_prior_kl(global_natparam, prior_natparam, expected_stats=None):
    """lds.py:16-20."""
    if expected_stats is None:
        expected_stats = lds_prior_expectedstats(global_natparam)
    return -_contract(_sub(prior_natparam, global_natparam), expected_stats) \
        + (lds_prior_logZ(prior_natparam) - lds_prior_logZ(global_natparam))


def lds_run_inference(prior_natparam, global_natparam, nn_potentials, eps):
    """lds.py:35-42 for ONE sequence; eps (T,S,n) is the sampler's noise.  Returns
    (samples (T,S,n), (E_init_stats, E_pair_stats), global_kl, local_kl)."""
    es = lds_prior_expectedstats(global_natparam)
    init_params = ef.unpack_dense(es[0])
    local_natparam = (init_params, es[1])
    node = lds_numpy._canonical_node_params(nn_potentials)
    messages, lognorm = lds_numpy.natural_filter_forward_general(init_params, es[1], node)
    E_init, E_pair, E_node = lds_numpy.natural_smoother_general(messages, es[1])
    samples = lds_numpy.natural_sample_backward_general(messages, es[1], eps)
    local_kl = _contract(node, E_node) - lognorm
    global_kl = lds_prior_kl(global_natparam, prior_natparam, es)
    return samples, (E_init, E_pair), global_kl, local_kl


def gmm_prior_kl(global_natparam, prior_natparam):
    """gmm.py:54-58."""
    es = (ef.dirichlet_expectedstats(global_natparam[0]), ef.niw_expectedstats(global_natparam[1]))
    diff = sum(np.sum((np.asarray(g) - np.asarray(p)) * e)
               for g, p, e in zip(global_natparam, prior_natparam, es))
    logZ = lambda q: ef.dirichlet_logZ(q[0]) + ef.niw_logZ(q[1])
    return diff - (logZ(global_natparam) - logZ(prior_natparam))
