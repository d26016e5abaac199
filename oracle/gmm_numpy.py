"""Plain-NumPy restatement of the reference's GMM mean-field E-step.  TEST INFRASTRUCTURE.

  local_meanfield        svae/models/gmm.py:62-88
  meanfield_fixed_point  svae/models/gmm.py:90-110
  gaussian_meanfield     svae/models/gmm.py:112-117
  label_meanfield        svae/models/gmm.py:119-124
  initialize_meanfield   svae/models/gmm.py:126-128  (RNG injected: pass `label_init`)

The reference draws the initial responsibilities from the global NumPy RNG inside
`initialize_meanfield`; here they are an explicit argument so that the oracle, the reference and
the HIP kernel can be run on identical inputs.
"""
import numpy as np
from . import expfam_numpy as ef

# util.py:45 -- tensordot over the trailing `axes` axes of both arguments
_pflat = lambda a, axes: np.reshape(a, a.shape[:-axes] + (-1,))


def gaussian_meanfield(gaussian_globals, node_potentials, label_stats):
    global_potentials = np.tensordot(label_stats, gaussian_globals, [1, 0])
    natparam = node_potentials + global_potentials
    stats = ef.gaussian_expectedstats(natparam)
    kl = np.tensordot(node_potentials, stats, 3) - ef.gaussian_logZ(natparam)
    return natparam, stats, kl


def label_meanfield(label_global, gaussian_globals, gaussian_stats):
    node_potentials = np.tensordot(gaussian_stats, gaussian_globals, [[1, 2], [1, 2]])
    natparam = node_potentials + label_global
    stats = ef.categorical_expectedstats(natparam)
    kl = np.tensordot(stats, node_potentials) - ef.categorical_logZ(natparam)
    return natparam, stats, kl


def meanfield_fixed_point(label_global, gaussian_globals, node_potentials, label_init,
                          tol=1e-3, max_iter=100, return_iters=False):
    kl = np.inf
    label_stats = label_init
    it = 0
    for i in range(max_iter):
        it = i + 1
        gaussian_natparam, gaussian_stats, gaussian_kl = \
            gaussian_meanfield(gaussian_globals, node_potentials, label_stats)
        label_natparam, label_stats, label_kl = \
            label_meanfield(label_global, gaussian_globals, gaussian_stats)
        # recompute gaussian_kl linear term with new label_stats b/c labels were updated
        gaussian_global_potentials = np.tensordot(label_stats, gaussian_globals, [1, 0])
        linear_difference = gaussian_natparam - gaussian_global_potentials - node_potentials
        gaussian_kl = gaussian_kl + np.tensordot(linear_difference, gaussian_stats, 3)
        kl, prev_kl = label_kl + gaussian_kl, kl
        if abs(kl - prev_kl) < tol:
            break
    return (label_stats, it) if return_iters else label_stats


def local_meanfield(label_global, gaussian_globals, node_potentials, label_init,
                    tol=1e-3, max_iter=100):
    """gmm.py:62-88 with the two global->local maps (dirichlet/niw expectedstats, :67-68) already
    applied by the caller, and node_potentials = (J diag (T,N), h (T,N))."""
    node_potentials = ef.pack_dense(*node_potentials)
    label_stats, iters = meanfield_fixed_point(label_global, gaussian_globals, node_potentials,
                                               label_init, tol, max_iter, return_iters=True)
    gaussian_natparam, gaussian_stats, gaussian_kl = \
        gaussian_meanfield(gaussian_globals, node_potentials, label_stats)
    label_natparam, label_stats, label_kl = \
        label_meanfield(label_global, gaussian_globals, gaussian_stats)
    dirichlet_stats = np.sum(label_stats, 0)
    niw_stats = np.tensordot(label_stats, gaussian_stats, [0, 0])
    local_stats = label_stats, gaussian_stats
    prior_stats = dirichlet_stats, niw_stats
    natparam = label_natparam, gaussian_natparam
    kl = label_kl + gaussian_kl
    return local_stats, prior_stats, natparam, kl, iters
