"""NumPy restatement of the reference's SLDS-SVAE local inference.  TEST INFRASTRUCTURE.

Follows svae/models/slds_svae.py:80-310 for ONE sequence (the reference has no batch axis).

PARITY PINNED (round 3): the reference module itself runs through oracle/ref_py2.py:
load_reference_slds (its four dead import lines aliased to the modules that exist in the tree,
hmm_estep evaluated with the reference's compiled hmm_logZ / hmm_logZ_grad); its outputs for the
glue functions, a whole optimize_local_meanfield run and run_inference are committed as
tests/golden/slds_*.npz (tests/golden/make_golden.py: slds_case) and this restatement is held to
them in tests/test_oracle.py.

`cython_init_logZ`: the reference's compiled filter reads init_params[2] only
(cython_lds_inference.pyx:32), so with the SLDS's 4-tuple init potential (J, h, a, b) the term
b = 1/2 E log|J| never enters lds_vlb AS SHIPPED; the reference's Python twin sums the tail
(lds_inference.py:62-63).  True (default since round 5: the reference AS SHIPPED) reproduces the shipped value,
False the Python twin's.
"""
import numpy as np

from . import expfam_numpy as ef
from . import hmm_numpy, lds_numpy


def get_all_lds_local_natparams(lds_global_natparams):
    """slds_svae.py:86-89."""
    inits = [ef.unpack_dense(ef.niw_expectedstats(niw)) for niw, _ in lds_global_natparams]
    pairs = [ef.mniw_expectedstats(mniw) for _, mniw in lds_global_natparams]
    return inits, pairs


def get_var_lds_local_natparam(inits, pairs, expected_states):
    """slds_svae.py:92-103."""
    init = tuple(sum(w * np.asarray(p[i]) for w, p in zip(expected_states[0], inits)) for i in range(4))
    pair = tuple(np.stack([sum(w * np.asarray(p[i]) for w, p in zip(ws, pairs)) for ws in expected_states[1:]])
                 for i in range(4))
    return init, pair


def get_arhmm_local_nodeparams(inits, pairs, init_stats, pair_stats):
    """slds_svae.py:131-147."""
    ExxT0, Ex0 = init_stats
    T = pair_stats[0].shape[0] + 1
    K = len(inits)
    node = np.zeros((T, K))
    for k in range(K):
        J, h, a, b = inits[k]
        node[0, k] = np.sum(ExxT0 * J) + np.dot(Ex0, h) + a + b
        P = pairs[k]
        for t in range(T - 1):
            node[t + 1, k] = sum(np.sum(pair_stats[i][t] * P[i]) for i in range(3)) + P[3]
    return node


def lds_meanfield(inits, pairs, node_potentials, expected_states, cython_init_logZ=True):
    """slds_svae.py:80-84 -> (vlb, init_stats, pair_stats, node_stats, natparam)."""
    natparam = get_var_lds_local_natparam(inits, pairs, expected_states)
    est = (natparam[0][:3], natparam[1]) if cython_init_logZ else natparam
    lognorm, (E_init, E_pair, E_node) = lds_numpy.natural_lds_estep_general(est, node_potentials)
    return lognorm, (E_init[0], E_init[1]), E_pair[:3], E_node[:2], natparam


def initialize_local_meanfield(node_potentials, eps):
    """slds_svae.py:203-226; eps (T,1,n)."""
    n = node_potentials[1].shape[1]
    A = 0.9 * np.eye(n)
    natparam = ((-0.5 * np.eye(n), np.zeros(n), 0.),
                (-0.5 * A.T.dot(A), A.T, -0.5 * np.eye(n), 0.))
    node = lds_numpy._canonical_node_params(node_potentials)
    messages, _ = lds_numpy.natural_filter_forward_general(natparam[0], natparam[1], node)
    x = lds_numpy.natural_sample_backward_general(messages, natparam[1], eps)[:, 0]
    outer = lambda a, b: a[..., :, None] * b[..., None, :]
    return (outer(x[0], x[0]), x[0]), (outer(x[:-1], x[:-1]), outer(x[:-1], x[1:]), outer(x[1:], x[1:]))


def optimize_local_meanfield(global_natparam, node_potentials, init_eps, tol=1e-2, max_iter=100,
                             cython_init_logZ=True):
    """slds_svae.py:159-175."""
    (dir_nat, mdir_nat), lds_global = global_natparam
    hmm_init, hmm_pair = ef.dirichlet_expectedstats(dir_nat), ef.dirichlet_expectedstats(mdir_nat)
    inits, pairs = get_all_lds_local_natparams(lds_global)
    init_stats, pair_stats = initialize_local_meanfield(node_potentials, init_eps)
    vlb = -np.inf
    for it in range(1, max_iter + 1):
        node_hmm = get_arhmm_local_nodeparams(inits, pairs, init_stats, pair_stats)
        hmm_vlb, hmm_stats = hmm_numpy.hmm_estep((hmm_init, hmm_pair, node_hmm))
        lds_vlb, init_stats, pair_stats, node_stats, lds_nat = lds_meanfield(inits, pairs, node_potentials, hmm_stats[2],
                                                                             cython_init_logZ)
        new_vlb = hmm_vlb + lds_vlb
        if abs(new_vlb - vlb) < tol:
            break
        vlb = new_vlb
    return dict(hmm_stats=hmm_stats, init_stats=init_stats, pair_stats=pair_stats, node_stats=node_stats,
                lds_natparam=lds_nat, hmm_vlb=hmm_vlb, lds_vlb=lds_vlb, iters=it, node_hmm=node_hmm)


def get_global_stats(hmm_stats, init_stats, pair_stats):
    """slds_svae.py:229-243 for one sequence."""
    Ei, Et, Es = hmm_stats
    K = Es.shape[1]
    g_init = (np.stack([Es[0, k] * init_stats[0] for k in range(K)]), np.stack([Es[0, k] * init_stats[1] for k in range(K)]),
              Es[0].copy(), Es[0].copy())
    g_pair = tuple(np.einsum("tk,tij->kij", Es[1:], p) for p in pair_stats) + (Es[1:].sum(0),)
    return (Ei, Et), (g_init, g_pair)


def optimize_local_meanfield_withlabels(global_natparam, node_potentials, labels):
    """slds_svae.py:178-200."""
    (dir_nat, _), lds_global = global_natparam
    N = np.asarray(dir_nat).shape[0]
    labels = np.asarray(labels)
    ind = np.eye(N)[labels]
    trans = np.vstack([np.bincount(labels[1:][labels[:-1] == i], minlength=N) for i in range(N)])
    soft = ind + 1e-2
    hmm_stats = (ind[0], trans, soft / soft.sum(1, keepdims=True))
    inits, pairs = get_all_lds_local_natparams(lds_global)
    lds_vlb, init_stats, pair_stats, node_stats, lds_nat = lds_meanfield(inits, pairs, node_potentials, hmm_stats[2])
    return dict(hmm_stats=hmm_stats, init_stats=init_stats, pair_stats=pair_stats, node_stats=node_stats, lds_vlb=lds_vlb)
