"""NumPy restatement of the reference's HMM log-normaliser and E-step.  TEST INFRASTRUCTURE.

  hmm_logZ_python  svae/hmm/hmm_inference.py:44-51  (= cython hmm_logZ, cython_hmm_inference.pyx:93-121)
  E-step = gradient of hmm_logZ w.r.t. its natural parameters (hmm_inference.py:65
  `hmm_estep_slow = vgrad(hmm_logZ)`; compiled: hmm_logZ_grad, cython_hmm_inference.pyx:126-166),
  restated here as the classic log-space forward-backward.
"""
import numpy as np
from scipy.special import logsumexp


def hmm_logZ(natparam):
    init_params, pair_params, node_params = natparam
    log_alpha = init_params + node_params[0]
    for node_param in node_params[1:]:
        log_alpha = logsumexp(log_alpha[:, None] + pair_params, axis=0) + node_param
    return logsumexp(log_alpha)


def hmm_estep(natparam):
    init_params, pair_params, node_params = (np.asarray(x, float) for x in natparam)
    T, K = node_params.shape
    la = np.zeros((T, K)); lb = np.zeros((T, K))
    la[0] = init_params + node_params[0]
    for t in range(1, T):
        la[t] = logsumexp(la[t - 1][:, None] + pair_params, axis=0) + node_params[t]
    for t in range(T - 2, -1, -1):
        lb[t] = logsumexp(pair_params + (node_params[t + 1] + lb[t + 1])[None, :], axis=1)
    logZ = logsumexp(la[-1])
    E_states = np.exp(la + lb - logZ)
    E_trans = np.zeros((K, K))
    for t in range(T - 1):
        E_trans += np.exp(la[t][:, None] + pair_params + (node_params[t + 1] + lb[t + 1])[None, :] - logZ)
    return logZ, (E_states[0], E_trans, E_states)
