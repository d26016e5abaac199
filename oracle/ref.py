"""Thin loader for the REFERENCE's own compiled E-step built by oracle/build_ref.py.
TEST INFRASTRUCTURE ONLY (checker + CPU baseline).  Never imported by svae_amd/.

Wraps the functions imported at /root/reference/svae/lds/lds_inference.py:18-24 exactly as
`cython_natural_lds_estep_general` (lds_inference.py:232-237) composes them; the only change is
materialising Python-3 `map` objects (cython_lds_inference.pyx:23,208 were written for Python 2).
"""
import glob
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_mods = {}


def available():
    return bool(glob.glob(os.path.join(_HERE, "_ref", "cython_lds_inference*.so")))


def _load(name):
    if name not in _mods:
        paths = glob.glob(os.path.join(_HERE, "_ref", name + "*.so"))
        if not paths:
            raise ImportError("oracle/_ref/%s*.so not built; run python oracle/build_ref.py" % name)
        spec = importlib.util.spec_from_file_location(name, paths[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _mods[name] = mod
    return _mods[name]


def filter_forward(init_params, pair_params, node_params):
    """cython_lds_inference.pyx:28-90.  init_params must be the 3-tuple (J, h, logZ)."""
    m = _load("cython_lds_inference")
    (messages, lognorm), aux = m.natural_filter_forward_general(init_params, pair_params, node_params)
    return messages, lognorm, aux


def smoother(forward_messages, pair_params):
    """cython_lds_inference.pyx:149-210."""
    m = _load("cython_lds_inference")
    (E_init, E_pair, E_node), aux = m.natural_smoother_general(forward_messages, pair_params)
    return (tuple(E_init), tuple(E_pair), tuple(E_node)), aux


def estep(natparam, node_params):
    """cython_natural_lds_estep_general, lds_inference.py:232-237 -> (lognorm, stats)."""
    init_params, pair_params = natparam
    init_params = (init_params[0], init_params[1], sum(init_params[2:]))
    messages, lognorm, _ = filter_forward(init_params, pair_params, node_params)
    stats, _ = smoother(messages, pair_params)
    return lognorm, stats


def hmm_logZ(natparam):
    """cython_hmm_inference.pyx:93-121."""
    m = _load("cython_hmm_inference")
    return m.hmm_logZ(natparam)


def hmm_logZ_grad(g, aux):
    """cython_hmm_inference.pyx:126-166; at g=1 this is the HMM E-step (hmm_inference.py:65)."""
    m = _load("cython_hmm_inference")
    return m.hmm_logZ_grad(g, aux)


def sample_backward(natparam, node_params, num_samples, seed):
    """cython_natural_lds_sample (lds_inference.py:260-264): filter + natural_sample_backward
    (cython_lds_inference.pyx:310-355).  The compiled sampler draws `flipud(randn(T,S,N))` from the
    global NumPy RNG (:333); we seed it and return, next to the samples (T,S,N), the noise it used,
    re-indexed so that eps[t] is the draw applied at time t."""
    import numpy as np
    m = _load("cython_lds_inference")
    init_params, pair_params = natparam
    init_params = (init_params[0], init_params[1], sum(init_params[2:]))
    messages, _, _ = filter_forward(init_params, pair_params, node_params)
    T, N = np.asarray(node_params[1]).shape
    np.random.seed(seed)
    samples, _ = m.natural_sample_backward(messages, pair_params, num_samples)
    np.random.seed(seed)
    eps = np.random.randn(T, num_samples, N)[::-1].copy()
    return np.asarray(samples), eps


def estep_vjp(natparam, node_params, g_lognorm, g_E_node, g_samples=None, seed=0, g_E_init=None,
              g_E_pair=None):
    """Composite reverse-mode derivative w.r.t. node_params, wired exactly as the reference wires its
    autograd primitives (svae/lds/lds_inference.py:26-39):
        natural_filter_grad            cython_lds_inference.pyx:92-145   (argnum 2 = node_params)
        natural_smoother_general_grad  cython_lds_inference.pyx:236-306  (argnum 0 = messages)
        natural_sample_backward_grad   cython_lds_inference.pyx:357-409  (argnum 0 = messages)
    g_E_node = (g_diagExxT (T,n), g_Ex (T,n)); g_E_init = (g_ExxT0 (n,n), g_Ex0 (n)) and g_E_pair =
    (g0, g1, g2), each (n,n) or per step (T-1,n,n), default to zero (the LDS model code never
    differentiates them: svae.py:21 stores them in `saved.stats`; the SLDS does, slds_svae.py:300).
    Returns (g_node_J, g_node_h, g_node_logZ) and, if sampling, the eps used (re-indexed by time)."""
    import numpy as np
    m = _load("cython_lds_inference")
    init_params, pair_params = natparam
    init_params = (init_params[0], init_params[1], sum(init_params[2:]))
    (messages, lognorm), aux_f = m.natural_filter_forward_general(init_params, pair_params, node_params)
    T, n = np.asarray(node_params[1]).shape
    (E_init, E_pair, E_node), aux_s = m.natural_smoother_general(messages, pair_params)
    cp = lambda x: np.array(x, dtype=float, copy=True)
    gi = (np.zeros((n, n)), np.zeros(n)) if g_E_init is None else (cp(g_E_init[0]), cp(g_E_init[1]))
    gp = (np.zeros((n, n)),) * 3 if g_E_pair is None else tuple(cp(x) for x in g_E_pair[:3])
    g_stats = ((gi[0], gi[1], 0., 0.),
               (cp(gp[0]), cp(gp[1]), cp(gp[2]), 0.),
               (np.array(g_E_node[0], dtype=float, copy=True), np.array(g_E_node[1], dtype=float, copy=True),
                np.zeros(T)))   # copies: _compute_stats_grad (:212-234) accumulates into its inputs
    (gJp, ghp), (gJf, ghf) = m.natural_smoother_general_grad(g_stats, aux_s)
    gJp, ghp, gJf, ghf = [np.array(x, dtype=float, copy=True) for x in (gJp, ghp, gJf, ghf)]
    eps = None
    if g_samples is not None:
        S = g_samples.shape[1]
        np.random.seed(seed)
        samples, aux_x = m.natural_sample_backward(messages, pair_params, S)
        np.random.seed(seed)
        eps = np.random.randn(T, S, n)[::-1].copy()
        (aJp, ahp), (aJf, ahf) = m.natural_sample_backward_grad(np.array(g_samples, dtype=float, copy=True), aux_x)
        gJp += aJp; ghp += ahp; gJf += aJf; ghf += ahf
    g = (((gJp, ghp), (gJf, ghf)), float(g_lognorm))
    gJ, gh, gz = m.natural_filter_grad(g, aux_f)
    return (np.asarray(gJ), np.asarray(gh), np.asarray(gz)), eps
