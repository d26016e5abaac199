"""Run the reference's own *Python 2 + autograd* modules under Python 3, in memory.
TEST INFRASTRUCTURE ONLY; used here (where /root/reference exists) to generate golden fixtures.

The reference's pure-Python path (svae/models/gmm.py, svae/distributions/*.py, svae/lds/*.py,
svae/util.py) is Python-2-only and depends on autograd@0f026ab, neither of which exists in this
image.  This loader does NOT copy or port those sources: it reads each module
where it lies under /root/reference, rewrites the Python-2 syntax with the standard library's
lib2to3 fixers in memory, and executes it against a minimal stand-in for the few autograd/toolz
names the forward (non-differentiated) code paths touch: `autograd.numpy` is NumPy itself,
`primitive` / `getval` are identities, `grad` raises.  Values computed this way are the reference's
own arithmetic on NumPy 2.x.

    ref = load_reference()            # -> the `svae` package (sys.modules['svae'])
    from svae.models import gmm       # etc.
"""
import functools
import importlib.abc
import importlib.util
import inspect
import os
import sys
import types
import warnings

REF = os.environ.get("SVAE_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_shims():
    import numpy as np
    import scipy.linalg
    import scipy.special

    if "autograd" in sys.modules and getattr(sys.modules["autograd"], "_svae_shim", False):
        return

    def _no_grad(*a, **k):
        # grad(f) / value_and_grad(f) may be formed at import time (hmm_inference.py:65); only
        # CALLING the derivative is impossible here.
        def _raise(*a2, **k2):
            raise NotImplementedError("autograd is not available; forward values only")
        return _raise

    class _Prim(object):
        def __init__(self, f, aux=False):
            self.f, self.aux = f, aux
            functools.update_wrapper(self, f)

        def __call__(self, *a, **k):
            out = self.f(*a, **k)
            return out[0] if self.aux else out      # primitive_with_aux hides the aux output

        def defgrad(self, *a, **k):
            return None

        defgrads = defvjp = defgrad

    def flatten(struct):
        leaves = []

        def walk(s):
            if isinstance(s, (tuple, list)):
                return [walk(x) for x in s]
            a = np.asarray(s, dtype=float)
            leaves.append(a.ravel())
            return a.shape

        shapes = walk(struct)
        flat = np.concatenate(leaves) if leaves else np.zeros(0)

        def unflatten(v):
            pos = [0]

            def build(sh):
                if isinstance(sh, list):
                    return tuple(build(x) for x in sh)
                n = int(np.prod(sh)) if sh else 1
                out = np.reshape(v[pos[0]:pos[0] + n], sh)
                pos[0] += n
                return out
            return build(shapes)
        return flat, unflatten

    # `autograd.numpy` = NumPy, with the NumPy-1.x rule of linalg.solve the reference was written
    # against (b is a stack of VECTORS when b.ndim == a.ndim - 1; NumPy 2 changed that), used at
    # distributions/gaussian.py:14,22 and distributions/mniw.py:45.
    def solve_np1(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if b.ndim == a.ndim - 1:
            return np.linalg.solve(a, b[..., None])[..., 0]
        return np.linalg.solve(a, b)

    linalg1 = types.ModuleType("autograd.numpy.linalg")
    linalg1.__dict__.update({k: v for k, v in np.linalg.__dict__.items() if not k.startswith("__")})
    linalg1.solve = solve_np1
    np1 = types.ModuleType("autograd.numpy")
    np1.__dict__.update({k: v for k, v in np.__dict__.items() if not k.startswith("__")})
    np1.linalg = linalg1
    if "object" not in np.__dict__:
        np1.object = object        # util.py:169 names np.object (removed in NumPy 1.24)
    sys.modules["autograd.numpy"] = np1
    sys.modules["autograd.numpy.linalg"] = linalg1
    sys.modules["autograd.numpy.random"] = np.random

    ag = _mod("autograd", grad=_no_grad, value_and_grad=_no_grad, _svae_shim=True)
    ag.numpy = np1
    agsp = _mod("autograd.scipy")
    agsp.special = _mod("autograd.scipy.special", digamma=scipy.special.digamma,
                        gammaln=scipy.special.gammaln, multigammaln=scipy.special.multigammaln)
    agsp.misc = _mod("autograd.scipy.misc", logsumexp=scipy.special.logsumexp)
    sys.modules["autograd.scipy.linalg"] = scipy.linalg
    agsp.linalg = scipy.linalg
    ag.scipy = agsp
    ag.util = _mod("autograd.util", flatten=flatten, make_tuple=lambda *a: tuple(a))
    ag.core = _mod("autograd.core", getval=lambda x: x, primitive=lambda f: _Prim(f),
                   primitive_with_aux=lambda f: _Prim(f, aux=True))
    ag.container_types = _mod("autograd.container_types",
                              TupleNode=type("TupleNode", (), {}), ListNode=type("ListNode", (), {}))
    ag.convenience_wrappers = _mod("autograd.convenience_wrappers", grad_and_aux=_no_grad,
                                   value_and_grad=_no_grad)
    ag.optimizers = _mod("autograd.optimizers", sgd=_no_grad, adam=_no_grad)

    def curry(f):
        nreq = len([p for p in inspect.signature(f).parameters.values()
                    if p.default is p.empty and p.kind == p.POSITIONAL_OR_KEYWORD])

        def curried(*a, **k):
            if len(a) + len([x for x in k if x in inspect.signature(f).parameters]) >= nreq:
                return f(*a, **k)
            return lambda *a2, **k2: curried(*(a + a2), **dict(k, **k2))
        return functools.wraps(f)(curried)
    if "toolz" not in sys.modules:
        _mod("toolz", curry=curry)


# Python-2 constructs lib2to3 cannot rewrite (a lambda with TWO tuple parameters); replaced, in
# memory, by the equivalent single-parameter form before the fixers run.  Syntax only.
_PRE_SUBS = {
    "svae/lds/lds_inference.py": [
        ("sample = lambda (J11, J12), (J_filt, h_filt): lambda next_sample: \\\n"
         "        natural_sample(*natural_condition_on(J_filt, h_filt, next_sample, J11, J12))",
         "sample = lambda _p, _f: lambda next_sample: \\\n"
         "        natural_sample(*natural_condition_on(_f[0], _f[1], next_sample, _p[0], _p[1]))"),
    ],
}


class _Py2Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Imports svae.* from /root/reference through lib2to3, without touching the disk."""

    def __init__(self, root):
        self.root = root
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from lib2to3 import refactor
        fixers = refactor.get_fixers_from_package("lib2to3.fixes")
        self.tool = refactor.RefactoringTool(fixers)

    def _path(self, fullname):
        rel = fullname.replace(".", "/")
        for cand, ispkg in ((rel + "/__init__.py", True), (rel + ".py", False)):
            p = os.path.join(self.root, cand)
            if os.path.isfile(p):
                return p, ispkg
        return None, False

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "svae" and not fullname.startswith("svae."):
            return None
        p, ispkg = self._path(fullname)
        if p is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, origin=p, is_package=ispkg)

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        p = module.__spec__.origin
        if module.__spec__.submodule_search_locations is not None:
            module.__path__ = [os.path.dirname(p)]
        src = open(p).read()
        if not src.endswith("\n"):
            src += "\n"
        for old, new in _PRE_SUBS.get(os.path.relpath(p, self.root), ()):
            assert old in src, (p, old)
            src = src.replace(old, new)
        py3 = str(self.tool.refactor_string(src, p)) if src.strip() else ""
        exec(compile(py3, p, "exec"), module.__dict__)


def load_reference(with_cython=True):
    if not os.path.isdir(os.path.join(REF, "svae")):
        raise ImportError("reference tree not present at %s" % REF)
    _install_shims()
    if not any(isinstance(f, _Py2Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Py2Finder(REF))
    if with_cython:
        # svae/lds/lds_inference.py:18-24 imports the compiled module; hand it the one built by
        # oracle/build_ref.py from the very same .pyx.
        from . import ref as _ref
        sys.modules.setdefault("svae.lds.cython_lds_inference", _ref._load("cython_lds_inference"))
        sys.modules.setdefault("svae.hmm.cython_hmm_inference", _ref._load("cython_hmm_inference"))
    import svae
    return svae


def load_reference_slds():
    """The reference's svae/models/slds_svae.py (and svae/models/lds.py), executed as shipped.

    Both modules fail to import in the reference tree itself because of four DEAD import lines, not
    because of anything they compute with:
      * `from svae.lds import niw, mniw` (slds_svae.py:16, models/lds.py:12): the exp-family modules
        live in svae/distributions/ in this tree -> aliased to svae.distributions.{niw,mniw}.
      * `from svae.hmm import dirichlet` (slds_svae.py:18) -> svae.distributions.dirichlet.
      * `from lds_svae import lds_prior_expectedstats` (slds_svae.py:19): no such module.  The name
        is rebuilt from the reference's own functions with the reference's own definition
        (svae/lds/lds_prior.py:7-9, models/lds.py:24-26: `niw.expectedstats, mniw.expectedstats`),
        composed with the reference's gaussian.unpack_dense (distributions/gaussian.py:47-57)
        because the LDS code consumes init params as the tuple (J, h, a, b)
        (cython_lds_inference.pyx:30-32) while distributions/niw.py:25 returns them dense-packed.
      * `pyhsmm.internals.hmm_messages_interface` (hmm_inference.py:8-10; un-vendored, un-pinned):
        stand-in names that raise; `hmm_estep` is then re-bound inside slds_svae to the
        reference's own `hmm_estep_slow = vgrad(hmm_logZ)` (hmm_inference.py:65) evaluated with
        the COMPILED hmm_logZ / hmm_logZ_grad(1.0, .) of cython_hmm_inference.pyx:93-166.
    Everything else -- get_var_lds_local_natparam (:92-103), hmm_prior_expectedstats (:120-128),
    get_arhmm_local_nodeparams (:131-147), optimize_local_meanfield (:159-175),
    initialize_local_meanfield (:203-226), get_global_stats (:229-243), run_inference
    (:289-310, forward values) -- is the reference's code on the reference's compiled kernels.
    Returns the module svae.models.slds_svae."""
    import numpy as np
    svae = load_reference(with_cython=True)
    from . import ref as _ref
    import svae.distributions.niw as niw
    import svae.distributions.mniw as mniw
    import svae.distributions.dirichlet as dirichlet
    import svae.distributions.gaussian as dgauss
    import svae.lds
    import svae.hmm
    sys.modules["svae.lds.niw"] = svae.lds.niw = niw
    sys.modules["svae.lds.mniw"] = svae.lds.mniw = mniw
    sys.modules["svae.hmm.dirichlet"] = svae.hmm.dirichlet = dirichlet
    chmm = _ref._load("cython_hmm_inference")
    sys.modules.setdefault("cython_hmm_inference", chmm)       # hmm_inference.py:12 (py2 implicit relative)

    def _absent(*a, **k):
        raise NotImplementedError("pyhsmm is not vendored by the reference")
    _mod("pyhsmm")
    _mod("pyhsmm.internals")
    _mod("pyhsmm.internals.hmm_messages_interface", messages_backwards_log=_absent,
         messages_forwards_log=_absent, expected_statistics_log=_absent, viterbi=_absent)

    def lds_prior_expectedstats(natparam):
        niw_natparam, mniw_natparam = natparam
        J, h, a, b = dgauss.unpack_dense(niw.expectedstats(niw_natparam))
        return (J, h, a, b), tuple(mniw.expectedstats(mniw_natparam))
    _mod("lds_svae", lds_prior_expectedstats=lds_prior_expectedstats)

    import svae.models.slds_svae as slds

    def hmm_estep(natparam):
        natparam = tuple(np.require(x, np.double, "C") for x in natparam)
        logZ, aux = chmm.hmm_logZ(natparam)
        E_init, E_trans, E_states = chmm.hmm_logZ_grad(1.0, aux)
        return logZ, (np.asarray(E_init), np.asarray(E_trans), np.asarray(E_states))
    slds.hmm_estep = hmm_estep
    return slds
