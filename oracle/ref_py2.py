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
        raise NotImplementedError("autograd is not available; forward values only")

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
