"""CPU oracle for the SVAE structured E-step hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain NumPy, the algorithm of the reference
(mattjj/svae) for the natural-parameter LDS E-step and the GMM mean-field update, and wraps the
reference's own compiled Cython path when it has been built into ``oracle/_ref/``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the CPU baseline.  The product package ``svae_amd`` never does.

Parity status
-------------
* LDS E-step (filter + smoother + expected stats): PINNED.  The reference ships no golden vectors
  for this path, so the restatement in ``lds_numpy.py`` is pinned against
  (1) the reference's own compiled implementation (``oracle/_ref/cython_lds_inference*.so`` built
  from ``/root/reference/svae/lds/cython_lds_inference.pyx`` by ``oracle/build_ref.py``), both live
  in this container and through the committed fixtures in ``tests/golden/`` generated from it by
  ``tests/golden/make_golden.py``, and (2) a brute-force dense (Tn x Tn) joint-Gaussian solve.
* GMM mean field and the exponential-family maps: pinned against outputs of the reference's own
  Python modules executed in this container through a lib2to3 + autograd-shim loader
  (``oracle/ref_py2.py``), committed as fixtures by ``tests/golden/make_golden.py``.
"""
