"""Build the REFERENCE's own compiled E-step (Cython -> C -> .so) into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing under svae_amd/ may import anything under oracle/.

Recipe (adapted so that no reference source is copied into this repo):
the .pyx files are compiled *where they lie* under /root/reference; Cython's generated C and all
object files go to a temporary directory outside the repo; only the resulting extension modules
(`cython_lds_inference*.so`, `cython_hmm_inference*.so`) land in oracle/_ref/ (git-ignored, but
shipped to the GPU box by gpurun so bench.py can time the reference CPU path there).

Reference sources compiled:
  /root/reference/svae/lds/cython_lds_inference.pyx   (+ cython_gaussian_grads.pxd,
  /root/reference/svae/cython_util.pxd, /root/reference/svae/cython_linalg_grads.pxd)
  /root/reference/svae/hmm/cython_hmm_inference.pyx

Usage:  python oracle/build_ref.py        (no-op with exit code 0 if /root/reference is absent
                                           and prebuilt modules exist; exit 2 if neither)
"""
import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("SVAE_REFERENCE", "/root/reference")


def have_prebuilt():
    return bool(glob.glob(os.path.join(OUT, "cython_lds_inference*.so")))


def build(force=False):
    if have_prebuilt() and not force:
        return True
    if not os.path.isdir(os.path.join(REF, "svae")):
        return have_prebuilt()
    import numpy as np
    from Cython.Build import cythonize
    from setuptools import Extension
    from setuptools.dist import Distribution
    from setuptools.command.build_ext import build_ext

    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="svae_ref_build_")
    try:
        exts = [
            Extension("cython_lds_inference",
                      [os.path.join(REF, "svae/lds/cython_lds_inference.pyx")],
                      include_dirs=[np.get_include()],
                      extra_compile_args=["-O2", "-w"]),
            Extension("cython_hmm_inference",
                      [os.path.join(REF, "svae/hmm/cython_hmm_inference.pyx")],
                      include_dirs=[np.get_include()],
                      extra_compile_args=["-O2", "-w"]),
        ]
        # language_level=2: the reference is Python-2 era Cython (implicit relative cimports).
        exts = cythonize(
            exts, language_level=2, build_dir=os.path.join(tmp, "cy"), quiet=True,
            include_path=[REF, os.path.join(REF, "svae/lds"), os.path.join(REF, "svae/hmm")])
        dist = Distribution({"ext_modules": exts})
        cmd = build_ext(dist)
        cmd.build_lib = OUT
        cmd.build_temp = os.path.join(tmp, "obj")
        cmd.ensure_finalized()
        cmd.run()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return have_prebuilt()


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(OUT, "*.so"))))
    sys.exit(0 if ok else 2)
