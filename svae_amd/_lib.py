"""ctypes binding of libsvae_hip.so (C ABI declared in include/svae_hip.h).

The library must be loaded AFTER torch so that its DT_NEEDED ``libamdhip64.so.7`` resolves to the
HIP runtime torch already mapped (one runtime per process: torch's streams and device pointers are
handed straight to the kernels).  Missing library => ImportError; there is no fallback path.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVAE_AMD_LIB", os.path.join(_HERE, "libsvae_hip.so"))  # env: experiments only

ABI_VERSION = 14
LDS_MAX_N = 15        # register/DPP path (E-step, sampler, VJP)
LDS_TILE_MAX_N = 64   # LDS-tiled MFMA path (E-step only)

# per-call `options` word of the LDS entry points (include/svae_hip.h, SVAE_OPT_*): 0 = the library's choice
OPT_DEFAULT, OPT_TWOEND_OFF, OPT_TWOEND_FULL = 0x00, 0x01, 0x02
OPT_LAYOUT_SPLIT, OPT_LAYOUT_PACKED, OPT_PRODUCERS_ON, OPT_PRODUCERS_OFF = 0x04, 0x08, 0x10, 0x20
OPT_TILE_FORWARD, OPT_TILE_BACKWARD = 0x40, 0x80     # 16 <= n <= 64 only: one half of the E-step per call
OPT_LEAN_ON, OPT_LEAN_OFF, OPT_INFER_RECORDS = 0x100, 0x200, 0x400   # record format of svae_lds_inference_f64 / its VJP
KEEP_SIGMA = 4                                       # keep bit of svae_lds_estep_f64, 16 <= n <= 64 only (SVAE_KEEP_SIGMA)
# names used by tests / tools / bench.py --kernel for the E-step kernel families
KERNEL_OPTIONS = {"auto": OPT_DEFAULT, "twoend": OPT_DEFAULT, "twoend_full": OPT_TWOEND_FULL,
                  "twoend_seq": OPT_LAYOUT_SPLIT, "twoend_rpc": OPT_LAYOUT_PACKED,
                  "split": OPT_TWOEND_OFF | OPT_LAYOUT_SPLIT, "packed": OPT_TWOEND_OFF | OPT_LAYOUT_PACKED}

_c_double_p = ctypes.c_void_p   # raw device pointers travel as integers
_c_int_p = ctypes.c_void_p

# name -> (restype, argtypes).  Must list every symbol of include/svae_hip.h (tests check this).
SIGNATURES = {
    "svae_hip_abi_version": (ctypes.c_int, []),
    "svae_lds_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "svae_lds_workspace_bytes_ex": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "svae_lds_tile_sigma_offset_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "svae_lds_estep_f64": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_uint] + [_c_double_p] * 15
                           + [_c_int_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_lds_filter_f64": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.c_uint] + [_c_double_p] * 15
                            + [_c_int_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_slds_lds_meanfield_lds_bytes": (ctypes.c_size_t, [ctypes.c_int] * 2),
    "svae_slds_lds_meanfield_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "svae_slds_lds_meanfield_f64": (ctypes.c_int, [ctypes.c_int] * 5 + [_c_double_p] * 9 + [_c_int_p]
                                    + [_c_double_p] * 5 + [_c_int_p, ctypes.c_void_p, ctypes.c_size_t,
                                                           ctypes.c_uint, ctypes.c_void_p]),
    "svae_lds_reduce_stats_f64": (ctypes.c_int, [ctypes.c_int] * 2 + [_c_double_p] * 4
                                  + [ctypes.c_void_p]),
    "svae_lds_vjp_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "svae_lds_estep_vjp_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 9
                               + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                  ctypes.c_void_p]),
    "svae_lds_estep_vjp_ex_f64": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_uint] + [_c_double_p] * 13
                                  + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.c_void_p]),
    "svae_lds_estep_vjp_dense_f64": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_uint] + [_c_double_p] * 14
                                     + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p]),
    "svae_lds_inference_is_lean": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_uint]),
    "svae_lds_inference_f64": (ctypes.c_int, [ctypes.c_int] * 7 + [ctypes.c_uint] + [_c_double_p] * 17
                               + [_c_int_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_lds_sample_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.c_uint] + [_c_double_p] * 2
                            + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_hmm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "svae_hmm_estep_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 7
                           + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_slds_hmm_meanfield_f64": (ctypes.c_int, [ctypes.c_int] * 5 + [_c_double_p] * 9 + [_c_int_p] + [_c_double_p] * 5
                                    + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_slds_sweep_glue_f64": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_double, _c_int_p] + [_c_double_p] * 7
                                 + [_c_int_p] * 4 + [ctypes.c_void_p]),
    "svae_slds_path_nodeparams_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 9 + [ctypes.c_void_p]),
    "svae_slds_mix_pair_natparam_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 9 + [ctypes.c_void_p]),
    "svae_slds_pair_contract_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 6 + [ctypes.c_int, ctypes.c_void_p]),
    "svae_gmm_meanfield_f64": (ctypes.c_int, [ctypes.c_int] * 3 + [_c_double_p] * 5
                               + [ctypes.c_double, ctypes.c_int] + [_c_double_p] * 8
                               + [_c_int_p] * 3 + [ctypes.c_void_p]),
    "svae_lds_tile_vjp_workspace_doubles": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "svae_lds_tile_vjp_f64": (ctypes.c_int, [ctypes.c_int] * 9 + [_c_double_p] * 11
                              + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_lds_tile_noise_f64": (ctypes.c_int, [ctypes.c_int] * 7 + [_c_double_p] * 2
                                + [ctypes.c_void_p, ctypes.c_void_p, _c_int_p, ctypes.c_void_p]),
    "svae_lds_tile_sample_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 2 + [ctypes.c_void_p, ctypes.c_void_p]),
    "svae_lds_global_step_f64": (ctypes.c_int, [ctypes.c_int] + [_c_double_p] * 19 + [_c_int_p, ctypes.c_void_p]),
    "svae_lds_diag_sample_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "svae_lds_diag_sample_f64": (ctypes.c_int, [ctypes.c_int] * 3 + [_c_double_p] * 9 + [_c_int_p, ctypes.c_void_p, ctypes.c_size_t,
                                                                                   ctypes.c_void_p]),
    "svae_lds_global_step_multi_f64": (ctypes.c_int, [ctypes.c_int] * 2 + [ctypes.POINTER(ctypes.c_void_p)] * 5 +
                                       [_c_double_p] * 8 + [_c_int_p, ctypes.c_void_p]),
    "svae_lds_natgrad_f64": (ctypes.c_int, [ctypes.c_int] * 2 + [_c_double_p] * 3 + [ctypes.c_double] * 2
                             + [_c_double_p, ctypes.c_void_p]),
    "svae_gmm_mw_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "svae_gmm_mw_begin": (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_gmm_mw_step_f64": (ctypes.c_int, [ctypes.c_int] * 5 + [_c_double_p] * 5
                             + [ctypes.c_double, ctypes.c_int] + [_c_double_p] * 8
                             + [_c_int_p] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "svae_gmm_mw_kl_hist": (ctypes.c_void_p, [ctypes.c_void_p]),
    "svae_ipc_mailbox_bytes": (ctypes.c_size_t, [ctypes.c_int] * 2),
    "svae_ipc_allreduce_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.c_uint] * 2 + [_c_double_p] * 2
                               + [ctypes.c_void_p, _c_int_p, ctypes.c_void_p]),
    "svae_gmm_global_step_f64": (ctypes.c_int, [ctypes.c_int] * 2 + [_c_double_p] * 7 + [_c_int_p, ctypes.c_void_p]),
    "svae_gmm_sample_f64": (ctypes.c_int, [ctypes.c_int] * 3 + [_c_double_p] * 3 + [ctypes.c_void_p]),
    "svae_gmm_local_vjp_f64": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_double_p] * 11 + [ctypes.c_void_p]),
    "svae_gmm_mw_fixed_point_f64": (ctypes.c_int, [ctypes.c_int] * 3 + [_c_double_p] * 5
                                    + [ctypes.c_double, ctypes.c_int] + [_c_double_p] * 8
                                    + [_c_int_p] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises ImportError if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(
                "svae_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C svae_amd/csrc -j8` (there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError => stale library: fail loudly
            fn.restype, fn.argtypes = res, args
        if lib.svae_hip_abi_version() != ABI_VERSION:
            raise ImportError("svae_amd: libsvae_hip.so ABI %d != expected %d"
                              % (lib.svae_hip_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: rc=%d (%s)" % (
            what, rc, "bad argument #%d" % -rc if -100 < rc < 0 else "launch error"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def source_hash():
    """First 16 hex digits of the SHA-256 over the library's sources (svae_amd/csrc/*.hip, *.hpp, the Makefile and
    include/svae_hip.h, in name order): what bench.py compares with the hash stored in a committed rocprofv3 PMC
    summary before quoting its traffic figures -- a kernel change without a re-profile must not quote stale counters."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.hpp")))
    files += [os.path.join(here, "csrc", "Makefile"), os.path.join(os.path.dirname(here), "include", "svae_hip.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
