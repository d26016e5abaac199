"""CPU tests of the drop-in boundary: libsvae_hip.so loads without a GPU, exports every symbol
include/svae_hip.h declares, and rejects bad arguments before touching the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "svae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svae_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = _header_symbols()
    for s in ("svae_lds_estep_f64", "svae_lds_workspace_bytes", "svae_lds_workspace_bytes_ex", "svae_lds_filter_f64",
              "svae_lds_reduce_stats_f64",
              "svae_lds_sample_f64", "svae_lds_estep_vjp_f64", "svae_lds_vjp_workspace_bytes",
              "svae_hmm_estep_f64", "svae_hmm_workspace_bytes", "svae_slds_lds_meanfield_f64",
              "svae_slds_lds_meanfield_lds_bytes", "svae_slds_lds_meanfield_workspace_bytes", "svae_gmm_mw_workspace_bytes", "svae_gmm_mw_begin",
              "svae_gmm_mw_step_f64", "svae_gmm_mw_kl_hist", "svae_gmm_mw_fixed_point_f64", "svae_lds_global_step_f64", "svae_lds_global_step_multi_f64", "svae_lds_diag_sample_f64", "svae_lds_diag_sample_workspace_bytes", "svae_lds_natgrad_f64", "svae_lds_tile_vjp_f64",
              "svae_lds_tile_vjp_workspace_doubles", "svae_lds_tile_sample_f64", "svae_lds_tile_noise_f64",
              "svae_lds_tile_sigma_offset_bytes", "svae_gmm_meanfield_f64", "svae_hip_abi_version"):
        assert s in syms


def test_library_loads_and_exports_every_header_symbol():
    from svae_amd import _lib
    lib = _lib.load()                       # raises if not built: no fallback
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    for s in _header_symbols():
        assert hasattr(lib, s), s
    assert lib.svae_hip_abi_version() == _lib.ABI_VERSION


def test_workspace_size_formula():
    from svae_amd import _lib
    lib = _lib.load()
    # main region: n rows of [P^-1 J12 | c] (stride even(n+1)) + n rows of P^-1 (stride even(n));
    # factor region: n*n + n
    # plus one constant page (12 + 10 doubles) per sequence
    # and the cross-moment region ((n+1) rows of stride even(n+1)) per step
    # n <= 10: the main region holds BOTH that and the two-ended kernel's layout (two chains per sequence: a
    # constant page of 2(n+2)+2 doubles + T/2 + 1 records of n rows [P^-1 | P^-1 J12 | c | pad]): a launch that
    # keeps the sampler / VJP hand-off runs the one-directional filter and the two-ended E-step side by side;
    # one scratch double per sequence follows in every case
    one = lambda T, n: (n + 1 + (n + 1) % 2) + (n + n % 2) + T * n * ((n + 1 + (n + 1) % 2) + (n + n % 2))
    two = lambda T, n: 2 * (2 * (n + 2) + 2 + (T // 2 + 1) * n * (2 * n + 2))
    assert one(200, 10) == 22 + 200 * 10 * (12 + 10)
    assert lib.svae_lds_workspace_bytes(512, 200, 10) == 512 * (one(200, 10) + two(200, 10) + 1 + 200 * (10 * 10 + 10 + 11 * 12)) * 8
    assert lib.svae_lds_workspace_bytes(3, 7, 5) == 3 * (one(7, 5) + two(7, 5) + 1 + 7 * (5 * 5 + 5 + 6 * 6)) * 8
    assert lib.svae_lds_workspace_bytes(3, 7, 12) == 3 * (one(7, 12) + 1 + 7 * (12 * 12 + 12 + 13 * 14)) * 8
    # VJP scratch per (sequence, step): G^ (n rows, stride even(n+1)) | factors of the sampler share (2 n per sample, 4
    # samples) | lower triangle of the symmetric share of Pbar (padded to even) | Pbar(direct) (n rows, stride even(n))
    assert lib.svae_lds_vjp_workspace_bytes(3, 7, 5) == 3 * 7 * (5 * 6 + 2 * 4 * 5 + 16 + 5 * 6) * 8
    assert lib.svae_lds_vjp_workspace_bytes(512, 200, 10) == 512 * 200 * (10 * 12 + 80 + 56 + 100) * 8
    # n > 15: tiled path, per step X and P^-1 (NP x NP, NP = n rounded up to 16) and c (NP)
    # + the pair parameters re-packed in fragment order: 2 slots (homogeneous) or T-1 per set, 3 NP^2 each
    assert lib.svae_lds_workspace_bytes(1, 1, 16) == (2 * 16 * 16 + 16) * 8
    assert lib.svae_lds_workspace_bytes(2, 3, 40) == (2 * 3 * (2 * 48 * 48 + 48) + 2 * 3 * 48 * 48) * 8
    assert lib.svae_lds_workspace_bytes_ex(2, 3, 40, 0, 0) == lib.svae_lds_workspace_bytes(2, 3, 40)
    assert lib.svae_lds_workspace_bytes_ex(2, 3, 40, 1, 0) == (2 * 3 * (2 * 48 * 48 + 48) + 2 * 3 * 48 * 48) * 8
    assert lib.svae_lds_workspace_bytes_ex(2, 5, 40, 1, 1) == (2 * 5 * (2 * 48 * 48 + 48) + 2 * 4 * 3 * 48 * 48) * 8
    assert lib.svae_lds_workspace_bytes_ex(3, 7, 5, 1, 1) == lib.svae_lds_workspace_bytes(3, 7, 5)
    assert lib.svae_lds_workspace_bytes(1, 1, 65) == 0     # n > 64: unsupported
    assert lib.svae_lds_vjp_workspace_bytes(1, 1, 16) == 0  # VJP / sampler: register path only
    assert lib.svae_lds_workspace_bytes(0, 5, 3) == 0


def test_bad_arguments_are_rejected_on_the_host():
    """Argument errors return -k before any HIP call (safe without a GPU)."""
    from svae_amd import _lib
    lib = _lib.load()
    null = None
    args = [null] * 15 + [null, null, 0, null]
    assert lib.svae_lds_estep_f64(4, 0, 3, 0, 0, 0, 0, *args) == -2       # T < 1
    assert lib.svae_lds_estep_f64(4, 5, 65, 0, 0, 0, 0, *args) == -3      # n too large
    assert lib.svae_lds_estep_f64(4, 5, 16, 0, 0, 1, 0, *args) == -23     # tiled path keeps no sampler factors
    assert lib.svae_lds_estep_f64(4, 5, 3, 0, 1, 0, 0, *args) == -5       # batched pair w/o inhomog
    assert lib.svae_lds_estep_f64(4, 5, 3, 0, 0, 0, 0, *args) == -6       # NULL init_J
    assert lib.svae_lds_sample_f64(4, 5, 3, 0, 0, null, null, null, 0, null) == -4    # S < 1
    assert lib.svae_gmm_meanfield_f64(10, 9, 3, *([null] * 5), 1e-3, 100,
                                      *([null] * 8), null, null, null, null) == -2
    assert lib.svae_gmm_meanfield_f64(10, 2, 65, *([null] * 5), 1e-3, 100,
                                      *([null] * 8), null, null, null, null) == -3


def test_round3_entry_points_reject_bad_arguments_on_the_host():
    """svae_slds_path_nodeparams_f64 / svae_slds_mix_pair_natparam_f64 and the step ranges of svae_lds_tile_vjp_f64 /
    svae_lds_tile_noise_f64: argument errors come back before any HIP call."""
    import ctypes as C
    from svae_amd import _lib
    lib = _lib.load()
    buf = (C.c_double * 64)()
    ptr = C.cast(buf, C.c_void_p)
    nine = [ptr] * 9
    assert lib.svae_slds_path_nodeparams_f64(2, 3, 17, 4, *nine, None) == -3       # K > 16
    assert lib.svae_slds_path_nodeparams_f64(2, 3, 4, 16, *nine, None) == -4       # n > 15: no quadratic-form kernel
    assert lib.svae_slds_path_nodeparams_f64(2, 0, 4, 4, *nine, None) == -2        # T < 1
    assert lib.svae_slds_mix_pair_natparam_f64(2, 3, 4, 19, *nine, None) == -4     # 3 n^2 + 1 > 1024 threads
    assert lib.svae_slds_mix_pair_natparam_f64(2, 3, 17, 4, *nine, None) == -3
    assert lib.svae_slds_mix_pair_natparam_f64(2, 3, 4, 4, None, *nine[1:], None) == -5
    nws = lib.svae_lds_tile_vjp_workspace_doubles(1, 4, 16, 0)
    assert nws == 1 * 4 * 256 * 2 + 3 * 256 + 4 * 16 + 0 + (16 * 18 + 64)           # ... | phase-2 state between ranges
    big = (C.c_double * nws)()
    bp = C.cast(big, C.c_void_p)
    vjp = lambda phase, t0, t1: lib.svae_lds_tile_vjp_f64(phase, 1, 4, 16, 0, t0, t1, 0, 0, ptr, ptr, None, None, None, None,
                                                         None, None, ptr, ptr, ptr, bp, bp, nws, None)
    assert vjp(2, 3, 2) == -20 and vjp(2, 0, 5) == -20 and vjp(2, -1, 4) == -20     # bad range
    assert vjp(0, 1, 4) == -20 and vjp(1, 0, 3) == -20                               # phases 0 / 1: the whole chain only
    info = (C.c_int32 * 1)()
    ip = C.cast(info, C.c_void_p)
    assert lib.svae_lds_tile_noise_f64(0, 1, 4, 16, 1, 2, 2, ptr, ptr, bp, None, ip, None) == -20
    assert lib.svae_lds_tile_noise_f64(0, 1, 4, 16, 1, 0, 5, ptr, ptr, bp, None, ip, None) == -20


def test_keep_sigma_is_a_tile_path_bit_with_its_own_workspace_section():
    """SVAE_KEEP_SIGMA (ABI 9): accepted for 16 <= n <= 64 only, and only with room for the (B,T,n,n) section at
    svae_lds_tile_sigma_offset_bytes; the register path's keep bits stay 0..3."""
    import ctypes as C
    from svae_amd import _lib
    lib = _lib.load()
    assert lib.svae_lds_tile_sigma_offset_bytes(3, 5, 10, 0, 0) == 0
    base = lib.svae_lds_workspace_bytes_ex(3, 5, 40, 0, 0)
    off = lib.svae_lds_tile_sigma_offset_bytes(3, 5, 40, 0, 0)
    assert off >= base and off % 256 == 0 and off - base < 256
    buf = (C.c_double * 64)()
    ptr = C.cast(buf, C.c_void_p)
    info = (C.c_int32 * 1)()
    ip = C.cast(info, C.c_void_p)

    def estep(n, keep, ws_bytes):
        return lib.svae_lds_estep_f64(3, 5, n, 0, 0, keep, 0, *([ptr] * 10), *([ptr] * 5), ip, ptr, ws_bytes, None)
    assert estep(10, _lib.KEEP_SIGMA, 1 << 40) == -23          # register path: bits 0 and 1 only
    assert estep(40, 1, 1 << 40) == -23                        # tile path: SVAE_KEEP_SIGMA only
    assert estep(40, _lib.KEEP_SIGMA, off + 3 * 5 * 40 * 40 * 8 - 8) == -22      # no room for the section


def test_contradictory_or_unknown_options_are_rejected():
    """The per-call selection word (SVAE_OPT_*): contradictory pairs and unknown bits return -24 after the pointer
    checks and before any HIP call; the library exports no process-global selectors any more."""
    import ctypes as C
    from svae_amd import _lib
    lib = _lib.load()
    buf = (C.c_double * 4096)()
    ptr = C.cast(buf, C.c_void_p)
    ws = lib.svae_lds_workspace_bytes(1, 4, 2)
    assert 0 < ws <= 8 * 4096
    args = [ptr] * 15 + [ptr, ptr, ws, None]
    for bad in (_lib.OPT_TWOEND_OFF | _lib.OPT_TWOEND_FULL, _lib.OPT_LAYOUT_SPLIT | _lib.OPT_LAYOUT_PACKED,
                _lib.OPT_PRODUCERS_ON | _lib.OPT_PRODUCERS_OFF, 0x40, 0x80000000):
        assert lib.svae_lds_estep_f64(1, 4, 2, 0, 0, 0, bad, *args) == -24
        assert lib.svae_lds_sample_f64(1, 4, 2, 1, bad, ptr, ptr, ptr, ws, None) == -24
    # one half of the E-step per call: latent dimension 16 .. 64 only, and not both halves at once
    assert lib.svae_lds_estep_f64(1, 4, 2, 0, 0, 0, _lib.OPT_TILE_FORWARD, *args) == -24
    assert lib.svae_lds_estep_f64(1, 4, 2, 0, 0, 0, _lib.OPT_TILE_BACKWARD, *args) == -24
    for name in ("svae_lds_set_twoend", "svae_lds_set_split_max_b", "svae_lds_set_prod_max_b"):
        assert not hasattr(lib, name)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from svae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_round4_entry_points_reject_bad_arguments():
    """GMM global step / sampler / adjoint and the IPC mailbox all-reduce: argument checks run before any launch."""
    from svae_amd import _lib
    lib = _lib.load()
    one = 0x1000                 # any non-NULL address: the checks below must fail before it is used
    assert lib.svae_gmm_global_step_f64(0, 2, one, one, None, None, one, one, None, one, None) == -1
    assert lib.svae_gmm_global_step_f64(5, 9, one, one, None, None, one, one, None, one, None) == -2
    assert lib.svae_gmm_global_step_f64(5, 2, None, one, None, None, one, one, None, one, None) == -3
    assert lib.svae_gmm_global_step_f64(5, 2, one, one, None, None, one, one, one, one, None) == -5      # kl without a prior
    assert lib.svae_gmm_global_step_f64(5, 2, one, one, None, None, one, one, None, None, None) == -10
    assert lib.svae_gmm_sample_f64(-1, 2, 1, one, one, one, None) == -1
    assert lib.svae_gmm_sample_f64(4, 0, 1, one, one, one, None) == -2
    assert lib.svae_gmm_sample_f64(4, 2, 1, None, one, one, None) == -4
    assert lib.svae_gmm_sample_f64(4, 2, 1, one, None, one, None) == -5
    assert lib.svae_gmm_sample_f64(0, 2, 1, None, None, None, None) == 0                                 # nothing to do
    assert lib.svae_gmm_local_vjp_f64(4, 2, 0, 1, one, one, one, one, one, one, None, None, None, one, one, None) == -3
    assert lib.svae_gmm_local_vjp_f64(4, 2, 5, 1, None, one, one, one, one, one, None, None, None, one, one, None) == -5
    assert lib.svae_gmm_local_vjp_f64(4, 2, 5, 1, one, one, one, one, one, one, None, None, one, one, one, None) == -12  # sample cotangents without eps
    assert lib.svae_gmm_local_vjp_f64(4, 2, 5, 1, one, one, one, one, one, one, None, None, None, None, one, None) == -14
    assert lib.svae_ipc_mailbox_bytes(415, 8) == 2 * 8 * 2 * 415 * 8 and lib.svae_ipc_mailbox_bytes(415, 17) == 0
    # (n, capacity, rank, world, epoch, spin_limit, in, out, mailboxes, info, stream)
    assert lib.svae_ipc_allreduce_f64(0, 4, 0, 2, 1, 0, one, one, one, one, None) == -1
    assert lib.svae_ipc_allreduce_f64(5, 4, 0, 2, 1, 0, one, one, one, one, None) == -2      # n beyond the mailbox capacity
    assert lib.svae_ipc_allreduce_f64(4, 4, 2, 2, 1, 0, one, one, one, one, None) == -3
    assert lib.svae_ipc_allreduce_f64(4, 4, 0, 17, 1, 0, one, one, one, one, None) == -4
    assert lib.svae_ipc_allreduce_f64(4, 4, 0, 2, 0, 0, one, one, one, one, None) == -5
    assert lib.svae_ipc_allreduce_f64(4, 4, 0, 2, 1, 0, one, one, None, one, None) == -9
