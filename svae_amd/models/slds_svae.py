"""SLDS-SVAE local inference on MI355X; mirrors /root/reference/svae/models/slds_svae.py:80-310.

  lds_meanfield / get_var_lds_local_natparam      (:80-103)  -> svae_lds_estep_f64, per-sequence
                                                              time-inhomogeneous pair parameters
  hmm_meanfield / get_arhmm_local_nodeparams      (:108-147) -> svae_hmm_estep_f64
  optimize_local_meanfield                        (:159-175) coordinate ascent between the two
  initialize_local_meanfield                      (:203-226) one sample of a random-walk LDS posterior
  get_global_stats                                (:229-243)
  run_inference                                   (:289-310)

The reference module is stale as shipped (imports svae.lds.niw/mniw, svae.hmm.dirichlet, lds_svae,
none of which exist; hmm_estep needs the un-vendored pyhsmm).  Formulas are taken from it with
svae/distributions/{niw,mniw,dirichlet}.py; tests/golden/slds_*.npz hold the outputs of the reference
module itself (dead imports aliased, oracle/ref_py2.py) and the GPU tests check against them.  The two
message-passing hot loops run in the HIP kernels; the contractions between them (O(B T K n^2) einsums)
are torch ops on the device.

`reference_compat`: the reference's COMPILED filter reads init_params[2] only
(cython_lds_inference.pyx:32), so the 4th entry of the SLDS's init potential (b = 1/2 E log|J|, mixed
over E[z_0]) is missing from lds_vlb / local_vlb as shipped, while its Python twin sums the tail
(lds_inference.py:62-63).  Default True (since round 5) = the reference AS SHIPPED: the compiled path, bit for bit
in that term (it also moves the stopping test); False = the Python twin (the value that is actually a bound).
Batched: node potentials (B,T,n); every sequence runs its own coordinate ascent and stops on its own
|delta vlb| < tol like the reference (converged sequences are frozen).
"""
import torch

from .. import _lib
from ..distributions import expfam
from ..parallel import allreduce_nested
from ..hmm.hmm_inference import hmm_estep, hmm_logZ_differentiable
from ..lds.lds_inference import (LDSEStepPlan, lds_inference_differentiable, natural_lds_estep_general,
                                     natural_lds_sample)


_STATUS = {}     # (path name, device) -> (1,) int32 status word, PERSISTENT per device: every call of the path ORs into it


def _status_word(name, device):
    """The persistent (1,) int32 status word of (path, device), handed to the kernels as their `info` pointer: they write
    it only on failure (atomicMax / compare-and-swap, never a clear), so a failure of an EARLIER call survives later
    successful ones whatever the number of devices, streams and plans in flight -- and the hot path pays nothing for it
    (the dict used to keep only the last call's freshly allocated word)."""
    key = (name, str(torch.device(device)))
    word = _STATUS.get(key)
    if word is None:
        _STATUS[key] = word = torch.zeros(1, dtype=torch.int32, device=device)
    return word


def check_info():
    """Synchronising check of the status words the SLDS device paths leave on the device -- the global -> local maps
    (a non-positive-definite NIW / MNIW scale), the diagonal initial sample path (a non-positive precision) and the
    fused LDS mean-field kernel (a non-positive pivot) -- like gmm.check_info(): the ascent itself never reads them
    (no host synchronisation on the hot path; the reference ignores LAPACK `info` altogether,
    cython_gaussian_grads.pxd:54-76).  Raises FloatingPointError naming the path."""
    bad = []
    for (name, device), word in list(_STATUS.items()):
        v = int(word.item())
        if v != 0:
            word.zero_()                       # (every word is read and cleared before anything is raised)
            bad.append("%s on %s: status word %d" % (name, device, v))
    if bad:
        raise FloatingPointError("SLDS " + "; ".join(bad) + " (parameters / potentials not positive definite)")


def _dev64(x, device):
    # dtype given up front: torch.as_tensor(python_float) alone would round to float32
    t = x.detach() if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=torch.float64)
    return t.to(device=device, dtype=torch.float64).contiguous()


def hmm_prior_expectedstats(natparam):
    """(:124-130) Dirichlet rows -> (E log pi_0 (K), E log P (K,K))."""
    dir_natparam, mdir_natparam = natparam
    return expfam.dirichlet_expectedstats(dir_natparam), expfam.dirichlet_expectedstats(mdir_natparam)


def get_all_lds_local_natparams(lds_global_natparams):
    """(:86-89) per discrete state k: init 4-tuple and pair 4-tuple, stacked over k."""
    inits, pairs = [], []
    for niw_nat, mniw_nat in lds_global_natparams:
        inits.append(expfam.unpack_dense(expfam.niw_expectedstats(niw_nat)))
        pairs.append(expfam.mniw_expectedstats(mniw_nat))
    stack = lambda tuples: tuple(torch.stack([torch.as_tensor(t[i], dtype=torch.float64) for t in tuples]) for i in range(4))
    return stack(inits), stack(pairs)          # each: 4 tensors with leading K


def _global_to_local_maps_device(global_natparam, device):
    """global_to_local_maps with the K (NIW, MNIW) factor pairs through the LDS global-step kernel -- ONE launch for all K
    states (svae_lds_global_step_multi_f64; K > 16: one launch per state) writing the stacked outputs -- and the Dirichlet
    rows as two digamma expressions on the device: no host arithmetic, no host -> device copies of the results."""
    from .lds import GLOBAL_STEP_MAX_N
    hmm_global, lds_global = global_natparam
    K = len(lds_global)
    n = int(torch.as_tensor(lds_global[0][0]).shape[-1]) - 2
    if not (1 <= n <= GLOBAL_STEP_MAX_N):
        return None
    f64 = dict(dtype=torch.float64, device=device)
    c = lambda x: _dev64(x, device).contiguous()
    hmm_init, hmm_pair = hmm_prior_expectedstats(tuple(c(x) for x in hmm_global))
    D = n + 2
    # ONE launch for the K factor pairs (svae_lds_global_step_multi_f64): host arrays of K device pointers in, stacked
    # outputs (K, ...) out
    import ctypes
    f = lambda shape: torch.empty(*shape, **f64)
    init_J, init_h, init_lz = f((K, n, n)), f((K, n)), f((K,))
    J11, J12, J22, lzp = f((K, n, n)), f((K, n, n)), f((K, n, n)), f((K,))
    esb = f((K, D, D))
    info = _status_word("global -> local maps", device)
    lib, p = _lib.load(), _lib.ptr
    keep = [[c(niw), c(A), c(Bm), c(C), c(d).reshape(1)] for niw, (A, Bm, C, d) in lds_global]
    arrs = [(ctypes.c_void_p * K)(*[t[i].data_ptr() for t in keep]) for i in range(5)]
    if K <= 16:
        rc = lib.svae_lds_global_step_multi_f64(K, n, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4],
                                                p(init_J), p(init_h), p(init_lz), p(J11), p(J12), p(J22), p(lzp), p(esb),
                                                p(info), _lib.current_stream(device))
        _lib.check(rc, "svae_lds_global_step_multi_f64")
    else:
        for k, t in enumerate(keep):
            rc = lib.svae_lds_global_step_f64(
                n, p(t[0]), p(t[1]), p(t[2]), p(t[3]), p(t[4]), None, None, None, None, None,
                p(init_J[k]), p(init_h[k]), p(init_lz[k:k + 1]), p(J11[k]), p(J12[k]), p(J22[k]), p(lzp[k:k + 1]), p(esb[k]),
                None, p(info), _lib.current_stream(device))
            _lib.check(rc, "svae_lds_global_step_f64")
    global_to_local_maps.last_info = info
    dense_init = (esb[:, :n, :n].contiguous(), esb[:, :n, n].contiguous(), esb[:, n, n].contiguous(),
                  esb[:, n + 1, n + 1].contiguous())
    dense_pair = (J11, J12, J22, lzp)
    return hmm_init.contiguous(), hmm_pair.contiguous(), dense_init, dense_pair


def global_to_local_maps(global_natparam, device):
    """The once-per-step global -> local maps of the SLDS (hmm_prior_expectedstats :124-130 and
    get_all_lds_local_natparams :86-89) -> (hmm_init (K), hmm_pair (K,K), dense_init 4-tuple, dense_pair 4-tuple).
    On a GPU: K launches of the LDS global-step kernel (round 4; _global_to_local_maps_device).  Otherwise (CPU tensors,
    n > 64) the same maps in float64 torch on the host, moved over as 10 small tensors (as torch device ops they are a
    few hundred tiny launches, 7 ms)."""
    hmm_global, lds_global = global_natparam
    if torch.device(device).type == "cuda":
        # (also for parameters that still live on the host: 5 K small copies in -- the host path's O(K n^3) torch CPU
        #  operators take anything from 2 to 100+ ms on a box whose core quota is below its hardware thread count)
        out = _global_to_local_maps_device(global_natparam, torch.device(device))
        if out is not None:
            return out
    cpu = torch.device("cpu")
    hmm_init, hmm_pair = hmm_prior_expectedstats(tuple(_dev64(x, cpu) for x in hmm_global))
    lds_cpu = [(_dev64(a, cpu), tuple(_dev64(y, cpu) for y in m)) for a, m in lds_global]
    dense_init, dense_pair = get_all_lds_local_natparams(lds_cpu)
    to = lambda x: x.to(device).contiguous()
    return to(hmm_init), to(hmm_pair), tuple(to(x) for x in dense_init), tuple(to(x) for x in dense_pair)


def get_var_lds_local_natparam(dense_init, dense_pair, expected_states):
    """(:92-103) expected_states (B,T,K) -> init params (B,..) and per-step pair params (B,T-1,..)."""
    w0, w1 = expected_states[:, 0], expected_states[:, 1:]
    init = tuple(torch.tensordot(w0, p, dims=1) for p in dense_init)
    B, T, K = expected_states.shape
    n = dense_pair[0].shape[-1]
    if expected_states.is_cuda and T > 1 and K <= 16 and 3 * n * n + 1 <= 1024 \
            and not any(x.requires_grad for x in (expected_states,) + tuple(dense_pair)):
        # the (B,T-1) x K x (3 n^2 + 1) mixture in one kernel bound by the output it writes (svae_slds_mix_pair_natparam_f64)
        dev = expected_states.device
        f64 = dict(dtype=torch.float64, device=dev)
        out = tuple(torch.empty(B, T - 1, n, n, **f64) for _ in range(3)) + (torch.empty(B, T - 1, **f64),)
        p = _lib.ptr
        params = tuple(_dev64(x, dev).contiguous() for x in dense_pair)
        rc = _lib.load().svae_slds_mix_pair_natparam_f64(B, T, K, n, p(_dev64(expected_states, dev).contiguous()),
                                                         *(p(x) for x in params), *(p(x) for x in out),
                                                         _lib.current_stream(dev))
        _lib.check(rc, "svae_slds_mix_pair_natparam_f64")
        return init, out
    pair = tuple(torch.tensordot(w1, p, dims=1) for p in dense_pair)
    return init, pair


def _packed_pair_stats(pair_stats):
    """Per-step pair statistics as ONE (B,T-1,3 n^2) matrix per sequence: `pair_stats` is the kernels' packed
    (B,T-1,3,n,n) tensor, or the reference's 3-tuple of (B,T-1,n,n) blocks (stacked: one copy)."""
    if isinstance(pair_stats, torch.Tensor):
        B, Tm1 = pair_stats.shape[:2]
        return pair_stats.reshape(B, Tm1, -1)
    Exx = pair_stats[0]
    B, Tm1, n = Exx.shape[:3]
    base = getattr(Exx, "_base", None)
    if base is not None and tuple(base.shape) == (B, Tm1, 3, n, n) and base.is_contiguous() and all(
            getattr(x, "_base", None) is base and x.data_ptr() == base.data_ptr() + 8 * i * n * n
            and x.stride() == (3 * Tm1 * n * n, 3 * n * n, n, 1) for i, x in enumerate(pair_stats[:3])):
        return (base.detach() if not Exx.requires_grad else base).reshape(B, Tm1, -1)   # the kernels' own packed buffer
    return torch.stack(tuple(pair_stats[:3]), 2).reshape(B, Tm1, -1)


def get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, pair_stats):
    """(:131-147) node[b,0,k] = <init_stats_b, init_params_k>, node[b,t+1,k] = <pair_stats_bt, pair_params_k>.
    pair_stats: 3-tuple of (B,T-1,n,n) or the packed (B,T-1,3,n,n) tensor (one GEMM against the K stacked
    parameter sets instead of three strided contractions)."""
    ExxT0, Ex0 = init_stats
    K = dense_init[0].shape[0]
    n0 = ExxT0.reshape(ExxT0.shape[0], -1) @ dense_init[0].reshape(K, -1).T + Ex0 @ dense_init[1].T \
        + dense_init[2] + dense_init[3]
    P = torch.cat([dense_pair[i].reshape(K, -1) for i in range(3)], 1)          # (K, 3 n^2)
    nt = _packed_pair_stats(pair_stats) @ P.T + dense_pair[3]
    return torch.cat([n0[:, None], nt], 1)


def _pair_contract_applies(dense_init, pair_stats):
    """where svae_slds_pair_contract_f64 serves: the kernels' packed (B,T-1,3,n,n) statistics on the device, n <= 10, K <= 8"""
    if not (isinstance(pair_stats, torch.Tensor) and pair_stats.is_cuda and pair_stats.dim() == 5
            and pair_stats.is_contiguous()):
        return False
    B, Tm1, _, n, _ = pair_stats.shape
    return n <= 10 and dense_init[0].shape[0] <= 8 and Tm1 >= 1 and B >= 1


def final_pass_contractions(dense_init, dense_pair, init_stats, pair_stats, expected_states):
    """The pair part of get_arhmm_local_nodeparams (:131-147) and of get_global_stats (:229-243) on the per-step pair
    statistics of the final LDS E-step in ONE pass over them (svae_slds_pair_contract_f64; they are 2.45 GB at configs[3]
    and were read by two library GEMMs).  pair_stats: the kernels' packed (B,T-1,3,n,n) tensor.
    -> (node_hmm (B,T,K), g_pair_sums (K,3,n,n)), or None where the kernel does not apply (the callers then use the two
    functions above)."""
    if not _pair_contract_applies(dense_init, pair_stats) or pair_stats.requires_grad:
        return None
    B, Tm1, _, n, _ = pair_stats.shape
    K = dense_init[0].shape[0]
    dev = pair_stats.device
    T = Tm1 + 1
    ExxT0, Ex0 = init_stats
    f64 = dict(dtype=torch.float64, device=dev)
    P = torch.cat([_dev64(dense_pair[i], dev).reshape(K, -1) for i in range(3)], 1).contiguous()
    lz = _dev64(dense_pair[3], dev).contiguous()
    Es = _dev64(expected_states, dev).contiguous()
    node = torch.empty(B, T, K, **f64)
    blocks = min(B, 2 * torch.cuda.get_device_properties(dev).multi_processor_count)
    gbuf = torch.empty(blocks * 8 * 3 * n * n, **f64)
    gpart = gbuf.view(blocks, 8, 3 * n * n)
    p = _lib.ptr
    rc = _lib.load().svae_slds_pair_contract_f64(B, T, K, n, p(pair_stats), p(P), p(lz), p(Es), p(node), p(gbuf), blocks,
                                                 _lib.current_stream(dev))
    _lib.check(rc, "svae_slds_pair_contract_f64")
    node[:, 0] = ExxT0.reshape(B, -1) @ dense_init[0].reshape(K, -1).T + Ex0 @ dense_init[1].T \
        + dense_init[2] + dense_init[3]
    return node, gpart.sum(0)[:K].reshape(K, 3, n, n)


class _PairContract(torch.autograd.Function):
    """final_pass_contractions with the gradient of the HMM node potentials w.r.t. the pair statistics attached:
    node[b,t+1,k] = <S[b,t], P_k> + lz_k  =>  dL/dS[b,t] = sum_k g[b,t+1,k] P_k (one library GEMM in the backward pass);
    the weighted sums are formed from the values only (the reference detaches the statistics it returns).
    The K parameter sets (dense_init, dense_pair) are CONSTANTS of this node: no gradient flows to the global
    parameters through it (the SVAE never asks for one -- the global step is the closed-form natural gradient,
    svae.py:33-34); the caller checks that they do not require grad and that the kernel applies
    (_pair_contract_applies), and otherwise takes get_arhmm_local_nodeparams."""

    @staticmethod
    def forward(ctx, pair_stats, ExxT0, Ex0, dense_init, dense_pair, expected_states):
        node, sums = final_pass_contractions(dense_init, dense_pair, (ExxT0.detach(), Ex0.detach()), pair_stats.detach(),
                                             expected_states)
        K = dense_init[0].shape[0]
        ctx.P = torch.cat([dense_pair[i].reshape(K, -1) for i in range(3)], 1)
        ctx.Jk, ctx.hk = dense_init[0].reshape(K, -1), dense_init[1]
        ctx.shapes = (pair_stats.shape, ExxT0.shape)
        ctx.mark_non_differentiable(sums)
        return node, sums

    @staticmethod
    def backward(ctx, g_node, _g_sums):
        g_pair = (g_node[:, 1:] @ ctx.P).reshape(ctx.shapes[0])
        g0 = g_node[:, 0]
        return g_pair, (g0 @ ctx.Jk).reshape(ctx.shapes[1]), g0 @ ctx.hk, None, None, None


def initialize_local_meanfield(node_potentials, eps):
    """(:203-226) statistics of ONE posterior sample path of a random-walk LDS; eps (B,T,1,n)."""
    x = _initial_sample_path(node_potentials, eps)
    out = lambda a, b: a.unsqueeze(-1) * b.unsqueeze(-2)
    init_stats = (out(x[:, 0], x[:, 0]), x[:, 0])
    pair_stats = (out(x[:, :-1], x[:, :-1]), out(x[:, :-1], x[:, 1:]), out(x[:, 1:], x[:, 1:]))
    return init_stats, pair_stats


class SLDSMeanfieldPlan(object):
    """Buffers of the FUSED LDS mean-field step (svae_slds_lds_meanfield_f64): the K per-state parameter sets
    stay in LDS, the HMM marginals stream in as weights, and the pair statistics leave the kernel already
    contracted with the K sets (B,T,2,K) -- neither the per-step pair parameters nor the per-step pair
    statistics ((B,T-1,n,n) x 3 each, slds_svae.py:92-103 / :141-146) are ever materialised.  The buffers
    are the persistent state of the coordinate ascent: a launch with a `seq_index` list rewrites only the
    rows of the sequences still iterating (and costs only their share of the work)."""

    def __init__(self, B, T, n, K, device, options=0):
        self.lib = _lib.load()
        self.B, self.T, self.n, self.K = B, T, n, K
        # kernel-selection word of svae_slds_lds_meanfield_f64 (0 = the library's choice: row-per-chain consumers + MFMA
        # producer wavefronts for K <= 8; _lib.OPT_LAYOUT_SPLIT = the one-sequence-per-wavefront table kernel of rounds
        # 2 - 4; OPT_LAYOUT_PACKED | OPT_PRODUCERS_OFF = reference producers, test infrastructure)
        self.options = int(options)
        self.device = torch.device(device)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.ws_bytes = int(self.lib.svae_slds_lds_meanfield_workspace_bytes(max(B, 1), T, n))
        self.ws = torch.empty(self.ws_bytes // 8, **f64)
        # (no zero fills: the first launch of an ascent runs on every row and writes all of these)
        self.lognorm = torch.empty(B, **f64)
        self.E_init = torch.empty(B, n * n + n, **f64)
        self.E_node_diagxx = torch.empty(B, T, n, **f64)
        self.E_node_x = torch.empty(B, T, n, **f64)
        self.pair_contr = torch.empty(B, T, 2, K, **f64)
        self.info = _status_word("fused LDS mean field", self.device)     # (shared by the plans of a device)

    @staticmethod
    def supported(n, T, K):
        """The fused kernel covers n <= 10, T >= 4 and K parameter sets that fit the 160 KiB of LDS."""
        if n > 10 or T < 4 or K > 16:
            return False
        nbytes = int(_lib.load().svae_slds_lds_meanfield_lds_bytes(n, K))
        return 0 < nbytes <= 160 * 1024

    def launch(self, dense_init, dense_pair, weights, node, seq_index=None, nrun=None):
        """dense_init = (J (K,n,n), h (K,n), a (K), b (K)), dense_pair = (J11, J12, J22 (K,n,n), logZ (K)),
        weights (B,T,K), node = (J, h[, logZ]); seq_index: int32 tensor of the rows to process (its first `nrun`
        entries; default all of it), or None = all rows."""
        p = _lib.ptr
        if nrun is None:
            nrun = self.B if seq_index is None else int(seq_index.numel())
        rc = self.lib.svae_slds_lds_meanfield_f64(
            nrun, self.B, self.T, self.n, self.K, p(dense_init[0]), p(dense_init[1]),
            p(dense_pair[0]), p(dense_pair[1]), p(dense_pair[2]), p(weights),
            p(node[0]), p(node[1]), p(node[2]) if len(node) > 2 else None, p(seq_index),
            p(self.lognorm), p(self.E_init), p(self.E_node_diagxx), p(self.E_node_x), p(self.pair_contr),
            p(self.info), p(self.ws), self.ws_bytes, self.options, _lib.current_stream(self.device))
        _lib.check(rc, "svae_slds_lds_meanfield_f64")

    def hmm_nodeparams(self, dense_init, dense_pair, rows=None):
        """get_arhmm_local_nodeparams (:131-147) from the kernel's contracted pair statistics
        (rows: int64 index tensor = only those sequences)."""
        n = self.n
        E_init = self.E_init if rows is None else self.E_init.index_select(0, rows)
        pc = self.pair_contr if rows is None else self.pair_contr.index_select(0, rows)
        ExxT0, Ex0 = E_init[:, :n * n], E_init[:, n * n:]
        n0 = ExxT0 @ dense_init[0].reshape(self.K, n * n).T + Ex0 @ dense_init[1].T + dense_init[2] + dense_init[3]
        nt = pc[:, :-1, 0] + pc[:, 1:, 1] + dense_pair[3]
        return torch.cat([n0[:, None], nt], 1)

    def lds_vlb(self, dense_init, dense_pair, weights, reference_compat=True):
        """log-normaliser of the mixed LDS = kernel part + the mixed constants."""
        const0 = dense_init[2] if reference_compat else dense_init[2] + dense_init[3]
        return self.lognorm + weights[:, 0] @ const0 + (weights[:, 1:] @ dense_pair[3]).sum(1)


def _arhmm_nodeparams_from_path(dense_init, dense_pair, x):
    """get_arhmm_local_nodeparams (:131-147) for the statistics of ONE sample path x (B,T,n)
    (initialize_local_meanfield, :203-226) as quadratic forms -- without building the outer products."""
    B, T, n = x.shape
    K = dense_init[0].shape[0]
    if x.is_cuda and n <= 15 and K <= 16:
        # three quadratic forms per (sequence, step, state) in one kernel (svae_slds_path_nodeparams_f64)
        dev = x.device
        c = lambda v: _dev64(v, dev).contiguous()
        out = torch.empty(B, T, K, dtype=torch.float64, device=dev)
        p = _lib.ptr
        rc = _lib.load().svae_slds_path_nodeparams_f64(B, T, K, n, p(c(x)), p(c(dense_init[0])), p(c(dense_init[1])),
                                                       p(c(dense_init[2] + dense_init[3])), p(c(dense_pair[0])),
                                                       p(c(dense_pair[1])), p(c(dense_pair[2])), p(c(dense_pair[3])),
                                                       p(out), _lib.current_stream(dev))
        _lib.check(rc, "svae_slds_path_nodeparams_f64")
        return out
    x0, xa, xb = x[:, 0], x[:, :-1], x[:, 1:]
    n0 = torch.einsum("bi,kij,bj->bk", x0, dense_init[0], x0) + x0 @ dense_init[1].T + dense_init[2] + dense_init[3]
    q = lambda u, M, v: (torch.einsum("bti,kij->btkj", u, M) * v.unsqueeze(2)).sum(-1)
    nt = q(xa, dense_pair[0], xa) + q(xa, dense_pair[1], xb) + q(xb, dense_pair[2], xb) + dense_pair[3]
    return torch.cat([n0[:, None], nt], 1)


def _optimize_local_meanfield_fused(hmm_init, hmm_pair, dense_init, dense_pair, node, init_eps, tol, max_iter,
                                    reference_compat=True):
    """The coordinate ascent of optimize_local_meanfield on the fused kernels.  Same iteration as the materialised path
    (and the reference): hmm_meanfield -> lds_meanfield -> |delta vlb| < tol per sequence.  A sweep is three launches on
    the list of sequences still iterating -- the HMM kernel (node potentials built on the fly from the previous LDS
    step's contracted statistics), the fused LDS kernel, and the glue kernel (bounds, stopping test, sweep counters,
    compaction of the list) -- with ONE 4-byte read by the host (the length of the next list, which sizes the next
    launches); no torch arithmetic in the loop."""
    B, T, n = node[1].shape
    K = dense_init[0].shape[0]
    dev = node[1].device
    lib, p = _lib.load(), _lib.ptr
    plan = SLDSMeanfieldPlan(B, T, n, K, dev)
    dense_init = tuple(x.contiguous() for x in dense_init)
    dense_pair = tuple(x.contiguous() for x in dense_pair)
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    x = _initial_sample_path(node, init_eps)
    node_hmm = _arhmm_nodeparams_from_path(dense_init, dense_pair, x).contiguous()      # sweep 0 (:203-226)
    cinit_hmm = (dense_init[2] + dense_init[3]).contiguous()       # HMM node potential of step 0 (:138: both constants)
    # ... and of the LDS bound: the compiled reference filter drops the second one (module docstring)
    cinit_vlb = dense_init[2].contiguous() if reference_compat else cinit_hmm
    lz = dense_pair[3].contiguous()
    # (sweep 0 runs on all B rows: every buffer below is written before it is read)
    st = dict(Ei=torch.empty(B, K, **f64), Et=torch.empty(B, K, K, **f64), Es=torch.empty(B, T, K, **f64),
              hmm_vlb=torch.empty(B, **f64), node_hmm=node_hmm)
    lds_vlb = torch.empty(B, **f64)
    vlb = torch.full((B,), -float("inf"), **f64)
    iters = torch.zeros(B, **i32)
    hws_bytes = int(lib.svae_hmm_workspace_bytes(max(B, 1), T, K))
    hws = torch.empty(hws_bytes // 8, **f64)
    lists = [torch.arange(B, **i32), torch.empty(B, **i32)]
    keep, count = torch.empty(B, **i32), torch.empty(1, **i32)
    stream = _lib.current_stream(dev)
    # (Measured and not kept, round 4: the batch as two halves on two streams, staggered by one phase so that one half's
    #  HMM kernel runs under the other's LDS kernel.  The fused LDS kernel on half the batch takes 1.0 ms, not 0.75 -- a
    #  round of both halves 2.0 ms against 2.2 -- and in the late sweeps, a handful of sequences per half, every half
    #  waits for the other's latency-bound kernel: 32 ms instead of 26 for the ascent.)
    # The host learns the length of a sweep's new list ONE SWEEP LATE (an asynchronous 4-byte copy and an event per sweep
    # instead of a blocking read): sweep i is launched on `nrun` slots = the length known from sweep i - 2, an upper bound
    # -- the glue kernel pads the compacted list with -1 up to the launch size and every kernel skips negative slots --
    # so the three launches of consecutive sweeps follow each other without a host round trip (~45 us per sweep).  The
    # sweep after the last sequence has converged runs on an all-unused list and costs three empty launches.
    pinned = torch.empty(max_iter + 1, dtype=torch.int32).pin_memory()
    events = []
    nrun, cur = B, 0
    ts = torch.cuda.current_stream(dev)          # the stream the kernels are launched on (dev need not be the current device)
    for it in range(max_iter):
        if nrun == 0:
            break
        idx = lists[cur]
        rc = lib.svae_slds_hmm_meanfield_f64(
            nrun, B, T, K, n, p(hmm_init), p(hmm_pair), p(node_hmm) if it == 0 else None,
            p(plan.pair_contr), p(plan.E_init), p(dense_init[0]), p(dense_init[1]), p(cinit_hmm), p(lz), p(idx),
            p(st["hmm_vlb"]), p(st["Ei"]), p(st["Et"]), p(st["Es"]), None if it == 0 else p(st["node_hmm"]),
            p(hws), hws_bytes, stream)
        _lib.check(rc, "svae_slds_hmm_meanfield_f64")
        plan.launch(dense_init, dense_pair, st["Es"], node, idx, nrun)
        rc = lib.svae_slds_sweep_glue_f64(nrun, T, K, float(tol), p(idx), p(st["Es"]), p(cinit_vlb), p(lz), p(plan.lognorm),
                                          p(st["hmm_vlb"]), p(lds_vlb), p(vlb), p(iters), p(keep), p(lists[1 - cur]),
                                          p(count), stream)
        _lib.check(rc, "svae_slds_sweep_glue_f64")
        cur = 1 - cur
        with torch.cuda.stream(ts):
            pinned[it:it + 1].copy_(count, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(ts)
        events.append(ev)
        if it >= 1:
            events[it - 1].synchronize()            # (long complete: the next sweep is already queued behind it)
            nrun = int(pinned[it - 1])              # list length after sweep it - 1 >= the length sweep it + 1 will find
    return plan, st, lds_vlb, iters.to(torch.int64)


_RW_CACHE = {}


def _random_walk_natparam(n, dev):
    """Natural parameters of the random-walk LDS of initialize_local_meanfield (:203-226: x_0 ~ N(0, I), A = 0.9 I,
    unit noise) -- constants of (n, device), built once (a dozen tiny launches otherwise, every ascent)."""
    key = (n, str(dev))
    if key not in _RW_CACHE:
        eye = torch.eye(n, dtype=torch.float64, device=dev)
        A = 0.9 * eye
        zero = torch.zeros((), dtype=torch.float64, device=dev)
        _RW_CACHE[key] = ((-0.5 * eye, torch.zeros(n, dtype=torch.float64, device=dev), zero),
                          ((-0.5 * A.T @ A).contiguous(), A.T.contiguous(), -0.5 * eye, zero))
    return _RW_CACHE[key]


def _initial_sample_path(node_potentials, eps):
    """(:203-226) ONE posterior sample path x (B,T,n) of a random-walk LDS given the node potentials."""
    nJ = node_potentials[0]
    B, T, n = nJ.shape
    dev = nJ.device
    if nJ.is_cuda and nJ.dim() == 3 and eps is not None:
        # every matrix of this model is diagonal: n scalar recursions per sequence (svae_lds_diag_sample_f64), the same
        # eps -> sample map as the dense filter + sampler below (tests/test_slds_hip.py pins the two against each other)
        key = ("diag", n, str(dev))
        if key not in _RW_CACHE:
            v = lambda c: torch.full((n,), c, dtype=torch.float64, device=dev)
            _RW_CACHE[key] = (v(-0.5), v(0.0), v(-0.5 * (0.9 * 0.9)), v(0.9), v(-0.5))
        iJ, ih, j11, j12, j22 = _RW_CACHE[key]
        lib, p = _lib.load(), _lib.ptr
        nJc, nhc = _dev64(nJ, dev).contiguous(), _dev64(node_potentials[1], dev).contiguous()
        e = _dev64(eps, dev).reshape(B, T, n).contiguous()
        out = torch.empty(B, T, n, dtype=torch.float64, device=dev)
        wsb = int(lib.svae_lds_diag_sample_workspace_bytes(max(B, 1), T, n))
        ws = torch.empty(wsb // 8, dtype=torch.float64, device=dev)
        info = _status_word("initial sample path", dev)
        rc = lib.svae_lds_diag_sample_f64(B, T, n, p(iJ), p(ih), p(j11), p(j12), p(j22), p(nJc), p(nhc), p(e), p(out),
                                          p(info), p(ws), wsb, _lib.current_stream(dev))
        _lib.check(rc, "svae_lds_diag_sample_f64")
        _initial_sample_path.last_info = info
        return out
    natparam = _random_walk_natparam(n, dev)
    x = natural_lds_sample(natparam, node_potentials, num_samples=1, eps=eps)     # filter + sampler, no smoother (:222)
    return x[:, :, 0]                                                # (B,T,n)


def optimize_local_meanfield(global_natparam, node_potentials, init_eps, tol=1e-2, max_iter=100, fused=None,
                             pair_stats=True, reference_compat=True, local_maps=None):
    """(:159-175).  Returns ((hmm_stats, lds_stats), (hmm_natparam, lds_natparam), (hmm_vlb, lds_vlb), iters).

    fused=None picks the fused LDS mean-field kernel (SLDSMeanfieldPlan) when it covers the shape, else the
    path that materialises per-step pair parameters and statistics; True / False force one.  On the fused
    path the per-step pair statistics of the reference's `lds_stats` tuple exist only if `pair_stats` (one
    extra E-step on the converged mean field): callers that run their own final pass (run_inference) skip it
    and get `None` in that slot.  local_maps: the result of global_to_local_maps(global_natparam, device) where the caller
    has it already (run_inference needs it again behind the ascent: rebuilt there, its small host-to-device copies queue
    behind the final pass's kernels and block the host for their duration)."""
    hmm_global, lds_global = global_natparam
    dev = node_potentials[0].device
    node = tuple(_dev64(x, dev) for x in node_potentials)
    B, T, n = node[1].shape
    hmm_init, hmm_pair, dense_init, dense_pair = local_maps if local_maps is not None else \
        global_to_local_maps(global_natparam, dev)
    K = dense_init[0].shape[0]
    if fused is None or (fused and not SLDSMeanfieldPlan.supported(n, T, K)):
        fused = SLDSMeanfieldPlan.supported(n, T, K)       # (a forced fused=True outside the kernel's coverage: the materialised path)
    if fused:
        fplan, st, lds_vlb, iters = _optimize_local_meanfield_fused(
            hmm_init, hmm_pair, dense_init, dense_pair, node, _dev64(init_eps, dev), tol, max_iter, reference_compat)
        lds_init, lds_pair = get_var_lds_local_natparam(dense_init, dense_pair, st["Es"])
        init_stats = (fplan.E_init[:, :n * n].reshape(B, n, n), fplan.E_init[:, n * n:])
        pstats = None
        if pair_stats:
            plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
            _, (_, Ep, _) = _lds_estep_batched_init(plan, lds_init, lds_pair, node)
            pstats = tuple(Ep[:3])
        lds_stats = (init_stats, pstats, (fplan.E_node_diagxx, fplan.E_node_x))
        return ((st["Ei"], st["Et"], st["Es"]), lds_stats), \
            ((hmm_init, hmm_pair, st["node_hmm"]), (lds_init, lds_pair)), (st["hmm_vlb"], lds_vlb), iters
    plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)

    init_stats, pair_stats = initialize_local_meanfield(node, _dev64(init_eps, dev))
    vlb = torch.full((B,), -float("inf"), dtype=torch.float64, device=dev)
    active = torch.ones(B, dtype=torch.bool, device=dev)
    iters = torch.zeros(B, dtype=torch.int64, device=dev)
    # State of the ascent, one persistent buffer each; a sweep overwrites the rows of the sequences still
    # active (converged ones are frozen, like the reference's per-sequence `break`).  The big ones --
    # per-step pair statistics, 3 (T-1) n^2 doubles per sequence -- are only ever copied row-wise.
    state = {}

    def commit(name, val, all_active):
        if name not in state:
            state[name] = val.clone()
        elif all_active:
            state[name].copy_(val)
        else:
            idx = active.nonzero(as_tuple=True)[0]
            state[name].index_copy_(0, idx, val.index_select(0, idx))

    for _ in range(max_iter):
        node_hmm = get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, pair_stats)
        hmm_vlb, (Ei, Et, Es) = hmm_estep((hmm_init, hmm_pair, node_hmm))
        lds_init, lds_pair = get_var_lds_local_natparam(dense_init, dense_pair, Es)
        # the E-step API takes one shared init potential per launch: fold each sequence's init
        # potential into its first node potential instead (identical model: both multiply x_0's factor)
        lds_vlb, (Ei_l, Ep_l, En_l) = _lds_estep_batched_init(plan, lds_init, lds_pair, node,
                                                              reference_compat=reference_compat)
        all_active = bool(active.all())
        for name, val in (("Ei", Ei), ("Et", Et), ("Es", Es), ("node_hmm", node_hmm), ("E_init", plan.E_init),
                          ("E_pair", plan.E_pair), ("dxx", En_l[0]), ("ex", En_l[1]), ("hmm_vlb", hmm_vlb),
                          ("lds_vlb", lds_vlb)):
            commit(name, val, all_active)
        init_stats = (state["E_init"][:, :n * n].reshape(B, n, n), state["E_init"][:, n * n:])
        pair_stats = tuple(state["E_pair"][:, :, i] for i in range(3))
        new_vlb = state["hmm_vlb"] + state["lds_vlb"]
        iters += active.to(torch.int64)
        done = (new_vlb - vlb).abs() < tol
        vlb = new_vlb
        active = active & ~done
        if not bool(active.any()):
            break
    # the natural parameters of the frozen solution (cheap to rebuild from the final HMM marginals)
    lds_init, lds_pair = get_var_lds_local_natparam(dense_init, dense_pair, state["Es"])
    lds_stats = (init_stats, pair_stats, (state["dxx"], state["ex"]))
    return ((state["Ei"], state["Et"], state["Es"]), lds_stats), \
        ((hmm_init, hmm_pair, state["node_hmm"]), (lds_init, lds_pair)), (state["hmm_vlb"], state["lds_vlb"]), iters


def optimize_local_meanfield_withlabels(global_natparam, node_potentials, labels):
    """(:178-200) discrete states given: HMM stats from the labels (B,T) int, then one LDS mean-field step."""
    hmm_global, lds_global = global_natparam
    dev = node_potentials[0].device
    node = tuple(_dev64(x, dev) for x in node_potentials)
    B, T, n = node[1].shape
    K = torch.as_tensor(hmm_global[0]).shape[0]
    labels = torch.as_tensor(labels, device=dev).long()
    ind = torch.nn.functional.one_hot(labels, K).to(torch.float64)             # (B,T,K)
    E_trans = torch.einsum("bti,btj->bij", ind[:, :-1], ind[:, 1:])
    soft = ind + 1e-2
    hmm_stats = (ind[:, 0], E_trans, soft / soft.sum(-1, keepdim=True))
    _, _, dense_init, dense_pair = global_to_local_maps(global_natparam, dev)
    lds_init, lds_pair = get_var_lds_local_natparam(dense_init, dense_pair, hmm_stats[2])
    plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
    lds_vlb, (Ei, Ep, En) = _lds_estep_batched_init(plan, lds_init, lds_pair, node)
    lds_stats = ((Ei[0].clone(), Ei[1].clone()), tuple(x.clone() for x in Ep[:3]), tuple(x.clone() for x in En[:2]))
    return (hmm_stats, lds_stats), (None, (lds_init, lds_pair)), (torch.zeros_like(lds_vlb), lds_vlb.clone())


def _lds_estep_batched_init(plan, lds_init, lds_pair, node, keep_factor=False, reference_compat=True, eps=None,
                            inplace=False):
    """LDS E-step with a PER-SEQUENCE init potential (the SLDS mixes K init potentials by E[z_0]):
    run the kernel with a zero shared init potential and add each sequence's (J0, h0) to its first
    node potential's dense block... the kernel's node potentials are diagonal, so instead the init
    potential is passed through the batched pair-parameter path: J11 of step 0 absorbs J0, and h0 is
    added to node_h[:,0]; log-normaliser constants are added back on the host.
    eps (B,T,S,n): E-step + backward sampler in ONE call (svae_lds_inference_f64, forward values only: lean records for
    batches above 1024 sequences); the samples are left in plan._infer_samples.  inplace: J0 is added INTO the caller's
    J11[:, 0] (the caller owns the per-step parameters and will not reuse them: saves a copy of (B,T-1,n,n))."""
    J0, h0, a0, b0 = lds_init                      # (B,n,n), (B,n), (B), (B)
    J11, J12, J22, lz = lds_pair                   # (B,T-1,...)
    B, T, n = node[1].shape
    dev = node[1].device
    nJ, nh = node[0], node[1].clone()
    nh[:, 0] += h0
    zero = torch.zeros((), dtype=torch.float64, device=dev)
    if T > 1:
        if not (inplace and J11.is_contiguous()):
            J11 = J11.clone()
        J11[:, 0] += J0                            # -1/2 x0' J0 x0 multiplies the same variable as J11[0]
        natparam = ((torch.zeros(n, n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev), zero),
                    (J11.contiguous(), J12.contiguous(), J22.contiguous(), lz.contiguous()))
        lognorm, stats = natural_lds_estep_general(natparam, (nJ, nh) + tuple(node[2:]), plan=plan,
                                                   keep_factor=keep_factor, _infer_eps=eps)
        return (lognorm + a0 if reference_compat else lognorm + a0 + b0), stats
    raise NotImplementedError("SLDS needs T > 1")


def get_global_stats(hmm_stats, init_stats, pair_stats, pair_sums=None):
    """(:229-243) -> (hmm (E_init, E_trans) summed over the batch, per state k: (init stats, pair stats)).
    pair_sums: the weighted sums of the pair statistics (K,3,n,n) where final_pass_contractions has formed them."""
    Ei, Et, Es = hmm_stats
    w0, w1 = Es[:, 0], Es[:, 1:]
    ExxT0, Ex0 = init_stats
    ones = torch.ones_like(w0)
    B, n = Ex0.shape
    K = w0.shape[1]
    g_init = ((w0.T @ ExxT0.reshape(B, n * n)).reshape(K, n, n), w0.T @ Ex0, w0.sum(0), w0.sum(0))
    # sum_{b,t} w1[b,t,k] * stats[b,t]: per sequence (K x T-1)(T-1 x 3n^2), then the batch sum (one long
    # reduction axis as a single GEMM is pathological in rocBLAS: 110 ms at B T = 1e6)
    gp = pair_sums if pair_sums is not None else \
        torch.bmm(w1.transpose(1, 2), _packed_pair_stats(pair_stats)).sum(0).reshape(K, 3, n, n)
    # (the sum over the strided view Es[:, 1:] costs 0.29 ms at configs[3]; over the contiguous tensor 0.02: measured)
    g_pair = (gp[:, 0], gp[:, 1], gp[:, 2], Es.sum((0, 1)) - w0.sum(0) if Es.is_contiguous() else w1.sum((0, 1)))
    return (Ei.sum(0), Et.sum(0)), (g_init, g_pair)


def global_stats_as_natparam(stats):
    """Re-nest get_global_stats' stacked tensors like the global natural parameters
    ((dirichlet (K), dirichlet rows (K,K)), [(niw dense (n+2,n+2), mniw 4-tuple)] * K), so that the
    natural-gradient expression of make_gradfun (svae.py:33-34: prior + stats - params on the
    flattened structures) lines up (the reference zips per-state tuples the same way, :240)."""
    hmm, (g_init, g_pair) = stats
    K = g_init[2].shape[0]
    lds = [(expfam.pack_dense(g_init[0][k], g_init[1][k], g_init[2][k], g_init[3][k]),
            tuple(p[k] for p in g_pair)) for k in range(K)]
    return hmm, lds


def run_inference(prior_natparam, global_natparam, nn_potentials, num_samples, init_eps=None, eps=None,
                  generator=None, tol=1e-2, group=None, reference_compat=True):
    """(:289-310) -> (samples (B,T,S,n), expected_stats, global_vlb, local_vlb); forward values only
    (see run_inference_differentiable).  Under torch.distributed the sequences are this rank's shard (every
    sequence runs its own coordinate ascent: no collective inside it); the statistics and local_vlb are summed
    over the ranks of `group` with ONE all-reduce."""
    dev = nn_potentials[1].device
    node = tuple(_dev64(x, dev) for x in nn_potentials)
    B, T, n = node[1].shape
    if init_eps is None:
        init_eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=generator)
    # (host copies of the global parameters for the prior term FIRST: made behind the kernels below, the device-to-host
    #  copies of device-resident parameters would wait for all of them and leave the host arithmetic of slds_prior_vlb,
    #  3.4 ms, for afterwards instead of next to the final pass)
    host_params = (_HostParamsLater(global_natparam), _HostParamsLater(prior_natparam))
    maps = global_to_local_maps(global_natparam, dev)
    (hmm_stats, _), (hmm_nat, (lds_init, lds_pair)), _, _ = optimize_local_meanfield(
        global_natparam, node, init_eps, tol, pair_stats=False, reference_compat=reference_compat, local_maps=maps)
    plan = LDSEStepPlan(B, T, n, dev, inhomog=True, pair_batched=True)
    S = int(num_samples)
    if eps is None:
        eps = torch.randn(B, T, S, n, dtype=torch.float64, device=dev, generator=generator)
    eps = _dev64(eps, dev)
    if n <= _lib.LDS_MAX_N and S <= 16:
        # final E-step + sampler in ONE call on the per-step parameters of the converged mean field (lean records above
        # 1024 sequences: csrc/lds_lean_estep.hpp, INH), J0 added into the mixed J11 in place
        lognorm, (Ei, Ep, En) = _lds_estep_batched_init(plan, lds_init, lds_pair, node, keep_factor=True,
                                                        reference_compat=reference_compat, eps=eps, inplace=True)
        samples = plan._infer_samples
    else:
        lognorm, (Ei, Ep, En) = _lds_estep_batched_init(plan, lds_init, lds_pair, node, keep_factor=True,
                                                        reference_compat=reference_compat)
        samples = plan.sample(eps)
    _, _, dense_init, dense_pair = maps
    fused = final_pass_contractions(dense_init, dense_pair, (Ei[0], Ei[1]), plan.E_pair, hmm_stats[2])
    node_hmm, pair_sums = fused if fused is not None else \
        (get_arhmm_local_nodeparams(dense_init, dense_pair, (Ei[0], Ei[1]), plan.E_pair), None)
    hmm_vlb, _ = hmm_estep((hmm_nat[0], hmm_nat[1], node_hmm))
    expected_stats = get_global_stats(hmm_stats, (Ei[0], Ei[1]), plan.E_pair, pair_sums)
    lds_vlb = lognorm - ((node[0] * En[0]).sum((1, 2)) + (node[1] * En[1]).sum((1, 2)))
    local_vlb = (hmm_vlb + lds_vlb).sum()
    expected_stats, local_vlb = allreduce_nested(expected_stats, local_vlb, group)
    global_vlb = slds_prior_vlb(host_params[0], host_params[1], dev)
    return samples, expected_stats, global_vlb, local_vlb


def _slds_params_on_host(natparam):
    """A nested copy of SLDS global natural parameters on the host.  Host-resident parameters are used as they are; a
    `_HostParamsLater` (device-resident ones on their way, below) is waited for here."""
    if isinstance(natparam, _HostParamsLater):
        return natparam.get()
    cpu = torch.device("cpu")
    (d, md), lds = natparam
    return (_dev64(d, cpu), _dev64(md, cpu)), [(_dev64(a, cpu), tuple(_dev64(y, cpu) for y in m)) for a, m in lds]


class _HostParamsLater(object):
    """Device-resident SLDS global parameters on their way to the host for the prior term (slds_prior_vlb is K small
    matrices of host arithmetic): ONE device-side concatenation, ONE asynchronous copy into pinned memory and an event,
    issued BEFORE the local step's kernels are queued -- the host goes on queueing and only waits when it needs the
    values, by which time the copy has long landed.  (Round 5 made ~50 blocking device-to-host copies at this point:
    0.6 ms during which nothing was queued; made behind the kernels they waited for all of them.)"""

    def __init__(self, natparam):
        (d, md), lds = natparam
        self.leaves = [d, md] + [x for a, m in lds for x in (a,) + tuple(m)]
        self.K = len(lds)
        self.m_len = len(lds[0][1]) if lds else 0
        self.pending = None
        if all(isinstance(x, torch.Tensor) and x.is_cuda for x in self.leaves):
            flat = torch.cat([x.detach().to(torch.float64).reshape(-1) for x in self.leaves])
            self.host = torch.empty(flat.shape, dtype=torch.float64, device="cpu", pin_memory=True)
            self.host.copy_(flat, non_blocking=True)
            self.pending = torch.cuda.Event()
            self.pending.record()
            self._flat = flat                                   # (kept alive until the copy has completed)
        else:
            self.value = _slds_params_on_host(natparam)

    def get(self):
        if self.pending is not None:
            self.pending.synchronize()
            out, o = [], 0
            for x in self.leaves:
                k = x.numel()
                out.append(self.host[o:o + k].reshape(x.shape).clone())
                o += k
            it = iter(out)
            d, md = next(it), next(it)
            lds = []
            for _ in range(self.K):
                a = next(it)
                lds.append((a, tuple(next(it) for _ in range(self.m_len))))
            self.value, self.pending, self._flat = ((d, md), lds), None, None
        return self.value


def slds_prior_vlb(global_natparam, prior_natparam, dev):
    """(:248-286) <prior - global, E_global[stats]> - (logZ(prior) - logZ(global)).
    K small matrices: evaluated on the host, one scalar moved back (parameters that already are host copies --
    _slds_params_on_host -- are used as they are)."""
    out_dev = dev
    (gd, gmd), glds = _slds_params_on_host(global_natparam)
    (pd, pmd), plds = _slds_params_on_host(prior_natparam)
    val = ((pd - gd) * expfam.dirichlet_expectedstats(gd)).sum() + ((pmd - gmd) * expfam.dirichlet_expectedstats(gmd)).sum()
    logZ = lambda d, md, lds: expfam.dirichlet_logZ(d) + expfam.dirichlet_logZ(md) + \
        sum(expfam.niw_logZ(a) + expfam.mniw_logZ(m) for a, m in lds)
    for (ga, gm), (pa, pm) in zip(glds, plds):
        val = val + ((pa - ga) * expfam.niw_expectedstats(ga)).sum()
        val = val + sum(((x - y) * e).sum() for x, y, e in zip(pm, gm, expfam.mniw_expectedstats(gm)))
    return (val - (logZ(pd, pmd, plds) - logZ(gd, gmd, glds))).to(out_dev)


def final_pass_differentiable(global_natparam, hmm_natparam, lds_natparam, nn_potentials, eps,
                              reference_compat=True, local_maps=None, expected_states=None):
    """The part of run_inference that depends on nn_potentials with gradients attached
    (slds_svae.py:295-307, "recompute terms that depend on nn_potentials at optimum"): the LDS
    E-step + sampler on the FIXED mean-field natural parameters, the HMM bound evaluated on its
    statistics, and the local bound.  Returns (samples, (E_init, E_pair) per sequence, local_vlb); with
    `expected_states` (the HMM marginals of the ascent) a fourth entry: the pair statistics summed with those weights
    (K,3,n,n) for get_global_stats, or None -- formed by the kernel that also builds the HMM node potentials."""
    dev = nn_potentials[1].device
    nJ, nh = nn_potentials[0], nn_potentials[1]
    B, T, n = nh.shape
    (J0, h0, a0, b0), (J11, J12, J22, lz) = lds_natparam
    # per-sequence init potential folded into pair 0 / node 0 (see _lds_estep_batched_init)
    J11 = J11.clone()
    J11[:, 0] += J0
    first = torch.zeros_like(nh)
    first[:, 0] = h0
    nh_eff = nh + first
    zero = torch.zeros((), dtype=torch.float64, device=dev)
    natparam = ((torch.zeros(n, n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev), zero),
                (J11.contiguous(), J12.contiguous(), J22.contiguous(), lz.contiguous()))
    lognorm, (dxx, ex), samples, (E_init, E_pair) = lds_inference_differentiable(natparam, (nJ, nh_eff), eps=eps)
    lognorm = lognorm + a0 if reference_compat else lognorm + a0 + b0
    _, _, dense_init, dense_pair = local_maps if local_maps is not None else global_to_local_maps(global_natparam, dev)
    init_stats = (E_init[:, :n * n].reshape(B, n, n), E_init[:, n * n:])
    pair_stats = (E_pair[:, :, 0], E_pair[:, :, 1], E_pair[:, :, 2])
    pair_sums = None
    K = dense_init[0].shape[0]
    no_global_grad = not any(isinstance(x, torch.Tensor) and x.requires_grad for x in tuple(dense_init) + tuple(dense_pair))
    if expected_states is not None and no_global_grad and _pair_contract_applies(dense_init, E_pair):
        node_hmm, pair_sums = _PairContract.apply(E_pair, init_stats[0], init_stats[1], dense_init, dense_pair,
                                                  expected_states)
    else:
        node_hmm = get_arhmm_local_nodeparams(dense_init, dense_pair, init_stats, E_pair)
    hmm_vlb = hmm_logZ_differentiable((hmm_natparam[0], hmm_natparam[1], node_hmm))
    lds_vlb = lognorm - ((nJ * dxx).sum((1, 2)) + (nh * ex).sum((1, 2)))
    if expected_states is not None:
        return samples, (init_stats, pair_stats), (hmm_vlb + lds_vlb).sum(), pair_sums
    return samples, (init_stats, pair_stats), (hmm_vlb + lds_vlb).sum()


def run_inference_differentiable(prior_natparam, global_natparam, nn_potentials, num_samples, init_eps=None,
                                 eps=None, generator=None, tol=1e-2, group=None, reference_compat=True):
    """run_inference (slds_svae.py:289-310) with torch autograd attached to nn_potentials = (J, h),
    each (B,T,n): the local mean field is optimised on detached values (the reference's `unbox`),
    then the final pass is differentiated through the E-step / sampler VJP kernels and the HMM
    kernel.  -> (samples (B,T,S,n), expected_stats, global_vlb, local_vlb)."""
    dev = nn_potentials[1].device
    node_d = tuple(_dev64(x, dev) for x in nn_potentials)
    B, T, n = node_d[1].shape
    if init_eps is None:
        init_eps = torch.randn(B, T, 1, n, dtype=torch.float64, device=dev, generator=generator)
    host_params = (_HostParamsLater(global_natparam), _HostParamsLater(prior_natparam))
    maps = global_to_local_maps(global_natparam, dev)
    (hmm_stats, _), (hmm_nat, lds_nat), _, _ = optimize_local_meanfield(
        global_natparam, node_d, init_eps, tol, pair_stats=False, reference_compat=reference_compat, local_maps=maps)
    if eps is None:
        eps = torch.randn(B, T, int(num_samples), n, dtype=torch.float64, device=dev, generator=generator)
    samples, (init_stats, pair_stats), local_vlb, pair_sums = final_pass_differentiable(
        global_natparam, hmm_nat, lds_nat, (nn_potentials[0], nn_potentials[1]), _dev64(eps, dev), reference_compat, maps,
        hmm_stats[2])
    expected_stats = get_global_stats(hmm_stats, tuple(x.detach() for x in init_stats),
                                      tuple(x.detach() for x in pair_stats), pair_sums)
    expected_stats, local_vlb = allreduce_nested(expected_stats, local_vlb, group)
    global_vlb = slds_prior_vlb(host_params[0], host_params[1], dev)
    return samples, expected_stats, global_vlb, local_vlb


def make_hmm_global_natparam(num_states, alpha=1., sticky_bias=0., random=False, generator=None,
                             dtype=torch.float64, device="cpu"):
    """(slds_svae.py:38-50) -> (dirichlet natparam of the initial state (K), of the transition rows (K,K))."""
    kw = dict(dtype=dtype, device=device)
    row = lambda: alpha * torch.ones(num_states, **kw) if not random \
        else alpha + torch.rand(num_states, generator=generator, **kw)
    dir_natparam = row()
    mdir_natparam = sticky_bias * torch.eye(num_states, **kw) + torch.stack([row() for _ in range(num_states)])
    return dir_natparam, mdir_natparam


def make_lds_global_natparams(num_states, state_dim, random=False, generator=None, dtype=torch.float64,
                              device="cpu"):
    """(slds_svae.py:56-75) -> [(NIW natparam, MNIW natparam)] per discrete state."""
    kw = dict(dtype=dtype, device=device)
    n = state_dim
    eye = torch.eye(n, **kw)
    r = lambda: float(torch.rand((), generator=generator, **kw))
    out = []
    for _ in range(num_states):
        if not random:
            nu, S, mu, kappa = n + 10., (n + 10.) * eye, torch.zeros(n, **kw), 10.
            nu2, S2, M, K = n + 10., (n + 10.) * eye, torch.zeros(n, n, **kw), 10. * eye
        else:
            nu, S, mu, kappa = n + 4. + r(), (n + r()) * eye, torch.randn(n, generator=generator, **kw), r()
            nu2, S2, M, K = n + 4. + r(), (n + r()) * eye, 1e-2 * torch.randn(n, n, generator=generator, **kw), (n + r()) * eye
        t = lambda x: torch.as_tensor(x, **kw)
        out.append((expfam.niw_standard_to_natural(S, mu, t(kappa), t(nu)),
                    expfam.mniw_standard_to_natural(t(nu2), S2, M, K)))
    return out


def make_slds_global_natparam(num_states, state_dim, alpha=5., sticky_bias=0., random=False, generator=None,
                              dtype=torch.float64, device="cpu"):
    """(slds_svae.py:27-31)"""
    return (make_hmm_global_natparam(num_states, alpha, sticky_bias, random, generator, dtype, device),
            make_lds_global_natparams(num_states, state_dim, random, generator, dtype, device))
