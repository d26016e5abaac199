/* svae_hip.h -- C ABI of libsvae_hip.so: the MI355X (gfx950) structured E-step of mattjj/svae.
 *
 * Every entry point replaces one function at the reference's Python->Cython boundary
 * (/root/reference/svae/lds/lds_inference.py:18-24; svae/hmm/hmm_inference.py:11-13) or one hot loop
 * of svae/models/gmm.py, batched over independent sequences / data points.  Plain pointers and sizes only; no torch types.
 *
 * Conventions (identical to the reference's API level):
 *   - all arrays are float64, C-order, time-major, and hold NATURAL parameters:
 *       Gaussian potentials are (-1/2 J, h); pair potentials are
 *       (J11 = -1/2 A'Q^-1 A, J12 = A'Q^-1, J22 = -1/2 Q^-1, logZ)        [gaussian.py:130-143]
 *   - every pointer is a DEVICE pointer owned by the caller; calls are asynchronous on `stream`
 *     (a hipStream_t, passed as void*); nothing is allocated internally.
 *   - return value: 0 = launched, <0 = bad argument number -k (nothing launched).
 *   - numerical failures (non-positive pivot = potentials not positive definite; the reference
 *     silently ignores LAPACK `info`, cython_gaussian_grads.pxd:54-76) are reported through the
 *     device-side `info` word: 0 = ok, k>0 = 1 + index of the first offending sequence/point.
 */
#ifndef SVAE_HIP_H
#define SVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVAE_HIP_ABI_VERSION 14   /* 14: + svae_lds_estep_vjp_dense_f64 (cotangent of dense node potentials), svae_hmm_* up to K = 64, svae_lds_inference_f64 (E-step + sampler in one call; lean per-step records for large homogeneous batches), svae_lds_inference_is_lean, SVAE_OPT_LEAN_ON / _OFF / SVAE_OPT_INFER_RECORDS; 13: + svae_slds_pair_contract_f64 (the two contractions of the SLDS final pass over the per-step pair statistics in one pass); 12: svae_gmm_global_step_f64 writes kl[0..1] (as spelled | as shipped), svae_ipc_allreduce_f64 takes the mailbox stride and never writes `out` on a timeout, + svae_slds_lds_meanfield options; 11: + svae_lds_global_step_multi_f64 (K parameter sets in one launch: the SLDS global -> local maps), svae_lds_diag_sample_f64 (filter + sampler of an all-diagonal LDS: the SLDS initial path); 10: + svae_ipc_allreduce_f64 / svae_ipc_mailbox_bytes, svae_gmm_sample_f64, svae_gmm_local_vjp_f64, svae_gmm_global_step_f64 (the differentiable tail and the global side of the GMM local step); 9: keep bit SVAE_KEEP_SIGMA of svae_lds_estep_f64 (16 <= n <= 64) + svae_lds_tile_sigma_offset_bytes; 8: step ranges (t_begin, t_end) in svae_lds_tile_vjp_f64 / svae_lds_tile_noise_f64, SVAE_OPT_TILE_FORWARD / _BACKWARD; 7: + svae_slds_path_nodeparams_f64, svae_slds_mix_pair_natparam_f64; 6: per-call `options` word replaces the process-global svae_lds_set_* selectors (re-entrant library), + svae_slds_hmm_meanfield_f64, svae_slds_sweep_glue_f64, g_E_pair in svae_lds_tile_vjp_f64; 5: + svae_lds_set_prod_max_b; 4: + svae_slds_lds_meanfield_f64, svae_gmm_mw_*, svae_lds_global_step_f64, svae_lds_natgrad_f64, svae_lds_tile_vjp_f64; 2: + svae_lds_workspace_bytes_ex, svae_lds_estep_vjp_ex_f64, svae_hmm_*, tiled path (n <= 64) */
#define SVAE_HMM_MAX_K 64   /* svae_hmm_estep_f64 / svae_slds_hmm_meanfield_f64: K <= 16 one DPP row per sequence; 17 <= K <= 64 one wavefront per sequence (round 6) */
#define SVAE_LDS_MAX_N 15   /* register/DPP path: one 16-lane row per sequence, n+1 <= 16 */
#define SVAE_LDS_TILE_MAX_N 64   /* 16 <= n <= 64: LDS-tiled MFMA path (keep: SVAE_KEEP_SIGMA or 0) */
#define SVAE_KEEP_SIGMA 4         /* keep bit of svae_lds_estep_f64, 16 <= n <= 64 only: see svae_lds_tile_sigma_offset_bytes */

/* Library/ABI version (host only, no GPU needed). */
int svae_hip_abi_version(void);

/* Bytes of scratch `svae_lds_estep_f64` / `svae_lds_sample_f64` need for (B, T, n): the per-step
 * backward kernels (G_t, c_t, P_t^-1: (2n+1)n doubles) written by the forward filter, followed by
 * the factor region (unit LDL' factor + pivots of P_t: n*n+n doubles) the sampler reads and the
 * cross-moment region (W~_t) the VJP reads. */
size_t svae_lds_workspace_bytes(int B, int T, int n);

/* Same, exact for the tiled path (n > SVAE_LDS_MAX_N), whose workspace also holds the pair
 * parameters re-packed in MFMA fragment order: 2 slots when they are homogeneous (what
 * svae_lds_workspace_bytes assumes), T-1 slots per parameter set when they are per-step
 * (`inhomog`), one set per sequence when `pair_batched`.  Equal to svae_lds_workspace_bytes for
 * n <= SVAE_LDS_MAX_N. */
size_t svae_lds_workspace_bytes_ex(int B, int T, int n, int inhomog, int pair_batched);
/* 16 <= n <= 64, keep & SVAE_KEEP_SIGMA: the backward half of the E-step also leaves the smoothed covariances
 * Sigma_t = Cov(x_t | y) as a compact (B,T,n,n) array at THIS byte offset of the workspace (svae_lds_workspace_bytes_ex
 * rounded up to 256), which must then be at least offset + B T n n 8 bytes long.  That array is the first section of the
 * workspace of svae_lds_tile_vjp_f64: a caller that lets its VJP workspace START at workspace + offset (one allocation
 * of offset + 8 svae_lds_tile_vjp_workspace_doubles(..) bytes) skips phase 0 of the VJP, which would rebuild the same
 * matrices from the hand-off.  0 for n <= 15. */
size_t svae_lds_tile_sigma_offset_bytes(int B, int T, int n, int inhomog, int pair_batched);

/* Batched LDS E-step = filter + RTS smoother + expected sufficient statistics + log-normalizer.
 *
 * Replaces, for B independent sequences sharing (init, pair) parameters,
 *   cython_natural_lds_estep_general(natparam, node_params) -> (lognorm, expected_stats)
 *     /root/reference/svae/lds/lds_inference.py:232-237, i.e.
 *   natural_filter_forward_general  /root/reference/svae/lds/cython_lds_inference.pyx:28-90
 *   natural_smoother_general        /root/reference/svae/lds/cython_lds_inference.pyx:149-195
 *   _compute_stats                  /root/reference/svae/lds/cython_lds_inference.pyx:197-210
 *
 *  in : init_J (n,n), init_h (n), init_logZ (1)           = init_params (sum of trailing logZ terms)
 *       J11, J12, J22: (n,n) if !inhomog, else (T-1,n,n) per sequence-independent step;
 *       logZ_pair: (1) if !inhomog else (T-1)
 *       pair_batched != 0 (inhomog only): J11/J12/J22 are (B,T-1,n,n), logZ_pair (B,T-1)
 *         (the SLDS case, slds_svae.py:92-103, where pair params depend on each sequence's
 *          discrete-state marginals)
 *       keep: bit 0 = also write the factor region of the workspace so that svae_lds_sample_f64 can
 *         follow; bit 1 = also write the cross-moment region so that svae_lds_estep_vjp_f64 can
 *         follow (each costs a few % of the E-step); 16 <= n <= 64: SVAE_KEEP_SIGMA or 0
 *       node_J (B,T,n) diagonal of -1/2 precision, node_h (B,T,n), node_logZ (B,T) or NULL (=0)
 *  out: lognorm (B)
 *       E_init  (B, n*n + n)      = [E[x0 x0'] (n,n) | E[x0] (n)]   (the two trailing 1's of the
 *                                    reference tuple are implied)
 *       E_pair  (B, 3, n, n)      = [sum_t E[x_t x_t'], sum_t E[x_t x_{t+1}'], sum_t E[x_{t+1}x_{t+1}']]
 *                                    t = 0..T-2 (4th entry of the reference tuple is T-1)
 *                                    if inhomog: (B, T-1, 3, n, n) per-step blocks (no sum)
 *       E_node_diagxx (B,T,n) = diag E[x_t x_t'],   E_node_x (B,T,n) = E[x_t]
 *       info (1) int32, must be zeroed by the caller (or by a previous successful call)
 */
int svae_lds_estep_f64(int B, int T, int n, int inhomog, int pair_batched, int keep, unsigned options,
                       const double* init_J, const double* init_h, const double* init_logZ,
                       const double* J11, const double* J12, const double* J22,
                       const double* logZ_pair,
                       const double* node_J, const double* node_h, const double* node_logZ,
                       double* lognorm, double* E_init, double* E_pair,
                       double* E_node_diagxx, double* E_node_x,
                       int32_t* info, void* workspace, size_t ws_bytes, void* stream);

/* Filter only = natural_filter_forward_general (/root/reference/svae/lds/cython_lds_inference.pyx:28-90),
 * the first of the six functions the reference imports at lds_inference.py:18-24, batched.  Same inputs as
 * svae_lds_estep_f64 (n <= SVAE_LDS_MAX_N).
 *  out: lognorm (B); the forward messages in the reference's scaling (natural parameters, :84-85), each
 *       optional (NULL = not wanted):  J_pred, J_filt (B,T,n,n) = -1/2 precision,  h_pred, h_filt (B,T,n).
 * The workspace afterwards holds what svae_lds_sample_f64 needs: filter + sampler without the smoother is
 * cython_natural_lds_sample (lds_inference.py:260-264). */
int svae_lds_filter_f64(int B, int T, int n, int inhomog, int pair_batched, unsigned options,
                        const double* init_J, const double* init_h, const double* init_logZ,
                        const double* J11, const double* J12, const double* J22, const double* logZ_pair,
                        const double* node_J, const double* node_h, const double* node_logZ,
                        double* lognorm, double* J_pred, double* h_pred, double* J_filt, double* h_filt,
                        int32_t* info, void* workspace, size_t ws_bytes, void* stream);

/* `options` word of svae_lds_estep_f64 / svae_lds_filter_f64 / svae_lds_sample_f64 / svae_lds_estep_vjp_ex_f64
 * (n <= SVAE_LDS_MAX_N; ignored above).  0 = the library's own choice; the bits force one of the kernel variants --
 * all give the same results up to rounding -- for A/B measurements and for the tests that run every kernel.  The
 * library keeps NO process-global selection state and reads no environment variables: two host threads / streams /
 * devices may use different options at the same time.
 *
 * Default dispatch of svae_lds_estep_f64: with keep == 0, n <= 10 and T >= 4 the two-ended kernel (block elimination
 * from both ends of the chain, meeting in the middle: half the serial depth; lean hand-off record) -- up to 512
 * sequences one sequence per workgroup (shortest instruction stream per sequence), above two sequences per wavefront
 * (fewest instructions per sequence).  With keep != 0,
 * n <= 10, T >= 4 and B <= 1023 the call runs TWO kernels side by side, joined by events before it returns to the
 * caller's stream: the two-ended E-step (statistics, log-normaliser, and the cross moments the VJP reads) and the
 * one-directional filter (hand-off records and LDL' factors for svae_lds_sample_f64 / svae_lds_estep_vjp_f64); the
 * helper stream and the two events are created once per (device, caller stream).  Otherwise batches with B <= 1023
 * run the one-directional small-batch variant (one sequence per wavefront, product stages split across the four DPP
 * rows), larger ones the packed kernel (four sequences per wavefront).
 * svae_lds_sample_f64 (S <= 4) and the sweeps of svae_lds_estep_vjp_f64 run, for batches B <= 1024, with PRODUCER
 * wavefronts: the serial loop of a workgroup's sequences reads its per-step records from an LDS ring that more
 * wavefronts of the workgroup fill several steps ahead (and, in the first sweep, HELPER wavefronts take the work that
 * does not feed the recursion); larger batches hide the latency by occupancy instead. */
#define SVAE_OPT_DEFAULT        0x00u
#define SVAE_OPT_TWOEND_OFF     0x01u   /* never the two-ended kernel (one-directional kernels only) */
#define SVAE_OPT_TWOEND_FULL    0x02u   /* two-ended kernel with the full hand-off record (no lean record) */
#define SVAE_OPT_LAYOUT_SPLIT   0x04u   /* one sequence per wavefront whatever B */
#define SVAE_OPT_LAYOUT_PACKED  0x08u   /* several sequences per wavefront whatever B (two-ended kernel: two, one DPP row
                                           per elimination chain; one-directional kernels: four) */
#define SVAE_OPT_PRODUCERS_ON   0x10u   /* producer / helper wavefronts whatever B */
#define SVAE_OPT_PRODUCERS_OFF  0x20u   /* never */
#define SVAE_OPT_ALL            0x3fu   /* (contradictory pairs or unknown bits: the call returns -24) */
/* Record format of svae_lds_inference_f64 (below); accepted and ignored by the other entry points, so that a caller can
 * pass one word with every call of a plan. */
#define SVAE_OPT_LEAN_ON        0x100u  /* lean per-step records whatever B (where they apply: see svae_lds_inference_f64) */
#define SVAE_OPT_LEAN_OFF       0x200u  /* never */
#define SVAE_OPT_INFER_RECORDS  0x400u  /* svae_lds_estep_vjp_ex_f64 only: `workspace` was written by svae_lds_inference_f64 */
/* svae_lds_estep_f64 with 16 <= n <= 64 only (any other use: -24): run HALF of the E-step.  The forward half (filter,
 * hand-off, log-normaliser) and the backward half (smoother + statistics, from the hand-off a forward-only call left
 * in the same workspace) meet only through the hand-off, so a training step can run the backward half on one stream
 * NEXT to the kernels that need the hand-off alone (svae_lds_tile_noise_f64 + svae_lds_tile_sample_f64, phase 0 of
 * svae_lds_tile_vjp_f64) on others -- each is one workgroup per sequence or shorter than the smoother. */
#define SVAE_OPT_TILE_FORWARD   0x40u
#define SVAE_OPT_TILE_BACKWARD  0x80u

/* LDS mean-field step of the SLDS-SVAE coordinate ascent with the mixing of the K per-state parameter sets
 * and the contraction of the pair statistics FUSED into the E-step:
 *   lds_meanfield + get_var_lds_local_natparam  /root/reference/svae/models/slds_svae.py:80-103
 *   the pair part of get_arhmm_local_nodeparams /root/reference/svae/models/slds_svae.py:131-147
 * For sequence b the LDS has init potential sum_k w[b,0,k] (init_J_k, init_h_k) and, at step t, pair
 * parameters sum_k w[b,t+1,k] (J11_k, J12_k, J22_k) -- never materialised: the K sets sit in LDS, w streams.
 *  in : init_J (K,n,n), init_h (K,n), J11/J12/J22 (K,n,n) natural parameters per discrete state;
 *       weights (rows,T,K) = E[z_t = k]; node potentials (rows,T,n) as in svae_lds_estep_f64;
 *       seq_index (B) int32 or NULL: the launch processes B <= rows sequences, slot i working on row
 *       seq_index[i] of every array and of the workspace (NULL: row i); rows not listed are left untouched;
 *       a NEGATIVE entry marks an unused slot (skipped): a caller may launch on more slots than are live
 *       (converged sequences of the coordinate ascent are frozen, slds_svae.py:170-172: the caller lists the
 *       ones still iterating)
 *  out: lognorm (rows) WITHOUT the mixed constants sum_k w[b,0,k] init_logZ_k + sum_t sum_k w[b,t+1,k] logZ_k
 *       (a (B,T,K) x (K) product the caller adds); E_init, E_node_diagxx, E_node_x as in svae_lds_estep_f64;
 *       pair_contr (rows,T,2,K):  [b,t,0,k] = <E x_t x_t', J11_k> + <E x_t x_{t+1}', J12_k>   (t < T-1)
 *                              [b,t,1,k] = <E x_t x_t', J22_k>                              (t > 0)
 *         so that the HMM node potential of slds_svae.py:141-146 is
 *         node[b,t+1,k] = pair_contr[b,t,0,k] + pair_contr[b,t+1,1,k] + logZ_k
 *         (which of the two slots carries the cross term depends on the chain that owns node t; only the sum
 *          above is defined).
 * n <= 10, T >= 4, K <= 16 and svae_slds_lds_meanfield_lds_bytes(n, K) <= 160 KiB (else -4: use the
 * per-step entry point).  workspace: svae_slds_lds_meanfield_workspace_bytes(rows, T, n) (the two-ended kernel's
 * records only: a third of svae_lds_workspace_bytes).
 * options (ABI 12; 0 = the library's choice, a pure function of the arguments):
 *   SVAE_OPT_LAYOUT_PACKED  row-per-chain consumers (two sequences per wavefront, lean records) + producer wavefronts that
 *                           form the K-state mixing and the K-state contraction on v_mfma_f64_16x16x4, 8 sequences per
 *                           workgroup in lock-step (csrc/lds_estep_twoend_rpcmix.hpp): K <= 8 and a workspace below
 *                           4 GiB, else -24; the default where it applies;
 *   SVAE_OPT_LAYOUT_SPLIT   one sequence per wavefront, the K parameter sets as LDS tables (rounds 2 - 4; any K <= 16);
 *   SVAE_OPT_PRODUCERS_OFF  with the packed layout: the producers compute with plain loops instead of MFMA (slow; test
 *                           infrastructure that separates the lock-step protocol from the MFMA operand layouts). */
size_t svae_slds_lds_meanfield_lds_bytes(int n, int K);
size_t svae_slds_lds_meanfield_workspace_bytes(int rows, int T, int n);
int svae_slds_lds_meanfield_f64(int B, int rows, int T, int n, int K,
                                const double* init_J, const double* init_h,
                                const double* J11, const double* J12, const double* J22,
                                const double* weights,
                                const double* node_J, const double* node_h, const double* node_logZ,
                                const int32_t* seq_index,
                                double* lognorm, double* E_init, double* E_node_diagxx, double* E_node_x,
                                double* pair_contr, int32_t* info,
                                void* workspace, size_t ws_bytes, unsigned options, void* stream);

/* Reverse-mode derivative of the E-step (+ sampler) w.r.t. the node potentials for latent dimension 16 <= n <= 64,
 * on the hand-off the LDS-tiled E-step kernel leaves in `handoff_workspace` (the workspace of the LAST
 * svae_lds_estep_f64 call with this (B,T,n)): natural_filter_grad / natural_smoother_general_grad /
 * natural_sample_backward_grad, /root/reference/svae/lds/cython_lds_inference.pyx:92-145, 236-306, 357-409,
 * composed as /root/reference/svae/lds/lds_inference.py:26-39.  Three phases, one call each, in order:
 *   0: smoothed covariances (backward in time); 1: adjoint of the smoother / sampler recursions (forward in
 *   time); 2: adjoint of the filter (backward in time) -> g_node_J, g_node_h (B,T,n).
 * Between phases 1 and 2 the caller adds the Cholesky adjoint of the sampler's noise factor -- batched over all
 * (sequence, step) pairs, from `xbar` and eps -- into the `pinv_bar` section of `workspace`
 * (layout [sig (B,T,n,n) | pinv_bar (B,T,n,n) | g_bar (B,T-1,n,n) | c_bar (B,T,n) | xbar (B,T,S,n)],
 * svae_lds_tile_vjp_workspace_doubles); without sample cotangents (g_samples NULL) there is nothing to add.
 * Step ranges: phases 0 and 1 take (t_begin, t_end) = (0, T).  Phase 2 may be split into launches over
 * [t_begin, t_end), called from the LAST range down to the first (the recursion runs backward in time; its state
 * travels through the tail of `workspace`): a caller can then run the Cholesky adjoint of the next (earlier) range
 * (svae_lds_tile_noise_f64, mode 1, same range arguments) on another stream NEXT to phase 2 of the current one -- one
 * workgroup per sequence leaves most of the chip to it.  Bad ranges: -20.
 * Cotangents: g_lognorm (B); g_E_node_diagxx, g_E_node_x (B,T,n) or NULL; g_E_init (B, n*n+n) or NULL;
 * g_E_pair (B,T-1,3,n,n) or NULL: of the per-step pair statistics (inhomog only; _compute_stats_grad,
 * cython_lds_inference.pyx:212-234); g_samples (B,T,S,n) or NULL with the samples drawn (S <= 16).  J12 as in svae_lds_estep_vjp_ex_f64. */
size_t svae_lds_tile_vjp_workspace_doubles(int B, int T, int n, int S);
int svae_lds_tile_vjp_f64(int phase, int B, int T, int n, int S, int t_begin, int t_end, int inhomog, int pair_batched,
                          const double* J12, const double* g_lognorm, const double* g_E_node_diagxx,
                          const double* g_E_node_x, const double* g_E_init, const double* g_E_pair,
                          const double* g_samples,
                          const double* samples, const double* E_node_x, double* g_node_J, double* g_node_h,
                          const void* handoff_workspace, void* workspace, size_t ws_doubles, void* stream);

/* Backward sampling (natural_sample_backward, /root/reference/svae/lds/cython_lds_inference.pyx:310-355) for latent
 * dimension 16 <= n <= 64 from the hand-off of the tiled svae_lds_estep_f64: the serial recursion
 * x_t = c_t + noise_t + G_t x_{t+1}; `noise` (B,T,S,n) = chol(P_t)^-T eps_t is the caller's batched factorisation
 * of the hand-off's P_t^-1 (it does not depend on the recursion).  S <= 16. */
int svae_lds_tile_sample_f64(int B, int T, int n, int S, const double* noise, double* samples,
                             const void* handoff_workspace, void* stream);

/* The sampler's noise factor for latent dimension 16 <= n <= 64 and its adjoint, from the P_t^-1 of the tiled
 * E-step's hand-off, one workgroup per (sequence, step): the reference's noise map is chol(P_t)^-T eps_t
 * (/root/reference/svae/lds/cython_gaussian_grads.pxd:431-454), and chol(P_t)^-T is the upper-triangular M with
 * P_t^-1 = M M'.
 *   mode 0: noise (B,T,S,n) = M_t eps_t   (input of svae_lds_tile_sample_f64)
 *   mode 1: adds the Cholesky-path cotangent sym(M^-T Phi(M' Mbar) M^-1), Mbar = triu(sum_s xbar_s eps_s'), into the
 *           pinv_bar section of `vjp_workspace` -- between phases 1 and 2 of svae_lds_tile_vjp_f64
 *           (the adjoint of cython_gaussian_grads.pxd:456-487 `_natural_sample_grad`).           S <= 16.
 * Only the steps t_begin <= t < t_end of every sequence are processed ((0, T): all; bad ranges: -20). */
int svae_lds_tile_noise_f64(int mode, int B, int T, int n, int S, int t_begin, int t_end, const double* eps, double* noise,
                            const void* handoff_workspace, void* vjp_workspace, int32_t* info, void* stream);

/* The once-per-step GLOBAL side of the LDS-SVAE in one launch ("global->local maps
 * on device"): niw.expectedstats (/root/reference/svae/distributions/niw.py:15-25) and mniw.expectedstats
 * (/root/reference/svae/distributions/mniw.py:19-20, 33-55) of the global factors -> the LDS init and pair
 * potentials (svae/models/lds.py:23-25), and the prior KL of svae/models/lds.py:16-20 (with niw.logZ niw.py:27-31
 * and mniw.logZ mniw.py:13-17) against `prior_*` (all five NULL: no KL).
 *  in : niw (n+2,n+2) dense-packed NIW natural parameter; mniw_A, mniw_B, mniw_C (n,n), mniw_d (1)
 *  out: init_J (n,n) = -1/2 E[J], init_h (n) = E[h], init_logZ (1) = -1/2 E[h'J^-1 h] + 1/2 E[log|J|];
 *       J11, J12, J22 (n,n), logZ_pair (1) = the MNIW expected statistics;
 *       niw_expectedstats (n+2,n+2) dense-packed or NULL; global_kl (1) or NULL;
 *       info: 1 if an inversion met a non-positive pivot (natural parameters outside the domain).   n <= 64. */
int svae_lds_global_step_f64(int n, const double* niw, const double* mniw_A, const double* mniw_B,
                             const double* mniw_C, const double* mniw_d,
                             const double* prior_niw, const double* prior_A, const double* prior_B,
                             const double* prior_C, const double* prior_d,
                             double* init_J, double* init_h, double* init_logZ,
                             double* J11, double* J12, double* J22, double* logZ_pair,
                             double* niw_expectedstats, double* global_kl, int32_t* info, void* stream);

/* The same maps for K parameter sets in ONE launch (the SLDS global -> local maps, get_all_lds_local_natparams,
 * /root/reference/svae/models/slds_svae.py:86-89: K factor pairs, no prior / KL): the five inputs are HOST arrays of K
 * device pointers (shapes as above), the outputs are stacked over the leading K axis -- init_J (K,n,n), init_h (K,n),
 * init_logZ (K), J11 / J12 / J22 (K,n,n), logZ_pair (K), niw_expectedstats (K,n+2,n+2) or NULL.   K <= 16. */
int svae_lds_global_step_multi_f64(int K, int n, const double* const* niw, const double* const* mniw_A,
                                   const double* const* mniw_B, const double* const* mniw_C,
                                   const double* const* mniw_d,
                                   double* init_J, double* init_h, double* init_logZ,
                                   double* J11, double* J12, double* J22, double* logZ_pair,
                                   double* niw_expectedstats, int32_t* info, void* stream);

/* Filter + backward sampler (cython_natural_lds_sample, /root/reference/svae/lds/lds_inference.py:260-264) of an LDS
 * whose natural parameters are all DIAGONAL -- the random-walk model of initialize_local_meanfield
 * (/root/reference/svae/models/slds_svae.py:203-226) under diagonal recognition potentials: n independent scalar
 * recursions per sequence, one lane each.  Same eps -> sample map as svae_lds_filter_f64 + svae_lds_sample_f64 on the
 * dense form of the same model (one sample per sequence).
 *  in : init_J, init_h (n); J11, J12, J22 (n): the diagonals (natural parameters as in svae_lds_estep_f64);
 *       node_J, node_h (B,T,n); eps (B,T,n)
 *  out: samples (B,T,n); info: b + 1 of a sequence with a non-positive precision, else untouched
 *  workspace: svae_lds_diag_sample_workspace_bytes(B,T,n) */
size_t svae_lds_diag_sample_workspace_bytes(int B, int T, int n);
int svae_lds_diag_sample_f64(int B, int T, int n, const double* init_J, const double* init_h,
                             const double* J11, const double* J12, const double* J22,
                             const double* node_J, const double* node_h, const double* eps,
                             double* samples, int32_t* info, void* workspace, size_t ws_bytes, void* stream);

/* Natural-gradient expression of /root/reference/svae/svae.py:33-34 for the LDS global parameter, straight from
 * the (all-reduced) buffer of svae_lds_reduce_stats_f64:
 *   natgrad = -scale * (prior + num_batches * stats - params)
 * over the flat parameter [NIW dense (n+2)^2 | A (n^2) | B (n^2) | C (n^2) | d (1)], where stats =
 * (pack_dense(sum E[x0 x0'], sum E[x0], count, count), (E_pair sums, count (T-1))).  One launch. */
int svae_lds_natgrad_f64(int n, int T, const double* packed_stats, const double* prior_flat,
                         const double* params_flat, double num_batches, double scale,
                         double* natgrad_flat, void* stream);

/* Deterministic sum over the batch of the per-sequence global statistics (the quantity that is
 * all-reduced across GPUs for the natural-gradient step, svae.py:33-34):
 *   out (n*n + n + 3*n*n + 2) = [sum_b E_init | sum_b E_pair | sum_b lognorm | B]
 * Homogeneous E_pair layout only. */
int svae_lds_reduce_stats_f64(int B, int n, const double* E_init, const double* E_pair,
                              const double* lognorm, double* out, void* stream);

/* Backward sampling given the forward messages held in `workspace` by the LAST call of
 * svae_lds_estep_f64 with the same (B,T,n) and keep_factor != 0 [natural_sample_backward,
 * /root/reference/svae/lds/cython_lds_inference.pyx:310-355].
 *   eps     (B,T,S,n) standard-normal draws (the reference draws them inside, :333; passing them
 *           in makes the op deterministic and parity-testable)
 *   samples (B,T,S,n)
 */
int svae_lds_sample_f64(int B, int T, int n, int S, unsigned options, const double* eps, double* samples,
                        const void* workspace, size_t ws_bytes, void* stream);

/* E-step + backward sampling in ONE call: the composite the reference's model layer uses,
 *   cython_natural_lds_inference_general(natparam, node_params, num_samples) -> (samples, expected_stats, lognorm)
 *     /root/reference/svae/lds/lds_inference.py:196-202  (= natural_filter_forward_general, natural_smoother_general,
 *     natural_sample_backward: /root/reference/svae/lds/cython_lds_inference.pyx:28-90, 149-210, 310-355),
 * keeping in `workspace` -- keep_vjp != 0 -- what svae_lds_estep_vjp_ex_f64 needs (called with SVAE_OPT_INFER_RECORDS, the
 * same B, T, n, S, inhomog and `options`); keep_vjp = 0: forward values only (no cross-moment record; lean records then
 * also serve per-step / per-sequence pair parameters: the final pass of the SLDS's run_inference, slds_svae.py:289-310).  Arguments as svae_lds_estep_f64 (n <= SVAE_LDS_MAX_N) plus eps, samples (B,T,S,n) as
 * svae_lds_sample_f64; S = 0: no sampling (eps, samples may be NULL).  Same results as svae_lds_estep_f64 with keep = 3
 * followed by svae_lds_sample_f64 -- and in general exactly those two calls.  What the single entry point buys: for
 * homogeneous pair parameters, n <= 10, T >= 2, S <= 2 and batches of more than 2048 sequences (or SVAE_OPT_LEAN_ON)
 * the per-step records that travel through HBM shrink from (4n+3)n + (n+1)(n+2) doubles to n(n+1)/2 + n (+ the cross
 * moments): the forward pass keeps only U_t = chol(P_t)^-T and c_t, every reader rebuilds P_t^-1 = U U' and P_t^-1 J12,
 * the sampler runs inside the smoother's loop, and the first VJP sweep hands the second one symmetric triangle
 * (csrc/lds_lean_estep.hpp, lds_lean_vjp.hpp).  After a lean call svae_lds_sample_f64 cannot follow (the samples were
 * drawn here) and the VJP takes no cotangents of E_init / E_pair.  svae_lds_inference_is_lean: which format a call with
 * these arguments uses (1 = lean; host only, a pure function of its arguments).
 * Return values as svae_lds_estep_f64; -4: bad S / missing eps or samples; -100 + k: the sampler stage returned k. */
int svae_lds_inference_is_lean(int B, int T, int n, int S, int inhomog, int keep_vjp, unsigned options);
int svae_lds_inference_f64(int B, int T, int n, int S, int inhomog, int pair_batched, int keep_vjp, unsigned options,
                           const double* init_J, const double* init_h, const double* init_logZ,
                           const double* J11, const double* J12, const double* J22, const double* logZ_pair,
                           const double* node_J, const double* node_h, const double* node_logZ,
                           const double* eps, double* samples,
                           double* lognorm, double* E_init, double* E_pair,
                           double* E_node_diagxx, double* E_node_x,
                           int32_t* info, void* workspace, size_t ws_bytes, void* stream);

/* Bytes of scratch svae_lds_estep_vjp_f64 needs in addition to the E-step workspace. */
size_t svae_lds_vjp_workspace_bytes(int B, int T, int n);

/* Reverse-mode derivative (vector-Jacobian product) of the E-step [+ sampler] w.r.t. the node
 * potentials, for the LAST svae_lds_estep_f64 call on `workspace` with keep = 3 (and, if g_samples
 * is given, the svae_lds_sample_f64 call that followed).  One entry point for what the reference
 * wires as three autograd primitives (/root/reference/svae/lds/lds_inference.py:26-39):
 *   natural_filter_grad            /root/reference/svae/lds/cython_lds_inference.pyx:92-145
 *   natural_smoother_general_grad  /root/reference/svae/lds/cython_lds_inference.pyx:236-306
 *   natural_sample_backward_grad   /root/reference/svae/lds/cython_lds_inference.pyx:357-409
 *  in : J12 (n,n) natural pair parameter (homogeneous pair parameters only)
 *       g_lognorm (B); g_E_node_diagxx, g_E_node_x (B,T,n) or NULL; g_samples (B,T,S,n) or NULL with
 *       the eps (B,T,S,n) and samples (B,T,S,n) of the sampler call (S <= 16)
 *       (cotangents of E_init / E_pair are taken as zero: the model code never differentiates the
 *        global statistics, /root/reference/svae/svae.py:21)
 *  out: g_node_J (B,T,n), g_node_h (B,T,n)   [g_node_logZ[b,t] = g_lognorm[b]: trivial, left to the caller]
 */
int svae_lds_estep_vjp_f64(int B, int T, int n, int S, const double* J12,
                           const double* g_lognorm, const double* g_E_node_diagxx,
                           const double* g_E_node_x, const double* g_samples,
                           const double* eps, const double* samples,
                           double* g_node_J, double* g_node_h,
                           const void* workspace, size_t ws_bytes,
                           void* vjp_workspace, size_t vjp_ws_bytes, void* stream);

/* The same with per-step pair parameters and cotangents of the remaining statistics -- what the
 * SLDS-SVAE differentiates (/root/reference/svae/models/slds_svae.py:295-300: lds_stats feed
 * get_hmm_vlb; _compute_stats_grad /root/reference/svae/lds/cython_lds_inference.pyx:212-234):
 *   inhomog / pair_batched: J12 is (T-1,n,n) / (B,T-1,n,n) as in svae_lds_estep_f64
 *   g_E_init (B, n*n+n) or NULL: cotangent of [E[x0 x0'] | E[x0]]
 *   g_E_pair (B,T-1,3,n,n) or NULL (inhomog only): cotangent of the per-step pair statistics; needs
 *     the forward outputs E_pair (B,T-1,3,n,n) and E_node_x (B,T,n) of the E-step call
 * All other arguments as svae_lds_estep_vjp_f64 (which is this with inhomog = 0 and NULLs). */
int svae_lds_estep_vjp_ex_f64(int B, int T, int n, int S, int inhomog, int pair_batched, unsigned options,
                              const double* J12, const double* g_lognorm,
                              const double* g_E_node_diagxx, const double* g_E_node_x,
                              const double* g_E_init, const double* g_E_pair,
                              const double* g_samples, const double* eps, const double* samples,
                              const double* E_pair, const double* E_node_x,
                              double* g_node_J, double* g_node_h,
                              const void* workspace, size_t ws_bytes,
                              void* vjp_workspace, size_t vjp_ws_bytes, void* stream);

/* The same with one more output: g_node_J_dense (B,T,n,n) = -2 Pbar_t, the cotangent of a DENSE node potential J_t --
 * the (T,n,n) node potentials of the reference's Python path (natural_condition_on_general,
 * /root/reference/svae/lds/gaussian.py:46-49; lds_inference.py:65-82, 205-218), which the host folds into per-step
 * pair parameters (svae_amd/lds/lds_inference.py:_fold_dense_nodes): J_t enters the recursion only through the pivot
 * block P_t = J_pred,t + J_t (+ J11,t), so its cotangent is the whole of P_t's, of which g_node_J is the diagonal.  Not
 * symmetrised (the caller takes the symmetric part); runs the packed sweeps whatever B (only they write it).  -27: NULL
 * g_node_J_dense; -8 also with SVAE_OPT_INFER_RECORDS on lean records. */
int svae_lds_estep_vjp_dense_f64(int B, int T, int n, int S, int inhomog, int pair_batched, unsigned options,
                                 const double* J12, const double* g_lognorm,
                                 const double* g_E_node_diagxx, const double* g_E_node_x,
                                 const double* g_E_init, const double* g_E_pair,
                                 const double* g_samples, const double* eps, const double* samples,
                                 const double* E_pair, const double* E_node_x,
                                 double* g_node_J, double* g_node_h, double* g_node_J_dense,
                                 const void* workspace, size_t ws_bytes,
                                 void* vjp_workspace, size_t vjp_ws_bytes, void* stream);

/* Batched HMM E-step: log-normaliser and expected statistics of B chains with K <= SVAE_HMM_MAX_K = 64 states
 * (K <= 16: one 16-lane DPP row per sequence, two-ended scaled recursions; 17 <= K <= 64, round 6: one wavefront per
 * sequence, lane = state, scaled recursions with a log-space redo of flagged sequences -- csrc/hmm_estep_wide.hip)
 * [hmm_logZ  /root/reference/svae/hmm/cython_hmm_inference.pyx:93-121 and hmm_logZ_grad :126-166 at
 *  g = 1, i.e. `hmm_estep_slow = vgrad(hmm_logZ)` /root/reference/svae/hmm/hmm_inference.py:65; the
 *  reference's hmm_estep (:21-41) delegates to the un-vendored pyhsmm].
 *  in : init_params (K) log initial potentials; pair_params (K,K) log transition potentials [j][k] =
 *       j -> k, or (B,K,K) if pair_batched; node_params (B,T,K) log node potentials
 *  out: logZ (B); E_init (B,K) = E[z_0]; E_trans (B,K,K) = sum_t E[z_t = j, z_{t+1} = k];
 *       E_states (B,T,K) = E[z_t]
 *  workspace: svae_hmm_workspace_bytes(B,T,K)
 */
size_t svae_hmm_workspace_bytes(int B, int T, int K);
int svae_hmm_estep_f64(int B, int T, int K, int pair_batched,
                       const double* init_params, const double* pair_params,
                       const double* node_params,
                       double* logZ, double* E_init, double* E_trans, double* E_states,
                       void* workspace, size_t ws_bytes, void* stream);

/* HMM step of the SLDS coordinate ascent on the rows `seq_index` lists (B of `rows`; NULL: rows 0..B-1; negative
 * entries = unused slots, which must follow the live ones -- the list svae_slds_sweep_glue_f64 writes):
 *   hmm_meanfield + get_arhmm_local_nodeparams   /root/reference/svae/models/slds_svae.py:108-115, 131-147
 * Node log-potentials: `node_params` (rows,T,K) if given; else built on the fly from the outputs of
 * svae_slds_lds_meanfield_f64 --  node[b,0,k] = <E x_0 x_0', init_J_k> + <E x_0, init_h_k> + cinit_k  (lds_E_init
 * (rows, n*n+n), init_J (K,n,n), init_h (K,n), cinit (K) = the init potential's constants),
 * node[b,t,k] = pair_contr[b,t-1,0,k] + pair_contr[b,t,1,k] + lz_k  (t >= 1; lz (K) = the pair potentials' constants).
 * Outputs as svae_hmm_estep_f64, written at the listed rows only; node_out (rows,T,K) or NULL: the potentials used. */
int svae_slds_hmm_meanfield_f64(int B, int rows, int T, int K, int n,
                                const double* hmm_init, const double* hmm_pair, const double* node_params,
                                const double* pair_contr, const double* lds_E_init, const double* init_J,
                                const double* init_h, const double* cinit, const double* lz,
                                const int32_t* seq_index,
                                double* logZ, double* E_init, double* E_trans, double* E_states,
                                double* node_out, void* workspace, size_t ws_bytes, void* stream);

/* End of one sweep of the SLDS coordinate ascent (optimize_local_meanfield, slds_svae.py:159-175), after the HMM and the
 * fused LDS kernels, for the B listed rows:  lds_vlb = lognorm + <E z_0, cinit> + sum_{t>=1} <E z_t, lz>;
 * vlb_new = hmm_vlb + lds_vlb;  iters += 1;  the row keeps iterating unless |vlb_new - vlb| < tol (:170-172);  vlb = vlb_new.
 * next_index (B) / next_count (1): the rows still iterating, in the order of `seq_index` (a different buffer), padded
 * with -1 up to B: the next sweep may be launched on B slots before the host has read next_count (negative slots are
 * skipped by the three kernels of a sweep).
 * keep_scratch: B int32.  No host arithmetic is needed between two sweeps (the caller reads next_count to size them). */
int svae_slds_sweep_glue_f64(int B, int T, int K, double tol, const int32_t* seq_index,
                             const double* E_states, const double* cinit, const double* lz,
                             const double* lognorm, const double* hmm_vlb, double* lds_vlb, double* vlb,
                             int32_t* iters, int32_t* keep_scratch, int32_t* next_index, int32_t* next_count,
                             void* stream);

/* The two dense contractions at the ends of the ascent, as kernels (they were library GEMMs over materialised outer
 * products / per-step parameter blocks):
 *   initialize_local_meanfield + get_arhmm_local_nodeparams  /root/reference/svae/models/slds_svae.py:203-226, 131-147
 *     HMM node potentials of sweep 0 from ONE sample path x (B,T,n):  node[b,0,k] = x_0' init_J_k x_0 + init_h_k' x_0 +
 *     cinit_k;  node[b,t,k] = x_{t-1}' J11_k x_{t-1} + x_{t-1}' J12_k x_t + x_t' J22_k x_t + lz_k  (t >= 1).  n <= 15.
 *   get_var_lds_local_natparam (the pair part)               /root/reference/svae/models/slds_svae.py:92-103
 *     out_J11/J12/J22 (B,T-1,n,n), out_logZ (B,T-1):  out[b,t] = sum_k E_states[b,t+1,k] P_k.
 * Parameter blocks carry a leading K axis (K <= 16). */
int svae_slds_path_nodeparams_f64(int B, int T, int K, int n, const double* x, const double* init_J,
                                  const double* init_h, const double* cinit, const double* J11, const double* J12,
                                  const double* J22, const double* lz, double* node_out, void* stream);
int svae_slds_mix_pair_natparam_f64(int B, int T, int K, int n, const double* E_states, const double* J11,
                                    const double* J12, const double* J22, const double* lz, double* out_J11,
                                    double* out_J12, double* out_J22, double* out_logZ, void* stream);

/* The two contractions of the SLDS final pass (run_inference, /root/reference/svae/models/slds_svae.py:289-310) over the
 * per-step pair statistics pair_stats (B,T-1,3,n,n) of the last LDS E-step, in ONE pass over them:
 *   get_arhmm_local_nodeparams :131-147   node_out[b,t+1,k] = <pair_stats[b,t], P_k> + lz_k   (rows 1..T-1 of (B,T,K);
 *                                         row 0 -- the init statistics -- is the caller's)
 *   get_global_stats :229-243             sum_{b,t} weights[b,t+1,k] pair_stats[b,t]: `blocks` partial sums
 *                                         gpart (blocks, 8, 3 n^2), to be added in index order (rows k >= K are zero)
 * P (K, 3 n^2) = [J11_k | J12_k | J22_k] flattened, lz (K), weights (B,T,K) = the HMM marginals.  n <= 10, K <= 8, T >= 2. */
int svae_slds_pair_contract_f64(int B, int T, int K, int n, const double* pair_stats, const double* P, const double* lz,
                                const double* weights, double* node_out, double* gpart, int blocks, void* stream);

/* GMM mean-field fixed point + global statistics for one minibatch of T points
 * [local_meanfield, /root/reference/svae/models/gmm.py:62-88; meanfield_fixed_point :90-110;
 *  gaussian_meanfield :112-117; label_meanfield :119-124].
 *
 *  in : label_global (K)             = dirichlet.expectedstats(dirichlet natparam)
 *       gaussian_globals (K,N+2,N+2) = niw.expectedstats(niw natparams), dense-packed
 *       node_J (T,N) diagonal -1/2 precision, node_h (T,N)      (the recognition potentials)
 *       label_init (T,K)  initial responsibilities (reference: normalize(rand(T,K)), gmm.py:126-128)
 *       tol, max_iter     (reference: 1e-3, 100); the stop rule is on the batch-total KL
 *  out: label_stats (T,K) responsibilities at the fixed point (after the final extra pass, :74-77)
 *       label_fixed (T,K) or NULL: the fixed point itself, i.e. what that final pass started from
 *         (:71; callers that keep the final pass on an autograd tape re-run it from here)
 *       gaussian_stats (T,N+2,N+2) dense-packed E[t(x_t)]
 *       label_natparam (T,K), gaussian_natparam (T,N+2,N+2)
 *       dirichlet_stats (K), niw_stats (K,N+2,N+2)      (sums over points, :80-81)
 *       kl (1) = label_kl + gaussian_kl (:86),  iters (1) number of fixed-point iterations run
 *       assign (T) int32 argmax_k label_stats
 *  Limits: N <= 8, K <= 64.  Single workgroup (the minibatch total KL is a block reduction).
 */
int svae_gmm_meanfield_f64(int T, int N, int K,
                           const double* label_global, const double* gaussian_globals,
                           const double* node_J, const double* node_h, const double* label_init,
                           double tol, int max_iter,
                           double* label_stats, double* label_fixed, double* gaussian_stats,
                           double* label_natparam, double* gaussian_natparam,
                           double* dirichlet_stats, double* niw_stats,
                           double* kl, int32_t* iters, int32_t* assign,
                           int32_t* info, void* stream);

/* The same fixed point spread over MANY workgroups -- and, through the caller's all-reduce, over many GPUs --
 * with the reference's stopping rule on the batch-total KL (/root/reference/svae/models/gmm.py:104-105)
 * kept exact: one launch per sweep, each ending in a fixed-order reduction of its workgroups' KL partials
 * into kl_hist[sweep] (first max_iter+1 doubles of the workspace; svae_gmm_mw_kl_hist); every launch first
 * scans kl_hist for an earlier sweep that met |kl_j - kl_{j-1}| < tol and is a no-op if there is one, so the
 * host enqueues max_iter sweeps blindly, with no synchronisation.  Multi-GPU (points sharded over ranks):
 * all-reduce kl_hist[sweep] in place after each sweep launch -- every rank then takes the same decision.
 *   svae_gmm_mw_begin     zeroes the state (once per fixed point)
 *   svae_gmm_mw_step_f64  phase 0: sweep number `sweep`; phase 1: the final pass of gmm.py:74-86 (writes
 *                         every per-point output, kl = this rank's total, iters); phase 2: the global
 *                         statistics (dirichlet_stats, niw_stats of this rank's points).
 * Arguments as svae_gmm_meanfield_f64.  Results agree with the single-workgroup kernel up to the summation
 * order of the KL total (same per-point arithmetic: identical labels unless a sweep sits on the tolerance). */
size_t svae_gmm_mw_workspace_bytes(int T, int N, int K, int max_iter);
int svae_gmm_mw_begin(int T, int N, int K, int max_iter, void* workspace, size_t ws_bytes, void* stream);
int svae_gmm_mw_step_f64(int phase, int sweep, int T, int N, int K,
                         const double* label_global, const double* gaussian_globals,
                         const double* node_J, const double* node_h,
                         const double* label_init, double tol, int max_iter,
                         double* label_stats, double* label_fixed, double* gaussian_stats,
                         double* label_natparam, double* gaussian_natparam,
                         double* dirichlet_stats, double* niw_stats,
                         double* kl, int32_t* iters, int32_t* assign, int32_t* info,
                         void* workspace, size_t ws_bytes, void* stream);
double* svae_gmm_mw_kl_hist(void* workspace);
/* Single GPU: the same sweeps, final pass and statistics as svae_gmm_mw_begin + max_iter x phase 0 + phase 1 + phase 2,
 * in ONE plain launch (+ the statistics launch): per sweep the workgroups exchange their KL partials as data-tagged
 * 8-byte agent-scope words (no grid barrier, no atomics read-modify-write, no cooperative launch), every workgroup sums
 * them in index order and takes the same stopping decision; with at most one point per thread and K <= 16 the
 * responsibilities stay in registers for the whole fixed point.  Removes the 16-21 us per-sweep launch cost that bounds
 * the fixed point below ~100 k points.  Same results as the per-sweep calls, bit for bit.  Every spin is bounded: a
 * workgroup that never sees a partner's partial sets *info = -77.  Returns -50 only when the device cannot be queried. */
int svae_gmm_mw_fixed_point_f64(int T, int N, int K,
                                const double* label_global, const double* gaussian_globals,
                                const double* node_J, const double* node_h,
                                const double* label_init, double tol, int max_iter,
                                double* label_stats, double* label_fixed, double* gaussian_stats,
                                double* label_natparam, double* gaussian_natparam,
                                double* dirichlet_stats, double* niw_stats,
                                double* kl, int32_t* iters, int32_t* assign, int32_t* info,
                                void* workspace, size_t ws_bytes, void* stream);


/* ---- GMM-SVAE local step, differentiable tail (csrc/gmm_train.hip) -------------------------------------------
 * Replaces /root/reference/svae/distributions/gaussian.py:27-33 (`natural_sample`) and autograd's reverse pass
 * through /root/reference/svae/models/gmm.py:74-86 + gmm.py:12-16 (`run_inference`: samples and local_kl are what
 * svae.py:21-30 differentiates w.r.t. the recognition network's node potentials).
 *   svae_gmm_sample_f64     samples[t,s,:] = J_t^-1 h_t + chol(J_t)^-T eps[t,s,:], (J_t, h_t) unpacked from the
 *                           dense (N+2)x(N+2) gaussian_natparam[t] the fixed-point kernels return.
 *   svae_gmm_local_vjp_f64  cotangents of the node potentials (g_node_J, g_node_h: (T,N)) given g_kl (device scalar,
 *                           cotangent of the final pass's local KL; NULL = 0) and g_samples ((T,S,N); NULL = none,
 *                           then eps may be NULL).  gaussian_natparam / label_natparam: the final pass's outputs.
 * One point per lane, N <= 8, K <= 64; device pointers, asynchronous on `stream`.  0 / -k (argument k).
 *   svae_gmm_global_step_f64  the global side of a step in ONE launch: label_global (K) = dirichlet.expectedstats
 *                           (dirichlet.py:5-7), gaussian_globals (K,N+2,N+2) = niw.expectedstats (niw.py:15-25) -- the two
 *                           potentials svae_gmm_meanfield_f64 / svae_gmm_mw_* take -- and, if `kl` (TWO doubles, ABI 12) is
 *                           given, the prior KL of gmm.py:54-58: kl[0] as the code spells it (full contraction;
 *                           dirichlet.logZ, niw.logZ), kl[1] as the reference AS SHIPPED evaluates it (util.py:39 `flat`
 *                           after util.py:166 rebinds `flatten`: the contraction keeps its first term only).  *info is
 *                           raised to 1 on a non-positive-definite NIW scale matrix. */
int svae_gmm_global_step_f64(int K, int N, const double* dirichlet_natparam, const double* niw_natparam,
                             const double* prior_dirichlet, const double* prior_niw,
                             double* label_global, double* gaussian_globals, double* kl, int32_t* info, void* stream);
int svae_gmm_sample_f64(int T, int N, int S, const double* gaussian_natparam, const double* eps,
                        double* samples, void* stream);
int svae_gmm_local_vjp_f64(int T, int N, int K, int S, const double* label_global,
                           const double* gaussian_globals, const double* node_J, const double* node_h,
                           const double* gaussian_natparam, const double* label_natparam,
                           const double* g_kl, const double* eps, const double* g_samples,
                           double* g_node_J, double* g_node_h, void* stream);


/* ---- one-shot all-reduce of a small buffer over IPC-mapped mailboxes (csrc/ipc_allreduce.hip) -----------------------
 * The exchange step (/root/reference/svae/svae.py:33-34: the batch-summed statistics) as ONE kernel: every rank stores
 * its n doubles, as tagged 8-byte words, into slot `rank` of EVERY rank's mailbox (fine-grained device memory of
 * svae_ipc_mailbox_bytes(capacity, world) bytes, zero-initialised once, IPC-mapped into every peer: mailboxes[q] = rank
 * q's mailbox as mapped into this process), then sums the words that arrived in its own mailbox in rank order -> out
 * (may alias in).  `capacity` (doubles) fixes the mailbox layout for its lifetime -- the same on every rank and in every
 * call; a call may reduce any n <= capacity, and consecutive calls may differ in n (ABI 12).  epoch: 1, 2, 3, .. per
 * call, the same on every rank.  Bit-identical results on every rank.  spin_limit: polls per word before a peer counts
 * as absent (0 = 2^28, minutes); an element whose words never arrive comes out as NaN and *info = -78 -- a timeout never
 * passes for a sum.  world <= 16.  Opt-in alternative to the RCCL all-reduce (svae_amd/ipc.py).
 * Returns 0 / -k (argument k). */
size_t svae_ipc_mailbox_bytes(int capacity, int world);
int svae_ipc_allreduce_f64(int n, int capacity, int rank, int world, unsigned epoch, unsigned spin_limit,
                           const double* in, double* out, void* const* mailboxes, int32_t* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVAE_HIP_H */
