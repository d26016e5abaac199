// lds_args.hpp -- launch arguments shared by the C-ABI dispatcher and the per-n kernel units.
#pragma once
#include <stdint.h>

namespace svae {

// Workspace layout per (sequence, step), in doubles: N rows of H = [P^-1 J12 | c] (N+1 entries, row
// stride padded to even => 16-byte aligned rows), then N rows of P^-1 (stride padded to even).
constexpr int ws_h_stride(int n) { return (n + 1) + ((n + 1) & 1); }
constexpr int ws_p_stride(int n) { return n + (n & 1); }
constexpr int ws_step_doubles(int n) { return n * (ws_h_stride(n) + ws_p_stride(n)); }
// Per sequence, ahead of the T step blocks: a constant page [e_n (H stride) | zeros (P stride)] that
// the lanes >= n read instead of a hand-off row (keeps every hot-loop load unconditional: a
// conditional load makes hipcc wait for it at the join, which defeats the prefetch).
constexpr int ws_zpage_doubles(int n) { return ws_h_stride(n) + ws_p_stride(n); }
constexpr long ws_seq_doubles(int n, int T) { return ws_zpage_doubles(n) + (long)T * ws_step_doubles(n); }

// ---- workspace geometry of the two-ended kernel (lds_estep_twoend.hpp) -------------------------------
// per chain (2 per sequence): a constant page, then one record per local step 0 .. T/2
constexpr int te_row_doubles(int n) { return 2 * n + 2; }           // [P^-1 row (n) | X row (n) | c_i | pad]
constexpr int te_step_doubles(int n) { return n * te_row_doubles(n); }
constexpr int te_lean_step_doubles(int n) { return (n * (n + 1) / 2 + n + 3) & ~1; }   // lean record: [tril P^-1 | c | 0 | trash]
constexpr int te_page_doubles(int n) { return 2 * (n + 2) + 2; }    // [e_n (n+2) | zeros (n+2) | trash (2)]
constexpr int te_elims(int T) { return T / 2; }                     // eliminations per chain
constexpr long te_chain_doubles(int n, int T) {
  return te_page_doubles(n) + (long)(te_elims(T) + 1) * te_step_doubles(n);
}
constexpr long te_seq_doubles(int n, int T) { return 2 * te_chain_doubles(n, T); }
constexpr int TE_MAX_N = 10;
constexpr int TE_MIN_T = 4;
constexpr int TE_S4_LDS_BYTES = 14 * 1024;   // static LDS of an S4 workgroup (transposition tiles, exchange buffer), rounded up
constexpr int TE_S4_MAX_B = 512;     // two-ended kernel: second wavefront per sequence in the smoother phase up to this batch
constexpr int TE_RPC_MIN_B = 1025;   // two-ended kernel: two sequences per wavefront (row-per-chain layout) from this batch
                                     // (measured T = 200, n = 10: 1024 sequences 0.200 vs 0.237 ms, 1536: 0.321 vs 0.272 ms)
// MIX launches of the two-ended kernel (K parameter sets mixed per step, svae_slds_lds_meanfield_f64): LDS
// tables of 16-byte entries (two states each), see lds_estep_twoend.hpp
constexpr int te_mix_nxl(int n) { return 15 - n; }                 // right-hand-side lanes n..14
constexpr int te_mix_ent_j(int n) { return 4 * te_mix_nxl(n) + 2 * n + 1; }   // J12 table: rhs lanes | Schur-operand lanes | zero
constexpr int te_mix_ent_c(int n) { return 4 * n + 1; }
constexpr long te_mix_lds_bytes(int n, int K) {
  const int kp2 = (K + 1) / 2, j = (n + 1) / 2, nc2 = (n + 1) / 2;
  return 16L * (kp2 * (n * te_mix_ent_j(n) + 2 * j * te_mix_ent_c(n)) + 4 * j * 2 * nc2 * 16);
}
constexpr int TE_MIX_MAX_K = 16;
constexpr long TE_MIX_MAX_LDS = 160 * 1024;

struct LdsArgs {
  int B, T;
  const double* __restrict__ init_J;
  const double* __restrict__ init_h;
  const double* __restrict__ init_logZ;
  const double* __restrict__ J11;
  const double* __restrict__ J12;
  const double* __restrict__ J22;
  const double* __restrict__ logZ_pair;
  const double* __restrict__ node_J;
  const double* __restrict__ node_h;
  const double* __restrict__ node_logZ;
  double* __restrict__ lognorm;
  double* __restrict__ E_init;
  double* __restrict__ E_pair;
  double* __restrict__ E_node_diagxx;
  double* __restrict__ E_node_x;
  int32_t* __restrict__ info;
  double* __restrict__ ws;
  double* __restrict__ ws2;   // factor region for the sampler (nullptr: not kept)
  double* __restrict__ ws3;   // cross-moment region for the VJP: W~_t, (n+1) rows x ws_h_stride per step
  long pair_seq_stride;  // doubles between consecutive sequences' pair blocks (0 = shared)
  // filter-only launches (svae_lds_filter_f64): the forward messages in the reference's scaling
  // (natural parameters, cython_lds_inference.pyx:84-85), each (B,T,n,n) / (B,T,n) or nullptr
  double* __restrict__ msg_Jp;
  double* __restrict__ msg_hp;
  double* __restrict__ msg_Jf;
  double* __restrict__ msg_hf;
  // MIX launches (svae_slds_lds_meanfield_f64): init_J/init_h/J11/J12/J22 carry a leading K axis
  const double* __restrict__ mix_w;      // (B,T,K) weights E[z_t = k]
  double* __restrict__ mix_out;          // (B,T,2,K) contractions with the K pair-parameter sets
  const int32_t* __restrict__ seq_index; // (B) or nullptr: slot i of the launch works on row seq_index[i]
  int mix_K;
  // S4 launches of the two-ended kernel: the last `lds_keep` hand-off records of each chain (the ones the smoother
  // reads first) stay in LDS instead of travelling through HBM (0: all records through HBM)
  int lds_keep;
  // tiled path (n > 15): smoothed covariances Sigma_t, compact (B,T,n,n), written by the backward half (nullptr: not kept)
  double* __restrict__ sig_out;
  // tiled path (n > 15): 0 = whole E-step, 1 = forward half only, 2 = backward half only (SVAE_OPT_TILE_FORWARD / _BACKWARD)
  int tile_half;
};

struct SampleArgs {
  int B, T, S;
  int prod_max_b;                     // largest batch that runs with producer wavefronts (svae_lds_set_prod_max_b)
  const double* __restrict__ eps;     // (B, T, S, n)
  double* __restrict__ samples;       // (B, T, S, n)
  const double* __restrict__ ws;      // main region (G~' rows, c, P^-1)
  const double* __restrict__ ws2;     // factor region
};

// reverse-mode sweeps (lds_vjp_kernel.hpp)
struct VjpArgs {
  int B, T, S;
  int prod_max_b;                        // largest batch that runs with producer / helper wavefronts
  const double* __restrict__ J12;        // natural pair parameter: (n,n), (T-1,n,n) or (B,T-1,n,n)
  long pair_t_stride;                    // doubles between consecutive steps' J12 (0 = homogeneous)
  long pair_seq_stride;                  // doubles between consecutive sequences' J12 blocks (0 = shared)
  const double* __restrict__ g_E_init;   // (B, n*n+n) cotangent of (E[x_0 x_0'], E[x_0]) or nullptr
  const double* __restrict__ g_E_pair;   // (B,T-1,3,n,n) cotangent of the per-step pair statistics or nullptr
  const double* __restrict__ E_pair;     // (B,T-1,3,n,n) forward output (with g_E_pair: supplies S~_{t+1})
  const double* __restrict__ E_node_x;   // (B,T,n)       forward output (with g_E_pair)
  const double* __restrict__ g_lognorm;  // (B)
  const double* __restrict__ g_diagxx;   // (B,T,n) or nullptr
  const double* __restrict__ g_x;        // (B,T,n) or nullptr
  const double* __restrict__ g_samples;  // (B,T,S,n) or nullptr
  const double* __restrict__ eps;        // (B,T,S,n)   (with g_samples)
  const double* __restrict__ samples;    // (B,T,S,n)   (with g_samples)
  double* __restrict__ g_node_J;         // (B,T,n)
  double* __restrict__ g_node_h;         // (B,T,n)
  const double* __restrict__ ws;         // E-step main region
  const double* __restrict__ ws2;        // factor region
  const double* __restrict__ ws3;        // cross-moment region
  double* __restrict__ adj;              // VJP scratch: vjp_step_doubles(n) per (b,t)
  double* __restrict__ g_P;              // (B,T,n,n) or nullptr: -2 Pbar_t = the cotangent of a DENSE node potential J_t (packed sweep 2 only)
};
// VJP scratch per (b,t), written by sweep 1 and read by sweep 2:
//   [0, n HS)        G^: the smoother share (two-role launches) or the total -- n rows x ws_h_stride
//   vjp_vec_off      the sampler share of G^ = sum_s xhat_s [x_{t+1,s}' | 1] as its factors (two-role launches, which
//                    take at most VJP_SPLIT_MAX_S samples): per sample [xhat_s (n) | x_{t+1,s} (n)]
//   vjp_pbp_off      -P^-1 Pinvbar P^-1: symmetric, lower triangle by rows (entry (i, c <= i) at i (i + 1) / 2 + c)
//   vjp_pex_off      Pbar(direct) of the sampler's noise factor (not symmetric): n rows x ws_p_stride
constexpr int VJP_SPLIT_MAX_S = 4;
constexpr int vjp_tri_doubles(int n) { return (n * (n + 1) / 2 + 1) & ~1; }
constexpr int vjp_vec_off(int n) { return n * ws_h_stride(n); }
constexpr int vjp_pbp_off(int n) { return vjp_vec_off(n) + 2 * VJP_SPLIT_MAX_S * n; }
constexpr int vjp_pex_off(int n) { return vjp_pbp_off(n) + vjp_tri_doubles(n); }
constexpr int vjp_step_doubles(int n) { return vjp_pex_off(n) + n * ws_p_stride(n); }

// ---- LEAN records of the large-batch training path (lds_lean_estep.hpp / lds_lean_vjp.hpp; round 6) -----------------
// svae_lds_inference_f64 with homogeneous pair parameters, n <= LEAN_MAX_N, 1 <= T, S <= LEAN_MAX_S and a batch the packed
// kernels serve: per (sequence, step) the forward pass keeps ONLY
//   [ U = chol(P_t)^-T = L^-T D^-1/2, upper triangle packed by rows (row k at lean_row_off(k), entries c = k .. n-1)
//   | c_t = P_t^-1 h_filt,t (n) | trash / pad ]                                              = lean_rec_doubles(n)
// (66 doubles at n = 10 against 330 of the full hand-off + factor region).  Every reader rebuilds P^-1 = U U' (n(n+1)/2
// DPP multiply-adds) and P^-1 J12 (n^2) from it; the sampler's noise is U eps.  The cross-moment region (W~_t) is
// unchanged.  Sweep 1 leaves [G^ (n rows x ws_h_stride) | lower triangle of -P^-1 Pinvbar P^-1 + sym(Pbar(direct))]:
// only the symmetric part of the sampler's direct share reaches the gradients (every map it goes through is a
// congruence, and the outputs take the diagonal), so the two shares travel as ONE triangle.
constexpr int LEAN_MAX_N = 10;
constexpr int LEAN_MAX_S = 2;
constexpr int LEAN_MIN_B = 1025;        // default dispatch: above the batches the producer-wavefront kernels serve (measured T = 200, n = 10: 1024 sequences 1.36 ms with producers vs 1.76 lean; 1100: 1.93 full vs 1.85 lean; 2048: 2.52 vs 1.93; 4096: 4.01 vs 2.61)
constexpr int lean_tri(int n) { return n * (n + 1) / 2; }
constexpr int lean_row_off(int n, int k) { return k * n - k * (k - 1) / 2; }
constexpr int lean_trash(int n) { return lean_tri(n) + n; }
constexpr int lean_rec_doubles(int n) { return (lean_tri(n) + n + 2) & ~1; }
constexpr int lean_adj_tri_off(int n) { return n * ws_h_stride(n); }
constexpr int lean_adj_doubles(int n) { return lean_adj_tri_off(n) + vjp_tri_doubles(n); }
struct LeanSample {
  int S;                              // 0: no sampling
  const double* __restrict__ eps;     // (B, T, S, n)
  double* __restrict__ samples;       // (B, T, S, n)
};

}  // namespace svae
