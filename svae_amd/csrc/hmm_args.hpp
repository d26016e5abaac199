// hmm_args.hpp -- launch arguments shared by the HMM kernel units (hmm_estep.hip: K <= 16; hmm_estep_wide.hip: K <= 64).
#pragma once
#include <stdint.h>

namespace svae {

struct HmmArgs {
  int B, T, K;
  long pair_stride;                       // doubles between sequences' pair params (0 = shared)
  const double* __restrict__ init_params; // (K)      log pi_0 (unnormalised ok)
  const double* __restrict__ pair_params; // (K,K) or (B,K,K)   log P[j][k]  (j -> k)
  const double* __restrict__ node_params; // (B,T,K)  log-likelihood potentials
  double* __restrict__ logZ;              // (B)
  double* __restrict__ E_init;            // (B,K)
  double* __restrict__ E_trans;           // (B,K,K)
  double* __restrict__ E_states;          // (B,T,K)
  double* __restrict__ ws;                // (B,T,HMM_WS)
  // indexed launches (SLDS coordinate ascent): slot i of the launch works on row seq_index[i] of every array
  const int32_t* __restrict__ seq_index;  // (B) or nullptr
  // FUSED node potentials (get_arhmm_local_nodeparams, slds_svae.py:131-147, from the fused LDS mean-field kernel's
  // outputs): node[b,0,k] = <E x0 x0', J_k> + <E x0, h_k> + cinit_k;  node[b,t,k] = pc[b,t-1,0,k] + pc[b,t,1,k] + lz_k
  int n;                                  // latent dimension of the LDS
  const double* __restrict__ pair_contr;  // (rows,T,2,K)
  const double* __restrict__ lds_E_init;  // (rows, n*n+n)
  const double* __restrict__ init_J;      // (K,n,n)
  const double* __restrict__ init_h;      // (K,n)
  const double* __restrict__ cinit;       // (K)
  const double* __restrict__ lz;          // (K)
  double* __restrict__ node_out;          // (rows,T,K) or nullptr: the node potentials used
  int redo_only;                          // hmm_estep_kernel behind hmm_estep2_kernel: only wavefronts with a flagged sequence run
};
// wide kernel (17 <= K <= 64, one wavefront per sequence): workspace record per (sequence, step) =
// [alpha^_t or log alpha_t (KP) | normaliser c_t | REDO flag of the sequence (first record only)]
constexpr int HMM_WIDE_MAX_K = 64;
constexpr int hmm_wide_kp(int K) { return K <= 32 ? 32 : 64; }
constexpr int hmm_wide_rec(int KP) { return KP + 2; }
constexpr double HMM_WIDE_TINY = 1e-200;

}  // namespace svae
