// lds_estep_n.hip -- one translation unit per latent dimension (compiled with -DSVAE_N=<n>), so
// that `make -j` builds the 15 specialisations of the E-step kernel in parallel.
#if defined(SVAE_N) && SVAE_N > 13
// n = 14, 15: the tiles exceed 256 VGPRs and hipcc moves values through AGPRs (VALU writes the DPP
// hazard bookkeeping in dpp.hpp cannot see): make every DPP statement self-fenced (slower, safe).
#define SVAE_DPP_ALWAYS_FENCED 1
#endif
#include "lds_estep_kernel.hpp"
#include "lds_estep_split.hpp"
#include "lds_estep_twoend.hpp"
#include "lds_estep_twoend_rpc.hpp"
#include "lds_estep_twoend_rpcmix.hpp"
#include "lds_filter_1r.hpp"
#include "lds_lean_estep.hpp"

#ifndef SVAE_N
#error "compile with -DSVAE_N=<latent dim>"
#endif
#define SVAE_CAT_(a, b) a##b
#define SVAE_CAT(a, b) SVAE_CAT_(a, b)

extern "C" int SVAE_CAT(svae_lds_launch_n, SVAE_N)(const svae::LdsArgs* a, int inhomog, void* stream) {
  return svae::launch_estep<SVAE_N>(*a, inhomog != 0, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_sample_n, SVAE_N)(const svae::SampleArgs* a, void* stream) {
  return svae::launch_sample<SVAE_N>(*a, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_launch_split_n, SVAE_N)(const svae::LdsArgs* a, int inhomog, void* stream) {
  return svae::launch_estep_split<SVAE_N>(*a, inhomog != 0, (hipStream_t)stream);
}

// layout: 0 = by batch size (one sequence per wavefront below TE_RPC_MIN_B, two per wavefront from there), 1 = one sequence
// per wavefront, 2 = two per wavefront (row-per-chain kernel; homogeneous lean launches without the cross-moment hand-off)
extern "C" int SVAE_CAT(svae_lds_launch_twoend_n, SVAE_N)(const svae::LdsArgs* a, int inhomog, int lean, int layout,
                                                          void* stream) {
  const bool rpc_ok = !inhomog && lean && !a->ws3 && a->T >= svae::TE_MIN_T;
  if (rpc_ok && (layout == 2 || (layout == 0 && a->B >= svae::TE_RPC_MIN_B)))
    return svae::launch_estep_twoend_rpc<SVAE_N>(*a, (hipStream_t)stream);
  return svae::launch_estep_twoend<SVAE_N>(*a, inhomog != 0, lean != 0, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_launch_twoend_mix_n, SVAE_N)(const svae::LdsArgs* a, void* stream) {
  return svae::launch_estep_twoend_mix<SVAE_N>(*a, (hipStream_t)stream);
}

// the SLDS mean-field step in the row-per-chain layout with producer wavefronts (refprod != 0: reference producers)
extern "C" int SVAE_CAT(svae_lds_launch_slds_rpc_n, SVAE_N)(const svae::LdsArgs* a, int refprod, int seq_ok, void* stream) {
  return svae::launch_slds_meanfield_rpc<SVAE_N>(*a, refprod, seq_ok, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_launch_filter_n, SVAE_N)(const svae::LdsArgs* a, int inhomog, void* stream) {
  return svae::launch_filter<SVAE_N>(*a, inhomog != 0, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_launch_filter_1r_n, SVAE_N)(const svae::LdsArgs* a, int inhomog, void* stream) {
  return svae::launch_filter_1r<SVAE_N>(*a, inhomog != 0, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_launch_forward_pair_n, SVAE_N)(const svae::LdsArgs* f, const svae::LdsArgs* e, int inhomog,
                                                                void* stream) {
  return svae::launch_forward_pair<SVAE_N>(*f, *e, inhomog != 0, (hipStream_t)stream);
}

extern "C" int SVAE_CAT(svae_lds_launch_filter_split_n, SVAE_N)(const svae::LdsArgs* a, int inhomog, void* stream) {
  return svae::launch_filter_split<SVAE_N>(*a, inhomog != 0, (hipStream_t)stream);
}

// E-step + sampler in one launch on lean records (lds_lean_estep.hpp): homogeneous pair parameters, n <= LEAN_MAX_N
extern "C" int SVAE_CAT(svae_lds_infer_lean_n, SVAE_N)(const svae::LdsArgs* a, const svae::LeanSample* ls, int inhomog, void* stream) {
  return svae::launch_infer_lean<SVAE_N>(*a, *ls, inhomog != 0, (hipStream_t)stream);
}
