// lds_vjp_n.hip -- one translation unit per latent dimension (-DSVAE_N=<n>) for the VJP sweeps.
// Register-heavy kernels: every DPP statement is self-fenced here (see dpp.hpp).
#define SVAE_DPP_ALWAYS_FENCED 1
#include "lds_vjp_kernel.hpp"

#ifndef SVAE_N
#error "compile with -DSVAE_N=<latent dim>"
#endif
#define SVAE_CAT_(a, b) a##b
#define SVAE_CAT(a, b) SVAE_CAT_(a, b)

extern "C" int SVAE_CAT(svae_lds_vjp_n, SVAE_N)(const svae::VjpArgs* a, void* stream) {
  return svae::launch_vjp<SVAE_N>(*a, (hipStream_t)stream);
}
