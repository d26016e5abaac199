// lds_vjp_n.hip -- one translation unit per latent dimension (-DSVAE_N=<n>) for the VJP sweeps.
// DPP hazards are handled per product stage (one fence on the DPP-read operand array, dpp.hpp /
// lds_vjp_kernel.hpp) and checked on the generated ISA for every n by `make audit`;
// -DSVAE_DPP_ALWAYS_FENCED=1 falls back to a self-fenced build (every DPP statement carries its own
// wait states, about twice the instructions).
#ifndef SVAE_DPP_ALWAYS_FENCED
#define SVAE_DPP_ALWAYS_FENCED 0
#endif
#include "lds_vjp_kernel.hpp"
#include "lds_lean_vjp.hpp"

#ifndef SVAE_N
#error "compile with -DSVAE_N=<latent dim>"
#endif
#define SVAE_CAT_(a, b) a##b
#define SVAE_CAT(a, b) SVAE_CAT_(a, b)

extern "C" int SVAE_CAT(svae_lds_vjp_n, SVAE_N)(const svae::VjpArgs* a, void* stream) {
  return svae::launch_vjp<SVAE_N>(*a, (hipStream_t)stream);
}

// the two sweeps on the lean records of svae_lds_inference_f64 (lds_lean_vjp.hpp)
extern "C" int SVAE_CAT(svae_lds_vjp_lean_n, SVAE_N)(const svae::VjpArgs* a, void* stream) {
  return svae::launch_vjp_lean<SVAE_N>(*a, (hipStream_t)stream);
}
