// dpp.hpp -- CDNA4 (gfx950) cross-lane primitives used by the SVAE message-passing kernels.
//
// Layout convention ("row tile"): a wavefront is 4 DPP rows of 16 lanes.  One row owns one
// sequence (or one data point group); lane c of the row holds COLUMN c of every small matrix, one
// VGPR pair per matrix row.  `bcast<K>(x)` returns lane K of the caller's row to all 16 lanes of
// that row in one DP-ALU DPP move (v_mov_b64_dpp row_newbcast:K) -- no LDS, no SGPR round trip, no
// cross-row traffic, so the 4 rows of a wave run 4 independent problems in lock-step.
#pragma once
#include <hip/hip_runtime.h>

namespace svae {

// llvm.amdgcn.update.dpp is type-generic in LLVM but clang's builtin is int-only in ROCm 7.2;
// binding the intrinsic by its IR name gives the f64 form (lowers to v_mov_b64_dpp on gfx950).
extern "C" __device__ double __svae_update_dpp_f64(double, double, int, int, int, bool)
    __asm("llvm.amdgcn.update.dpp.f64");

constexpr int DPP_ROW_NEWBCAST0 = 0x150;  // row_newbcast:0 .. row_newbcast:15 (gfx90a+)

template <int K>
__device__ __forceinline__ double bcast(double x) {
  static_assert(K >= 0 && K < 16, "row_newbcast lane out of range");
  return __svae_update_dpp_f64(0.0, x, DPP_ROW_NEWBCAST0 + K, 0xf, 0xf, true);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Reciprocal to full fp64 accuracy: v_rcp_f64 seed + two Newton steps (4 dependent FMAs).
__device__ __forceinline__ double rcp_nr(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// Sum over the 16 lanes of a DPP row (result valid in every lane of the row).
__device__ __forceinline__ double row_sum16(double x) {
  x += __shfl_xor(x, 1, 16);
  x += __shfl_xor(x, 2, 16);
  x += __shfl_xor(x, 4, 16);
  x += __shfl_xor(x, 8, 16);
  return x;
}

}  // namespace svae
