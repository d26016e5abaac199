// dpp.hpp -- CDNA4 (gfx950) cross-lane primitives used by the SVAE message-passing kernels.
//
// Layout convention ("row tile"): a wavefront is 4 DPP rows of 16 lanes.  One row owns one
// sequence; lane c of the row holds COLUMN c of every small matrix, one VGPR pair per matrix row.
// Every small dense product is then  acc[c] += M[i][k] * B[k][c]  with M[i][k] = "lane k of my row,
// register i": on gfx90a+ that broadcast is a DPP operand modifier (row_newbcast:k) that the DP-ALU
// accepts directly on v_fmac_f64, i.e. ONE instruction per multiply-accumulate, no LDS, no SGPR
// round trip, no cross-row traffic; the 4 rows of a wave run 4 independent problems in lock-step.
//
// hipcc (ROCm 7.2) cannot emit that form: clang's update_dpp builtin is 32-bit only, and LLVM's
// GCNDPPCombine does not fold a v_mov_b64_dpp into v_fmac_f64 (tied accumulator), which leaves
// mov+fma pairs plus an s_nop per pair.  So the multiply-accumulate is inline asm, and the two
// hazards the compiler would otherwise pad are handled here by construction:
//   (H1) VALU write of a VGPR -> DPP read of it needs 2 wait states;
//   (H2) VALU write of EXEC (v_cmpx) -> DPP op needs 5 wait states (gfx9 hipcc emits v_cmp +
//        s_and_saveexec, never v_cmpx; tools/audit_dpp_hazards.py checks the .s for both).
// All DPP asm statements are `volatile`, so they keep their source order relative to each other;
// `dpp_fence` (an s_nop 1 that "rewrites" its operands) separates compiler-produced values from
// their first DPP read, and the kernels order the statements so that an asm-produced register is
// never DPP-read within the next two instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace svae {

#ifndef SVAE_FUSED_DPP
#define SVAE_FUSED_DPP 1   // 0: compiler-only path (v_mov_b64_dpp + v_fma_f64), for A/B and debugging
#endif

// llvm.amdgcn.update.dpp is type-generic in LLVM but clang's builtin is int-only in ROCm 7.2;
// binding the intrinsic by its IR name gives the f64 form (lowers to v_mov_b64_dpp on gfx950).
extern "C" __device__ double __svae_update_dpp_f64(double, double, int, int, int, bool)
    __asm("llvm.amdgcn.update.dpp.f64");

constexpr int DPP_ROW_NEWBCAST0 = 0x150;  // row_newbcast:0 .. row_newbcast:15 (gfx90a+)

// lane K of the caller's 16-lane row, broadcast to the row (compiler-scheduled; hazards padded
// by hipcc).  Only for values produced by ordinary (non-asm) code.
template <int K>
__device__ __forceinline__ double bcast(double x) {
  static_assert(K >= 0 && K < 16, "row_newbcast lane out of range");
  return __svae_update_dpp_f64(0.0, x, DPP_ROW_NEWBCAST0 + K, 0xf, 0xf, true);
}

// acc (+/-)= bcast_K(src) * b   in one DP-ALU DPP instruction.  FENCED: the statement carries its
// own two wait states (src may have been written by the immediately preceding instruction).
#ifndef SVAE_DPP_ALWAYS_FENCED
#define SVAE_DPP_ALWAYS_FENCED 0   // 1: every DPP statement carries its own wait states (used where
#endif                             // register pressure makes the compiler shuffle VGPR<->AGPR)
template <int K, bool NEG = false, bool FENCED_ = false>
__device__ __forceinline__ void mac_bc(double& acc, double src, double b) {
  static_assert(K >= 0 && K < 16, "row_newbcast lane out of range");
  constexpr bool FENCED = FENCED_ || SVAE_DPP_ALWAYS_FENCED;
#if SVAE_FUSED_DPP
  if constexpr (NEG && FENCED)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(b), "n"(K));
  else if constexpr (NEG)
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(b), "n"(K));
  else if constexpr (FENCED)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(b), "n"(K));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(b), "n"(K));
#else
  acc = __builtin_fma(bcast<K>(src), NEG ? -b : b, acc);
#endif
}

// bcast_K(src) of a register that may have been written by the immediately preceding asm
// statements: carries its own two wait states (H1).
template <int K>
__device__ __forceinline__ double bcast_fenced(double src) {
#if SVAE_FUSED_DPP
  double out;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "=v"(out)
               : "v"(src), "n"(K));
  return out;
#else
  return bcast<K>(src);
#endif
}

// Order every earlier producer of x[0..M) before every later DPP read of them, with >= 2 wait
// states in between.  The empty statements emit no code; they only pin program order.
template <int M>
__device__ __forceinline__ void dpp_fence(double (&x)[M]) {
#if SVAE_FUSED_DPP
#pragma unroll
  for (int i = 0; i < M - 1; ++i) asm volatile("" : "+v"(x[i]));
  asm volatile("s_nop 1" : "+v"(x[M - 1]));
#endif
}
__device__ __forceinline__ void dpp_fence(double& x) {
#if SVAE_FUSED_DPP
  asm volatile("s_nop 1" : "+v"(x));
#endif
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Reciprocal to full fp64 accuracy: v_rcp_f64 seed (relative error e = 1 - p r ~ 2^-23, "2^29 ulp") + ONE cubic
// step r (1 + e + e^2) = 3 dependent FMAs; the neglected term e^3 ~ 2^-69 is below the rounding of the last FMA
// (two Newton steps, 4 FMAs, gave the same ~1 ulp result).
__device__ __forceinline__ double rcp_nr(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  e = __builtin_fma(e, e, e);
  return __builtin_fma(r, e, r);
}

// 1/sqrt(p) to full fp64 accuracy without the library's sqrt + divide (~70 instructions): v_rsq_f64 seed
// (~2^-24) + two coupled Newton steps (r <- r + r e (1/2 + 3/8 e), e = 1 - p r^2: cubic convergence).
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
  static_for<0, 2>([&](auto) {
    const double pr = p * r;
    const double e = __builtin_fma(-pr, r, 1.0);
    const double h = __builtin_fma(0.375, e, 0.5);
    r = __builtin_fma(r * e, h, r);
  });
  return r;
}

// Ordered pieces of the reciprocal chain (v_rcp_f64 + two Newton steps) for hand-pipelined pivoting:
// asm statements keep their place between the DPP row updates.
__device__ __forceinline__ double asm_rcp(double p) {
  double r;
  asm volatile("v_rcp_f64 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(p));   // trans result: 1 wait state
  return r;
}
__device__ __forceinline__ double asm_fnma1(double a, double b) {        // 1 - a*b
  double r;
  asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double asm_fma(double a, double b, double c) { // a*b + c
  double r;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Sum over the 16 lanes of a DPP row (result valid in every lane of the row).
// exp(x) for x <= 0 (max-subtracted log-weights): x = n ln2 + r, |r| <= ln2 / 2, Taylor to r^13 (remainder < 4e-18),
// scaled by 2^n; underflows to 0 like exp().  21 instructions against ~42 of the library's exp(): the kernels that call
// it once per step (HMM forward pass, GMM label update) are bound by their instruction count.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -746.0);
  const double n = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                                   // 1/13!
  p = __builtin_fma(p, r, 2.08767569878680989792e-09);                 // 1/12!
  p = __builtin_fma(p, r, 2.50521083854417187751e-08);                 // 1/11!
  p = __builtin_fma(p, r, 2.75573192239858906526e-07);                 // 1/10!
  p = __builtin_fma(p, r, 2.75573192239858906526e-06);                 // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873015873e-05);                 // 1/8!
  p = __builtin_fma(p, r, 1.98412698412698412698e-04);                 // 1/7!
  p = __builtin_fma(p, r, 1.38888888888888888889e-03);                 // 1/6!
  p = __builtin_fma(p, r, 8.33333333333333333333e-03);                 // 1/5!
  p = __builtin_fma(p, r, 4.16666666666666666667e-02);                 // 1/4!
  p = __builtin_fma(p, r, 1.66666666666666666667e-01);                 // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

__device__ __forceinline__ double row_sum16(double x) {
  x += __shfl_xor(x, 1, 16);
  x += __shfl_xor(x, 2, 16);
  x += __shfl_xor(x, 4, 16);
  x += __shfl_xor(x, 8, 16);
  return x;
}

// Sum over aligned groups of G = 4, 8 or 16 lanes with DPP lane permutations (quad_perm, row_half_mirror, row_mirror:
// no LDS crossbar); result valid in every lane of the group.
template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <int G>
__device__ __forceinline__ double group_sum(double x) {
  static_assert(G == 4 || G == 8 || G == 16, "group of 4, 8 or 16 lanes");
  x += dpp_perm_f64<0xB1>(x);                        // quad_perm [1,0,3,2]
  x += dpp_perm_f64<0x4E>(x);                        // quad_perm [2,3,0,1]
  if constexpr (G >= 8) x += dpp_perm_f64<0x141>(x); // row_half_mirror
  if constexpr (G == 16) x += dpp_perm_f64<0x140>(x);// row_mirror
  return x;
}

// Streaming stores.  The per-step records of the large-batch kernels (hand-off, LDL' factor, cross moments, adjoint
// records: GBs per launch, read once by a LATER phase or kernel) gain nothing from being allocated in L2 / MALL on the
// way out; written with the non-temporal hint they leave the cache to the operands being read.  Measured on
// slds_mix_pair_kernel (2.45 GB written in full 800-byte rows): 2.3 -> 3.8 TB/s.  NOT for the 88-byte row pieces the
// register kernels store per DPP row: as partial lines without the L2 to merge them the packed E-step's hand-off went
// 1.62 -> 2.2 ms at 4096 sequences (the adjoint records of sweep 1: -3 %, within the box-to-box spread; not kept).
template <bool NT>
__device__ __forceinline__ void st_stream(double* p, double v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

}  // namespace svae
