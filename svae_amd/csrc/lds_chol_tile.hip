// lds_chol_tile.hip -- the sampler's noise factor chol(P_t)^-T and its adjoint for latent dimension 16 <= n <= 64, one
// wavefront per (sequence, step), on the hand-off of the LDS-tiled E-step kernel (svae_lds_tile_noise_f64: what
// _natural_sample / _natural_sample_grad do with dpotrf / dtrtrs in the reference, cython_gaussian_grads.pxd:431-487).
// Its own translation unit: the fully unrolled 64 x 64 loops take minutes to compile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/svae_hip.h"
#include "per_device.hpp"

namespace svae {

constexpr int TV_MAX_S = 16;

// ---- the sampler's noise factor and its adjoint, one wavefront per (sequence, step) ----------------------------------
// The reference's noise map is noise_t = chol(P_t)^-T eps_t (cython_gaussian_grads.pxd:431-454).  From the hand-off
// only P_t^-1 is at hand: chol(P_t)^-T is the unique UPPER-triangular M with P_t^-1 = M M' (positive diagonal), a
// "UL" Cholesky factorisation computed from the last column backwards.
//   MODE 0 (sampler):  noise[s] = M eps[s]
//   MODE 1 (VJP):      pinv_bar += sym( M^-T Phi(M' Mbar) M^-1 ),  Mbar = triu(sum_s xbar_s eps_s'),
//                      Phi = upper triangle with the diagonal halved -- the Cholesky adjoint (Murray 2016) carried
//                      over to the UL form by the index-reversal permutation.
// Layout: lane c holds COLUMN c of the matrix being worked on in NC registers (NC = n rounded up to 16, identity
// padding; every loop fully unrolled), so the factorisation and both triangular solves are register FMAs whose
// second operand is a wave-uniform LDS read (the other column / the factor entry, at a compile-time offset); a
// first version on LDS-resident matrices spent 240 .. 780 us per matrix in barriers, index divisions and LDS
// latency against ~25 us of arithmetic.
__device__ __forceinline__ double rdiag(double p) {      // 1/p: v_rcp_f64 + two Newton steps
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(r, e, r);
}

// 1/sqrt(p) to full fp64 accuracy without the library's sqrt + divide (their slow-path branches cost ~70 instructions
// per pivot): v_rsq_f64 seed + two coupled Newton steps (cf. rsqrt_nr, dpp.hpp)
__device__ __forceinline__ double rsq_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double pr = p * r;
    const double e = __builtin_fma(-pr, r, 1.0);
    const double h = __builtin_fma(0.375, e, 0.5);
    r = __builtin_fma(r * e, h, r);
  }
  return r;
}

template <int NC>
struct CholCfg {
  static constexpr int LD = NC + 1;
  static constexpr int PAN = NC * LD;
  // ONE panel (mode 0: M row-major; mode 1: the transpositions): 33 KB at NC = 64, so that four one-wavefront workgroups
  // share a CU, one per SIMD (through round 3 the round-2 layout of 76 KB was still requested: two per CU, half the
  // SIMDs idle)
  static constexpr int LDS_DOUBLES = PAN;
};

// lane `src` (compile-time after unrolling) of a wavefront-wide double, as a wave-uniform value: two v_readlane_b32
// into SGPRs, which the consuming v_fma_f64 takes as a scalar operand -- no LDS round trip
__device__ __forceinline__ double bcast_lane(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

// Round 3: every wave-uniform operand (an entry of another lane's column) comes from that lane's REGISTER through
// v_readlane instead of from an LDS panel.  The round-2 form fed each FMA a wave-uniform LDS read; with the two
// column arrays filling the register file nothing could be batched, and the kernel ran at ~60 cycles per multiply-add
// (368 us per 64 x 64 matrix and wavefront).  The factorisation needs no LDS at all: the multiplier M[k][j] lane k
// wants at pivot j is, by the symmetry the running Schur complement keeps, its OWN register j.
template <int NC, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void tile_chol_kernel(int n, int S, int NP, int T, int t0, int tlen, const double* ws, const double* eps,
                                                       const double* xbar, double* noise, double* pinv_bar,
                                                       int32_t* info) {
  using Cfg = CholCfg<NC>;
  constexpr int LD = Cfg::LD;
  extern __shared__ double sm[];
  double* panM = sm;                       // MODE 0: M (upper), row-major
  double* pan2 = sm;                       // MODE 1: transpositions
  const long bt = (long)(blockIdx.x / tlen) * T + t0 + blockIdx.x % tlen;   // steps t0 .. t0 + tlen - 1 of every sequence
  const int c = threadIdx.x;               // lane = column
  const bool on = c < NC;
  const double* P = ws + bt * (2L * NP * NP + NP) + (long)NP * NP;
  double A[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) A[i] = (on && i < n && c < n) ? P[(long)i * NP + (c < n ? c : 0)] : ((i == c) ? 1.0 : 0.0);
  bool bad = false;
  // UL Cholesky, pivots NC-1 .. 0.  With d = A[j][j], r = 1 / sqrt(d):  column j becomes M[:, j] = a r, and every
  // column k < j loses M[i][j] M[k][j] = a_ij (r^2 a_kj) in its rows i < j.  Both are ONE fma per row with the
  // wave-uniform a_ij (lane j's register i) and a per-lane factor w: -(r^2 a_kj) for k < j -- a_kj read from the
  // lane's own register j (the Schur complement is symmetric and its lower entries are kept up to date by the same
  // updates) --, r - 1 for lane j itself (a + a (r - 1) = a r), 0 for the finished columns.
#pragma unroll
  for (int j = NC - 1; j >= 0; --j) {
    const double d = bcast_lane(A[j], j);
    bad = bad || !(d > 0.0);
    const double r = rsq_nr(d);
    const double w = (c < j) ? -(A[j] * r) * r : ((c == j) ? r - 1.0 : 0.0);
#pragma unroll
    for (int i = 0; i <= j; ++i) A[i] = __builtin_fma(bcast_lane(A[i], j), w, A[i]);   // (rows below the diagonal: unused)
  }
  if (bad && c == 0) atomicMax(info, 1);
  if constexpr (MODE == 0) {
    // noise[s][i] = sum_{j >= i} M[i][j] eps[s][j]   (lane = row i: M through a row-major LDS panel)
    if (on) {
#pragma unroll
      for (int i = 0; i < NC; ++i) panM[i * LD + c] = (i <= c) ? A[i] : 0.0;
    }
    __syncthreads();
    // row c of M into the lane's registers, eps_s[j] as a wave-uniform operand from lane j's register: fully unrolled,
    // no memory access inside the sums (the first version looped j = i .. n-1 over an LDS read and a global load of
    // eps[j] each -- a serial chain of ~64 memory latencies per sample that took longer than the factorisation)
    if (on) {
#pragma unroll
      for (int j = 0; j < NC; ++j) A[j] = panM[c * LD + j];
    }
    for (int s_ = 0; s_ < S; ++s_) {
      const double ev = (c < n) ? eps[(bt * S + s_) * n + c] : 0.0;
      double v0 = 0.0, v1 = 0.0;
#pragma unroll
      for (int j = 0; j < NC; j += 2) {
        v0 = __builtin_fma(A[j], bcast_lane(ev, j), v0);
        v1 = __builtin_fma(A[j + 1], bcast_lane(ev, j + 1), v1);
      }
      if (c < n) noise[(bt * S + s_) * n + c] = v0 + v1;
    }
  } else {
    // u_s = M' xbar_s (own column), then Phi(M' Mbar)[i][c] = sum_s u_s[i] eps_s[c] for i <= c (diagonal halved):
    // Mbar = triu(sum_s xbar_s eps_s') is never formed (its mask k <= j is implied by k <= i <= j)
    double K[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) K[i] = 0.0;
    for (int s_ = 0; s_ < S; ++s_) {
      const double* xb = xbar + (bt * S + s_) * n;
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < NC; ++k) v = __builtin_fma(A[k], (k <= c && k < n) ? xb[k < n ? k : 0] : 0.0, v);
      const double ev = (c < n) ? eps[(bt * S + s_) * n + c] : 0.0;
#pragma unroll
      for (int i = 0; i < NC; ++i) K[i] = __builtin_fma(bcast_lane(v, i), ev, K[i]);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) K[i] = (i < c) ? K[i] : ((i == c) ? 0.5 * K[i] : 0.0);
    // two passes of:  solve M' X = K column-wise (x_i = K_i / M[i][i];  K_r -= M[i][r] x_i for r > i), then
    // transpose through LDS:  Q1 = M^-T Phi;  Q' = M^-T Q1'  (Q = Q1 M^-1).  (Written as a loop of two trips so
    // that the unrolled body exists once and K stays in registers: as lambdas the array went to scratch memory.)
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const double x = K[i] * rdiag(bcast_lane(A[i], i));
        K[i] = x;
#pragma unroll
        for (int r = i + 1; r < NC; ++r) K[r] = __builtin_fma(-bcast_lane(A[i], r), x, K[r]);
      }
      if (pass == 0) {
        __syncthreads();
        if (on) {
#pragma unroll
          for (int i = 0; i < NC; ++i) pan2[i * LD + c] = K[i];
        }
        __syncthreads();
        if (on) {
#pragma unroll
          for (int i = 0; i < NC; ++i) K[i] = pan2[c * LD + i];
        }
      }
    }
    // pinv_bar += (Q + Q') / 2
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < NC; ++i) pan2[i * LD + c] = K[i];
    }
    __syncthreads();
    if (c < n) {
      double* out = pinv_bar + bt * n * n;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        if (i < n) out[i * n + c] += 0.5 * (K[i] + pan2[c * LD + i]);
    }
  }
}

}  // namespace svae

// The sampler's noise factor (mode 0: noise (B,T,S,n) = chol(P_t)^-T eps_t) and its adjoint (mode 1: adds the
// Cholesky-path cotangent into the pinv_bar section of the VJP workspace, between phases 1 and 2 of
// svae_lds_tile_vjp_f64), from the P_t^-1 of the tiled E-step's hand-off; one workgroup per (sequence, step).
extern "C" int svae_lds_tile_noise_f64(int mode, int B, int T, int n, int S, int t_begin, int t_end, const double* eps, double* noise,
                                       const void* handoff_workspace, void* vjp_workspace, int32_t* info, void* stream) {
  if (mode < 0 || mode > 1) return -1;
  if (B < 0) return -2;
  if (T < 1) return -3;
  if (n < 1 || n > 64) return -4;
  if (S < 1 || S > svae::TV_MAX_S) return -5;
  if (!eps) return -6;
  if (mode == 0 && !noise) return -7;
  if (!handoff_workspace) return -8;
  if (mode == 1 && !vjp_workspace) return -9;
  if (!info) return -10;
  if (t_begin < 0 || t_end > T || t_begin >= t_end) return -20;
  if (B == 0) return 0;
  const int NP = 16 * ((n + 15) / 16);
  const long BT = (long)B * T;
  const int tlen = t_end - t_begin;
  double* w = (double*)vjp_workspace;
  double* pinv_bar = mode == 1 ? w + (size_t)BT * n * n : nullptr;
  const double* xbar = mode == 1 ? w + (size_t)BT * n * n * 2 + (size_t)B * (T > 1 ? T - 1 : 0) * n * n + (size_t)BT * n : nullptr;
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto nc, auto md) -> int {
    constexpr int NC = decltype(nc)::value, MD = decltype(md)::value;
    const size_t lds = (size_t)svae::CholCfg<NC>::LDS_DOUBLES * sizeof(double);
    auto kern = svae::tile_chol_kernel<NC, MD>;
    static svae::LdsGrant grant;            // (one per instantiation of this lambda, per device inside)
    if (lds > 64 * 1024 && !grant.ensure(reinterpret_cast<const void*>(kern), (long)lds)) return -1001;
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * tlen)), dim3(64), lds, st, n, S, NP, T, t_begin, tlen,
                       (const double*)handoff_workspace, eps, xbar,
                       noise, pinv_bar, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (NP == 16) return mode ? go(std::integral_constant<int, 16>{}, I1{}) : go(std::integral_constant<int, 16>{}, I0{});
  if (NP == 32) return mode ? go(std::integral_constant<int, 32>{}, I1{}) : go(std::integral_constant<int, 32>{}, I0{});
  if (NP == 48) return mode ? go(std::integral_constant<int, 48>{}, I1{}) : go(std::integral_constant<int, 48>{}, I0{});
  return mode ? go(std::integral_constant<int, 64>{}, I1{}) : go(std::integral_constant<int, 64>{}, I0{});
}
