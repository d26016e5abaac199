// lds_chol_tile.hip -- the sampler's noise factor chol(P_t)^-T and its adjoint for latent dimension 16 <= n <= 64, one
// wavefront per (sequence, step), on the hand-off of the LDS-tiled E-step kernel (svae_lds_tile_noise_f64: what
// _natural_sample / _natural_sample_grad do with dpotrf / dtrtrs in the reference, cython_gaussian_grads.pxd:431-487).
// Blocked with 16 x 16 tiles: tile products on v_mfma_f64_16x16x4, tile factorisations on a DPP row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "per_device.hpp"

namespace svae {

constexpr int TV_MAX_S = 16;

// ---- the sampler's noise factor and its adjoint, one wavefront per (sequence, step) ----------------------------------
// The reference's noise map is noise_t = chol(P_t)^-T eps_t (cython_gaussian_grads.pxd:431-454).  From the hand-off
// only P_t^-1 is at hand: chol(P_t)^-T is the unique UPPER-triangular M with P_t^-1 = M M' (positive diagonal), a
// "UL" Cholesky factorisation.
//   mode 0 (sampler):  noise[s] = M eps[s]
//   mode 1 (VJP):      pinv_bar += sym( M^-T Phi(M' Mbar) M^-1 ),  Mbar = triu(sum_s xbar_s eps_s'),
//                      Phi = upper triangle with the diagonal halved -- the Cholesky adjoint (Murray 2016) carried
//                      over to the UL form by the index-reversal permutation.
// History: round 2 worked on LDS-resident matrices (240 .. 780 us per matrix in barriers and LDS latency); round 3 first
// held the matrix in registers, lane = column, every wave-uniform operand through a v_readlane pair (64 us per 64 x 64
// factor, 137 us per adjoint: ~10 cycles per instruction, and minutes of compile time for the unrolled loops) -- at 256
// sequences x 1000 steps that was 60 ms of a 98 ms training pass.  The blocked form below does the O(n^3) work on the
// matrix cores.
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

// lane `src` (compile-time after unrolling) of a wavefront-wide double, as a wave-uniform value: two v_readlane_b32
// into SGPRs, which the consuming v_fma_f64 takes as a scalar operand -- no LDS round trip
__device__ __forceinline__ double bcast_lane(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

// ---- the blocked factorisation ------------------------------------------------------------------------------------------
// Blocked with 16 x 16 tiles the factorisation of a 64 x 64 matrix is 16 tile products on v_mfma_f64_16x16x4 plus four
// 16 x 16 tile factorisations on a DPP row (the elimination of the E-step's pivot tiles, lds_estep_tile.hip).
// The UL factor of P^-1 (M upper triangular, P^-1 = M M') is the ordinary Cholesky factor of the index-reversed matrix:
// with J the reversal, A~ = J P^-1 J = L L' and M = J L J -- so the kernel reverses on load and store and runs a plain
// left-to-right blocked LL' in between, one wavefront per matrix, the matrix in a row-major LDS panel (row stride
// == 2 mod 32 doubles: conflict-free fragment reads):
//   per block column kb:  tile factor A~_kk = L_kk L_kk' (DPP row; also W = L_kk^-1),  panel L_ik = A~_ik W' (i > kb),
//                         trailing update A~_ij -= L_ik L_jk' (kb < j <= i)
//   then  noise~ = L eps~  with the row of L in the lane's registers and eps~_j through v_readlane.
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mma16(const d4 a, const d4 b, d4 c) {
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c, 0, 0, 0);
  return c;
}
// lane = 16 kq + r16.  frag_a: A operand of the row-major tile Tl at (row0, col0) == B operand of Tl';
// frag_b: B operand of Tl == the C/D layout
__device__ __forceinline__ d4 frag_a(const double* M, int ld, int row0, int col0, int r16, int kq) {
  const double* p = M + (row0 + r16) * ld + col0 + kq;
  return d4{p[0], p[4], p[8], p[12]};
}
__device__ __forceinline__ d4 frag_b(const double* M, int ld, int row0, int col0, int r16, int kq) {
  const double* p = M + (row0 + kq) * ld + col0 + r16;
  return d4{p[0], p[4 * ld], p[8 * ld], p[12 * ld]};
}
__device__ __forceinline__ void store_c(double* M, int ld, int row0, int col0, int r16, int kq, const d4 v) {
  double* p = M + (row0 + kq) * ld + col0 + r16;
  p[0] = v[0]; p[4 * ld] = v[1]; p[8 * ld] = v[2]; p[12 * ld] = v[3];
}

// 16 x 16 SPD tile (row-major at `tile`, row stride ld) -> its Cholesky factor L (lower, written back over the tile with
// zeros above the diagonal) and W = L^-1 (lower, row stride ldw).  Lane r16 = column, one register per row, the four DPP
// rows work redundantly.  Forward elimination A = Lu D Lu': row i > p gets row_i -= (A[i][p] / d_p) row_p with the
// multiplier written into lane p, so that lanes c < i of row i end as (Lu^-1)[i][c]; L = Lu D^1/2 leaves column by
// column (row p of the running Schur complement IS its column p), W = D^-1/2 Lu^-1 at the end.
__device__ __forceinline__ void chol_tile_factor(double* tile, int ld, double* W, int ldw, int r16, int kq, bool& bad) {
  double A[16];
  static_for<0, 16>([&](auto r) { A[r] = tile[r * ld + r16]; });
  dpp_fence(A);
  double rs[16];
  static_for<0, 16>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    const double pv = bcast_fenced<p>(A[p]);
    bad = bad || !(pv > 0.0);
    const double rq = rsq_nr(pv);
    rs[p] = rq;
    const double Ep = (r16 == p) ? 1.0 : 0.0;
    const double r = __builtin_fma(Ep, 1.0 - pv, A[p]) * (rq * rq);      // lane p: 1 / d_p; lane c != p: A[p][c] / d_p
    if (kq == 0) tile[r16 * ld + p] = (r16 >= p) ? A[p] * rq : 0.0;       // L[c][p] = A[p][c] / sqrt(d_p)
    // row updates in groups of four: the lane-p clears first (compiler code), then the DPP multiply-accumulates
    // (their DPP-read operand, the old row, was written at least four instructions earlier)
    auto update = [&](auto i0, auto cnt) {
      constexpr int I0 = decltype(i0)::value, C = decltype(cnt)::value;
      double olds[C], accs[C];
      static_for<0, C>([&](auto j) { olds[j] = A[I0 + j]; accs[j] = __builtin_fma(-olds[j], Ep, olds[j]); });
      static_for<0, C>([&](auto j) { mac_bc<p, true, (C < 4)>(accs[j], olds[j], r); A[I0 + j] = accs[j]; });
    };
    constexpr int REM = 15 - p;
    static_for<0, (REM + 3) / 4>([&](auto g) {
      constexpr int i0 = p + 1 + 4 * decltype(g)::value;
      constexpr int c = (16 - i0) < 4 ? (16 - i0) : 4;
      update(std::integral_constant<int, i0>{}, std::integral_constant<int, c>{});
    });
  });
  if (kq == 0) {
    static_for<0, 16>([&](auto r) { W[r * ldw + r16] = r16 < r ? A[r] * rs[r] : (r16 == r ? rs[r] : 0.0); });
  }
}

// Where W_kb = L_kk^-1 is kept.  KEEP = false: one 16 x 18 slot behind the panel (the caller does not need the inverses
// afterwards).  KEEP = true: one slot per block -- for NB >= 3 in the strictly upper tiles of the panel, which nothing else
// uses ((0,1) .. (0,NB-1), then (1,2)): the panel alone is 33.8 KB at n = 64, four one-wavefront workgroups per CU.
template <int NB, bool KEEP>
struct WSlots {
  static constexpr int LD = 16 * NB + 2;
  static constexpr bool in_panel = KEEP && NB >= 3;
  static constexpr int ldw = in_panel ? LD : 18;
  static constexpr int extra_doubles = in_panel ? 0 : (KEEP ? NB : 1) * 16 * 18;
  static __device__ __forceinline__ double* at(double* Am, int kb) {
    if (in_panel) return Am + 16 * (kb < NB - 1 ? 0 : 1) * LD + 16 * (kb < NB - 1 ? kb + 1 : 2);
    return Am + 16 * NB * LD + (KEEP ? kb : 0) * 16 * 18;
  }
};

// Loads A~ = J P^-1 J into the panel Am (NP x LD) and factors it in place: the lower tiles become L, W_kb goes to its slot.
template <int NB, bool KEEP>
__device__ __forceinline__ void load_and_factor(const double* P, int n, double* Am, int c, int r16, int kq, bool& bad) {
  constexpr int NP = 16 * NB, LD = NP + 2;
  using WS = WSlots<NB, KEEP>;
  // A~[i][c] = P^-1[n-1-i][n-1-c]; identity on the padding (which stays the TRAILING block: it does not touch L)
  if (c < NP) {
    const int cs = c < n ? n - 1 - c : 0;
#pragma unroll 16
    for (int i = 0; i < NP; ++i) {
      const double v = P[(long)(i < n ? n - 1 - i : 0) * NP + cs];
      Am[i * LD + c] = (i < n && c < n) ? v : (i == c ? 1.0 : 0.0);
    }
  }
  static_for<0, NB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
    double* W = WS::at(Am, kb);
    chol_tile_factor(Am + (16 * kb) * LD + 16 * kb, LD, W, WS::ldw, r16, kq, bad);
    if constexpr (kb + 1 < NB) {
      const d4 fw = frag_a(W, WS::ldw, 0, 0, r16, kq);                  // B operand of W'
      static_for<kb + 1, NB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const d4 l = mma16(frag_a(Am, LD, 16 * i, 16 * kb, r16, kq), fw, d4{0.0, 0.0, 0.0, 0.0});
        store_c(Am, LD, 16 * i, 16 * kb, r16, kq, l);
      });
      static_for<kb + 1, NB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const d4 nl = -frag_a(Am, LD, 16 * i, 16 * kb, r16, kq);
        static_for<kb + 1, i + 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const d4 u = mma16(nl, frag_a(Am, LD, 16 * j, 16 * kb, r16, kq), frag_b(Am, LD, 16 * i, 16 * j, r16, kq));
          store_c(Am, LD, 16 * i, 16 * j, r16, kq, u);
        });
      });
    }
  });
}

template <int NB>
__global__ __launch_bounds__(64) void tile_noise_mfma_kernel(int n, int S, int T, int t0, int tlen, const double* ws,
                                                             const double* eps, double* noise, int32_t* info) {
  constexpr int NP = 16 * NB, LD = NP + 2;
  extern __shared__ double sm[];
  double* Am = sm;                 // NP x LD: A~, its lower tiles become L  (+ one 16 x 18 slot: W of the current block)
  const long bt = (long)(blockIdx.x / tlen) * T + t0 + blockIdx.x % tlen;
  const int c = threadIdx.x, r16 = c & 15, kq = c >> 4;
  const double* P = ws + bt * (2L * NP * NP + NP) + (long)NP * NP;
  bool bad = false;
  load_and_factor<NB, false>(P, n, Am, c, r16, kq, bad);
  if (bad && c == 0) atomicMax(info, 1);
  // noise~ = L eps~: lane = row (the upper tiles of the panel still hold A~: masked), eps~_j = eps[n-1-j] from lane j
  double Lr[NP];
  const int cr = c < NP ? c : 0;
#pragma unroll
  for (int j = 0; j < NP; ++j) { const double v = Am[cr * LD + j]; Lr[j] = (j <= (cr | 15)) ? v : 0.0; }
  for (int s_ = 0; s_ < S; ++s_) {
    const double ev = (c < n) ? eps[(bt * S + s_) * n + (n - 1 - c)] : 0.0;
    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int j = 0; j < NP; j += 2) {
      v0 = __builtin_fma(Lr[j], bcast_lane(ev, j), v0);
      v1 = __builtin_fma(Lr[j + 1], bcast_lane(ev, j + 1), v1);
    }
    if (c < n) noise[(bt * S + s_) * n + (n - 1 - c)] = v0 + v1;
  }
}


// ---- mode 1: the adjoint, every O(n^3) stage a list of tile products with the intermediates in REGISTERS ----------------
// In the reversed coordinates (L lower, x~_s = J xbar_s, e~_s = J eps_s) the cotangent is
//     pinv_bar~ += 1/2 Linv' S Linv,   S = K + K',   K = Phi(L' Lbar) = [sum_s u_s e~_s']  (lower triangle, diagonal halved),
//     u_s = L' x~_s,  Linv = L^-1.
// What makes it LDS-free beyond the one panel: the MFMA C/D layout of a tile is the B-operand layout of the tile and the
// A-operand layout of its TRANSPOSE -- so U' = X~' L (C layout) serves as A operand of U and B operand of U'; S is built
// column by column directly in the layout the next product wants (S symmetric: A operand of S_ik = C layout of S_ki);
// Y' = S Linv comes out as the A operand of Y = Linv' S, and Q = Y Linv finishes.  x~ and e~ are loaded from global
// memory straight into operand fragments.  L^-1 overwrites L row by row (L Linv = I:  Linv_ij = -W_i sum_k L_ik Linv_kj).
// Tile products at n = 64: 16 (factor) + 10 (U') + 16 (Linv) + 20 (S) + 40 (Y') + 40 (Q) = 142.
template <int NB>
__global__ __launch_bounds__(64) void tile_noise_adj_mfma_kernel(int n, int S, int T, int t0, int tlen, const double* ws,
                                                                 const double* eps, const double* xbar, double* pinv_bar,
                                                                 int32_t* info) {
  constexpr int NP = 16 * NB, LD = NP + 2;
  extern __shared__ double sm[];
  double* Am = sm;                 // NP x LD: A~ -> L (lower tiles) -> Linv;  W_i = L_ii^-1 in their slots (WSlots)
  using WS = WSlots<NB, true>;
  const long bt = (long)(blockIdx.x / tlen) * T + t0 + blockIdx.x % tlen;
  const int c = threadIdx.x, r16 = c & 15, kq = c >> 4;
  const double* P = ws + bt * (2L * NP * NP + NP) + (long)NP * NP;
  // operand fragments of x~ and e~ (requested before the factorisation, consumed after it):
  //   xr[k][kb] = x~_{s = r16}[16 k + 4 kb + kq]      A operand of X~_k'  (X~_k: rows of block k x 16 samples)
  //   er[j][kb] = e~_{s = 4 kb + kq}[16 j + r16]      A operand of E_j == B operand of E_j'
  d4 xr[NB], er[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int rx = 16 * k + 4 * kb + kq, sx = r16;
      const bool okx = sx < S && rx < n;
      const double vx = xbar[(bt * S + (okx ? sx : 0)) * n + (okx ? n - 1 - rx : 0)];
      xr[k][kb] = okx ? vx : 0.0;
      const int re = 16 * k + r16, se = 4 * kb + kq;
      const bool oke = se < S && re < n;
      const double ve = eps[(bt * S + (oke ? se : 0)) * n + (oke ? n - 1 - re : 0)];
      er[k][kb] = oke ? ve : 0.0;
    }
  bool bad = false;
  load_and_factor<NB, true>(P, n, Am, c, r16, kq, bad);
  if (bad && c == 0) atomicMax(info, 1);
  const d4 z4 = {0.0, 0.0, 0.0, 0.0};
  // U'_i = sum_{k >= i} X~_k' L_ki   (C layout: [sample][row of block i])
  d4 ut[NB];
  static_for<0, NB>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    d4 acc = z4;
    static_for<i, NB>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      acc = mma16(xr[k], frag_b(Am, LD, 16 * k, 16 * i, r16, kq), acc);
    });
    ut[i] = acc;
  });
  // Linv over L, row by row
  static_for<0, NB>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    d4 li[i > 0 ? i : 1];
    const d4 fwa = frag_a(WS::at(Am, i), WS::ldw, 0, 0, r16, kq);
    static_for<0, i>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      d4 r = z4;
      static_for<j, i>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        r = mma16(frag_a(Am, LD, 16 * i, 16 * k, r16, kq), frag_b(Am, LD, 16 * k, 16 * j, r16, kq), r);
      });
      li[j] = mma16(fwa, -r, z4);
    });
    const d4 wd = frag_b(WS::at(Am, i), WS::ldw, 0, 0, r16, kq);
    static_for<0, i>([&](auto jc) { store_c(Am, LD, 16 * i, 16 * decltype(jc)::value, r16, kq, li[decltype(jc)::value]); });
    store_c(Am, LD, 16 * i, 16 * i, r16, kq, wd);
  });
  // Y' = S Linv, block row i at a time: column i of S (tiles S_ki, C layout) is the A operand of row i of S
  d4 yt[NB][NB];
  static_for<0, NB>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    d4 sc[NB];
    static_for<0, NB>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k > i) sc[k] = mma16(ut[k], er[i], z4);             // K_ki = U_k E_i'
      else if constexpr (k < i) sc[k] = mma16(er[k], ut[i], z4);        // K_ik' = E_k U_i'
      else {
        const d4 g = mma16(ut[i], er[i], z4), gt = mma16(er[i], ut[i], z4);
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[k][q] = (4 * q + kq >= r16) ? g[q] : gt[q];     // lower + diagonal: G; upper: G'
      }
    });
    static_for<0, NB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      d4 acc = z4;
      static_for<j, NB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        acc = mma16(sc[k], frag_b(Am, LD, 16 * k, 16 * j, r16, kq), acc);
      });
      yt[i][j] = acc;
    });
  });
  // Q = Y Linv (A operand of Y_ik = C layout of Y'_ki);  pinv_bar[n-1-row][n-1-col] += Q / 2
  double* out = pinv_bar + bt * n * n;
  static_for<0, NB>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    static_for<0, NB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // (the old values are requested unconditionally, at clamped addresses, in front of the tile's products: a load
      //  inside the bounds check would be waited for element by element)
      double prev[4];
      int idx[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * i + 4 * q + kq, col = 16 * j + r16;
        idx[q] = (row < n && col < n) ? (n - 1 - row) * n + (n - 1 - col) : -1;
        prev[q] = out[idx[q] < 0 ? 0 : idx[q]];
      }
      d4 acc = z4;
      static_for<j, NB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        acc = mma16(yt[k][i], frag_b(Am, LD, 16 * k, 16 * j, r16, kq), acc);
      });
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (idx[q] >= 0) out[idx[q]] = __builtin_fma(0.5, acc[q], prev[q]);
    });
  });
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
  auto go0 = [&](auto nb) -> int {        // mode 0: the blocked kernel
    constexpr int NB = decltype(nb)::value;
    const size_t lds = (size_t)(16 * NB * (16 * NB + 2) + svae::WSlots<NB, false>::extra_doubles) * sizeof(double);
    hipLaunchKernelGGL(svae::tile_noise_mfma_kernel<NB>, dim3((unsigned)((long)B * tlen)), dim3(64), lds, st, n, S, T, t_begin,
                       tlen, (const double*)handoff_workspace, eps, noise, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  };
  if (mode == 0) {
    if (NP == 16) return go0(std::integral_constant<int, 1>{});
    if (NP == 32) return go0(std::integral_constant<int, 2>{});
    if (NP == 48) return go0(std::integral_constant<int, 3>{});
    return go0(std::integral_constant<int, 4>{});
  }
  auto go1 = [&](auto nb) -> int {        // mode 1
    constexpr int NB = decltype(nb)::value;
    const size_t lds = (size_t)(16 * NB * (16 * NB + 2) + svae::WSlots<NB, true>::extra_doubles) * sizeof(double);
    hipLaunchKernelGGL(svae::tile_noise_adj_mfma_kernel<NB>, dim3((unsigned)((long)B * tlen)), dim3(64), lds, st, n, S, T,
                       t_begin, tlen, (const double*)handoff_workspace, eps, xbar, pinv_bar, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  };
  if (NP == 16) return go1(std::integral_constant<int, 1>{});
  if (NP == 32) return go1(std::integral_constant<int, 2>{});
  if (NP == 48) return go1(std::integral_constant<int, 3>{});
  return go1(std::integral_constant<int, 4>{});
}
