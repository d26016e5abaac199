// lds_estep_tile.hip -- LDS E-step for latent dimension 16 <= n <= 64 on MI355X (gfx950): the
// "MFMA only if latent dim >= 16 makes it a real contraction" leg of the hot path.
//
// What it replaces (reference = mattjj/svae): the same functions as the register path
// (lds_estep_kernel.hpp): natural_filter_forward_general svae/lds/cython_lds_inference.pyx:28-90,
// natural_smoother_general :149-195, _compute_stats :197-210, at latent sizes where one matrix no
// longer fits a 16-lane DPP row.
//
// Mapping: one workgroup (4 wavefronts) per sequence, two workgroups per CU.  The per-step
// matrices live in LDS as ONE row-major panel  M = [ P | R | h ]  of NP x (2 NP + 1) doubles
// (NP = n rounded up to 16; P = pivot block J_filt + J11, R = J12, h = filtered potential vector),
// cut in 16x16 tiles (h: a tile column of which only column 0 exists).  Every O(n^3) stage is a
// list of tile products on v_mfma_f64_16x16x4_f64 with A/B fragments read straight from the panel
// (row stride == 2 mod 32 doubles: the A-fragment read is bank-conflict free):
//   forward   in-place BLOCK Gauss-Jordan of [P | R | h] with 16x16 block pivots: P -> P^-1,
//             R -> X = P^-1 J12, h -> c = P^-1 h.  The pivot tile A_kk = L D L' is factored by one
//             wavefront with the DPP elimination of the register path (pivots -> log det P), which
//             leaves U = L^-1 and D^-1; the block row is then A_kk^-1 A_kj = U' D^-1 (U A_kj), two
//             tile products (applying the explicit inverse tile instead costs 2 digits on
//             ill-conditioned models).
//             Then the Schur step  P' = (J22 + J11) - J12' X,  h' = J12' c + node_h.
//   backward  moment form:  W = Sigma_{t+1} X',  Sigma_t = P^-1 + X W,  m_t = c + X m_{t+1};
//             tile column j of W stays in registers (the MFMA C layout IS the B-operand layout)
//             and feeds the second product directly.
// Padding rows/columns (n < NP) carry an identity diagonal: pivots 1, log det unchanged.
#define SVAE_DPP_ALWAYS_FENCED 1
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "lds_args.hpp"

namespace svae {

typedef double d4 __attribute__((ext_vector_type(4)));

// D = A(16x16) * B(16x16) + C as four 16x16x4 MFMAs; a[kb]/b[kb] = k-chunk kb of the fragments.
__device__ __forceinline__ d4 mma16(const d4 a, const d4 b, d4 c) {
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c, 0, 0, 0);
  return c;
}

// Fragment addressing (lane = 16 kq + r16).  For a row-major tile Tl at (row0, col0):
//   frag_a: A operand of Tl   (lane holds Tl[r16][4 kb + kq])   == B operand of Tl'
//   frag_b: B operand of Tl   (lane holds Tl[4 kb + kq][r16])   == A operand of Tl'  == C/D layout
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

template <int NB>
struct TileCfg {
  static constexpr int NP = 16 * NB;
  static constexpr int NTC = 2 * NB + 1;        // tile columns of [P | R | h]; the last one is 1 wide
  static constexpr int LDM = 2 * NP + 2;        // == 2 (mod 32)
  static constexpr int WSTEP = 2 * NP * NP + NP;   // hand-off per step: X, P^-1 (row-major NP x NP), c
  static constexpr int LDU = 18;                // pivot-factor tile U = L^-1, row stride
  static constexpr int UBUF = 16 * LDU + 16;    // U and D^-1; double-buffered
  static constexpr int LDS_DOUBLES = NP * LDM + 3 * NP + 16 + 2 * UBUF;
};

// 16x16 SPD tile A = L D L'  ->  U = L^-1 (unit lower triangular) and D^-1, by the calling wavefront
// (lane r16 = column, one register per row, the four DPP rows work redundantly).  Forward
// elimination only: row i > p gets  row_i -= (A[i][p] / d_p) * row_p, with the multiplier written
// into lane p, so that lanes c < i of row i end as U[i][c].  Accumulates log det as mantissa/exponent.
__device__ __forceinline__ void factor_pivot_tile(const double* tile, int ld, double* U, int ldu,
                                                  double* dinv, int r16, int kq,
                                                  const double (&E)[16], double& pmin, double& ldM,
                                                  int& ldE) {
  double A[16];
  static_for<0, 16>([&](auto r) { A[r] = tile[r * ld + r16]; });
  dpp_fence(A);
  double dv = 0.0;
  static_for<0, 16>([&](auto p) {
    const double pv = bcast_fenced<p>(A[p]);
    pmin = fmin(pmin, pv);
    ldM *= __builtin_amdgcn_frexp_mant(pv);
    ldE += __builtin_amdgcn_frexp_exp(pv);
    const double rinv = rcp_nr(pv);
    dv = __builtin_fma(E[p], rinv, dv);
    const double r = __builtin_fma(E[p], 1.0 - pv, A[p]) * rinv;     // lane p: 1/pivot
    static_for<p + 1, 16>([&](auto i) {
      const double old = A[i];
      double acc = __builtin_fma(-old, E[p], old);                    // lane p of the row -> 0
      mac_bc<p, true, true>(acc, old, r);                             // row_i -= A[i][p] * r
      A[i] = acc;
    });
  });
  ldE += __builtin_amdgcn_frexp_exp(ldM);
  ldM = __builtin_amdgcn_frexp_mant(ldM);
  if (kq == 0) {
    static_for<0, 16>([&](auto r) { U[r * ldu + r16] = r16 < r ? A[r] : E[r]; });
    dinv[r16] = dv;
  }
}

template <int NB, bool INHOMOG>
__global__ __launch_bounds__(256, 2) void lds_estep_tile_kernel(const LdsArgs a, const int n) {
  using Cfg = TileCfg<NB>;
  constexpr int NP = Cfg::NP, NTC = Cfg::NTC, LDM = Cfg::LDM, WSTEP = Cfg::WSTEP;
  extern __shared__ double smem[];
  double* M = smem;
  double* hvec = M + NP * LDM;     // forward: h_filt of the step; backward: c_t
  double* mv0 = hvec + NP;         // forward: next h_filt; backward: mean vectors (double buffer)
  double* mv1 = mv0 + NP;
  double* red = mv1 + NP;          // 16 doubles of reduction scratch
  double* ubuf = red + 16;         // 2 x (U = L^-1 of the pivot tile, row stride LDU | D^-1)
  constexpr int LDU = Cfg::LDU, UBUF = Cfg::UBUF;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x;
  const int T = a.T;
  const long nn = (long)n * n;
  const double* J11 = a.J11 + (long)b * a.pair_seq_stride;
  const double* J12 = a.J12 + (long)b * a.pair_seq_stride;
  const double* J22 = a.J22 + (long)b * a.pair_seq_stride;
  const double* nodeJ = a.node_J + (long)b * T * n;
  const double* nodeh = a.node_h + (long)b * T * n;
  double* wsb = a.ws + (long)b * T * WSTEP;
  auto pair_at = [&](const double* p, int t) { return INHOMOG ? p + (long)t * nn : p; };

  double E[16];
  static_for<0, 16>([&](auto i) { E[i] = (r16 == i) ? 1.0 : 0.0; });
  double ldM = 1.0, pmin = 1.0e300, qacc = 0.0;
  int ldE = 0;

  // B/C fragments by tile column jt of the panel; jt == 2 NB is the h column (one physical column:
  // lanes r16 > 0 see zeros and do not store)
  const double e0 = E[0];
  auto ld_b = [&](int row0, int jt) -> d4 {
    if (jt < 2 * NB) return frag_b(M, LDM, row0, 16 * jt, r16, kq);
    const double* p = M + (row0 + kq) * LDM + 2 * NP;
    return d4{p[0] * e0, p[4 * LDM] * e0, p[8 * LDM] * e0, p[12 * LDM] * e0};
  };
  auto st_c = [&](int row0, int jt, const d4 v) {
    if (jt < 2 * NB) {
      store_c(M, LDM, row0, 16 * jt, r16, kq, v);
    } else if (r16 == 0) {
      double* p = M + (row0 + kq) * LDM + 2 * NP;
      p[0] = v[0]; p[4 * LDM] = v[1]; p[8 * LDM] = v[2]; p[12 * LDM] = v[3];
    }
  };
  // A_kk^-1 (.) = U' D^-1 U (.) applied to a B-layout tile
  auto apply_pivot = [&](const double* U, const double* dinv, const d4 fb) -> d4 {
    d4 v = mma16(frag_a(U, LDU, 0, 0, r16, kq), fb, d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) v[qq] *= dinv[4 * qq + kq];
    return mma16(frag_b(U, LDU, 0, 0, r16, kq), v, d4{0.0, 0.0, 0.0, 0.0});
  };
  // tiles of block column kc that the elimination of block pivot kc left for later:
  //   A[i][kc] <- -A[i][kc] A_kk^-1 (computed transposed: A_kk^-1 A[i][kc]')   and   A[kc][kc] <- A_kk^-1
  auto deferred_column = [&](int kc, const double* U, const double* dinv, int w0, int nw) {
    for (int q = w0; q < NB; q += nw) {
      if (q == NB - 1) {
        d4 v = frag_b(U, LDU, 0, 0, r16, kq);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) v[qq] *= dinv[4 * qq + kq];
        store_c(M, LDM, 16 * kc, 16 * kc, r16, kq, mma16(frag_b(U, LDU, 0, 0, r16, kq), v, d4{0.0, 0.0, 0.0, 0.0}));
      } else {
        const int i = q + (q >= kc ? 1 : 0);
        const d4 rt = apply_pivot(U, dinv, frag_a(M, LDM, 16 * i, 16 * kc, r16, kq));   // (A[i][kc] W)'
        double* p = M + (16 * i + r16) * LDM + 16 * kc + kq;
        p[0] = -rt[0]; p[4] = -rt[1]; p[8] = -rt[2]; p[12] = -rt[3];
      }
    }
  };

  // ---- step 0: P = -2 (init_J + J11) + diag(-2 node_J[0]),  R = J12,  h = init_h + node_h[0] -----
  for (int idx = tid; idx < NP * NP; idx += 256) {
    const int row = idx / NP, col = idx % NP;
    const bool in = row < n && col < n;
    double v = (row == col) ? 1.0 : 0.0, r = 0.0;
    if (in) {
      v = -2.0 * a.init_J[row * n + col];
      if (T > 1) { v -= 2.0 * J11[row * n + col]; r = J12[row * n + col]; }
      if (row == col) v -= 2.0 * nodeJ[row];
    }
    M[row * LDM + col] = v;
    M[row * LDM + NP + col] = r;
  }
  if (tid < NP) {
    const double hv = tid < n ? a.init_h[tid] + nodeh[tid] : 0.0;
    hvec[tid] = hv;
    M[tid * LDM + 2 * NP] = hv;
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const bool last = (t == T - 1);
    // ---- in-place block Gauss-Jordan ---------------------------------------------------------------
    for (int k = 0; k < NB; ++k) {
      double* U = ubuf + (k & 1) * UBUF;
      double* dinv = U + 16 * LDU;
      if (wave == 0) {
        factor_pivot_tile(M + (16 * k) * LDM + 16 * k, LDM, U, LDU, dinv, r16, kq, E, pmin, ldM, ldE);
      } else if (k > 0) {
        const double* Up = ubuf + ((k - 1) & 1) * UBUF;
        deferred_column(k - 1, Up, Up + 16 * LDU, wave - 1, 3);
      }
      __syncthreads();
      // pivot row:  A[k][j] <- A_kk^-1 A[k][j]   (j != k)
      for (int q = wave; q < NTC - 1; q += 4) {
        const int j = q + (q >= k ? 1 : 0);
        st_c(16 * k, j, apply_pivot(U, dinv, ld_b(16 * k, j)));
      }
      __syncthreads();
      // elimination:  A[i][j] -= A[i][k] * A[k][j]   (i != k, j != k); column k itself is deferred
      for (int q = wave; q < (NB - 1) * (NTC - 1); q += 4) {
        const int iq = q / (NTC - 1), jq = q % (NTC - 1);
        const int i = iq + (iq >= k ? 1 : 0), j = jq + (jq >= k ? 1 : 0);
        const d4 fa = frag_a(M, LDM, 16 * i, 16 * k, r16, kq);
        st_c(16 * i, j, mma16(-fa, ld_b(16 * k, j), ld_b(16 * i, j)));
      }
      __syncthreads();
    }
    {
      const double* Up = ubuf + ((NB - 1) & 1) * UBUF;
      deferred_column(NB - 1, Up, Up + 16 * LDU, wave, 4);
    }
    __syncthreads();

    // ---- hand-off to the backward half: X, P^-1 (row-major NP x NP), c --------------------------
    double* w = wsb + (long)t * WSTEP;
    for (int idx = tid; idx < NP * NP; idx += 256) {
      const int row = idx / NP, col = idx % NP;
      w[idx] = M[row * LDM + NP + col];
      w[NP * NP + idx] = M[row * LDM + col];
    }
    if (tid < NP) {
      const double cv = M[tid * LDM + 2 * NP];
      w[2 * NP * NP + tid] = cv;
      qacc = __builtin_fma(hvec[tid], cv, qacc);            // h' P^-1 h
    }
    __syncthreads();

    if (!last) {
      // ---- Schur step:  P' = -2 (J22 + J11') - J12' X   (tile (i,j), j < NB);   h' = J12' c (j == NB) --
      const bool next_last = (t + 1 == T - 1);
      const double* pJ12 = pair_at(J12, t);
      const double* pJ22 = pair_at(J22, t);
      const double* pJ11n = next_last ? nullptr : pair_at(J11, t + 1);
      for (int q = wave; q < NB * (NB + 1); q += 4) {
        const int i = q % NB, j = q / NB;
        d4 c = {0.0, 0.0, 0.0, 0.0};
        if (j < NB) {
          const int col = 16 * j + r16;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int row = 16 * i + 4 * qq + kq;
            double v = (row == col) ? 1.0 : 0.0;
            if (row < n && col < n) {
              v = -2.0 * pJ22[row * n + col];
              if (pJ11n) v -= 2.0 * pJ11n[row * n + col];
            }
            c[qq] = v;
          }
        }
        for (int kk = 0; kk < NB; ++kk) {
          // A operand = -(J12')[tile i][tile kk]:  lane holds -J12[16 kk + 4 kb + kq][16 i + r16]
          d4 fa;
          const int col = 16 * i + r16;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const int row = 16 * kk + 4 * kb + kq;
            fa[kb] = (row < n && col < n) ? -pJ12[row * n + col] : 0.0;
          }
          c = mma16(fa, ld_b(16 * kk, NB + j), c);
        }
        if (j < NB) {
          store_c(M, LDM, 16 * i, 16 * j, r16, kq, c);
        } else if (r16 == 0) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int row = 16 * i + 4 * qq + kq;
            mv0[row] = row < n ? nodeh[(long)(t + 1) * n + row] - c[qq] : 0.0;   // c = -J12' c_t
          }
        }
      }
      __syncthreads();
      // ---- next step's right-hand sides and node diagonal ---------------------------------------
      const double* pJ12n = next_last ? nullptr : pair_at(J12, t + 1);
      for (int idx = tid; idx < NP * NP; idx += 256) {
        const int row = idx / NP, col = idx % NP;
        M[row * LDM + NP + col] = (pJ12n && row < n && col < n) ? pJ12n[row * n + col] : 0.0;
      }
      if (tid < NP) {
        const double hv = mv0[tid];
        hvec[tid] = hv;
        M[tid * LDM + 2 * NP] = hv;
        if (tid < n) M[tid * LDM + tid] -= 2.0 * nodeJ[(long)(t + 1) * n + tid];
      }
      __syncthreads();
    }
  }

  // ---- log-normaliser --------------------------------------------------------------------------
  {
    double z = qacc * 0.5;
    if (a.node_logZ) { for (int t = tid; t < T; t += 256) z += a.node_logZ[(long)b * T + t]; }
    if (INHOMOG) {
      const double* lz = a.logZ_pair + (a.pair_seq_stride ? (long)b * (T - 1) : 0);
      for (int t = tid; t < T - 1; t += 256) z += lz[t];
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) z += __shfl_xor(z, s, 64);
    if (lane == 0) red[wave] = z;
    __syncthreads();
    if (tid == 0) {
      double total = red[0] + red[1] + red[2] + red[3];
      total += a.init_logZ[0];
      if (!INHOMOG && T > 1) total += (double)(T - 1) * a.logZ_pair[0];
      total -= 0.5 * (::log(ldM) + (double)ldE * 0.6931471805599453094);
      a.lognorm[b] = total;
      const bool bad = !(pmin > 0.0) || !(total == total);
      if (bad) {
        int old = *(volatile int32_t*)a.info;
        while (old == 0 || old > b + 1) {
          const int seen = atomicCAS(a.info, old, b + 1);
          if (seen == old) break;
          old = seen;
        }
      }
    }
  }

  // ---- backward pass (moment form) -------------------------------------------------------------
  for (int idx = tid; idx < NP * NP; idx += 256) M[(idx / NP) * LDM + (idx % NP)] = 0.0;   // Sigma_T := 0
  if (tid < NP) { mv0[tid] = 0.0; mv1[tid] = 0.0; }
  double* mold = mv0;
  double* mnew = mv1;
  d4 cross[NB], sxx[NB], lastE[NB], firstE[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    cross[i] = d4{0.0, 0.0, 0.0, 0.0}; sxx[i] = cross[i]; lastE[i] = cross[i]; firstE[i] = cross[i];
  }
  const int j = wave;                       // this wavefront's tile column (waves >= NB only help copy)
  const int mycol = 16 * j + r16;
  double* oEx = a.E_node_x + (long)b * T * n;
  double* oExx = a.E_node_diagxx + (long)b * T * n;
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    const double* w = wsb + (long)t * WSTEP;
    for (int idx = tid; idx < NP * NP; idx += 256) M[(idx / NP) * LDM + NP + (idx % NP)] = w[idx];
    if (tid < NP) hvec[tid] = w[2 * NP * NP + tid];
    __syncthreads();

    {  // m_t = c_t + X_t m_{t+1}
      const int row = tid >> 2, part = tid & 3;
      double s = 0.0;
      if (row < NP) {
        const double* xr = M + row * LDM + NP;
        for (int cc = part; cc < NP; cc += 4) s = __builtin_fma(xr[cc], mold[cc], s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (row < NP && part == 0) mnew[row] = hvec[row] + s;
    }
    d4 Wt[NB], Sn[NB];
    if (j < NB) {
      d4 Bx[NB];
#pragma unroll
      for (int l = 0; l < NB; ++l) Bx[l] = frag_a(M, LDM, 16 * j, NP + 16 * l, r16, kq);   // (X_{jl})' as B
#pragma unroll
      for (int kk = 0; kk < NB; ++kk) {
        d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < NB; ++l) c = mma16(frag_a(M, LDM, 16 * kk, 16 * l, r16, kq), Bx[l], c);
        Wt[kk] = c;                            // W[kk][j] = (Sigma_{t+1} X')  = Cov(x_{t+1}, x_t)
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const double* pi = w + NP * NP + (16 * i + kq) * NP + mycol;
        d4 c = {pi[0], pi[4 * NP], pi[8 * NP], pi[12 * NP]};
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) c = mma16(frag_a(M, LDM, 16 * i, NP + 16 * kk, r16, kq), Wt[kk], c);
        Sn[i] = c;                             // Sigma_t tile (i, j)
      }
    }
    __syncthreads();

    if (j < NB) {
      const double mc = mnew[mycol];
      double* oP = INHOMOG && t < T - 1 ? a.E_pair + ((long)b * (T - 1) + t) * 3 * nn : nullptr;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        store_c(M, LDM, 16 * i, 16 * j, r16, kq, Sn[i]);
        d4 exx, ecr;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int row = 16 * i + 4 * qq + kq;
          exx[qq] = __builtin_fma(mnew[row], mc, Sn[i][qq]);      // E[x_t x_t'](row, mycol)
          ecr[qq] = __builtin_fma(mold[row], mc, Wt[i][qq]);      // E[x_{t+1} x_t'](row, mycol)
        }
        if (INHOMOG) {
          if (oP) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int row = 16 * i + 4 * qq + kq;
              if (row < n && mycol < n) {
                oP[row * n + mycol] = exx[qq];
                oP[nn + mycol * n + row] = ecr[qq];
                oP[2 * nn + row * n + mycol] = lastE[i][qq];      // E[x_{t+1} x_{t+1}'] (previous step)
              }
            }
          }
          lastE[i] = exx;
        } else {
          if (t == T - 1) lastE[i] = exx; else sxx[i] += exx;
          cross[i] += ecr;
        }
        if (t == 0) firstE[i] = exx;
      }
    }
    __syncthreads();
    if (tid < n) {
      const double mm = mnew[tid];
      oEx[(long)t * n + tid] = mm;
      oExx[(long)t * n + tid] = __builtin_fma(mm, mm, M[tid * LDM + tid]);
    }
    double* tmp = mold; mold = mnew; mnew = tmp;
  }

  // ---- E_init and the pair sums -----------------------------------------------------------------
  if (j < NB) {
    double* oI = a.E_init + (long)b * (nn + n);
    double* oP = a.E_pair + (long)b * 3 * nn;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = 16 * i + 4 * qq + kq;
        if (row < n && mycol < n) {
          oI[row * n + mycol] = firstE[i][qq];
          if (!INHOMOG) {
            oP[row * n + mycol] = sxx[i][qq];
            oP[nn + mycol * n + row] = cross[i][qq];
            oP[2 * nn + row * n + mycol] = sxx[i][qq] + lastE[i][qq] - firstE[i][qq];
          }
        }
      }
    }
  }
  if (tid < n) a.E_init[(long)b * (nn + n) + nn + tid] = mold[tid];    // mold == m_0 after the last swap
}

template <int NB>
static int launch_tile(const LdsArgs& a, int n, int inhomog, hipStream_t s) {
  const size_t lds = TileCfg<NB>::LDS_DOUBLES * sizeof(double);
  auto go = [&](auto kern) {
    static bool attr_set = false;     // per instantiation
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -1001;
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds, s, a, n);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  };
  return inhomog ? go(lds_estep_tile_kernel<NB, true>) : go(lds_estep_tile_kernel<NB, false>);
}

}  // namespace svae

extern "C" size_t svae_lds_tile_step_doubles(int n) {
  const int NP = 16 * ((n + 15) / 16);
  return (size_t)2 * NP * NP + NP;
}

extern "C" int svae_lds_launch_tile(const svae::LdsArgs* a, int n, int inhomog, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch ((n + 15) / 16) {
    case 1: return svae::launch_tile<1>(*a, n, inhomog, s);
    case 2: return svae::launch_tile<2>(*a, n, inhomog, s);
    case 3: return svae::launch_tile<3>(*a, n, inhomog, s);
    case 4: return svae::launch_tile<4>(*a, n, inhomog, s);
  }
  return -3;
}
