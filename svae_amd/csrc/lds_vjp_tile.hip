// lds_vjp_tile.hip -- reverse-mode derivative of the LDS E-step (+ sampler) w.r.t. the node potentials for
// latent dimension 16 <= n <= 64 (any n <= 64 works), on the hand-off of the LDS-tiled E-step kernel
// (lds_estep_tile.hip): what natural_filter_grad / natural_smoother_general_grad / natural_sample_backward_grad
// compute in the reference (svae/lds/cython_lds_inference.pyx:92-145, 236-306, 357-409), as the adjoint of THIS
// library's recursion (derivation and torch restatement: svae_amd/lds/lds_large.py, vjp_from_handoff).
//
// Three passes over time, each one workgroup (256 threads) per sequence with the n x n state in LDS:
//   phase 0 (t = T-1 .. 0)  Sigma_t = Pinv_t + G_t Sigma_{t+1} G_t'                      -> sig (B,T,n,n)
//   phase 1 (t = 0 .. T-1)  adjoint of the smoother / sampler recursions: Sigma_bar, m_bar, x_bar
//                           -> pinv_bar (B,T,n,n), g_bar (B,T-1,n,n), c_bar (B,T,n), xbar (B,T,S,n)
//   [the caller adds the Cholesky adjoint of the noise factor, batched over all (b,t), into pinv_bar]
//   phase 2 (t = T-1 .. 0)  adjoint of the filter -> g_node_J, g_node_h (B,T,n)
// Every O(n^3) product is C = A B on LDS-resident operands (row stride 66 doubles), each thread owning a 4 x 4
// block of C (16 x 16 threads), operands read as 16-byte pairs; transposed operands are produced when a
// matrix is copied into LDS, never inside the product.  fp64 VALU: v_mfma_f64 has the same rate on MI355X.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/svae_hip.h"
#include "per_device.hpp"

namespace svae {

// LDS matrices: row stride ld = n4 + 2 doubles (16-byte aligned rows, 2-way bank conflicts at most), one buffer
// = n4 * ld doubles (n4 = n rounded up to 4): 33.8 KB at n = 64, 8.7 KB at n = 32 -- sized at launch, so that
// smaller latent dimensions fit several workgroups per CU
__host__ __device__ constexpr int tv_ld(int n4) { return n4 + 2; }
__host__ __device__ constexpr int tv_mat(int n4) { return n4 * (n4 + 2); }
constexpr int TV_MAX_S = 16;

struct TileVjpArgs {
  int B, T, n, S, NP;                     // NP: padded dimension of the hand-off (n rounded up to 16)
  const double* ws;                       // tile-kernel hand-off: per (b,t) [G (NP x NP) | Pinv (NP x NP) | c (NP)]
  const double* J12; long pair_t_stride, pair_seq_stride;     // natural pair parameter
  const double* E_node_x;                 // (B,T,n) smoothed means
  const double* samples; const double* g_samples;             // (B,T,S,n) or nullptr
  const double* g_lognorm;                // (B)
  const double* g_dxx; const double* g_x; // (B,T,n) or nullptr
  const double* g_E_init;                 // (B, n*n+n) or nullptr
  const double* g_E_pair;                 // (B,T-1,3,n,n) or nullptr: cotangents of the per-step pair statistics
  double* sig; double* pinv_bar; double* g_bar; double* c_bar; double* xbar;   // VJP workspace
  double* g_node_J; double* g_node_h;
};

struct Acc { double v[4][4]; };

__device__ __forceinline__ void acc_zero(Acc& a) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) a.v[i][j] = 0.0;
}

// acc += A B for this thread's 4 x 4 block (rows 4 ty.., columns 4 tx..); A, B in LDS; k < n4 (multiple of 4)
__device__ __forceinline__ void gemm_nn(const double* A, const double* Bm, int n4, int ty, int tx, Acc& acc) {
  const int TV_LD = tv_ld(n4);
  if (4 * ty >= n4 || 4 * tx >= n4) return;
  const double* ap = A + (4 * ty) * TV_LD;
  const double* bp = Bm + 4 * tx;
#pragma unroll 4
  for (int k = 0; k < n4; k += 2) {
    double2 a[4], b0[2], b1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const double2*>(ap + i * TV_LD + k);
    b0[0] = *reinterpret_cast<const double2*>(bp + k * TV_LD);
    b0[1] = *reinterpret_cast<const double2*>(bp + k * TV_LD + 2);
    b1[0] = *reinterpret_cast<const double2*>(bp + (k + 1) * TV_LD);
    b1[1] = *reinterpret_cast<const double2*>(bp + (k + 1) * TV_LD + 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc.v[i][0] = __builtin_fma(a[i].x, b0[0].x, acc.v[i][0]);
      acc.v[i][1] = __builtin_fma(a[i].x, b0[0].y, acc.v[i][1]);
      acc.v[i][2] = __builtin_fma(a[i].x, b0[1].x, acc.v[i][2]);
      acc.v[i][3] = __builtin_fma(a[i].x, b0[1].y, acc.v[i][3]);
      acc.v[i][0] = __builtin_fma(a[i].y, b1[0].x, acc.v[i][0]);
      acc.v[i][1] = __builtin_fma(a[i].y, b1[0].y, acc.v[i][1]);
      acc.v[i][2] = __builtin_fma(a[i].y, b1[1].x, acc.v[i][2]);
      acc.v[i][3] = __builtin_fma(a[i].y, b1[1].y, acc.v[i][3]);
    }
  }
}

// global (row stride gld) -> LDS buffer, zero-padded to n4 x n4; TRANS: dst[c][r] = src[r][c]; scaled by `scale`.
// Thread (ty, tx) copies columns 4 tx .. 4 tx + 3 of rows ty, ty + 16, ty + 32, ty + 48: 16 independent loads in
// flight per thread, 512 contiguous bytes per row across the 16 tx (no index divisions).
template <bool TRANS>
__device__ __forceinline__ void load_mat(double* dst, const double* src, int gld, int n, int n4, double scale) {
  const int TV_LD = tv_ld(n4);
  const int ty = threadIdx.x >> 4, c0 = 4 * (threadIdx.x & 15);
  if (c0 >= n4) return;
  double v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (r < n && c0 + j < n) ? src[(long)r * gld + c0 + j] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i;
    if (r < n4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (TRANS) dst[(c0 + j) * TV_LD + r] = scale * v[i][j]; else dst[r * TV_LD + c0 + j] = scale * v[i][j];
      }
    }
  }
}

// LDS matrix -> global n x n (dense, row stride n), same thread mapping as load_mat
__device__ __forceinline__ void store_mat(double* dst, const double* src, int n) {
  const int TV_LD = tv_ld((n + 3) & ~3);
  const int ty = threadIdx.x >> 4, c0 = 4 * (threadIdx.x & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (r < n && c0 + j < n) dst[(long)r * n + c0 + j] = src[r * TV_LD + c0 + j];
  }
}

__device__ __forceinline__ void store_block(double* dst, const Acc& a, int ty, int tx, int n4, bool trans) {
  const int TV_LD = tv_ld(n4);
  if (4 * ty >= n4 || 4 * tx >= n4) return;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (trans) dst[(4 * tx + j) * TV_LD + 4 * ty + i] = a.v[i][j];
      else dst[(4 * ty + i) * TV_LD + 4 * tx + j] = a.v[i][j];
    }
}

// in place: M <- (M + M') / 2 on the n4 x n4 LDS matrix (barriers inside); each thread handles its 4 x 4 block
__device__ __forceinline__ void symmetrize_lds(double* M, int n4) {
  const int TV_LD = tv_ld(n4);
  const int r0 = 4 * (threadIdx.x >> 4), c0 = 4 * (threadIdx.x & 15);
  const bool on = r0 < n4 && c0 < n4;
  __syncthreads();
  double keep[4][4];
  if (on) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[i][j] = 0.5 * (M[(r0 + i) * TV_LD + c0 + j] + M[(c0 + j) * TV_LD + r0 + i]);
  }
  __syncthreads();
  if (on) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) M[(r0 + i) * TV_LD + c0 + j] = keep[i][j];
  }
  __syncthreads();
}

// y[i] = sum_j A[i][j] x[j]  (TRANS: A[j][i]) for i < n; A in LDS; x, y LDS vectors (y != x); barrier after
template <bool TRANS>
__device__ __forceinline__ void matvec(const double* A, const double* x, double* y, int n) {
  const int TV_LD = tv_ld((n + 3) & ~3);
  for (int i = threadIdx.x; i < n; i += 256) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s = __builtin_fma(TRANS ? A[j * TV_LD + i] : A[i * TV_LD + j], x[j], s);
    y[i] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ const double* handoff(const TileVjpArgs& a, int b, int t) {
  return a.ws + ((long)b * a.T + t) * (2L * a.NP * a.NP + a.NP);
}

// ---- phase 0 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_vjp_phase0(const TileVjpArgs a) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, n = a.n, n4 = (n + 3) & ~3, NP = a.NP, T = a.T;
  const int TV_MAT = tv_mat(n4);
  double *L0 = sm, *L1 = sm + TV_MAT, *L3 = sm + 2 * TV_MAT;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  load_mat<false>(L3, handoff(a, b, T - 1) + (long)NP * NP, NP, n, n4, 1.0);
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    if (t < T - 1) {
      const double* h = handoff(a, b, t);
      load_mat<false>(L0, h, NP, n, n4, 1.0);                     // G_t
      __syncthreads();
      Acc acc;
      acc_zero(acc);
      gemm_nn(L0, L3, n4, ty, tx, acc);                           // G Sigma
      store_block(L1, acc, ty, tx, n4, true);                     // (G Sigma)'
      __syncthreads();
      acc_zero(acc);
      gemm_nn(L0, L1, n4, ty, tx, acc);                           // G (G Sigma)' = G Sigma G'
      const double* P = h + (long)NP * NP;
      if (4 * ty < n4 && 4 * tx < n4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * ty + i, c = 4 * tx + j;
            acc.v[i][j] += (r < n && c < n) ? P[(long)r * NP + c] : 0.0;
          }
      }
      __syncthreads();                                            // everyone done reading L3
      store_block(L3, acc, ty, tx, n4, false);
      symmetrize_lds(L3, n4);
    }
    store_mat(a.sig + ((long)b * T + t) * n * n, L3, n);
  }
}

// ---- phase 1 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_vjp_phase1(const TileVjpArgs a) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, n = a.n, n4 = (n + 3) & ~3, NP = a.NP, T = a.T, S = a.g_samples ? a.S : 0;
  const int TV_LD = tv_ld(n4), TV_MAT = tv_mat(n4);
  double *L0 = sm, *L1 = sm + TV_MAT, *L2 = sm + 2 * TV_MAT, *L3 = sm + 3 * TV_MAT;
  double* vec = sm + 4 * TV_MAT;          // mb (64) | tmp (64) | mnext (64) | xb (S x 64) | xtmp (S x 64) | xnext (S x 64)
  double *mb = vec, *tmpv = vec + 64, *mnext = vec + 128, *xb = vec + 192, *xtmp = xb + TV_MAX_S * 64, *xnext = xtmp + TV_MAX_S * 64;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  for (int e = threadIdx.x; e < TV_MAT; e += 256) L3[e] = 0.0;                 // Sigma_bar
  for (int e = threadIdx.x; e < 192 + 3 * TV_MAX_S * 64; e += 256) vec[e] = 0.0;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const long bt = (long)b * T + t;
    const double* mt = a.E_node_x + bt * n;
    // direct cotangents of step t
    for (int i = threadIdx.x; i < n; i += 256) {
      const double gd = a.g_dxx ? a.g_dxx[bt * n + i] : 0.0;
      L3[i * TV_LD + i] += gd;
      mb[i] += (a.g_x ? a.g_x[bt * n + i] : 0.0) + 2.0 * gd * mt[i];
    }
    if (t == 0 && a.g_E_init) {
      __syncthreads();
      const double* gi = a.g_E_init + (long)b * (n * n + n);
      for (int e = threadIdx.x; e < n * n; e += 256) {
        const int r = e / n, c = e % n;
        L3[r * TV_LD + c] += 0.5 * (gi[r * n + c] + gi[c * n + r]);
      }
      for (int i = threadIdx.x; i < n; i += 256) {
        double s = gi[n * n + i];
        for (int j = 0; j < n; ++j) s = __builtin_fma(gi[i * n + j] + gi[j * n + i], mt[j], s);
        mb[i] += s;
      }
    }
    if (a.g_E_pair) {
      // per-step pair statistics (_compute_stats_grad, cython_lds_inference.pyx:212-234): E_pair[t] = (E x_t x_t',
      // E x_t x_{t+1}', E x_{t+1} x_{t+1}').  Node t is the FIRST node of pair t (cotangents A0, A1) and the SECOND of
      // pair t-1 (A2', A1'):  Sigma_bar_t += sym(A0) + sym(A2'),  m_bar_t += (A0 + A0' + A2' + A2'') m_t + A1 m_{t+1}
      // + A1'' m_{t-1}.  (The covariance part of the cross statistic, G_t Sigma_{t+1}, is handled below.)
      __syncthreads();
      const double* A = t < T - 1 ? a.g_E_pair + ((long)b * (T - 1) + t) * 3 * n * n : nullptr;
      const double* Ap = t > 0 ? a.g_E_pair + ((long)b * (T - 1) + t - 1) * 3 * n * n : nullptr;
      for (int e = threadIdx.x; e < n * n; e += 256) {
        const int r = e / n, c = e % n;
        double v = 0.0;
        if (A) v += 0.5 * (A[r * n + c] + A[c * n + r]);
        if (Ap) v += 0.5 * (Ap[2 * n * n + r * n + c] + Ap[2 * n * n + c * n + r]);
        L3[r * TV_LD + c] += v;
      }
      const double* mnx = a.E_node_x + (bt + 1) * n;
      const double* mpv = a.E_node_x + (bt - 1) * n;
      for (int i = threadIdx.x; i < n; i += 256) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
          if (A) s += (A[i * n + j] + A[j * n + i]) * mt[j] + A[n * n + i * n + j] * mnx[j];
          if (Ap) s += (Ap[2 * n * n + i * n + j] + Ap[2 * n * n + j * n + i]) * mt[j] + Ap[n * n + j * n + i] * mpv[j];
        }
        mb[i] += s;
      }
    }
    for (int e = threadIdx.x; e < S * n; e += 256) {
      const int s_ = e / n, i = e % n;
      xb[s_ * 64 + i] += a.g_samples[(bt * a.S + s_) * n + i];
    }
    __syncthreads();
    // records for phase 2
    store_mat(a.pinv_bar + bt * n * n, L3, n);
    for (int i = threadIdx.x; i < n; i += 256) {
      double s = mb[i];
      for (int s_ = 0; s_ < S; ++s_) s += xb[s_ * 64 + i];
      a.c_bar[bt * n + i] = s;
    }
    for (int e = threadIdx.x; e < S * n; e += 256) a.xbar[(bt * a.S + e / n) * n + e % n] = xb[(e / n) * 64 + e % n];
    if (t == T - 1) break;
    // propagate to t + 1
    const double* h = handoff(a, b, t);
    load_mat<false>(L0, h, NP, n, n4, 1.0);                                      // G_t
    load_mat<false>(L2, a.sig + (bt + 1) * n * n, n, n, n4, 1.0);               // Sigma_{t+1}
    __syncthreads();
    Acc acc;
    acc_zero(acc);
    gemm_nn(L3, L0, n4, ty, tx, acc);                                           // SG = Sigma_bar G
    store_block(L1, acc, ty, tx, n4, false);
    __syncthreads();
    acc_zero(acc);
    gemm_nn(L1, L2, n4, ty, tx, acc);                                           // SG Sigma_{t+1}
    Acc accp;                                                                    // A1 Sigma_{t+1} (pair-statistic cotangent)
    acc_zero(accp);
    if (a.g_E_pair) {
      // E x_t x_{t+1}' = G_t Sigma_{t+1} + m_t m_{t+1}':  G_bar += A1 Sigma_{t+1};  Sigma_bar_{t+1} += sym(G_t' A1).
      // Sigma_bar (L3) has been consumed (SG in L1, pinv_bar stored): A1 takes its buffer until the new one is formed.
      load_mat<false>(L3, a.g_E_pair + (((long)b * (T - 1) + t) * 3 + 1) * n * n, n, n, n4, 1.0);
      __syncthreads();
      gemm_nn(L3, L2, n4, ty, tx, accp);
    }
    {
      // G_bar = 2 SG Sigma_{t+1} + A1 Sigma_{t+1} + m_bar m_{t+1}' + sum_s x_bar_s x_{t+1,s}'
      const double* mn = a.E_node_x + (bt + 1) * n;
      double* gb = a.g_bar + ((long)b * (T - 1) + t) * n * n;
      if (4 * ty < n4 && 4 * tx < n4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * ty + i, c = 4 * tx + j;
            if (r < n && c < n) {
              double v = 2.0 * acc.v[i][j] + accp.v[i][j] + mb[r] * mn[c];
              for (int s_ = 0; s_ < S; ++s_) v = __builtin_fma(xb[s_ * 64 + r], a.samples[((bt + 1) * a.S + s_) * n + c], v);
              gb[r * n + c] = v;
            }
          }
      }
    }
    // m_bar <- G' m_bar ; x_bar <- x_bar G
    matvec<true>(L0, mb, mnext, n);
    for (int e = threadIdx.x; e < S * n; e += 256) {
      const int s_ = e / n, j = e % n;
      double v = 0.0;
      for (int i = 0; i < n; ++i) v = __builtin_fma(xb[s_ * 64 + i], L0[i * TV_LD + j], v);
      xnext[s_ * 64 + j] = v;
    }
    __syncthreads();                                                             // L2 (Sigma_{t+1}) and mb / xb no longer read
    for (int i = threadIdx.x; i < n; i += 256) mb[i] = mnext[i];
    for (int e = threadIdx.x; e < S * n; e += 256) xb[(e / n) * 64 + e % n] = xnext[(e / n) * 64 + e % n];
    load_mat<true>(L2, h, NP, n, n4, 1.0);                                       // G_t'
    __syncthreads();
    acc_zero(acc);
    gemm_nn(L2, L1, n4, ty, tx, acc);                                           // G' SG
    if (a.g_E_pair) gemm_nn(L2, L3, n4, ty, tx, acc);                           // + G' A1 (symmetrised below)
    __syncthreads();
    store_block(L3, acc, ty, tx, n4, false);
    symmetrize_lds(L3, n4);
  }
  (void)tmpv; (void)xtmp;
}

// ---- phase 2 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_vjp_phase2(const TileVjpArgs a) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, n = a.n, n4 = (n + 3) & ~3, NP = a.NP, T = a.T;
  const int TV_LD = tv_ld(n4), TV_MAT = tv_mat(n4);
  double *L0 = sm, *L1 = sm + TV_MAT, *L2 = sm + 2 * TV_MAT, *L3 = sm + 3 * TV_MAT;
  double* vec = sm + 4 * TV_MAT;          // hb | cb | Pc | tmp
  double *hb = vec, *cb = vec + 64, *Pc = vec + 128, *tmpv = vec + 192;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const double gl = a.g_lognorm[b];
  for (int e = threadIdx.x; e < TV_MAT; e += 256) L3[e] = 0.0;                 // J_bar of step t + 1
  for (int e = threadIdx.x; e < 256; e += 256) vec[e] = 0.0;
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const long bt = (long)b * T + t;
    const double* h = handoff(a, b, t);
    const double* ct = h + 2L * NP * NP;
    load_mat<false>(L0, h + (long)NP * NP, NP, n, n4, 1.0);                      // Pinv_t
    for (int i = threadIdx.x; i < n; i += 256) cb[i] = a.c_bar[bt * n + i];
    Acc pbar;
    acc_zero(pbar);
    if (t < T - 1) {
      // R = -J12 (info form):  X_bar = -R J_bar - G_bar = J12 J_bar - G_bar ;  c_bar -= R h_bar = += J12 h_bar
      const double* J12 = a.J12 + (long)b * a.pair_seq_stride + (long)t * a.pair_t_stride;
      load_mat<false>(L1, J12, n, n, n4, 1.0);
      __syncthreads();
      matvec<false>(L1, hb, tmpv, n);
      for (int i = threadIdx.x; i < n; i += 256) cb[i] += tmpv[i];
      Acc acc;
      acc_zero(acc);
      gemm_nn(L1, L3, n4, ty, tx, acc);                                         // J12 J_bar
      const double* gb = a.g_bar + ((long)b * (T - 1) + t) * n * n;
      if (4 * ty < n4 && 4 * tx < n4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * ty + i, c = 4 * tx + j;
            acc.v[i][j] -= (r < n && c < n) ? gb[r * n + c] : 0.0;
          }
      }
      store_block(L2, acc, ty, tx, n4, false);                                  // X_bar
      __syncthreads();
      acc_zero(acc);
      gemm_nn(L0, L2, n4, ty, tx, acc);                                         // PX = Pinv X_bar
      __syncthreads();
      store_block(L1, acc, ty, tx, n4, false);
      load_mat<true>(L2, h, NP, n, n4, 1.0);                                     // G_t'
      __syncthreads();
      gemm_nn(L1, L2, n4, ty, tx, pbar);                                        // P_bar = PX G'
      __syncthreads();
    } else {
      __syncthreads();
    }
    load_mat<false>(L1, a.pinv_bar + bt * n * n, n, n, n4, 1.0);                // Pinv_bar (direct + Cholesky part)
    __syncthreads();
    {
      Acc acc;
      acc_zero(acc);
      gemm_nn(L0, L1, n4, ty, tx, acc);                                         // Pinv Pinv_bar
      store_block(L2, acc, ty, tx, n4, false);
      __syncthreads();
      acc_zero(acc);
      gemm_nn(L2, L0, n4, ty, tx, acc);                                         // (Pinv Pinv_bar) Pinv
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) pbar.v[i][j] -= acc.v[i][j];
    }
    matvec<false>(L0, cb, Pc, n);                                               // Pc = Pinv c_bar  (barrier inside)
    // P_bar -= Pc c' + gl/2 c c' + gl/2 Pinv
    if (4 * ty < n4 && 4 * tx < n4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * ty + i, c = 4 * tx + j;
          if (r < n && c < n) pbar.v[i][j] -= Pc[r] * ct[c] + 0.5 * gl * (ct[r] * ct[c] + L0[r * TV_LD + c]);
          else pbar.v[i][j] = 0.0;
        }
    }
    __syncthreads();
    store_block(L3, pbar, ty, tx, n4, false);
    symmetrize_lds(L3, n4);
    for (int i = threadIdx.x; i < n; i += 256) {
      const double hf = Pc[i] + gl * ct[i];
      a.g_node_J[bt * n + i] = -2.0 * L3[i * TV_LD + i];
      a.g_node_h[bt * n + i] = hf;
      hb[i] = hf;
    }
    __syncthreads();
  }
}

// ---- backward sampler recursion ----------------------------------------------------------------------------------
// x_t = c_t + noise_t + G_t x_{t+1}  (t = T-1 .. 0; natural_sample_backward, cython_lds_inference.pyx:310-355, with the
// noise chol(P_t)^-T eps_t precomputed for all (sequence, step) pairs: it does not depend on the recursion).
// One workgroup per sequence, G_t staged in LDS, S <= 16 samples.
__global__ __launch_bounds__(256) void tile_sample_kernel(int T, int n, int S, int NP, const double* ws,
                                                          const double* noise, double* samples) {
  extern __shared__ double sm[];
  double* Gs = sm;                       // n x (n + 1)
  double* xn = sm + 64 * 65;             // S x 64: x_{t+1}
  const int b = blockIdx.x, ld = n + 1;
  // G_t of the NEXT step travels through registers (16 entries per thread: rows ty + 16 i, columns 4 tx ..) while the
  // current step computes from LDS: the 32 KB tile comes from HBM, further away than one matrix-vector product
  const int ty = threadIdx.x >> 4, c0 = 4 * (threadIdx.x & 15);
  double pre[4][4];
  auto fetch = [&](int t) {
    const double* h = ws + ((long)b * T + t) * (2L * NP * NP + NP);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ty + 16 * i, cq = c0 + j;
        pre[i][j] = (r < n && cq < n) ? h[(long)r * NP + cq] : 0.0;
      }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ty + 16 * i, cq = c0 + j;
        if (r < n && cq < n) Gs[r * ld + cq] = pre[i][j];
      }
  };
  if (T > 1) fetch(T - 2);
  for (int t = T - 1; t >= 0; --t) {
    const double* ct = ws + ((long)b * T + t) * (2L * NP * NP + NP) + 2L * NP * NP;
    if (t < T - 1) {
      stage();                           // G_t, requested one step ago
      if (t > 0) fetch(t - 1);
    }
    __syncthreads();
    double out[4];
    int cnt = 0;
    for (int e = threadIdx.x; e < S * n; e += 256) {
      const int s_ = e / n, i = e % n;
      double v = ct[i] + noise[(((long)b * T + t) * S + s_) * n + i];
      if (t < T - 1)
        for (int j = 0; j < n; ++j) v = __builtin_fma(Gs[i * ld + j], xn[s_ * 64 + j], v);
      out[cnt++] = v;
    }
    __syncthreads();
    cnt = 0;
    for (int e = threadIdx.x; e < S * n; e += 256) {
      const int s_ = e / n, i = e % n;
      xn[s_ * 64 + i] = out[cnt];
      samples[(((long)b * T + t) * S + s_) * n + i] = out[cnt++];
    }
    __syncthreads();
  }
}

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

template <int NC>
struct CholCfg {
  static constexpr int LD = NC + 1;
  static constexpr int PAN = NC * LD;
  static constexpr int LDS_DOUBLES = 2 * PAN + 2 * NC + TV_MAX_S * NC;     // M panel | transposition panel | pivot column x2 | u_s
};

template <int NC, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void tile_chol_kernel(int n, int S, int NP, const double* ws, const double* eps,
                                                       const double* xbar, double* noise, double* pinv_bar,
                                                       int32_t* info) {
  using Cfg = CholCfg<NC>;
  constexpr int LD = Cfg::LD;
  extern __shared__ double sm[];
  double* panM = sm;                       // M (upper), row-major
  double* pan2 = sm + Cfg::PAN;            // transpositions
  double* pcol = pan2 + Cfg::PAN;          // scaled pivot column, double-buffered
  double* us = pcol + 2 * NC;              // MODE 1: u_s = M' xbar_s
  const long bt = blockIdx.x;
  const int c = threadIdx.x;               // lane = column
  const bool on = c < NC;
  const double* P = ws + bt * (2L * NP * NP + NP) + (long)NP * NP;
  double A[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) A[i] = (on && i < n && c < n) ? P[(long)i * NP + (c < n ? c : 0)] : ((i == c) ? 1.0 : 0.0);
  bool bad = false;
  // UL Cholesky, pivots NC-1 .. 0: lane j scales its column and publishes it; lanes k < j subtract M[i][j] M[k][j]
#pragma unroll
  for (int j = NC - 1; j >= 0; --j) {
    double* pc = pcol + (j & 1) * NC;
    if (c == j) {
      const double d = A[j];
      bad = bad || !(d > 0.0);
      const double r = 1.0 / sqrt(d);
#pragma unroll
      for (int i = 0; i <= j; ++i) { A[i] *= r; pc[i] = A[i]; }
    }
    __syncthreads();
    const double mkj = (c < j) ? pc[c < j ? c : 0] : 0.0;
#pragma unroll
    for (int i = 0; i < j; ++i) A[i] = __builtin_fma(-pc[i], mkj, A[i]);      // (entries below the diagonal: unused)
  }
  if (__ballot(bad) != 0 && c == 0) atomicMax(info, 1);
  // M as a row-major panel for the wave-uniform reads below
  if (on) {
#pragma unroll
    for (int i = 0; i < NC; ++i) panM[i * LD + c] = (i <= c) ? A[i] : 0.0;
  }
  __syncthreads();
  if constexpr (MODE == 0) {
    // noise[s][i] = sum_{j >= i} M[i][j] eps[s][j]   (lane = row i)
    const int i = c;
    if (i < n)
      for (int s_ = 0; s_ < S; ++s_) {
        const double* ep = eps + (bt * S + s_) * n;
        double v = 0.0;
        for (int j = i; j < n; ++j) v = __builtin_fma(panM[i * LD + j], ep[j], v);
        noise[(bt * S + s_) * n + i] = v;
      }
  } else {
    // u_s = M' xbar_s (own column), then Phi(M' Mbar)[i][c] = sum_s u_s[i] eps_s[c] for i <= c (diagonal halved):
    // Mbar = triu(sum_s xbar_s eps_s') is never formed (its mask k <= j is implied by k <= i <= j)
    for (int s_ = 0; s_ < S; ++s_) {
      const double* xb = xbar + (bt * S + s_) * n;
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < NC; ++k) v = __builtin_fma(A[k], (k <= c && k < n) ? xb[k < n ? k : 0] : 0.0, v);
      if (on) us[s_ * NC + c] = v;
    }
    __syncthreads();
    double K[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) K[i] = 0.0;
    for (int s_ = 0; s_ < S; ++s_) {
      const double ev = (c < n) ? eps[(bt * S + s_) * n + c] : 0.0;
#pragma unroll
      for (int i = 0; i < NC; ++i) K[i] = __builtin_fma(us[s_ * NC + i], ev, K[i]);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) K[i] = (i < c) ? K[i] : ((i == c) ? 0.5 * K[i] : 0.0);
    // two passes of:  solve M' X = K column-wise (x_i = K_i / M[i][i];  K_r -= M[i][r] x_i for r > i), then
    // transpose through LDS:  Q1 = M^-T Phi;  Q' = M^-T Q1'  (Q = Q1 M^-1).  (Written as a loop of two trips so
    // that the unrolled body exists once and K stays in registers: as lambdas the array went to scratch memory.)
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const double x = K[i] * rdiag(panM[i * LD + i]);
        K[i] = x;
#pragma unroll
        for (int r = i + 1; r < NC; ++r) K[r] = __builtin_fma(-panM[i * LD + r], x, K[r]);
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

extern "C" size_t svae_lds_tile_vjp_workspace_doubles(int B, int T, int n, int S) {
  if (B <= 0 || T <= 0 || n <= 0 || n > 64 || S < 0) return 0;
  return (size_t)B * T * n * n * 2 + (size_t)B * (T > 1 ? T - 1 : 0) * n * n + (size_t)B * T * n + (size_t)B * T * (S > 0 ? S : 0) * n;
}

// phase 0, 1, 2 as described at the top; `workspace` holds [sig | pinv_bar | g_bar | c_bar | xbar] in that
// order (svae_lds_tile_vjp_workspace_doubles); between phase 1 and phase 2 the caller adds the Cholesky adjoint
// of the noise factor into pinv_bar (it is batched over all (sequence, step) pairs and needs xbar).
extern "C" int svae_lds_tile_vjp_f64(int phase, int B, int T, int n, int S, int inhomog, int pair_batched,
                                     const double* J12, const double* g_lognorm, const double* g_E_node_diagxx,
                                     const double* g_E_node_x, const double* g_E_init, const double* g_E_pair,
                                     const double* g_samples,
                                     const double* samples, const double* E_node_x, double* g_node_J, double* g_node_h,
                                     const void* handoff_workspace, void* workspace, size_t ws_doubles, void* stream) {
  if (phase < 0 || phase > 2) return -1;
  if (B < 0) return -2;
  if (T < 1) return -3;
  if (n < 1 || n > 64) return -4;
  if (g_samples && (S < 1 || S > svae::TV_MAX_S)) return -5;
  if (pair_batched && !inhomog) return -6;
  if (T > 1 && !J12) return -8;
  if (!g_lognorm) return -9;
  if (g_E_pair && !inhomog) return -13;          /* per-step pair statistics only (as svae_lds_estep_vjp_ex_f64) */
  if (g_samples && !samples) return -14;
  if (!E_node_x) return -15;
  if (!g_node_J || !g_node_h) return -16;
  if (!handoff_workspace) return -18;
  if (!workspace || ws_doubles < svae_lds_tile_vjp_workspace_doubles(B, T, n, g_samples ? S : 0)) return -19;
  if (B == 0) return 0;
  svae::TileVjpArgs a;
  a.B = B; a.T = T; a.n = n; a.S = g_samples ? S : 0; a.NP = 16 * ((n + 15) / 16);
  a.ws = (const double*)handoff_workspace;
  a.J12 = J12; a.pair_t_stride = inhomog ? (long)n * n : 0; a.pair_seq_stride = pair_batched ? (long)(T - 1) * n * n : 0;
  a.E_node_x = E_node_x; a.samples = samples; a.g_samples = g_samples; a.g_lognorm = g_lognorm;
  a.g_dxx = g_E_node_diagxx; a.g_x = g_E_node_x; a.g_E_init = g_E_init; a.g_E_pair = T > 1 ? g_E_pair : nullptr;
  double* w = (double*)workspace;
  a.sig = w; w += (size_t)B * T * n * n;
  a.pinv_bar = w; w += (size_t)B * T * n * n;
  a.g_bar = w; w += (size_t)B * (T > 1 ? T - 1 : 0) * n * n;
  a.c_bar = w; w += (size_t)B * T * n;
  a.xbar = w;
  a.g_node_J = g_node_J; a.g_node_h = g_node_h;
  hipStream_t s = (hipStream_t)stream;
  const int n4 = (n + 3) & ~3;
  const size_t lds0 = (size_t)3 * svae::tv_mat(n4) * sizeof(double);
  const size_t lds12 = (size_t)(4 * svae::tv_mat(n4) + 192 + 3 * svae::TV_MAX_S * 64) * sizeof(double);
  const size_t lds0_max = (size_t)3 * svae::tv_mat(64) * sizeof(double);
  const size_t lds12_max = (size_t)(4 * svae::tv_mat(64) + 192 + 3 * svae::TV_MAX_S * 64) * sizeof(double);
  static svae::LdsGrant grant0, grant1, grant2;
  if (!grant0.ensure(reinterpret_cast<const void*>(svae::tile_vjp_phase0), (long)lds0_max) ||
      !grant1.ensure(reinterpret_cast<const void*>(svae::tile_vjp_phase1), (long)lds12_max) ||
      !grant2.ensure(reinterpret_cast<const void*>(svae::tile_vjp_phase2), (long)lds12_max))
    return -1001;
  if (phase == 0) hipLaunchKernelGGL(svae::tile_vjp_phase0, dim3(B), dim3(256), lds0, s, a);
  else if (phase == 1) hipLaunchKernelGGL(svae::tile_vjp_phase1, dim3(B), dim3(256), lds12, s, a);
  else hipLaunchKernelGGL(svae::tile_vjp_phase2, dim3(B), dim3(256), lds12, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// Backward sampling for latent dimension 16 <= n <= 64 from the hand-off of the tiled E-step: `noise` (B,T,S,n) =
// chol(P_t)^-T eps_t (the caller's batched factorisation of P_t^-1), samples (B,T,S,n) out.
extern "C" int svae_lds_tile_sample_f64(int B, int T, int n, int S, const double* noise, double* samples,
                                        const void* handoff_workspace, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (n < 1 || n > 64) return -3;
  if (S < 1 || S > svae::TV_MAX_S) return -4;
  if (!noise) return -5;
  if (!samples) return -6;
  if (!handoff_workspace) return -7;
  if (B == 0) return 0;
  const size_t lds = (size_t)(64 * 65 + svae::TV_MAX_S * 64) * sizeof(double);
  hipLaunchKernelGGL(svae::tile_sample_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, T, n, S,
                     16 * ((n + 15) / 16), (const double*)handoff_workspace, noise, samples);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// The sampler's noise factor (mode 0: noise (B,T,S,n) = chol(P_t)^-T eps_t) and its adjoint (mode 1: adds the
// Cholesky-path cotangent into the pinv_bar section of the VJP workspace, between phases 1 and 2 of
// svae_lds_tile_vjp_f64), from the P_t^-1 of the tiled E-step's hand-off; one workgroup per (sequence, step).
extern "C" int svae_lds_tile_noise_f64(int mode, int B, int T, int n, int S, const double* eps, double* noise,
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
  if (B == 0) return 0;
  const int NP = 16 * ((n + 15) / 16);
  const long BT = (long)B * T;
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
    hipLaunchKernelGGL(kern, dim3((unsigned)BT), dim3(64), lds, st, n, S, NP, (const double*)handoff_workspace, eps, xbar,
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
