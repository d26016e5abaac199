// lds_vjp_tile.hip -- reverse-mode derivative of the LDS E-step (+ sampler) w.r.t. the node potentials for
// latent dimension 16 <= n <= 64 (any n <= 64 works), on the hand-off of the LDS-tiled E-step kernel
// (lds_estep_tile.hip): what natural_filter_grad / natural_smoother_general_grad / natural_sample_backward_grad
// compute in the reference (svae/lds/cython_lds_inference.pyx:92-145, 236-306, 357-409), as the adjoint of THIS
// library's recursion (derivation and torch restatement: svae_amd/lds/lds_large.py, vjp_from_handoff).
//
// Three passes over time, each one workgroup (four wavefronts) per sequence with the n x n state in LDS:
//   phase 0 (t = T-1 .. 0)  Sigma_t = Pinv_t + G_t Sigma_{t+1} G_t'                      -> sig (B,T,n,n)
//   phase 1 (t = 0 .. T-1)  adjoint of the smoother / sampler recursions: Sigma_bar, m_bar, x_bar
//                           -> pinv_bar (B,T,n,n), g_bar (B,T-1,n,n), c_bar (B,T,n), xbar (B,T,S,n)
//   [the caller adds the Cholesky adjoint of the noise factor, batched over all (b,t), into pinv_bar]
//   phase 2 (t = T-1 .. 0)  adjoint of the filter -> g_node_J, g_node_h (B,T,n)
// Every O(n^3) product is a list of 16 x 16 tile products on v_mfma_f64_16x16x4 with fragments read straight from
// LDS-resident operands (row stride NP + 2 doubles): the NB x NB output tiles are dealt round-robin to the four
// wavefronts; the fragment addressing makes a transposed operand free (no transposed copies), and a product reads
// 4x fewer LDS bytes than the register-blocked VALU form it replaces (round 2: 4 x 4 blocks per thread -- LDS-bound
// at ~7 us per 64^3 product against 2 us of arithmetic).  The global operands of a step (G_t, Sigma_{t+1}, P_t^-1,
// Pinv_bar_t, J12, the per-step vectors) are requested ahead of their use into registers -- one step ahead for what a
// step starts with, at the top of the step for what it needs two products later; see the note on the memory schedule
// below for what keeps such requests in flight.  The matrix-vector products of a step run on all four wavefronts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "per_device.hpp"

namespace svae {

// LDS matrices: NP x (NP + 2) doubles, NP = n rounded up to 16 (33.8 KB at n = 64, 8.7 KB at n = 32): sized by the
// instantiation (NB = NP / 16), so that smaller latent dimensions fit several workgroups per CU
__host__ __device__ constexpr int tv_mat(int np) { return np * (np + 2); }
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
  int t_begin, t_end;                     // phase 2: the steps of this launch, t_end - 1 .. t_begin
  double* p2state;                        // phase 2: per sequence [J_bar (NP x (NP + 2)) | h_bar (64)] between two ranges
  double* g_node_J; double* g_node_h;
};

typedef double d4 __attribute__((ext_vector_type(4)));

// D = A(16x16) * B(16x16) + C as four v_mfma_f64_16x16x4; a[kb] / b[kb] = k-chunk kb of the fragments
__device__ __forceinline__ d4 mma16(const d4 a, const d4 b, d4 c) {
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c, 0, 0, 0);
  return c;
}
// Fragment addressing (lane = 16 kq + r16) for a row-major tile Tl at (row0, col0) of an LDS matrix:
//   frag_a: A operand of Tl (lane holds Tl[r16][4 kb + kq])  == B operand of Tl'
//   frag_b: B operand of Tl (lane holds Tl[4 kb + kq][r16])  == A operand of Tl'  == the C/D layout
// so a transposed operand costs nothing: no transposed copies are ever made.
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

// LDS matrices of the MFMA phases: NP x NP (NP = 16 NB), row stride NP + 2 (== 2 mod 32 doubles: the fragment reads
// are bank-conflict free), zero-padded beyond n.  The NB x NB output tiles of a product are dealt round-robin to the
// four wavefronts (tile q = wave + 4 j -> (q / NB, q % NB)): every product and every element-wise epilogue of a phase
// uses the same ownership, so accumulators of different products add up lane by lane.
template <int NB> constexpr int tv_maxt() { return (NB * NB + 3) / 4; }

template <int NB>
__device__ __forceinline__ void acc_zero(d4 (&acc)[tv_maxt<NB>()]) {
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
}

// acc += op(A) op(B) for this wavefront's tiles (TA / TB: the operand is the TRANSPOSE of the LDS matrix)
template <int NB, bool TA, bool TB>
__device__ __forceinline__ void gemm_mfma(const double* A, const double* Bm, int wave, int r16, int kq,
                                          d4 (&acc)[tv_maxt<NB>()]) {
  constexpr int LD = 16 * NB + 2;
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j) {
    const int q = wave + 4 * j;
    if (q < NB * NB) {
      const int I = q / NB, J = q % NB;
#pragma unroll
      for (int K = 0; K < NB; ++K) {
        const d4 a = TA ? frag_b(A, LD, 16 * K, 16 * I, r16, kq) : frag_a(A, LD, 16 * I, 16 * K, r16, kq);
        const d4 b = TB ? frag_a(Bm, LD, 16 * J, 16 * K, r16, kq) : frag_b(Bm, LD, 16 * K, 16 * J, r16, kq);
        acc[j] = mma16(a, b, acc[j]);
      }
    }
  }
}

// f(j, i, row, col, value&) over the accumulator entries this lane owns
template <int NB, class F>
__device__ __forceinline__ void for_owned(d4 (&acc)[tv_maxt<NB>()], int wave, int r16, int kq, F&& f) {
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j) {
    const int q = wave + 4 * j;
    if (q < NB * NB) {
      const int I = q / NB, J = q % NB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double v = acc[j][i];
        f(16 * I + kq + 4 * i, 16 * J + r16, v);
        acc[j][i] = v;
      }
    }
  }
}

// Memory schedule of the phases: every global operand of a step is REQUESTED one step ahead and first touched where it
// is used.  Two things would undo that and are avoided throughout: a load inside a conditional block (hipcc waits for
// it at the end of the block -- so the requests are unconditional, with clamped indices, and the zero padding /
// validity mask is applied where the value is consumed), and __syncthreads() (its fences wait for ALL outstanding
// memory operations: the workgroup only ever exchanges data through LDS, so the barriers here wait for LDS only).
__device__ __forceinline__ void tile_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// this lane's C-layout entries of a global n x n matrix (row stride gld): raw (clamped addresses), requested a step
// ahead like the LDS operands; mask_c zeroes the entries outside n x n at the point of use
// (the element offsets are computed ONCE per kernel -- COff / MOff below: recomputed per request, with their clamps
//  and 64-bit multiplies, the ~35 requests of a step cost ~9 instructions each: 2.9 k cycles in phase 2)
template <int NB> struct COff { int o[tv_maxt<NB>()][4]; };
template <int NB>
__device__ __forceinline__ COff<NB> c_off(int gld, int n, int wave, int r16, int kq) {
  COff<NB> f;
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j) {
    const int q = wave + 4 * j < NB * NB ? wave + 4 * j : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * (q / NB) + kq + 4 * i, c = 16 * (q % NB) + r16;
      const bool ok = r < n && c < n;
      f.o[j][i] = (ok ? r : 0) * gld + (ok ? c : 0);
    }
  }
  return f;
}
template <int NB>
__device__ __forceinline__ void fetch_c(d4 (&v)[tv_maxt<NB>()], const double* src, const COff<NB>& f) {
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[j][i] = src[f.o[j][i]];
}
template <int NB>
__device__ __forceinline__ void mask_c(d4 (&v)[tv_maxt<NB>()], int n, int wave, int r16, int kq) {
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j) {
    const int q = wave + 4 * j;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * (q / NB) + kq + 4 * i, c = 16 * (q % NB) + r16;
      v[j][i] = (q < NB * NB && r < n && c < n) ? v[j][i] : 0.0;
    }
  }
}

template <int NB>
__device__ __forceinline__ void store_acc(double* M, d4 (&acc)[tv_maxt<NB>()], int wave, int r16, int kq) {
  constexpr int LD = 16 * NB + 2;
#pragma unroll
  for (int j = 0; j < tv_maxt<NB>(); ++j) {
    const int q = wave + 4 * j;
    if (q < NB * NB) store_c(M, LD, 16 * (q / NB), 16 * (q % NB), r16, kq, acc[j]);
  }
}

// A global matrix (row stride gld, n x n valid) travels through 16 registers per thread: fetch (global -> registers,
// issued a step ahead of its use: the operands come from HBM, ~2 us away) and stage (registers -> LDS, zero-padded to
// NP x NP, scaled).  Thread (ty, tx) holds columns 4 tx .. 4 tx + 3 of rows ty, ty + 16, ty + 32, ty + 48.
// (fetch: raw values from clamped addresses, no branch; stage: zero outside n x n)
// VEC: the row stride is a multiple of 4 doubles and the matrix starts 32-byte aligned (the NP-padded hand-off always;
// compact n x n arrays when n is a multiple of 4: the kernels' A4 instances) -- one 32-byte request per row instead of
// four 8-byte ones whose lanes sit 32 bytes apart: the scalar form touches every cache line four times, and the
// requests of a step (~50 per thread) were bound by the texture addresser, not by the data (phase 2: 5.2 k -> 4.4 k
// cycles for the section that issues them, tools/tile_vjp_timing.py).
struct MatRegs { double v[4][4]; };
template <bool VEC> struct MOff { int o[VEC ? 4 : 16]; };
template <bool VEC>
__device__ __forceinline__ MOff<VEC> mat_off(int gld, int n) {
  MOff<VEC> f;
  const int ty = threadIdx.x >> 4, c0 = 4 * (threadIdx.x & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i < n ? ty + 16 * i : 0;
    if constexpr (VEC) {
      f.o[i] = r * gld + (c0 < gld ? c0 : 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.o[4 * i + j] = r * gld + (c0 + j < n ? c0 + j : 0);
    }
  }
  return f;
}
template <bool VEC>
__device__ __forceinline__ void fetch_mat(MatRegs& m, const double* src, const MOff<VEC>& f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (VEC) {
      const d4 q = *(const d4*)(src + f.o[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) m.v[i][j] = q[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) m.v[i][j] = src[f.o[4 * i + j]];
    }
  }
}
template <int NB>
__device__ __forceinline__ void stage_mat(double* dst, const MatRegs& m, int n, double scale = 1.0) {
  constexpr int NPL = 16 * NB, LD = NPL + 2;
  const int ty = threadIdx.x >> 4, c0 = 4 * (threadIdx.x & 15);
  if (c0 >= NPL) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i;
    if (r < NPL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[r * LD + c0 + j] = (r < n && c0 + j < n) ? scale * m.v[i][j] : 0.0;
    }
  }
}
template <int NB, bool VEC>
__device__ __forceinline__ void load_mat(double* dst, const double* src, int gld, int n) {
  MatRegs m;
  fetch_mat<VEC>(m, src, mat_off<VEC>(gld, n));
  stage_mat<NB>(dst, m, n);
}

// LDS matrix -> global n x n (dense, row stride n), same thread mapping (VEC: n a multiple of 4 -> 32-byte stores)
template <int NB, bool VEC>
__device__ __forceinline__ void store_mat(double* dst, const double* src, int n) {
  constexpr int LD = 16 * NB + 2;
  const int ty = threadIdx.x >> 4, c0 = 4 * (threadIdx.x & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i;
    if constexpr (VEC) {
      if (r < n && c0 < n) {
        const double* q = src + r * LD + c0;
        *(d4*)(dst + (long)r * n + c0) = d4{q[0], q[1], q[2], q[3]};
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (r < n && c0 + j < n) dst[(long)r * n + c0 + j] = src[r * LD + c0 + j];
    }
  }
}

// M <- (A + A') / 2 for the product A held in the wavefronts' accumulators (C layout): A goes to LDS, every lane reads
// the mirror images of its own 16 entries, averages in registers, and the result goes back -- three barriers, 16 reads
// (symmetrize_lds below: four barriers with the store in front of it, 32 reads)
template <int NB>
__device__ __forceinline__ void symmetrize_acc(double* M, d4 (&acc)[tv_maxt<NB>()], int wave, int r16, int kq) {
  constexpr int LD = 16 * NB + 2;
  store_acc<NB>(M, acc, wave, r16, kq);
  tile_barrier();
  for_owned<NB>(acc, wave, r16, kq, [&](int r, int c, double& v) { v = 0.5 * (v + M[c * LD + r]); });
  tile_barrier();
  store_acc<NB>(M, acc, wave, r16, kq);
  tile_barrier();
}

// in place: M <- (M + M') / 2 on the NP x NP LDS matrix (barriers inside); each thread handles a 4 x 4 block
template <int NB>
__device__ __forceinline__ void symmetrize_lds(double* M) {
  constexpr int NPL = 16 * NB, LD = NPL + 2;
  const int r0 = 4 * (threadIdx.x >> 4), c0 = 4 * (threadIdx.x & 15);
  const bool on = r0 < NPL && c0 < NPL;
  tile_barrier();
  double keep[4][4];
  if (on) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[i][j] = 0.5 * (M[(r0 + i) * LD + c0 + j] + M[(c0 + j) * LD + r0 + i]);
  }
  tile_barrier();
  if (on) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) M[(r0 + i) * LD + c0 + j] = keep[i][j];
  }
  tile_barrier();
}

// y[i] = sum_j A[i][j] x[j]  (TRANS: A[j][i]) for i < n <= 64; A in LDS (row stride LD); x, y LDS vectors (y != x);
// `part`: 256 doubles of LDS scratch.  All four wavefronts: wavefront w sums the columns j = w, w + 4, .. of every row
// (lane = row; four independent accumulators so that the LDS reads overlap), the partial sums meet in `part`.  Two
// barriers inside, the second after y is complete.  (One thread per row and a serial 64-term loop took ~3 us a call.)
template <bool TRANS>
__device__ __forceinline__ void matvec(const double* A, int LD, const double* x, double* y, int n, double* part) {
  const int w = threadIdx.x >> 6, i = threadIdx.x & 63, ii = i < n ? i : 0;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int j = w + 4 * u, jj = j < n ? j : 0;
    const double av = TRANS ? A[jj * LD + ii] : A[ii * LD + jj];
    s[u & 3] = __builtin_fma(j < n ? av : 0.0, x[jj], s[u & 3]);
  }
  part[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
  tile_barrier();
  if (threadIdx.x < n) y[threadIdx.x] = (part[i] + part[64 + i]) + (part[128 + i] + part[192 + i]);
  tile_barrier();
}

__device__ __forceinline__ const double* handoff(const TileVjpArgs& a, int b, int t) {
  return a.ws + ((long)b * a.T + t) * (2L * a.NP * a.NP + a.NP);
}

// ---- phase 0 ------------------------------------------------------------------------------------------------
template <int NB, bool A4>
__global__ __launch_bounds__(256) void tile_vjp_phase0(const TileVjpArgs a) {
  constexpr int NPL = 16 * NB, LD = NPL + 2, MAT = NPL * LD;
  extern __shared__ double sm[];
  const int b = blockIdx.x, n = a.n, NP = a.NP, T = a.T;
  double *L0 = sm, *L1 = sm + MAT, *L3 = sm + 2 * MAT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
  for (int e = threadIdx.x; e < 3 * MAT; e += 256) sm[e] = 0.0;
  tile_barrier();
  load_mat<NB, true>(L3, handoff(a, b, T - 1) + (long)NP * NP, NP, n);
  MatRegs gpre;
  d4 ppre[tv_maxt<NB>()];
  const MOff<true> offNP = mat_off<true>(NP, n);
  const COff<NB> coffNP = c_off<NB>(NP, n, wave, r16, kq);
  {
    const int tp = T > 1 ? T - 2 : 0;
    fetch_mat<true>(gpre, handoff(a, b, tp), offNP);
    fetch_c<NB>(ppre, handoff(a, b, tp) + (long)NP * NP, coffNP);
  }
  tile_barrier();
  for (int t = T - 1; t >= 0; --t) {
    if (t < T - 1) {
      stage_mat<NB>(L0, gpre, n);                                    // G_t, requested one step ago
      d4 pcur[tv_maxt<NB>()];
#pragma unroll
      for (int j = 0; j < tv_maxt<NB>(); ++j) pcur[j] = ppre[j];  // P_t^-1 in the C layout
      mask_c<NB>(pcur, n, wave, r16, kq);
      {
        const int tp = t > 0 ? t - 1 : 0;                         // (unconditional: see the note on the memory schedule)
        fetch_mat<true>(gpre, handoff(a, b, tp), offNP);
        fetch_c<NB>(ppre, handoff(a, b, tp) + (long)NP * NP, coffNP);
      }
      tile_barrier();
      d4 acc[tv_maxt<NB>()];
      acc_zero<NB>(acc);
      gemm_mfma<NB, false, false>(L0, L3, wave, r16, kq, acc);    // G Sigma
      store_acc<NB>(L1, acc, wave, r16, kq);
      tile_barrier();                                            // (everyone is done reading L3, too)
      acc_zero<NB>(acc);
      gemm_mfma<NB, false, true>(L1, L0, wave, r16, kq, acc);     // (G Sigma) G'
#pragma unroll
      for (int j = 0; j < tv_maxt<NB>(); ++j) acc[j] += pcur[j];
      store_acc<NB>(L3, acc, wave, r16, kq);
      // (P^-1 + (G Sigma) G' is symmetric by construction: nothing to symmetrise but rounding, 1e-16 per step)
      tile_barrier();
    }
    store_mat<NB, A4>(a.sig + ((long)b * T + t) * n * n, L3, n);
  }
}

// per-section cycle counters of a timing build (-DSVAE_TV_TIMING, tools/tile_vjp_timing.py): phase 1 leaves them in
// c_bar[b, 0, :16], phase 2 in g_node_h[b, 0, :16] (overwriting results)
#ifdef SVAE_TV_TIMING
#define TV_TICK_INIT long long tm[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tlast_ = __builtin_readcyclecounter();
#define TV_TICK(i) { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tlast_; tlast_ = now_; }
#else
#define TV_TICK_INIT
#define TV_TICK(i)
#endif

// ---- phase 1 ------------------------------------------------------------------------------------------------
template <int NB, bool A4>
__global__ __launch_bounds__(256) void tile_vjp_phase1(const TileVjpArgs a) {
  constexpr int NPL = 16 * NB, LD = NPL + 2, MAT = NPL * LD;
  extern __shared__ double sm[];
  const int b = blockIdx.x, n = a.n, NP = a.NP, T = a.T, S = a.g_samples ? a.S : 0;
  double *L0 = sm, *L1 = sm + MAT, *L2 = sm + 2 * MAT, *L3 = sm + 3 * MAT;
  double* vec = sm + 4 * MAT;             // mb (64) | tmp (64) | mnext (64) | xb (S x 64) | xtmp (S x 64) | xnext (S x 64)
  double *mb = vec, *mnext = vec + 128, *xb = vec + 192, *xnext = xb + 2 * TV_MAX_S * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
  for (int e = threadIdx.x; e < 4 * MAT + 192 + 3 * TV_MAX_S * 64; e += 256) sm[e] = 0.0;     // Sigma_bar = 0, vectors = 0
  MatRegs gpre, spre;
  const MOff<true> offNP = mat_off<true>(NP, n);
  const MOff<A4> offN = mat_off<A4>(n, n);
  fetch_mat<true>(gpre, handoff(a, b, 0), offNP);
  fetch_mat<A4>(spre, a.sig + ((long)b * T + (T > 1 ? 1 : 0)) * n * n, offN);
  // direct cotangents and the mean of step t for thread i < n, requested one step ahead
  const int ti = threadIdx.x < n ? threadIdx.x : 0;
  double gdpre = 0.0, gxpre = 0.0, mtpre = 0.0;
  const double* gd_src = a.g_dxx ? a.g_dxx : a.E_node_x;      // (absent cotangents: a valid address, the value is dropped)
  const double* gx_src = a.g_x ? a.g_x : a.E_node_x;
  const double gd_on = a.g_dxx ? 1.0 : 0.0, gx_on = a.g_x ? 1.0 : 0.0;
  auto fetch_vec = [&](int t) {
    const long o = ((long)b * T + t) * n + ti;
    gdpre = gd_src[o];
    gxpre = gx_src[o];
    mtpre = a.E_node_x[o];
  };
  fetch_vec(0);
  // per-sample vectors, elements e = tid + 256 k: the sample cotangents of the NEXT step and the samples x_{t+1} the
  // epilogue of THIS step multiplies with (staged in LDS next to m_{t+1}: read from global memory inside the epilogue
  // they cost a round trip each)
  constexpr int XSL_S = 12;               // samples that fit the LDS staging area (more: read in place)
  double* mnl = vec + 64;                 // m_{t+1}
  double* xsl = xb + TV_MAX_S * 64 + 256; // x_{t+1,s}, s < XSL_S (behind the matvec scratch)
  const int SN = S * n;
  int eo[4], el[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = threadIdx.x + 256 * k, ee = e < SN ? e : 0;
    eo[k] = ee;                            // offset inside a step's (S, n) block
    el[k] = n > 0 ? (ee / n) * 64 + ee % n : 0;
  }
  const double* gs_src = S ? a.g_samples : a.E_node_x;
  const double* xs_src = S ? a.samples : a.E_node_x;
  double gspre[4], xspre[4];
  auto fetch_smp = [&](int tg, int tx) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gspre[k] = gs_src[((long)b * T + tg) * SN + eo[k]];
      xspre[k] = xs_src[((long)b * T + tx) * SN + eo[k]];
    }
  };
  fetch_smp(0, T > 1 ? 1 : 0);
  tile_barrier();
  TV_TICK_INIT
  for (int t = 0; t < T; ++t) {
    const long bt = (long)b * T + t;
    const double* mt = a.E_node_x + bt * n;
    double gscur[4], xscur[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { gscur[k] = gspre[k]; xscur[k] = xspre[k]; }
    // direct cotangents of step t
    if (threadIdx.x < n) {
      const int i = threadIdx.x;
      const double gd = gd_on * gdpre, gx = gx_on * gxpre;
      L3[i * LD + i] += gd;
      mb[i] += gx + 2.0 * gd * mtpre;
    }
    fetch_vec(t + 1 < T ? t + 1 : t);
    fetch_smp(t + 1 < T ? t + 1 : t, t + 2 < T ? t + 2 : T - 1);
    if (t == 0 && a.g_E_init) {
      tile_barrier();
      const double* gi = a.g_E_init + (long)b * (n * n + n);
      for (int e = threadIdx.x; e < n * n; e += 256) {
        const int r = e / n, c = e % n;
        L3[r * LD + c] += 0.5 * (gi[r * n + c] + gi[c * n + r]);
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
      tile_barrier();
      const double* A = t < T - 1 ? a.g_E_pair + ((long)b * (T - 1) + t) * 3 * n * n : nullptr;
      const double* Ap = t > 0 ? a.g_E_pair + ((long)b * (T - 1) + t - 1) * 3 * n * n : nullptr;
      for (int e = threadIdx.x; e < n * n; e += 256) {
        const int r = e / n, c = e % n;
        double v = 0.0;
        if (A) v += 0.5 * (A[r * n + c] + A[c * n + r]);
        if (Ap) v += 0.5 * (Ap[2 * n * n + r * n + c] + Ap[2 * n * n + c * n + r]);
        L3[r * LD + c] += v;
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
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (threadIdx.x + 256 * k < SN) xb[el[k]] += gscur[k];
    tile_barrier();
    TV_TICK(0)
    // records for phase 2
    store_mat<NB, A4>(a.pinv_bar + bt * n * n, L3, n);
    for (int i = threadIdx.x; i < n; i += 256) {
      double s = mb[i];
      for (int s_ = 0; s_ < S; ++s_) s += xb[s_ * 64 + i];
      a.c_bar[bt * n + i] = s;
    }
    for (int e = threadIdx.x; e < S * n; e += 256) a.xbar[(bt * a.S + e / n) * n + e % n] = xb[(e / n) * 64 + e % n];
    TV_TICK(1)
    if (t == T - 1) break;
    // propagate to t + 1
    if (threadIdx.x < n) mnl[threadIdx.x] = mtpre;                                  // m_{t+1} (requested at the top of the step)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (threadIdx.x + 256 * k < SN && el[k] < XSL_S * 64) xsl[el[k]] = xscur[k];
    stage_mat<NB>(L0, gpre, n);                                                     // G_t
    stage_mat<NB>(L2, spre, n);                                                     // Sigma_{t+1}
    tile_barrier();
    TV_TICK(2)
    // next step's matrices: each requested in FRONT of a product (a burst of requests stalls at the issue, phase 2)
    const int tq = t + 2 < T ? t + 1 : t;                                        // (unconditional, clamped)
    fetch_mat<true>(gpre, handoff(a, b, tq), offNP);
    d4 acc[tv_maxt<NB>()], accp[tv_maxt<NB>()];
    acc_zero<NB>(acc);
    gemm_mfma<NB, false, false>(L3, L0, wave, r16, kq, acc);                     // SG = Sigma_bar G
    store_acc<NB>(L1, acc, wave, r16, kq);
    tile_barrier();                                                             // (Sigma_bar in L3 is consumed)
    TV_TICK(3)
    fetch_mat<A4>(spre, a.sig + ((long)b * T + tq + 1) * n * n, offN);
    acc_zero<NB>(acc);
    gemm_mfma<NB, false, false>(L1, L2, wave, r16, kq, acc);                     // SG Sigma_{t+1}
    TV_TICK(4)
    acc_zero<NB>(accp);
    if (a.g_E_pair) {
      // E x_t x_{t+1}' = G_t Sigma_{t+1} + m_t m_{t+1}':  G_bar += A1 Sigma_{t+1};  Sigma_bar_{t+1} += sym(G_t' A1).
      load_mat<NB, A4>(L3, a.g_E_pair + (((long)b * (T - 1) + t) * 3 + 1) * n * n, n, n);
      tile_barrier();
      gemm_mfma<NB, false, false>(L3, L2, wave, r16, kq, accp);
    }
    {
      // G_bar = 2 SG Sigma_{t+1} + A1 Sigma_{t+1} + m_bar m_{t+1}' + sum_s x_bar_s x_{t+1,s}'
      double* gb = a.g_bar + ((long)b * (T - 1) + t) * n * n;
#pragma unroll
      for (int j = 0; j < tv_maxt<NB>(); ++j) {
        const int q = wave + 4 * j;
        if (q < NB * NB) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 16 * (q / NB) + kq + 4 * i, c = 16 * (q % NB) + r16;
            if (r < n && c < n) {
              double v = 2.0 * acc[j][i] + accp[j][i] + mb[r] * mnl[c];
              for (int s_ = 0; s_ < S; ++s_)
                v = __builtin_fma(xb[s_ * 64 + r], s_ < XSL_S ? xsl[s_ * 64 + c] : a.samples[((bt + 1) * a.S + s_) * n + c], v);
              gb[r * n + c] = v;
            }
          }
        }
      }
    }
    TV_TICK(5)
    // m_bar <- G' m_bar ; x_bar <- x_bar G
    matvec<true>(L0, LD, mb, mnext, n, xb + TV_MAX_S * 64);     // (scratch: the unused middle third of the sample area)
    for (int s_ = 0; s_ < S; ++s_) matvec<true>(L0, LD, xb + s_ * 64, xnext + s_ * 64, n, xb + TV_MAX_S * 64);
    // (the barrier that ends the last product: mb / xb are no longer read)
    for (int i = threadIdx.x; i < n; i += 256) mb[i] = mnext[i];
    for (int e = threadIdx.x; e < S * n; e += 256) xb[(e / n) * 64 + e % n] = xnext[(e / n) * 64 + e % n];
    TV_TICK(6)
    acc_zero<NB>(acc);
    gemm_mfma<NB, true, false>(L0, L1, wave, r16, kq, acc);                      // G' SG
    if (a.g_E_pair) gemm_mfma<NB, true, false>(L0, L3, wave, r16, kq, acc);      // + G' A1 (symmetrised below)
    tile_barrier();
    TV_TICK(7)
    store_acc<NB>(L3, acc, wave, r16, kq);
    // G' (Sigma_bar G) is symmetric by construction; only the pair-statistic term G' A1 needs its symmetric part taken
    if (a.g_E_pair) symmetrize_lds<NB>(L3);
    else tile_barrier();
    TV_TICK(8)
  }
#ifdef SVAE_TV_TIMING
  if (threadIdx.x == 0) for (int q = 0; q < 16; ++q) a.c_bar[(long)b * T * n + q] = (double)tm[q];
#endif
}

// ---- phase 2 ------------------------------------------------------------------------------------------------
template <int NB, bool A4>
__global__ __launch_bounds__(256) void tile_vjp_phase2(const TileVjpArgs a) {
  constexpr int NPL = 16 * NB, LD = NPL + 2, MAT = NPL * LD;
  extern __shared__ double sm[];
  const int b = blockIdx.x, n = a.n, NP = a.NP, T = a.T;
  double *L0 = sm, *L1 = sm + MAT, *L2 = sm + 2 * MAT, *L3 = sm + 3 * MAT;
  double* vec = sm + 4 * MAT;             // hb | cb | Pc | tmp
  double *hb = vec, *cb = vec + 64, *Pc = vec + 128, *tmpv = vec + 192;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
  const double gl = a.g_lognorm[b];
  for (int e = threadIdx.x; e < 4 * MAT + 256; e += 256) sm[e] = 0.0;           // J_bar of step t + 1 = 0, vectors = 0
  const int t_hi = a.t_end - 1, t_lo = a.t_begin;
  double* st = a.p2state + (long)b * (MAT + 64);
  if (a.t_end < T) {                       // a later range ran before: its J_bar / h_bar
    tile_barrier();
    for (int e = threadIdx.x; e < MAT; e += 256) L3[e] = st[e];
    if (threadIdx.x < 64) hb[threadIdx.x] = st[MAT + threadIdx.x];
  }
  // operands of a step, requested one step ahead: Pinv_t, Pinv_bar_t, G_t, J12 (per-step parameters)
  MatRegs ppre, bpre, gpre, jpre;
  d4 gbpre[tv_maxt<NB>()];                  // G_bar_t in the C layout
  double cbpre = 0.0, ctpre = 0.0;          // c_bar_t[i], c_t[i] for thread i < n
  double* ctv = vec + 256;                  // (the vector area has 256 + 3 * TV_MAX_S * 64 doubles: c_t lives behind tmp)
  // (loop-invariant bases: a selection inside the loop becomes a branch with the loads duplicated in its arms, whose
  // merge copies -- and therefore waits for -- the freshly requested registers.  T = 1 has no pair: any valid address)
  const double* j12_base = T > 1 ? a.J12 + (long)b * a.pair_seq_stride : handoff(a, b, 0);
  const long j12_step = T > 1 ? a.pair_t_stride : 0;
  const double* gb_base = T > 1 ? a.g_bar + (long)b * (T - 1) * n * n : handoff(a, b, 0);
  const long gb_step = T > 1 ? (long)n * n : 0;
  const int pair_ld = T > 1 ? n : NP;
  const MOff<true> offNP = mat_off<true>(NP, n);
  const MOff<A4> offN = mat_off<A4>(n, n), offPair = mat_off<A4>(pair_ld, n);
  const COff<NB> coffPair = c_off<NB>(pair_ld, n, wave, r16, kq);
  // operands the step STARTS with (Pinv_t, J12, G_bar_t, the vectors): requested during the previous step ...
  // Three parts, each issued in FRONT of a product of the previous step: 25 KB of requests per wavefront in one burst
  // fill the CU's texture-addresser queue (64 bytes per clock) and the wavefronts stall at the issue -- 2.9 k cycles per
  // step; in front of a product the queue drains while the MFMAs run.
  auto fetch_step_a = [&](int t) {
    const long bt = (long)b * T + t;
    const double* h = handoff(a, b, t);
    fetch_mat<true>(ppre, h + (long)NP * NP, offNP);
    const int i = threadIdx.x < n ? threadIdx.x : 0;
    cbpre = a.c_bar[bt * n + i];
    ctpre = h[2L * NP * NP + i];
  };
  // (the last step has no pair ahead: its requests are clamped to valid addresses and never used)
  auto fetch_step_b = [&](int t) {
    const int tj = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
    fetch_mat<A4>(jpre, j12_base + tj * j12_step, offPair);
  };
  auto fetch_step_c = [&](int t) {
    const int tj = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
    fetch_c<NB>(gbpre, gb_base + tj * gb_step, coffPair);
  };
  auto fetch_step = [&](int t) { fetch_step_a(t); fetch_step_b(t); fetch_step_c(t); };
  // ... and the ones it needs two products later (G_t, Pinv_bar_t): requested at its top -- they do not live across
  // the loop's back edge
  auto fetch_mid = [&](int t) {
    fetch_mat<true>(gpre, handoff(a, b, t), offNP);
    fetch_mat<A4>(bpre, a.pinv_bar + ((long)b * T + t) * n * n, offN);
  };
  fetch_step(t_hi);
  tile_barrier();
  TV_TICK_INIT
  for (int t = t_hi; t >= t_lo; --t) {
    const long bt = (long)b * T + t;
    const double* ct = ctv;
    stage_mat<NB>(L0, ppre, n);                                                     // Pinv_t
    if (threadIdx.x < n) { cb[threadIdx.x] = cbpre; ctv[threadIdx.x] = ctpre; }
    // P_bar = Pinv X_bar G' - Pinv Pinv_bar Pinv = Pinv (X_bar G' - Pinv_bar Pinv): three products after X_bar instead
    // of four (the bracket accumulates both terms in the same tiles; -Pinv_bar is staged with its sign)
    d4 pbar[tv_maxt<NB>()], gbcur[tv_maxt<NB>()], yacc[tv_maxt<NB>()];
    acc_zero<NB>(pbar);
    acc_zero<NB>(yacc);
#pragma unroll
    for (int j = 0; j < tv_maxt<NB>(); ++j) gbcur[j] = gbpre[j];
    if (t < T - 1) {
      mask_c<NB>(gbcur, n, wave, r16, kq);
      // R = -J12 (info form):  X_bar = -R J_bar - G_bar = J12 J_bar - G_bar ;  c_bar -= R h_bar = += J12 h_bar
      stage_mat<NB>(L1, jpre, n);
      tile_barrier();
      TV_TICK(0)
      fetch_mid(t);                                                              // (in front of the first product)
      matvec<false>(L1, LD, hb, tmpv, n, vec + 320);
      for (int i = threadIdx.x; i < n; i += 256) cb[i] += tmpv[i];
      TV_TICK(1)
      d4 acc[tv_maxt<NB>()];
      acc_zero<NB>(acc);
      gemm_mfma<NB, false, false>(L1, L3, wave, r16, kq, acc);                   // J12 J_bar
#pragma unroll
      for (int j = 0; j < tv_maxt<NB>(); ++j) acc[j] -= gbcur[j];
      store_acc<NB>(L2, acc, wave, r16, kq);                                     // X_bar
      tile_barrier();                                                           // (L1 = J12 and L3 = J_bar consumed)
      TV_TICK(2)
      stage_mat<NB>(L1, gpre, n);                                                   // G_t
    } else {
      tile_barrier();
      fetch_mid(t);
    }
    TV_TICK(13)
    stage_mat<NB>(L3, bpre, n, -1.0);                                               // -Pinv_bar (direct + Cholesky part)
    TV_TICK(14)
    const int tnx = t > t_lo ? t - 1 : t_lo;                                     // (requests: unconditional, clamped to the range)
    TV_TICK(15)
    tile_barrier();
    TV_TICK(3)
    fetch_step_a(tnx);
    if (t < T - 1) gemm_mfma<NB, false, true>(L2, L1, wave, r16, kq, yacc);      // X_bar G'
    TV_TICK(4)
    fetch_step_b(tnx);
    gemm_mfma<NB, false, false>(L3, L0, wave, r16, kq, yacc);                     // - Pinv_bar Pinv
    TV_TICK(5)
    tile_barrier();                                                             // (L2 = X_bar consumed)
    store_acc<NB>(L2, yacc, wave, r16, kq);                                      // Y
    tile_barrier();
    TV_TICK(6)
    fetch_step_c(tnx);
    gemm_mfma<NB, false, false>(L0, L2, wave, r16, kq, pbar);                    // P_bar = Pinv Y
    TV_TICK(7)
    TV_TICK(8)
    matvec<false>(L0, LD, cb, Pc, n, vec + 320);                                // Pc = Pinv c_bar  (barriers inside)
    TV_TICK(9)
    // P_bar -= Pc c' + gl/2 c c' + gl/2 Pinv
    for_owned<NB>(pbar, wave, r16, kq, [&](int r, int c, double& v) {
      if (r < n && c < n) v -= Pc[r] * ct[c] + 0.5 * gl * (ct[r] * ct[c] + L0[r * LD + c]);
      else v = 0.0;
    });
    tile_barrier();
    TV_TICK(10)
    symmetrize_acc<NB>(L3, pbar, wave, r16, kq);
    TV_TICK(11)
    for (int i = threadIdx.x; i < n; i += 256) {
      const double hf = Pc[i] + gl * ct[i];
      a.g_node_J[bt * n + i] = -2.0 * L3[i * LD + i];
      a.g_node_h[bt * n + i] = hf;
      hb[i] = hf;
    }
    tile_barrier();
    TV_TICK(12)
  }
  if (t_lo > 0) {                          // an earlier range follows
    for (int e = threadIdx.x; e < MAT; e += 256) st[e] = L3[e];
    if (threadIdx.x < 64) st[MAT + threadIdx.x] = hb[threadIdx.x];
  }
#ifdef SVAE_TV_TIMING
  if (threadIdx.x == 0) for (int q = 0; q < 16; ++q) a.g_node_h[(long)b * T * n + q] = (double)tm[q];
#endif
}

// ---- backward sampler recursion ----------------------------------------------------------------------------------
// x_t = c_t + noise_t + G_t x_{t+1}  (t = T-1 .. 0; natural_sample_backward, cython_lds_inference.pyx:310-355, with the
// noise chol(P_t)^-T eps_t precomputed for all (sequence, step) pairs: it does not depend on the recursion).
// One workgroup per sequence, G_t staged in LDS, S <= 16 samples.
__global__ __launch_bounds__(256) void tile_sample_kernel(int T, int n, int S, int NP, const double* ws,
                                                          const double* noise, double* samples) {
  // Four threads per row of G_t: thread (row, part) = (tid >> 2, tid & 3) holds the row's columns 16 part .. 16 part + 15
  // in registers, requested one step ahead straight from the hand-off (NP-padded with zeros: no masking of columns up to
  // NP); x_{t+1} of the S samples sits in 512-byte LDS vectors, double-buffered -- ONE barrier per step, no staging of
  // the matrix.  (The first version staged G_t through LDS behind sixteen per-element branches and took three barriers
  // per step: 1.6 us per step; it also competed for LDS with the E-step's backward half it runs next to.)
  __shared__ double xs[2][TV_MAX_S * 64];
  const int b = blockIdx.x, tid = threadIdx.x, row = tid >> 2, part = tid & 3;
  const bool rv = row < n;
  const int rc = rv ? row : 0;
  const long WS = 2L * NP * NP + NP, SN = (long)S * n;
  for (int e = tid; e < 2 * TV_MAX_S * 64; e += 256) (&xs[0][0])[e] = 0.0;      // x_T := 0; rows >= n stay 0
  int coff[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) { const int c = 16 * part + 4 * v; coff[v] = c < NP ? c : 0; }
  // Requests run PF steps ahead (a step is shorter than the memory latency: one step ahead left every step waiting for
  // its row): a ring of PF register sets, the time loop unrolled by PF so that the ring index is a compile-time constant.
  // Lane `part` finishes the samples s = part + 4 k: their noise terms, and c_t[row].
  constexpr int PF = 2;
  d4 g[PF][4];
  double cpre[PF], npre[PF][4];
  auto fetch = [&](auto slot, int t) {           // every request unconditional, at clamped addresses
    constexpr int u = decltype(slot)::value;
    const double* h = ws + ((long)b * T + (t > 0 ? t : 0)) * WS;
#pragma unroll
    for (int v = 0; v < 4; ++v) g[u][v] = *(const d4*)(h + (long)rc * NP + coff[v]);
    cpre[u] = h[2L * NP * NP + rc];
    const double* nz = noise + ((long)b * T + (t > 0 ? t : 0)) * SN + rc;
#pragma unroll
    for (int k = 0; k < 4; ++k) npre[u][k] = nz[(long)(part + 4 * k < S ? part + 4 * k : 0) * n];
  };
  static_for<0, PF>([&](auto slot) { fetch(slot, T - 1 - decltype(slot)::value); });
  __syncthreads();
  int cur = 0;
  for (int t0 = T - 1; t0 >= 0; t0 -= PF) {
    static_for<0, PF>([&](auto slot) {
      constexpr int u = decltype(slot)::value;
      const int t = t0 - u;
      const bool live = t >= 0;                    // (steps past the start of the chain compute on clamped data, write nothing)
      // (no masking of the column chunks beyond NP: their entries of x are zero -- rows >= n of xs are never written --
      //  and the clamped requests return finite entries of G)
      const d4 (&gc)[4] = g[u];
      const double (&nn)[4] = npre[u];
      const double cc = cpre[u];
      const double* xc = xs[cur] + 16 * part;
      double* xw = xs[cur ^ 1];
      double* out = samples + ((long)b * T + (live ? t : 0)) * SN;
      // (a runtime loop over the samples, the noise term picked by selects: fully unrolled over the 16 possible samples
      //  the kernel needed 180 .. 260 registers and no longer fitted beside the E-step's backward half on a SIMD)
      for (int s_ = 0; s_ < S; ++s_) {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const d4 x4 = *(const d4*)(xc + s_ * 64 + 4 * v);
          p0 = __builtin_fma(gc[v][0], x4[0], p0);
          p1 = __builtin_fma(gc[v][1], x4[1], p1);
          p0 = __builtin_fma(gc[v][2], x4[2], p0);
          p1 = __builtin_fma(gc[v][3], x4[3], p1);
        }
        const double p = group_sum<4>(p0 + p1);      // (G_t x_{t+1,s})[row]  (x_T = 0: nothing at t = T-1)
        const int k = s_ >> 2;
        const double nk = k == 0 ? nn[0] : k == 1 ? nn[1] : k == 2 ? nn[2] : nn[3];
        if ((s_ & 3) == part && rv && live) {
          const double v = (cc + nk) + p;
          xw[s_ * 64 + row] = v;
          out[(long)s_ * n + row] = v;
        }
      }
      fetch(slot, t - PF);                           // refill the slot: PF steps ahead
      tile_barrier();                                // (LDS only: __syncthreads() would wait for the requests just issued)
      cur ^= 1;
    });
  }
}

}  // namespace svae

extern "C" size_t svae_lds_tile_vjp_workspace_doubles(int B, int T, int n, int S) {
  if (B <= 0 || T <= 0 || n <= 0 || n > 64 || S < 0) return 0;
  const int NP = 16 * ((n + 15) / 16);
  return (size_t)B * T * n * n * 2 + (size_t)B * (T > 1 ? T - 1 : 0) * n * n + (size_t)B * T * n +
         (size_t)B * T * (S > 0 ? S : 0) * n + (size_t)B * (svae::tv_mat(NP) + 64);     // ... | phase-2 state between ranges
}

// phase 0, 1, 2 as described at the top; `workspace` holds [sig | pinv_bar | g_bar | c_bar | xbar] in that
// order (svae_lds_tile_vjp_workspace_doubles); between phase 1 and phase 2 the caller adds the Cholesky adjoint
// of the noise factor into pinv_bar (it is batched over all (sequence, step) pairs and needs xbar).
extern "C" int svae_lds_tile_vjp_f64(int phase, int B, int T, int n, int S, int t_begin, int t_end, int inhomog, int pair_batched,
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
  if (t_begin < 0 || t_end > T || t_begin >= t_end || (phase != 2 && (t_begin != 0 || t_end != T))) return -20;
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
  a.p2state = (double*)workspace + svae_lds_tile_vjp_workspace_doubles(B, T, n, g_samples ? S : 0) -
              (size_t)B * (svae::tv_mat(16 * ((n + 15) / 16)) + 64);
  a.t_begin = t_begin; a.t_end = t_end;
  a.g_node_J = g_node_J; a.g_node_h = g_node_h;
  hipStream_t s = (hipStream_t)stream;
  auto go2 = [&](auto nb, auto a4c) -> int {
    constexpr int NB = decltype(nb)::value;
    constexpr bool A4 = decltype(a4c)::value;
    const size_t lds0 = (size_t)3 * svae::tv_mat(16 * NB) * sizeof(double);
    const size_t lds12 = (size_t)(4 * svae::tv_mat(16 * NB) + 256 + 3 * svae::TV_MAX_S * 64) * sizeof(double);
    static svae::LdsGrant grant0, grant1, grant2;      // (per instantiation, per device inside)
    if (phase == 0) {
      if (!grant0.ensure(reinterpret_cast<const void*>(svae::tile_vjp_phase0<NB, A4>), (long)lds0)) return -1001;
      hipLaunchKernelGGL((svae::tile_vjp_phase0<NB, A4>), dim3(B), dim3(256), lds0, s, a);
    } else if (phase == 1) {
      if (!grant1.ensure(reinterpret_cast<const void*>(svae::tile_vjp_phase1<NB, A4>), (long)lds12)) return -1001;
      hipLaunchKernelGGL((svae::tile_vjp_phase1<NB, A4>), dim3(B), dim3(256), lds12, s, a);
    } else {
      if (!grant2.ensure(reinterpret_cast<const void*>(svae::tile_vjp_phase2<NB, A4>), (long)lds12)) return -1001;
      hipLaunchKernelGGL((svae::tile_vjp_phase2<NB, A4>), dim3(B), dim3(256), lds12, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  };
  // A4: every compact n x n operand (row stride n) can be moved 32 bytes at a time
  auto go = [&](auto nb) -> int {
    return (n & 3) == 0 ? go2(nb, std::true_type{}) : go2(nb, std::false_type{});
  };
  switch ((n + 15) / 16) {
    case 1: return go(std::integral_constant<int, 1>{});
    case 2: return go(std::integral_constant<int, 2>{});
    case 3: return go(std::integral_constant<int, 3>{});
    default: return go(std::integral_constant<int, 4>{});
  }
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
  hipLaunchKernelGGL(svae::tile_sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, T, n, S,
                     16 * ((n + 15) / 16), (const double*)handoff_workspace, noise, samples);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
