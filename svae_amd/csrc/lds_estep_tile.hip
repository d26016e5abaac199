// lds_estep_tile.hip -- LDS E-step for latent dimension 16 <= n <= 64 on MI355X (gfx950): the
// "MFMA only if latent dim >= 16 makes it a real contraction" leg of the hot path.
//
// What it replaces (reference = mattjj/svae): the same functions as the register path
// (lds_estep_kernel.hpp): natural_filter_forward_general svae/lds/cython_lds_inference.pyx:28-90,
// natural_smoother_general :149-195, _compute_stats :197-210, at latent sizes where one matrix no
// longer fits a 16-lane DPP row.
//
// Mapping: one workgroup (4 wavefronts) per sequence, two workgroups per CU.  The per-step
// matrices live in LDS as ONE row-major panel  M = [ P | R | c ]  of NP x (2 NP + 1) doubles
// (NP = n rounded up to 16; P = pivot block J_filt + J11, R = J12, last column: c = P^-1 h),
// cut in 16x16 tiles.  Every O(n^3) stage is a
// list of tile products on v_mfma_f64_16x16x4_f64 with A/B fragments read straight from the panel
// (row stride == 2 mod 32 doubles: the A-fragment read is bank-conflict free):
//   forward   in-place BLOCK Gauss-Jordan of [P | R] with 16x16 block pivots: P -> P^-1,
//             R -> X = P^-1 J12; c = P^-1 h and J12' c are matrix-vector products on the vector ALU
//             (as a 1-wide tile column of the panel h cost a full tile product per row update and
//             Schur row).  The pivot tile A_kk = L D L' is factored by one
//             wavefront with the DPP elimination of the register path (pivots -> log det P), which
//             leaves U = L^-1 and D^-1; the block row is then A_kk^-1 A_kj = U' D^-1 (U A_kj), two
//             tile products (applying the explicit inverse tile instead costs 2 digits on
//             ill-conditioned models).  Look-ahead: while wavefronts 1..3 eliminate with block
//             pivot k (one tile row each), wavefront 0 updates and factors pivot tile k+1; the
//             first pivot tile of the next step is factored while the others reload the right-hand
//             sides.  Pair-parameter operands come pre-packed in fragment order (tile_pack_pairs_kernel).
//             Then the Schur step  P' = (J22 + J11) - J12' X,  h' = J12' c + node_h.
//   backward  moment form:  W = Sigma_{t+1} X',  Sigma_t = P^-1 + X W,  m_t = c + X m_{t+1};
//             tile column j of W stays in registers (the MFMA C layout IS the B-operand layout)
//             and feeds the second product directly.
// Padding rows/columns (n < NP) carry an identity diagonal: pivots 1, log det unchanged.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "lds_args.hpp"
#include "per_device.hpp"

#ifndef SVAE_TILE_SGB
#define SVAE_TILE_SGB 7
#endif

// Scheduling hint: NM groups of {1 MFMA, NR LDS reads, NW LDS writes}.  An in-order wavefront stalls
// on the next MFMA while the pipe is busy, so independent LDS work must sit BETWEEN the MFMAs in
// program order to overlap with them (measured: -16 % on the elimination's tile updates).
#define SVAE_SGB_(ID, NM, NR, NW)                                               \
  _Pragma("unroll") for (int g_ = 0; g_ < (NM); ++g_) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, ID);                        \
    if ((NR) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (NR), ID);       \
    if ((NW) > 0) __builtin_amdgcn_sched_group_barrier(0x200, (NW), ID);       \
  }
// SVAE_TILE_SGB: bit 0 = elimination row updates, bit 1 = Schur stage, bit 2 = backward Sigma update.  (Bit 1 was off
// through round 2 and most of round 3: with it hipcc 7.2 produced wrong code for the per-step-parameter n = 64
// instantiation of the kernel as it was then -- caught by tests/test_lds_tile_hip.py.  With the register budgets and
// single-half instances of round 3 every instantiation passes with it; 0.7 % at 512 sequences.  -DSVAE_TILE_SGB=5
// switches it off again.)
#define SVAE_SGB(ID, NM, NR, NW) if constexpr (((SVAE_TILE_SGB) >> ((ID) - 1)) & 1) { SVAE_SGB_(ID, NM, NR, NW) }

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

// Two independent products sharing the A fragment, MFMAs interleaved (the second chain fills the
// result latency of the first).
__device__ __forceinline__ void mma16x2(const d4 a, const d4 b0, d4& c0, const d4 b1, d4& c1) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb], b0[kb], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb], b1[kb], c1, 0, 0, 0);
  }
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

// writes the TRANSPOSE of a C-layout tile at (row0, col0)  (same addressing as frag_a)
__device__ __forceinline__ void store_ct(double* M, int ld, int row0, int col0, int r16, int kq, const d4 v) {
  double* p = M + (row0 + r16) * ld + col0 + kq;
  p[0] = v[0]; p[4] = v[1]; p[8] = v[2]; p[12] = v[3];
}
// Symmetric NB x NB tile grid, every unordered pair {i, j} formed once: column j takes the rows (j + q) mod NB,
// q < sym_cnt(NB, j)
constexpr int sym_cnt(int NB, int j) { return (NB % 2 == 0 && j >= NB / 2) ? NB / 2 : NB / 2 + 1; }
// The Schur stage's variant of the same idea (NB >= 3, one wavefront per tile row): wavefront 0 forms ONLY tile (0, 0) --
// it factors that tile, the first pivot of the next step, while the others are still busy -- and tile row i >= 1 takes
// (i, i), its share of the pairs among the rows 1 .. NB-1 (the cyclic assignment on that sub-grid) and (i, 0):
// NB = 4 -> 1, 3, 3, 3 tiles;  schur_col(NB, i, s) = tile column of the s-th tile of row i.
constexpr int schur_cnt(int NB, int i) { return i == 0 ? 1 : sym_cnt(NB - 1, i - 1) + 1; }
constexpr int schur_col(int NB, int i, int s) {
  return i == 0 ? 0 : (s < sym_cnt(NB - 1, i - 1) ? 1 + (i - 1 + s) % (NB - 1) : 0);
}

template <int NB>
struct TileCfg {
  static constexpr int NP = 16 * NB;
  static constexpr int NTC = 2 * NB;            // tile columns of the panel [P | R] (column 2 NP of a row: c = P^-1 h)
  static constexpr int LDM = 2 * NP + 2;        // == 2 (mod 32)
  static constexpr int WSTEP = 2 * NP * NP + NP;   // hand-off per step: X, P^-1 (row-major NP x NP), c
  static constexpr int LDU = 18;                // pivot-factor tile U = L^-1, row stride
  static constexpr int UBUF = 16 * LDU + 16;    // U and D^-1; double-buffered
  static constexpr int LDS_DOUBLES = NP * LDM + 3 * NP + 16 + 2 * UBUF + 2;
};

// 16x16 SPD tile A = L D L'  ->  U = L^-1 (unit lower triangular) and D^-1, by the calling wavefront
// (lane r16 = column, one register per row, the four DPP rows work redundantly).  Forward
// elimination only: row i > p gets  row_i -= (A[i][p] / d_p) * row_p, with the multiplier written
// into lane p, so that lanes c < i of row i end as U[i][c].  Accumulates log det as mantissa/exponent.
__device__ __forceinline__ void factor_pivot_tile(const double* tile, int ld, double* U, int ldu,
                                                  double* dinv, int r16, int kq,
                                                  double& pmin, double& ldM, int& ldE) {
  double A[16];
  static_for<0, 16>([&](auto r) { A[r] = tile[r * ld + r16]; });
  dpp_fence(A);
  double dv = 0.0;
  double pv = bcast_fenced<0>(A[0]);
  double rinv = rcp_nr(pv);
  double pprod = 1.0;
  static_for<0, 16>([&](auto p) {
    const double Ep = (r16 == p) ? 1.0 : 0.0;       // per-lane selects done arithmetically (x * Ep, exact)
    pmin = fmin(pmin, pv);
    pprod *= pv;
    dv = __builtin_fma(Ep, rinv, dv);
    const double r = __builtin_fma(Ep, 1.0 - pv, A[p]) * rinv;       // lane p: 1/pivot
    // row updates in groups of four: the four lane-p clears first (independent), then the four DPP
    // multiply-accumulates, so that no instruction waits for its predecessor's 8-cycle latency
    // (the exact two-instruction form: the single-FMA form of the register path, gauss_jordan in
    // lds_estep_kernel.hpp, loses a factor 3 on ill-conditioned n = 64 models here and gains nothing)
    auto update4 = [&](auto i0, auto cnt) {
      constexpr int I0 = decltype(i0)::value, C = decltype(cnt)::value;
      double olds[C], accs[C];
      static_for<0, C>([&](auto j) { olds[j] = A[I0 + j]; accs[j] = __builtin_fma(-olds[j], Ep, olds[j]); });
      static_for<0, C>([&](auto j) { mac_bc<p, true, false>(accs[j], olds[j], r); A[I0 + j] = accs[j]; });
    };
    if constexpr (p + 1 < 16) {
      // software pipelining by hand: row p+1 first, broadcast its pivot, then the reciprocal chain of
      // the NEXT pivot between the remaining row updates (cf. gauss_jordan, lds_estep_kernel.hpp)
      update4(std::integral_constant<int, p + 1>{}, std::integral_constant<int, 1>{});
      const double pn = bcast_fenced<p + 1>(A[p + 1]);
      double t0 = 0.0, e0 = 0.0, t1 = 0.0, e1 = 0.0, rn = 0.0;
      constexpr int REM = 14 - p;                      // row updates still to come
      constexpr int NG = (REM + 3) / 4;                // ... in groups of four
      auto chain = [&](auto s) {
        if constexpr (s == 0) t0 = asm_rcp(pn);
        else if constexpr (s == 1) e0 = asm_fnma1(pn, t0);
        else if constexpr (s == 2) t1 = asm_fma(t0, e0, t0);
        else if constexpr (s == 3) e1 = asm_fnma1(pn, t1);
        else if constexpr (s == 4) rn = asm_fma(t1, e1, t1);
      };
      if constexpr (NG == 0) static_for<0, 5>(chain);
      static_for<0, NG>([&](auto g) {
        constexpr int i0 = p + 2 + 4 * g;
        constexpr int c = (16 - i0) < 4 ? (16 - i0) : 4;
        constexpr int lo = g * 5 / NG, hi = (g + 1) * 5 / NG;
        static_for<lo, hi>(chain);                     // chain steps ahead of the group they overlap with
        update4(std::integral_constant<int, i0>{}, std::integral_constant<int, c>{});
      });
      pv = pn;
      rinv = rn;
    }
    if constexpr (p == 7 || p == 15) {             // keep the running product of pivots in range
      ldE += __builtin_amdgcn_frexp_exp(pprod);
      ldM *= __builtin_amdgcn_frexp_mant(pprod);
      pprod = 1.0;
    }
  });
  ldE += __builtin_amdgcn_frexp_exp(ldM);
  ldM = __builtin_amdgcn_frexp_mant(ldM);
  if (kq == 0) {
    static_for<0, 16>([&](auto r) { U[r * ldu + r16] = r16 < r ? A[r] : (r16 == r ? 1.0 : 0.0); });
    dinv[r16] = dv;
  }
}

// WPC = workgroups per CU the register allocation is sized for.  Two of them (256 registers per lane) is what lets
// 512 sequences run in one round, at the price of ~570 spilled dwords whose scratch traffic sits between the operand
// requests of every step; up to one workgroup per CU (B <= 256) the one-per-CU instance (512 registers, no spills) is
// the faster one: n = 64, T = 1000: 64 sequences 27.2 -> 23.3 ms, 256: 23.5 ms; 512 sequences 42.8 (two per CU) vs
// 46.8 ms (one per CU, two rounds).
// HALF: 0 = the half (or both) a.tile_half names at run time; 1 / 2 = an instance that CONTAINS only the forward /
// backward half.  Compiled together at 256 registers per lane the two halves spill ~510 dwords; each alone fits (the
// forward half: 254 registers, no scratch) -- so batches of more than one workgroup per CU run the E-step as two
// launches of the single-half instances.
template <int NB, bool INHOMOG, int WPC, int HALF>
__global__ __launch_bounds__(256, WPC) void lds_estep_tile_kernel(const LdsArgs a, const int n,
                                                                const double* __restrict__ pk_base,
                                                                const int pk_batched) {
  // a.tile_half: 0 = the whole E-step; 1 = the forward half only (filter, hand-off, log-normaliser); 2 = the backward
  // half only (smoother and statistics from the hand-off a forward-only launch left).  The halves meet only through
  // the hand-off in global memory, so a training step runs the backward half NEXT to the kernels that need the
  // hand-off alone (noise factor + sampler recursion, phase 0 of the VJP): one workgroup per sequence leaves them room.
  const int half = HALF ? HALF : a.tile_half;
  using Cfg = TileCfg<NB>;
  constexpr int NP = Cfg::NP, LDM = Cfg::LDM, WSTEP = Cfg::WSTEP;
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
  const double* nodeJ = a.node_J + (long)b * T * n;
  const double* nodeh = a.node_h + (long)b * T * n;
  double* wsb = a.ws + (long)b * T * WSTEP;
  // packed pair parameters: per set, (INHOMOG ? T-1 : 2) slots of [pA | pC | pR] (3 NP^2 doubles)
  const double* packed = pk_base + (pk_batched ? (long)b * (T - 1) * (3 * NP * NP) : 0);

  double ldM = 1.0, pmin = 1.0e300, qacc = 0.0;
  int ldE = 0;
#ifdef SVAE_TILE_SKEW_NS
  // Round-6 experiment ("two sequences skewed by half a step"): with two workgroups per CU
  // blocks b and b + gridDim/2 share a CU (profiles/r5_tile_n64_b512/wave_placement_hwid.txt); the second one starts
  // SVAE_TILE_SKEW_NS later, so that its DPP pivot factorisations fall into the first one's MFMA phases (and vice versa)
  // for as long as the two keep their distance.  Measured: DESIGN.md section 8.  Not compiled in by default.
  if (WPC == 2 && blockIdx.x >= (gridDim.x + 1) / 2) {
    const long long t0_ = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0_ < (SVAE_TILE_SKEW_NS) / 10) __builtin_amdgcn_s_sleep(8);
  }
#endif
#ifdef SVAE_TILE_TIMING   // per-phase cycle counts of wave 0 (tools/tile_timing.py); results are not written
  long long tm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast_ = __builtin_readcyclecounter();
#define TICK(i) { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tlast_; tlast_ = now_; }
#else
#define TICK(i)
#endif

  // B/C fragments by tile column jt < 2 NB of the panel [P | R].  The right-hand side h does NOT ride through the
  // elimination (as a 1-wide tile column it cost a full 16-wide MFMA product per row update and per Schur row: 1/8 and
  // 1/5 of those phases): c = P^-1 h is a 4-thread-per-row product with the finished inverse in the hand-off phase,
  // J12' c a per-lane product with the A fragments the Schur stage holds anyway.  Column 2 NP of M carries c.
  auto ld_b = [&](int row0, int jt) -> d4 { return frag_b(M, LDM, row0, 16 * jt, r16, kq); };
  auto st_c = [&](int row0, int jt, const d4 v) { store_c(M, LDM, row0, 16 * jt, r16, kq, v); };
  // A_kk^-1 (.) = U' D^-1 U (.) applied to a B-layout tile
  auto apply_pivot = [&](const double* U, const double* dinv, const d4 fb) -> d4 {
    d4 v = mma16(frag_a(U, LDU, 0, 0, r16, kq), fb, d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) v[qq] *= dinv[4 * qq + kq];
    return mma16(frag_b(U, LDU, 0, 0, r16, kq), v, d4{0.0, 0.0, 0.0, 0.0});
  };
  // ---- roles ------------------------------------------------------------------------------------
  // Schur product: tile row si per wavefront (WPR wavefronts share a row when NB <= 2), tile columns
  // sj0, sj0 + WPR, ... < NB; the wavefront with sj0 == WPR - 1 also forms the row's share of J12' c
  constexpr int WPR = NB <= 2 ? 4 / NB : 1;
  constexpr int SCOLS = WPR == 1 ? (NB - 1) / 2 + 2 : (NB + WPR - 1) / WPR;   // (WPR == 1: symmetric tile pairs once, see the Schur step)
  constexpr int RL4 = (NP * NP / 4 + 191) / 192; // 4-double chunks per thread of an NP x NP copy by 3 wavefronts
  int* flag = (int*)(ubuf + 2 * UBUF);           // look-ahead hand-shake (see below)
  if (tid == 0) *flag = 0;

  if (HALF != 2 && half != 2) {
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
  if (tid < NP) hvec[tid] = tid < n ? a.init_h[tid] + nodeh[tid] : 0.0;
  __syncthreads();

  // The forward half is instantiated per wavefront index W (wave-uniform switch below): every tile
  // coordinate is then a compile-time constant and every LDS access is base + immediate.
  auto forward = [&](auto wc) {
  constexpr int W = decltype(wc)::value;
  constexpr bool schur_on = W < NB * WPR;
  constexpr int si = W % NB, sj0 = W / NB;
  constexpr bool h_on = schur_on && sj0 == WPR - 1;
  // Operands taken from the pair parameters come pre-packed (tile_pack_pairs_kernel below) in the
  // exact register order, so each is one coalesced 32-byte load per lane at base + immediate:
  //   sA = -(J12') tiles of row si (A operands of the Schur product), sC = its C input
  //   -2 (J22 + J11') (identity on the padding), rl = this thread's share of the next J12.
  // They are requested inside the last block pivot of a step and consumed after the hand-off: the
  // L2 latency hides behind the elimination.
  d4 sA[NB], sC[SCOLS], rl[RL4];
#ifndef SVAE_TILE_LATE_OPERANDS
#define SVAE_TILE_LATE_OPERANDS 1
#endif
  constexpr bool late_operands = SVAE_TILE_LATE_OPERANDS && WPC == 2;
  constexpr bool lean_schur = WPC == 2 && WPR == 1;   // see the Schur step
  auto load_operands = [&](const double* pk) {
    if constexpr (schur_on) {
#pragma unroll
      for (int kk = 0; kk < NB; ++kk) sA[kk] = *(const d4*)(pk + ((si * NB + kk) * 64 + lane) * 4);
#pragma unroll
      for (int s2 = 0; s2 < SCOLS; ++s2) {
        const int j = WPR == 1 ? schur_col(NB, si, s2) : sj0 + s2 * WPR;
        sC[s2] = d4{0.0, 0.0, 0.0, 0.0};
        if constexpr (lean_schur) continue;           // (requested one tile ahead inside the Schur step)
        if (WPR == 1 ? s2 < schur_cnt(NB, si) : j < NB) sC[s2] = *(const d4*)(pk + NP * NP + ((si * NB + j) * 64 + lane) * 4);
      }
    }
    if constexpr (W > 0) {
#pragma unroll
      for (int u = 0; u < RL4; ++u) {
        const int c4 = tid - 64 + 192 * u;
        rl[u] = d4{0.0, 0.0, 0.0, 0.0};
        if (c4 * 4 < NP * NP) rl[u] = *(const d4*)(pk + 2 * NP * NP + c4 * 4);
      }
    }
  };
  if constexpr (W == 0) factor_pivot_tile(M, LDM, ubuf, LDU, ubuf + 16 * LDU, r16, kq, pmin, ldM, ldE);
  for (int t = 0; t < T; ++t) {
    const bool last = (t == T - 1);
    const bool next_last = (t + 1 == T - 1);
    TICK(0)
    double njn = 0.0, nhn = 0.0, njd = 0.0;
    // ---- in-place block Gauss-Jordan with look-ahead -----------------------------------------------
    // Per block pivot k:  (P1) all wavefronts scale the pivot row with U_k;  (P2) wavefronts 1..3 own
    // the other tile rows (eliminate, then rewrite their pivot-column tile) while wavefront 0 updates
    // the NEXT pivot tile first and factors it, so that the serial DPP factorisation overlaps the MFMA
    // work.  The owner of row k+1 must not overwrite tile (k+1,k) before wavefront 0 has read it: an
    // LDS flag carries that one dependency (LDS operations of a wavefront execute in order).
    // (the first pivot tile of this step was factored during the previous step's reload phase)
    TICK(1)
    static_for<0, NB>([&](auto kc) {
      constexpr int k = kc;
      const double* U = ubuf + (k & 1) * UBUF;
      const double* dinv = U + 16 * LDU;
      const int gen = t * NB + k + 1;
      __syncthreads();
      TICK(2)
      {  // (P1) pivot row:  A[k][j] <- A_kk^-1 A[k][j] = U' D^-1 (U A[k][j])   (j != k), two tiles at a time
        constexpr int NT1 = (2 * NB - 1 + 3) / 4;           // 2 NB - 1 tiles beside the pivot tile
        static_assert(NT1 <= 2, "at most two pivot-row tiles per wavefront");
        const int q0 = W, q1 = W + 4;
        const int j0 = q0 + (q0 >= k ? 1 : 0), j1 = q1 + (q1 >= k ? 1 : 0);
        const d4 z4 = {0.0, 0.0, 0.0, 0.0};
        if (q1 < 2 * NB - 1) {
          const d4 fu = frag_a(U, LDU, 0, 0, r16, kq), fut = frag_b(U, LDU, 0, 0, r16, kq);
          d4 v0 = z4, v1 = z4;
          mma16x2(fu, ld_b(16 * k, j0), v0, ld_b(16 * k, j1), v1);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) { const double dq = dinv[4 * qq + kq]; v0[qq] *= dq; v1[qq] *= dq; }
          d4 t0 = z4, t1 = z4;
          mma16x2(fut, v0, t0, v1, t1);
          st_c(16 * k, j0, t0);
          st_c(16 * k, j1, t1);
        } else if (q0 < 2 * NB - 1) {
          st_c(16 * k, j0, apply_pivot(U, dinv, ld_b(16 * k, j0)));
        }
      }
      __syncthreads();
      TICK(3)
      if constexpr (k == NB - 1) {
        // operands of the Schur stage / next right-hand side / next node potentials: requested now,
        // used after the hand-off (unconditional, clamped addresses: a load under a branch is waited
        // for at the join)
        // (two workgroups per CU, SVAE_TILE_LATE_OPERANDS: requested at the point of use instead -- the other workgroup
        //  covers the latency and the ~110 registers are not live through the last block pivot and the hand-off)
        if constexpr (!late_operands) {
          if (!last) load_operands(packed + (long)(INHOMOG ? t : (next_last ? 1 : 0)) * (3 * NP * NP));
        }
        const long tn = (long)(last ? t : t + 1) * n;
        njn = nodeJ[tn + (tid < n ? tid : n - 1)];
        if constexpr (h_on) {
          const int row = 16 * si + r16;
          nhn = nodeh[tn + (row < n ? row : n - 1)];
        }
        if constexpr (schur_on && WPR == 1) {      // the node diagonal of this wavefront's diagonal tile (added in registers)
          const int row = 16 * si + r16;
          njd = nodeJ[tn + (row < n ? row : n - 1)];
        }
      }
      auto inverse_tile = [&]() {   // A[k][k] <- A_kk^-1 = U' D^-1 U
        d4 v = frag_b(U, LDU, 0, 0, r16, kq);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) v[qq] *= dinv[4 * qq + kq];
        store_c(M, LDM, 16 * k, 16 * k, r16, kq, mma16(frag_b(U, LDU, 0, 0, r16, kq), v, d4{0.0, 0.0, 0.0, 0.0}));
      };
      if constexpr (W == 0) {
        if constexpr (k + 1 == NB) inverse_tile();      // (else: done by the owner of row k+1, one task short)
        if constexpr (k + 1 < NB) {
          const d4 fa = frag_a(M, LDM, 16 * (k + 1), 16 * k, r16, kq);
          const d4 c = mma16(-fa, ld_b(16 * k, k + 1), ld_b(16 * (k + 1), k + 1));
          __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (k+1,k) has been read
          st_c(16 * (k + 1), k + 1, c);
          double* Un = ubuf + ((k + 1) & 1) * UBUF;
          factor_pivot_tile(M + (16 * (k + 1)) * LDM + 16 * (k + 1), LDM, Un, LDU, Un + 16 * LDU, r16, kq,
                            pmin, ldM, ldE);
        }
      } else {
        static_for<0, (NB - 1 - (W - 1) + 2) / 3>([&](auto rc) {
          constexpr int r = W - 1 + 3 * decltype(rc)::value;
          constexpr int i = r + (r >= k ? 1 : 0);
          constexpr int skipn = (i == k + 1) ? 2 : 1;         // row k+1: tile (k+1,k+1) belongs to wave 0
          constexpr int cnt = 2 * NB - skipn;
          const d4 fa = frag_a(M, LDM, 16 * i, 16 * k, r16, kq);
          const d4 nfa = -fa;
          auto jmap = [&](int jq) { return jq + (jq >= k ? skipn : 0); };
          // software pipeline over the row's tiles: fragments of tile q+1 are requested, tile q runs on
          // the MFMA pipe, tile q-1 is stored -- a store issued right behind its own MFMAs would stall
          // the wavefront for the result latency with the pipe idle
          d4 cq = ld_b(16 * i, jmap(0)), bq = ld_b(16 * k, jmap(0));
          d4 rprev = cq;
#pragma unroll
          for (int q = 0; q < cnt; ++q) {
            d4 cn = cq, bn = bq;
            if (q + 1 < cnt) { cn = ld_b(16 * i, jmap(q + 1)); bn = ld_b(16 * k, jmap(q + 1)); }
            const d4 rq = mma16(nfa, bq, cq);
            if (q > 0) st_c(16 * i, jmap(q - 1), rprev);
            rprev = rq; cq = cn; bq = bn;
            SVAE_SGB(1, 4, 2, 1)
          }
          st_c(16 * i, jmap(cnt - 1), rprev);
          TICK(1)     // (timing build: row updates of wavefronts 1..3 land in slot 1, the rest of P2 in slot 4)
          // pivot-column tile:  A[i][k] <- -A[i][k] A_kk^-1, computed transposed as A_kk^-1 A[i][k]'
          const d4 rt = apply_pivot(U, dinv, fa);
          if constexpr (i == k + 1) {
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < gen)
              __builtin_amdgcn_s_sleep(1);
          }
          double* p = M + (16 * i + r16) * LDM + 16 * k + kq;
          p[0] = -rt[0]; p[4] = -rt[1]; p[8] = -rt[2]; p[12] = -rt[3];
          if constexpr (i == k + 1) inverse_tile();
        });
      }
      TICK(4)
    });
    __syncthreads();
    TICK(5)

    // ---- hand-off to the backward half: X, P^-1 (row-major NP x NP), c --------------------------
    double* w = wsb + (long)t * WSTEP;
    constexpr int TPR = NP / 4;                  // threads per row of the hand-off copy
    if constexpr (TPR == 4 || TPR == 8 || TPR == 16) {
      // c = P^-1 h rides on the copy: each thread multiplies the four entries of P^-1 it moves anyway with its four
      // entries of h; the TPR threads of a row are one aligned lane group, summed with DPP permutations
      const int col = (tid * 4) % NP;
      const d4 hv4 = *(const d4*)(hvec + col);
      for (int c4 = tid; c4 * 4 < NP * NP; c4 += 256) {
        const int row = (c4 * 4) / NP;
        const d4 pv = *(const d4*)(M + row * LDM + col);
        *(d4*)(w + c4 * 4) = *(const d4*)(M + row * LDM + NP + col);
        *(d4*)(w + NP * NP + c4 * 4) = pv;
        double cv = pv[0] * hv4[0];
        cv = __builtin_fma(pv[1], hv4[1], cv);
        cv = __builtin_fma(pv[2], hv4[2], cv);
        cv = __builtin_fma(pv[3], hv4[3], cv);
        cv = group_sum<TPR>(cv);
        if ((tid & (TPR - 1)) == 0) {
          M[row * LDM + 2 * NP] = cv;
          w[2 * NP * NP + row] = cv;
          qacc = __builtin_fma(hvec[row], cv, qacc);          // h' P^-1 h
        }
      }
    } else {
      {  // c = P^-1 h: four threads per row of the finished inverse
        // (opaque thread index: the handful of LDS / global addresses below are recomputed every step instead of being
        //  hoisted out of the time loop -- as loop-invariant registers they end up in scratch in the 256-register instance)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int row = tq >> 2, part = tq & 3;
        double cv = 0.0;
        if (row < NP) {
          const double* prow = M + row * LDM;
#pragma unroll
          for (int cc = 0; cc < NP; cc += 4) cv = __builtin_fma(prow[cc + part], hvec[cc + part], cv);
        }
        cv += __shfl_xor(cv, 1, 64);
        cv += __shfl_xor(cv, 2, 64);
        if (row < NP && part == 0) {
          M[row * LDM + 2 * NP] = cv;
          w[2 * NP * NP + row] = cv;
          qacc = __builtin_fma(hvec[row], cv, qacc);          // h' P^-1 h
        }
      }
      for (int c4 = tid; c4 * 4 < NP * NP; c4 += 256) {
        const int row = (c4 * 4) / NP, col = (c4 * 4) % NP;
        *(d4*)(w + c4 * 4) = *(const d4*)(M + row * LDM + NP + col);
        *(d4*)(w + NP * NP + c4 * 4) = *(const d4*)(M + row * LDM + col);
      }
    }
    __syncthreads();
    TICK(6)

    if constexpr (late_operands) {
      if (!last) load_operands(packed + (long)(INHOMOG ? t : (next_last ? 1 : 0)) * (3 * NP * NP));
    }
    if (!last) {
      // ---- Schur step:  P' = -2 (J22 + J11') + diag(-2 node_J') - J12' X   (tile (si,j), j < NB);
      //                   h' = node_h' + J12' c ------------------------------------------------------
      if constexpr (schur_on && WPR == 1) {
        // P' is symmetric: each mirror pair of tiles is formed once and stored at both positions (the mirror image
        // transposed); the assignment (schur_col: NB = 4 -> 1, 3, 3, 3 tiles instead of 4 each) leaves wavefront 0 time to
        // factor tile (0, 0) -- the next step's first pivot tile -- BEFORE the barrier instead of behind it, where the
        // other three waited for it (3.5 k cycles per step).  The node diagonal goes onto the diagonal tile in registers.
        constexpr int CNT = schur_cnt(NB, si);
        const double njd2 = (16 * si + r16 < n) ? 2.0 * njd : 0.0;
        auto add_diag = [&](d4& c) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) c[qq] -= (4 * qq + kq == r16) ? njd2 : 0.0;
        };
        if constexpr (lean_schur) {
          // two workgroups per CU: B fragments read where they are consumed and the C inputs requested one tile ahead
          // (the other workgroup covers the latencies; the buffers below are ~90 registers the 256-register instance
          // does not have: it kept the C inputs in scratch)
          const double* pkc = packed + (long)(INHOMOG ? t : (next_last ? 1 : 0)) * (3 * NP * NP) + NP * NP
                              + (si * NB * 64 + lane) * 4;
          d4 cnx = *(const d4*)(pkc + schur_col(NB, si, 0) * 256);
          static_for<0, CNT>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int j = schur_col(NB, si, q), jn = schur_col(NB, si, q + 1 < CNT ? q + 1 : q);
            d4 c = cnx;
            if constexpr (q + 1 < CNT) cnx = *(const d4*)(pkc + jn * 256);
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) c = mma16(sA[kk], ld_b(16 * kk, NB + j), c);
            if constexpr (j == si) add_diag(c);
            store_c(M, LDM, 16 * si, 16 * j, r16, kq, c);
            if constexpr (j != si) store_ct(M, LDM, 16 * j, 16 * si, r16, kq, c);
            __builtin_amdgcn_sched_barrier(0);     // (keeps the next tile's fragment reads from being hoisted above: spills)
          });
        } else {
          d4 fb[NB];
#pragma unroll
          for (int kk = 0; kk < NB; ++kk) fb[kk] = ld_b(16 * kk, NB + schur_col(NB, si, 0));
          static_for<0, CNT>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int j = schur_col(NB, si, q), jn = schur_col(NB, si, q + 1 < CNT ? q + 1 : q);
            d4 fn[NB];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) fn[kk] = (q + 1 < CNT) ? ld_b(16 * kk, NB + jn) : fb[kk];
            d4 c = sC[q];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) c = mma16(sA[kk], fb[kk], c);
            SVAE_SGB(2, 4 * NB, 1, 0)
            if constexpr (j == si) add_diag(c);
            store_c(M, LDM, 16 * si, 16 * j, r16, kq, c);
            if constexpr (j != si) store_ct(M, LDM, 16 * j, 16 * si, r16, kq, c);
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) fb[kk] = fn[kk];
          });
        }
      } else if constexpr (schur_on && sj0 < NB) {
        d4 fb[NB];
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) fb[kk] = ld_b(16 * kk, NB + sj0);
#pragma unroll
        for (int s2 = 0; s2 < SCOLS; ++s2) {
          const int j = sj0 + s2 * WPR;
          if (j < NB) {
            d4 fn[NB];
            const int jn = j + WPR;
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) fn[kk] = fb[kk];
            if (s2 + 1 < SCOLS && jn < NB) {
#pragma unroll
              for (int kk = 0; kk < NB; ++kk) fn[kk] = ld_b(16 * kk, NB + jn);
            }
            d4 c = sC[s2];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) c = mma16(sA[kk], fb[kk], c);
            SVAE_SGB(2, 4 * NB, 1, 0)
            store_c(M, LDM, 16 * si, 16 * j, r16, kq, c);
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) fb[kk] = fn[kk];
          }
        }
      }
      if constexpr (h_on) {
        // sA[kk][kb] = (-J12')[16 si + r16][16 kk + 4 kb + kq]: each lane a quarter of its row's product with c, the
        // four DPP rows summed through the cross-lane network
        double s0 = 0.0, s1 = 0.0;
        int kqx = kq;
        asm volatile("" : "+v"(kqx));               // (as above: no loop-invariant address register)
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) {
          const double* cp = M + (16 * kk + kqx) * LDM + 2 * NP;
          s0 = __builtin_fma(sA[kk][0], cp[0], s0);
          s1 = __builtin_fma(sA[kk][1], cp[4 * LDM], s1);
          s0 = __builtin_fma(sA[kk][2], cp[8 * LDM], s0);
          s1 = __builtin_fma(sA[kk][3], cp[12 * LDM], s1);
        }
        double sh = s0 + s1;
        sh += __shfl_xor(sh, 16, 64);
        sh += __shfl_xor(sh, 32, 64);
        const int row = 16 * si + r16;
        if (kq == 0) mv0[row] = row < n ? nhn - sh : 0.0;     // sh = -(J12' c_t)[row]
      }
      // (one wavefront per tile row: wavefront 0 has formed only tile (0, 0), node diagonal included, and factors it
      //  now -- nothing else reads that tile or the factor buffer of pivot 0 in this stage)
      if constexpr (W == 0 && WPR == 1) factor_pivot_tile(M, LDM, ubuf, LDU, ubuf + 16 * LDU, r16, kq, pmin, ldM, ldE);
      __syncthreads();
      TICK(7)
      // ---- next step: wavefronts 1..3 write the right-hand sides (latent dimension <= 32, several wavefronts per tile
      //      row: wavefront 0 adds the node diagonal and factors the first pivot tile here) ------------------------------
      if constexpr (W == 0) {
        if constexpr (WPR > 1) {
          if (tid < n) M[tid * LDM + tid] -= 2.0 * njn;
          factor_pivot_tile(M, LDM, ubuf, LDU, ubuf + 16 * LDU, r16, kq, pmin, ldM, ldE);
        }
      } else {
        const int t3 = tid - 64;
#pragma unroll
        for (int u = 0; u < RL4; ++u) {
          const int c4 = t3 + 192 * u;
          if (c4 * 4 < NP * NP) *(d4*)(M + ((c4 * 4) / NP) * LDM + NP + ((c4 * 4) % NP)) = rl[u];
        }
        if (t3 < NP) hvec[t3] = mv0[t3];
      }
      __syncthreads();
      TICK(8)
    }
  }
  };
  switch (wave) {
    case 0: forward(std::integral_constant<int, 0>{}); break;
    case 1: forward(std::integral_constant<int, 1>{}); break;
    case 2: forward(std::integral_constant<int, 2>{}); break;
    default: forward(std::integral_constant<int, 3>{}); break;
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

  }   // forward half
#ifdef SVAE_TILE_TIMING
  if ((HALF == 1 || half == 1) && lane == 0 && wave < 2) {
    for (int q = 0; q < 12; ++q) a.E_init[(long)b * (nn + n) + 12 * wave + q] = (double)tm[q];
  }
#endif
  if (HALF == 1 || half == 1) return;
  __syncthreads();
#ifdef SVAE_TILE_FWD_ONLY   // register-pressure experiments
  return;
#endif
  // ---- backward pass (moment form) -------------------------------------------------------------
  // Per step: (B0) the prefetched X_t, c_t go to LDS; (B1) W = Sigma_{t+1} X_t' for this wavefront's tile
  // column (kept in registers) and m_t = c_t + X_t m_{t+1}; (B2) Sigma_t = P_t^-1 + X_t W tile by tile,
  // straight into LDS, with the statistics accumulated from the same registers.
  for (int idx = tid; idx < NP * NP; idx += 256) M[(idx / NP) * LDM + (idx % NP)] = 0.0;   // Sigma_T := 0
  if (tid < NP) { mv0[tid] = 0.0; mv1[tid] = 0.0; }
  double* mold = mv0;
  double* mnew = mv1;
  d4 cross[NB], sxx[NB / 2 + 1], prevE[NB / 2 + 1];   // sxx / prevE: one per COMPUTED tile of the wavefront's column (B2)
#pragma unroll
  for (int i = 0; i < NB; ++i) cross[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NB / 2 + 1; ++i) { sxx[i] = d4{0.0, 0.0, 0.0, 0.0}; prevE[i] = sxx[i]; }
  double* oEx = a.E_node_x + (long)b * T * n;
  double* oExx = a.E_node_diagxx + (long)b * T * n;
  double* oI = a.E_init + (long)b * (nn + n);
  double* oPh = a.E_pair + (long)b * 3 * nn;            // homogeneous: the three sums
  constexpr int RL4B = (NP * NP / 4 + 255) / 256;     // 4-double chunks per thread of an NP x NP copy
  d4 xr[RL4B];
  double cpre = 0.0;
  auto prefetch_step = [&](const double* w) {       // X_t (row-major NP x NP) and c_t of one step
#pragma unroll
    for (int u = 0; u < RL4B; ++u) {
      const int c4 = tid + 256 * u;
      xr[u] = d4{0.0, 0.0, 0.0, 0.0};
      if (c4 * 4 < NP * NP) xr[u] = *(const d4*)(w + c4 * 4);
    }
    cpre = w[2 * NP * NP + (tid < NP ? tid : 0)];
  };
  // Two workgroups per CU (256 registers): the step's X_t / P_t^-1 are requested where they are consumed instead of a
  // step / a phase ahead -- the other workgroup covers the latency, and ~64 registers are not live through B1 / B2
#ifndef SVAE_TILE_LEAN_BWD
#define SVAE_TILE_LEAN_BWD 1
#endif
  constexpr bool lean_bwd = (WPC == 2) && SVAE_TILE_LEAN_BWD;
  if constexpr (!lean_bwd) prefetch_step(wsb + (long)(T - 1) * WSTEP);
  __syncthreads();

  // SVAE_KEEP_SIGMA: Sigma_t, compact n x n, for the VJP (svae_lds_tile_sigma_offset_bytes): thread (ty, tx) moves columns
  // 4 tx .. 4 tx + 3 of rows ty, ty + 16, ..  (32-byte stores when n is a multiple of 4)
  auto store_sigma = [&](int ts) {
    double* dst = a.sig_out + ((long)b * T + ts) * nn;
    const int ty = tid >> 4, c0 = 4 * (tid & 15);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = ty + 16 * i;
      if (r < n && c0 < n) {
        const d4 v = *(const d4*)(M + r * LDM + c0);
        if ((n & 3) == 0) {
          *(d4*)(dst + (long)r * n + c0) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (c0 + j < n) dst[(long)r * n + c0 + j] = v[j];
        }
      }
    }
  };
  // Like the forward half, instantiated per wavefront index (= tile column J of the wavefront).
  auto backward = [&](auto jc) {
  constexpr int J = decltype(jc)::value;
  const int mycol = 16 * J + r16;
  for (int t = T - 1; t >= 0; --t) {
    TICK(9)
    const double* w = wsb + (long)t * WSTEP;
    if constexpr (lean_bwd) prefetch_step(w);
#pragma unroll
    for (int u = 0; u < RL4B; ++u) {
      const int c4 = tid + 256 * u;
      if (c4 * 4 < NP * NP) *(d4*)(M + ((c4 * 4) / NP) * LDM + NP + ((c4 * 4) % NP)) = xr[u];
    }
    if (tid < NP) hvec[tid] = cpre;
    __syncthreads();
    if (a.sig_out && t + 1 < T) store_sigma(t + 1);     // (the panel's P columns hold Sigma_{t+1} until B2 of this step)
    if (t + 1 < T && tid < n) {          // node statistics of step t+1 (Sigma_{t+1} is complete now)
      const double mm = mold[tid];
      oEx[(long)(t + 1) * n + tid] = mm;
      oExx[(long)(t + 1) * n + tid] = __builtin_fma(mm, mm, M[tid * LDM + tid]);
    }
    // B2 forms each symmetric tile pair of Sigma_t ONCE: wavefront J (= tile column J) computes the tile rows
    // i = (J + q) mod NB, q < CNT (cyclic assignment: NB = 4 -> 3, 3, 2, 2 tiles instead of 4 each), and stores the
    // tile at (i, J) and transposed at (J, i).  P_t^-1 is read for those tiles only.
    constexpr int CNT = J < NB ? sym_cnt(NB, J) : 0;
    d4 pin[CNT > 0 ? CNT : 1];           // P_t^-1 tiles (i, J): consumed in B2
    auto load_pin_i = [&](int i) -> d4 {
      const double* pi = w + NP * NP + (16 * i + kq) * NP + mycol;
      return d4{pi[0], pi[4 * NP], pi[8 * NP], pi[12 * NP]};
    };
    if constexpr (!lean_bwd) {
#pragma unroll
      for (int q = 0; q < CNT; ++q) pin[q] = load_pin_i((J + q) % NB);
    }
    TICK(10)
    {  // m_t = c_t + X_t m_{t+1}
      const int row = tid >> 2, part = tid & 3;
      double s = 0.0;
      if (row < NP) {
        const double* xrow = M + row * LDM + NP;
        for (int cc = part; cc < NP; cc += 4) s = __builtin_fma(xrow[cc], mold[cc], s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (row < NP && part == 0) mnew[row] = hvec[row] + s;
    }
    d4 Wt[NB];
    if constexpr (J < NB) {
      d4 Bx[NB];
#pragma unroll
      for (int l = 0; l < NB; ++l) Bx[l] = frag_a(M, LDM, 16 * J, NP + 16 * l, r16, kq);   // (X_{jl})' as B
#pragma unroll
      for (int kk = 0; kk < NB; ++kk) {
        d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < NB; ++l) c = mma16(frag_a(M, LDM, 16 * kk, 16 * l, r16, kq), Bx[l], c);
        Wt[kk] = c;                            // W[kk][j] = (Sigma_{t+1} X')  = Cov(x_{t+1}, x_t)
      }
    }
    __syncthreads();
    TICK(11)

    d4 pnext = {0.0, 0.0, 0.0, 0.0};             // lean: P_t^-1 tile one computed tile ahead
    if constexpr (lean_bwd) { if constexpr (J < NB) pnext = load_pin_i(J); }
    else prefetch_step(t > 0 ? w - WSTEP : w);   // step t-1, in flight during B2
    if constexpr (J < NB) {
      const double mc = mnew[mycol];
      double* oP = INHOMOG && t < T - 1 ? a.E_pair + ((long)b * (T - 1) + t) * 3 * nn : nullptr;
      const bool rare = INHOMOG || t == 0 || t == T - 1;
      d4 fx[NB];                           // A fragments of X tile row i, fetched one tile ahead (lean: in place)
      if constexpr (!lean_bwd) {
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) fx[kk] = frag_a(M, LDM, 16 * J, NP + 16 * kk, r16, kq);
      }
      static_for<0, CNT>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int i = (J + q) % NB, inx = (J + q + 1) % NB;
        d4 fn[lean_bwd ? 1 : NB];
        if constexpr (!lean_bwd) {
#pragma unroll
          for (int kk = 0; kk < NB; ++kk) fn[kk] = (q + 1 < CNT) ? frag_a(M, LDM, 16 * inx, NP + 16 * kk, r16, kq) : fx[kk];
        }
        d4 c;
        if constexpr (lean_bwd) { c = pnext; if constexpr (q + 1 < CNT) pnext = load_pin_i(inx); }
        else c = pin[q];
        if constexpr (lean_bwd) {        // fragments read where they are consumed (one live at a time)
#pragma unroll
          for (int kk = 0; kk < NB; ++kk) c = mma16(frag_a(M, LDM, 16 * i, NP + 16 * kk, r16, kq), Wt[kk], c);
        } else {
#pragma unroll
          for (int kk = 0; kk < NB; ++kk) c = mma16(fx[kk], Wt[kk], c);
        }
        SVAE_SGB(3, 4 * NB, 1, 0)
        if constexpr (!lean_bwd) {
#pragma unroll
          for (int kk = 0; kk < NB; ++kk) fx[kk] = fn[kk];
        }
        store_c(M, LDM, 16 * i, 16 * J, r16, kq, c);          // Sigma_t tile (i, J) ...
        if constexpr (i != J) store_ct(M, LDM, 16 * J, 16 * i, r16, kq, c);   // ... and its mirror image (J, i)
        d4 exx;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) exx[qq] = __builtin_fma(mnew[16 * i + 4 * qq + kq], mc, c[qq]);   // E[x_t x_t'](row, mycol)
#ifndef SVAE_TILE_TIMING
        if (rare) {
          // (homogeneous model: stores at the two ends of the chain only.  The opaque copy of the column index keeps
          //  their loop-invariant 64-bit addresses from being hoisted out of the time loop into live registers)
          int mcx = mycol;
          if constexpr (!INHOMOG) asm volatile("" : "+v"(mcx));
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int row = 16 * i + 4 * qq + kq;
            const bool in = row < n && mcx < n;
            const int p1 = row * n + mcx, p2 = mcx * n + row;       // the element and its mirror image
            if (INHOMOG) {
              if (oP && in) {
                oP[p1] = exx[qq];
                oP[2 * nn + p1] = prevE[q][qq];                     // E[x_{t+1} x_{t+1}'] (previous step)
                if constexpr (i != J) { oP[p2] = exx[qq]; oP[2 * nn + p2] = prevE[q][qq]; }
              }
            } else if (in) {
              // sum_{t>=1} E[x_t x_t'] = sum_{t<=T-2} + last - first: the last term waits in its output slot
              if (t == T - 1) oPh[2 * nn + p1] = exx[qq];
              if (t == 0) {
                const double s0 = sxx[q][qq] + (T > 1 ? exx[qq] : 0.0);
                const double s2 = s0 + oPh[2 * nn + p1] - exx[qq];
                oPh[p1] = s0;
                oPh[2 * nn + p1] = s2;
                if constexpr (i != J) { oPh[p2] = s0; oPh[2 * nn + p2] = s2; }
              }
            }
            if (t == 0 && in) {
              oI[p1] = exx[qq];
              if constexpr (i != J) oI[p2] = exx[qq];
            }
          }
        }
#endif
        if (INHOMOG) prevE[q] = exx;
        else if (t < T - 1) sxx[q] += exx;
      });
      // cross moments E[x_{t+1} x_t'](row, mycol) = W + m_{t+1} m_t': every tile row of the wavefront's column
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        d4 ecr;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) ecr[qq] = __builtin_fma(mold[16 * i + 4 * qq + kq], mc, Wt[i][qq]);
#ifndef SVAE_TILE_TIMING
        if (rare) {
          int mcx = mycol;
          if constexpr (!INHOMOG) asm volatile("" : "+v"(mcx));
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int row = 16 * i + 4 * qq + kq;
            const bool in = row < n && mcx < n;
            if (INHOMOG) {
              if (oP && in) oP[nn + mcx * n + row] = ecr[qq];
            } else if (in && t == 0) {
              oPh[nn + mcx * n + row] = cross[i][qq] + ecr[qq];
            }
          }
        }
#endif
        if (!INHOMOG && t < T - 1) cross[i] += ecr;
      }
    }
    double* tmp = mold; mold = mnew; mnew = tmp;
    __syncthreads();
    TICK(9)
  }
  };
  switch (wave) {
    case 0: backward(std::integral_constant<int, 0>{}); break;
    case 1: backward(std::integral_constant<int, 1>{}); break;
    case 2: backward(std::integral_constant<int, 2>{}); break;
    default: backward(std::integral_constant<int, 3>{}); break;
  }
#ifdef SVAE_TILE_TIMING
  if (lane == 0 && wave < 2) { for (int q = 0; q < 12; ++q) a.E_init[(long)b * (nn + n) + 12 * wave + q] = (double)tm[q]; }
  return;
#endif
  if (a.sig_out) store_sigma(0);
  if (tid < n) {                          // node statistics of step 0, E[x_0]
    const double mm = mold[tid];
    oEx[tid] = mm;
    oExx[tid] = __builtin_fma(mm, mm, M[tid * LDM + tid]);
    oI[nn + tid] = mm;
  }
}


// Packs the pair parameters of one step (slot) into the register order of the main kernel:
//   pA[((i NB + kk) 64 + lane) 4 + kb] = -J12[16 kk + 4 kb + kq][16 i + r16]       (A operand of -(J12'))
//   pC[((i NB + j) 64 + lane) 4 + qq]  = -2 (J22 + w J11n)[16 i + 4 qq + kq][16 j + r16], identity on the padding
//   pR[row NP + col]                   = J12n[row][col] (zero-padded; all zero when the next step is the last)
// Homogeneous parameters: slot 0 = regular step, slot 1 = the step before the last one (no J11 term,
// zero right-hand side).  Per-step parameters: slot t = transition t -> t+1 (t = 0 .. T-2).
template <int NB>
__global__ __launch_bounds__(256) void tile_pack_pairs_kernel(const double* __restrict__ J11,
                                                              const double* __restrict__ J12,
                                                              const double* __restrict__ J22,
                                                              int n, int T, int inhomog, long set_stride,
                                                              double* __restrict__ out) {
  constexpr int NP = 16 * NB;
  const int slot = blockIdx.x, set = blockIdx.y;
  const int nslots = inhomog ? T - 1 : 2;
  const long nn = (long)n * n;
  const int t = inhomog ? slot : 0;
  const bool next_last = inhomog ? (t + 1 == T - 1) : (slot == 1);
  const double* j12 = J12 + set * set_stride + (long)t * nn;
  const double* j22 = J22 + set * set_stride + (long)t * nn;
  const double* j11n = J11 + set * set_stride + (long)(inhomog && !next_last ? t + 1 : t) * nn;
  const double* j12n = J12 + set * set_stride + (long)(inhomog && !next_last ? t + 1 : t) * nn;
  double* o = out + ((long)set * nslots + slot) * (3 * NP * NP);
  for (int e = threadIdx.x; e < NP * NP; e += 256) {
    const int qq = e & 3, lane = (e >> 2) & 63, tile = e >> 8;
    const int r16 = lane & 15, kq = lane >> 4, ti = tile / NB, tj = tile % NB;
    {  // pA: tile (i = ti, kk = tj), kb = qq
      const int row = 16 * tj + 4 * qq + kq, col = 16 * ti + r16;
      o[e] = (row < n && col < n) ? -j12[row * n + col] : 0.0;
    }
    {  // pC: tile (i = ti, j = tj)
      const int row = 16 * ti + 4 * qq + kq, col = 16 * tj + r16;
      double v = (row == col) ? 1.0 : 0.0;
      if (row < n && col < n) v = -2.0 * j22[row * n + col] - (next_last ? 0.0 : 2.0 * j11n[row * n + col]);
      o[NP * NP + e] = v;
    }
    {  // pR
      const int row = e / NP, col = e % NP;
      o[2 * NP * NP + e] = (!next_last && row < n && col < n) ? j12n[row * n + col] : 0.0;
    }
  }
}

template <int NB>
static int launch_tile(const LdsArgs& a, int n, int inhomog, hipStream_t s) {
  constexpr int NP = 16 * NB;
  const size_t lds = TileCfg<NB>::LDS_DOUBLES * sizeof(double);
  const int T = a.T;
  // workspace: [hand-off region: B T WSTEP][packed pair parameters]
  double* pk = a.ws + (size_t)a.B * T * TileCfg<NB>::WSTEP;
  const int batched = a.pair_seq_stride != 0;
  if (T > 1 && a.tile_half != 2) {        // (the backward half reads the hand-off only)
    const int nslots = inhomog ? T - 1 : 2;
    hipLaunchKernelGGL((tile_pack_pairs_kernel<NB>), dim3(nslots, batched ? a.B : 1), dim3(256), 0, s,
                       a.J11, a.J12, a.J22, n, T, inhomog, (long)a.pair_seq_stride, pk);
    if (hipGetLastError() != hipSuccess) return -1000;
  }
  static LdsGrant grants[6];          // per kernel instance (and per device inside)
  auto go = [&](auto kern, int which) {
    if (!grants[which].ensure((const void*)kern, (long)lds)) return -1001;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds, s, a, n, (const double*)pk, batched);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  };
  (void)NP;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // at most one workgroup per CU: the 512-register instance (both halves, or the one a.tile_half names); more: the
  // single-half instances sized for two workgroups per CU, the whole E-step as two launches
  if (a.B <= cus) return inhomog ? go(lds_estep_tile_kernel<NB, true, 1, 0>, 0) : go(lds_estep_tile_kernel<NB, false, 1, 0>, 1);
  int rc = 0;
  if (a.tile_half != 2)
    rc = inhomog ? go(lds_estep_tile_kernel<NB, true, 2, 1>, 2) : go(lds_estep_tile_kernel<NB, false, 2, 1>, 3);
  if (rc == 0 && a.tile_half != 1)
    rc = inhomog ? go(lds_estep_tile_kernel<NB, true, 2, 2>, 4) : go(lds_estep_tile_kernel<NB, false, 2, 2>, 5);
  return rc;
}

}  // namespace svae

extern "C" size_t svae_lds_tile_step_doubles(int n) {
  const int NP = 16 * ((n + 15) / 16);
  return (size_t)2 * NP * NP + NP;
}

/* doubles of packed pair parameters behind the hand-off region */
extern "C" size_t svae_lds_tile_packed_doubles(int B, int T, int n, int inhomog, int pair_batched) {
  const int NP = 16 * ((n + 15) / 16);
  if (T < 2) return 0;
  const size_t nslots = inhomog ? (size_t)(T - 1) : 2;
  return (pair_batched ? (size_t)B : 1) * nslots * 3 * NP * NP;
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
