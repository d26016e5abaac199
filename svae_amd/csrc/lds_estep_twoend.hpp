// lds_estep_twoend.hpp -- two-ended ("burn at both ends") LDS E-step for MI355X, latent dim n <= 10.
//
// Same contract and results as lds_estep_kernel.hpp (filter + RTS smoother + expected statistics +
// log-normaliser: cython_lds_inference.pyx:28-90, 149-210), different schedule.  The chain
// x_0 .. x_{T-1} is a Markov chain in both directions, so block elimination may start at BOTH ends:
//   chain A eliminates x_0, x_1, ...        (the reference's forward filter, natural_filter_forward_general)
//   chain B eliminates x_{T-1}, x_{T-2}, ... (the same filter on the time-reversed chain: pair blocks
//                                            (J11,J12,J22) -> (J22,J12',J11), no initial potential)
// and the two meet in the middle: the node where they meet receives both messages, its marginal is
// one more inversion, and from there each chain runs its own moment-form smoother back to its end.
// The serial dependency chain is T/2 eliminations + T/2 smoother steps instead of T + T, and the two
// chains of a sequence advance in ONE instruction stream:
//   * one sequence per wavefront; DPP rows 0,1 carry chain A, rows 2,3 chain B (all arithmetic is
//     row-local, so the same instruction serves both chains);
//   * the two DPP rows of a chain share every product stage by output row (row i of a tile lives in
//     DPP row i & 1, "slot" i >> 1) and re-replicate slot tiles with v_permlane16_swap (gfx950): no LDS,
//     no waitcnt on the critical path;
//   * the Gauss-Jordan works on ONE register per matrix row: lanes 0..n-1 hold P (replicated in the
//     two DPP rows), lanes n..14 hold the right-hand-side columns J12[:, x] (column x in DPP row x & 1,
//     lane n + (x >> 1)), lane 15 the potential vector h -- one v_fmac_f64_dpp updates a whole row of
//     [P | J12 | h].  The inverse's columns are kept UNSCALED during the elimination (column k holds
//     the multipliers f_i, the pivot row -1) and scaled by -1/p_k once at the end: exact (no 1 + 1/p
//     rounding), one instruction per row update.
//   * log|P_t| and the positive-definiteness check come from the vector of -1/p_k (one multiply and
//     one max per step instead of per pivot).
// Hand-off to the smoother: per (chain, step) n rows [P^-1 row | (P^-1 J12) row | c_i | pad] in an HBM
// workspace written and re-read by the same lanes' wavefront (layout te_* below).
//
// Used for keep == 0 (no sampler / VJP hand-off: those follow the reference's one-directional
// factorisation, whose LDL' factors define the eps -> sample map), 4 <= T, n <= 10.
#pragma once
#include "lds_estep_kernel.hpp"
#include "per_device.hpp"
#include "gj1r_gen.hpp"
#include "lds_estep_twoend_s4.hpp"

#ifndef SVAE_TE_LDS_SPLIT
#define SVAE_TE_LDS_SPLIT 3     // bit 0: kernels without the cross-moment store, bit 1: CROSS (runs next to the filter kernel)
#endif

namespace svae {

// which variants re-replicate the next pivot block through LDS (else: v_permlane16_swap shuffles); MIX never does
template <bool MIX, bool CROSS> constexpr bool TE_LDS_SPLIT() { return !MIX && (((SVAE_TE_LDS_SPLIT) >> (CROSS ? 1 : 0)) & 1); }

__device__ __forceinline__ double asm_sub(double a, double b) {        // a - b, kept in program order
  double r;
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ double asm_max(double a, double b) {        // fmax as ONE v_max_f64 (hipcc canonicalises both operands first)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ---- in-place Gauss-Jordan on one register per row --------------------------------------------------
// M[i]: lanes < N = row i of P (SPD), other lanes = right-hand-side columns.  On exit lanes < N hold
// the inverse in "unscaled column" form (true inverse = M[i][c] * v[c], v[c] = -1/p_c accumulated in
// vfull, whose other lanes are left untouched), the other lanes P^-1 * rhs.  qacc += rhs_k^2 / p_k per
// lane (lane 15: h' P^-1 h).  Software pipelining as in gauss_jordan<> (lds_estep_kernel.hpp): the
// reciprocal chain of the next pivot is issued between the row updates of the current one.
template <int N>
__device__ __forceinline__ void gauss_jordan_1r(double (&M)[N], const double (&E)[N], double& qacc,
                                                double& vfull) {
  double p = bcast_fenced<0>(M[0]);
  double rinv = rcp_nr(p);
  static_for<0, N>([&](auto k) {
    const double mk0 = __builtin_fma(-p, E[k], M[k]);      // lane k -> exactly 0
    const double ru = mk0 * rinv;                          // scaled pivot row (lane k: 0)
    qacc = __builtin_fma(M[k], ru, qacc);
    vfull = __builtin_fma(-rinv, E[k], vfull);             // lane k <- -1/p_k
    auto update = [&](auto i, auto fenced) {
      mac_bc<k, true, decltype(fenced)::value>(M[i], M[i], ru);   // lane k keeps the multiplier f_i
    };
    if constexpr (k + 1 < N) {
      update(std::integral_constant<int, k + 1>{}, std::true_type{});
      const double pn = bcast_fenced<k + 1>(M[k + 1]);
      const double stored = asm_sub(ru, E[k]);             // pivot row as kept: lane k = -1
      double t0 = 0.0, e0 = 0.0, t1 = 0.0, e1 = 0.0, rn = 0.0;
      constexpr int REM = N - 2;
      auto chain = [&](auto s) {
        if constexpr (s == 0) t0 = asm_rcp(pn);
        else if constexpr (s == 1) e0 = asm_fnma1(pn, t0);
        else if constexpr (s == 2) t1 = asm_fma(t0, e0, t0);
        else if constexpr (s == 3) e1 = asm_fnma1(pn, t1);
        else if constexpr (s == 4) rn = asm_fma(t1, e1, t1);
      };
      if constexpr (REM == 0) static_for<0, 5>(chain);
      static_for<0, N>([&](auto i) {
        if constexpr (i != k && i != k + 1) {
          constexpr int pos = i - (i > k ? 1 : 0) - (i > k + 1 ? 1 : 0);
          update(i, std::false_type{});
          constexpr int lo = pos * 5 / REM, hi = (pos + 1) * 5 / REM;
          static_for<lo, hi>(chain);
        }
      });
      M[k] = stored;
      p = pn;
      rinv = rn;
    } else {
      const double stored = asm_sub(ru, E[k]);
      static_for<0, N>([&](auto i) {
        if constexpr (i != k) {
          if constexpr (i == 0 || (k == 0 && i == 1)) update(i, std::true_type{});
          else update(i, std::false_type{});
        }
      });
      M[k] = stored;
    }
  });
}

#ifndef SVAE_GJ_GENERATED
#define SVAE_GJ_GENERATED 1   // 1: one hand-scheduled asm block per pivot (gj1r_gen.hpp, tools/gen_gj_asm.py);
#endif                        // 0: the statement-by-statement version above (same arithmetic), for A/B
template <int N>
__device__ __forceinline__ void te_gauss_jordan(double (&M)[N], const double (&E)[N], double& qacc, double& vfull) {
#if SVAE_GJ_GENERATED
  gauss_jordan_1r_asm<N>(M, E, qacc, vfull);
#else
  gauss_jordan_1r<N>(M, E, qacc, vfull);
#endif
}

#ifdef SVAE_PHASE_TIMING
#define TE_TICK(i) { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tlast_; tlast_ = now_; }
#else
#define TE_TICK(i)
#endif

// LEAN hand-off (the default): per (chain, step) only the lower triangle of P^-1 and c = P^-1 h_filt
// (te_lean_step_doubles: 68 doubles at n = 10 instead of 220); the smoother rebuilds G = -P^-1 J12 with one
// split product per step and transposes it through 3 KB of LDS.  Algorithmic HBM bytes x ~4 instead of x 11.
//
// MIX (svae_slds_lds_meanfield_f64, the SLDS mean field: /root/reference/svae/models/slds_svae.py:92-103,
// 131-147): the pair parameters of step t are sum_k w[b,t+1,k] * P_k for K parameter sets resident in LDS
// (mixed just in time, straight into the registers that consume them: M, Bt, the Schur accumulators), and
// instead of per-step E_pair blocks the smoother contracts its tiles with the K sets on the spot,
//   out[b,t,0,k] = <E x_t x_t', J11_k> (+ <E x_t x_{t+1}', J12_k> on chain A)      -> pair t
//   out[b,t,1,k] = <E x_t x_t', J22_k> (+ <E x_{t-1} x_t', J12_k> on chain B)      -> pair t-1
// (lane k of the DPP row accumulates state k: the lane reduction is the broadcast of the DPP operand).
// W sequences (wavefronts) per workgroup share the tables; `seq_index` lists the rows a launch works on.
//
// CROSS: also write the cross moments W~_t = E[x~_{t+1} x~_t'] ((n+1) x (n+1), t = 0 .. T-1, the record of
// t = T-1 being e_n [mu_{T-1}; 1]') in the layout the reverse-mode sweeps read (a.ws3; lds_vjp_kernel.hpp).
// They are posterior moments -- the same whichever way the chain was eliminated -- so the training step may take
// statistics and cross moments from this kernel while the one-directional FILTER (whose factorisation defines the
// sampler's eps -> sample map and the hand-off the sweeps differentiate) runs concurrently on the SIMDs a small
// batch leaves idle (lds_estep.hip: svae_lds_estep_f64 with keep != 0).
//
// S4 (small batches, homogeneous lean variant): the workgroup has a SECOND wavefront that waits at a barrier through
// the elimination phase and then runs chain B's smoother while this one runs chain A's, each with four DPP rows per
// chain (lds_estep_twoend_s4.hpp).
// (the body as a device function of the workgroup index `blk`: lds_forward_pair_kernel, lds_filter_1r.hpp, runs it next
//  to the one-directional filter in ONE launch)
// RING (round 5; the one-sequence consumer of lds_estep_twoend_rpcmix.hpp's producer wavefronts): per-step parameters of
// the SLDS mean field as in MIX, but mixed by PRODUCER wavefronts of the same workgroup into an LDS ring (`ring_m`: per
// column [nat J12 (n^2) | C lower triangle | zeros], columns 0 / 1 = chain A / B, slot = local step & 1), the node's
// tiles for the K-state contraction left in a second ring (`ring_s`, `ring_depth` slots: [S~ lower triangle | junk | W~]);
// lean records; one LDS-only barrier per step, 2 e + 3 in all (the producers execute the same count).
struct TeRing {
  double* m; double* s;      // mixing ring, tile ring (LDS)
  int cs, mixslot;           // doubles: column stride / slot stride of the mixing ring
  int ss, sslot, woff;       // ... of the tile ring; offset of W~ in a column
  int depth;                 // tile-ring slots in use
};

template <int N, bool INHOMOG, bool LEAN, bool MIX = false, bool CROSS = false, bool S4 = false, bool RING = false>
__device__ __forceinline__ void lds_estep_twoend_body(const LdsArgs& a, const int blk, const TeRing ring = TeRing{}) {
  static_assert(!S4 || (LEAN && !INHOMOG && !MIX && !CROSS), "S4: the homogeneous lean variant only");
  static_assert(!RING || (INHOMOG && LEAN && !MIX && !CROSS && !S4), "RING: per-step parameters from the ring, lean records");
  static_assert(N >= 1 && N <= TE_MAX_N && N + ((N - 1) >> 1) <= 14,
                "the right-hand-side columns must fit lanes N..14 of two DPP rows");
  static_assert(!MIX || (INHOMOG && !LEAN), "MIX: per-step parameters, full hand-off record");
  // (CROSS runs next to the one-directional filter: one wavefront per SIMD, see lds_estep_split.hpp)
  // (S4: 2 B wavefronts on 1024 SIMDs -- without the marker two of them may share a SIMD: 180 us instead of 152)
  if constexpr (CROSS || S4) asm volatile("v_accvgpr_write_b32 a63, 0" ::: "a63");
  constexpr int RW = te_row_doubles(N), ZP = te_page_doubles(N);
  constexpr int WS = LEAN ? te_lean_step_doubles(N) : te_step_doubles(N);
  constexpr int TRI = N * (N + 1) / 2;    // LEAN record: [lower triangle | c (N) | 0.0 | trash | pad]
  constexpr int LZERO = TRI + N, LTRASH = TRI + N + 1;
  constexpr int J = (N + 1) / 2;          // slots holding rows 0..N-1 (row i = 2j + gl)
  constexpr int J1 = (N + 2) / 2;         // slots holding rows 0..N
  constexpr int HL = 15;                  // lane of the h column
  constexpr int RSL = (N + 3) & ~1;       // LDS row stride of the transposition tile (even, >= N + 1)
  __shared__ double tab_static[MIX ? 2 : (S4 ? 4 : 2) * 16 * 16];   // [chain][row 0..15][RSL]: G~ rows for the transposed read (S4: + the gather tiles)
  __shared__ double xch_static[S4 ? 2 * ((N + 3) / 4) * 64 + 256 : 2];   // S4: chain B's sums on their way to chain A
  extern __shared__ double2 te_dyn[];     // MIX: the parameter tables (te_mix_lds_bytes)

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int dir = g >> 1;                 // 0: chain A (forward in time), 1: chain B (reversed)
  const int gl = g & 1;                   // DPP row within the chain's pair
  // one sequence per wavefront.  MIX: consecutive slots go to DIFFERENT workgroups (slot = wavefront * workgroups +
  // workgroup): a launch sized for more slots than are live (negative seq_index entries, see below) then leaves one live
  // wavefront per CU instead of eight sharing a CU's LDS bandwidth
  const int bslot = MIX ? wv * (int)gridDim.x + blk : blk;
  // MIX: row of every array (and of the workspace) this launch slot works on (surplus slots: any valid row)
  // (a negative entry of seq_index marks an unused slot: its wavefront helps stage the tables, then leaves)
  const int braw = (!MIX && !RING) ? bslot : (a.seq_index ? a.seq_index[bslot < a.B ? bslot : 0] : bslot);
  const int b = braw < 0 ? 0 : braw;
  double* tab = tab_static;               // (MIX keeps the full hand-off record: no transposition tile)
  // re-replication tile of the elimination phase, [chain][row 0..N][16]: S4 borrows the exchange buffer (used at the very
  // end), the other variants the transposition tiles (used by the smoother phase only, zeroed again in between)
  double* const split_tile = S4 ? xch_static : tab_static;
  static_assert(MIX || 2 * (N + 1) * 16 <= (S4 ? 2 * ((N + 3) / 4) * 64 + 256 : 2 * 16 * 16), "split tile fits");
  // S4: the last `keep` records of each chain stay in (dynamic) LDS -- the smoother reads them first -- instead of
  // travelling through HBM: [chain][slot][WS] doubles, slot = local step - (e + 1 - keep)
  double* const lrecs = reinterpret_cast<double*>(te_dyn);
  int keep = 0;
  if constexpr (S4) {
    const int e_ = te_elims(a.T);
    keep = a.lds_keep < e_ + 1 ? a.lds_keep : e_ + 1;
    if (keep < 3) keep = 0;
    if (wv == 1) {                        // chain B's smoother wavefront: sleeps until the records are complete
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      te_smooth4<N>(a, b, 1, lane, tab_static + 256, tab_static + 768, xch_static, lrecs, keep);
      return;
    }
  }
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int T = a.T;
  const int e = te_elims(T);              // eliminations per chain; the meeting node is local index e
  const int jx = T - 1 - e;               // eliminations done when the partner's message is taken
  const bool oddT = (T & 1) != 0;

  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EH = (c == HL) ? 1.0 : 0.0;
  const double EN = (c == N) ? 1.0 : 0.0;

  // ---- MIX: the K parameter sets as LDS tables, one 16-byte entry = states (2kp, 2kp+1) -----------------
  //   t_j[kp][i][ENT_J]    ONE table for the two places J12' enters a step, whose lanes are disjoint:
  //                        right-hand-side lanes: -J12'_k[i][x]          (entry (dir*2+gl)*NXL + c-N)
  //                        lanes < N (Schur operand): nat J12'_k[i][c]   (entry 4 NXL + dir*N + c)
  //   t_22, t_11[kp][j][ENT_C]  lanes < N: -2 q22_k / -2 q11_k [2j+gl][c] (entry (dir*2+gl)*N + c)
  //   the last entry of every row is zero (lanes with nothing to add);
  //   ct[blk][j][gl][c/2][16]   entry = (P_k[2j+gl][c], P_k[2j+gl][c+1]) in lane k: blk 0 J11, 1 J22,
  //                             2 J12 transposed (chain A's cross moment), 3 J12 (chain B's)
  constexpr int NXL = te_mix_nxl(N);
  constexpr int ENT_J = te_mix_ent_j(N), ENT_C = te_mix_ent_c(N);
  constexpr int NC2 = (N + 1) / 2;
  const int K = (MIX || RING) ? a.mix_K : 0, KP2 = (K + 1) >> 1;
  double2* t_j = te_dyn;
  double2* t_22 = t_j + KP2 * N * ENT_J;
  double2* t_11 = t_22 + KP2 * J * ENT_C;
  double2* ct = t_11 + KP2 * J * ENT_C;
  if constexpr (MIX) {
    auto P = [&](const double* base, int k, int idx) { return k < K ? base[(long)k * N * N + idx] : 0.0; };
    const int nth = blockDim.x, tid = threadIdx.x;
    for (int q = tid; q < KP2 * N * ENT_J; q += nth) {
      const int ent = q % ENT_J, i = (q / ENT_J) % N, kp = q / (ENT_J * N);
      double2 v = make_double2(0.0, 0.0);
      if (ent < 4 * NXL) {
        const int d = ent / (2 * NXL), r = (ent / NXL) & 1, x = 2 * (ent % NXL) + r;
        if (x < N) {
          const int idx = d ? x * N + i : i * N + x;
          v = make_double2(-P(a.J12, 2 * kp, idx), -P(a.J12, 2 * kp + 1, idx));
        }
      } else if (ent < 4 * NXL + 2 * N) {
        const int e2 = ent - 4 * NXL, d = e2 / N, cq = e2 % N;
        const int idx = d ? cq * N + i : i * N + cq;
        v = make_double2(P(a.J12, 2 * kp, idx), P(a.J12, 2 * kp + 1, idx));
      }
      t_j[q] = v;
    }
    for (int q = tid; q < KP2 * J * ENT_C; q += nth) {
      const int ent = q % ENT_C, j = (q / ENT_C) % J, kp = q / (ENT_C * J);
      double2 v22 = make_double2(0.0, 0.0), v11 = v22;
      if (ent < 4 * N) {
        const int d = ent / (2 * N), r = (ent / N) & 1, cq = ent % N, i = 2 * j + r;
        if (i < N) {
          const double* Q22 = d ? a.J11 : a.J22;
          const double* Q11 = d ? a.J22 : a.J11;
          v22 = make_double2(-2.0 * P(Q22, 2 * kp, i * N + cq), -2.0 * P(Q22, 2 * kp + 1, i * N + cq));
          v11 = make_double2(-2.0 * P(Q11, 2 * kp, i * N + cq), -2.0 * P(Q11, 2 * kp + 1, i * N + cq));
        }
      }
      t_22[q] = v22;
      t_11[q] = v11;
    }
    for (int q = tid; q < 4 * J * 2 * NC2 * 16; q += nth) {
      const int k = q & 15, cp = (q >> 4) % NC2, r = (q / (16 * NC2)) & 1, j = (q / (32 * NC2)) % J, blk = q / (32 * NC2 * J);
      const int i = 2 * j + r, c0 = 2 * cp, c1 = 2 * cp + 1;
      const double* base = blk == 0 ? a.J11 : (blk == 1 ? a.J22 : a.J12);
      auto at = [&](int cq) { return (i < N && cq < N) ? P(base, k, blk == 2 ? cq * N + i : i * N + cq) : 0.0; };
      ct[q] = make_double2(at(c0), at(c1));
    }
    __syncthreads();
    if (bslot >= a.B || braw < 0) return;
  }

  // ---- pair parameters in the chain's own orientation ---------------------------------------------
  // chain A: (J11, J12, J22); chain B: (J22, J12', J11).  Local pair l joins local nodes l, l+1
  // (global pair index l for A, T-2-l for B).
  const double* q11 = (dir ? a.J22 : a.J11) + (long)b * a.pair_seq_stride;
  const double* q22 = (dir ? a.J11 : a.J22) + (long)b * a.pair_seq_stride;
  const double* q12 = a.J12 + (long)b * a.pair_seq_stride;
  const int si = dir ? 1 : N, sc = dir ? N : 1;       // J12'[i][x] = J12[i * si + x * sc]
  auto pair_off = [&](int l) -> long { return INHOMOG ? (long)(dir ? T - 2 - l : l) * N * N : 0; };
  const int xq = 2 * (c - N) + gl;                    // right-hand-side column of this lane (c >= N)
  const bool xok = c >= N && c < HL && xq < N;
  const int xx = xok ? xq : 0;
  // MIX: this lane's table entries (registers and state pairs are immediate offsets from these)
  const double2* tj_l = t_j + (xok ? (dir * 2 + gl) * NXL + (c - N) : (col ? 4 * NXL + dir * N + c : 4 * NXL + 2 * N));
  const double rmask = xok ? 1.0 : 0.0, jcmask = col ? 1.0 : 0.0;
  const double2* c22_l = t_22 + (col ? (dir * 2 + gl) * N + c : 4 * N);
  const double2* c11_l = t_11 + (col ? (dir * 2 + gl) * N + c : 4 * N);
  const double2* cta_l = ct + gl * NC2 * 16 + c;
  const double2* ctc_l = cta_l + J * 2 * NC2 * 16;
  const double2* ctx_l = cta_l + (2 + dir) * J * 2 * NC2 * 16;
  // acc[r] += sum_k w_k tbl[kp][r][entry]   (w_k = lane k of wreg: the DPP broadcast does the indexing)
  // acc0[r] += sum_k w0_k t0[k][r],  acc1[r] += sum_k w1_k t1[k][r]   (r < R; tables [state pair][R][entries],
  // acc0 / acc1 may be the same array).  A real loop over the state pairs: the weights sit in the lanes of
  // w0 / w1 and are rotated two lanes per trip (row_ror:14: lane i <- lane i + 2), so that the DPP operand is
  // always lanes 0 and 1.  The table reads are software-pipelined BY HAND, half a trip ahead: inline-asm
  // statements are scheduling boundaries for hipcc, so a read written next to its FMAs is issued there and
  // waited for at once (measured: ~90 cycles per read, 7 k cycles per step; tools/te_mix_phase_timing.py).
  auto ror2 = [&](double x) { return __svae_update_dpp_f64(0.0, x, 0x12E, 0xf, 0xf, true); };
  // (chunk sizes R0 / R1 registers, ENT entries per register row, STRIDE entries per state pair in both tables)
  auto mix2 = [&](auto nreg0, auto nreg1, auto nent, auto stride, auto& acc0, auto& acc1, const double2* t0,
                  const double2* t1, double w0, double w1) {
    constexpr int R0 = decltype(nreg0)::value, R1 = decltype(nreg1)::value, ENT = decltype(nent)::value,
                  STRIDE = decltype(stride)::value;
    double2 va[R0], vb[R1 > 0 ? R1 : 1];
    static_for<0, R0>([&](auto r) { va[r] = t0[r * ENT]; });
    for (int kp = 0; kp < KP2; ++kp) {
      static_for<0, R1>([&](auto r) { vb[r] = t1[r * ENT]; });
      t1 += STRIDE;
      double wf[2] = {w0, w1};
      dpp_fence(wf);
      static_for<0, R0>([&](auto r) { mac_bc<0>(acc0[r], wf[0], va[r].x); });
      static_for<0, R0>([&](auto r) { mac_bc<1>(acc0[r], wf[0], va[r].y); });
      t0 += (kp + 1 < KP2) ? STRIDE : 0;            // last trip: re-reads its own group (unused)
      static_for<0, R0>([&](auto r) { va[r] = t0[r * ENT]; });
      static_for<0, R1>([&](auto r) { mac_bc<0>(acc1[r], wf[1], vb[r].x); });
      static_for<0, R1>([&](auto r) { mac_bc<1>(acc1[r], wf[1], vb[r].y); });
      w0 = ror2(wf[0]);
      w1 = ror2(wf[1]);
    }
  };
  const double wmask = (c < K) ? 1.0 : 0.0;
  const double* wrow = a.mix_w + (MIX ? (long)b * T * K + (c < K ? c : 0) : 0);    // + r*K: weights of node r
  auto wnode = [&](int l) { int r = dir ? T - 1 - l : l + 1; r = r < 0 ? 0 : (r > T - 1 ? T - 1 : r); return r; };   // node whose weights mix local pair l
  //   EX[i]: lanes < N identity row i; lane of column x: info-form J12'[i][x] = -nat J12'[i][x]
  //   NJ12c[k]: lanes < N: nat J12'[k][c] (= -J12'[k][c]); other lanes 0
  //   Cc[j] (row i = 2j+gl): lanes < N: info-form J22'(pair l) + J11'(pair l+1); other lanes 0
  double EX[N], NJ12c[N], Cc[J];
  auto load_pair = [&](int l, bool with_cc) {
    const long o = pair_off(l), o1 = pair_off(l + 1);
    static_for<0, N>([&](auto i) {
      const double rx = q12[o + i * si + xx * sc], rc = q12[o + i * si + cc * sc];
      EX[i] = xok ? -rx : E[i];
      NJ12c[i] = col ? rc : 0.0;
    });
    if (with_cc) static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl;
      const int ii = i < N ? i : 0;
      const double r22 = q22[o + ii * N + cc], r11 = q11[o1 + ii * N + cc];
      Cc[j] = (col && i < N) ? -2.0 * (r22 + r11) : 0.0;
    });
  };
  if (!INHOMOG) load_pair(0, true);
  // RING: this lane's ring entries (bytes within a mixing-ring slot; column = the chain).  rJ[i]: J12'[i][x] in the
  // right-hand-side lanes, J12'[i][c] in lanes < N (chain B: J12' = J12^T), a zero entry elsewhere; rC[j]: C[2j+gl][c]
  // from the triangle (lanes < N of real rows), else a zero entry
  constexpr int RTRI = N * (N + 1) / 2;
  unsigned rJ[RING ? N : 1], rC[RING ? J : 1];
  if constexpr (RING) {
    const unsigned cb = (unsigned)(dir * ring.cs), zero_e = (unsigned)(N * N + RTRI);
    static_for<0, N>([&](auto i) {
      const unsigned ent = xok ? (unsigned)(dir ? xx * N + i : i * N + xx) : (col ? (unsigned)(dir ? c * N + i : i * N + c) : zero_e);
      rJ[i] = 8u * (cb + ent);
    });
    static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl, hi = i > c ? i : c, lo = i > c ? c : i;
      rC[j] = 8u * (cb + (unsigned)((col && i < N) ? N * N + hi * (hi + 1) / 2 + lo : N * N + RTRI));
    });
  }
  auto ring_pair = [&](int slot, bool with_cc) {      // what load_pair does, from ring slot `slot`
    const char* base = reinterpret_cast<const char*>(ring.m) + 8u * (unsigned)(slot * ring.mixslot);
    static_for<0, N>([&](auto i) {
      const double v = *reinterpret_cast<const double*>(base + rJ[RING ? (int)i : 0]);
      EX[i] = __builtin_fma(-rmask, v, E[i]);         // right-hand-side lanes: info-form J12' = -nat; lanes < N: identity row
      NJ12c[i] = jcmask * v;
    });
    if (with_cc) static_for<0, J>([&](auto j) {
      Cc[j] = *reinterpret_cast<const double*>(base + rC[RING ? (int)j : 0]);
    });
  };

  // ---- elimination (filter) phase -------------------------------------------------------------------
  // An (replicated over the chain's two DPP rows): lanes < N = pivot block of the next node without its
  // node potential (incoming message + J11' of the pair ahead), lane 15 = incoming potential vector,
  // lanes N..14 zero.
  double An[N];
  if constexpr (MIX || RING) {
    // init potential mixed by E[z_0], J11' of the chain's pair 0 by its own node's weights (once per sequence:
    // straight from global memory)
    const double* w0 = a.mix_w + (long)b * T * K;
    const double* wq = w0 + (long)wnode(0) * K;
    static_for<0, N>([&](auto i) { An[i] = 0.0; });
    for (int k = 0; k < K; ++k) {
      const double wi = dir ? 0.0 : w0[k], wp = wq[k];
      const double* ij = a.init_J + (long)k * N * N, *ih = a.init_h + (long)k * N;
      const double* j11 = (dir ? a.J22 : a.J11) + (long)k * N * N;
      static_for<0, N>([&](auto i) {
        const double v = -2.0 * (wi * ij[i * N + cc] + wp * j11[i * N + cc]);
        An[i] += col ? v : ((c == HL) ? wi * ih[i] : 0.0);
      });
    }
  } else {
    const long o0 = pair_off(0);
    static_for<0, N>([&](auto i) {
      const double ij = a.init_J[i * N + cc], ih = a.init_h[i], j11 = q11[o0 + i * N + cc];
      An[i] = col ? -2.0 * ((dir ? 0.0 : ij) + j11) : ((c == HL && !dir) ? ih : 0.0);
    });
  }

  // node potentials of local step s: global node t = s (A) / T-1-s (B); lanes >= N read element 0
  const double* nJb = a.node_J + ((long)b * T) * N + cc;
  const double* nhb = a.node_h + ((long)b * T) * N + cc;
  auto node_off = [&](int s) -> long { return (long)(dir ? T - 1 - s : s) * N; };

  // chain workspace: constant page [e_N (N+2) | zeros (N+2) | trash (2)], then the records
  double* wsb = a.ws + (long)b * te_seq_doubles(N, T);            // uniform: this sequence's two chains
  const unsigned choff = (unsigned)(dir * te_chain_doubles(N, T)) + ZP;   // per lane: its chain's records
  double* zpage = wsb + (long)dir * te_chain_doubles(N, T);
  double* rec0 = zpage + ZP;
  double* trash = zpage + 2 * (N + 2);
  if (gl == 0) {
    if (c < N + 2) zpage[c] = EN;
    if (c < N + 2) zpage[N + 2 + c] = 0.0;
  }
  // hand-off store of register i: ONE unconditional instruction (a conditional one would make hipcc
  // wait for the previous step's stores at the loop head); lanes with nothing to hand over write the
  // record's trash / pad entry.
  //   full:  row i of the record = [P^-1 row | X row | c_i | pad]:       base + i * RW + stoff
  //   LEAN:  [lower triangle | c | 0.0 | trash]: lane c <= i -> tri(i) + c, lane 15 -> TRI + i
  const bool stp = (gl == 0 && col) || xok || (gl == 0 && c == HL);
  const int stoff = !stp ? 2 * N + 1 : (col ? c : (c == HL ? 2 * N : N + xx));
  unsigned loff[LEAN ? N : 1], lloff[S4 ? N : 1];
  if constexpr (LEAN) {
    static_for<0, N>([&](auto i) {
      const unsigned ent = (gl == 0 && c <= i) ? i * (i + 1) / 2 + c : ((gl == 0 && c == HL) ? TRI + i : LTRASH);
      loff[i] = 8u * (choff + ent);                              // bytes
      // (LDS: lanes with nothing to hand over get a dummy slot each behind the records -- fifty lanes storing to one
      //  trash address would serialise in its bank)
      if constexpr (S4) lloff[i] = 8u * (ent == (unsigned)LTRASH ? (unsigned)(2 * keep * WS) + lane : (unsigned)(dir * keep * WS) + ent);
    });
    if constexpr (S4) {
      for (int r = gl * 16 + c; r < keep; r += 32) lrecs[(long)(dir * keep + r) * WS + LZERO] = 0.0;   // the LDS records' zero entry
    }
    for (int q = lane; q < 2 * 16 * 16; q += 64) tab[q] = 0.0;     // rows of the transposition tile never written
    for (int r = gl * 16 + c; r <= e; r += 32) rec0[(long)r * WS + LZERO] = 0.0;   // the records' zero entry
  }

  double qacc = 0.0;        // lane 15: sum_t h' P^-1 h
  double ldM = 1.0;         // per lane c < N: running product of -1/p_c (log|P| = -sum log|.|)
  int ldE = 0;
  double vworst = -1.0;     // max over steps of -1/p_c (>= 0 <=> some pivot was not positive)

  // node potentials are prefetched one step ahead.  (hipcc merges the memory-counter state of the loop
  // entry with the back edge's, so the wait at the loop head also covers the previous step's hand-off
  // stores; loading by inline asm with a hand-placed s_waitcnt vmcnt(N) was tried and is NOT safe: the
  // compiler may copy the destination register before the wait.)
  // (walking per-lane pointers: recomputing node_off(s + 1) per step was two 64-bit multiply-adds and three selects)
  const double* pJn = nJb + node_off(0);
  const double* phn = nhb + node_off(0);
  const long nstep = dir ? -(long)N : (long)N;
  double Jo_n = *pJn;
  double ho_n = *phn;
  const double jm2 = col ? -2.0 : 0.0, jadd = col ? 0.0 : 1.0;      // JoX = col ? -2 Jo : 1 as ONE multiply-add
  double Mp[N];             // partner chain's An at the hand-over point
  static_for<0, N>([&](auto i) { Mp[i] = 0.0; });
  // ... and this chain's log-normaliser accumulators at that point: with even T the partner's last
  // elimination removes this chain's meeting node a second time, so the partner counts the snapshot
  double qacc_s = 0.0, ldM_s = 1.0;
  int ldE_s = 0;
  auto take_partner = [&]() {
    static_for<0, N>([&](auto i) { Mp[i] = __shfl_xor(An[i], 32); });
    qacc_s = qacc; ldM_s = ldM; ldE_s = ldE;
  };
  const int s0 = e + 1 - keep;              // S4: records of local steps >= s0 go to LDS
  auto hand_off = [&](int s, const double (&M)[N], double vfull, auto to_lds) {
    if constexpr (S4 && decltype(to_lds)::value) {
      char* w = reinterpret_cast<char*>(lrecs + (long)(s - s0) * WS);
      static_for<0, N>([&](auto i) { *reinterpret_cast<double*>(w + lloff[i]) = M[i] * vfull; });
    } else if constexpr (LEAN) {
      char* w = reinterpret_cast<char*>(wsb + (long)s * WS);     // uniform base + 32-bit lane offset
      // (the offset passes through an empty asm: hipcc otherwise hoists its zero-extension out of the loop and pays
      //  a 64-bit add per store instead of the SGPR-base + 32-bit-offset addressing mode)
      static_for<0, N>([&](auto i) {
        asm volatile("" : "+v"(loff[i]));
        *reinterpret_cast<double*>(w + loff[i]) = M[i] * vfull;
      });
    } else {
      double* w = rec0 + (long)s * WS + stoff;
      static_for<0, N>([&](auto i) { w[i * RW] = M[i] * vfull; });
    }
  };

  // hipcc merges the memory-counter state of the loop entry with the back edge's and waits for the more
  // pessimistic of the two: a dummy hand-off (N stores into the meeting record, rewritten later) behind the
  // first loads makes both look alike, so the wait at the loop head is vmcnt(N) -- the prefetched node
  // potentials -- instead of a wait for the previous step's hand-off stores.
  hand_off(e, An, 1.0, std::false_type{});
#ifdef SVAE_PHASE_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast_ = __builtin_readcyclecounter();
#endif
  // MIX: weights of local pairs s and s + 1 (raw, fetched one step ahead like the node potentials)
  double wr0 = MIX ? wrow[(long)wnode(0) * K] : 0.0, wr1 = MIX ? wrow[(long)wnode(1) * K] : 0.0;
  auto elim_step = [&](int s, auto to_lds) {
    if (s == jx) take_partner();
    const double JoX = __builtin_fma(Jo_n, jm2, jadd);
    double ho = ho_n;
    pJn += nstep; phn += nstep;            // node s + 1 <= e: the meeting node's potentials included
    Jo_n = *pJn;
    ho_n = *phn;
    if constexpr (RING) ring_pair(s & 1, true);
    else if (INHOMOG && !MIX) load_pair(s, true);
    double wq[2] = {wr0 * wmask, wr1 * wmask};     // MIX: lane k = weight of state k (pair s, pair s + 1)
    if constexpr (MIX) {
      wr0 = wr1;
      wr1 = wrow[(long)wnode(s + 2) * K];
      dpp_fence(wq);
    }

    // condition on the node potential; right-hand sides ride in the upper lanes
    double M[N], Bt[N];
    if constexpr (MIX) {
      static_for<0, N>([&](auto i) { M[i] = __builtin_fma(JoX, E[i], An[i]); });
    } else {
      static_for<0, N>([&](auto i) { M[i] = __builtin_fma(JoX, EX[i], An[i]); });
    }
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(M[i], ho, EH); });      // lane 15: h_filt = h_pred + h_node
    // B operand of the Schur stage: lanes < N: -J12'[k][c]; lane 15: -h_filt,k
    if constexpr (MIX) {
      // mixed J12' of this pair, once, for both of its uses (disjoint lanes; zero in lane 15: h_filt is left alone)
      constexpr int R0 = (N + 1) / 2, R1 = N - R0;
      double mjA[R0], mjB[R1 > 0 ? R1 : 1];
      static_for<0, R0>([&](auto r) { mjA[r] = 0.0; });
      static_for<0, R1>([&](auto r) { mjB[r] = 0.0; });
      mix2(std::integral_constant<int, R0>{}, std::integral_constant<int, R1>{}, std::integral_constant<int, ENT_J>{},
           std::integral_constant<int, N * ENT_J>{}, mjA, mjB, tj_l, tj_l + R0 * ENT_J, wq[0], wq[0]);
      static_for<0, N>([&](auto k) {
        double mj;
        if constexpr (k < R0) mj = mjA[k]; else mj = mjB[k - R0];
        Bt[k] = __builtin_fma(jcmask, mj, -EH * M[k]);
        M[k] = __builtin_fma(rmask, mj, M[k]);
      });
    } else {
      static_for<0, N>([&](auto k) { Bt[k] = __builtin_fma(-EH, M[k], NJ12c[k]); });
    }
    dpp_fence(M);
    TE_TICK(0)

    double vfull = col ? 0.0 : 1.0;
    te_gauss_jordan<N>(M, E, qacc, vfull);
    TE_TICK(1)

    // next pivot block, slot layout:  AnD[j] = Cc[j] + sum_k X[k][i] * Bt[k]   (row i = 2j + gl;
    // X[k][i] = lane N + j of M[k] in this DPP row)
    double AnD[J];
    if constexpr (MIX) {
      static_for<0, J>([&](auto j) { AnD[j] = 0.0; });
      mix2(std::integral_constant<int, J>{}, std::integral_constant<int, J>{}, std::integral_constant<int, ENT_C>{},
           std::integral_constant<int, J * ENT_C>{}, AnD, AnD, c22_l, c11_l, wq[0], wq[1]);
    } else {
      static_for<0, J>([&](auto j) { AnD[j] = Cc[j]; });
    }
    asm volatile("s_nop 1");
    static_for<0, N>([&](auto k) {
      static_for<0, J>([&](auto j) { mac_bc<N + j>(AnD[j], M[k], Bt[k]); });
    });
    TE_TICK(2)

    // The next pivot block leaves the Schur stage in slot layout (row 2j + gl in DPP row gl of the chain's pair) and
    // the Gauss-Jordan wants every row in both DPP rows.  Round 4: the re-replication goes through LDS -- J stores of
    // [row][16 lanes], N reads of a row each -- and its round trip is covered by the hand-off (column scaling +
    // stores) and the log-determinant bookkeeping, which do not depend on it; as register shuffles (pair_split: two
    // v_permlane16_swap per register and their copies) it was 33 instructions on the serial chain of the step.
    // (MIX keeps the shuffles: its LDS is the parameter tables.)
    if constexpr (TE_LDS_SPLIT<MIX, CROSS>()) {
      double* sp = split_tile + dir * (N + 1) * 16;
      __builtin_amdgcn_wave_barrier();
      static_for<0, J>([&](auto j) { sp[(2 * j + gl) * 16 + c] = AnD[j]; });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // scale the inverse's columns, hand the record to the smoother phase
    vworst = asm_max(vworst, vfull);
    ldM *= vfull;
    if ((s & 3) == 3) {
      asm volatile("; renormalise the determinant product");      // (keeps this a branch: as selects it is 6 instructions per step)
      ldE += __builtin_amdgcn_frexp_exp(ldM);
      ldM = __builtin_amdgcn_frexp_mant(ldM);
    }
    hand_off(s, M, vfull, to_lds);

    if constexpr (TE_LDS_SPLIT<MIX, CROSS>()) {
      const double* sp = split_tile + dir * (N + 1) * 16 + c;
      static_for<0, N>([&](auto i) { An[i] = sp[i * 16]; });
      __builtin_amdgcn_wave_barrier();
    } else {
      dpp_fence(AnD);
      static_for<0, J>([&](auto j) {
        if constexpr (2 * j + 1 < N) pair_split(AnD[j], An[2 * j], An[2 * j + 1]);
        else { double dummy; pair_split(AnD[j], An[2 * j], dummy); }
      });
    }
    if constexpr (RING) lds_barrier();     // the next step's parameters are in the other ring slot
    TE_TICK(3)
  };
  if constexpr (RING) {
    lds_barrier();                          // #0: the rings are zeroed
    lds_barrier();                          // #1: the producers have left step 0's parameters in slot 0
  }
  {
    // (two loops, not a branch around the stores: a conditional store makes hipcc wait for the previous step's stores)
    const int s_hbm = keep > 0 ? (s0 < e ? s0 : e) : e;       // steps whose record goes to the HBM workspace
    int s = 0;
    for (; s < s_hbm; ++s) elim_step(s, std::false_type{});
    if constexpr (S4) for (; s < e; ++s) elim_step(s, std::true_type{});
  }
  if (jx == e) take_partner();

  // ---- meeting node ---------------------------------------------------------------------------------
  // Both chains' An count J22'(pair e-1) + J11'(pair e) (the same two blocks in either orientation):
  // P_m = An_own + An_partner - (J22' + J11') + node.  No pair potential ahead: the right-hand sides are
  // h alone (G = 0), and the record is the smoother's starting point (c = mu_m, P^-1 = Sigma_m).
  double qacc_m = 0.0, vfull_m = col ? 0.0 : 1.0;
  {
    const long o = pair_off(e - 1), o1 = pair_off(e);
    const double JoX = col ? -2.0 * Jo_n : 1.0;
    double ho = ho_n;
    double M[N];
    if constexpr (MIX) {
      double dbl[N];
      static_for<0, N>([&](auto i) { dbl[i] = 0.0; });
      const double* wa = a.mix_w + ((long)b * T + wnode(e - 1)) * K;
      const double* wb = a.mix_w + ((long)b * T + wnode(e)) * K;
      for (int k = 0; k < K; ++k) {
        const double va = wa[k], vb = wb[k];
        const double* p22 = (dir ? a.J11 : a.J22) + (long)k * N * N + cc;
        const double* p11 = (dir ? a.J22 : a.J11) + (long)k * N * N + cc;
        static_for<0, N>([&](auto i) { dbl[i] = __builtin_fma(va, p22[i * N], __builtin_fma(vb, p11[i * N], dbl[i])); });
      }
      static_for<0, N>([&](auto i) {
        M[i] = __builtin_fma(JoX, E[i], (An[i] + Mp[i]) - (col ? -2.0 * dbl[i] : 0.0));
      });
    } else if constexpr (RING) {
      // C of the last elimination step (ring slot (e - 1) & 1, untouched since), every row in every DPP row
      const char* base = reinterpret_cast<const char*>(ring.m) + 8u * (unsigned)(((e - 1) & 1) * ring.mixslot);
      static_for<0, N>([&](auto i) {
        const int hi = i > c ? i : c, lo = i > c ? c : i;
        const unsigned ent = (unsigned)(dir * ring.cs) + (unsigned)(col ? N * N + hi * (hi + 1) / 2 + lo : N * N + RTRI);
        const double dbl = *reinterpret_cast<const double*>(base + 8u * ent);
        M[i] = __builtin_fma(JoX, E[i], (An[i] + Mp[i]) - dbl);
      });
    } else {
      static_for<0, N>([&](auto i) {
        const double r22 = q22[o + i * N + cc], r11 = q11[o1 + i * N + cc];
        const double dbl = col ? -2.0 * (r22 + r11) : 0.0;
        M[i] = __builtin_fma(JoX, E[i], (An[i] + Mp[i]) - dbl);
      });
    }
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(M[i], ho, EH); });
    dpp_fence(M);
    te_gauss_jordan<N>(M, E, qacc_m, vfull_m);
    if (S4 && keep > 0) hand_off(e, M, vfull_m, std::true_type{});
    else hand_off(e, M, vfull_m, std::false_type{});
  }

  // ---- log-normaliser --------------------------------------------------------------------------------
  {
    // per chain: 1/2 sum h'P^-1h - 1/2 sum log p = 1/2 [ q + sum_c log|prod_t (-1/p_c)| ]
    auto chain_part = [&](double q, double m, int ee) {
      const int ex = __builtin_amdgcn_frexp_exp(m);
      const double mant = __builtin_amdgcn_frexp_mant(m);
      double part = col ? (::log(fabs(mant)) + (double)(ee + ex) * 0.6931471805599453094) : 0.0;
      if (c == HL) part = q;
      return 0.5 * row_sum16(part);
    };
    double pm = col ? ::log(fabs(vfull_m)) : 0.0;
    if (c == HL) pm = qacc_m;
    const double meet_total = 0.5 * row_sum16(pm);
    // all of this chain's eliminations + the partner's up to its hand-over + this chain's meeting node
    const double chain_total = chain_part(qacc, ldM, ldE) + __shfl_xor(chain_part(qacc_s, ldM_s, ldE_s), 32);
    double z = 0.0;
    if (a.node_logZ) {
      for (int t = c; t < T; t += 16) z += a.node_logZ[(long)b * T + t];
    }
    if (INHOMOG && !MIX && !RING) {
      const double* lz = a.logZ_pair + (a.pair_seq_stride ? (long)b * (T - 1) : 0);
      for (int t = c; t < T - 1; t += 16) z += lz[t];
    }
    // (MIX: the mixed log-normaliser constants sum_k w_k logZ_k are the caller's, a (B,T,K) x (K) product)
    double total = row_sum16(z) + chain_total + meet_total + ((MIX || RING) ? 0.0 : a.init_logZ[0]);
    if (!INHOMOG) total += (double)(T - 1) * a.logZ_pair[0];
    if (lane == 0) a.lognorm[b] = total;
    const bool lane_bad = col && (!(vworst < 0.0) || !(vfull_m < 0.0));
    const bool bad = __ballot(lane_bad) != 0 || !(total == total);
    if (bad && lane == 0) {   // rare path: keep the smallest failing index (+1); 0 = ok
      int old = *(volatile int32_t*)a.info;
      while (old == 0 || old > b + 1) {
        const int seen = atomicCAS(a.info, old, b + 1);
        if (seen == old) break;
        old = seen;
      }
    }
  }
  TE_TICK(4)
  if constexpr (S4) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    te_smooth4<N>(a, b, 0, lane, tab_static, tab_static + 512, xch_static, lrecs, keep);
    return;
  }

  if constexpr (TE_LDS_SPLIT<MIX, CROSS>() && !S4) {   // the transposition tiles served as the elimination phase's split tile
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < 2 * 16 * 16; q += 64) tab[q] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- smoother phase: moment form on homogeneous coordinates, local steps e, e-1, .., 0 ---------------
  // S~ in slot layout (row i = 2j+gl of the (N+1) x (N+1) tile, lane = column); starts from e_N e_N' so that
  // the generic step at the meeting record (G = 0, c = mu) yields [[Sigma + mu mu', mu], [mu', 1]].
  double ED[J1];                          // ED[j][c] = (c == 2j+gl): picks S[i][i] in slot layout
  static_for<0, J1>([&](auto j) { ED[j] = (c == 2 * j + gl && c < N) ? 1.0 : 0.0; });
  const double CN = (gl == (N & 1)) ? EN : 0.0;     // row N of G~ = e_N (slot N/2 of DPP row N & 1)
  double S[J1];
  static_for<0, J1>([&](auto j) { S[j] = (j == N / 2) ? CN : 0.0; });
  dpp_fence(S);
  double sumS[J], sumW[J], Stop[J];
  static_for<0, J>([&](auto j) { sumS[j] = 0.0; sumW[j] = 0.0; Stop[j] = 0.0; });
  const bool own_N = (gl == (N & 1));     // the DPP row holding row N (E[x_t]) in slot N/2
  // chain B, even T: its first smoother step repeats pair e-1, which chain A counts
  const bool skip2nd = dir && !oddT;
  const double wsp = skip2nd ? 0.0 : 1.0;
  const bool own_e = oddT && !dir;        // who reports the meeting node

  // node statistics: unconditional stores through per-lane walking pointers (idle lanes -> trash)
  const bool dlane = col && (c & 1) == gl, xlane = col && own_N;
  const long nstride = dir ? N : -N;      // towards smaller s
  double* pdg = trash;
  double* pex = trash + 1;
  auto node_ptrs = [&](int s) {
    const long o = ((long)b * T + (dir ? T - 1 - s : s)) * N + c;
    pdg = dlane ? a.E_node_diagxx + o : trash;
    pex = xlane ? a.E_node_x + o : trash + 1;
  };
  if (own_e) node_ptrs(e);

  // operands of one step (prefetched one step ahead, every load unconditional)
  //   full:  H[k] = [X | c][c][k] (row c of the record; lane N: e_N; lanes > N: 0), Gc[j][c] = [X | c][2j+gl][c]
  //          (row N: e_N), Pi[j][c] = P^-1[2j+gl][c]
  //   LEAN:  Pi[j] = [P^-1 | c][2j+gl][c] (lane N: c_i; rows >= N, lanes > N: the record's zero entry)
  int rslot = 0;                          // RING: tile-ring slot of the current smoother step
  struct Ops { double H[LEAN ? 1 : N + 1]; double Gc[LEAN ? 1 : J1]; double Pi[J1]; };
  const double* hp_ = col ? rec0 + c * RW + N : (c == N ? zpage : zpage + N + 2);
  const long hstride = col ? WS : 0;
  const double* rp_[J1];                  // full: row i of the record at this lane's P^-1 column (page for idle lanes)
  long rstride[J1];
  int goff[J1];                           // full: from rp_ to this lane's entry of [X | c] (row N: e_N in the page)
  unsigned poff[LEAN ? J1 : 1];
  static_for<0, J1>([&](auto j) {
    const int i = 2 * j + gl;
    const bool rowok = i < N && c <= N;
    rp_[j] = rowok ? rec0 + i * RW + cc : zpage + N + 2;
    rstride[j] = rowok ? WS : 0;
    goff[j] = rowok ? N + (c - cc) : ((i == N && c < N + 2) ? c - (N + 2) : 0);
    if constexpr (LEAN) {
      const int hi = i > c ? i : c, lo = i > c ? c : i;
      poff[j] = 8u * (choff + ((i < N && col) ? hi * (hi + 1) / 2 + lo : ((i < N && c == N) ? TRI + i : LZERO)));   // bytes
    }
  });
  const double cmask = col ? 1.0 : 0.0;
  if constexpr (!LEAN) {
    hp_ += (long)e * hstride;
    static_for<0, J1>([&](auto j) { rp_[j] += (long)e * rstride[j]; });
  }
  const double* lrec = wsb + (long)e * WS;           // LEAN: uniform record pointer
  auto load_ops = [&](Ops& o, bool more) {           // records e, e-1, .. (more: another record follows)
    if constexpr (LEAN) {
      static_for<0, J1>([&](auto j) { o.Pi[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lrec) + poff[j]); });
      lrec -= more ? WS : 0;
    } else {
      load_row<N + 1>(hp_, o.H);
      static_for<0, J1>([&](auto j) {
        o.Gc[j] = rp_[j][goff[j]];            // lane c: X[i][c], lane N: c_i; row N: e_N; else 0
        o.Pi[j] = *rp_[j];                    // lane N of a real row reads P^-1[i][0]: masked where used
      });
      const long mv = more ? 1 : 0;
      hp_ -= mv * hstride;
      static_for<0, J1>([&](auto j) { rp_[j] -= mv * rstride[j]; });
    }
  };

  // one smoother step.  KIND: 0 generic, 1 first (meeting record), 2 second (weight of the repeated pair)
  auto step = [&](auto kind, int s, Ops& cur, Ops& nxt) {
    constexpr int KIND = decltype(kind)::value;
    load_ops(nxt, s > 1);                    // prefetch record s-1 (s = 0: re-reads record 0, unused)
    if constexpr (RING) { if (KIND != 1) ring_pair(s & 1, false); }
    else if (INHOMOG && LEAN && KIND != 1) load_pair(s, false);   // G~ of step s uses pair s (the meeting record: G = 0)

    double Gc[J1], H[N + 1];
    if constexpr (LEAN) {
      // G~ rows of this DPP row: X[i][c] = sum_k P^-1[i][k] J12'[k][c] (lanes < N), c_i (lane N); row N = e_N
      static_for<0, J1>([&](auto j) { Gc[j] = (j == N / 2) ? __builtin_fma(EN, cur.Pi[j], CN) : EN * cur.Pi[j]; });
      if (KIND != 1) {
        dpp_fence(cur.Pi);
        static_for<0, N>([&](auto k) {
          static_for<0, J>([&](auto j) { mac_bc<k, true>(Gc[j], cur.Pi[j], NJ12c[k]); });
        });
      }
      // transposed, replicated copy through LDS: H[k][lane c] = G~[c][k]
      // Lanes > N (zeros) store into their OWN row's padding column N + 1 (RSL >= N + 2).  At column c they would land in
      // the rows behind (row i, lane c >= RSL = row i + 1, column c - RSL), correct only while those rows' own stores come
      // later -- and per lane the J1 addresses differ, so the compiler may order them as it likes.  (Round 5's "n = 4
      // only" failure of the ring consumer's two-steps-per-trip loop was exactly that: hipcc scheduled the store of rows
      // 4 / 5 ahead of rows 0 .. 3, lane 10 of row 3 zeroed e_N[N], and the mean was gone from the second step on.  Found
      // round 6 by dumping registers from patched ISA: docs/experiments/r6_ring_two_per_trip_reproducer.md.)
      double* tb = tab + dir * 16 * RSL;
      const int cw = c <= N ? c : N + 1;
      __builtin_amdgcn_wave_barrier();
      static_for<0, J1>([&](auto j) { tb[(2 * j + gl) * RSL + cw] = Gc[j]; });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      static_for<0, (N + 2) / 2>([&](auto q) {
        const double2 v = reinterpret_cast<const double2*>(tb + c * RSL)[q];
        H[2 * q] = v.x;
        if constexpr (2 * q + 1 <= N) H[2 * q + 1] = v.y;
      });
      __builtin_amdgcn_wave_barrier();
    } else {
      static_for<0, J1>([&](auto j) { Gc[j] = cur.Gc[j]; });
      static_for<0, N + 1>([&](auto k) { H[k] = cur.H[k]; });
      dpp_fence(Gc);
    }

    // W~[i] = S~[i] G~'  for my rows:  sum_k -/+ bcast_k(S[j]) H[k]
    double W[J1];
    static_for<0, J1>([&](auto j) { W[j] = 0.0; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(W[j], S[j], H[k]); });
    });
    dpp_fence(W);
    if constexpr (CROSS && KIND != 1) {
      // chain A: W[i][c] = W~_s[i][c] (record s);  chain B: W[i][c] = E[x~_{t-1,i} x~_{t,c}] = W~_{t-1}[c][i]
      // (record T-2-s, transposed).  One unconditional store per slot; lanes / rows outside the tile, and chain B's
      // repeat of pair e-1 with even T, go to the trash slot.
      constexpr int HSX = ws_h_stride(N);
      const int tr = dir ? T - 2 - s : s;
      double* w3 = a.ws3 + ((long)b * T + tr) * (N + 1) * HSX;
      const bool skipx = KIND == 2 && skip2nd;
      static_for<0, J1>([&](auto j) {
        const int i = 2 * j + gl;
        const bool ok = i <= N && c <= N && !skipx;
        double* q = ok ? w3 + (dir ? c * HSX + i : i * HSX + c) : trash;
        *q = W[j];
      });
    }
    if constexpr (!INHOMOG && KIND == 0) static_for<0, J>([&](auto j) { sumW[j] += W[j]; });   // (W dies in the split)
    double WR[N + 2];
    static_for<0, J1>([&](auto j) { pair_split(W[j], WR[2 * j], WR[(2 * j + 1 <= N) ? 2 * j + 1 : N + 1]); });
    // S~_t[i] = P^-1[i] + G~[i] W~ = Pi + sum_k -/+ bcast_k(Gc[j]) WR[k]
    double Sn[J1];
    static_for<0, J1>([&](auto j) { Sn[j] = LEAN ? __builtin_fma(-EN, cur.Pi[j], cur.Pi[j]) : cur.Pi[j] * cmask; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(Sn[j], Gc[j], WR[k]); });
    });

    const int t = dir ? T - 1 - s : s;
    if constexpr (MIX) {
      // contraction of this node's tiles with the K parameter sets: lane k accumulates state k
      //   acc0 = <S~_t, J11_k>, acc1 = <S~_t, J22_k>, acc2 = <W~, J12_k> (orientation by chain), my rows only
      const bool own = KIND != 1 || own_e;
      const double fX = (KIND != 1 && !(KIND == 2 && skip2nd)) ? 1.0 : 0.0;
      double acc[3] = {0.0, 0.0, 0.0};
      dpp_fence(Sn);
      // groups of two column pairs x three tables, read one group ahead (see mix_rows)
      constexpr int CG = 2, NCG = (NC2 + CG - 1) / CG, NQ = J * NCG;
      double2 cb[2][3 * CG];
      auto cload = [&](auto q, double2 (&b)[3 * CG]) {
        constexpr int j = q / NCG, c0 = (q % NCG) * CG;
        static_for<0, CG>([&](auto g) {
          if constexpr (c0 + g < NC2) {
            constexpr int off = (j * 2 * NC2 + c0 + g) * 16;
            b[3 * g] = cta_l[off]; b[3 * g + 1] = ctc_l[off]; b[3 * g + 2] = ctx_l[off];
          }
        });
      };
      auto cfma = [&](auto q, const double2 (&b)[3 * CG]) {
        constexpr int j = q / NCG, c0 = (q % NCG) * CG;
        static_for<0, CG>([&](auto g) {
          if constexpr (c0 + g < NC2) {
            constexpr int cp = c0 + g;
            mac_bc<2 * cp>(acc[0], Sn[j], b[3 * g].x);
            mac_bc<2 * cp>(acc[1], Sn[j], b[3 * g + 1].x);
            mac_bc<2 * cp>(acc[2], W[j], b[3 * g + 2].x);
            if constexpr (2 * cp + 1 < N) {
              mac_bc<2 * cp + 1>(acc[0], Sn[j], b[3 * g].y);
              mac_bc<2 * cp + 1>(acc[1], Sn[j], b[3 * g + 1].y);
              mac_bc<2 * cp + 1>(acc[2], W[j], b[3 * g + 2].y);
            }
          }
        });
      };
      cload(std::integral_constant<int, 0>{}, cb[0]);
      static_for<0, NQ>([&](auto q) {
        if constexpr (q + 1 < NQ) cload(std::integral_constant<int, q + 1>{}, cb[(q + 1) & 1]);
        cfma(q, cb[q & 1]);
      });
      // chain A's cross moment is pair t (slot 0), chain B's pair t-1 (slot 1); then the two DPP rows' shares
      double o1 = __builtin_fma(dir ? 0.0 : fX, acc[2], acc[0]);
      double o3 = __builtin_fma(dir ? fX : 0.0, acc[2], acc[1]);
      double e1, d1, e3, d3;
      pair_split(o1, e1, d1);
      pair_split(o3, e3, d3);
      const bool wr = own && gl == 0 && c < K;
      double* q1 = wr ? a.mix_out + (((long)b * T + t) * 2) * K + c : trash;
      double* q3 = wr ? q1 + K : trash + 1;
      *q1 = e1 + d1;
      *q3 = e3 + d3;
    } else if constexpr (RING) {
      // this node's tiles for the producers' contraction with the K parameter sets: S~ rows < N as a lower triangle,
      // the cross moment W~ rows < N -- zero where the pair is not this chain's to count (the meeting record has G = 0,
      // hence W~ rows < N = 0); everything else of the slot registers goes to the column's junk entries
      char* base = reinterpret_cast<char*>(ring.s) + 8u * (unsigned)(rslot * ring.sslot + dir * ring.ss);
      const double fX = (KIND == 2 && skip2nd) ? 0.0 : 1.0;
      static_for<0, J>([&](auto j) {
        const int i = 2 * j + gl;
        const int junk = RTRI + (c <= N ? c : N + 1);               // the column's n + 2 junk entries
        const unsigned es = (unsigned)((i < N && c <= i) ? i * (i + 1) / 2 + c : junk);
        const unsigned ew = (unsigned)((i < N && col) ? ring.woff + i * N + c : junk);
        *reinterpret_cast<double*>(base + 8u * es) = Sn[j];
        *reinterpret_cast<double*>(base + 8u * ew) = fX * W[j];
      });
      rslot = rslot + 1 == ring.depth ? 0 : rslot + 1;
    } else if (INHOMOG) {
      // per-step pair blocks [E x_t x_t' | E x_t x_{t+1}' | E x_{t+1} x_{t+1}'] for pair index p:
      // the owner of node t writes S~_t into pair t (first block) and pair t-1 (third block); the
      // cross moment W~ = E[x~_{prev} x~_{this}'] is pair s (transposed) for A, pair T-2-s for B.
      double* EP = a.E_pair + (long)b * (T - 1) * 3 * N * N;
      const bool own = KIND != 1 || own_e;
      const bool crossw = KIND != 1 && !(KIND == 2 && skip2nd);
      const int p = dir ? T - 2 - s : s;
      static_for<0, J>([&](auto j) {
        const int i = 2 * j + gl;
        if (i < N && col) {
          if (own && t < T - 1) EP[((long)t * 3 + 0) * N * N + i * N + c] = Sn[j];
          if (own && t > 0) EP[((long)(t - 1) * 3 + 2) * N * N + i * N + c] = Sn[j];
          if (crossw) EP[((long)p * 3 + 1) * N * N + (dir ? i * N + c : c * N + i)] = W[j];
        }
      });
    } else {
      if constexpr (KIND == 1) {
        static_for<0, J>([&](auto j) { Stop[j] = Sn[j]; });
      } else if constexpr (KIND == 2) {
        // (multiplications by 0 / 1, exact: a select here makes hipcc merge the stores through a pointer
        //  phi, which leaves three accumulators on the stack)
        static_for<0, J>([&](auto j) {
          sumS[j] = wsp * Sn[j];
          sumW[j] = wsp * W[j];
          Stop[j] = __builtin_fma(wsp, Stop[j], __builtin_fma(-wsp, Sn[j], Sn[j]));
        });
      } else {
        static_for<0, J>([&](auto j) { sumS[j] += Sn[j]; });
      }
    }

    // node statistics: diag E[x_t x_t'] (lane i of DPP row i & 1), E[x_t] = row N
    double dg = 0.0;
    static_for<0, J>([&](auto j) { dg = __builtin_fma(ED[j], Sn[j], dg); });
    *pdg = dg;
    *pex = Sn[N / 2];
    if constexpr (KIND == 1) node_ptrs(e - 1);
    else { pdg += dlane ? nstride : 0; pex += xlane ? nstride : 0; }
    static_for<0, J1>([&](auto j) { S[j] = Sn[j]; });
    if constexpr (RING) lds_barrier();       // the tiles are in the ring; the next step's J12 is in its slot
  };

  {
    Ops A, Bq;
    constexpr std::integral_constant<int, 0> GEN{};
    load_ops(A, true);
    step(std::integral_constant<int, 1>{}, e, A, Bq);
    step(std::integral_constant<int, 2>{}, e - 1, Bq, A);
    int s = e - 2;
#ifndef SVAE_RING_TWO_PER_TRIP
#define SVAE_RING_TWO_PER_TRIP 0      // 1: the ring consumer with the two-steps-per-trip loop as well (`make ring2`: the regression build)
#endif
    if constexpr (RING && !SVAE_RING_TWO_PER_TRIP) {
      // one step per trip, the stage copied: same speed as two per trip here (0.51 ms at T = 500), half the code.  (Round 5
      // chose it because the other form was WRONG at N = 4: the transposition-tile stores above, not the loop -- fixed in
      // round 6; tests/test_slds_hip.py runs the two-per-trip build of N = 4 against the table kernel.)
      for (; s >= 1; --s) { step(GEN, s, A, Bq); A = Bq; }
    } else {
      for (; s >= 2; s -= 2) {          // two steps per trip: the prefetch buffers ping-pong
        step(GEN, s, A, Bq);
        step(GEN, s - 1, Bq, A);
      }
      if (s == 1) { step(GEN, 1, A, Bq); A = Bq; }     // (copy: no swapped call sites, the buffers stay in registers)
    }
    step(GEN, 0, A, Bq);
  }
#ifdef SVAE_PHASE_TIMING
  TE_TICK(5)
  if (lane == 0) { for (int q = 0; q < 6; ++q) a.E_init[(long)b * (N * N + N) + q] = (double)tm[q]; }
  return;
#endif

  // ---- global statistics ------------------------------------------------------------------------------
  // S = S~ at the chain's end node (x_0 for A, x_{T-1} for B).  Sums over the chain's pairs:
  //   sumS = sum S~(s) over its counted steps;  sumP = sum S~(s+1) = (sumS - S~(0)) + Stop;  sumW = sum W~(s)
  // A: first block += sumS, third += sumP, cross += sumW';  B: first += sumP, third += sumS, cross += sumW.
  if (!INHOMOG) {
    double sumP[J];
    static_for<0, J>([&](auto j) { sumP[j] = (sumS[j] - S[j]) + Stop[j]; });
    // chain B's sums travel to chain A's lanes (lane ^ 32); its cross sum is transposed through LDS
    double oS[J], oP[J], oWt[J];
    static_for<0, J>([&](auto j) {
      oS[j] = __shfl_xor(sumS[j], 32);
      oP[j] = __shfl_xor(sumP[j], 32);
    });
    __builtin_amdgcn_wave_barrier();
    if (dir) static_for<0, J>([&](auto j) { tab[(gl * 16 + c) * 8 + j] = sumW[j]; });   // [gl][c][j]: W_B[2j+gl][c]
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // A's lane (gl, c) holds W_A[i=2j+gl][c], which lands at cross[c][i]: it needs W_B[c][i]
    static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl;
      const int ii = i < N ? i : 0;
      oWt[j] = tab[((cc & 1) * 16 + ii) * 8 + (cc >> 1)];
    });
    if (!dir) {
      static_for<0, J>([&](auto j) {
        const int i = 2 * j + gl;
        if (i < N && col) {
          double* ep = a.E_pair + (long)b * 3 * N * N;
          ep[i * N + c] = sumS[j] + oP[j];
          ep[N * N + c * N + i] = sumW[j] + oWt[j];
          ep[2 * N * N + i * N + c] = sumP[j] + oS[j];
        }
      });
    }
  }
  if constexpr (CROSS) {
    // record T-1: W~_{T-1} = e_n [mu_{T-1}; 1]'  (what the one-directional smoother's first step leaves there)
    if (dir) {
      constexpr int HSX = ws_h_stride(N);
      double* w3 = a.ws3 + ((long)b * T + (T - 1)) * (N + 1) * HSX;
      static_for<0, J1>([&](auto j) {
        const int i = 2 * j + gl;
        if (i <= N && c <= N) w3[i * HSX + c] = (i == N) ? S[j] : 0.0;
      });
    }
  }
  if (!dir) {
    static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl;
      if (i < N && col) a.E_init[(long)b * (N * N + N) + i * N + c] = S[j];
    });
    if (col && own_N) a.E_init[(long)b * (N * N + N) + N * N + c] = S[N / 2];
  }
}

template <int N, bool INHOMOG, bool LEAN, bool MIX = false, bool CROSS = false, bool S4 = false>
__global__ __launch_bounds__(MIX ? 512 : (S4 ? 128 : 64)) void lds_estep_twoend_kernel(const LdsArgs a) {
  lds_estep_twoend_body<N, INHOMOG, LEAN, MIX, CROSS, S4>(a, (int)blockIdx.x);
}

template <int N>
static int launch_estep_twoend(const LdsArgs& a, bool inhomog, bool lean, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    dim3 grid(a.B), block(64);
    if (a.ws3) {     // with the cross moments for the reverse-mode sweeps (CROSS): homogeneous lean / per-step full record
      if (inhomog)
        hipLaunchKernelGGL((lds_estep_twoend_kernel<N, true, false, false, true>), grid, block, 0, stream, a);
      else
        hipLaunchKernelGGL((lds_estep_twoend_kernel<N, false, true, false, true>), grid, block, 0, stream, a);
      return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (inhomog && lean)
      hipLaunchKernelGGL((lds_estep_twoend_kernel<N, true, true>), grid, block, 0, stream, a);
    else if (inhomog)
      hipLaunchKernelGGL((lds_estep_twoend_kernel<N, true, false>), grid, block, 0, stream, a);
    else if (lean && a.B <= TE_S4_MAX_B) {   // one chain per wavefront in the smoother phase while SIMDs are idle
      // ... and the records the smoother reads first stay in LDS: as many per chain as fit when a CU takes one
      // workgroup MORE than its even share ceil(B / 256) of the launch (the dispatcher does not balance exactly: sized
      // for the even share, a few workgroups would wait for a second round and double the kernel time)
      auto kern = lds_estep_twoend_kernel<N, false, true, false, false, true>;
      // (CU count and LDS per CU from the device: a partitioned part -- fewer CUs -- or one with less LDS gets a smaller
      //  window instead of an oversubscribed one)
      int dev = 0, cus = 256, lds_cu = 160 * 1024;
      if (hipGetDevice(&dev) == hipSuccess) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev);
        if (cus < 1) cus = 256;
        if (lds_cu < 64 * 1024) lds_cu = 64 * 1024;
        (void)hipGetLastError();
      }
      const int wg_per_cu = (a.B + cus - 1) / cus + 1;
      const long budget = (long)lds_cu / wg_per_cu - TE_S4_LDS_BYTES - 64 * (long)sizeof(double);
      const long rec_bytes = 2L * te_lean_step_doubles(N) * sizeof(double);          // one record of each chain
      // Measured (T = 200, n = 10): with one workgroup per CU (B <= 256) the LDS records cost nothing (0.142 ms either
      // way) and the HBM traffic per launch falls to 1.6x the algorithmic bytes; with two workgroups per CU (B = 512)
      // the kernel is 2 % slower with them (0.152 vs 0.149 ms), so they stay in HBM there: time before traffic.
      long keep = (a.B <= cus && budget > 0) ? budget / rec_bytes : 0;
#ifdef SVAE_S4_KEEP_OVERRIDE
      keep = SVAE_S4_KEEP_OVERRIDE;      // (experiments: tools/build_variant.sh ... -DSVAE_S4_KEEP_OVERRIDE=<records>)
#endif
      if (keep > te_elims(a.T) + 1) keep = te_elims(a.T) + 1;
      if (keep < 3) keep = 0;
      LdsArgs a2 = a;
      a2.lds_keep = (int)keep;
      const long bytes = keep * rec_bytes + (keep > 0 ? 64 * sizeof(double) : 0);    // + one dummy slot per lane
      static LdsGrant grant;
      if (bytes > 0 && !grant.ensure(reinterpret_cast<const void*>(kern), bytes)) {
        // the window was refused: every record through HBM (always valid) instead of an error
        a2.lds_keep = 0;
        hipLaunchKernelGGL(kern, grid, dim3(128), 0, stream, a2);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
      }
      hipLaunchKernelGGL(kern, grid, dim3(128), (size_t)bytes, stream, a2);
    }
    else if (lean)
      hipLaunchKernelGGL((lds_estep_twoend_kernel<N, false, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((lds_estep_twoend_kernel<N, false, false>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

// MIX launch: up to 8 sequences (wavefronts) per workgroup share the LDS tables
template <int N>
static int launch_estep_twoend_mix(const LdsArgs& a, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    const long bytes = te_mix_lds_bytes(N, a.mix_K);
    if (a.mix_K < 1 || a.mix_K > TE_MIX_MAX_K || bytes > TE_MIX_MAX_LDS) return -30;
    static LdsGrant grant;                 // largest dynamic-LDS size granted so far (this instantiation, per device)
    auto kern = lds_estep_twoend_kernel<N, true, false, true>;
    if (!grant.ensure(reinterpret_cast<const void*>(kern), bytes)) return -31;
    // Sequences (wavefronts) per workgroup: the tables are 100+ KB, so a CU holds ONE workgroup, and the mixing's table
    // reads (195 KB per wavefront-step) share that CU's LDS bandwidth -- eight wavefronts per CU are LDS-bound at twice
    // the step time of a lone one.  Only as many per workgroup as it takes to place the launch on the chip: the late
    // sweeps of the SLDS ascent (a few hundred sequences still iterating) then run one wavefront per CU.
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int W = (a.B + cus - 1) / cus;
    W = W < 1 ? 1 : (W > 8 ? 8 : W);
    dim3 grid((a.B + W - 1) / W), block(64 * W);
    hipLaunchKernelGGL(kern, grid, block, (size_t)bytes, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
