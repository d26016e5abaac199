// lds_estep_twoend_rpcmix.hpp -- the SLDS local mean field's LDS step (svae_slds_lds_meanfield_f64) in the ROW-PER-CHAIN
// layout, with the K-state mixing and the K-state contraction taken OFF the elimination / smoother wavefronts and put on
// the matrix cores of PRODUCER wavefronts (round 5).
//
// What it replaces: /root/reference/svae/models/slds_svae.py:80-103 (lds_meanfield + get_var_lds_local_natparam) and
// the pair part of get_arhmm_local_nodeparams (:131-147) -- the same contract as the MIX mode of lds_estep_twoend.hpp
// (one sequence per wavefront; kept for K > 8 and as the A/B partner of this kernel).
//
// Why: the MIX kernel forms  sum_k w[t,k] (J11, J12, J22)_k  and  <E x x', P_k>  as table reads -- per wavefront step 160
// + 150 DPP multiply-adds fed by 155 16-byte LDS reads, on top of the elimination / smoother arithmetic: 861 + 607
// instructions per sequence step, two wavefronts per SIMD at 2048 sequences, 4.9 cycles per instruction -- 8.5 % of the
// fp64 peak.  Both are small GEMMs:
//     mixing       out[e, col] = sum_k  Ptab[e, k] * w[k, col]        e: the 2 n^2 entries a step needs, col: 16 chains
//     contraction  out[(q,k), col] = sum_e  Ptab[(q,k), e] * S[e, col]  e: the 2 n^2 entries of (E x x', E x_next x')
// Here a workgroup is FOUR consumer wavefronts -- the row-per-chain kernel of lds_estep_twoend_rpc.hpp: every DPP row one
// elimination chain, two sequences per wavefront, lean hand-off records -- and FOUR producer wavefronts on
// v_mfma_f64_16x16x4: the 16 columns of an MFMA are the workgroup's 16 chains (8 sequences x 2 chains), advancing in
// lock-step, one `s_waitcnt lgkmcnt(0); s_barrier` per step.
//   * elimination phase: the producers write, one step ahead, a 2-slot LDS ring [slot][column][ nat J12 (n^2) | C (n^2) ]
//     with C = -2 (J22' w_s + J11' w_{s+1}) in the column's own orientation (chain B: the two weight vectors swapped in
//     the B operand; J12' = J12^T is a transposed READ by the consumer lane).  A consumer step reads 2 n registers
//     (ds_read_b64) instead of running 160 multiply-adds on 80 table reads;
//   * smoother phase: producer 3 keeps the J12 ring filled (the lean record's G = -P^-1 J12 is rebuilt per step); the
//     consumers leave their tiles (S~_t rows < n and the cross moment W~) in a 3-slot ring [slot][column][2 n^2], and
//     producers 0 - 2 take one slot each in turn: two 16 x 16 x n^2 products (rows: [J11_k ; J22_k] on S~, [J12_k^T ;
//     J12_k] on W~ -- a chain-A column keeps the first half of the second product, a chain-B column the second), spread
//     over the two steps that follow, then the (T,2,K) outputs straight from the accumulators.
// Load balance: with four live consumers every producer shares a SIMD with one, and a step ends when the slowest SIMD
// is done -- so a tile-ring slot is contracted by ONE producer over the three steps after it was filled (13 MFMAs a
// step; four slots in flight, one per producer) and the producer that has no segment in a step forms the J12 mix (14):
// 13 | 13 | 13 | 14 MFMAs per smoother step instead of 14 | 25 | 14 | 0 (same-box A/B: 0.964 -> 0.91 ms at 2048 sequences).
// (Issue priorities, s_setprio 3 on the producers or on the consumers: +1 - 2 %, off.)
// Symmetry is used where it is free: C and S~ travel as lower triangles (the ring columns have the layout of a lean
// hand-off record, so the consumer lanes' symmetric reads / in-order overwriting stores need no selects), which cuts the
// products from 42 + 14 + 50 to 30 + 14 + 39 MFMAs per step pair -- fp64 MFMAs and the consumers' fp64 vector
// instructions contend for the SIMD (measured), so every MFMA saved is consumer time.
// The parameter tables live in the producers' REGISTERS (they are the A operands: <= 22 + 39 doubles per lane).
// LDS per workgroup (n = 10): 2 x 21.4 KB mixing ring + 3 x 21.6 KB tile ring + 24.6 KB transposition tiles = 132 KB
// (one workgroup per CU: up to 8 sequences, 8 wavefronts, two per SIMD -- a consumer and its producer).
// REFPROD: the producers compute the same ring contents / outputs with plain loops from global memory (no MFMA) -- slow,
// test infrastructure: isolates the lock-step protocol and the consumer side from the MFMA operand layouts.
// K <= 8 (the 2 K output rows of the contraction are one MFMA tile), n <= 10, T >= 4.
#pragma once
#include "lds_estep_twoend_rpc.hpp"

namespace svae {

typedef double rm_d4 __attribute__((ext_vector_type(4)));

template <int N>
struct RpcMixCfg {
  static constexpr int NN = N * N;
  static constexpr int TRI = N * (N + 1) / 2;
  static constexpr int NT = (NN + 15) / 16;             // 16-entry tiles of J12 (n^2 entries)
  static constexpr int NTC = (TRI + 15) / 16;           // 16-entry tiles of C (symmetric: its lower triangle)
  // mixing ring, one column: [ nat J12 (n^2) | C, lower triangle by rows (TRI) | zeros (n + 1) ] -- the C part has the
  // layout of a lean hand-off record: lane N reads entry TRI + i, lanes > N entry TRI + N, both zero
  static constexpr int CS = (NN + TRI + N + 1) | 1;     // column stride (odd: conflict-free tile stores)
  static constexpr int MIXSLOT = 16 * CS;
  // tile ring, one column: [ S~ lower triangle (TRI) | junk (n + 2: what lanes >= n store) | pad | W~ (n^2) ]
  static constexpr int WOFF = (TRI + N + 2 + 3) & ~3;   // W~ starts on a k-block boundary
  static constexpr int SS = (WOFF + NN) | 1;
  static constexpr int SSLOT = 16 * SS;
  static constexpr int NKB1 = (TRI + 3) / 4;            // 4-entry k-blocks of the first product (S~ triangle)
  static constexpr int NKB2 = (NN + 3) / 4;             //   ... of the second (W~)
  static constexpr int RSL = (N + 3) & ~1;              // row stride of a consumer's transposition tile
  static constexpr int TAB = 4 * 16 * RSL;              // per consumer wavefront
  static constexpr int OFF_S = 2 * MIXSLOT;
  static constexpr int SDEPTH = 4;                      // tile-ring slots (a launch uses 3 or 4: one per contraction producer)
  static constexpr int OFF_TAB = OFF_S + SDEPTH * SSLOT;
  static constexpr int LDS_DOUBLES = OFF_TAB + 4 * TAB;
};
constexpr long rpcmix_lds_bytes(int n) {
  const int nn = n * n, tri = n * (n + 1) / 2, cs = (nn + tri + n + 1) | 1, woff = (tri + n + 2 + 3) & ~3;
  const int ss = (woff + nn) | 1, rsl = (n + 3) & ~1;
  return 8L * (2 * 16 * cs + 4 * 16 * ss + 4 * 4 * 16 * rsl);
}
// (row, column) of entry e of a lower triangle stored by rows
__device__ __forceinline__ void rm_tri_rc(int e, int& i, int& c) {
  i = 0;
  while ((i + 1) * (i + 2) / 2 <= e) ++i;
  c = e - i * (i + 1) / 2;
}
constexpr int RPCMIX_MAX_K = 8;

// row of the arrays that launch position `pos` (0 .. 7) of workgroup `blk` works on; consecutive slots go to DIFFERENT
// workgroups (a launch with few live slots leaves one live sequence per CU); < 0: unused
__device__ __forceinline__ int rm_row_of(const LdsArgs& a, int pos, int blk, int G) {
  const int slot = pos * G + blk;
  if (slot >= a.B) return -1;
  return a.seq_index ? a.seq_index[slot] : slot;
}

// which wavefronts of the workgroup have work: bit w = consumer wavefront w has a live sequence
__device__ __forceinline__ int rm_live_waves(const LdsArgs& a, int blk, int G) {
  int m = 0;
  for (int pos = 0; pos < 8; ++pos) m |= (rm_row_of(a, pos, blk, G) >= 0) ? (1 << (pos >> 1)) : 0;
  return m;
}

// ---- producers, reference form (plain loops; test infrastructure) ------------------------------------------------------
template <int N>
__device__ void rm_producer_ref(const LdsArgs& a, double* mring, double* sring, const int pt, const int blk, const int G,
                                const int ncp) {
  using C = RpcMixCfg<N>;
  constexpr int NN = C::NN, TRI = C::TRI;
  const int T = a.T, K = a.mix_K, e = te_elims(T);
  const bool oddT = (T & 1) != 0;
  auto erow_of = [&](int j) { int r = rm_row_of(a, j >> 1, blk, G); if (r < 0) r = rm_row_of(a, (j >> 1) ^ 1, blk, G); return r; };
  auto wvec = [&](int j, int m) -> const double* {          // weights that mix local pair m of column j
    const int er = erow_of(j), dirj = j & 1;
    int node = dirj ? T - 1 - m : m + 1;
    node = node < 0 ? 0 : (node > T - 1 ? T - 1 : node);
    return a.mix_w + ((long)(er < 0 ? 0 : er) * T + node) * K;
  };
  auto mix = [&](int m, bool withC) {                       // E_m -> slot m & 1
    double* slot = mring + (m & 1) * C::MIXSLOT;
    for (int o = pt; o < 16 * (NN + TRI); o += 256) {
      const int j = o & 15, ee = o >> 4, dirj = j & 1;
      const double* w0 = wvec(j, m), *w1 = wvec(j, m + 1);
      double v = 0.0;
      if (ee < NN) {
        for (int k = 0; k < K; ++k) v = __builtin_fma(a.J12[(long)k * NN + ee], w0[k], v);
      } else if (withC) {
        int i, cq;
        rm_tri_rc(ee - NN, i, cq);
        const int e2 = i * N + cq;
        const double* x = dirj ? w1 : w0, *y = dirj ? w0 : w1;
        for (int k = 0; k < K; ++k)
          v = __builtin_fma(-2.0 * a.J22[(long)k * NN + e2], x[k], __builtin_fma(-2.0 * a.J11[(long)k * NN + e2], y[k], v));
      } else {
        continue;
      }
      slot[j * C::CS + ee] = v;
    }
  };
  auto contract = [&](int kk) {                             // tile-ring slot kk % ncp -> pair_contr rows of step s = e - kk
    const int j = pt & 15, r = pt >> 4, q = r >> 3, k = r & 7, dirj = j & 1;
    const int row = rm_row_of(a, j >> 1, blk, G);
    const double* src = sring + (kk % ncp) * C::SSLOT + j * C::SS;
    if (k >= K || row < 0) return;
    if (kk == 0 && !(oddT && dirj == 0)) return;            // the meeting node is reported by chain A, odd T only
    const double* Pq = (q ? a.J22 : a.J11) + (long)k * NN;
    const double* Px = a.J12 + (long)k * NN;
    double acc = 0.0;
    for (int ee = 0; ee < TRI; ++ee) {
      int i, cq;
      rm_tri_rc(ee, i, cq);
      acc = __builtin_fma(src[ee], i == cq ? Pq[i * N + i] : Pq[i * N + cq] + Pq[cq * N + i], acc);
    }
    if (q == dirj) {
      for (int ee = 0; ee < NN; ++ee) {
        const int i = ee / N, cq = ee % N;
        acc = __builtin_fma(src[C::WOFF + ee], dirj ? Px[ee] : Px[cq * N + i], acc);
      }
    }
    const int s = e - kk, t = dirj ? T - 1 - s : s;
    a.mix_out[(((long)row * T + t) * 2 + q) * K + k] = acc;
  };
  lds_barrier();                                            // #0: the rings are zeroed
  mix(0, true);
  lds_barrier();                                            // #1: E_0 is in slot 0
  for (int s = 0; s < e; ++s) {
    if (s + 1 <= e - 1) mix(s + 1, true);
    lds_barrier();
  }
  for (int k = 0; k <= e; ++k) {                            // smoother step k works on local step s = e - k
    const int s = e - k;
    if (k >= 1 && s >= 1) mix(s - 1, false);
    if (k >= 1) contract(k - 1);
    lds_barrier();
  }
  contract(e);
}

// ---- producers on the matrix cores ------------------------------------------------------------------------------------
// p: producer index 0 .. 3 (it shares a SIMD with consumer wavefront p).  `live_waves` = 1 (only consumer 0 has work: small
// launches, the late sweeps of the coordinate ascent): producers 1 - 3 share everything and producer 0 only keeps the
// barrier count -- fp64 MFMAs and the consumer's fp64 vector instructions contend for the SIMD (measured), so the one
// chain that is the launch's latency keeps its SIMD to itself.
template <int N, bool SOLO>
__device__ void rm_producer_mfma(const LdsArgs& a, double* mring, double* sring, const int p, const int lane, const int blk,
                                 const int G) {
  using C = RpcMixCfg<N>;
  constexpr int NN = C::NN, TRI = C::TRI, NT = C::NT, NTC = C::NTC, NKB1 = C::NKB1, NKB2 = C::NKB2;
  constexpr int NUC = (NTC + 2) / 3;                        // C tiles of one producer, at most (three or four share them)
  const int T = a.T, K = a.mix_K, e = te_elims(T);
  const bool oddT = (T & 1) != 0;
  constexpr bool solo = SOLO;
  constexpr int np = solo ? 3 : 4;                          // producers sharing the mixing tiles
  const int pr = solo ? p - 1 : p;                          // my rank among them (-1: none of it)
  constexpr int ncp = solo ? 3 : 4;                         // contraction producers = tile-ring slots in use
  const int cr = solo ? p - 1 : p;                          // my rank among them (-1: none)
  const int kq = lane >> 4, r16 = lane & 15;
  const int j = r16, dirj = j & 1;                          // this lane's column (B operand / C-D layout: column = lane % 16)
  const int row = rm_row_of(a, j >> 1, blk, G);
  const int sib = rm_row_of(a, (j >> 1) ^ 1, blk, G);
  const int erow = row >= 0 ? row : (sib >= 0 ? sib : 0);
  // ---- A operands (lane holds A[r16][4 kb + kq]) ------------------------------------------------------------------------
  double tj[NT][2];                                         // J12 tiles:  A[e][k] = nat J12_k[e]
  double tc[NUC][4];                                        // my C tiles (ti = pr + np u): A[e][0..7] = -2 J22_k, [8..15] = -2 J11_k
  static_for<0, NT>([&](auto ti) {
    const int ee = ti * 16 + r16;
    static_for<0, 2>([&](auto kb) {
      const int k = kb * 4 + kq;
      const bool ok = ee < NN && k < K;
      tj[ti][kb] = ok ? a.J12[ok ? (long)k * NN + ee : 0] : 0.0;
    });
  });
  static_for<0, NUC>([&](auto u) {
    const int ti = pr + np * u, ee = ti * 16 + r16;
    int i = 0, cq = 0;
    rm_tri_rc(ee < TRI && pr >= 0 ? ee : 0, i, cq);
    static_for<0, 4>([&](auto kb) {
      const int k = (kb & 1) * 4 + kq;
      const double* base = kb < 2 ? a.J22 : a.J11;
      const bool ok = pr >= 0 && ti < NTC && ee < TRI && k < K;
      tc[u][kb] = ok ? -2.0 * base[ok ? (long)k * NN + i * N + cq : 0] : 0.0;
    });
  });
  // contraction: rows r16 = (q, k).  First product [J11_k ; J22_k] on the lower triangle of S~ (off-diagonal entries
  // count twice: J[i][c] + J[c][i]); second [J12_k^T ; J12_k] on W~
  double a1[NKB1], a2[NKB2];
  {
    const int q = r16 >> 3, k = r16 & 7;
    const double* Pq = q ? a.J22 : a.J11;
    static_for<0, NKB1>([&](auto kb) {
      const int ee = kb * 4 + kq;
      const bool ok = cr >= 0 && ee < TRI && k < K;
      int i = 0, cq = 0;
      rm_tri_rc(ok ? ee : 0, i, cq);
      const long b0 = ok ? (long)k * NN : 0;
      a1[kb] = ok ? (i == cq ? Pq[b0 + i * N + i] : Pq[b0 + i * N + cq] + Pq[b0 + cq * N + i]) : 0.0;
    });
    static_for<0, NKB2>([&](auto kb) {
      const int ee = kb * 4 + kq, i = ee / N, cq = ee % N;
      const bool ok = cr >= 0 && ee < NN && k < K;
      a2[kb] = ok ? a.J12[ok ? (long)k * NN + (q ? ee : cq * N + i) : 0] : 0.0;
    });
  }
  // ---- weights: B operand, lane holds w[k = 4 kb + kq] of ITS column's node -------------------------------------------------
  const double* wcol = a.mix_w + (long)erow * T * K;
  const int k0 = kq < K ? kq : 0, k1 = kq + 4 < K ? kq + 4 : 0;
  const double m0 = kq < K ? 1.0 : 0.0, m1 = kq + 4 < K ? 1.0 : 0.0;
  auto node_of = [&](int m) { int nd = dirj ? T - 1 - m : m + 1; return nd < 0 ? 0 : (nd > T - 1 ? T - 1 : nd); };
  struct W2 { double v0, v1; };
  auto wload = [&](int m) { const double* q = wcol + (long)node_of(m) * K; return W2{q[k0] * m0, q[k1] * m1}; };
  double* const dstl = mring + j * C::CS + kq;
  const rm_d4 zero4 = {0.0, 0.0, 0.0, 0.0};
  // E_m (or, smoother phase, its J12 part) -> ring slot m & 1; `all_j`: this producer takes every J12 tile
  auto mix = [&](int m, const W2 wa, const W2 wb, bool withC, bool all_j) {
    double* dst = dstl + (m & 1) * C::MIXSLOT;
    static_for<0, NT>([&](auto ti) {
      if (all_j || (ti % np) == pr) {
        rm_d4 d = zero4;
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(tj[ti][0], wa.v0, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(tj[ti][1], wa.v1, d, 0, 0, 0);
        static_for<0, 4>([&](auto rr) {
          if (ti * 16 + 4 * rr + kq < NN) dst[ti * 16 + 4 * rr] = d[(int)rr];
        });
      }
    });
    if (withC && pr >= 0) {
      const double f0 = dirj ? wb.v0 : wa.v0, f1 = dirj ? wb.v1 : wa.v1;     // multiplies -2 J22
      const double g0 = dirj ? wa.v0 : wb.v0, g1 = dirj ? wa.v1 : wb.v1;     // multiplies -2 J11
      static_for<0, NUC>([&](auto u) {
        const int ti = pr + np * u;
        if (ti < NTC) {
          rm_d4 d = zero4;
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(tc[u][0], f0, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(tc[u][1], f1, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(tc[u][2], g0, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f64_16x16x4f64(tc[u][3], g1, d, 0, 0, 0);
          static_for<0, 4>([&](auto rr) {
            if (ti * 16 + 4 * rr + kq < TRI) dst[NN + ti * 16 + 4 * rr] = d[(int)rr];
          });
        }
      });
    }
  };
  // ---- contraction of one tile-ring slot = NKB1 + NKB2 MFMAs in a fixed order (first product, then second), cut into
  // ncp - 1 segments that the slot's producer runs in the ncp - 1 steps after the slot was filled; outputs after the last
  rm_d4 D1 = zero4, D2 = zero4;
  const double* const srcl = sring + j * C::SS + kq;
  constexpr int NKBT = NKB1 + NKB2;
  auto seg_run = [&](int kk, auto m0_c, auto m1_c) {
    constexpr int M0 = decltype(m0_c)::value, M1 = decltype(m1_c)::value;
    const double* src = srcl + (kk % ncp) * C::SSLOT;
    if constexpr (M0 == 0) { D1 = zero4; D2 = zero4; }
    static_for<M0, M1>([&](auto m) {
      if constexpr (m < NKB1) D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[m], src[m * 4], D1, 0, 0, 0);
      else D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[m - NKB1], src[C::WOFF + (m - NKB1) * 4], D2, 0, 0, 0);
    });
    if constexpr (M1 == NKBT) {
      // rows: register rr holds row 4 rr + kq = (q = rr >> 1, k = 4 (rr & 1) + kq); the cross term of a chain-A column is
      // rows 0 .. 7 of the second product (-> q = 0), of a chain-B column rows 8 .. 15 (-> q = 1)
      const bool own = kk != 0 || (oddT && dirj == 0);
      const int s = e - kk, t = dirj ? T - 1 - s : s;
      double* out = a.mix_out + ((long)(row < 0 ? 0 : row) * T + t) * 2 * K;
      static_for<0, 4>([&](auto rr) {
        const int q = rr >> 1, k = 4 * (rr & 1) + kq;
        const double v = D1[(int)rr] + ((q == dirj) ? D2[(int)rr] : 0.0);
        if (row >= 0 && own && k < K) out[q * K + k] = v;
      });
    }
  };
  // segment g (0-based) of slot kk; three producers: two segments per slot (halves: with the one-sequence consumer the
  // producers ARE the step's critical path -- 19 | 20 | 14 MFMAs instead of 14 | 25 | 14: 0.544 -> 0.504 ms), four: three
  constexpr int B3_1 = NKBT / 2, B4_1 = NKBT / 3, B4_2 = 2 * NKBT / 3;
  using I0 = std::integral_constant<int, 0>;
  using IT = std::integral_constant<int, NKBT>;
  auto segment = [&](int kk, int g) {
    if constexpr (ncp == 3) {
      if (g == 0) seg_run(kk, I0{}, std::integral_constant<int, B3_1>{});
      else seg_run(kk, std::integral_constant<int, B3_1>{}, IT{});
    } else {
      if (g == 0) seg_run(kk, I0{}, std::integral_constant<int, B4_1>{});
      else if (g == 1) seg_run(kk, std::integral_constant<int, B4_1>{}, std::integral_constant<int, B4_2>{});
      else seg_run(kk, std::integral_constant<int, B4_2>{}, IT{});
    }
  };

  lds_barrier();                                            // #0: the rings are zeroed
  W2 w_0 = wload(0), w_1 = wload(1), w_2 = wload(2), w_3 = wload(3);      // weights of local pairs m, m+1, m+2, m+3
  if (pr >= 0) mix(0, w_0, w_1, true, false);
  lds_barrier();                                            // #1: E_0 is in slot 0
  for (int s = 0; s < e; ++s) {
    const W2 w_4 = wload(s + 4 <= e ? s + 4 : e);           // (pairs beyond e are never used)
    if (pr >= 0 && s + 1 <= e - 1) mix(s + 1, w_1, w_2, true, false);
    w_0 = w_1; w_1 = w_2; w_2 = w_3; w_3 = w_4;
    lds_barrier();
  }
  // smoother phase: step k works on local step s = e - k
  W2 v_0 = wload(e >= 2 ? e - 2 : 0), v_1 = wload(e >= 3 ? e - 3 : 0), v_2 = wload(e >= 4 ? e - 4 : 0);
  for (int k = 0; k <= e; ++k) {
    const int s = e - k;
    const W2 v_3 = wload(s - 4 >= 0 ? s - 4 : 0);           // pair (s - 1) - 3
    if (cr >= 0) {
      // slot k - 1 - g is in its segment g now (g = 0 .. ncp - 2): at most one of them is mine; the producer that has
      // none this step (rank k % ncp) forms the J12 mix of local step s - 1 -- every step the same load on every SIMD
      for (int g = ncp - 2; g >= 0; --g) {
        const int kk = k - 1 - g;
        if (kk >= 0 && kk % ncp == cr) segment(kk, g);
      }
      if (k % ncp == cr && k >= 1 && s >= 1) mix(s - 1, v_0, v_0, false, true);
    }
    if (k >= 1) { v_0 = v_1; v_1 = v_2; v_2 = v_3; }
    lds_barrier();
  }
  if (cr >= 0) {                                            // the last slots' remaining segments (nobody overwrites them)
    for (int k = e + 1; k <= e + ncp - 1; ++k)
      for (int g = ncp - 2; g >= 0; --g) {
        const int kk = k - 1 - g;
        if (kk >= 0 && kk <= e && kk % ncp == cr) segment(kk, g);
      }
  }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
template <int N, bool REFPROD>
__global__ __launch_bounds__(512) void slds_meanfield_rpc_kernel(const LdsArgs a) {
  static_assert(N >= 1 && N <= TE_MAX_N, "latent dimension");
  using C = RpcMixCfg<N>;
  constexpr int NN = C::NN;
  constexpr int ZP = te_page_doubles(N), WS = te_lean_step_doubles(N);
  constexpr int TRI = N * (N + 1) / 2;    // lean record: [lower triangle of P^-1 | c (N) | 0.0 | trash | pad]
  constexpr int LZERO = TRI + N, LTRASH = TRI + N + 1;
  constexpr int HL = N;                   // lane of the h column (= the homogeneous coordinate's column in the smoother)
  constexpr int RSL = C::RSL;
  extern __shared__ double rm_lds[];
  double* const mring = rm_lds;
  double* const sring = rm_lds + C::OFF_S;

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int blk = blockIdx.x, G = gridDim.x;
  const int T = a.T, K = a.mix_K;
  const int e = te_elims(T);              // eliminations per chain; the meeting node is local index e
  const int NBAR = 2 * e + 3;             // barriers every wavefront of the workgroup executes

  const int live_waves = rm_live_waves(a, blk, G);
  if (live_waves == 0) return;            // an oversized launch: nothing listed for this workgroup (no barrier executed yet)
  for (int q = threadIdx.x; q < C::LDS_DOUBLES; q += 512) rm_lds[q] = 0.0;   // (NaN bit patterns left by an earlier kernel
                                                                             //  must not meet the zero table entries)
  if (wv >= 4) {
#ifdef SVAE_RM_PRODPRIO
    __builtin_amdgcn_s_setprio(SVAE_RM_PRODPRIO);      // (experiment: the producers' issue priority against their SIMD's consumer)
#endif
    if constexpr (REFPROD) rm_producer_ref<N>(a, mring, sring, (wv - 4) * 64 + lane, blk, G, live_waves == 1 ? 3 : 4);
    else if (live_waves == 1) rm_producer_mfma<N, true>(a, mring, sring, wv - 4, lane, blk, G);
    else rm_producer_mfma<N, false>(a, mring, sring, wv - 4, lane, blk, G);
    return;
  }

#ifdef SVAE_RM_CONSPRIO
  __builtin_amdgcn_s_setprio(SVAE_RM_CONSPRIO);        // (experiment: the consumers' issue priority against their SIMD's producer)
#endif
  // ---- consumers: lds_estep_twoend_rpc.hpp with per-step parameters from the ring -----------------------------------------------
  const int c = lane & 15;
  const int g = lane >> 4;
  const int dir = g & 1;                  // 0: chain A (forward in time), 1: chain B (reversed)
  const int sq = g >> 1;                  // which of the wavefront's two sequences
  const int r0 = rm_row_of(a, 2 * wv, blk, G), r1 = rm_row_of(a, 2 * wv + 1, blk, G);
  if (r0 < 0 && r1 < 0) {                 // nothing to do here: keep the workgroup's barrier count
    for (int q = 0; q < NBAR; ++q) lds_barrier();
    return;
  }
  // a dead position repeats its live sibling (same values to the same addresses, as the odd-batch tail of the rpc kernel)
  const int b = sq ? (r1 >= 0 ? r1 : r0) : (r0 >= 0 ? r0 : r1);
  const int colj = 4 * wv + g;            // this chain's column of the rings
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int jx = T - 1 - e;               // eliminations done when the partner's message is taken
  const bool oddT = (T & 1) != 0;

  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EH = (c == HL) ? 1.0 : 0.0;
  const double EN = (c == N) ? 1.0 : 0.0;

  // ring addresses of this lane (bytes from the start of a mixing-ring slot):
  //   offJ[k]: nat J12'[k][c] -- chain A entry k n + c, chain B (J12' = J12^T) entry c n + k; lanes >= n: a zero entry
  //   offC[i]: C[i][c] = C[c][i]: entry tri(max, min) of the column's triangle; lane n: TRI + i, lanes > n: TRI + n (zeros)
  unsigned offJ[N], offC[N];
  static_for<0, N>([&](auto k) {
    offJ[k] = 8u * (unsigned)(colj * C::CS + (col ? (dir ? c * N + k : k * N + c) : NN + TRI));
    const int hi = k > c ? k : c, lo = k > c ? c : k;
    offC[k] = 8u * (unsigned)(colj * C::CS + NN + (col ? hi * (hi + 1) / 2 + lo : ((c == N) ? TRI + k : TRI + N)));
  });
  const char* const mbase = reinterpret_cast<const char*>(mring);
  auto ringJ = [&](unsigned slotbytes, double (&dst)[N]) {
    static_for<0, N>([&](auto k) { dst[k] = *reinterpret_cast<const double*>(mbase + slotbytes + offJ[k]); });
  };
  auto ringC = [&](unsigned slotbytes, double (&dst)[N]) {
    static_for<0, N>([&](auto i) { dst[i] = *reinterpret_cast<const double*>(mbase + slotbytes + offC[i]); });
  };

  // An: lanes < N = pivot block of the next node without its node potential (incoming message + J11' of the pair
  // ahead), lane N = incoming potential vector, other lanes zero.  The init potential is mixed by E[z_0], J11' of the
  // chain's pair 0 by its own node's weights (once per sequence: straight from global memory)
  double An[N];
  {
    const double* w0 = a.mix_w + (long)b * T * K;
    const int nd0 = dir ? T - 1 : 1;
    const double* wq = w0 + (long)nd0 * K;
    static_for<0, N>([&](auto i) { An[i] = 0.0; });
    for (int k = 0; k < K; ++k) {
      const double wi = dir ? 0.0 : w0[k], wp = wq[k];
      const double* ij = a.init_J + (long)k * NN, *ih = a.init_h + (long)k * N;
      const double* j11 = (dir ? a.J22 : a.J11) + (long)k * NN;
      static_for<0, N>([&](auto i) {
        const double v = -2.0 * (wi * ij[i * N + cc] + wp * j11[i * N + cc]);
        An[i] += col ? v : ((c == HL) ? wi * ih[i] : 0.0);
      });
    }
  }

  // node potentials of local step s: global node t = s (A) / T-1-s (B); lanes >= N read element 0
  const double* nJb = a.node_J + ((long)b * T) * N + cc;
  const double* nhb = a.node_h + ((long)b * T) * N + cc;
  auto node_off = [&](int s) -> long { return (long)(dir ? T - 1 - s : s) * N; };

  // chain workspace (layout of lds_estep_twoend.hpp, lean records): constant page [e_N (N+2) | zeros (N+2) | trash (2)], records.
  // Addressed as the UNIFORM base a.ws + a 32-bit byte offset per lane (its sequence, chain and entry): the launcher
  // checks that the whole workspace is below 4 GiB (the two sequences of a wavefront are arbitrary rows)
  double* const wsb = a.ws;
  const unsigned choff = (unsigned)((long)b * te_seq_doubles(N, T) + (long)dir * te_chain_doubles(N, T)) + ZP;   // doubles
  double* zpage = a.ws + (long)b * te_seq_doubles(N, T) + (long)dir * te_chain_doubles(N, T);
  double* rec0 = zpage + ZP;
  double* trash = zpage + 2 * (N + 2);
  if (c < N + 2) { zpage[c] = EN; zpage[N + 2 + c] = 0.0; }
  // hand-off store of register i: lane c <= i -> tri(i) + c, lane N -> TRI + i, the others -> the record's trash entry
  unsigned loff[N];
  static_for<0, N>([&](auto i) {
    loff[i] = 8u * (choff + (unsigned)((c <= i) ? i * (i + 1) / 2 + c : ((c == HL) ? TRI + i : LTRASH)));   // bytes
  });
  for (int r = c; r <= e; r += 16) rec0[(long)r * WS + LZERO] = 0.0;   // the records' zero entry

  double qacc = 0.0;        // lane N: sum_t h' P^-1 h
  double ldM = 1.0;         // per lane c < N: running product of -1/p_c (log|P| = -sum log|.|)
  int ldE = 0;
  double vworst = -1.0;     // max over steps of -1/p_c (>= 0 <=> some pivot was not positive)

  double Jo_n = nJb[node_off(0)];
  double ho_n = nhb[node_off(0)];
  double Mp[N];             // partner chain's An at the hand-over point
  static_for<0, N>([&](auto i) { Mp[i] = 0.0; });
  double qacc_s = 0.0, ldM_s = 1.0;
  int ldE_s = 0;
  auto take_partner = [&]() {
    static_for<0, N>([&](auto i) { Mp[i] = __shfl_xor(An[i], 16); });
    qacc_s = qacc; ldM_s = ldM; ldE_s = ldE;
  };
  auto hand_off = [&](int s, const double (&M)[N], double vfull) {
    char* w = reinterpret_cast<char*>(wsb + (long)s * WS);       // uniform base + 32-bit lane offset
    static_for<0, N>([&](auto i) {
      asm volatile("" : "+v"(loff[i]));
      *reinterpret_cast<double*>(w + loff[i]) = M[i] * vfull;
    });
  };

  const bool live = c <= N;               // lanes N+1 .. 15 carry nothing: switched off (EXEC) in the two loops
  double qacc_m = 0.0, vfull_m = col ? 0.0 : 1.0;
  double Cc[N];                           // C of the current step (the last one is the meeting node's correction)
  static_for<0, N>([&](auto i) { Cc[i] = 0.0; });

  lds_barrier();                          // #0: the rings are zeroed
  lds_barrier();                          // #1: the producers have left step 0's parameters in slot 0
  if (live) {
    auto elim_step = [&](int s, auto slot_c) {
      constexpr unsigned SLOTB = 8u * (unsigned)(decltype(slot_c)::value * C::MIXSLOT);
      if (s == jx) take_partner();
      double NJ[N];
      ringJ(SLOTB, NJ);                   //   lanes < N: nat J12'[k][c]; other lanes 0
      ringC(SLOTB, Cc);                   //   lanes < N: info-form J22'(pair s) + J11'(pair s + 1); other lanes 0
      const double JoX = col ? -2.0 * Jo_n : 1.0;
      double ho = ho_n;
      Jo_n = nJb[node_off(s + 1)];        // s + 1 <= e: the meeting node's potentials included
      ho_n = nhb[node_off(s + 1)];

      // condition on the node potential; the right-hand sides (info-form J12' = -nat) in the second register set
      double MA[N], MB[N], Bt[N];
      static_for<0, N>([&](auto i) { MA[i] = __builtin_fma(JoX, E[i], An[i]); });
      dpp_fence(ho);
      static_for<0, N>([&](auto i) { mac_bc<i>(MA[i], ho, EH); });      // lane N: h_filt = h_pred + h_node
      // B operand of the Schur stage: lanes < N: -J12'[k][c] (info form) = nat; lane N: -h_filt,k
      static_for<0, N>([&](auto k) {
        Bt[k] = __builtin_fma(-EH, MA[k], NJ[k]);
        MB[k] = asm_neg(NJ[k]);
      });
      dpp_fence(MA);

      double vfull = col ? 0.0 : 1.0;
      gauss_jordan_2r_asm<N>(MA, MB, E, qacc, vfull);

      // next pivot block: An[i] = Cc[i] + sum_k X[k][i] * Bt[k]   (X[k][i] = lane i of MB[k]); row i in register i
      static_for<0, N>([&](auto i) { An[i] = Cc[i]; });
      asm volatile("s_nop 1");
      static_for<0, N>([&](auto k) {
        static_for<0, N>([&](auto i) { mac_bc<i>(An[i], MB[k], Bt[k]); });
      });

      vworst = fmax(vworst, vfull);
      ldM *= vfull;
      if ((s & 3) == 3) {
        ldE += __builtin_amdgcn_frexp_exp(ldM);
        ldM = __builtin_amdgcn_frexp_mant(ldM);
      }
      hand_off(s, MA, vfull);
      lds_barrier();                      // step s + 1's parameters are in the other slot
    };
    int s = 0;
    for (; s + 1 < e; s += 2) {           // two steps per trip: the ring slot is a compile-time constant
      elim_step(s, std::integral_constant<int, 0>{});
      elim_step(s + 1, std::integral_constant<int, 1>{});
    }
    if (s < e) elim_step(s, std::integral_constant<int, 0>{});
    if (jx == e) take_partner();

    // ---- meeting node: P_m = An_own + An_partner - (J22' + J11') + node; right-hand side h alone ----------------------
    {
      const double JoX = col ? -2.0 * Jo_n : 1.0;
      double ho = ho_n;
      double M[N];
      static_for<0, N>([&](auto i) { M[i] = __builtin_fma(JoX, E[i], (An[i] + Mp[i]) - Cc[i]); });
      dpp_fence(ho);
      static_for<0, N>([&](auto i) { mac_bc<i>(M[i], ho, EH); });
      dpp_fence(M);
      gauss_jordan_1r_asm<N>(M, E, qacc_m, vfull_m);
      hand_off(e, M, vfull_m);
    }
  }

  // ---- log-normaliser (without the mixed constants: the caller's) -----------------------------------------------------------
  {
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
    const double chain_total = chain_part(qacc, ldM, ldE) + __shfl_xor(chain_part(qacc_s, ldM_s, ldE_s), 16);
    double z = 0.0;
    if (a.node_logZ) {
      for (int t = c; t < T; t += 16) z += a.node_logZ[(long)b * T + t];
    }
    const double total = row_sum16(z) + chain_total + meet_total;
    if (c == 0 && dir == 0) a.lognorm[b] = total;
    const bool lane_bad = col && (!(vworst < 0.0) || !(vfull_m < 0.0));
    const unsigned long long bal = __ballot(lane_bad);
    const bool bad = ((bal >> (32 * sq)) & 0xffffffffull) != 0 || !(total == total);
    if (bad && c == 0 && dir == 0) {   // rare path: keep the smallest failing index (+1); 0 = ok
      int old = *(volatile int32_t*)a.info;
      while (old == 0 || old > b + 1) {
        const int seen = atomicCAS(a.info, old, b + 1);
        if (seen == old) break;
        old = seen;
      }
    }
  }

  // ---- smoother phase: moment form on homogeneous coordinates, local steps e, e-1, .., 0 ------------------------------
  double S[N + 1];
  static_for<0, N + 1>([&](auto i) { S[i] = (i == N) ? EN : 0.0; });
  dpp_fence(S);
  const bool skip2nd = dir && !oddT;      // chain B, even T: its first smoother step repeats pair e-1, which chain A counts
  const double wsp = skip2nd ? 0.0 : 1.0;
  const bool own_e = oddT && !dir;        // who reports the meeting node

  // node statistics: unconditional stores through per-lane walking pointers (idle lanes -> trash)
  const long nstride = dir ? N : -N;      // towards smaller s
  double* pdg = trash;
  double* pex = trash + 1;
  auto node_ptrs = [&](int s) {
    const long o = ((long)b * T + (dir ? T - 1 - s : s)) * N + c;
    pdg = col ? a.E_node_diagxx + o : trash;
    pex = col ? a.E_node_x + o : trash + 1;
  };
  if (own_e) node_ptrs(e);

  // operands of one step: Pi[i] = [P^-1 | c][i][c] of the lean record (lane N: c_i; lanes > N: its zero entry),
  // Pd = P^-1[c][c] (lanes < N; else the zero entry).  Prefetched one step ahead, every load unconditional.
  struct Ops { double Pi[N]; double Pd; };
  unsigned poff[N], pdoff;
  static_for<0, N>([&](auto i) {
    const int hi = i > c ? i : c, lo = i > c ? c : i;
    poff[i] = 8u * (choff + (unsigned)(col ? hi * (hi + 1) / 2 + lo : ((c == N) ? TRI + i : LZERO)));   // bytes
  });
  pdoff = 8u * (choff + (unsigned)(col ? c * (c + 1) / 2 + c : LZERO));
  const double* lrec = wsb + (long)e * WS;           // uniform record pointer
  int nextrec = e;                                   // index of the record the next load_ops fetches (.., 1, 0, 0, ..)
  auto load_ops = [&](Ops& o) {
    static_for<0, N>([&](auto i) {
      asm volatile("" : "+v"(poff[i]));
      o.Pi[i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lrec) + poff[i]);
    });
    asm volatile("" : "+v"(pdoff));
    o.Pd = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lrec) + pdoff);
    lrec -= nextrec > 0 ? WS : 0;
    nextrec -= nextrec > 0 ? 1 : 0;
  };
  double* tb = rm_lds + C::OFF_TAB + wv * C::TAB + g * 16 * RSL;
  // lanes > N (zeros) write the row's padding column N + 1 (RSL >= N + 2); row N (= e_N: constant) is written once here
  double* tbw = tb + (c <= N ? c : N + 1);
  if (c <= N) tb[N * RSL + c] = EN;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // tile ring, this lane's entries of its column:  S~[i][c] -> entry tri(max, min) of the triangle -- the stores of
  // register i hit row i of the triangle from lanes c <= i and, from lanes c > i, the mirror entries of rows c, which
  // the LATER store of register c overwrites with its own (lower-triangle) value: a wavefront's LDS stores retire in
  // order, so the ring ends up holding the lower triangle, deterministically; lane n -> TRI + i, lanes > n -> TRI + n
  // (junk the products multiply by zero).  W~[i][c] (lanes < n) -> WOFF + i n + c.
  unsigned soff[N];
  static_for<0, N>([&](auto i) {
    const int hi = i > c ? i : c, lo = i > c ? c : i;
    soff[i] = 8u * (unsigned)(colj * C::SS + (col ? hi * (hi + 1) / 2 + lo : ((c == N) ? TRI + i : TRI + N)));
  });
  char* const sring_b = reinterpret_cast<char*>(sring);
  char* const wbase = sring_b + 8u * (unsigned)(colj * C::SS + C::WOFF + cc);

  // one smoother step.  KIND: 0 generic, 1 first (meeting record: G = 0), 2 second (weight of the repeated pair);
  // kslot: tile-ring slot of this step (its index in the phase modulo the ring depth), s: the local step (J12 ring slot s & 1)
  auto step = [&](auto kind, int s, int kslot, Ops& cur, Ops& fill) {
    constexpr int KIND = decltype(kind)::value;
    load_ops(fill);                          // the next record, into the stage the previous step used
    // G~ rows: X[i][c] = sum_k P^-1[i][k] J12'[k][c] (lanes < N), c_i (lane N); row N = e_N (constant, in the tile)
    double Gc[N], H[N + 1];
    static_for<0, N>([&](auto i) { Gc[i] = EN * cur.Pi[i]; });
    if (KIND != 1) {
      double NJ[N];
      ringJ(8u * (unsigned)((s & 1) * C::MIXSLOT), NJ);
      dpp_fence(cur.Pi);
      static_for<0, N>([&](auto k) {
        static_for<0, N>([&](auto i) { mac_bc<k, true>(Gc[i], cur.Pi[i], NJ[k]); });
      });
    }
    // transposed copy through LDS: H[k][lane c] = G~[c][k]
    __builtin_amdgcn_wave_barrier();
    static_for<0, N>([&](auto i) { tbw[i * RSL] = Gc[i]; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    static_for<0, (N + 2) / 2>([&](auto q) {
      const double2 v = reinterpret_cast<const double2*>(tb + c * RSL)[q];
      H[2 * q] = v.x;
      if constexpr (2 * q + 1 <= N) H[2 * q + 1] = v.y;
    });
    __builtin_amdgcn_wave_barrier();

    // W~[i] = S~[i] G~'  :  sum_k -/+ bcast_k(S[i]) H[k]      (rows 0 .. N)
    double W[N + 1];
    static_for<0, N + 1>([&](auto i) { W[i] = 0.0; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, N + 1>([&](auto i) { mac_bc<k, (k < N)>(W[i], S[i], H[k]); });
    });
    dpp_fence(W);
    // S~_t[i] = P^-1[i] + G~[i] W~ = Pi + sum_k -/+ bcast_k(Gc[i]) W[k]   (rows < N; row N of G~ = e_N: S~_t[N] = W~[N])
    double Sn[N + 1];
    static_for<0, N>([&](auto i) { Sn[i] = __builtin_fma(-EN, cur.Pi[i], cur.Pi[i]); });
    Sn[N] = W[N];
    dpp_fence(Gc);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, N>([&](auto i) { mac_bc<k, (k < N)>(Sn[i], Gc[i], W[k]); });
    });
    // diag E[x_t x_t'] is lane-local: S~_t[c][c] = P^-1[c][c] + sum_k -/+ G~[c][k] W~[k][c], G~[c][k] = H[k] in lane c
    double dg = cur.Pd;
    static_for<0, N + 1>([&](auto k) {
      if constexpr (k < N) dg = __builtin_fma(-H[k], W[k], dg); else dg = __builtin_fma(H[k], W[k], dg);
    });

    // this node's tiles for the contraction with the K parameter sets (producers): S~ rows < N, and the cross moment
    // W~ rows < N -- zero where the pair is not this chain's to count (the meeting record: G = 0 gives W~ rows < N = 0)
    {
      const unsigned sb = 8u * (unsigned)(kslot * C::SSLOT);
      static_for<0, N>([&](auto i) { *reinterpret_cast<double*>(sring_b + sb + soff[i]) = Sn[i]; });
      if (col) {
        static_for<0, N>([&](auto i) {
          *reinterpret_cast<double*>(wbase + sb + 8 * (i * N)) = (KIND == 2) ? wsp * W[i] : W[i];
        });
      }
    }

    *pdg = dg;
    *pex = Sn[N];
    if constexpr (KIND == 1) node_ptrs(e - 1);
    else { pdg += col ? nstride : 0; pex += col ? nstride : 0; }
    static_for<0, N + 1>([&](auto i) { S[i] = Sn[i]; });
    lds_barrier();                           // the tiles are in the ring; the next step's J12 is in its slot
  };

  if (live) {
    Ops R0, R1;
    constexpr std::integral_constant<int, 0> GEN{};
    load_ops(R0);                                                 // record e
    int kslot = 0;
    const int sdepth = live_waves == 1 ? 3 : 4;                   // tile-ring slots in use (= contraction producers)
    auto nxt = [&]() { const int r = kslot; kslot = kslot == sdepth - 1 ? 0 : kslot + 1; return r; };
    step(std::integral_constant<int, 1>{}, e, nxt(), R0, R1);     // local step e (fetches record e - 1)
    step(std::integral_constant<int, 2>{}, e - 1, nxt(), R1, R0); // e - 1
    int s = e - 2;                                                // (e >= 2: T >= TE_MIN_T)
    for (; s >= 1; s -= 2) {           // two steps per trip: the stages ping-pong, no copies, no branch inside
      step(GEN, s, nxt(), R0, R1);
      step(GEN, s - 1, nxt(), R1, R0);
    }
    if (s == 0) step(GEN, 0, nxt(), R0, R1);
  }

  // ---- E_init: S~ at chain A's end node x_0 ----------------------------------------------------------------------------------
  if (!dir && col) {
    double* ei = a.E_init + (long)b * (NN + N);
    static_for<0, N>([&](auto i) { ei[i * N + c] = S[i]; });
    ei[NN + c] = S[N];
  }
}

// ---- one sequence per workgroup (launches of at most one sequence per CU: the late sweeps of the coordinate ascent) ------
// The row-per-chain consumer issues every instruction for TWO sequences; a lone sequence pays for both (1070 instructions
// per step pair).  Here the consumer is the ONE-sequence two-ended body (lds_estep_twoend.hpp, RING: both chains in one
// instruction stream, one-register Gauss-Jordan, lean records) -- wavefront 0 -- next to three producers (wavefronts 1 - 3,
// the solo split of rm_producer_mfma): four wavefronts on four SIMDs, the consumer alone on its own.
template <int N>
__global__ __launch_bounds__(256) void slds_meanfield_seq_kernel(const LdsArgs a) {
  static_assert(N >= 1 && N <= TE_MAX_N, "latent dimension");
  using C = RpcMixCfg<N>;
  extern __shared__ double rm_lds[];
  double* const mring = rm_lds;
  double* const sring = rm_lds + C::OFF_S;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x, G = gridDim.x;             // G = a.B: position 0 of workgroup blk is slot blk, the rest unused
  if (rm_row_of(a, 0, blk, G) < 0) return;               // unused slot: the whole workgroup leaves before its first barrier
  for (int q = threadIdx.x; q < C::OFF_TAB; q += 256) rm_lds[q] = 0.0;
  if (wv >= 1) {
    rm_producer_mfma<N, true>(a, mring, sring, wv, lane, blk, G);
    return;
  }
  TeRing ring;
  ring.m = mring; ring.s = sring;
  ring.cs = C::CS; ring.mixslot = C::MIXSLOT;
  ring.ss = C::SS; ring.sslot = C::SSLOT; ring.woff = C::WOFF;
  ring.depth = 3;
  lds_estep_twoend_body<N, true, true, false, false, false, true>(a, blk, ring);
}

// refprod: 0 = MFMA producers; 1 = reference producers (plain loops: test infrastructure).  seq_ok: launches of at most
// one sequence per CU may take the one-sequence consumer (0: always the row-per-chain consumers)
template <int N>
static int launch_slds_meanfield_rpc(const LdsArgs& a, int refprod, int seq_ok, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    if (a.mix_K < 1 || a.mix_K > RPCMIX_MAX_K || a.T < TE_MIN_T) return -30;
    const long bytes = rpcmix_lds_bytes(N);
    static_assert(rpcmix_lds_bytes(N) == 8L * RpcMixCfg<N>::LDS_DOUBLES, "LDS size");
    static LdsGrant grant_mfma, grant_ref;
    // sequences per workgroup: only as many as it takes to place the launch on the chip (a workgroup always carries its
    // four producers; a lone sequence on a CU runs at the latency of its own chain -- the late sweeps of the ascent)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    if (!refprod && seq_ok && a.B <= cus) {                // at most one sequence per CU: the one-sequence consumer
      static LdsGrant grant_seq;
      auto kern = slds_meanfield_seq_kernel<N>;
      const long sbytes = 8L * RpcMixCfg<N>::OFF_TAB;
      if (!grant_seq.ensure(reinterpret_cast<const void*>(kern), sbytes)) return -31;
      hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), (size_t)sbytes, stream, a);
      return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    int W = (a.B + cus - 1) / cus;
    W = W < 1 ? 1 : (W > 8 ? 8 : W);
    const int grid = (a.B + W - 1) / W;
    if (refprod) {
      auto kern = slds_meanfield_rpc_kernel<N, true>;
      if (!grant_ref.ensure(reinterpret_cast<const void*>(kern), bytes)) return -31;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)bytes, stream, a);
    } else {
      auto kern = slds_meanfield_rpc_kernel<N, false>;
      if (!grant_mfma.ensure(reinterpret_cast<const void*>(kern), bytes)) return -31;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)bytes, stream, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
