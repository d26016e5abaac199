// lds_estep_twoend_s4.hpp -- smoother phase of the two-ended E-step with ONE CHAIN PER WAVEFRONT (small batches).
//
// The two-ended kernel (lds_estep_twoend.hpp) runs both chains of a sequence in one instruction stream, two DPP
// rows per chain.  Up to 512 sequences that leaves half of the chip's 1024 SIMDs idle.  After the meeting node the
// two chains' smoothers are independent, so the workgroup carries a second wavefront that sleeps at a barrier
// through the elimination phase and then takes chain B's smoother while the first keeps chain A's: each chain now
// owns all FOUR DPP rows of its wavefront, every product stage is split by output row over four rows instead of
// two (row i of a tile lives in DPP row i & 3, slot i >> 2): 96 instead of 182 DPP multiply-adds per step, the
// all-gather of a slot tile is v_permlane16_swap + v_permlane32_swap.  Same recursion, same accumulation order per
// output element as the two-row smoother (moment form on homogeneous coordinates:
// cython_lds_inference.pyx:149-210 is what it replaces), same lean hand-off records.
// Homogeneous parameters, lean records, statistics summed over time (the headline configuration).
#pragma once
#include "lds_estep_kernel.hpp"

namespace svae {

// x: one value per DPP row  ->  the four rows' values, each replicated over the wavefront.
// v_permlane16_swap(x, x) leaves the even rows' values replicated over their row pairs in its first result and the
// odd rows' in its second; v_permlane32_swap(y, y) replicates the lower / upper half of y.
__device__ __forceinline__ void quad_gather(double x, double& r0, double& r1, double& r2, double& r3) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const auto el = __builtin_amdgcn_permlane32_swap(pl[0], pl[0], false, false);   // rows 0 | 2
  const auto eh = __builtin_amdgcn_permlane32_swap(ph[0], ph[0], false, false);
  const auto ol = __builtin_amdgcn_permlane32_swap(pl[1], pl[1], false, false);   // rows 1 | 3
  const auto oh = __builtin_amdgcn_permlane32_swap(ph[1], ph[1], false, false);
  r0 = __hiloint2double(eh[0], el[0]);
  r2 = __hiloint2double(eh[1], el[1]);
  r1 = __hiloint2double(oh[0], ol[0]);
  r3 = __hiloint2double(oh[1], ol[1]);
}

__device__ __forceinline__ double reg_copy(double x) {       // a copy the compiler cannot fold away
  double y;
  asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));
  return y;
}

// dir = the chain (0: A, forward in time; 1: B, reversed) = the wavefront's index in the workgroup.
// tabw: 16 x RSL doubles of LDS of this wavefront, zero-initialised;  xch: exchange buffer of the workgroup.
// Both wavefronts of the workgroup call this function; it contains ONE __syncthreads().
// lrecs / keep: the last `keep` records of each chain (local steps e+1-keep .. e, the ones read first) live in LDS
// ([chain][slot][WS] doubles, written by the elimination phase) instead of the HBM workspace; keep = 0 or >= 2.
template <int N>
__device__ __forceinline__ void te_smooth4(const LdsArgs& a, const int b, const int dir, const int lane,
                                           double* tabw, double* xch, const double* lrecs, const int keep) {
  constexpr int ZP = te_page_doubles(N), WS = te_lean_step_doubles(N);
  constexpr int TRI = N * (N + 1) / 2, LZERO = TRI + N;
  constexpr int J = (N + 3) / 4;          // slots holding rows 0..N-1 (row i = 4j + r)
  constexpr int J1 = (N + 4) / 4;         // slots holding rows 0..N
  constexpr int RSL = (N + 3) & ~1;       // LDS row stride of the transposition tile
  constexpr int NS = N >> 2, NR = N & 3;  // row N of the homogeneous tile: slot NS of DPP row NR
  const int c = lane & 15, r = lane >> 4;
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int T = a.T;
  const int e = te_elims(T);
  const bool oddT = (T & 1) != 0;
  const double EN = (c == N) ? 1.0 : 0.0;

  double* wsb = a.ws + (long)b * te_seq_doubles(N, T);
  const unsigned choff = (unsigned)(dir * te_chain_doubles(N, T)) + ZP;
  double* zpage = wsb + (long)dir * te_chain_doubles(N, T);
  double* trash = zpage + 2 * (N + 2);

  // nat J12' of the chain's orientation: NJ12c[k][c] = J12'[k][c] (chain A: J12[k][c]; chain B: J12[c][k])
  const double* q12 = a.J12 + (long)b * a.pair_seq_stride;
  const int si = dir ? 1 : N, sc = dir ? N : 1;
  double NJ12c[N];
  static_for<0, N>([&](auto k) { const double v = q12[k * si + cc * sc]; NJ12c[k] = col ? v : 0.0; });

  double ED[J];                           // ED[j][c] = (c == 4j+r): picks S[i][i] in slot layout
  static_for<0, J>([&](auto j) { ED[j] = (c == 4 * j + r && c < N) ? 1.0 : 0.0; });
  const bool own_N = (r == NR);
  const double CN = own_N ? EN : 0.0;     // row N of G~ = e_N
  double S[J1];
  static_for<0, J1>([&](auto j) { S[j] = (j == NS) ? CN : 0.0; });
  dpp_fence(S);
  double sumS[J], sumW[J], Stop[J];
  static_for<0, J>([&](auto j) { sumS[j] = 0.0; sumW[j] = 0.0; Stop[j] = 0.0; });
  const bool skip2nd = dir && !oddT;      // chain B, even T: its first smoother step repeats pair e-1, which chain A counts
  const double wsp = skip2nd ? 0.0 : 1.0;
  const bool own_e = oddT && !dir;        // who reports the meeting node

  // node statistics: unconditional stores through per-lane walking pointers (idle lanes -> trash)
  const bool dlane = col && (c & 3) == r, xlane = col && own_N;
  const long nstride = dir ? N : -N;      // towards smaller s
  double* pdg = trash;
  double* pex = trash + 1;
  auto node_ptrs = [&](int s) {
    const long o = ((long)b * T + (dir ? T - 1 - s : s)) * N + c;
    pdg = dlane ? a.E_node_diagxx + o : trash;
    pex = xlane ? a.E_node_x + o : trash + 1;
  };
  if (own_e) node_ptrs(e);

  // operands of a step: Pi[j] = [P^-1 | c][4j+r][c] of the lean record (lane N: c_i; rows >= N, lanes > N: its zero entry)
  struct Ops { double Pi[J1]; };
  unsigned poff[J1];
  static_for<0, J1>([&](auto j) {
    const int i = 4 * j + r;
    const int hi = i > c ? i : c, lo = i > c ? c : i;
    poff[j] = 8u * (choff + ((i < N && col) ? hi * (hi + 1) / 2 + lo : ((i < N && c == N) ? TRI + i : LZERO)));   // bytes
  });
  // Records are fetched FOUR steps ahead into a ring of register stages (3 registers each): a step is ~200
  // instructions (~0.45 us) and the records, written by the other wavefront during the elimination phase, are an L2
  // round trip away -- one step of lookahead left the loop bound by that latency.
  const int s0 = e + 1 - keep;                       // records s >= s0 are in LDS, records < s0 in the HBM workspace
  const int g0 = s0 > 0 ? s0 - 1 : 0;                // first record the HBM ring fetches
  const double* lrec = wsb + (long)g0 * WS;
  int nextrec = g0;                                  // index of the record the next load_ops fetches (.., 1, 0, 0, ..)
  auto load_ops = [&](Ops& o) {
    // (offsets through an empty asm: keeps the SGPR-base + 32-bit-offset addressing mode, see hand_off in lds_estep_twoend.hpp)
    static_for<0, J1>([&](auto j) {
      asm volatile("" : "+v"(poff[j]));
      o.Pi[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lrec) + poff[j]);
    });
    lrec -= nextrec > 0 ? WS : 0;
    nextrec -= nextrec > 0 ? 1 : 0;
  };

  // one smoother step.  KIND: 0 generic, 1 first (meeting record: G = 0), 2 second (weight of the repeated pair)
  auto core = [&](auto kind, int s, Ops& cur) {
    constexpr int KIND = decltype(kind)::value;
    // G~ rows of this DPP row: X[i][c] = sum_k P^-1[i][k] J12'[k][c] (lanes < N), c_i (lane N); row N = e_N
    double Gc[J1], H[N + 1];
    static_for<0, J1>([&](auto j) { Gc[j] = (j == NS) ? __builtin_fma(EN, cur.Pi[j], CN) : EN * cur.Pi[j]; });
    if (KIND != 1) {
      dpp_fence(cur.Pi);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k, true>(Gc[j], cur.Pi[j], NJ12c[k]); });
      });
    }
    // transposed, replicated copy through LDS: H[k][lane c] = G~[c][k]
    __builtin_amdgcn_wave_barrier();
    static_for<0, J1>([&](auto j) { tabw[(4 * j + r) * RSL + c] = Gc[j]; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    static_for<0, (N + 2) / 2>([&](auto q) {
      const double2 v = reinterpret_cast<const double2*>(tabw + c * RSL)[q];
      H[2 * q] = v.x;
      if constexpr (2 * q + 1 <= N) H[2 * q + 1] = v.y;
    });
    __builtin_amdgcn_wave_barrier();

    // W~[i] = S~[i] G~'  for my rows
    double W[J1];
    static_for<0, J1>([&](auto j) { W[j] = 0.0; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(W[j], S[j], H[k]); });
    });
    dpp_fence(W);
    if constexpr (KIND == 0) static_for<0, J>([&](auto j) { sumW[j] += W[j]; });
    double WR[4 * J1];
    static_for<0, J1>([&](auto j) { quad_gather(W[j], WR[4 * j], WR[4 * j + 1], WR[4 * j + 2], WR[4 * j + 3]); });
    // S~_t[i] = P^-1[i] + G~[i] W~
    double Sn[J1];
    static_for<0, J1>([&](auto j) { Sn[j] = __builtin_fma(-EN, cur.Pi[j], cur.Pi[j]); });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(Sn[j], Gc[j], WR[k]); });
    });

    if constexpr (KIND == 1) {
      static_for<0, J>([&](auto j) { Stop[j] = Sn[j]; });
    } else if constexpr (KIND == 2) {
      static_for<0, J>([&](auto j) {
        sumS[j] = wsp * Sn[j];
        sumW[j] = wsp * W[j];
        Stop[j] = __builtin_fma(wsp, Stop[j], __builtin_fma(-wsp, Sn[j], Sn[j]));
      });
    } else {
      static_for<0, J>([&](auto j) { sumS[j] += Sn[j]; });
    }

    // node statistics: diag E[x_t x_t'] (lane i of DPP row i & 3), E[x_t] = row N
    double dg = 0.0;
    static_for<0, J>([&](auto j) { dg = __builtin_fma(ED[j], Sn[j], dg); });
    *pdg = dg;
    *pex = Sn[NS];
    if constexpr (KIND == 1) node_ptrs(e - 1);
    else { pdg += dlane ? nstride : 0; pex += xlane ? nstride : 0; }
    static_for<0, J1>([&](auto j) { S[j] = Sn[j]; });
  };
  // a step on a record of the HBM ring
  auto step = [&](auto kind, int s, Ops& stage) {
    // (an explicit register copy: the stage's live range ends HERE, so the refill below can land in the same
    //  registers and the loop-carried value needs no copy at the back edge -- a compiler-made copy of a freshly
    //  loaded register there costs a wait for the youngest load, i.e. the whole prefetch distance)
    Ops cur;
    static_for<0, J1>([&](auto j) { cur.Pi[j] = reg_copy(stage.Pi[j]); });
    load_ops(stage);                         // refill the stage with the record four steps further down
    core(kind, s, cur);
  };
  // records kept in LDS: read one step ahead (slot of local step s = s - s0, clamped: the prefetch past the window
  // is never used)
  unsigned lpoff[J1];
  static_for<0, J1>([&](auto j) { lpoff[j] = poff[j] - 8u * choff + 8u * (unsigned)(dir * keep * WS); });
  auto lds_ops = [&](Ops& o, int s) {
    const int slot = s > s0 ? s - s0 : 0;
    const char* base = reinterpret_cast<const char*>(lrecs + (long)slot * WS);
    static_for<0, J1>([&](auto j) { o.Pi[j] = *reinterpret_cast<const double*>(base + lpoff[j]); });
  };

  constexpr std::integral_constant<int, 0> GEN{};
  if (keep > 0) {
    Ops R0, R1, R2, R3, LA, LB;
    if (s0 > 0) { load_ops(R0); load_ops(R1); load_ops(R2); load_ops(R3); }   // HBM records s0-1 .. s0-4: long on their way
    lds_ops(LA, e);
    lds_ops(LB, e - 1);
    core(std::integral_constant<int, 1>{}, e, LA);
    lds_ops(LA, e - 2);
    core(std::integral_constant<int, 2>{}, e - 1, LB);
    int s = e - 2;
    for (; s - 1 >= s0; s -= 2) {      // LA holds record s
      lds_ops(LB, s - 1);
      core(GEN, s, LA);
      lds_ops(LA, s - 2);
      core(GEN, s - 1, LB);
    }
    if (s >= s0) { core(GEN, s, LA); --s; }
    for (; s >= 3; s -= 4) {           // s = s0 - 1: the HBM ring (R0 holds record s)
      step(GEN, s, R0);
      step(GEN, s - 1, R1);
      step(GEN, s - 2, R2);
      step(GEN, s - 3, R3);
    }
    if (s >= 0) step(GEN, s, R0);
    if (s >= 1) step(GEN, s - 1, R1);
    if (s >= 2) step(GEN, s - 2, R2);
  } else {
    Ops R0, R1, R2, R3;
    load_ops(R0); load_ops(R1); load_ops(R2); load_ops(R3);       // records e, e-1, e-2, e-3
    step(std::integral_constant<int, 1>{}, e, R0);
    step(std::integral_constant<int, 2>{}, e - 1, R1);
    step(GEN, e - 2, R2);                                         // (e >= 2: T >= TE_MIN_T)
    int s = e - 3;
    if (s >= 0) { step(GEN, s, R3); --s; }
    for (; s >= 3; s -= 4) {           // four steps per trip, no branch inside (hipcc's wait counts stay exact)
      step(GEN, s, R0);
      step(GEN, s - 1, R1);
      step(GEN, s - 2, R2);
      step(GEN, s - 3, R3);
    }
    if (s >= 0) step(GEN, s, R0);
    if (s >= 1) step(GEN, s - 1, R1);
    if (s >= 2) step(GEN, s - 2, R2);
  }

  // ---- global statistics: chain B's sums travel to chain A's wavefront through LDS ---------------------------
  // S = S~ at the chain's end node (x_0 for A, x_{T-1} for B).  sumS = sum S~(s) over the chain's counted steps,
  // sumP = sum S~(s+1) = (sumS - S~(0)) + Stop, sumW = sum W~(s).
  // A: first block += sumS, third += sumP, cross += sumW';  B: first += sumP, third += sumS, cross += sumW.
  double sumP[J];
  static_for<0, J>([&](auto j) { sumP[j] = (sumS[j] - S[j]) + Stop[j]; });
  double* xS = xch;                        // [J][64]
  double* xP = xch + J * 64;               // [J][64]
  double* xW = xch + 2 * J * 64;           // [16][16]: W_B[i][c]
  if (dir) {
    static_for<0, J>([&](auto j) {
      xS[j * 64 + lane] = sumS[j];
      xP[j * 64 + lane] = sumP[j];
      xW[(4 * j + r) * 16 + c] = sumW[j];
    });
  }
  __syncthreads();
  if (!dir) {
    double* ep = a.E_pair + (long)b * 3 * N * N;
    double* ei = a.E_init + (long)b * (N * N + N);
    static_for<0, J>([&](auto j) {
      const int i = 4 * j + r;
      const int ii = i < N ? i : 0;
      const double oS = xS[j * 64 + lane], oP = xP[j * 64 + lane];
      const double oWt = xW[cc * 16 + ii];            // W_B[c][i]: A's W_A[i][c] lands at cross[c][i]
      if (i < N && col) {
        ep[i * N + c] = sumS[j] + oP;
        ep[N * N + c * N + i] = sumW[j] + oWt;
        ep[2 * N * N + i * N + c] = sumP[j] + oS;
        ei[i * N + c] = S[j];
      }
    });
    if (col && own_N) ei[N * N + c] = S[NS];
  }
}

}  // namespace svae
