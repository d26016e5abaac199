// lds_estep_twoend_s4.hpp -- smoother phase of the two-ended E-step with ONE CHAIN PER WAVEFRONT (small batches).
//
// The two-ended kernel (lds_estep_twoend.hpp) runs both chains of a sequence in one instruction stream, two DPP
// rows per chain.  Up to 512 sequences that leaves half of the chip's 1024 SIMDs idle.  After the meeting node the
// two chains' smoothers are independent, so the workgroup carries a second wavefront that sleeps at a barrier
// through the elimination phase and then takes chain B's smoother while the first keeps chain A's: each chain now
// owns all FOUR DPP rows of its wavefront, every product stage is split by output row over four rows instead of
// two (row i of a tile lives in DPP row i & 3, slot i >> 2): 96 instead of 182 DPP multiply-adds per step; the
// all-gather of the one tile that sits on the recursion's serial chain goes through LDS, its round trip covered by the
// preparation of the next step (round 4; as v_permlane16/32_swap shuffles it was 36 of 194 instructions).  Same
// recursion, same accumulation order per
// output element as the two-row smoother (moment form on homogeneous coordinates:
// cython_lds_inference.pyx:149-210 is what it replaces), same lean hand-off records.
// Homogeneous parameters, lean records, statistics summed over time (the headline configuration).
#pragma once
#include "lds_estep_kernel.hpp"

namespace svae {

// rows 2j (DPP row 0 of the pair) and 2j+1 (DPP row 1) of a slot register -> two registers replicated
// over the pair.  v_permlane16_swap exchanges the odd rows of its first operand with the even rows of
// its second; the compiler pads the VALU -> permlane hazard of the copies it makes (the inputs must
// already be fenced from asm producers, see dpp_fence).
__device__ __forceinline__ void pair_split(double x, double& even_row, double& odd_row) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  even_row = __hiloint2double(rh[0], rl[0]);
  odd_row = __hiloint2double(rh[1], rl[1]);
}

// x: one value per DPP row  ->  the four rows' values, each replicated over the wavefront (register shuffles: used by the
// reverse-mode sweeps, lds_vjp_kernel.hpp; the smoother below gathers through LDS instead).
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

// an empty asm that reads x: the compiler waits for its pending load HERE.  Applied to the register a batch of LDS reads
// fills LAST (LDS returns in order: that wait covers the whole batch) it replaces the one wait per 16-byte read that
// hipcc otherwise places in front of each read's first use.
__device__ __forceinline__ void touch(const double& x) { asm volatile("" : : "v"(x)); }

__device__ __forceinline__ double reg_copy(double x) {       // a copy the compiler cannot fold away
  double y;
  asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));
  return y;
}

// dir = the chain (0: A, forward in time; 1: B, reversed) = the wavefront's index in the workgroup.
// tabw: 16 x RSL doubles of LDS of this wavefront, zero-initialised;  wtw: 16 x RSL more (the gather tile of the
// software-pipelined steps, any contents);  xch: exchange buffer of the workgroup.
// Both wavefronts of the workgroup call this function; it contains ONE __syncthreads().
// lrecs / keep: the last `keep` records of each chain (local steps e+1-keep .. e, the ones read first) live in LDS
// ([chain][slot][WS] doubles, written by the elimination phase) instead of the HBM workspace; keep = 0 or >= 3.
template <int N>
__device__ __forceinline__ void te_smooth4(const LdsArgs& a, const int b, const int dir, const int lane,
                                           double* tabw, double* wtw, double* xch, const double* lrecs, const int keep) {
  constexpr int ZP = te_page_doubles(N), WS = te_lean_step_doubles(N);
  constexpr int TRI = N * (N + 1) / 2, LZERO = TRI + N;
  constexpr int J = (N + 3) / 4;          // slots holding rows 0..N-1 (row i = 4j + r)
  constexpr int J1 = (N + 4) / 4;         // slots holding rows 0..N
  constexpr int RSL = (N + 3) & ~1;       // LDS row stride of the transposition tile
  constexpr int RSW = 4 * ((N + 4) / 4);  // row stride of the gather tile: every slot row 4j + r, j < J1, has its own entry
  constexpr int NS = N >> 2, NR = N & 3;  // row N of the homogeneous tile: slot NS of DPP row NR
  const int c = lane & 15, r = lane >> 4;
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int cw = c <= N ? c : N + 1;      // column of this lane's transposition-tile stores
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

  // records kept in LDS: read one step ahead (slot of local step s = s - s0, clamped: the prefetch past the window
  // is never used)
  unsigned lpoff[J1];
  static_for<0, J1>([&](auto j) { lpoff[j] = poff[j] - 8u * choff + 8u * (unsigned)(dir * keep * WS); });
  auto lds_ops = [&](Ops& o, int s) {
    const int slot = s > s0 ? s - s0 : 0;
    const char* base = reinterpret_cast<const char*>(lrecs + (long)slot * WS);
    static_for<0, J1>([&](auto j) { o.Pi[j] = *reinterpret_cast<const double*>(base + lpoff[j]); });
  };

  // ---- SOFTWARE-PIPELINED steps -------------------------------------------------------------------------------------
  // The products of a step split its rows over the four DPP rows, so W~ has to be all-gathered between them -- on the
  // serial chain S~ -> W~ -> S~.  As register shuffles (quad_gather above) that is 36 of the step's 194 instructions.
  // Here the gather goes through LDS instead (3 stores, 6 16-byte loads: lane c stores W~[i][c] at [c][i] and reads
  // back row [c][0..N] = column c of every row) and its round trip is covered by work that does not depend on the
  // recursion: the PREPARATION of the next step -- G~ of record s-1 rebuilt from its lean record (one split product)
  // and transposed through the other LDS tile -- issued between the stores and the first use of the gathered tile.
  struct Prep { double Gc[J1], H[N + 1], PiZ[J1]; };
  // operands of the step on record `pi`: G~ rows (slot layout), its transposed replicated copy H, P^-1 rows
  auto prep_products = [&](auto first, const Ops& pi, Prep& p) {          // VALU part + the tile stores
    static_for<0, J1>([&](auto j) { p.Gc[j] = (j == NS) ? __builtin_fma(EN, pi.Pi[j], CN) : EN * pi.Pi[j]; });
    if constexpr (!decltype(first)::value) {
      double piv[J1];
      static_for<0, J1>([&](auto j) { piv[j] = pi.Pi[j]; });
      dpp_fence(piv);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k, true>(p.Gc[j], piv[j], NJ12c[k]); });
      });
    }
    static_for<0, J1>([&](auto j) { p.PiZ[j] = __builtin_fma(-EN, pi.Pi[j], pi.Pi[j]); });
    __builtin_amdgcn_wave_barrier();
    // (lanes > N: the row's padding column, see lds_estep_twoend.hpp -- a store at column c would land in the next rows)
    static_for<0, J1>([&](auto j) { tabw[(4 * j + r) * RSL + cw] = p.Gc[j]; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto prep_read = [&](Prep& p) {
    static_for<0, (N + 2) / 2>([&](auto q) {
      const double2 v = reinterpret_cast<const double2*>(tabw + c * RSL)[q];
      p.H[2 * q] = v.x;
      if constexpr (2 * q + 1 <= N) p.H[2 * q + 1] = v.y;
    });
    __builtin_amdgcn_wave_barrier();
  };
  // one step on the prepared operands `p` (record s); on return p holds the operands of record s-1 (from `nxt`)
  auto pstep = [&](auto kind, int s, Prep& p, const Ops& nxt) {
    constexpr int KIND = decltype(kind)::value;
    // W~[i] = S~[i] G~'  for my rows
    double W[J1];
    static_for<0, J1>([&](auto j) { W[j] = 0.0; });
    // (the LAST register the operand's LDS reads fill is "used" here: one wait for the batch -- issued a whole product
    //  ago -- instead of one per 16-byte read in front of its first multiply-add: 5 issue slots per product)
    touch(p.H[N]);
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(W[j], S[j], p.H[k]); });
    });
    // all-gather through LDS: store [c][i], read back [c][0..N]
    __builtin_amdgcn_wave_barrier();
    static_for<0, J1>([&](auto j) { wtw[c * RSW + 4 * j + r] = W[j]; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double WR[N + 2];
    static_for<0, (N + 2) / 2>([&](auto q) {
      const double2 v = reinterpret_cast<const double2*>(wtw + c * RSW)[q];
      WR[2 * q] = v.x;
      WR[2 * q + 1] = v.y;
    });
    __builtin_amdgcn_wave_barrier();
    if constexpr (KIND == 0) static_for<0, J>([&](auto j) { sumW[j] += W[j]; });
    double Wk[J];
    static_for<0, J>([&](auto j) { Wk[j] = W[j]; });
    // this step's G~ rows / P^-1 rows move out of `p`, which takes the next step's operands while the gather is in flight
    double Gc[J1], Sn[J1];
    static_for<0, J1>([&](auto j) { Gc[j] = p.Gc[j]; Sn[j] = p.PiZ[j]; });
    prep_products(std::false_type{}, nxt, p);
    prep_read(p);
    // S~_t[i] = P^-1[i] + G~[i] W~
    touch(WR[2 * ((N + 2) / 2) - 1]);
    dpp_fence(Gc);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(Sn[j], Gc[j], WR[k]); });
    });
    if constexpr (KIND == 1) {
      static_for<0, J>([&](auto j) { Stop[j] = Sn[j]; });
    } else if constexpr (KIND == 2) {
      static_for<0, J>([&](auto j) {
        sumS[j] = wsp * Sn[j];
        sumW[j] = wsp * Wk[j];
        Stop[j] = __builtin_fma(wsp, Stop[j], __builtin_fma(-wsp, Sn[j], Sn[j]));
      });
    } else {
      static_for<0, J>([&](auto j) { sumS[j] += Sn[j]; });
    }
    double dg = 0.0;
    static_for<0, J>([&](auto j) { dg = __builtin_fma(ED[j], Sn[j], dg); });
    *pdg = dg;
    *pex = Sn[NS];
    if constexpr (KIND == 1) node_ptrs(e - 1);
    else { pdg += dlane ? nstride : 0; pex += xlane ? nstride : 0; }
    static_for<0, J1>([&](auto j) { S[j] = Sn[j]; });
  };
  // a step whose NEXT record sits in `stage` of the HBM ring (copied out, then the stage is refilled four records on)
  auto rstep = [&](auto kind, int s, Prep& p, Ops& stage) {
    Ops nxt;
    static_for<0, J1>([&](auto j) { nxt.Pi[j] = reg_copy(stage.Pi[j]); });
    load_ops(stage);
    pstep(kind, s, p, nxt);
  };
  constexpr std::integral_constant<int, 0> GEN{};
  constexpr std::integral_constant<int, 1> FIRST{};
  constexpr std::integral_constant<int, 2> SECOND{};
  Ops R0, R1, R2, R3;
  Prep P;
  int s;
  if (keep > 0) {
    // records e .. s0 in LDS (keep >= 3: the first two steps find their next record there), s0-1 .. 0 in the HBM ring
    if (s0 > 0) { load_ops(R0); load_ops(R1); load_ops(R2); load_ops(R3); }   // HBM records s0-1 .. s0-4: long on their way
    Ops nx;
    lds_ops(nx, e);
    prep_products(std::true_type{}, nx, P);                        // the meeting record: G = 0
    prep_read(P);
    lds_ops(nx, e - 1);
    pstep(FIRST, e, P, nx);
    lds_ops(nx, e - 2);
    pstep(SECOND, e - 1, P, nx);
    for (s = e - 2; s - 1 >= s0; --s) {
      lds_ops(nx, s - 1);
      pstep(GEN, s, P, nx);
    }
    // s == s0: the next records come from the ring (s0 == 0: none is needed, any valid record will do)
    if (s0 == 0) {
      lds_ops(nx, 0);
      pstep(GEN, 0, P, nx);
      s = -1;
    }
  } else {
    load_ops(R0); load_ops(R1); load_ops(R2); load_ops(R3);       // records e, e-1, e-2, e-3
    {
      Ops first;
      static_for<0, J1>([&](auto j) { first.Pi[j] = reg_copy(R0.Pi[j]); });
      load_ops(R0);                                                // record e-4
      prep_products(std::true_type{}, first, P);                   // the meeting record: G = 0
      prep_read(P);
    }
    rstep(FIRST, e, P, R1);                                        // next: record e-1
    rstep(SECOND, e - 1, P, R2);                                   // (e >= 2: T >= TE_MIN_T)
    s = e - 2;
    if (s >= 0) { rstep(GEN, s, P, R3); --s; }                     // next: e-3 (s = 0: the clamped ring re-reads record 0, unused)
  }
  for (; s >= 3; s -= 4) {             // four steps per trip, no branch inside (hipcc's wait counts stay exact)
    rstep(GEN, s, P, R0);
    rstep(GEN, s - 1, P, R1);
    rstep(GEN, s - 2, P, R2);
    rstep(GEN, s - 3, P, R3);
  }
  if (s >= 0) rstep(GEN, s, P, R0);
  if (s >= 1) rstep(GEN, s - 1, P, R1);
  if (s >= 2) rstep(GEN, s - 2, P, R2);

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
