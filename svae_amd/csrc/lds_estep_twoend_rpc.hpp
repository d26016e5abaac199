// lds_estep_twoend_rpc.hpp -- two-ended LDS E-step, ROW-PER-CHAIN layout: two sequences per wavefront (throughput form).
//
// Same algorithm, same results and the same lean hand-off records as lds_estep_twoend.hpp (block elimination from both
// ends of the chain, meeting in the middle, moment-form smoother back to the ends: what replaces
// natural_filter_forward_general / natural_smoother_general / _compute_stats, cython_lds_inference.pyx:28-90, 149-210),
// re-mapped for batches that fill the chip.  lds_estep_twoend.hpp gives one sequence a whole wavefront -- each chain two
// DPP rows, the matrix replicated in both, product stages split between them -- which minimises the LENGTH of one
// sequence's instruction stream (the right thing while SIMDs idle: B <= 512) but issues every instruction for one
// sequence only.  Here every DPP row is one chain on its own:
//     DPP row 0: sequence 2w, chain A      row 1: sequence 2w, chain B      rows 2, 3: sequence 2w + 1
//   * Gauss-Jordan on TWO registers per matrix row: A[i] = [P row (lanes < N) | .. | h_i (lane 15)], B[i] = the
//     right-hand-side columns J12[:, c] (lanes < N).  A row update is two DPP multiply-adds (both broadcast the
//     multiplier, lane k of A[i]) -- per SEQUENCE the same count as the one-register form, but the pivot bookkeeping
//     (reciprocal chain, scaled pivot row, log-determinant) is issued once for two sequences;
//   * no replication: the Schur complement  An[i] = C[i] + sum_k bcast_i(B[k]) Bt[k]  lands row i in register i, which is
//     the next step's input as it stands (the one-register form re-replicates five slot registers with
//     v_permlane16_swap per step), and the smoother's products run on whole (N+1) x (N+1) tiles: no slot split, no
//     all-gather;
//   * diag E[x x'] is lane-local: (G~ W~)[c][c] = sum_k G~'[k][c] W~[k][c], both operands already in registers.
// Instruction stream per wavefront step, N = 10: elimination 542, smoother 455 -- for TWO sequences (one-register form:
// 406 + 295 for one).  Two wavefronts per SIMD (227 registers); the slowest wavefront of a full launch issues one
// instruction per 4.2 cycles of its SIMD: the kernel is at the issue limit of its instruction count.
// Homogeneous pair parameters, lean records, statistics summed over time: the headline configuration from
// TE_RPC_MIN_B sequences.
#pragma once
#include "lds_estep_kernel.hpp"
#include "gj1r_gen.hpp"

namespace svae {

#ifdef SVAE_PHASE_TIMING
#define RPC_TICK(i) { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tlast_; tlast_ = now_; }
#else
#define RPC_TICK(i)
#endif

__device__ __forceinline__ double asm_neg(double x) {        // -x as ONE VALU instruction (the compiler's: xor + mov)
  double r;
  asm volatile("v_add_f64 %0, -%1, 0" : "=v"(r) : "v"(x));
  return r;
}

template <int N>
__global__ __launch_bounds__(64) void lds_estep_twoend_rpc_kernel(const LdsArgs a) {
  static_assert(N >= 1 && N <= TE_MAX_N, "latent dimension");
  constexpr int ZP = te_page_doubles(N), WS = te_lean_step_doubles(N);
  constexpr int TRI = N * (N + 1) / 2;    // lean record: [lower triangle of P^-1 | c (N) | 0.0 | trash | pad]
  constexpr int LZERO = TRI + N, LTRASH = TRI + N + 1;
  constexpr int HL = N;                   // lane of the h column (= the homogeneous coordinate's column in the smoother)
  constexpr int RSL = (N + 3) & ~1;       // LDS row stride of the transposition tile (even, >= N + 1)
  __shared__ double tab[4 * 16 * RSL];    // [DPP row][tile row 0..15][RSL]: G~ rows for the transposed read

  const int lane = threadIdx.x & 63;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int dir = g & 1;                  // 0: chain A (forward in time), 1: chain B (reversed)
  const int sq = g >> 1;                  // which of the wavefront's two sequences
  const int b0 = 2 * blockIdx.x;          // (uniform) first sequence of the wavefront
  // odd batch: the last wavefront's second row pair repeats sequence B-1 (same values to the same addresses)
  const int bo = (b0 + sq < a.B) ? sq : 0;
  const int b = b0 + bo;
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

  // ---- pair parameters in the chain's own orientation: A (J11, J12, J22); B (J22, J12', J11) ------------------------
  const double* q11 = dir ? a.J22 : a.J11;
  const double* q22 = dir ? a.J11 : a.J22;
  const int si = dir ? 1 : N, sc = dir ? N : 1;       // J12'[i][x] = J12[i * si + x * sc]
  //   NJ12c[k]: lanes < N: nat J12'[k][c] (= -info-form J12'[k][c]); other lanes 0
  //   Cc[i]:    lanes < N: info-form J22' + J11' (the pair behind + the pair ahead); other lanes 0
  double NJ12c[N], Cc[N];
  static_for<0, N>([&](auto i) {
    const double rc = a.J12[i * si + cc * sc], r22 = q22[i * N + cc], r11 = q11[i * N + cc];
    NJ12c[i] = col ? rc : 0.0;
    Cc[i] = col ? -2.0 * (r22 + r11) : 0.0;
  });

  // An: lanes < N = pivot block of the next node without its node potential (incoming message + J11' of the pair
  // ahead), lane 15 = incoming potential vector, other lanes zero
  double An[N];
  static_for<0, N>([&](auto i) {
    const double ij = a.init_J[i * N + cc], ih = a.init_h[i], j11 = q11[i * N + cc];
    An[i] = col ? -2.0 * ((dir ? 0.0 : ij) + j11) : ((c == HL && !dir) ? ih : 0.0);
  });

  // node potentials of local step s: global node t = s (A) / T-1-s (B); lanes >= N read element 0
  const double* nJb = a.node_J + ((long)b * T) * N + cc;
  const double* nhb = a.node_h + ((long)b * T) * N + cc;
  auto node_off = [&](int s) -> long { return (long)(dir ? T - 1 - s : s) * N; };

  // chain workspace (layout of lds_estep_twoend.hpp): constant page [e_N (N+2) | zeros (N+2) | trash (2)], records
  double* wsb = a.ws + (long)b0 * te_seq_doubles(N, T);            // uniform: the wavefront's first sequence
  const unsigned choff = (unsigned)(bo * te_seq_doubles(N, T) + dir * te_chain_doubles(N, T)) + ZP;   // per lane: its chain's records
  double* zpage = wsb + (long)bo * te_seq_doubles(N, T) + (long)dir * te_chain_doubles(N, T);
  double* rec0 = zpage + ZP;
  double* trash = zpage + 2 * (N + 2);
  if (c < N + 2) { zpage[c] = EN; zpage[N + 2 + c] = 0.0; }
  // hand-off store of register i: ONE unconditional instruction, lane c <= i -> tri(i) + c, lane 15 -> TRI + i, the
  // others -> the record's trash entry (a conditional store makes hipcc wait for the previous step's stores)
  unsigned loff[N];
  static_for<0, N>([&](auto i) {
    loff[i] = 8u * (choff + ((c <= i) ? i * (i + 1) / 2 + c : ((c == HL) ? TRI + i : LTRASH)));   // bytes
  });
  for (int q = lane; q < 4 * 16 * RSL; q += 64) tab[q] = 0.0;      // rows of the transposition tiles never written
  for (int r = c; r <= e; r += 16) rec0[(long)r * WS + LZERO] = 0.0;   // the records' zero entry

  double qacc = 0.0;        // lane 15: sum_t h' P^-1 h
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
  hand_off(e, An, 1.0);     // (dummy: makes the loop-head wait vmcnt(N), see lds_estep_twoend.hpp)

#ifdef SVAE_PHASE_TIMING
  long long tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long wall0_ = __builtin_amdgcn_s_memrealtime();     // 100 MHz, the same clock on every CU
  long long tlast_ = __builtin_readcyclecounter();
#endif
  // ---- elimination (filter) phase ----------------------------------------------------------------------------------
  // Two wavefronts share a SIMD at full batches and the arbiter favours the older one: it would finish ~40 % earlier and
  // leave the younger alone on the SIMD (single-wavefront issue rate) for the last third of the kernel.  The priority
  // therefore alternates in TIME (bit 9 of the 100 MHz real-time counter: every 5 us), in opposite phase for the older
  // and the younger half of the grid: the two wavefronts of a SIMD see the same clock, so exactly one of them holds
  // priority at any moment and they stay abreast.
#ifndef SVAE_RPC_SETPRIO
#define SVAE_RPC_SETPRIO 1
#endif
  const int young = (2 * blockIdx.x >= gridDim.x) ? 1 : 0;
  // Lanes N+1 .. 15 of every DPP row carry nothing: the two loops run with them switched off (EXEC), which is
  // 5 of 16 lanes of every fp64 operation not toggling -- the kernel is power-limited at full batches (the shader
  // clock sags to ~1.7 GHz with all 64 lanes live).
#ifndef SVAE_RPC_LANEMASK
#define SVAE_RPC_LANEMASK 1
#endif
  const bool live = !SVAE_RPC_LANEMASK || c <= N;
  double qacc_m = 0.0, vfull_m = col ? 0.0 : 1.0;
  if (live) {
  for (int s = 0; s < e; ++s) {
#if SVAE_RPC_SETPRIO
    if (((int)(__builtin_amdgcn_s_memrealtime() >> 9) ^ young) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
    if (s == jx) take_partner();
    const double JoX = col ? -2.0 * Jo_n : 1.0;
    double ho = ho_n;
    Jo_n = nJb[node_off(s + 1)];           // s + 1 <= e: the meeting node's potentials included
    ho_n = nhb[node_off(s + 1)];

    // condition on the node potential; the right-hand sides (info-form J12' = -nat) in the second register set
    double MA[N], MB[N], Bt[N];
    static_for<0, N>([&](auto i) { MA[i] = __builtin_fma(JoX, E[i], An[i]); });
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(MA[i], ho, EH); });      // lane 15: h_filt = h_pred + h_node
    // B operand of the Schur stage: lanes < N: -J12'[k][c] (info form) = nat; lane 15: -h_filt,k
    static_for<0, N>([&](auto k) {
      Bt[k] = __builtin_fma(-EH, MA[k], NJ12c[k]);
      MB[k] = asm_neg(NJ12c[k]);
    });
    dpp_fence(MA);
    RPC_TICK(0)

    double vfull = col ? 0.0 : 1.0;
    gauss_jordan_2r_asm<N>(MA, MB, E, qacc, vfull);
    RPC_TICK(1)

    // next pivot block: An[i] = Cc[i] + sum_k X[k][i] * Bt[k]   (X[k][i] = lane i of MB[k]); row i in register i
    static_for<0, N>([&](auto i) { An[i] = Cc[i]; });
    asm volatile("s_nop 1");
    static_for<0, N>([&](auto k) {
      static_for<0, N>([&](auto i) { mac_bc<i>(An[i], MB[k], Bt[k]); });
    });
    RPC_TICK(2)

    vworst = fmax(vworst, vfull);
    ldM *= vfull;
    if ((s & 3) == 3) {
      ldE += __builtin_amdgcn_frexp_exp(ldM);
      ldM = __builtin_amdgcn_frexp_mant(ldM);
    }
    hand_off(s, MA, vfull);
    RPC_TICK(3)
  }
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
  }   // live

  // ---- log-normaliser ------------------------------------------------------------------------------------------------
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
    const double total = row_sum16(z) + chain_total + meet_total + a.init_logZ[0] + (double)(T - 1) * a.logZ_pair[0];
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

  RPC_TICK(4)
  // ---- smoother phase: moment form on homogeneous coordinates, local steps e, e-1, .., 0 ------------------------------
  // S~ = (N+1) x (N+1) tile, row i in register i (lane = column); starts from e_N e_N' so that the generic step at the
  // meeting record (G = 0, c = mu) yields [[Sigma + mu mu', mu], [mu', 1]].
  double S[N + 1];
  static_for<0, N + 1>([&](auto i) { S[i] = (i == N) ? EN : 0.0; });
  dpp_fence(S);
  // S~ of the chain's first counted step ("Stop", needed once, at the end) waits in this sequence's E_pair output
  // block (chain A: first n x n block, chain B: second; the final statistics overwrite them): 20 registers less
  // (running sums in registers: accumulating them in LDS with ds_add_f64 frees 40 registers but saturates the LDS
  //  pipe with eight wavefronts per CU -- measured 6.9 instead of 4.4 cycles per instruction in this phase)
  double sumS[N], sumW[N];
  static_for<0, N>([&](auto i) { sumS[i] = 0.0; sumW[i] = 0.0; });
  const bool skip2nd = dir && !oddT;      // chain B, even T: its first smoother step repeats pair e-1, which chain A counts
  const double wsp = skip2nd ? 0.0 : 1.0;
  const bool own_e = oddT && !dir;        // who reports the meeting node
  // S~ of the chain's first counted step ("Stop", needed once, at the end) waits in this sequence's E_pair output
  // block (chain A: first n x n block, chain B: second; the final statistics overwrite all three): 20 registers
  // less.  Stores are unconditional with a per-lane address: lanes with nothing to keep write into the third block.
  double* const epb = a.E_pair + (long)b * 3 * N * N;
  double* const stop_dummy = epb + 2 * N * N + (c % N);
  double* stop_p = col ? epb + dir * N * N + c : stop_dummy;
  double* stop2_p = (col && skip2nd) ? stop_p : stop_dummy;   // second step: replaces it where the first pair is not counted

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
    poff[i] = 8u * (choff + (col ? hi * (hi + 1) / 2 + lo : ((c == N) ? TRI + i : LZERO)));   // bytes
  });
  pdoff = 8u * (choff + (col ? c * (c + 1) / 2 + c : LZERO));
  // Records are fetched one step ahead into the stage the previous step used (a three-stage ring, two steps ahead,
  // measured no gain: the phase is issue-bound, not latency-bound).
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
  double* tb = tab + g * 16 * RSL;
  // lanes > N (zeros) write the row's padding column N + 1 (RSL >= N + 2): a full-lane store at column c would run
  // into the next row, and row N (= e_N: constant, written once here) is never rewritten
  double* tbw = tb + (c <= N ? c : N + 1);
  if (c <= N) tb[N * RSL + c] = EN;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // one smoother step.  KIND: 0 generic, 1 first (meeting record: G = 0), 2 second (weight of the repeated pair)
  auto step = [&](auto kind, Ops& cur, Ops& fill) {
    constexpr int KIND = decltype(kind)::value;
#if SVAE_RPC_SETPRIO
    if (((int)(__builtin_amdgcn_s_memrealtime() >> 9) ^ young) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
    load_ops(fill);                          // the next record, into the stage the previous step used
    // G~ rows: X[i][c] = sum_k P^-1[i][k] J12'[k][c] (lanes < N), c_i (lane N); row N = e_N (constant, in the tile)
    double Gc[N], H[N + 1];
    static_for<0, N>([&](auto i) { Gc[i] = EN * cur.Pi[i]; });
    if (KIND != 1) {
      dpp_fence(cur.Pi);
      static_for<0, N>([&](auto k) {
        static_for<0, N>([&](auto i) { mac_bc<k, true>(Gc[i], cur.Pi[i], NJ12c[k]); });
      });
    }
    RPC_TICK(5)
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

    RPC_TICK(6)
    // W~[i] = S~[i] G~'  :  sum_k -/+ bcast_k(S[i]) H[k]      (rows 0 .. N)
    double W[N + 1];
    static_for<0, N + 1>([&](auto i) { W[i] = 0.0; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, N + 1>([&](auto i) { mac_bc<k, (k < N)>(W[i], S[i], H[k]); });
    });
    dpp_fence(W);
    RPC_TICK(7)
    if constexpr (KIND == 0) static_for<0, N>([&](auto i) { sumW[i] += W[i]; });
    // S~_t[i] = P^-1[i] + G~[i] W~ = Pi + sum_k -/+ bcast_k(Gc[i]) W[k]   (rows < N; row N of G~ = e_N: S~_t[N] = W~[N])
    double Sn[N + 1];
    static_for<0, N>([&](auto i) { Sn[i] = __builtin_fma(-EN, cur.Pi[i], cur.Pi[i]); });
    Sn[N] = W[N];
    dpp_fence(Gc);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, N>([&](auto i) { mac_bc<k, (k < N)>(Sn[i], Gc[i], W[k]); });
    });
    RPC_TICK(8)
    // diag E[x_t x_t'] is lane-local: S~_t[c][c] = P^-1[c][c] + sum_k -/+ G~[c][k] W~[k][c], G~[c][k] = H[k] in lane c
    double dg = cur.Pd;
    static_for<0, N + 1>([&](auto k) {
      if constexpr (k < N) dg = __builtin_fma(-H[k], W[k], dg); else dg = __builtin_fma(H[k], W[k], dg);
    });

    if constexpr (KIND == 1) {
      static_for<0, N>([&](auto i) { stop_p[i * N] = Sn[i]; });
    } else if constexpr (KIND == 2) {
      // (multiplications by 0 / 1, exact; Stop = wsp * Stop + (1 - wsp) * Sn as an address select)
      static_for<0, N>([&](auto i) {
        sumS[i] = wsp * Sn[i];
        sumW[i] = wsp * W[i];
        stop2_p[i * N] = Sn[i];
      });
    } else {
      static_for<0, N>([&](auto i) { sumS[i] += Sn[i]; });
    }

    *pdg = dg;
    *pex = Sn[N];
    if constexpr (KIND == 1) node_ptrs(e - 1);
    else { pdg += col ? nstride : 0; pex += col ? nstride : 0; }
    static_for<0, N + 1>([&](auto i) { S[i] = Sn[i]; });
    RPC_TICK(9)
  };

  if (live) {
    Ops R0, R1;
    constexpr std::integral_constant<int, 0> GEN{};
    load_ops(R0);                                                 // record e
    step(std::integral_constant<int, 1>{}, R0, R1);               // local step e (fetches record e - 1)
    step(std::integral_constant<int, 2>{}, R1, R0);               // e - 1
    int s = e - 2;                                                // (e >= 2: T >= TE_MIN_T)
    for (; s >= 1; s -= 2) {           // two steps per trip: the stages ping-pong, no copies, no branch inside
      step(GEN, R0, R1);
      step(GEN, R1, R0);
    }
    if (s == 0) step(GEN, R0, R1);
  }

#ifdef SVAE_PHASE_TIMING
  if (lane == 0) {
    for (int q = 0; q < 10; ++q) a.E_init[(long)b * (N * N + N) + q] = (double)tm[q];
    a.E_init[(long)b * (N * N + N) + 10] = (double)wall0_;
    a.E_init[(long)b * (N * N + N) + 11] = (double)__builtin_amdgcn_s_memrealtime();
  }
  return;
#endif
  // ---- global statistics -------------------------------------------------------------------------------------------
  // S = S~ at the chain's end node (x_0 for A, x_{T-1} for B).  Sums over the chain's pairs:
  //   sumS = sum S~(s) over its counted steps;  sumP = sum S~(s+1) = (sumS - S~(0)) + Stop;  sumW = sum W~(s)
  // A: first block += sumS, third += sumP, cross += sumW';  B: first += sumP, third += sumS, cross += sumW.
  double sumP[N], Stop[N];
  static_for<0, N>([&](auto i) { Stop[i] = stop_p[i * N]; });
  static_for<0, N>([&](auto i) { sumP[i] = (sumS[i] - S[i]) + Stop[i]; });
  double oS[N], oP[N], oWt[N];
  static_for<0, N>([&](auto i) {
    oS[i] = __shfl_xor(sumS[i], 16);
    oP[i] = __shfl_xor(sumP[i], 16);
  });
  // chain B's cross sum, transposed through its tile: A's lane c of register i holds W_A[i][c], which lands at
  // cross[c][i] and needs W_B[c][i]
  __builtin_amdgcn_wave_barrier();
  if (dir && col) static_for<0, N>([&](auto i) { tb[i * RSL + c] = sumW[i]; });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const double* tbB = tab + (g | 1) * 16 * RSL;
  static_for<0, N>([&](auto i) { oWt[i] = tbB[cc * RSL + i]; });
  if (!dir && col) {
    double* ep = a.E_pair + (long)b * 3 * N * N;
    double* ei = a.E_init + (long)b * (N * N + N);
    static_for<0, N>([&](auto i) {
      ep[i * N + c] = sumS[i] + oP[i];
      ep[N * N + c * N + i] = sumW[i] + oWt[i];
      ep[2 * N * N + i * N + c] = sumP[i] + oS[i];
      ei[i * N + c] = S[i];
    });
    ei[N * N + c] = S[N];
  }
}

template <int N>
static int launch_estep_twoend_rpc(const LdsArgs& a, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    hipLaunchKernelGGL((lds_estep_twoend_rpc_kernel<N>), dim3((a.B + 1) / 2), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
