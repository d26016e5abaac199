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

__device__ __forceinline__ double asm_sub(double a, double b) {        // a - b, kept in program order
  double r;
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
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

#ifdef SVAE_PHASE_TIMING
#define TE_TICK(i) { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tlast_; tlast_ = now_; }
#else
#define TE_TICK(i)
#endif

template <int N, bool INHOMOG>
__global__ __launch_bounds__(64) void lds_estep_twoend_kernel(const LdsArgs a) {
  static_assert(N >= 1 && N <= TE_MAX_N && N + ((N - 1) >> 1) <= 14,
                "the right-hand-side columns must fit lanes N..14 of two DPP rows");
  constexpr int RW = te_row_doubles(N), WS = te_step_doubles(N), ZP = te_page_doubles(N);
  constexpr int J = (N + 1) / 2;          // slots holding rows 0..N-1 (row i = 2j + gl)
  constexpr int J1 = (N + 2) / 2;         // slots holding rows 0..N
  constexpr int HL = 15;                  // lane of the h column
  __shared__ double tab[2 * 16 * 16];     // final transpose of chain B's cross-moment sums

  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int dir = g >> 1;                 // 0: chain A (forward in time), 1: chain B (reversed)
  const int gl = g & 1;                   // DPP row within the chain's pair
  const int b = blockIdx.x;               // one sequence per wavefront
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
  //   EX[i]: lanes < N identity row i; lane of column x: info-form J12'[i][x] = -nat J12'[i][x]
  //   NJ12c[k]: lanes < N: nat J12'[k][c] (= -J12'[k][c]); other lanes 0
  //   Cc[j] (row i = 2j+gl): lanes < N: info-form J22'(pair l) + J11'(pair l+1); other lanes 0
  double EX[N], NJ12c[N], Cc[J];
  auto load_pair = [&](int l) {
    const long o = pair_off(l), o1 = pair_off(l + 1);
    static_for<0, N>([&](auto i) {
      const double rx = q12[o + i * si + xx * sc], rc = q12[o + i * si + cc * sc];
      EX[i] = xok ? -rx : E[i];
      NJ12c[i] = col ? rc : 0.0;
    });
    static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl;
      const int ii = i < N ? i : 0;
      const double r22 = q22[o + ii * N + cc], r11 = q11[o1 + ii * N + cc];
      Cc[j] = (col && i < N) ? -2.0 * (r22 + r11) : 0.0;
    });
  };
  if (!INHOMOG) load_pair(0);

  // ---- elimination (filter) phase -------------------------------------------------------------------
  // An (replicated over the chain's two DPP rows): lanes < N = pivot block of the next node without its
  // node potential (incoming message + J11' of the pair ahead), lane 15 = incoming potential vector,
  // lanes N..14 zero.
  double An[N];
  {
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

  double* zpage = a.ws + ((long)b * 2 + dir) * te_chain_doubles(N, T);   // [e_N | zeros]
  double* rec0 = zpage + ZP;
  if (gl == 0) {
    if (c < N + 2) zpage[c] = EN;
    if (c < N + 2) zpage[N + 2 + c] = 0.0;
  }
  // hand-off store of register i: one instruction, per-lane destination inside row i of the record
  const bool stp = (gl == 0 && col) || xok || (gl == 0 && c == HL);
  const int stoff = col ? c : (c == HL ? 2 * N : N + xx);

  double qacc = 0.0;        // lane 15: sum_t h' P^-1 h
  double ldM = 1.0;         // per lane c < N: running product of -1/p_c (log|P| = -sum log|.|)
  int ldE = 0;
  double vworst = -1.0;     // max over steps of -1/p_c (>= 0 <=> some pivot was not positive)

  double Jo_n = nJb[node_off(0)];
  double ho_n = nhb[node_off(0)];
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

#ifdef SVAE_PHASE_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast_ = __builtin_readcyclecounter();
#endif
  for (int s = 0; s < e; ++s) {
    if (s == jx) take_partner();
    const double JoX = col ? -2.0 * Jo_n : 1.0;
    double ho = ho_n;
    Jo_n = nJb[node_off(s + 1)];           // s + 1 <= e: the meeting node's potentials included
    ho_n = nhb[node_off(s + 1)];
    if (INHOMOG) load_pair(s);

    // condition on the node potential; right-hand sides ride in the upper lanes
    double M[N], Bt[N];
    static_for<0, N>([&](auto i) { M[i] = __builtin_fma(JoX, EX[i], An[i]); });
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(M[i], ho, EH); });      // lane 15: h_filt = h_pred + h_node
    // B operand of the Schur stage: lanes < N: -J12'[k][c]; lane 15: -h_filt,k
    static_for<0, N>([&](auto k) { Bt[k] = __builtin_fma(-EH, M[k], NJ12c[k]); });
    dpp_fence(M);
    TE_TICK(0)

    double vfull = col ? 0.0 : 1.0;
    gauss_jordan_1r<N>(M, E, qacc, vfull);
    TE_TICK(1)

    // next pivot block, slot layout:  AnD[j] = Cc[j] + sum_k X[k][i] * Bt[k]   (row i = 2j + gl;
    // X[k][i] = lane N + j of M[k] in this DPP row)
    double AnD[J];
    static_for<0, J>([&](auto j) { AnD[j] = Cc[j]; });
    asm volatile("s_nop 1");
    static_for<0, N>([&](auto k) {
      static_for<0, J>([&](auto j) { mac_bc<N + j>(AnD[j], M[k], Bt[k]); });
    });
    TE_TICK(2)

    // scale the inverse's columns, hand the record to the smoother phase
    vworst = fmax(vworst, vfull);
    ldM *= vfull;
    if ((s & 3) == 3) {
      ldE += __builtin_amdgcn_frexp_exp(ldM);
      ldM = __builtin_amdgcn_frexp_mant(ldM);
    }
    double* w = rec0 + (long)s * WS + stoff;
    if (stp) static_for<0, N>([&](auto i) { w[i * RW] = M[i] * vfull; });

    dpp_fence(AnD);
    static_for<0, J>([&](auto j) {
      if constexpr (2 * j + 1 < N) pair_split(AnD[j], An[2 * j], An[2 * j + 1]);
      else { double dummy; pair_split(AnD[j], An[2 * j], dummy); }
    });
    TE_TICK(3)
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
    static_for<0, N>([&](auto i) {
      const double r22 = q22[o + i * N + cc], r11 = q11[o1 + i * N + cc];
      const double dbl = col ? -2.0 * (r22 + r11) : 0.0;
      M[i] = __builtin_fma(JoX, E[i], (An[i] + Mp[i]) - dbl);
    });
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(M[i], ho, EH); });
    dpp_fence(M);
    gauss_jordan_1r<N>(M, E, qacc_m, vfull_m);
    double* w = rec0 + (long)e * WS + stoff;
    if (stp) static_for<0, N>([&](auto i) { w[i * RW] = M[i] * vfull_m; });
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
    if (INHOMOG) {
      const double* lz = a.logZ_pair + (a.pair_seq_stride ? (long)b * (T - 1) : 0);
      for (int t = c; t < T - 1; t += 16) z += lz[t];
    }
    double total = row_sum16(z) + chain_total + meet_total + a.init_logZ[0];
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
#ifdef SVAE_PHASE_TIMING
  TE_TICK(4)
#endif

  // ---- smoother phase: moment form on homogeneous coordinates, local steps e, e-1, .., 0 ---------------
  // S~ in slot layout (row i = 2j+gl of the (N+1) x (N+1) tile, lane = column); starts from e_N e_N' so that
  // the generic step at the meeting record (G = 0, c = mu) yields [[Sigma + mu mu', mu], [mu', 1]].
  double ED[J1];                          // ED[j][c] = (c == 2j+gl): picks S[i][i] in slot layout
  static_for<0, J1>([&](auto j) { ED[j] = (c == 2 * j + gl && c < N) ? 1.0 : 0.0; });
  double S[J1];
  static_for<0, J1>([&](auto j) { S[j] = (2 * j + gl == N) ? EN : 0.0; });
  dpp_fence(S);
  double sumS[J], sumW[J], Stop[J];
  static_for<0, J>([&](auto j) { sumS[j] = 0.0; sumW[j] = 0.0; Stop[j] = 0.0; });
  const bool own_N = (gl == (N & 1));     // the DPP row holding row N (E[x_t]) in slot N/2
  // chain B, even T: its first smoother step repeats pair e-1, which chain A counts
  const double wsp = (dir && !oddT) ? 0.0 : 1.0;

  // loads of one step: replicated H[k] = [X | c][c][k] (row c of the record; lane N: e_N; lanes > N: 0)
  // and, in slot layout / column form, Gc[j][c] = [X | c][2j+gl][c] (row N: e_N), Pi[j][c] = P^-1[2j+gl][c]
  const double* hp_ = col ? rec0 + c * RW + N : (c == N ? zpage : zpage + N + 2);
  const long tstride = col ? WS : 0;
  const double* gptr[J1];
  const double* pptr[J1];
  long gstride[J1], pstride[J1];
  static_for<0, J1>([&](auto j) {
    const int i = 2 * j + gl;
    const bool gok = i < N && c <= N, pok = i < N && col;
    gptr[j] = gok ? rec0 + i * RW + N + c : ((i == N && c < N + 2) ? zpage + c : zpage + N + 2);
    pptr[j] = pok ? rec0 + i * RW + c : zpage + N + 2;
    gstride[j] = gok ? WS : 0;
    pstride[j] = pok ? WS : 0;
  });
  hp_ += (long)e * tstride;
  static_for<0, J1>([&](auto j) { gptr[j] += (long)e * gstride[j]; pptr[j] += (long)e * pstride[j]; });
  auto load_step = [&](double (&H)[N + 1], double (&Gc)[J1], double (&Pi)[J1]) {   // steps e, e-1, ..
    load_row<N + 1>(hp_, H);
    hp_ -= tstride;
    static_for<0, J1>([&](auto j) {
      Gc[j] = *gptr[j];
      Pi[j] = *pptr[j];
      gptr[j] -= gstride[j];
      pptr[j] -= pstride[j];
    });
  };

  auto step = [&](int s, double (&H)[N + 1], double (&Gc)[J1], double (&Pi)[J1],
                  double (&Hn)[N + 1], double (&Gcn)[J1], double (&Pin)[J1]) {
    if (s > 0) load_step(Hn, Gcn, Pin);      // prefetch: hides the L2/HBM latency
    dpp_fence(Gc);
    const int t = dir ? T - 1 - s : s;
    const bool own = s < e || (oddT && !dir);

    // W~[i] = S~[i] G~'  for my rows:  sum_k -/+ bcast_k(S[j]) H[k]
    double W[J1];
    static_for<0, J1>([&](auto j) { W[j] = 0.0; });
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(W[j], S[j], H[k]); });
    });
    dpp_fence(W);
    double WR[N + 2];
    static_for<0, J1>([&](auto j) { pair_split(W[j], WR[2 * j], WR[(2 * j + 1 <= N) ? 2 * j + 1 : N + 1]); });
    // S~_t[i] = P^-1[i] + G~[i] W~ = Pi + sum_k -/+ bcast_k(Gc[j]) WR[k]
    double Sn[J1];
    static_for<0, J1>([&](auto j) { Sn[j] = Pi[j]; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(Sn[j], Gc[j], WR[k]); });
    });

    if (INHOMOG) {
      // per-step pair blocks [E x_t x_t' | E x_t x_{t+1}' | E x_{t+1} x_{t+1}'] for pair index p:
      // the owner of node t writes S~_t into pair t (first block) and pair t-1 (third block); the
      // cross moment W~ = E[x~_{prev} x~_{this}'] is pair s (transposed) for A, pair T-2-s for B.
      double* EP = a.E_pair + (long)b * (T - 1) * 3 * N * N;
      const bool crossw = s < e && !(dir && !oddT && s == e - 1);
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
      if (s == e) {
        static_for<0, J>([&](auto j) { Stop[j] = Sn[j]; });
      } else {
        const double wgt = (s == e - 1) ? wsp : 1.0;
        static_for<0, J>([&](auto j) {
          sumS[j] = __builtin_fma(wgt, Sn[j], sumS[j]);
          sumW[j] = __builtin_fma(wgt, W[j], sumW[j]);
        });
        if (s == e - 1) static_for<0, J>([&](auto j) { Stop[j] = (wsp == 0.0) ? Sn[j] : Stop[j]; });
      }
    }

    // node statistics: diag E[x_t x_t'] (lane i of DPP row i & 1), E[x_t] = row N
    double dg = 0.0;
    static_for<0, J>([&](auto j) { dg = __builtin_fma(ED[j], Sn[j], dg); });
    if (own && col && (c & 1) == gl) a.E_node_diagxx[((long)b * T + t) * N + c] = dg;
    if (own && col && own_N) a.E_node_x[((long)b * T + t) * N + c] = Sn[N / 2];
    static_for<0, J1>([&](auto j) { S[j] = Sn[j]; });
  };

  {
    double Ha[N + 1], Gca[J1], Pia[J1], Hb[N + 1], Gcb[J1], Pib[J1];
    load_step(Ha, Gca, Pia);
    int s = e;
    for (; s >= 1; s -= 2) {          // two steps per trip: the prefetch buffers ping-pong
      step(s, Ha, Gca, Pia, Hb, Gcb, Pib);
      step(s - 1, Hb, Gcb, Pib, Ha, Gca, Pia);
    }
    if (s == 0) step(0, Ha, Gca, Pia, Hb, Gcb, Pib);
  }
#ifdef SVAE_PHASE_TIMING
  TE_TICK(5)
  if (lane == 0) { for (int q = 0; q < 6; ++q) a.E_init[(long)b * (N * N + N) + q] = (double)tm[q]; }
  return;
#endif

  // ---- global statistics ------------------------------------------------------------------------------
  // S = S~ at the chain's end node (x_0 for A, x_{T-1} for B).  Sums over the chain's pairs:
  //   sumS = sum S~(s), s < e(weighted);  sumP = sum S~(s+1) = (sumS - S~(0)) + Stop;  sumW = sum W~(s)
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
  if (!dir) {
    static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl;
      if (i < N && col) a.E_init[(long)b * (N * N + N) + i * N + c] = S[j];
    });
    if (col && own_N) a.E_init[(long)b * (N * N + N) + N * N + c] = S[N / 2];
  }
}

template <int N>
static int launch_estep_twoend(const LdsArgs& a, bool inhomog, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    dim3 grid(a.B), block(64);
    if (inhomog)
      hipLaunchKernelGGL((lds_estep_twoend_kernel<N, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((lds_estep_twoend_kernel<N, false>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
