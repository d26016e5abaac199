// lds_estep_split.hpp -- small-batch ("latency") variant of the LDS E-step kernel.
//
// Same algorithm, results and workspace format as lds_estep_kernel.hpp, different mapping: ONE
// sequence per wavefront.  All four DPP rows hold the same sequence; the Gauss-Jordan inversion runs
// replicated in the four rows (same instructions, no extra cost), while every product stage
//     OUT[i] = sum_k bcast_k(SRC[i]) * B[k]
// is split by OUTPUT ROW across the DPP rows: DPP row g computes matrix rows i = 4j + g ("slot" j),
// so a stage costs ceil(n/4) * n instructions instead of n * n.  The broadcast operand's index k is
// the same in all DPP rows, as row_newbcast requires; B must be replicated, SRC / OUT live in slot
// layout.  Two slot -> replicated all-gathers per time step (next pivot block; cross moments) go
// through 1.5 KB of LDS (same wavefront: no barrier).
//
// When: batches too small to give every SIMD a wavefront with four sequences each (the packed
// kernel leaves 7/8 of an MI355X idle at B = 512).  The dispatcher picks it for B <= SVAE_SPLIT_MAX_B.
#pragma once
#include "lds_estep_kernel.hpp"

namespace svae {

// slot-distributed tile -> replicated tile through LDS.  tab: [16 lanes][RS] doubles per wavefront.
template <int J, int M, int RS>
__device__ __forceinline__ void all_gather_rows(double* tab, int g, int c, const double (&d)[J],
                                                double (&r)[M]) {
  static_assert(RS % 2 == 0 && RS >= 4 * J && RS >= M, "row stride");
  __builtin_amdgcn_wave_barrier();
  static_for<0, J>([&](auto j) { tab[c * RS + 4 * j + g] = d[j]; });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  static_for<0, (M + 1) / 2>([&](auto q) {
    const double2 v = reinterpret_cast<const double2*>(tab + c * RS)[q];
    r[2 * q] = v.x;
    if constexpr (2 * q + 1 < M) r[2 * q + 1] = v.y;
  });
  __builtin_amdgcn_wave_barrier();
}

// FILT: forward filter only (stops after the log-normaliser; the hand-off and the factor region are what the
// sampler and the VJP sweeps read) -- the small-batch twin of lds_estep_kernel<.., FILT = true>, without the
// message outputs.
template <int N, bool INHOMOG, bool CHOL, bool FILT = false>
__global__ __launch_bounds__(64) void lds_estep_split_kernel(const LdsArgs a) {
  static_assert(N >= 1 && N <= SVAE_LDS_MAX_N, "n+1 lanes must fit a 16-lane DPP row");
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N);
  constexpr int J = (N + 3) / 4;          // slots holding rows 0..N-1
  constexpr int J1 = (N + 4) / 4;         // slots holding rows 0..N
  constexpr int RS = 4 * J1;              // LDS row stride (even, >= N+1)
  __shared__ double tab[16 * RS];
  // FILT runs concurrently with the two-ended E-step kernel (svae_lds_estep_f64, keep != 0): touching a high AGPR
  // pushes the wavefront's register allocation past half of the SIMD's file, so that no two wavefronts -- of either
  // kernel -- can share a SIMD (measured without it: the filter slows from 0.34 to 0.46 ms when they do)
  if constexpr (FILT) asm volatile("v_accvgpr_write_b32 a63, 0" ::: "a63");

  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int b = blockIdx.x;               // one sequence per wavefront
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int T = a.T;

  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EN = (c == N) ? 1.0 : 0.0;
  double ED[J1];                          // ED[j][c] = (c == 4j+g): picks S[i][i] in slot layout
  static_for<0, J1>([&](auto j) { ED[j] = (c == 4 * j + g && c < N) ? 1.0 : 0.0; });

  // ---- pair parameters --------------------------------------------------------------------------
  //   replicated: J12c[k][c] = J12[k][c]
  //   slot layout (row i = 4j+g): NJ12T[j][c] = -J12[c][i],  Cc[j][c] = J22[i][c] + J11[i][c]
  double J12c[N], NJ12T[J], Cc[J];
  const double* pJ11 = a.J11 + (long)b * a.pair_seq_stride;
  const double* pJ12 = a.J12 + (long)b * a.pair_seq_stride;
  const double* pJ22 = a.J22 + (long)b * a.pair_seq_stride;
  auto load_pair = [&](int t, bool with_next_J11) {
    const long o = INHOMOG ? (long)t * N * N : 0;
    const long o1 = INHOMOG ? (long)(t + 1) * N * N : 0;
    static_for<0, N>([&](auto k) { const double r = pJ12[o + k * N + cc]; J12c[k] = col ? -r : 0.0; });
    static_for<0, J>([&](auto j) {
      const int i = 4 * j + g;
      const bool ok = col && i < N;
      const int ii = i < N ? i : 0;
      const double r12t = pJ12[o + cc * N + ii], r22 = pJ22[o + ii * N + cc];
      const double r11 = with_next_J11 ? pJ11[o1 + ii * N + cc] : 0.0;
      NJ12T[j] = ok ? r12t : 0.0;
      Cc[j] = ok ? -2.0 * (r22 + r11) : 0.0;
    });
  };
  if (!INHOMOG && T > 1) { load_pair(0, true); dpp_fence(NJ12T); }

  // ---- forward filter (An replicated: lanes < N pivot block, lane N = h_pred) --------------------
  double An[N];
  static_for<0, N>([&](auto i) {
    const double ij = a.init_J[i * N + cc], ih = a.init_h[i], j11 = T > 1 ? pJ11[i * N + cc] : 0.0;
    An[i] = col ? -2.0 * (ij + j11) : ((c == N) ? ih : 0.0);
  });

  const double* nJ = a.node_J + ((long)b * T) * N + cc;
  const double* nh = a.node_h + ((long)b * T) * N + cc;
  double* zpage = a.ws + (long)b * ws_seq_doubles(N, T);   // [e_N | 0] rows for the lanes >= N
  double* wsb = zpage + ws_zpage_doubles(N);
  if (c < HS) zpage[c] = EN;
  if (c < PS) zpage[HS + c] = 0.0;
  double* ws2b = CHOL ? a.ws2 + ((long)b * T) * (N * N + N) + cc : nullptr;

  double qacc = 0.0, ldM = 1.0, pmin = 1.0;
  int ldE = 0;
  // prefetched RAW (no arithmetic on a prefetched value before its step: the multiply would make
  // the wave wait for the load -- and, vmcnt being in-order, for the previous step's stores)
  double Jo_n = nJ[0];             // unconditional loads (lanes >= N read a valid dummy element)
  double ho_n = nh[0];

  for (int t = 0; t < T; ++t) {
    const bool last = (t == T - 1);
    const double Jo = -2.0 * Jo_n;     // scaled one step AFTER its load was issued (see below)
    double ho = ho_n;
    if (!last) {
      Jo_n = nJ[(long)(t + 1) * N];
      ho_n = nh[(long)(t + 1) * N];
    }
    if (INHOMOG && !last) { load_pair(t, t + 1 < T - 1); dpp_fence(NJ12T); }

    double P[N], X[N];
    static_for<0, N>([&](auto i) { P[i] = __builtin_fma(Jo, E[i], An[i]); });
    if (last) {
      asm volatile("; last step: no pair potential, G = 0");
      static_for<0, N>([&](auto i) { X[i] = EN * An[i]; });
    } else {
      static_for<0, N>([&](auto i) { X[i] = __builtin_fma(EN, An[i], J12c[i]); });
    }
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(X[i], ho, EN); });
    dpp_fence(P);

    // replicated Gauss-Jordan: identical to the packed kernel (all four DPP rows do the same work)
    double pv = 0.0;                                   // CHOL: lane k <- pivot k
    double* w2 = CHOL ? ws2b + (long)t * (N * N + N) : nullptr;
    gauss_jordan<N, CHOL>(P, X, E, qacc, pmin, ldM, ldE, pv,
                          [&](auto kk, double r) { if (col && g == 0) w2[kk * N] = r; });
    if constexpr (CHOL) { if (col && g == 0) w2[N * N] = pv; }

    // hand-off rows (identical in the four DPP rows: row 0 stores them)
    double* w = wsb + (long)t * WS;
    if (g == 0 && c <= N) static_for<0, N>([&](auto i) { w[i * HS + c] = X[i]; });
    if (g == 0 && col) static_for<0, N>([&](auto i) { w[N * HS + i * PS + c] = P[i]; });

    if (!last) {
      // next pivot block, slot layout:  An_D[j] = C[j] + sum_k bcast_k(NJ12T[j]) X[k];  lane N: h_pred'
      double AnD[J];
      const bool next_last = (t + 1 == T - 1);
      if (!INHOMOG && next_last) {
        asm volatile("; next step is the last: its pivot block has no J11 term");
        static_for<0, J>([&](auto j) {
          const int i = 4 * j + g;
          const double r = pJ22[(i < N ? i : 0) * N + cc];
          AnD[j] = (col && i < N) ? -2.0 * r : 0.0;
        });
      } else {
        static_for<0, J>([&](auto j) { AnD[j] = Cc[j]; });
      }
      asm volatile("s_nop 1");
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k>(AnD[j], NJ12T[j], X[k]); });
      });
      all_gather_rows<J, N, RS>(tab, g, c, AnD, An);
    }
  }

  // ---- log-normaliser (DPP row 0 reports) --------------------------------------------------------
  {
    double z = 0.0;
    if (a.node_logZ) {
      for (int t = c; t < T; t += 16) z += a.node_logZ[(long)b * T + t];
    }
    if (INHOMOG) {
      const double* lz = a.logZ_pair + (a.pair_seq_stride ? (long)b * (T - 1) : 0);
      for (int t = c; t < T - 1; t += 16) z += lz[t];
    }
    double total = row_sum16(__builtin_fma(0.5, qacc * EN, z));
    total += a.init_logZ[0];
    if (!INHOMOG && T > 1) total += (double)(T - 1) * a.logZ_pair[0];
    total -= 0.5 * (::log(ldM) + (double)ldE * 0.6931471805599453094);
    if (lane == 0) a.lognorm[b] = total;
    const bool bad = !(pmin > 0.0) || !(total == total);
    if (bad && lane == 0) {
      int old = *(volatile int32_t*)a.info;
      while (old == 0 || old > b + 1) {
        const int seen = atomicCAS(a.info, old, b + 1);
        if (seen == old) break;
        old = seen;
      }
    }
  }

  if constexpr (FILT) return;
  // ---- backward pass: S~ in slot layout (row i = 4j+g in DPP row g), W~ gathered per step ---------
  double S[J1];
  static_for<0, J1>([&](auto j) { S[j] = (4 * j + g == N) ? EN : 0.0; });
  dpp_fence(S);
  double sumA[J], sumW[J], Slast[J];
  static_for<0, J>([&](auto j) { sumA[j] = 0.0; sumW[j] = 0.0; Slast[j] = 0.0; });
  const bool own_N = (g == N % 4);        // the DPP row holding row N (E[x_t]) in slot N/4

  // loads of one step: replicated H[k] = H[c][k] (row c of the hand-off tile; lane N: e_N) and, in
  // slot layout / column form, Gc[j][c] = H[4j+g][c] (row N of G~: e_N), Pi[j][c] = P^-1[4j+g][c]
  // every load unconditional: lanes / rows outside the tile read the constant page (stride 0)
  const double* hrow0 = col ? wsb + cc * HS : zpage;
  const long tstride = col ? WS : 0;
  const double* gptr[J1];
  const double* pptr[J1];
  long gstride[J1], pstride[J1];
  static_for<0, J1>([&](auto j) {
    const int i = 4 * j + g;
    const bool gok = i < N && c <= N, pok = i < N && col;
    // row N of G~ is e_N (= the page's first row); rows > N and lanes > N: zeros (the page's 2nd row)
    gptr[j] = gok ? wsb + i * HS + c : ((i == N && c < HS) ? zpage + c : zpage + HS);
    pptr[j] = pok ? wsb + N * HS + i * PS + c : zpage + HS;
    gstride[j] = gok ? WS : 0;
    pstride[j] = pok ? WS : 0;
  });
  // walking pointers (one 64-bit add per pointer and step instead of a 64-bit multiply)
  const double* hp_ = hrow0 + (long)(T - 1) * tstride;
  static_for<0, J1>([&](auto j) { gptr[j] += (long)(T - 1) * gstride[j]; pptr[j] += (long)(T - 1) * pstride[j]; });
  // (unconditional: a prefetch skipped by a branch makes hipcc wait for the loads just issued at the join;
  //  `more` = another step follows, else the pointers stay and the last record is read once more, unused)
  auto load_step = [&](long more, double (&H)[N + 1], double (&Gc)[J1], double (&Pi)[J1]) {   // steps T-1, T-2, ..
    load_row<N + 1>(hp_, H);
    hp_ -= more * tstride;
    static_for<0, J1>([&](auto j) {
      Gc[j] = *gptr[j];
      Pi[j] = *pptr[j];
      gptr[j] -= more * gstride[j];
      pptr[j] -= more * pstride[j];
    });
  };

#ifndef SVAE_WARM_AHEAD
#define SVAE_WARM_AHEAD 4
#endif
  double warm = 0.0, sink = 0.0;
  (void)warm; (void)sink;
  auto step = [&](int t, double (&H)[N + 1], double (&Gc)[J1], double (&Pi)[J1],
                  double (&Hn)[N + 1], double (&Gcn)[J1], double (&Pin)[J1]) {
    load_step(t > 1 ? 1 : 0, Hn, Gcn, Pin);         // prefetch of step t-1: hides the L2/HBM latency
#if SVAE_WARM_AHEAD > 0
    // L2 warming: one 128-byte line per lane of the hand-off record SVAE_WARM_AHEAD steps ahead (the
    // forward half wrote it ~T steps ago: HBM, further away than one step of arithmetic); the value
    // is folded into a dummy one step later, when (vmcnt being in-order) it has long arrived
    sink += warm;
    {
      const int tw = t > SVAE_WARM_AHEAD ? t - SVAE_WARM_AHEAD : 0;
      const int lo = lane * 16 < WS ? lane * 16 : WS - 1;
      warm = wsb[(long)tw * WS + lo];
    }
#endif
    dpp_fence(Gc);

    // W~[i] = S~[i] G~'  for my rows:  sum_k -/+ bcast_k(S[j]) H[k]
    double W[J1];
    static_for<0, J1>([&](auto j) { W[j] = 0.0; });
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(W[j], S[j], H[k]); });
    });
    double WR[N + 1];
    all_gather_rows<J1, N + 1, RS>(tab, g, c, W, WR);
    if (a.ws3) {   // VJP mode: keep W~_t (replicated after the gather: DPP row 0 stores it)
      double* w3 = a.ws3 + ((long)b * T + t) * (N + 1) * HS + c;
      if (g == 0 && c <= N) static_for<0, N + 1>([&](auto i) { w3[i * HS] = WR[i]; });
    }
    // S~_t[i] = P^-1[i] + G~[i] W~ = Pi + sum_k -/+ bcast_k(Gc[j]) WR[k]
    static_for<0, J1>([&](auto j) { S[j] = Pi[j]; });
    asm volatile("s_nop 1");
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k, (k < N)>(S[j], Gc[j], WR[k]); });
    });

    if (INHOMOG) {
      static_for<0, J>([&](auto j) {
        const int i = 4 * j + g;
        if (i < N && col) {
          if (t < T - 1) {
            a.E_pair[(((long)b * (T - 1) + t) * 3 + 0) * N * N + i * N + c] = S[j];
            a.E_pair[(((long)b * (T - 1) + t) * 3 + 1) * N * N + c * N + i] = W[j];
          }
          if (t > 0) a.E_pair[(((long)b * (T - 1) + t - 1) * 3 + 2) * N * N + i * N + c] = S[j];
        }
      });
    } else {
      if (t < T - 1) static_for<0, J>([&](auto j) { sumA[j] += S[j]; sumW[j] += W[j]; });
      else static_for<0, J>([&](auto j) { Slast[j] = S[j]; });
    }

    // node statistics: diag E[x_t x_t'] (lane i of DPP row i%4), E[x_t] = row N
    double dg = 0.0;
    static_for<0, J>([&](auto j) { dg = __builtin_fma(ED[j], S[j], dg); });
    if (col && (c & 3) == g) a.E_node_diagxx[((long)b * T + t) * N + c] = dg;
    if (col && own_N) a.E_node_x[((long)b * T + t) * N + c] = S[N / 4];
  };

  {
    double Ha[N + 1], Gca[J1], Pia[J1], Hb[N + 1], Gcb[J1], Pib[J1];
    load_step(T > 1 ? 1 : 0, Ha, Gca, Pia);
    int t = T - 1;
    for (; t >= 1; t -= 2) {          // two steps per trip: the prefetch buffers ping-pong
      step(t, Ha, Gca, Pia, Hb, Gcb, Pib);
      step(t - 1, Hb, Gcb, Pib, Ha, Gca, Pia);
    }
    if (t == 0) step(0, Ha, Gca, Pia, Hb, Gcb, Pib);
  }

  // ---- global statistics ---------------------------------------------------------------------------
  static_for<0, J>([&](auto j) {
    const int i = 4 * j + g;
    if (i < N && col) {
      double* ei = a.E_init + (long)b * (N * N + N);
      ei[i * N + c] = S[j];
      if (!INHOMOG) {
        double* ep = a.E_pair + (long)b * 3 * N * N;
        ep[i * N + c] = sumA[j];
        ep[N * N + c * N + i] = sumW[j];
        ep[2 * N * N + i * N + c] = (sumA[j] - S[j]) + Slast[j];
      }
    }
  });
  if (col && own_N) a.E_init[(long)b * (N * N + N) + N * N + c] = S[N / 4];
#if SVAE_WARM_AHEAD > 0
  if (sink == 1.2345e300) a.lognorm[b] = sink;        // keeps the touches alive; never true
#endif
}

// filter only, one sequence per wavefront (keeps the factor region: a.ws2 must be set)
template <int N>
static int launch_filter_split(const LdsArgs& a, bool inhomog, hipStream_t stream) {
  dim3 grid(a.B), block(64);
  if (inhomog)
    hipLaunchKernelGGL((lds_estep_split_kernel<N, true, true, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((lds_estep_split_kernel<N, false, true, true>), grid, block, 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <int N>
static int launch_estep_split(const LdsArgs& a, bool inhomog, hipStream_t stream) {
  dim3 grid(a.B), block(64);
  const bool chol = a.ws2 != nullptr;
  if (inhomog && chol)
    hipLaunchKernelGGL((lds_estep_split_kernel<N, true, true>), grid, block, 0, stream, a);
  else if (inhomog)
    hipLaunchKernelGGL((lds_estep_split_kernel<N, true, false>), grid, block, 0, stream, a);
  else if (chol)
    hipLaunchKernelGGL((lds_estep_split_kernel<N, false, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((lds_estep_split_kernel<N, false, false>), grid, block, 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace svae
