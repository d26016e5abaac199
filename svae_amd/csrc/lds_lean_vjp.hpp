// lds_lean_vjp.hpp -- the two VJP sweeps on the LEAN records of lds_lean_estep.hpp (round 6).
//
// What it replaces (reference = mattjj/svae, /root/reference), as lds_vjp_kernel.hpp:
//   natural_filter_grad            svae/lds/cython_lds_inference.pyx:92-145
//   natural_smoother_general_grad  svae/lds/cython_lds_inference.pyx:236-306
//   natural_sample_backward_grad   svae/lds/cython_lds_inference.pyx:357-409
// The SAME adjoint recursions as the packed sweeps of lds_vjp_kernel.hpp (four sequences per wavefront, one role-2
// workgroup doing the smoother and the sampler adjoints; derivation there), for batches where those sweeps wait on HBM
// (round 5 at 4096 x 200 x 10: 5.5 + 3.6 GB per pass).  What differs is the traffic:
//   * the forward record of a step is [U = chol(P)^-T packed | c] (66 doubles at n = 10; lds_args.hpp lean_*): each sweep
//     rebuilds P^-1 = U U' and P^-1 J12 (155 DPP multiply-adds), and the noise adjoint takes U as stored (the packed
//     sweep re-derived it from the LDL' factor: 45 multiply-adds + a reciprocal square root);
//   * sweep 1 hands sweep 2 ONE symmetric triangle per step, -P^-1 Pinvbar P^-1 + sym(Pbar(direct)), instead of a
//     triangle and a full n x n matrix: Pbar only ever acts through congruences (X Pbar X' into the previous step) and
//     through its diagonal (g_node_J), all of which commute with symmetrisation.
#pragma once
#include "lds_vjp_kernel.hpp"

namespace svae {

// P^-1 (rows, lanes < N; zero elsewhere) from the lean record's U rows (DPP-read) and masked transposed rows
template <int N>
__device__ __forceinline__ void lean_pinv(double (&Pi)[N], double (&Ur)[N], const double (&Ut)[N]) {
  static_for<0, N>([&](auto i) { Pi[i] = 0.0; });
  dpp_fence(Ur);
  static_for<0, N>([&](auto k) {
    constexpr int kk = decltype(k)::value;
    static_for<0, kk + 1>([&](auto i) { mac_bc<kk>(Pi[i], Ur[i], Ut[kk]); });
  });
}

// ---- sweep 1: smoother + sampler adjoints, forward in time ---------------------------------------------------------
// Register budget (the packed sweep 1 with samples already sits at 256 VGPRs): the loop carries S^ (n+1 rows), the
// prefetched W~', the masks and -- instead of X_{t-1} AND xhat_{t-1} -- their product X_{t-1}' xhat_{t-1}, formed at the end
// of the step that owns X_{t-1}; the sampler role runs first and leaves its two contributions (sym Pbar(direct), its
// share of G^) behind, then the smoother role stores G^ and the triangle as soon as they are complete and propagates S^.
// The identity tile is spelled through the masks ((c == i) = M[i] - M[i-1]); J12 comes from L1 each step.
template <int N, bool SAMP>
__global__ __launch_bounds__(64) void lds_vjp_sweep1_lean_kernel(const VjpArgs a) {
  static_assert(N >= 1 && N <= LEAN_MAX_N, "lean records: n <= 10");
  constexpr int HS = ws_h_stride(N), W3 = (N + 1) * HS;
  constexpr int TRI = lean_tri(N), LR = lean_rec_doubles(N), AS = lean_adj_doubles(N), TO = lean_adj_tri_off(N);
  __shared__ double tabs[4 * 256];
  const int lane = threadIdx.x & 63;
  const int c = lane & 15;
  double* tab = tabs + (lane >> 4) * 256;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N, colN = c <= N;
  const int T = a.T, S = a.S;
  const int cN = colN ? c : 0, cc = col ? c : 0;
  const double EN = (c == N) ? 1.0 : 0.0;
  const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);        // G~ row k = H row k * sg
  const double cm = col ? 1.0 : 0.0;
  double M[N];                                                // zeros of U': M[k][c] = (c <= k)
  static_for<0, N>([&](auto k) { M[k] = (col && c <= k) ? 1.0 : 0.0; });
  const double* pJ12 = a.J12 + cc;                            // (T > 1; nat J12 rows: info form = -nat)

  const double* recb = a.ws + (long)b * T * LR;
  const double* ur0 = recb + cc;                              // row k of U: + row_off(k) - k
  const double* ut0 = recb + (lean_row_off(N, cc) - cc);      // U[c][k]:   + k
  double Sh[N + 1];
  static_for<0, N + 1>([&](auto i) { Sh[i] = 0.0; });
  double xc[SAMP ? N : 1];                                    // X_{t-1}' xhat_{t-1}  (register k, lane = sample)
  if constexpr (SAMP) static_for<0, N>([&](auto k) { xc[k] = 0.0; });
  const bool sv = c < S;                                      // this lane carries a sample
  const int ss = sv ? c : 0;
  const int SN = S * N;
  const bool has_gx = a.g_x != nullptr, has_gd = a.g_diagxx != nullptr;

  // operands the smoother role STARTS with, fetched one step ahead: W~' and the direct cotangents
  double WTn[N + 1], gxn = 0.0, gdn = 0.0;
  auto fetch_next = [&](int t) {
    load_row<N + 1>(a.ws3 + ((long)b * T + t) * W3 + cN * HS, WTn);
    const long o = ((long)b * T + t) * N + cc;
    gxn = has_gx ? a.g_x[o] : 0.0;
    gdn = has_gd ? a.g_diagxx[o] : 0.0;
  };
  // (lanes N+1 .. 15 of every DPP row switched off through the time loop: see lds_lean_estep.hpp, SVAE_LEAN_LANEMASK)
#ifndef SVAE_LEAN_LANEMASK
#define SVAE_LEAN_LANEMASK 1
#endif
  if (SVAE_LEAN_LANEMASK && !colN) return;
  fetch_next(0);
  for (int t = 0; t < T; ++t) {
    double* ad = a.adj + ((long)b * T + t) * AS;
    // the step's lean record and the sampler role's operands, raw
    double Ur[N], Utr[N];
    static_for<0, N>([&](auto k) {
      Ur[k] = ur0[(long)t * LR + (lean_row_off(N, k) - k)];
      Utr[k] = ut0[(long)t * LR + k];
    });
    const double cvr = recb[(long)t * LR + TRI + cc];
    [[maybe_unused]] double gsv[SAMP ? N : 1], x1r[SPRE], epr[SPRE];
    if constexpr (SAMP) {
      const double* gs = a.g_samples + (((long)b * T + t) * S + ss) * N;
      static_for<0, N>([&](auto k) { gsv[k] = gs[k]; });
      const double* x1rec = a.samples + ((long)b * T + (t + 1 < T ? t + 1 : t)) * SN;   // x_{t+1}, per sample
      const double* eprec = a.eps + ((long)b * T + t) * SN;
      static_for<0, SPRE>([&](auto s) {
        const int sq = s < S ? (int)s : 0;
        x1r[s] = x1rec[sq * N + cc];
        epr[s] = eprec[sq * N + cc];
      });
    }

    // rebuild P^-1 and H = [P^-1 J12 | c] (rows; lanes > N zero)
    double Pi[N], Hc[N];
    {
      double Ut[N];
      static_for<0, N>([&](auto k) { Ut[k] = Utr[k] * M[k]; });
      lean_pinv<N>(Pi, Ur, Ut);
    }
    {
      double cvm = cvr * cm, J12c[N];
      if (T > 1) static_for<0, N>([&](auto k) { J12c[k] = pJ12[k * N]; });
      else static_for<0, N>([&](auto k) { J12c[k] = 0.0; });
      static_for<0, N>([&](auto k) { J12c[k] = col ? J12c[k] : 0.0; });
      static_for<0, N>([&](auto i) { Hc[i] = 0.0; });
      dpp_fence(cvm);
      static_for<0, N>([&](auto i) { mac_bc<i>(Hc[i], cvm, EN); });                // lane N: c_i
      mm_ab<N, N, true>(Hc, Pi, J12c);                                             // - P^-1 natJ12 = P^-1 J12(info)
    }

    double Gb[N], Pbp[N];                      // G^ rows i < N (lanes 0..N);  the symmetric share of Pbar
    static_for<0, N>([&](auto i) { Gb[i] = 0.0; Pbp[i] = 0.0; });
    if constexpr (SAMP) {
      // ---- sampler role ------------------------------------------------------------------------------------------------
      // xhat_t = g_samples_t - X_{t-1}' xhat_{t-1}   (register k, lane = sample)
      double xh[N];
      static_for<0, N>([&](auto k) { xh[k] = (sv ? gsv[k] : 0.0) - xc[k]; });
      dpp_fence(xh);
      // cbar_t += sum_s xhat (lane N);  Xbar_t -= sum_s xhat x_{t+1}'  (i.e. G^[:, :n] += ...)
      const double* x1rec = a.samples + ((long)b * T + (t + 1 < T ? t + 1 : t)) * SN;
      const double* eprec = a.eps + ((long)b * T + t) * SN;
      for_samples<0>(S, [&](auto s) {
        double v = EN;
        if (t + 1 < T) {
          const double x1 = (s < SPRE) ? x1r[s < SPRE ? (int)s : 0] : x1rec[s * N + cc];
          v += col ? x1 : 0.0;
        }
        asm volatile("s_nop 1");   // block entry after the branch: two wait states before the DPP reads (audit rule)
        static_for<0, N>([&](auto i) { mac_bc<s>(Gb[i], xh[i], v); });
      });
      // noise adjoint:  Pbar_t(direct) = -U (Lh U'),  U = chol(P)^-T as stored (upper triangular: only lanes >= the row
      // index are ever broadcast),  Lh from E' = sum_s eps_s z_s',  z = U' xhat
      double z[N], ET[N];
      static_for<0, N>([&](auto j) { z[j] = 0.0; ET[j] = 0.0; });
      asm volatile("s_nop 1");     // block entry behind the sample loop: two wait states before the DPP reads (audit rule; n = 1)
      static_for<0, N>([&](auto i) {
        constexpr int ii = decltype(i)::value;
        static_for<ii, N>([&](auto j) { mac_bc<j>(z[j], Ur[ii], xh[ii]); });
      });
      dpp_fence(z);
      for_samples<0>(S, [&](auto s) {
        const double e1 = (s < SPRE) ? epr[s < SPRE ? (int)s : 0] : eprec[s * N + cc];
        const double ev = col ? e1 : 0.0;
        asm volatile("s_nop 1");   // block entry (audit rule)
        static_for<0, N>([&](auto j) { mac_bc<s>(ET[j], z[j], ev); });         // ET[j][c] = E[c][j]
      });
      // Lh' rows: ET[j] * ((c > j) + 1/2 (c == j)),  (c > j) + 1/2 (c == j) = 1 - 1/2 (M[j] + M[j-1]) on the lanes < N
      double K[N], KT[N], Pex[N], PexT[N];
      static_for<0, N>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        double m;
        if constexpr (jj == 0) m = __builtin_fma(-0.5, M[0], 1.0);
        else m = __builtin_fma(-0.5, M[jj] + M[jj - 1], 1.0);
        ET[j] *= m;
        K[j] = 0.0; Pex[j] = 0.0;
      });
      static_for<0, N>([&](auto k) {                                             // K = U Lh'
        constexpr int kk = decltype(k)::value;
        static_for<0, kk + 1>([&](auto i) { mac_bc<kk>(K[i], Ur[i], ET[kk]); });
      });
      transpose_tile<N>(tab, c, K, KT);                                          // KT = Lh U'
      static_for<0, N>([&](auto k) {                                             // Pex = -U Lh U'
        constexpr int kk = decltype(k)::value;
        static_for<0, kk + 1>([&](auto i) { mac_bc<kk, true>(Pex[i], Ur[i], KT[kk]); });
      });
      transpose_tile<N>(tab, c, Pex, PexT);
      static_for<0, N>([&](auto i) { Pbp[i] = 0.5 * (Pex[i] + PexT[i]); });
      // what the next step's xhat needs of this one:  X_t' xhat_t
      static_for<0, N>([&](auto k) { xc[k] = 0.0; });
      dpp_fence(Hc);
      static_for<0, N>([&](auto j) {
        static_for<0, N>([&](auto k) { mac_bc<k>(xc[k], Hc[j], xh[j]); });
      });
    }

    // ---- smoother role ---------------------------------------------------------------------------------------------------
    {
      double WT[N + 1];
      static_for<0, N + 1>([&](auto k) { WT[k] = colN ? 2.0 * WTn[k] : 0.0; });    // 2 W~'
      double gx = col ? 0.5 * gxn : 0.0;         // direct cotangents, symmetrised
      const double gd = col ? gdn : 0.0;
      fetch_next(t + 1 < T ? t + 1 : t);
      Sh[N] += gx;
      dpp_fence(gx);
      static_for<0, N>([&](auto i) {
        constexpr int ii = decltype(i)::value;
        mac_bc<ii>(Sh[ii], gx, EN);
        if constexpr (ii == 0) Sh[0] = __builtin_fma(gd, M[0], Sh[0]);             // S^[i][i] += g_diagxx[i]
        else Sh[ii] = __builtin_fma(gd, M[ii] - M[ii - 1], Sh[ii]);
      });
      // G^ rows i < N:  2 S^ W~'   (lanes 0..N)
      mm_ab<N, N + 1, false>(Gb, Sh, WT);
      if (valid && colN) static_for<0, N>([&](auto i) { ad[i * HS + c] = Gb[i]; });
    }
    {
      // -P^-1 Pinvbar P^-1,  Pinvbar = S^[:n,:n] (before the propagation below)
      double Pib[N], T1[N];
      static_for<0, N>([&](auto i) { Pib[i] = Sh[i] * cm; T1[i] = 0.0; });
      mm_ab<N, N, false>(T1, Pib, Pi);
      mm_ab<N, N, true>(Pbp, Pi, T1);
      if (valid) static_for<0, N>([&](auto i) { if (c <= i) ad[TO + i * (i + 1) / 2 + c] = Pbp[i]; });
    }
    // S^ <- G~' (S^ G~)
    {
      double Gc[N + 1], Mm[N + 1], Sn[N + 1];
      static_for<0, N>([&](auto k) { Gc[k] = Hc[k] * sg; });
      Gc[N] = EN;
      static_for<0, N + 1>([&](auto i) { Mm[i] = 0.0; Sn[i] = 0.0; });
      mm_ab<N + 1, N + 1, false>(Mm, Sh, Gc);
      mm_atb<N + 1, N + 1, false>(Sn, Gc, Mm);
      static_for<0, N + 1>([&](auto i) { Sh[i] = Sn[i]; });
    }
  }
}

// ---- sweep 2: filter adjoint, backward in time ------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void lds_vjp_sweep2_lean_kernel(const VjpArgs a) {
  static_assert(N >= 1 && N <= LEAN_MAX_N, "lean records: n <= 10");
  constexpr int HS = ws_h_stride(N);
  constexpr int TRI = lean_tri(N), LR = lean_rec_doubles(N), AS = lean_adj_doubles(N), TO = lean_adj_tri_off(N);
  const int lane = threadIdx.x & 63;
  const int T = a.T;
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N, colN = c <= N;
  const int cN = colN ? c : 0, cc = col ? c : 0;
  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EN = (c == N) ? 1.0 : 0.0;
  const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);
  const double cm = col ? 1.0 : 0.0;
  double M[N], J12c[N], NJ12T[N];           // zeros of U';  info-form J12 rows (= -nat);  nat J12 transposed
  static_for<0, N>([&](auto k) {
    M[k] = (col && c <= k) ? 1.0 : 0.0;
    const double v = T > 1 ? a.J12[k * N + cc] : 0.0, vt = T > 1 ? a.J12[cc * N + k] : 0.0;
    J12c[k] = col ? -v : 0.0;
    NJ12T[k] = col ? vt : 0.0;
  });
  dpp_fence(NJ12T);
  const double g = a.g_lognorm[b];
  const double* recb = a.ws + (long)b * T * LR;
  const double* ur0 = recb + cc;
  const double* ut0 = recb + (lean_row_off(N, cc) - cc);
  double Ab[N];                                               // [Abar | hbar] of step t+1
  static_for<0, N>([&](auto i) { Ab[i] = 0.0; });
  int symo[N];                                                 // entry (i, c) of the symmetric triangle
  static_for<0, N>([&](auto i) { symo[i] = cc <= i ? i * (i + 1) / 2 + cc : cc * (cc + 1) / 2 + i; });

  double gbn[N];                                               // G^ of the step, fetched one step ahead
  auto fetch_g = [&](int t) {
    const double* ad = a.adj + ((long)b * T + t) * AS;
    static_for<0, N>([&](auto i) { gbn[i] = ad[i * HS + cN]; });
  };
  if (SVAE_LEAN_LANEMASK && !colN) return;   // (lanes N+1 .. 15 of every DPP row: nothing to do in this kernel)
  fetch_g(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    const double* ad = a.adj + ((long)b * T + t) * AS;
    double Xc[N];
    static_for<0, N>([&](auto i) { Xc[i] = colN ? gbn[i] * sg : 0.0; });      // [Xbar | cbar] = [-G^ | G^[:,n]]
    double Ur[N], Utr[N], Pb[N];
    static_for<0, N>([&](auto k) {
      Ur[k] = ur0[(long)t * LR + (lean_row_off(N, k) - k)];
      Utr[k] = ut0[(long)t * LR + k];
    });
    const double cvr = recb[(long)t * LR + TRI + cc];
    static_for<0, N>([&](auto i) { Pb[i] = ad[TO + symo[i]]; });                // sweep-1 share of Pbar
    fetch_g(t > 0 ? t - 1 : 0);
    // [Xbar | cbar] -= J12_t [Abar | hbar]_{t+1}
    if (t < T - 1) mm_ab<N, N, true>(Xc, J12c, Ab);
    // rebuild P^-1 and H' (HT[k][c] = (P^-1 J12)[c][k], HT[N] = c')
    double Ut[N], Pi[N], HT[N + 1];
    static_for<0, N>([&](auto k) { Ut[k] = Utr[k] * M[k]; });
    double cvm = cvr * cm;
    lean_pinv<N>(Pi, Ur, Ut);
    static_for<0, N>([&](auto k) { HT[k] = 0.0; });
    static_for<0, N>([&](auto j) {
      static_for<0, N>([&](auto k) { mac_bc<j, true>(HT[k], NJ12T[k], Pi[j]); });
    });
    HT[N] = cvm;
    // Bbar = P^-1 [Xbar | cbar]
    double Bb[N];
    static_for<0, N>([&](auto i) { Bb[i] = 0.0; });
    mm_ab<N, N, false>(Bb, Pi, Xc);
    // Pbar = [sweep-1 share] - Bbar H' - 1/2 g (c c' + P^-1)
    mm_ab<N, N + 1, true>(Pb, Bb, HT);
    double cvs = -0.5 * g * cvm;
    dpp_fence(cvs);
    static_for<0, N>([&](auto i) {
      mac_bc<i>(Pb[i], cvs, cvm);
      Pb[i] = __builtin_fma(-0.5 * g, Pi[i], Pb[i]);
    });
    // outputs and the adjoint handed to step t-1:  Ab = [Pbar | Bbar[:,n] + g c]
    double gc = g * cvm, gJ = 0.0, gh = 0.0;
    dpp_fence(gc);
    static_for<0, N>([&](auto i) {
      double hfb = Bb[i];
      mac_bc<i>(hfb, gc, EN);                                  // lane N: hfbar_i = Bbar[i][n] + g c_i
      Ab[i] = __builtin_fma(EN, hfb - Pb[i], Pb[i]);
      gJ = __builtin_fma(E[i], Pb[i], gJ);
    });
    dpp_fence(Ab);
    static_for<0, N>([&](auto i) { mac_bc<N>(gh, Ab[i], E[i]); });
    if (valid && col) {
      a.g_node_J[((long)b * T + t) * N + c] = -2.0 * gJ;
      a.g_node_h[((long)b * T + t) * N + c] = gh;
    }
  }
}

template <int N>
static int launch_vjp_lean(const VjpArgs& a, hipStream_t stream) {
  if constexpr (N <= LEAN_MAX_N) {
    dim3 grid((a.B + 3) / 4), block(64);
    if (a.g_samples) hipLaunchKernelGGL((lds_vjp_sweep1_lean_kernel<N, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lds_vjp_sweep1_lean_kernel<N, false>), grid, block, 0, stream, a);
    hipLaunchKernelGGL((lds_vjp_sweep2_lean_kernel<N>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
