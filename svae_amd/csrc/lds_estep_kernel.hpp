// lds_estep_kernel.hpp -- batched natural-parameter LDS E-step for MI355X (gfx950, wave64, fp64).
//
// What it replaces (reference = mattjj/svae, /root/reference):
//   natural_filter_forward_general   svae/lds/cython_lds_inference.pyx:28-90
//     _natural_condition_diag        svae/lds/cython_gaussian_grads.pxd:125-144
//     _natural_predict               svae/lds/cython_gaussian_grads.pxd:38-82
//     _natural_lognorm               svae/lds/cython_gaussian_grads.pxd:167-188
//   natural_smoother_general         svae/lds/cython_lds_inference.pyx:149-195
//     _rts_backward_step             svae/lds/cython_gaussian_grads.pxd:240-296
//     _info_to_mean                  svae/lds/cython_gaussian_grads.pxd:208-220
//   _compute_stats                   svae/lds/cython_lds_inference.pyx:197-210
//
// This is NOT a translation of those LAPACK call sequences.  Mapping to the hardware:
//   * one DPP row (16 lanes) per sequence, 4 sequences per wavefront, one wavefront per workgroup;
//     lane c holds column c of every n x n block, one VGPR pair per block row (n <= 15);
//   * every small dense product is "broadcast lane k of my row" x "my column": a v_mov_b64_dpp
//     row_newbcast + v_fma_f64 -- no LDS traffic, no SGPR round trips, no cross-row shuffles;
//   * the forward pass inverts P_t = J_filt,t + J11 by in-place Gauss-Jordan (pivots = the LDL'
//     diagonal, so log|P_t| comes for free; reciprocal = v_rcp_f64 + 2 Newton steps, no sqrt/div);
//   * the backward pass runs in MOMENT form on homogeneous coordinates x~ = [x;1]:
//         S~_t = G~_t S~_{t+1} G~_t' + diag(P_t^-1, 0),   G~_t = [[-P_t^-1 J12, P_t^-1 h_filt,t],[0,1]]
//         E[x~_{t+1} x~_t'] = S~_{t+1} G~_t'
//     i.e. two (n+1)^3 products per step and NO factorisation on the backward critical path (the
//     reference's information-form RTS step does 2 Cholesky + potri + 3 triangular solves per step).
//     S~ carries E[x x'], E[x] and 1 together, so means, second moments and cross moments fall out
//     of the same two products; sums of PSD terms only (no cancelling subtractions);
//   * J12 and h_filt ride through the elimination as extra right-hand-side columns (h in lane n), so
//     G_t = -P^-1 J12 and c_t = P^-1 h_filt have solve-quality accuracy (not inverse-times-matrix);
//   * forward -> backward hand-off ([P^-1 J12 | c] and P^-1 per step) goes through an HBM workspace
//     written and re-read by the SAME wave (no inter-workgroup communication); the backward half
//     reads it row-wise with 16-byte loads (the transposition G -> G' is done by the addressing).
//
// Algorithmic HBM bytes per sequence: 8*[T(2n+1) + 2Tn + 4n^2 + n + 1];
// workspace traffic on top of that is 2 * 8 * T * ws_step_doubles(n) (write + read).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "lds_args.hpp"

namespace svae {

// rows i0, i0+1, .. i0+IL-1 of  OUT[i] += sum_k (+/-) bcast_k(SRC[i]) * Bt[k]   (k = 0..KN-1; the
// first NEGK terms are subtracted), the IL accumulation chains interleaved so that consecutive
// instructions are independent.
template <int IL, int I0, int KN, int NEGK, int MS, int MB, int MO>
__device__ __forceinline__ void rows_src_bcast(double (&out)[MO], const double (&src)[MS],
                                               const double (&bt)[MB]) {
  static_for<0, KN>([&](auto k) {
    static_for<0, IL>([&](auto j) {
      if constexpr (I0 + j < MO && I0 + j < MS) mac_bc<k, (k < NEGK)>(out[I0 + j], src[I0 + j], bt[k]);
    });
  });
}

// rows i0.. of  OUT[i] += sum_k (+/-) Gt[k] * bcast_i(W[k])   (the transposed product of the backward step)
template <int IL, int I0, int KN, int NEGK, int MW, int MG, int MO>
__device__ __forceinline__ void rows_lane_bcast(double (&out)[MO], const double (&w)[MW],
                                                const double (&gt)[MG]) {
  static_for<0, KN>([&](auto k) {
    static_for<0, IL>([&](auto j) {
      if constexpr (I0 + j < MO) mac_bc<I0 + j, (k < NEGK)>(out[I0 + j], w[k], gt[k]);
    });
  });
}

// ---- in-place Gauss-Jordan on [P | X] ------------------------------------------------------------
// P (SPD, column layout) -> P^-1, X -> P^-1 X; pivots = LDL' diagonal (no pivoting needed).  Per pivot k:
// scaled pivot rows r, rx; every other row i: lane k of P[i] becomes the inverse's column (0 - m/p),
// the other lanes P[i] - m r, and X[i] -= m rx, with m = P[i][k] broadcast from lane k (DPP).
// Software pipelining by hand (hipcc would otherwise put the whole reciprocal chain of pivot k+1
// behind the row updates of pivot k, stalling ~100 cycles per pivot): row k+1 is updated first, its
// pivot is broadcast, and v_rcp_f64 + the two Newton steps are issued as asm statements BETWEEN the
// remaining row updates.
#ifndef SVAE_GJ_ONE_PLUS
#define SVAE_GJ_ONE_PLUS 0   // 0: exact two-instruction form (lane clear + FMA); 1: the one-FMA shortcut (A/B only)
#endif
template <int N, bool CHOL, class StoreR>
__device__ __forceinline__ void gauss_jordan(double (&P)[N], double (&X)[N], const double (&E)[N],
                                             double& qacc, double& pmin, double& ldM, int& ldE,
                                             double& pv, StoreR&& store_r) {
  double p = bcast_fenced<0>(P[0]);
  double rinv = rcp_nr(p);
  double pprod = 1.0;
  static_for<0, N>([&](auto k) {
    pmin = fmin(pmin, p);
    pprod *= p;
    // scaled pivot rows; lane k of r gets 1/p (the inverse's diagonal entry)
    const double r = __builtin_fma(E[k], 1.0 - p, P[k]) * rinv;
    const double rx = X[k] * rinv;
    qacc = __builtin_fma(X[k], rx, qacc);            // lane N: z_k^2 / d_k  (h' P^-1 h = sum_k)
    if constexpr (CHOL) {
      pv = __builtin_fma(E[k], p, pv);
      store_r(k, r);                                 // lanes j > k hold L_unit[j][k] (P = L D L')
    }
    // One instruction per row and block (E-step without hand-off to the sampler / VJP): lane k of the
    // row holds the multiplier f and must become -f/p; with r'[k] = 1 + 1/p the same FMA that updates
    // the other lanes gives f - f (1 + 1/p).  Rounding 1 + 1/p costs a relative (1 + p) 2^-53 in that
    // entry: against the extended-precision arbiter on models with pivots up to 1e8 the statistics
    // move from 8e-16 to 3e-13 relative at worst, far inside the conditioning noise of realistic
    // models (1e-12 .. 1e-9); it removes 27 % of the Gauss-Jordan's instructions (+6.6 % on the
    // headline).  With CHOL (factor kept for the sampler and the VJP) the exact two-instruction form
    // stays: gradients of ill-conditioned per-step models amplified the difference to 5e-6.
    constexpr bool ONE_PLUS = (SVAE_GJ_ONE_PLUS != 0) && !CHOL;
    const double rp = r + E[k];
    auto update = [&](auto i, auto fenced) {
      if constexpr (ONE_PLUS) {
        mac_bc<k, true, decltype(fenced)::value>(X[i], P[i], rx);
        mac_bc<k, true>(P[i], P[i], rp);
      } else {
        const double old = P[i];
        double acc = __builtin_fma(-old, E[k], old);
        mac_bc<k, true, decltype(fenced)::value>(acc, old, r);
        mac_bc<k, true>(X[i], old, rx);
        P[i] = acc;
      }
    };
    if constexpr (k + 1 < N) {
      update(std::integral_constant<int, k + 1>{}, std::true_type{});
      const double pn = bcast_fenced<k + 1>(P[k + 1]);
      // reciprocal chain of the NEXT pivot, one step after every other remaining row update
      double t0 = 0.0, e0 = 0.0, t1 = 0.0, e1 = 0.0, rn = 0.0;
      constexpr int REM = N - 2;                       // row updates still to come
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
          constexpr int pos = i - (i > k ? 1 : 0) - (i > k + 1 ? 1 : 0);     // 0 .. REM-1
          update(i, std::false_type{});
          // spread the 5 chain steps over the REM gaps (all remaining ones after the last update)
          constexpr int lo = pos * 5 / REM, hi = (pos + 1) * 5 / REM;
          static_for<lo, hi>(chain);
        }
      });
      p = pn;
      rinv = rn;
    } else {
      static_for<0, N>([&](auto i) {
        if constexpr (i != k) {
          if constexpr (i == 0 || (k == 0 && i == 1)) update(i, std::true_type{});
          else update(i, std::false_type{});
        }
      });
    }
    P[k] = r;
    X[k] = rx;
    if constexpr (k == N / 2 || k == N - 1) {      // keep the running product in range
      ldE += __builtin_amdgcn_frexp_exp(pprod);
      ldM *= __builtin_amdgcn_frexp_mant(pprod);
      pprod = 1.0;
    }
  });
  const int e = __builtin_amdgcn_frexp_exp(ldM);
  ldM = __builtin_amdgcn_frexp_mant(ldM);
  ldE += e;
}

#ifndef SVAE_IL
#define SVAE_IL 4     // independent accumulation chains interleaved per DPP product stage
#endif

// contiguous row of CNT doubles at a 16-byte aligned address -> registers (8- or 16-byte loads)
template <int CNT, int M>
__device__ __forceinline__ void load_row(const double* __restrict__ p, double (&r)[M]) {
  static_for<0, CNT / 2>([&](auto j) {
    const double2 v = reinterpret_cast<const double2*>(p)[j];
    r[2 * j] = v.x;
    if constexpr (2 * j + 1 < M) r[2 * j + 1] = v.y;
  });
  if constexpr (CNT % 2 == 1) r[CNT - 1] = p[CNT - 1];
}

// FILT: filter only (natural_filter_forward_general, cython_lds_inference.pyx:28-90): stops after the
// log-normaliser, keeps the hand-off and the factor region for the sampler (cython_natural_lds_sample,
// lds_inference.py:260-264) and, on request, writes the forward messages in the reference's scaling.
template <int N, bool INHOMOG, bool CHOL, bool FILT = false>
__global__ __launch_bounds__(64) void lds_estep_kernel(const LdsArgs a) {
  static_assert(N >= 1 && N <= SVAE_LDS_MAX_N, "n+1 lanes must fit a 16-lane DPP row");
  constexpr int IL = SVAE_IL;
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N);
  // Above n = 10 the register tiles no longer fit 256 VGPRs: keep fewer constants resident (reload
  // J12/J22 from L2 when needed) and do not double-buffer the backward loads.  (AGPR spill moves are
  // VALU writes, i.e. DPP hazards the compiler cannot see around the inline asm: avoid them.)
  constexpr bool LOWREG = N > 10;
  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;   // surplus rows recompute the last sequence, stores masked
  const bool col = c < N;
  const bool st = valid && col;
  const bool sth = valid && c <= N;
  const int cc = col ? c : 0;
  const int T = a.T;

  // identity tile: E[i][c] = (c == i), EN[c] = (c == N).  Per-lane selects are done arithmetically
  // with it (x*E, exact) instead of v_cndmask: on gfx950 v_cndmask throughput is shared by the whole
  // CU (measured: tools/ubench/valu_rates.hip), which throttles the kernel once >1 wave runs per CU.
  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EN = (c == N) ? 1.0 : 0.0;

  // ---- pair parameters (info form: J = -2 natJ, J12 = -natJ12), column c per lane -------------
  //   NJ12T[i][c] = -J12[c][i]      J12c[k][c] = J12[k][c]      (lanes >= N: 0)
  //   Cc[i][c]    =  J22[i][c] + J11[i][c]   (next step's pivot block without the node diagonal)
  double NJ12T[N], J12c[LOWREG ? 1 : N], Cc[N];   // (J22 alone is re-read from L2 for the one step that needs it)
  const double* pJ11 = a.J11 + (long)b * a.pair_seq_stride;
  const double* pJ12 = a.J12 + (long)b * a.pair_seq_stride;
  const double* pJ22 = a.J22 + (long)b * a.pair_seq_stride;
  auto load_pair = [&](int t, bool with_next_J11) {   // pair t (and J11 of pair t+1)
    const long o = INHOMOG ? (long)t * N * N : 0;
    const long o1 = INHOMOG ? (long)(t + 1) * N * N : 0;
    static_for<0, N>([&](auto i) {      // unconditional loads, selected afterwards
      const double r12t = pJ12[o + cc * N + i], r22 = pJ22[o + i * N + cc], r12 = pJ12[o + i * N + cc];
      const double r11 = with_next_J11 ? pJ11[o1 + i * N + cc] : 0.0;
      NJ12T[i] = col ? r12t : 0.0;
      if constexpr (!LOWREG) J12c[i] = col ? -r12 : 0.0;
      Cc[i] = col ? -2.0 * (r22 + r11) : 0.0;
    });
  };
  if (!INHOMOG && T > 1) { load_pair(0, true); dpp_fence(NJ12T); }

  // ---- forward filter --------------------------------------------------------------------------
  // An: lanes < N = pivot block of the current step without the node diagonal (J_pred + J11; J_pred
  // alone at t = T-1); lane N = h_pred (column layout: register i holds component i).
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
  // CHOL: also keep the unit-upper factor rows (the scaled pivot rows) and the pivots of P_t for the
  // backward sampler (svae_lds_sample_f64): N*N + N doubles per step in the second workspace region.
  double* ws2b = CHOL ? a.ws2 + ((long)b * T) * (N * N + N) + cc : nullptr;

  double qacc = 0.0;       // lane N: sum_t h_filt' P^-1 h_filt (other lanes: unused partials)
  double ldM = 1.0;        // log|P_t| accumulated as mantissa product ...
  int ldE = 0;             // ... and exponent sum (one log at the very end)
  double pmin = 1.0;       // smallest pivot seen (<= 0 => not positive definite)

  // prefetched RAW (no arithmetic on a prefetched value before its step: the multiply would make
  // the wave wait for the load -- and, vmcnt being in-order, for the previous step's stores)
  double Jo_n = nJ[0];             // unconditional loads (lanes >= N read a valid dummy element)
  double ho_n = nh[0];

#ifdef SVAE_PHASE_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TICK(i) { const long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tlast_; tlast_ = now_; }
  long long tlast_ = __builtin_readcyclecounter();
#else
#define TICK(i)
#endif
  for (int t = 0; t < T; ++t) {
    const bool last = (t == T - 1);
    const double Jo = -2.0 * Jo_n;     // scaled one step AFTER its load was issued (see below)
    double ho = ho_n;
    if (!last) {
      Jo_n = nJ[(long)(t + 1) * N];
      ho_n = nh[(long)(t + 1) * N];
    }
    if (INHOMOG && !last) { load_pair(t, t + 1 < T - 1); dpp_fence(NJ12T); }

    // condition on the node potential: P = A + diag(J_node); right-hand sides X = [J12 | h_filt]
    // (h_filt = h_pred + h_node lands in lane N, register i <- lane i of the row-layout h_node)
    double P[N], X[N];
    static_for<0, N>([&](auto i) { P[i] = __builtin_fma(Jo, E[i], An[i]); });
    if (last) {
      asm volatile("; last step: no pair potential, G = 0");   // keep this a branch
      static_for<0, N>([&](auto i) { X[i] = EN * An[i]; });
    } else {
      if constexpr (LOWREG) {
        const long o = INHOMOG ? (long)t * N * N : 0;
        static_for<0, N>([&](auto i) { const double r = pJ12[o + i * N + cc]; X[i] = __builtin_fma(EN, An[i], col ? -r : 0.0); });
      } else {
        static_for<0, N>([&](auto i) { X[i] = __builtin_fma(EN, An[i], J12c[i]); });
      }
    }
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(X[i], ho, EN); });
    dpp_fence(P);
    TICK(0)
    if constexpr (FILT) {
      // messages of step t (natural scaling): J_filt = J_pred + diag(J_node) = -1/2 (P - J11_info);
      // h_filt sits in lane N of X, h_pred in lane N of An
      const long oj = INHOMOG ? (long)t * N * N : 0;
      const long mo = ((long)b * T + t) * N * N, vo = ((long)b * T + t) * N;
      static_for<0, N>([&](auto i) {
        const double j11 = last ? 0.0 : pJ11[oj + i * N + cc];
        const double jf = -0.5 * P[i] - j11;
        if (st && a.msg_Jf) a.msg_Jf[mo + i * N + c] = jf;
        if (st && a.msg_Jp) a.msg_Jp[mo + i * N + c] = __builtin_fma(0.5 * Jo, E[i], jf);
        if (valid && c == N && a.msg_hf) a.msg_hf[vo + i] = X[i];
        if (valid && c == N && a.msg_hp) a.msg_hp[vo + i] = An[i];
      });
    }

    // in-place Gauss-Jordan: P -> P^-1, X -> P^-1 X (gauss_jordan above)
    double pv = 0.0;                                   // CHOL: lane k <- pivot k
    double* w2 = CHOL ? ws2b + (long)t * (N * N + N) : nullptr;
    gauss_jordan<N, CHOL>(P, X, E, qacc, pmin, ldM, ldE, pv,
                          [&](auto kk, double r) { if (st) w2[kk * N] = r; });
    if constexpr (CHOL) { if (st) w2[N * N] = pv; }
    TICK(1)

    // hand-off to the backward half: row i of [P^-1 J12 | c] (lanes 0..N) and of P^-1 (lanes < N)
    double* w = wsb + (long)t * WS;
    if (sth) static_for<0, N>([&](auto i) { w[i * HS + c] = X[i]; });
    if (st) static_for<0, N>([&](auto i) { w[N * HS + i * PS + c] = P[i]; });
    TICK(2)

    if (!last) {
      // next pivot block  A' = (J22 + J11) - J12' P^-1 J12  (lanes < N),  h_pred' = -J12' c  (lane N)
      const bool next_last = (t + 1 == T - 1);
      if (!INHOMOG && next_last) {
        asm volatile("; next step is the last: its pivot block has no J11 term");   // keep a branch
        static_for<0, N>([&](auto i) { const double r = pJ22[i * N + cc]; An[i] = col ? -2.0 * r : 0.0; });
      } else {
        static_for<0, N>([&](auto i) { An[i] = Cc[i]; });
      }
      asm volatile("s_nop 1");   // block entry: two wait states before the first DPP read (audit rule)
      static_for<0, (N + IL - 1) / IL>([&](auto g) { rows_src_bcast<IL, g * IL, N, 0>(An, NJ12T, X); });
    }
    TICK(3)
  }

  // ---- log-normaliser --------------------------------------------------------------------------
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
    if (valid && c == 0) a.lognorm[b] = total;
    const bool bad = !(pmin > 0.0) || !(total == total);
    if (bad && valid && c == 0) {   // rare path: keep the smallest failing index (+1); 0 = ok
      int old = *(volatile int32_t*)a.info;
      while (old == 0 || old > b + 1) {
        const int seen = atomicCAS(a.info, old, b + 1);
        if (seen == old) break;
        old = seen;
      }
    }
  }

#ifdef SVAE_PHASE_TIMING
  if (valid && c == 0) { for (int q = 0; q < 4; ++q) a.E_init[(long)b * (N * N + N) + q] = (double)tm[q]; }
  return;
#endif
  if constexpr (FILT) return;
  // ---- backward pass in moment form on homogeneous coordinates ---------------------------------
  // S[i] = row i of S~ (i = 0..N), lane c = column c (c = 0..N).  Start from S~_T := e_N e_N' so
  // that the generic step at t = T-1 (where G = 0, c = mu_{T-1}) yields [[Sigma+mu mu', mu],[mu',1]].
  // With H = [P^-1 J12 | c]:  G~' row k = -H[.][k] (k < N),  row N = c';  lane N: (0,..,0,1).
  double S[N + 1];
  static_for<0, N + 1>([&](auto i) { S[i] = 0.0; });
  S[N] = EN;
  dpp_fence(S);
  double sumA[N], sumW[N], Slast[LOWREG ? 1 : N];   // LOWREG: S~_{T-1} waits in its output slot
  static_for<0, N>([&](auto i) { sumA[i] = 0.0; sumW[i] = 0.0; });
  if constexpr (!LOWREG) static_for<0, N>([&](auto i) { Slast[i] = 0.0; });

  double* oEx = a.E_node_x + ((long)b * T) * N + cc;
  double* oExx = a.E_node_diagxx + ((long)b * T) * N + cc;
  double* oPair = INHOMOG ? a.E_pair + ((long)b * (T - 1)) * 3 * N * N + cc : nullptr;

  // lane c < N reads ROW c of H (-> H[k] = H[c][k], the transposition G -> G') and row c of the
  // symmetric P^-1, both contiguous and 16-byte aligned.
  // lanes >= N read the constant page ([e_N | 0], stride 0) instead: no branch around the loads
  const double* hrow0 = col ? wsb + cc * HS : zpage;
  const double* prow0 = col ? wsb + N * HS + cc * PS : zpage + HS;
  const long tstride = col ? WS : 0;
  const double* hp_ = hrow0 + (long)(T - 1) * tstride;     // walking pointers: steps T-1, T-2, ..
  const double* pp_ = prow0 + (long)(T - 1) * tstride;
  auto load_step = [&](long more, double (&h)[N + 1], double (&pi)[N]) {   // more: another step follows
    load_row<N + 1>(hp_, h);
    load_row<N>(pp_, pi);
    hp_ -= more * tstride;
    pp_ -= more * tstride;
  };

  // one backward step: consumes (H, Pi) of step t, prefetches step t-1 into (Hn, Pin)
  auto step = [&](int tt, double (&H)[N + 1], double (&Pi)[N], double (&Hn)[N + 1], double (&Pin)[N]) {
    const int t = tt < 0 ? -1 - tt : tt;
    if constexpr (!LOWREG) load_step(t > 1 ? 1 : 0, Hn, Pin);    // unconditional prefetch of step t-1 (t = 0: re-reads record 0, unused)
    dpp_fence(H);   // not a DPP source, but keeps the loads' consumers behind this point

    // W~ = S~_{t+1} G~'   (W[i][c] = E[x~_{t+1,i} x~_{t,c}])
    double W[N + 1];
    static_for<0, N + 1>([&](auto i) { W[i] = 0.0; });
    static_for<0, (N + 1 + IL - 1) / IL>([&](auto g) { rows_src_bcast<IL, g * IL, N + 1, N>(W, S, H); });
    if (a.ws3) {   // VJP mode: keep W~_t (rows 0..N, lanes 0..N)
      double* w3 = a.ws3 + ((long)b * T + t) * (N + 1) * HS + c;
      if (sth) static_for<0, N + 1>([&](auto i) { w3[i * HS] = W[i]; });
    }
    // S~_t = G~ W~ + diag(P^-1, 0), computed through its transpose (S~ symmetric):
    //   S~[c][i] = sum_k G~[c][k] W~[k][i] = sum_k G~'[k](lane c) * W[k](lane i)
    static_for<0, N>([&](auto i) { S[i] = Pi[i]; });
    S[N] = 0.0;
    static_for<0, (N + 1 + IL - 1) / IL>([&](auto g) { rows_lane_bcast<IL, g * IL, N + 1, N>(S, W, H); });

    if (INHOMOG) {
      // per-step pair blocks: [E x_t x_t' | E x_t x_{t+1}' | E x_{t+1} x_{t+1}'] for pair index t;
      // S~_t completes pair t (first block) and pair t-1 (third block); W~ (t < T-1) is pair t.
      if (st) {
        if (t < T - 1) {
          double* o = oPair + (long)t * 3 * N * N;
          static_for<0, N>([&](auto i) { o[i * N] = S[i]; });
          double* o2 = a.E_pair + (((long)b * (T - 1) + t) * 3 + 1) * N * N + (long)cc * N;
          static_for<0, N>([&](auto i) { o2[i] = W[i]; });
        }
        if (t > 0) {
          double* o = oPair + ((long)(t - 1) * 3 + 2) * N * N;
          static_for<0, N>([&](auto i) { o[i * N] = S[i]; });
        }
      }
    } else {
      if (t < T - 1) static_for<0, N>([&](auto i) { sumA[i] += S[i]; sumW[i] += W[i]; });
      else {
        if constexpr (LOWREG) { if (st) static_for<0, N>([&](auto i) { a.E_pair[(long)b * 3 * N * N + 2 * N * N + i * N + cc] = S[i]; }); }
        else static_for<0, N>([&](auto i) { Slast[i] = S[i]; });
      }
    }

    // node statistics: E[x_t] = row N of S~, diag E[x_t x_t'] = sum_i E[i] * S[i]
    double dg = 0.0, dg1 = 0.0;
    static_for<0, N>([&](auto i) {
      if constexpr (i % 2 == 0) dg = __builtin_fma(E[i], S[i], dg); else dg1 = __builtin_fma(E[i], S[i], dg1);
    });
    if (st) {
      oEx[(long)t * N] = S[N];
      oExx[(long)t * N] = dg + dg1;
    }
  };

  if constexpr (LOWREG) {
    double Ha[N + 1], Pia[N];
    for (int t = T - 1; t >= 0; --t) {
      load_step(1, Ha, Pia);
      step(-1 - t, Ha, Pia, Ha, Pia);   // negative: "no prefetch" (see step)
    }
  } else {
    double Ha[N + 1], Pia[N], Hb[N + 1], Pib[N];
    load_step(T > 1 ? 1 : 0, Ha, Pia);
    int t = T - 1;
    for (; t >= 1; t -= 2) {          // two steps per trip: the prefetch buffers ping-pong
      step(t, Ha, Pia, Hb, Pib);
      step(t - 1, Hb, Pib, Ha, Pia);
    }
    if (t == 0) step(0, Ha, Pia, Hb, Pib);
  }

  // ---- global statistics -----------------------------------------------------------------------
  if (st) {
    double* ei = a.E_init + (long)b * (N * N + N);
    static_for<0, N>([&](auto i) { ei[i * N + cc] = S[i]; });   // E[x0 x0']
    ei[N * N + cc] = S[N];                                      // E[x0]
    if (!INHOMOG) {
      double* ep = a.E_pair + (long)b * 3 * N * N;
      static_for<0, N>([&](auto i) {
        ep[i * N + cc] = sumA[i];                               // sum_{t<T-1} E[x_t x_t']
        ep[N * N + cc * N + i] = sumW[i];                       // sum_t E[x_t x_{t+1}'] = (sum_t W_t)'
        const double sl = LOWREG ? ep[2 * N * N + i * N + cc] : Slast[LOWREG ? 0 : (int)i];
        ep[2 * N * N + i * N + cc] = (sumA[i] - S[i]) + sl;      // sum_{t>=1} E[x_t x_t']
      });
    }
  }
}

template <int N>
static int launch_estep(const LdsArgs& a, bool inhomog, hipStream_t stream) {
  dim3 grid((a.B + 3) / 4), block(64);
  const bool chol = a.ws2 != nullptr;
  if (inhomog && chol)
    hipLaunchKernelGGL((lds_estep_kernel<N, true, true>), grid, block, 0, stream, a);
  else if (inhomog)
    hipLaunchKernelGGL((lds_estep_kernel<N, true, false>), grid, block, 0, stream, a);
  else if (chol)
    hipLaunchKernelGGL((lds_estep_kernel<N, false, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((lds_estep_kernel<N, false, false>), grid, block, 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <int N>
static int launch_filter(const LdsArgs& a, bool inhomog, hipStream_t stream) {
  dim3 grid((a.B + 3) / 4), block(64);
  if (inhomog)
    hipLaunchKernelGGL((lds_estep_kernel<N, true, true, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((lds_estep_kernel<N, false, true, true>), grid, block, 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load and
// store (s_waitcnt vmcnt(0)), i.e. for the producer's whole prefetch ring and the consumer's output stores, each step
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef double vd2 __attribute__((ext_vector_type(2)));   // (a native vector: stays in registers)
// one pipeline stage of a producer wavefront: a (sequence, step) record pair as 16-byte pieces, one per lane and k
template <int KW, int KA> struct Stage { vd2 w[KW], ad[KA]; };
template <int WP, int AP, int KW, int KA>
__device__ __forceinline__ void prod_issue(Stage<KW, KA>& sg, const vd2* wrec, const vd2* arec, int t, int lane) {
  static_for<0, KW>([&](auto k) {
    const int q = k * 64 + lane;
    sg.w[k] = wrec[(long)t * WP + (q < WP ? q : WP - 1)];
  });
  static_for<0, KA>([&](auto k) {
    const int q = k * 64 + lane;
    sg.ad[k] = arec[(long)t * AP + (q < AP ? q : AP - 1)];
  });
}
template <int WP, int AP, int KW, int KA>
__device__ __forceinline__ void prod_publish(const Stage<KW, KA>& sg, vd2* slot, int lane) {
  static_for<0, KW>([&](auto k) {
    const int q = k * 64 + lane;
    slot[q < WP ? q : WP - 1] = sg.w[k];                  // (clamped lanes rewrite the last pair)
  });
  static_for<0, KA>([&](auto k) {
    const int q = k * 64 + lane;
    slot[WP + (q < AP ? q : AP - 1)] = sg.ad[k];
  });
}

// producer stage of sweep 1: the step's record is GATHERED from several arrays (8-byte pieces, piece k of lane l is
// element k*64 + l of the concatenated record); base / stride / step offset of every piece are set up once
template <int KT> struct GStage { double v[KT]; };
// (every piece is loaded and published every step, also those beyond the record's end -- they re-read one valid
// address: a load count that is not a compile-time constant makes hipcc wait for vmcnt(0), i.e. for the whole prefetch
// ring, before each publish)
template <int KT>
__device__ __forceinline__ void gather_issue(GStage<KT>& sg, const double* const (&base)[KT], const int (&stp)[KT],
                                             const int (&off)[KT], int t, int T) {
  static_for<0, KT>([&](auto k) {
    int tt = t + off[k];
    tt = tt < T ? tt : T - 1;
    sg.v[k] = base[k][(long)tt * stp[k]];
  });
}
template <int KT>
__device__ __forceinline__ void gather_publish(const GStage<KT>& sg, double* slot, int lane) {
  static_for<0, KT>([&](auto k) { slot[k * 64 + lane] = sg.v[k]; });
}

// ---- backward sampler ---------------------------------------------------------------------------
// natural_sample_backward (svae/lds/cython_lds_inference.pyx:310-355; _natural_sample
// cython_gaussian_grads.pxd:431-454, _natural_condition_on :489-508):
//   x_{T-1} ~ N(J_f^-1 h_f, J_f^-1),   x_t | x_{t+1} ~ N(P_t^-1 (h_f,t - J12 x_{t+1}), P_t^-1),
//   noise = chol(P_t)^-T eps_t   (the reference's dtrtrs 'L','T'), so that equal eps give equal samples.
// From the forward pass: mean = c_t - (P^-1 J12) x_{t+1} (H rows in the main workspace) and
// P_t = L D L' (unit factor rows + pivots in the second region): noise = L^-T D^-1/2 eps by back
// substitution.  Layout: one DPP row per sequence as in the E-step, but lanes are SAMPLES (16 per
// pass) and the vector index lives in the register number: every coefficient is then a
// row_newbcast operand, so one v_fmac_f64_dpp advances 16 samples of 4 sequences.
template <int N>
__global__ __launch_bounds__(64) void lds_sample_kernel(const SampleArgs a) {
  constexpr int HS = ws_h_stride(N), WS = ws_step_doubles(N);
  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int T = a.T, S = a.S;
  const double* wsb = a.ws + (long)b * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
  const double* ws2b = a.ws2 + ((long)b * T) * (N * N + N) + cc;

  for (int s0 = 0; s0 < S; s0 += 16) {
    const int s = s0 + c;                      // this lane's sample
    const bool sv = valid && s < S;
    const int ss = s < S ? s : S - 1;
    double Xn[N];
    static_for<0, N>([&](auto k) { Xn[k] = 0.0; });
    double one = 1.0;
    // operands of one step, fetched one step ahead (raw: no arithmetic before their step, and every
    // load unconditional -- lanes >= N read row 0 and are masked when used)
    double Hn[N + 1], Rn[N], Yn[N], pvn;
    auto fetch = [&](int t) {
      const double* w2 = ws2b + (long)t * (N * N + N);
      load_row<N + 1>(wsb + (long)t * WS + cc * HS, Hn);                 // H[j] = H[c][j]
      static_for<0, N>([&](auto k) { Rn[k] = w2[k * N]; });
      pvn = w2[N * N];
      const double* e = a.eps + (((long)b * T + t) * S + ss) * N;
      static_for<0, N>([&](auto k) { Yn[k] = e[k]; });
    };
    fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      double H[N + 1], R[N], Y[N];
      static_for<0, N + 1>([&](auto k) { H[k] = col ? Hn[k] : 0.0; });
      static_for<0, N>([&](auto k) { R[k] = col ? Rn[k] : 0.0; Y[k] = Yn[k]; });
      const double pv = col ? pvn : 1.0;
      fetch(t > 0 ? t - 1 : 0);                // (unconditional: see load_step)
      double dis = rsqrt_nr(pv);               // lane k: D_k^-1/2
      dpp_fence(H);
      dpp_fence(R);
      dpp_fence(dis);
      // y = D^-1/2 eps, then back substitution with the unit upper factor L'
      static_for<0, N>([&](auto k) {
        double acc = 0.0;
        mac_bc<k>(acc, dis, Y[k]);
        Y[k] = acc;
      });
      static_for<1, N>([&](auto jj) {
        constexpr int j = N - jj;              // j = N-1 .. 1
        static_for<0, j>([&](auto k) { mac_bc<j, true>(Y[k], R[k], Y[j]); });
      });
      // x_t = noise + c_t - (P^-1 J12) x_{t+1}:  coefficient [k][j] = H[j] at lane k
      static_for<0, N>([&](auto k) { mac_bc<k>(Y[k], H[N], one); });
      static_for<0, N>([&](auto j) {
        static_for<0, N>([&](auto k) { mac_bc<k, true>(Y[k], H[j], Xn[j]); });
      });
      if (sv) {
        double* o = a.samples + (((long)b * T + t) * S + s) * N;
        static_for<0, N>([&](auto k) { o[k] = Y[k]; });
      }
      static_for<0, N>([&](auto k) { Xn[k] = Y[k]; });
    }
  }
}

// Few samples per sequence (S <= 4, the training step's S = 1): lanes are VECTOR COMPONENTS instead (lane c holds
// x[c]), one register per sample.  Per step and sample: z = D^-1/2 eps locally; the back substitution with L'
// column by column (y_i final -> every lane j < i subtracts L[i][j] y_i: one fenced DPP FMA per column, the
// factor read transposed by the addressing: lane c loads row c of the stored factor block = column c of L);
// the mean term with x_{t+1}[j] as the broadcast operand.  ~25 instructions per step and sample instead of
// ~165 for a 16-sample pass whose lanes would be mostly idle.
template <int N>
__global__ __launch_bounds__(64) void lds_sample_vec_kernel(const SampleArgs a) {
  constexpr int HS = ws_h_stride(N), WS = ws_step_doubles(N);
  constexpr int SMAX = 4;
  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int T = a.T, S = a.S;
  const double* wsb = a.ws + (long)b * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
  const double* ws2b = a.ws2 + ((long)b * T) * (N * N + N);
  const bool st = valid && col;

  double X[SMAX];                              // x_{t+1}[c] per sample
  static_for<0, SMAX>([&](auto s) { X[s] = 0.0; });
  // Operands are fetched NST steps ahead into a ring of register stages: the records were written by the
  // E-step ~T steps earlier and come from HBM / the memory-side cache, further away than one step of arithmetic
  // (one stage ahead: 0.76 us per step, the load latency; the arithmetic is ~0.25 us).
  constexpr int NST = N <= 10 ? 3 : 2;
  struct Stage { double H[N + 1], L[N], pv, E[SMAX]; };
  Stage ring[NST];
  auto fetch = [&](Stage& g, int t) {
    const double* w2 = ws2b + (long)t * (N * N + N);
    load_row<N + 1>(wsb + (long)t * WS + cc * HS, g.H);                  // H[j] = [P^-1 J12 | c][c][j]
    static_for<0, N>([&](auto i) { g.L[i] = w2[cc * N + i]; });          // factor block row c: lane i of register c = L[i][c]
    g.pv = w2[N * N + cc];
    static_for<0, SMAX>([&](auto s) { g.E[s] = a.eps[(((long)b * T + t) * S + (s < S ? s : S - 1)) * N + cc]; });
  };
  static_for<0, NST>([&](auto q) { fetch(ring[q], T - 1 - q > 0 ? T - 1 - q : 0); });
  for (int t0 = T - 1; t0 >= 0; t0 -= NST) {
    static_for<0, NST>([&](auto q) {
      const int t = t0 - q;
      if (t >= 0) {
        double H[N + 1], Lc[N], Y[SMAX];
        static_for<0, N + 1>([&](auto k) { H[k] = col ? ring[q].H[k] : 0.0; });
        static_for<0, N>([&](auto i) { Lc[i] = (c < i && col) ? ring[q].L[i] : 0.0; });   // L[i][c] below the diagonal only
        const double dis = rsqrt_nr(col ? ring[q].pv : 1.0);
        static_for<0, SMAX>([&](auto s) { Y[s] = col ? dis * ring[q].E[s] : 0.0; });
        fetch(ring[q], t - NST > 0 ? t - NST : 0);
        dpp_fence(X);
        dpp_fence(Y);
        static_for<0, SMAX>([&](auto s) {
          if (s < S) {
            // y = L^-T z: columns N-1 .. 1
            static_for<1, N>([&](auto jj) {
              constexpr int i = N - jj;
              mac_bc<i, true, true>(Y[s], Y[s], Lc[i]);
            });
            // x_t = y + c_t - (P^-1 J12) x_{t+1}
            double acc0 = Y[s] + H[N], acc1 = 0.0;
            static_for<0, N>([&](auto j) {
              if constexpr (j % 2 == 0) mac_bc<j, true>(acc0, X[s], H[j]); else mac_bc<j, true>(acc1, X[s], H[j]);
            });
            const double xt = acc0 + acc1;
            if (st) a.samples[(((long)b * T + t) * S + s) * N + c] = xt;
            X[s] = xt;
          }
        });
      }
    });
  }
}

// The same sampler with PRODUCER wavefronts (small batches): at ~0.2 us of arithmetic per step a three-stage register
// ring cannot cover ~2 us of HBM latency (0.67 us per step measured).  Four more wavefronts of the workgroup, one per
// sequence, keep SAMPLE_PD steps of the sequence's records in flight in their own registers (H rows, LDL' factor,
// eps: gathered as 8-byte pieces) and publish the oldest into a two-slot LDS ring each step; the consumer reads LDS
// only.  One s_waitcnt lgkmcnt(0) + s_barrier per step; every wavefront executes T barriers.
constexpr int SAMPLE_PD = 8;
template <int N>
__global__ __launch_bounds__(320) void lds_sample_vec_prod_kernel(const SampleArgs a) {
  constexpr int HS = ws_h_stride(N), WS = ws_step_doubles(N);
  constexpr int SMAX = 4, PD = SAMPLE_PD;
  constexpr int R1 = N * HS + N * N + N;
  // record in a ring slot: [H rows | LDL' factor | pivots | eps (SMAX x N) | zeros (N + 1)]: the consumer's idle lanes
  // (c >= N) read the zero row instead of masking every operand (2 v_cndmask per double: ~40 of ~145 instructions per step)
  constexpr int ZO = R1 + SMAX * N;
  static_assert(ZO % 2 == 0, "the zero row is read as 16-byte pairs");
  constexpr int KT = (ZO + N + 1 + 63) / 64, REC = KT * 64, SLOT = 4 * REC;
  __shared__ double ring[2 * SLOT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int T = a.T, S = a.S, SN = S * N;
  if (wv >= 1) {
    // ---- producer wavefronts ---------------------------------------------------------------------------------------
    const int r = wv - 1;
    const int br = blockIdx.x * 4 + r;
    const long bb = br < a.B ? br : a.B - 1;
    const double* wsq = a.ws + bb * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
    const double* base[KT];
    int stp[KT], off[KT];
    // (pieces outside the three segments -- the unused sample rows and the zero row -- re-read a 0.0 of the sequence's
    //  constant page, which every forward kernel that keeps the factor writes: [e_n (HS) | zeros])
    const double* zero = wsq - ws_zpage_doubles(N) + HS;
    static_for<0, KT>([&](auto k) { base[k] = zero; stp[k] = 0; off[k] = 0; });
    int start = 0;
    auto seg = [&](const double* p, int len, int stride) {
      static_for<0, KT>([&](auto k) {
        const int f = k * 64 + lane - start;
        if (f >= 0 && f < len) { base[k] = p + f; stp[k] = stride; }
      });
      start += len;
    };
    seg(wsq, N * HS, WS);
    seg(a.ws2 + bb * T * (N * N + N), N * N + N, N * N + N);
    seg(a.eps + bb * T * SN, SN, SN);
    GStage<KT> s0, s1, s2, s3, s4, s5, s6, s7;           // stage of step t: (T - 1 - t) % 8
    static_assert(PD == 8, "eight named stages");
    double* slot0 = ring + r * REC;
    auto rec = [&](int t) { return t > 0 ? t : 0; };
    gather_issue<KT>(s0, base, stp, off, rec(T - 1), T);
    gather_issue<KT>(s1, base, stp, off, rec(T - 2), T);
    gather_issue<KT>(s2, base, stp, off, rec(T - 3), T);
    gather_issue<KT>(s3, base, stp, off, rec(T - 4), T);
    gather_issue<KT>(s4, base, stp, off, rec(T - 5), T);
    gather_issue<KT>(s5, base, stp, off, rec(T - 6), T);
    gather_issue<KT>(s6, base, stp, off, rec(T - 7), T);
    gather_issue<KT>(s7, base, stp, off, rec(T - 8), T);
    gather_publish<KT>(s0, slot0 + ((T - 1) & 1) * SLOT, lane);
    gather_issue<KT>(s0, base, stp, off, rec(T - 9), T);
    lds_barrier();                                       // barrier 0: step T-1 is in its slot
    // consumer iteration t reads slot t%2: publish step t-1 into the other, refill the stage with step t-1-PD
    // (no branch inside the steady-state loop: across one, hipcc's wait counts degrade to vmcnt(0))
#define SVAE_PROD_STEP(sg, t)                                                 \
    {                                                                           \
      gather_publish<KT>(sg, slot0 + (((t) - 1) & 1) * SLOT, lane);             \
      gather_issue<KT>(sg, base, stp, off, rec((t) - 1 - PD), T);               \
      lds_barrier();                                                            \
    }
    int t0 = T - 1;
    for (; t0 >= 8; t0 -= PD) {
      SVAE_PROD_STEP(s1, t0)
      SVAE_PROD_STEP(s2, t0 - 1)
      SVAE_PROD_STEP(s3, t0 - 2)
      SVAE_PROD_STEP(s4, t0 - 3)
      SVAE_PROD_STEP(s5, t0 - 4)
      SVAE_PROD_STEP(s6, t0 - 5)
      SVAE_PROD_STEP(s7, t0 - 6)
      SVAE_PROD_STEP(s0, t0 - 7)
    }
    if (t0 >= 1) SVAE_PROD_STEP(s1, t0)
    if (t0 >= 2) SVAE_PROD_STEP(s2, t0 - 1)
    if (t0 >= 3) SVAE_PROD_STEP(s3, t0 - 2)
    if (t0 >= 4) SVAE_PROD_STEP(s4, t0 - 3)
    if (t0 >= 5) SVAE_PROD_STEP(s5, t0 - 4)
    if (t0 >= 6) SVAE_PROD_STEP(s6, t0 - 5)
    if (t0 >= 7) SVAE_PROD_STEP(s7, t0 - 6)
#undef SVAE_PROD_STEP
    return;
  }
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N;
  const int cc = col ? c : 0;
  const bool st = valid && col;
  const double* ringrow = ring + (lane >> 4) * REC;
  double X[SMAX];                              // x_{t+1}[c] per sample
  static_for<0, SMAX>([&](auto s) { X[s] = 0.0; });
  const int hoff = col ? c * HS : ZO;          // this lane's row of H, or the zero row
  double mL[N];                                // L[i][c] below the diagonal only: the other entries of the stored block are
  static_for<0, N>([&](auto i) { mL[i] = (c < i && col) ? 1.0 : 0.0; });   // finite by-products of the elimination
  const double cm = col ? 1.0 : 0.0;
  for (int t = T - 1; t >= 0; --t) {
    lds_barrier();                             // step t is in slot t%2
    const double* rec = ringrow + (t & 1) * SLOT;
    double H[N + 1], Lc[N], Y[SMAX], E[SMAX];
    load_row<N + 1>(rec + hoff, H);                                          // H[j] = [P^-1 J12 | c][c][j]
    static_for<0, N>([&](auto i) { Lc[i] = rec[N * HS + cc * N + i]; });     // lane c of register i = L[i][c]
    const double pv = rec[N * HS + N * N + cc];
    static_for<0, SMAX>([&](auto s) { E[s] = rec[R1 + (s < S ? s : S - 1) * N + cc]; });
    static_for<0, N>([&](auto i) { Lc[i] *= mL[i]; });
    const double dis = rsqrt_nr(col ? pv : 1.0) * cm;
    static_for<0, SMAX>([&](auto s) { Y[s] = dis * E[s]; });
    dpp_fence(X);
    dpp_fence(Y);
    static_for<0, SMAX>([&](auto s) {
      if (s < S) {
        static_for<1, N>([&](auto jj) {                                      // y = L^-T z: columns N-1 .. 1
          constexpr int i = N - jj;
          mac_bc<i, true, true>(Y[s], Y[s], Lc[i]);
        });
        double acc0 = Y[s] + H[N], acc1 = 0.0;                               // x_t = y + c_t - (P^-1 J12) x_{t+1}
        static_for<0, N>([&](auto j) {
          if constexpr (j % 2 == 0) mac_bc<j, true>(acc0, X[s], H[j]); else mac_bc<j, true>(acc1, X[s], H[j]);
        });
        const double xt = acc0 + acc1;
        if (st) a.samples[(((long)b * T + t) * S + s) * N + c] = xt;
        X[s] = xt;
      }
    });
  }
}

template <int N>
static int launch_sample(const SampleArgs& a, hipStream_t stream) {
  if (a.S <= 4 && a.B <= a.prod_max_b) hipLaunchKernelGGL((lds_sample_vec_prod_kernel<N>), dim3((a.B + 3) / 4), dim3(320), 0, stream, a);
  else if (a.S <= 4) hipLaunchKernelGGL((lds_sample_vec_kernel<N>), dim3((a.B + 3) / 4), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((lds_sample_kernel<N>), dim3((a.B + 3) / 4), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace svae
