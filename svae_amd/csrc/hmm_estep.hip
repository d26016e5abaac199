// hmm_estep.hip -- batched HMM E-step (forward-backward + expected statistics) for MI355X (gfx950).
//
// What it replaces (reference = mattjj/svae, /root/reference):
//   hmm_logZ        svae/hmm/cython_hmm_inference.pyx:93-121   (log-space forward pass)
//   hmm_logZ_grad   svae/hmm/cython_hmm_inference.pyx:126-166  (its reverse pass at g = 1 IS the
//                   E-step: expected initial state, transition counts, state marginals;
//                   `hmm_estep_slow = vgrad(hmm_logZ)`, svae/hmm/hmm_inference.py:65)
//   hmm_estep       svae/hmm/hmm_inference.py:21-41, which delegates to the un-vendored pyhsmm
//                   messages (parity pinned on hmm_logZ / hmm_logZ_grad).
// Used by the SLDS-SVAE coordinate ascent (svae/models/slds_svae.py:108-115, 159-175).
//
// Mapping: one DPP row (16 lanes) per sequence, 4 sequences per wavefront, lane k = discrete state k
// (K <= 16).  The matrix-vector products alpha' P and P (e . beta) are row_newbcast FMAs like the
// LDS kernels.  Scaled (not log-space) recursions: per step the node log-potentials are shifted by
// their maximum and exponentiated ONCE per lane (the reference does K logsumexp's of K terms per
// step), alpha is renormalised to sum 1, and log Z accumulates the scales as mantissa/exponent
// pairs (one log per sequence).  Forward quantities needed by the backward pass (alpha_t, e_t/c_t)
// go through a caller-owned workspace of HMM_WS doubles per (sequence, step).
// Dynamic range: a scaled step underflows when every path into the states the next observation allows
// goes through transition potentials ~700 nats below the matrix' maximum (exp underflows to 0 where the
// reference's log-space pass keeps e^-800).  Such a step (normaliser c_t below 1e-200: rare) is redone
// in LOG SPACE for its sequence -- K log-sum-exps like the reference -- and flagged in the workspace; the
// backward pass treats flagged steps in log space too: no NaN, and the reference's value whenever the
// paths that matter at a flagged step carried more than 1e-300 of the mass one step earlier.  (Components
// of alpha below that are flushed to zero by the scaled steps in between -- only an all-log-space pass,
// 4x the work, would keep them.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "hmm_args.hpp"

#ifndef SVAE_HMM_TWOEND
#define SVAE_HMM_TWOEND 1      // two-ended kernel + one-directional fallback for flagged sequences (0: the one-directional kernel alone; A/B)
#endif

namespace svae {

// workspace record per (sequence, step).  One-directional kernel (hmm_estep_kernel): [alpha (16) | e/c or its log-space
// stand-in (16) | flag | ..].  Two-ended kernel (hmm_estep2_kernel): [alpha^ | e | w = e o beta^] in slots of 8 (K <= 8) or
// 16 lanes, then the maximum of the node potentials; entry HMM_REDO of the sequence's FIRST record is its REDO flag.
constexpr int HMM_WS = 50;
constexpr int HMM_REDO = 49;
constexpr double HMM_TINY = 1e-200;

// Maximum over the 16 lanes of a DPP row, in every lane: four rotate-and-max steps (row_ror:8/4/2/1 on the two halves
// of the double; compiler-scheduled, hazards padded by hipcc) instead of K broadcasts.
template <int R>
__device__ __forceinline__ double row_ror(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + R, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + R, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_max16(double x) {
  x = __builtin_fmax(x, row_ror<8>(x));
  x = __builtin_fmax(x, row_ror<4>(x));
  x = __builtin_fmax(x, row_ror<2>(x));
  x = __builtin_fmax(x, row_ror<1>(x));
  return x;
}

template <int K, bool FUSED = false>
__global__ __launch_bounds__(64) void hmm_estep_kernel(const HmmArgs a) {
  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const int bslot = brow < a.B ? brow : a.B - 1;
  // (indexed launches: a negative entry marks an unused slot -- the list is compacted, so a wavefront's row 0 is live
  //  whenever any of its rows is; idle rows repeat it)
  const int braw = a.seq_index ? a.seq_index[bslot] : bslot;
  const bool valid = brow < a.B && braw >= 0;
  if (!__any(valid)) return;
  const int b = valid ? braw : __shfl(braw, 0);
  const bool col = c < K;
  const int cc = col ? c : 0;
  const int T = a.T;
  const double NEG_BIG = -1.0e300;
  if (a.redo_only) {
    // fallback pass behind the two-ended kernel: only for sequences it flagged (a step whose normaliser underflowed)
    const double redo = a.ws[((long)b * T) * HMM_WS + HMM_REDO];
    if (!__any(redo != 0.0)) return;
  }

  // transition matrix in both layouts, shifted by its maximum (the shift goes into logZ)
  const double* pp = a.pair_params + (long)b * a.pair_stride;
  double lp[K], lpT[K];
  static_for<0, K>([&](auto j) {
    const double v = pp[j * K + cc], vt = pp[cc * K + j];
    lp[j] = col ? v : NEG_BIG;            // lp[j][c] = log P[j][c]
    lpT[j] = col ? vt : NEG_BIG;          // lpT[k][c] = log P[c][k]
  });
  double pmax = NEG_BIG;
  static_for<0, K>([&](auto j) { pmax = fmax(pmax, lp[j]); });
  static_for<0, 4>([&](auto s) { pmax = fmax(pmax, __shfl_xor(pmax, 1 << s, 16)); });
  double P[K], PT[K];
  static_for<0, K>([&](auto j) { P[j] = col ? exp(lp[j] - pmax) : 0.0; PT[j] = col ? exp(lpT[j] - pmax) : 0.0; });

  const double* node = FUSED ? nullptr : a.node_params + ((long)b * T) * K + cc;
  // FUSED: node potential of step t >= 1 = pc[t-1][0] + pc[t][1] + lz (two loads instead of one)
  const double* pc = FUSED ? a.pair_contr + ((long)b * T) * 2 * K + cc : nullptr;
  const double lzc = FUSED ? a.lz[cc] : 0.0;
  // (callers clamp t to [1, T-1]; with T = 1 the second FUSED read is clamped into the row as well: the value is unused)
  auto node_at = [&](int t) -> double {
    if constexpr (FUSED) return (pc[(long)(t - 1) * 2 * K] + pc[(long)(t < T ? t : T - 1) * 2 * K + K]) + lzc;
    else return node[(long)(t < T ? t : T - 1) * K];
  };
  double node0;
  if constexpr (FUSED) {
    // <E x0 x0', J_c> + <E x0, h_c> + cinit_c: lane c = state (once per sequence)
    const int n = a.n;
    const double* ei = a.lds_E_init + (long)b * (n * n + n);
    const double* Jc = a.init_J + (long)cc * n * n;
    const double* hc = a.init_h + (long)cc * n;
    double s0 = 0.0;
    for (int q = 0; q < n * n; ++q) s0 = __builtin_fma(ei[q], Jc[q], s0);
    double s1 = 0.0;
    for (int q = 0; q < n; ++q) s1 = __builtin_fma(ei[n * n + q], hc[q], s1);
    node0 = (s0 + s1) + a.cinit[cc];
  } else {
    node0 = node[0];
  }
  double* nout = a.node_out ? a.node_out + ((long)b * T) * K + cc : nullptr;
  double* wsb = a.ws + ((long)b * T) * HMM_WS + c;
  double* wflag = a.ws + ((long)b * T) * HMM_WS + 32;
  double one = 1.0;

  // ---- forward ------------------------------------------------------------------------------------
  double lzM = 1.0;            // product of scales (mantissa) ...
  long lzE = 0;                // ... exponent
  double lzS = 0.0;            // sum of the subtracted maxima
  double alpha = 0.0;
  double nd_n = node0;
  for (int t = 0; t < T; ++t) {
    double nd = col ? nd_n : NEG_BIG;
    if (nout && valid && col) nout[(long)t * K] = nd_n;
    // next step's potentials, UNCONDITIONALLY (clamped): a load inside `if (t + 1 < T)` is waited for at the end of
    // its block, i.e. every step stalled for the full memory latency (0.55 -> ... us per step)
    nd_n = node_at(t + 1 < T ? t + 1 : (T > 1 ? T - 1 : 1));
    if (t == 0) nd += col ? a.init_params[cc] : 0.0;
    const double m = row_max16(nd);         // m = max_k node[k]
    const double e = col ? exp_nonpos(nd - m) : 0.0;      // (nd - m <= 0: m is the row maximum)
    double pred;
    if (t == 0) {
      pred = col ? 1.0 : 0.0;
    } else {
      pred = 0.0;
      dpp_fence(alpha);
      static_for<0, K>([&](auto j) { mac_bc<j>(pred, alpha, P[j]); });   // sum_j alpha[j] P[j][k]
    }
    double al = pred * e;
    double cs = 0.0;
    dpp_fence(al);
    static_for<0, K>([&](auto k) { mac_bc<k>(cs, al, one); });            // c_t = sum_k
    double rc = rcp_nr(cs);                 // (v_rcp_f64 + two Newton steps: no fp64 divide on the serial chain)
    double u_st = e * rc, shift = m + (t > 0 ? pmax : 0.0), flag = 0.0;
    const bool tiny = !(cs > HMM_TINY);
    if (__any(tiny)) {
      // log-space redo of this step for the rows that underflowed (wave-uniform branch; rows that did
      // not underflow keep their scaled result)
      const double la = alpha > 0.0 ? ::log(alpha) : NEG_BIG;       // alpha_{t-1}
      double lpred = 0.0;
      if (t > 0) {
        double sj[K], m2 = NEG_BIG;
        static_for<0, K>([&](auto j) {
          const double lpj = col ? pp[j * K + cc] : NEG_BIG;        // log P[j][c], reloaded (rare path)
          sj[j] = bcast<j>(la) + lpj;
          m2 = fmax(m2, sj[j]);
        });
        double sum = 0.0;
        static_for<0, K>([&](auto j) { sum += exp(sj[j] - m2); });
        lpred = m2 + ::log(sum);
      }
      const double lal = col ? lpred + nd : NEG_BIG;
      double M = NEG_BIG;
      static_for<0, K>([&](auto k) { M = fmax(M, bcast<k>(lal)); });
      const double al2 = col ? exp(lal - M) : 0.0;
      double cs2 = 0.0;
      static_for<0, K>([&](auto k) { cs2 += bcast<k>(al2); });
      if (tiny) {
        cs = cs2;
        rc = rcp_nr(cs2);
        al = al2;
        shift = M;
        flag = 1.0;
        u_st = nd - M - ::log(cs2);          // log of (likelihood / normaliser): the backward pass adds log P
      }
    }
    alpha = al * rc;
    if (valid) { wsb[(long)t * HMM_WS] = alpha; wsb[(long)t * HMM_WS + 16] = u_st; }
    if (valid && c == 0) wflag[(long)t * HMM_WS] = flag;
    lzM *= __builtin_amdgcn_frexp_mant(cs);
    lzE += __builtin_amdgcn_frexp_exp(cs);
    lzS += shift;
    if ((t & 15) == 15) { lzE += __builtin_amdgcn_frexp_exp(lzM); lzM = __builtin_amdgcn_frexp_mant(lzM); }
  }
  if (valid && c == 0) a.logZ[b] = lzS + ::log(lzM) + (double)lzE * 0.6931471805599453094;

  // ---- backward + statistics ----------------------------------------------------------------------
  double beta = col ? 1.0 : 0.0;
  double acc[K];                       // xi sums without the transition factor (scaled steps)
  static_for<0, K>([&](auto j) { acc[j] = 0.0; });
  double accS[K];                      // xi sums of the log-space steps (complete terms)
  static_for<0, K>([&](auto j) { accS[j] = 0.0; });
  bool any_slow = false;
  double* oS = a.E_states + ((long)b * T) * K + cc;
  {
    const double gam = alpha * beta;      // t = T-1
    if (valid && col) oS[(long)(T - 1) * K] = gam;
    if (T == 1 && valid && col) a.E_init[(long)b * K + c] = gam;
  }
  // (u, alpha) of a step are fetched one step ahead: the loop is serial in t and the forward pass
  // wrote them ~T steps ago
  double u_n = T > 1 ? wsb[(long)(T - 1) * HMM_WS + 16] : 0.0, al_n = T > 1 ? wsb[(long)(T - 2) * HMM_WS] : 0.0;
  double f_n = T > 1 ? wflag[(long)(T - 1) * HMM_WS] : 0.0;
  for (int t = T - 2; t >= 0; --t) {
    const double u = u_n;                              // e_{t+1} / c_{t+1}  (flagged step: its log-space stand-in)
    const double al = al_n;
    const bool slow = f_n != 0.0;
    {
      const int tp = t > 0 ? t - 1 : 0;                // unconditional (clamped) prefetch of step t-1
      u_n = wsb[(long)(tp + 1) * HMM_WS + 16];
      al_n = wsb[(long)tp * HMM_WS];
      f_n = wflag[(long)(tp + 1) * HMM_WS];
    }
    if (__any(slow)) {
      // step t+1 was redone in log space: beta_t[j] = sum_k exp(log P[j][k] + ul[k] + log beta[k]),
      // xi_t[j][k] = alpha_t[j] * that term  (K exponentials per lane; rows not flagged take the scaled
      // formulas below through the selects)
      any_slow = true;
      const double lw = (col && beta > 0.0) ? u + ::log(beta) : NEG_BIG;     // lane k
      double bn2 = 0.0, term[K];
      static_for<0, K>([&](auto k) {
        const double lpk = col ? pp[cc * K + k] : NEG_BIG;                   // log P[c][k]
        // (clamped: a state the forward pass excludes may have a future e^800 times likelier than the
        //  states that carry the mass; its beta only ever multiplies an alpha or a likelihood of exactly 0)
        const double x = lpk + bcast<k>(lw);
        term[k] = x > -745.0 ? fmin(exp(fmin(x, 700.0)), 1e300) : 0.0;
        bn2 += term[k];                                                      // lane j = c: sum over k
      });
      // xi: lane k of accS[j] += alpha_t[j] * term_{lane j}[k]: transpose through broadcasts
      if (slow) {
        static_for<0, K>([&](auto j) {
          // value at (j, k) lives in lane j, register k; lane k needs it: K x K broadcasts (rare path)
          double row = 0.0;
          static_for<0, K>([&](auto k) {
            const double v = bcast<j>(term[k]);       // (unconditional: the source lane must be active)
            row = (c == k) ? v : row;
          });
          accS[j] += bcast<j>(al) * row;
        });
      }
      double w0 = slow ? 0.0 : u * beta, al0 = slow ? 0.0 : al;
      dpp_fence(w0);
      dpp_fence(al0);
      double bn = 0.0;
      static_for<0, K>([&](auto k) { mac_bc<k>(bn, w0, PT[k]); });
      static_for<0, K>([&](auto j) { mac_bc<j>(acc[j], al0, w0); });
      beta = slow ? fmin(bn2, 1e300) : bn;
      const double gam = al * beta;
      if (valid && col) oS[(long)t * K] = gam;
      if (t == 0 && valid && col) a.E_init[(long)b * K + c] = gam;
      continue;
    }
    double w = u * beta;
    double al_f = al;
    dpp_fence(w);
    dpp_fence(al_f);
    double bn = 0.0;
    static_for<0, K>([&](auto k) { mac_bc<k>(bn, w, PT[k]); });            // beta_t[j] = sum_k P[j][k] w[k]
    static_for<0, K>([&](auto j) { mac_bc<j>(acc[j], al_f, w); });         // xi sums (without P)
    beta = bn;
    const double gam = al * beta;
    if (valid && col) oS[(long)t * K] = gam;
    if (t == 0 && valid && col) a.E_init[(long)b * K + c] = gam;
  }
  if (valid && col) {
    static_for<0, K>([&](auto j) { a.E_trans[(long)b * K * K + j * K + c] = __builtin_fma(acc[j], P[j], accS[j]); });
    (void)any_slow;
  }
}

// ---- two-ended scaled forward-backward (round 4) ----------------------------------------------------------------------
// The one-directional kernel above runs 2 T dependent steps of ~190 / ~150 instructions on ONE wavefront per four
// sequences: 0.35 - 0.6 ms at T = 500 whatever the batch (latency-bound: the SLDS ascent launches it a dozen times per
// step).  Most of a step does not depend on the recursion, and alpha and beta do not depend on each other:
//   phase 1 (parallel in t, both wavefronts of the workgroup, half of the steps each): node potentials (FUSED: built from
//            the LDS kernel's contractions), their row maximum m_t, e_t = exp(node_t - m_t)            -> workspace
//   phase 2 (serial, concurrent): wavefront 0  alpha^_t = normalise((alpha^_{t-1} P) o e_t), log Z     (~55 instructions)
//                                  wavefront 1  beta^_t = normalise(P w_{t+1}),  w_t = e_t o beta^_t   (~50 instructions)
//            -- each with its OWN scaling: gamma and xi are normalised per step in phase 3, so the scalings cancel
//   phase 3 (parallel in t, half each): q = P w_{t+1},  Z_t = <alpha^_t, q>,  gamma_t = alpha^_t o q / Z_t,
//            xi sums  acc[j][k] += alpha^_t[j] w_{t+1}[k] / Z_t  (E_trans = P o acc), summed over the two wavefronts in LDS.
// Every phase streams its operands through a register ring HMM2_D steps deep (a serial step is ~0.15 us of arithmetic
// against 1 - 2 us of memory latency); no branch inside the steady-state loops (hipcc's wait counts degrade to
// vmcnt(0) across one).  The records are compact -- [alpha^ | e | w] in slots of 8 lanes for K <= 8 -- inside the common
// record stride: half the traffic of full-width slots at 2048 sequences.
// A normaliser below HMM_TINY anywhere (the one-directional kernel's log-space case) raises the sequence's REDO flag:
// the one-directional kernel is launched behind this one with redo_only = 1 and recomputes the flagged wavefronts,
// log-space steps and all (its wavefronts exit at once otherwise).  Same mapping: one DPP row per sequence, lane = state.
constexpr int HMM2_D = 8;
#ifndef SVAE_HMM2_NORM
#define SVAE_HMM2_NORM 4
#endif
constexpr int HMM2_NORM = SVAE_HMM2_NORM;     // renormalise the forward / backward messages every HMM2_NORM-th step (1: every step)
static_assert(HMM2_D % HMM2_NORM == 0, "ring positions fix the renormalisation steps");
constexpr int HMM2_WAVES = 4;     // wavefronts per workgroup: all of them share phases 1 and 3, two run the recursions
template <int K, bool FUSED>
__global__ __launch_bounds__(64 * HMM2_WAVES) void hmm_estep2_kernel(const HmmArgs a) {
  constexpr int D = HMM2_D, NW = HMM2_WAVES;
  constexpr int KS = K <= 8 ? 8 : 16;                    // slot width
  constexpr int OA = 0, OE = KS, OW = 2 * KS, OM = 3 * KS;
  static_assert(OM < HMM_REDO, "record layout");
  __shared__ double xacc[(NW - 1) * 16 * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const int bslot = brow < a.B ? brow : a.B - 1;
  const int braw = a.seq_index ? a.seq_index[bslot] : bslot;         // negative: unused slot (see hmm_estep_kernel)
  const bool valid = brow < a.B && braw >= 0;
  if (!__any(valid)) return;                                         // (the same rows in every wavefront of the workgroup)
  const int b = valid ? braw : __shfl(braw, 0);
  const bool col = c < K;
  const int cc = col ? c : 0;
  const int T = a.T;
  const double NEG_BIG = -1.0e300;
  const bool st = valid && col;

  const double* pp = a.pair_params + (long)b * a.pair_stride;
  double lp[K], lpT[K];
  static_for<0, K>([&](auto j) {
    const double v = pp[j * K + cc], vt = pp[cc * K + j];
    lp[j] = col ? v : NEG_BIG;
    lpT[j] = col ? vt : NEG_BIG;
  });
  double pmax = NEG_BIG;
  static_for<0, K>([&](auto j) { pmax = fmax(pmax, lp[j]); });
  static_for<0, 4>([&](auto s) { pmax = fmax(pmax, __shfl_xor(pmax, 1 << s, 16)); });
  double P[K], PT[K];        // P[j][lane c] = P[j][c];  PT[k][lane c] = P[c][k]   (shifted by the matrix' maximum; 0 in idle lanes)
  static_for<0, K>([&](auto j) { P[j] = col ? exp(lp[j] - pmax) : 0.0; PT[j] = col ? exp(lpT[j] - pmax) : 0.0; });

  // this lane's column of the records (idle lanes alias lane 0: they only ever load, and what they load is multiplied
  // by the zero columns of P / PT), and the records' scalars
  double* wsb = a.ws + ((long)b * T) * HMM_WS + cc;
  double* wsr = a.ws + ((long)b * T) * HMM_WS;
  double one = 1.0;
  bool bad = false;
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };

  // ---- phase 1: e_t, m_t (and node_out) for this wavefront's half of the steps ---------------------------------------------
  {
    const double* node = FUSED ? nullptr : a.node_params + ((long)b * T) * K + cc;
    const double* pc = FUSED ? a.pair_contr + ((long)b * T) * 2 * K + cc : nullptr;
    const double lzc = FUSED ? a.lz[cc] : 0.0;
    double* nout = a.node_out ? a.node_out + ((long)b * T) * K + cc : nullptr;
    auto emit = [&](int t, double ndraw, double extra) {
      if (nout && st) nout[(long)t * K] = ndraw;
      const double nd = col ? ndraw + extra : NEG_BIG;
      const double m = row_max16(nd);
      const double e = col ? exp_nonpos(nd - m) : 0.0;
      if (st) wsb[(long)t * HMM_WS + OE] = e;
      if (valid && c == 0) wsr[(long)t * HMM_WS + OM] = m;
    };
    int p0 = (int)((long)T * wv / NW);
    const int p1 = (int)((long)T * (wv + 1) / NW);
    if (wv == 0) {
      if (valid && c == 0) wsr[HMM_REDO] = 0.0;
      double nd0;
      if constexpr (FUSED) {
        // <E x0 x0', J_c> + <E x0, h_c> + cinit_c: lane c = state (once per sequence)
        const int n = a.n;
        const double* ei = a.lds_E_init + (long)b * (n * n + n);
        const double* Jc = a.init_J + (long)cc * n * n;
        const double* hc = a.init_h + (long)cc * n;
        double s0 = 0.0;
        for (int q = 0; q < n * n; ++q) s0 = __builtin_fma(ei[q], Jc[q], s0);
        double s1 = 0.0;
        for (int q = 0; q < n; ++q) s1 = __builtin_fma(ei[n * n + q], hc[q], s1);
        nd0 = (s0 + s1) + a.cinit[cc];
      } else {
        nd0 = node[0];
      }
      emit(0, nd0, a.init_params[cc]);
    }
    if (p0 == 0) p0 = 1;                                 // (step 0 is wavefront 0's, above; T < NW: another range may start at 0)
    // steps p0 .. p1-1 (all >= 1), potentials requested D steps ahead
    auto node_at = [&](int t, double& x0, double& x1) {
      if constexpr (FUSED) { x0 = pc[(long)(t - 1) * 2 * K]; x1 = pc[(long)t * 2 * K + K]; }
      else { x0 = node[(long)t * K]; x1 = 0.0; }
    };
    if (p0 < p1) {
      double r0[D], r1[D];
      static_for<0, D>([&](auto u) { node_at(clampi(p0 + u, 1, p1 - 1), r0[u], r1[u]); });
      int t = p0;
      for (; t + D <= p1; t += D) {
        static_for<0, D>([&](auto u) {
          const double x0 = r0[u], x1 = r1[u];
          node_at(clampi(t + u + D, 1, p1 - 1), r0[u], r1[u]);
          emit(t + u, FUSED ? (x0 + x1) + lzc : x0, 0.0);
        });
      }
      static_for<0, D>([&](auto u) { if (t + u < p1) emit(t + u, FUSED ? (r0[u] + r1[u]) + lzc : r0[u], 0.0); });
    }
  }
  __syncthreads();

  // ---- phase 2: the two recursions, concurrently ------------------------------------------------------------------------------
  if (wv == 0) {
    double lzM = 1.0, lzS = 0.0, alpha = 0.0;
    long lzE = 0;
    const double colone = col ? 1.0 : 0.0;
    // (the message is renormalised every HMM2_NORM-th step only -- the sum, its reciprocal and the scaling are half of a
    //  step's dependent chain; phase 3 normalises per step anyway, so any positive scaling of alpha^_t will do.  Between
    //  two renormalisations the message can shrink by at most (1e-200)^(HMM2_NORM - 1) before the next sum flags the sequence)
    auto fstep = [&](int t, double e, double m, auto norm, auto renorm) {
      double p0 = 0.0, p1 = 0.0;                        // two accumulators: half the dependent chain
      dpp_fence(alpha);
      static_for<0, K>([&](auto j) { if constexpr (j % 2 == 0) mac_bc<j>(p0, alpha, P[j]); else mac_bc<j>(p1, alpha, P[j]); });
      const double first = t == 0 ? 1.0 : 0.0;          // (alpha starts at 0: pred_0 = 1 in the live lanes)
      const double pred = __builtin_fma(first, colone, p0 + p1);
      double al = pred * e;
      if constexpr (decltype(norm)::value) {
        double c0 = 0.0, c1 = 0.0;
        dpp_fence(al);
        static_for<0, K>([&](auto k) { if constexpr (k % 2 == 0) mac_bc<k>(c0, al, one); else mac_bc<k>(c1, al, one); });
        const double cs = c0 + c1;
        bad = bad || !(cs > HMM_TINY);
        al *= rcp_nr(cs);
        lzM *= __builtin_amdgcn_frexp_mant(cs);
        lzE += __builtin_amdgcn_frexp_exp(cs);
      }
      alpha = al;
      if (st) wsb[(long)t * HMM_WS + OA] = alpha;
      lzS += m + (t > 0 ? pmax : 0.0);
      if constexpr (decltype(renorm)::value) { lzE += __builtin_amdgcn_frexp_exp(lzM); lzM = __builtin_amdgcn_frexp_mant(lzM); }
    };
    double er[D], mr[D];
    static_for<0, D>([&](auto u) {
      const int tt = clampi(u, 0, T - 1);
      er[u] = wsb[(long)tt * HMM_WS + OE];
      mr[u] = wsr[(long)tt * HMM_WS + OM];
    });
    int t = 0;
    for (; t + D <= T; t += D) {
      static_for<0, D>([&](auto u) {
        const double e = er[u], m = mr[u];
        const int tn = clampi(t + u + D, 0, T - 1);
        er[u] = wsb[(long)tn * HMM_WS + OE];
        mr[u] = wsr[(long)tn * HMM_WS + OM];
        fstep(t + u, e, m, std::integral_constant<bool, (u % HMM2_NORM) == HMM2_NORM - 1>{}, std::integral_constant<bool, u == D - 1>{});
      });
    }
    static_for<0, D>([&](auto u) {
      if (t + u < T) fstep(t + u, er[u], mr[u], std::integral_constant<bool, (u % HMM2_NORM) == HMM2_NORM - 1>{},
                           std::integral_constant<bool, u == D - 1>{});
    });
    {
      // what the last renormalisation left unscaled
      double f0 = 0.0, f1 = 0.0;
      dpp_fence(alpha);
      static_for<0, K>([&](auto k) { if constexpr (k % 2 == 0) mac_bc<k>(f0, alpha, one); else mac_bc<k>(f1, alpha, one); });
      const double fs = f0 + f1;
      bad = bad || !(fs > HMM_TINY);
      lzM *= __builtin_amdgcn_frexp_mant(fs);
      lzE += __builtin_amdgcn_frexp_exp(fs);
    }
    if (valid && c == 0) a.logZ[b] = lzS + ::log(lzM) + (double)lzE * 0.6931471805599453094;
  } else if (wv == 1) {
    // w_{T-1} = e_{T-1} (beta_{T-1} = 1);  t = T-2 .. 1:  beta^_t = P w_{t+1} / sum,  w_t = e_t o beta^_t
    double w = col ? wsb[(long)(T - 1) * HMM_WS + OE] : 0.0;
    if (st) wsb[(long)(T - 1) * HMM_WS + OW] = w;
    auto bstep = [&](int t, double e, auto norm) {
      double q0 = 0.0, q1 = 0.0;
      dpp_fence(w);
      static_for<0, K>([&](auto k) { if constexpr (k % 2 == 0) mac_bc<k>(q0, w, PT[k]); else mac_bc<k>(q1, w, PT[k]); });
      double q = q0 + q1;                                 // lane j: sum_k P[j][k] w[k]
      if constexpr (decltype(norm)::value) {
        double d0 = 0.0, d1 = 0.0;
        dpp_fence(q);
        static_for<0, K>([&](auto j) { if constexpr (j % 2 == 0) mac_bc<j>(d0, q, one); else mac_bc<j>(d1, q, one); });
        const double d = d0 + d1;
        bad = bad || !(d > HMM_TINY);
        q *= rcp_nr(d);
      }
      w = e * q;
      if (st) wsb[(long)t * HMM_WS + OW] = w;
    };
    const int nsteps = T - 2 > 0 ? T - 2 : 0;            // step i: t = T-2-i  (t >= 1)
    if (nsteps > 0) {
      double er[D];
      static_for<0, D>([&](auto u) { er[u] = wsb[(long)clampi(T - 2 - u, 1, T - 1) * HMM_WS + OE]; });
      int i = 0;
      for (; i + D <= nsteps; i += D) {
        static_for<0, D>([&](auto u) {
          const double e = er[u];
          er[u] = wsb[(long)clampi(T - 2 - (i + u + D), 1, T - 1) * HMM_WS + OE];
          bstep(T - 2 - (i + u), e, std::integral_constant<bool, (u % HMM2_NORM) == HMM2_NORM - 1>{});
        });
      }
      static_for<0, D>([&](auto u) {
        if (i + u < nsteps) bstep(T - 2 - (i + u), er[u], std::integral_constant<bool, (u % HMM2_NORM) == HMM2_NORM - 1>{});
      });
    }
  }
  __syncthreads();

  // ---- phase 3: marginals and transition counts, half of the steps each -----------------------------------------------------------
  double acc[K];
  static_for<0, K>([&](auto j) { acc[j] = 0.0; });
  {
    const int q0_ = (int)((long)T * wv / NW);
    const int p1 = (int)((long)T * (wv + 1) / NW);
    const int q1_ = p1 < T - 1 ? p1 : T - 1;             // steps q0_ .. q1_-1 have a successor
    double* oS = a.E_states + ((long)b * T) * K + cc;
    double gam0 = 0.0;
    auto cstep = [&](int t, double al, double w1) {
      double s0 = 0.0, s1 = 0.0;
      dpp_fence(w1);
      static_for<0, K>([&](auto k) { if constexpr (k % 2 == 0) mac_bc<k>(s0, w1, PT[k]); else mac_bc<k>(s1, w1, PT[k]); });
      double g = al * (s0 + s1);
      double z0 = 0.0, z1 = 0.0;
      dpp_fence(g);
      static_for<0, K>([&](auto j) { if constexpr (j % 2 == 0) mac_bc<j>(z0, g, one); else mac_bc<j>(z1, g, one); });
      const double Z = z0 + z1;
      bad = bad || !(Z > HMM_TINY);
      const double rz = rcp_nr(Z);
      const double gam = g * rz;
      if (st) oS[(long)t * K] = gam;
      gam0 = t == 0 ? gam : gam0;
      double alz = al * rz;
      dpp_fence(alz);
      static_for<0, K>([&](auto j) { mac_bc<j>(acc[j], alz, w1); });     // lane k: alpha^_t[j] w_{t+1}[k] / Z_t
    };
    if (q0_ < q1_) {
      double ar[D], wr[D];
      static_for<0, D>([&](auto u) {
        const int tt = clampi(q0_ + u, 0, T - 2);
        ar[u] = wsb[(long)tt * HMM_WS + OA];
        wr[u] = wsb[(long)(tt + 1) * HMM_WS + OW];
      });
      int t = q0_;
      for (; t + D <= q1_; t += D) {
        static_for<0, D>([&](auto u) {
          const double al = ar[u], w1 = wr[u];
          const int tn = clampi(t + u + D, 0, T - 2);
          ar[u] = wsb[(long)tn * HMM_WS + OA];
          wr[u] = wsb[(long)(tn + 1) * HMM_WS + OW];
          cstep(t + u, al, w1);
        });
      }
      static_for<0, D>([&](auto u) { if (t + u < q1_) cstep(t + u, ar[u], wr[u]); });
    }
    if (q0_ == 0 && q0_ < q1_ && st) a.E_init[(long)b * K + c] = gam0;      // (the wavefront whose range holds step 0)
    if (wv == NW - 1) {                                   // beta_{T-1} = 1: gamma = alpha^ / its sum
      double al = col ? wsb[(long)(T - 1) * HMM_WS + OA] : 0.0;
      double f0 = 0.0, f1 = 0.0;
      dpp_fence(al);
      static_for<0, K>([&](auto k) { if constexpr (k % 2 == 0) mac_bc<k>(f0, al, one); else mac_bc<k>(f1, al, one); });
      const double fs = f0 + f1;
      bad = bad || !(fs > HMM_TINY);
      al *= rcp_nr(fs);
      if (st) oS[(long)(T - 1) * K] = al;
      if (T == 1 && st) a.E_init[(long)b * K + c] = al;
    }
  }
  if (bad && valid && c == 0) wsr[HMM_REDO] = 1.0;
  if (wv > 0) static_for<0, K>([&](auto j) { xacc[((wv - 1) * 16 + j) * 64 + lane] = acc[j]; });
  __syncthreads();
  if (wv == 0 && st)
    static_for<0, K>([&](auto j) {
      double s = acc[j];
      static_for<0, NW - 1>([&](auto w) { s += xacc[(w * 16 + j) * 64 + lane]; });       // (fixed order)
      a.E_trans[(long)b * K * K + j * K + c] = s * P[j];
    });
}

template <int K>
static int launch_hmm(const HmmArgs& a, hipStream_t s) {
#if SVAE_HMM_TWOEND
  {
    HmmArgs r = a;
    r.redo_only = 1;
    if (a.pair_contr) {
      hipLaunchKernelGGL((hmm_estep2_kernel<K, true>), dim3((a.B + 3) / 4), dim3(64 * HMM2_WAVES), 0, s, a);
      hipLaunchKernelGGL((hmm_estep_kernel<K, true>), dim3((a.B + 3) / 4), dim3(64), 0, s, r);
    } else {
      hipLaunchKernelGGL((hmm_estep2_kernel<K, false>), dim3((a.B + 3) / 4), dim3(64 * HMM2_WAVES), 0, s, a);
      hipLaunchKernelGGL((hmm_estep_kernel<K, false>), dim3((a.B + 3) / 4), dim3(64), 0, s, r);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  }
#endif
  if (a.pair_contr) hipLaunchKernelGGL((hmm_estep_kernel<K, true>), dim3((a.B + 3) / 4), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((hmm_estep_kernel<K, false>), dim3((a.B + 3) / 4), dim3(64), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// ---- glue of one SLDS coordinate-ascent sweep (slds_svae.py:159-175), after the HMM and the fused LDS kernels ----------
// one wavefront per active slot: lds_vlb = lognorm + <E z_0, cinit> + sum_{t>=1} <E z_t, lz>;  vlb_new = hmm_vlb + lds_vlb;
// iters += 1;  keep[slot] = |vlb_new - vlb| >= tol;  vlb = vlb_new
struct SldsGlueArgs {
  int nrun, T, K;
  double tol;
  const int32_t* __restrict__ seq_index;   // (nrun) or nullptr
  const double* __restrict__ E_states;     // (rows,T,K)
  const double* __restrict__ cinit;        // (K)
  const double* __restrict__ lz;           // (K)
  const double* __restrict__ lognorm;      // (rows) fused LDS kernel
  const double* __restrict__ hmm_vlb;      // (rows)
  double* __restrict__ lds_vlb;            // (rows) out
  double* __restrict__ vlb;                // (rows) in/out
  int32_t* __restrict__ iters;             // (rows) in/out
  int32_t* __restrict__ keep;              // (nrun) out: 1 = still iterating
};

__global__ __launch_bounds__(256) void slds_glue_kernel(const SldsGlueArgs a) {
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slot >= a.nrun) return;
  const int r = a.seq_index ? a.seq_index[slot] : slot;
  if (r < 0) { if (lane == 0) a.keep[slot] = 0; return; }            // unused slot
  const double* es = a.E_states + (long)r * a.T * a.K;
  const int K = a.K, TK = a.T * K;
  double acc = 0.0;
  for (int q = lane; q < TK; q += 64) {
    const int k = q % K;
    acc = __builtin_fma(es[q], q < K ? a.cinit[k] : a.lz[k], acc);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);       // (fixed order: reproducible)
  if (lane == 0) {
    const double lv = a.lognorm[r] + acc;
    const double nv = a.hmm_vlb[r] + lv;
    a.lds_vlb[r] = lv;
    a.iters[r] += 1;
    a.keep[slot] = !(fabs(nv - a.vlb[r]) < a.tol) ? 1 : 0;
    a.vlb[r] = nv;
  }
}

// stable compaction of the active list: out = [seq_index[i] : keep[i]], count[0] = its length (one workgroup)
__global__ __launch_bounds__(1024) void slds_compact_kernel(int nrun, const int32_t* __restrict__ seq_index,
                                                            const int32_t* __restrict__ keep, int32_t* __restrict__ out,
                                                            int32_t* __restrict__ count) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (nrun + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < nrun ? lo + per : nrun;
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += keep[i] != 0;
  part[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - cnt;
  for (int i = lo; i < hi; ++i)
    if (keep[i] != 0) out[pos++] = seq_index ? seq_index[i] : i;
  // the slots behind the new list are marked unused (-1): a caller may launch the next sweep on `nrun` slots before it
  // has read the new count (slds_svae.py: the host reads the count one sweep late); the kernels skip negative entries
  const int total = part[1023];
  for (int i = total + tid; i < nrun; i += 1024) out[i] = -1;
  if (tid == 1023) count[0] = total;
}

// ---- the two dense contractions at the ends of the SLDS coordinate ascent ------------------------------------------------------
// Node potentials of the HMM for ONE sample path x (B,T,n) (initialize_local_meanfield + get_arhmm_local_nodeparams,
// slds_svae.py:203-226, 131-147): the statistics of a path are outer products, so the contraction with state k's
// parameters is three quadratic forms.  One thread per (sequence, step), loop over the states: the parameters are
// wave-uniform (scalar loads), the path values sit in registers, each thread writes its K potentials contiguously.
template <int N>
__global__ __launch_bounds__(256) void slds_path_nodeparams_kernel(int B, int T, int K, const double* __restrict__ x,
                                                                   const double* __restrict__ init_J,
                                                                   const double* __restrict__ init_h,
                                                                   const double* __restrict__ cinit,
                                                                   const double* __restrict__ J11,
                                                                   const double* __restrict__ J12,
                                                                   const double* __restrict__ J22,
                                                                   const double* __restrict__ lz,
                                                                   double* __restrict__ out) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= (long)B * T) return;
  const int t = (int)(q % T);
  double xa[N], xb[N];                       // x_{t-1} (t >= 1), x_t
  const double* xt = x + q * N;
  static_for<0, N>([&](auto i) { xb[i] = xt[i]; xa[i] = t > 0 ? xt[i - N] : 0.0; });
  double* o = out + q * K;
  for (int k = 0; k < K; ++k) {
    double acc;
    if (t == 0) {
      const double* J = init_J + (long)k * N * N;
      acc = cinit[k];
      static_for<0, N>([&](auto i) {
        double r = init_h[(long)k * N + i];
        static_for<0, N>([&](auto j) { r = __builtin_fma(J[i * N + j], xb[j], r); });
        acc = __builtin_fma(xb[i], r, acc);
      });
    } else {
      const double* A = J11 + (long)k * N * N;
      const double* C = J12 + (long)k * N * N;
      const double* D = J22 + (long)k * N * N;
      acc = lz[k];
      static_for<0, N>([&](auto i) {
        double ra = 0.0, rb = 0.0;
        static_for<0, N>([&](auto j) {
          ra = __builtin_fma(A[i * N + j], xa[j], ra);
          ra = __builtin_fma(C[i * N + j], xb[j], ra);
          rb = __builtin_fma(D[i * N + j], xb[j], rb);
        });
        acc = __builtin_fma(xa[i], ra, acc);
        acc = __builtin_fma(xb[i], rb, acc);
      });
    }
    o[k] = acc;
  }
}

// Per-step pair parameters of the LDS factor under the HMM marginals (get_var_lds_local_natparam, slds_svae.py:92-103):
// out_m[b,t,:] = sum_k w[b,t+1,k] P_m[k,:] for the three n x n blocks and the constant.  Bound by the 3 n^2 + 1 doubles
// written per (sequence, step): thread e of a workgroup owns output element e and keeps its K parameters in
// registers; the weights of a row are wave-uniform (scalar loads); workgroups stride over the rows.
__global__ __launch_bounds__(1024) void slds_mix_pair_kernel(int B, int T, int K, int nn, const double* __restrict__ w,
                                                             const double* __restrict__ J11, const double* __restrict__ J12,
                                                             const double* __restrict__ J22, const double* __restrict__ lz,
                                                             double* __restrict__ o11, double* __restrict__ o12,
                                                             double* __restrict__ o22, double* __restrict__ olz) {
  const int e = threadIdx.x;
  const int m = e / nn, ee = e - m * nn;             // block (0..2: the matrices, 3: the constant), element
  if (m > 3 || (m == 3 && ee > 0)) return;
  const double* P = m == 0 ? J11 : (m == 1 ? J12 : (m == 2 ? J22 : lz));
  double* O = m == 0 ? o11 : (m == 1 ? o12 : (m == 2 ? o22 : olz));
  const int es = m == 3 ? 1 : nn;                    // elements per state / per output row
  double pk[16];
  static_for<0, 16>([&](auto k) { pk[k] = k < K ? P[(long)k * es + ee] : 0.0; });
#ifndef SVAE_MIXPAIR_U
#define SVAE_MIXPAIR_U 4
#endif
#ifndef SVAE_MIXPAIR_NT
#define SVAE_MIXPAIR_NT 1
#endif
  constexpr int U = SVAE_MIXPAIR_U;                  // steps in flight per thread (their weight loads and stores overlap)
  for (long b = blockIdx.y; b < B; b += gridDim.y) {  // (no division in the loops: grid rows stride over the sequences)
    const double* wb = w + (b * T + 1) * K;
    double* Ob = O + b * (T - 1) * es + ee;
    for (int t0 = blockIdx.x * U; t0 < T - 1; t0 += gridDim.x * U) {
      double acc[U];
      static_for<0, U>([&](auto u) {
        const int t = t0 + u < T - 1 ? t0 + u : T - 2;
        const double* wr = wb + (long)t * K;
        acc[u] = 0.0;
        static_for<0, 16>([&](auto k) { if (k < K) acc[u] = __builtin_fma(wr[k], pk[k], acc[u]); });
      });
      // (non-temporal: 2.45 GB at configs[3], read by a later kernel -- 1.07 -> 0.64 ms)
      static_for<0, U>([&](auto u) { if (t0 + u < T - 1) st_stream<SVAE_MIXPAIR_NT != 0>(&Ob[(long)(t0 + u) * es], acc[u]); });
    }
  }
}


// The two contractions of the SLDS final pass over the per-step pair statistics S (B, T-1, E = 3 n^2) of the last LDS
// E-step (run_inference, /root/reference/svae/models/slds_svae.py:289-310), in ONE pass over the 2.45 GB they are at
// configs[3] (they were two library GEMMs, each reading S: 2.2 + 1.0 ms):
//   node[b, t+1, k] = <S[b,t], P_k> + lz_k           get_arhmm_local_nodeparams  :131-147
//   G[k]           += w[b, t+1, k] S[b,t]            get_global_stats            :229-243 (summed over b and t)
// A wavefront takes one step at a time: lane l holds the entries l, l + 64, .. of S[b,t] (coalesced 512-byte rows, two
// steps prefetched), the K parameter rows of its entries and the K x EJ accumulators of G in registers; the K inner
// products are reduced across the wavefront by ONE multi-value butterfly (7 shuffles for 8 sums: each exchange halves the
// number of sums a lane still carries); the weights of a step are wave-uniform scalars.  Workgroups stride over the
// sequences and leave one (KK, E) partial of G each (summed by the caller in index order: deterministic).
#ifndef SVAE_PC_WPC
#define SVAE_PC_WPC 2
#endif
template <int EJ, int KK>
__global__ __launch_bounds__(256, SVAE_PC_WPC) void slds_pair_contract_kernel(int B, int T, int K, int E, const double* __restrict__ S,
                                                                   const double* __restrict__ P, const double* __restrict__ lz,
                                                                   const double* __restrict__ w, double* __restrict__ node,
                                                                   double* __restrict__ gpart) {
  static_assert(KK == 8, "the butterfly below is written for eight sums");
  __shared__ double red[EJ * KK * 64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  double pk[EJ][KK], ga[EJ][KK];
  int eo[EJ];
  static_for<0, EJ>([&](auto j) {
    const int e = lane + 64 * j;
    eo[j] = 8 * (e < E ? e : E - 1);                 // byte offset inside a row of S
    static_for<0, KK>([&](auto k) { pk[j][k] = (e < E && k < K) ? P[(long)(k < K ? k : 0) * E + eo[j] / 8] : 0.0; ga[j][k] = 0.0; });
  });
  const double lzl = lz[lane < K ? lane : 0];
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  // (no branch inside the time loop -- across one hipcc's wait counts degrade to vmcnt(0) and the prefetched rows are
  //  waited for at once: state indices are clamped and masked, and the lanes without a state store to a dummy word)
  int kx[KK];
  double mk[KK];
  static_for<0, KK>([&](auto k) { kx[k] = k < K ? k : K - 1; mk[k] = k < K ? 1.0 : 0.0; });
  const bool writer = lane < K;
  const int steps = T - 1;
  // store target of the lanes without a state: one word per WAVEFRONT inside this workgroup's own partial block (entries
  // e = wv of row k = 0), which the reduction below overwrites after its barriers -- no word shared between workgroups,
  // nothing the caller has to over-allocate
  double* const dummy = gpart + (long)blockIdx.x * KK * E + (wv < E ? wv : 0);
  for (long b = blockIdx.x; b < B; b += gridDim.x) {
    const double* Sb = S + b * steps * E;
    const double* wb = w + (b * T + 1) * K;
    double* nb = node + (b * T + 1) * K + (writer ? lane : 0);
    // register ring of D rows, every slot with a fixed role in the unrolled body (a rotation by moves would make each
    // move wait for the load it forwards): slot r is consumed and at once re-requested D steps further on
#ifndef SVAE_PC_D
#define SVAE_PC_D 2
#endif
    constexpr int D = SVAE_PC_D;
    double ring[D][EJ], wring[D];          // wring: the step's K weights, one per lane (read back with v_readlane:
    const int wl = 8 * (lane < K ? lane : K - 1);     // as eight scalar loads they were waited for one by one)
    // (row base through readfirstlane: a scalar base + the lane's 32-bit offset per load; left to itself hipcc keeps one
    //  64-bit pointer per (slot, chunk) in vector registers -- 30 of them -- and spills)
    auto load = [&](int t, double (&v)[EJ], double& wv_) {
      const uint64_t a = (uint64_t)(Sb + (long)(t < steps ? t : steps - 1) * E);
      const char* r = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)a));
      static_for<0, EJ>([&](auto j) { v[j] = *(const double*)(r + eo[j]); });
      const uint64_t aw = (uint64_t)(wb + (long)(t < steps ? t : steps - 1) * K);
      const char* rw = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(aw >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)aw));
      wv_ = *(const double*)(rw + wl);
    };
    static_for<0, D>([&](auto r) { load(wv + 4 * r, ring[r], wring[r]); });
    for (int t0 = wv; t0 < steps; t0 += 4 * D) {
      static_for<0, D>([&](auto r) {
        const int t = t0 + 4 * r;
        const bool live = t < steps;                       // (wave-uniform; a dead step computes on a clamped row)
        const double lv = live ? 1.0 : 0.0;
        double acc[KK];
        static_for<0, KK>([&](auto k) { acc[k] = 0.0; });
        static_for<0, KK>([&](auto k) {
          const double wk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wring[r]), kx[k]),
                                             __builtin_amdgcn_readlane(__double2loint(wring[r]), kx[k])) * (mk[k] * lv);
          static_for<0, EJ>([&](auto j) {
            acc[k] = __builtin_fma(ring[r][j], pk[j][k], acc[k]);
            ga[j][k] = __builtin_fma(wk, ring[r][j], ga[j][k]);
          });
        });
        load(t + 4 * D, ring[r], wring[r]);
        double r4[4], r2[2];
        static_for<0, 4>([&](auto i) {
          const double send = b0 ? acc[2 * i] : acc[2 * i + 1], keep = b0 ? acc[2 * i + 1] : acc[2 * i];
          r4[i] = keep + __shfl_xor(send, 1, 64);
        });
        static_for<0, 2>([&](auto i) {
          const double send = b1 ? r4[2 * i] : r4[2 * i + 1], keep = b1 ? r4[2 * i + 1] : r4[2 * i];
          r2[i] = keep + __shfl_xor(send, 2, 64);
        });
        double tot = (b2 ? r2[1] : r2[0]) + __shfl_xor(b2 ? r2[0] : r2[1], 4, 64);     // lane: the sum of state (lane & 7)
        tot += __shfl_xor(tot, 8, 64);
        tot += __shfl_xor(tot, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        double* dst = (writer && live) ? nb + (long)t * K : dummy;
        *dst = tot + lzl;
      });
    }
  }
  // the workgroup's partial of G: the four wavefronts' accumulators added in index order through LDS
  for (int q = 0; q < 4; ++q) {
    if (wv == q) {
      static_for<0, EJ>([&](auto j) {
        static_for<0, KK>([&](auto k) {
          double* r = red + (j * KK + k) * 64 + lane;
          *r = q == 0 ? ga[j][k] : *r + ga[j][k];
        });
      });
    }
    __syncthreads();
  }
  double* gp = gpart + (long)blockIdx.x * KK * E;
  for (int idx = threadIdx.x; idx < EJ * KK * 64; idx += 256) {
    const int l = idx & 63, k = (idx >> 6) % KK, j = idx / (64 * KK);
    const int e = l + 64 * j;
    if (e < E) gp[(long)k * E + e] = red[idx];
  }
}

}  // namespace svae

extern "C" int svae_hmm_wide_launch(const svae::HmmArgs* a, void* stream);     // hmm_estep_wide.hip: 17 <= K <= 64

extern "C" size_t svae_hmm_workspace_bytes(int B, int T, int K) {
  if (B <= 0 || T <= 0 || K <= 0 || K > SVAE_HMM_MAX_K) return 0;
  if (K > 16) return (size_t)B * T * svae::hmm_wide_rec(svae::hmm_wide_kp(K)) * sizeof(double);
  return (size_t)B * T * svae::HMM_WS * sizeof(double);
}

static int hmm_dispatch(const svae::HmmArgs& a, void* stream);

extern "C" int svae_hmm_estep_f64(int B, int T, int K, int pair_batched,
                                  const double* init_params, const double* pair_params,
                                  const double* node_params,
                                  double* logZ, double* E_init, double* E_trans, double* E_states,
                                  void* workspace, size_t ws_bytes, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (K < 1 || K > SVAE_HMM_MAX_K) return -3;
  if (!init_params) return -5;
  if (!pair_params) return -6;
  if (!node_params) return -7;
  if (!logZ) return -8;
  if (!E_init) return -9;
  if (!E_trans) return -10;
  if (!E_states) return -11;
  if (!workspace || ws_bytes < svae_hmm_workspace_bytes(B, T, K)) return -12;
  if (B == 0) return 0;
  svae::HmmArgs a;
  a.B = B; a.T = T; a.K = K; a.pair_stride = pair_batched ? (long)K * K : 0;
  a.init_params = init_params; a.pair_params = pair_params; a.node_params = node_params;
  a.logZ = logZ; a.E_init = E_init; a.E_trans = E_trans; a.E_states = E_states;
  a.ws = (double*)workspace;
  a.seq_index = nullptr; a.n = 0; a.pair_contr = nullptr; a.lds_E_init = nullptr; a.init_J = nullptr; a.init_h = nullptr;
  a.cinit = nullptr; a.lz = nullptr; a.node_out = nullptr; a.redo_only = 0;
  return hmm_dispatch(a, stream);
}

// HMM step of the SLDS coordinate ascent (hmm_meanfield, /root/reference/svae/models/slds_svae.py:108-115) on the rows
// `seq_index` lists, the node potentials taken from `node_params` (rows,T,K) if given, else built on the fly from the
// fused LDS mean-field kernel's outputs (get_arhmm_local_nodeparams, :131-147).
extern "C" int svae_slds_hmm_meanfield_f64(int B, int rows, int T, int K, int n,
                                           const double* hmm_init, const double* hmm_pair, const double* node_params,
                                           const double* pair_contr, const double* lds_E_init, const double* init_J,
                                           const double* init_h, const double* cinit, const double* lz,
                                           const int32_t* seq_index,
                                           double* logZ, double* E_init, double* E_trans, double* E_states,
                                           double* node_out, void* workspace, size_t ws_bytes, void* stream) {
  if (B < 0 || B > rows) return -1;
  if (T < 1) return -3;
  if (K < 1 || K > SVAE_HMM_MAX_K) return -4;
  if (n < 1 || n > 64) return -5;
  if (!hmm_init) return -6;
  if (!hmm_pair) return -7;
  if (!node_params && (K > 16 || !pair_contr || !lds_E_init || !init_J || !init_h || !cinit || !lz)) return -8;   /* K > 16: node potentials given */
  if (!logZ) return -16;
  if (!E_init) return -17;
  if (!E_trans) return -18;
  if (!E_states) return -19;
  if (!workspace || ws_bytes < svae_hmm_workspace_bytes(rows, T, K)) return -21;
  if (B == 0) return 0;
  svae::HmmArgs a;
  a.B = B; a.T = T; a.K = K; a.pair_stride = 0;
  a.init_params = hmm_init; a.pair_params = hmm_pair; a.node_params = node_params;
  a.logZ = logZ; a.E_init = E_init; a.E_trans = E_trans; a.E_states = E_states;
  a.ws = (double*)workspace;
  a.seq_index = seq_index; a.n = n;
  a.pair_contr = node_params ? nullptr : pair_contr; a.lds_E_init = lds_E_init; a.init_J = init_J; a.init_h = init_h;
  a.cinit = cinit; a.lz = lz; a.node_out = node_out; a.redo_only = 0;
  return hmm_dispatch(a, stream);
}

// After the HMM and the fused LDS kernels of a sweep: per listed row the LDS bound with its mixed constants, the
// stopping test |vlb_new - vlb| < tol of slds_svae.py:170-172, the sweep counter, and the compacted list of the rows
// still iterating (`next_index`, `next_count[0]`; stable order) -- no host arithmetic between two sweeps.
extern "C" int svae_slds_sweep_glue_f64(int B, int T, int K, double tol, const int32_t* seq_index,
                                        const double* E_states, const double* cinit, const double* lz,
                                        const double* lognorm, const double* hmm_vlb, double* lds_vlb, double* vlb,
                                        int32_t* iters, int32_t* keep_scratch, int32_t* next_index, int32_t* next_count,
                                        void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (K < 1 || K > 16) return -3;
  if (!E_states || !cinit || !lz || !lognorm || !hmm_vlb) return -6;
  if (!lds_vlb || !vlb || !iters || !keep_scratch || !next_index || !next_count) return -11;
  hipStream_t s = (hipStream_t)stream;
  if (B > 0) {
    svae::SldsGlueArgs g;
    g.nrun = B; g.T = T; g.K = K; g.tol = tol; g.seq_index = seq_index; g.E_states = E_states; g.cinit = cinit; g.lz = lz;
    g.lognorm = lognorm; g.hmm_vlb = hmm_vlb; g.lds_vlb = lds_vlb; g.vlb = vlb; g.iters = iters; g.keep = keep_scratch;
    hipLaunchKernelGGL(svae::slds_glue_kernel, dim3((B + 3) / 4), dim3(256), 0, s, g);
    if (hipGetLastError() != hipSuccess) return -1000;
  }
  hipLaunchKernelGGL(svae::slds_compact_kernel, dim3(1), dim3(1024), 0, s, B, seq_index, keep_scratch, next_index, next_count);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// Sweep 0 of the ascent: HMM node potentials from one sample path (see slds_path_nodeparams_kernel).
extern "C" int svae_slds_path_nodeparams_f64(int B, int T, int K, int n, const double* x, const double* init_J,
                                             const double* init_h, const double* cinit, const double* J11,
                                             const double* J12, const double* J22, const double* lz, double* node_out,
                                             void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (K < 1 || K > 16) return -3;
  if (n < 1 || n > 15) return -4;
  if (!x) return -5;
  if (!init_J || !init_h || !cinit) return -6;
  if (T > 1 && (!J11 || !J12 || !J22 || !lz)) return -9;
  if (!node_out) return -13;
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(((long)B * T + 255) / 256)), block(256);
  switch (n) {
#define SVAE_CASE(NN) case NN: hipLaunchKernelGGL((svae::slds_path_nodeparams_kernel<NN>), grid, block, 0, s, B, T, K, x, \
                                                  init_J, init_h, cinit, J11, J12, J22, lz, node_out); break;
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7) SVAE_CASE(8)
    SVAE_CASE(9) SVAE_CASE(10) SVAE_CASE(11) SVAE_CASE(12) SVAE_CASE(13) SVAE_CASE(14) SVAE_CASE(15)
#undef SVAE_CASE
  }
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// The per-step pair parameters of the converged mean field (see slds_mix_pair_kernel): E_states (B,T,K) ->
// J11 / J12 / J22 (B,T-1,n,n), logZ (B,T-1).
extern "C" int svae_slds_mix_pair_natparam_f64(int B, int T, int K, int n, const double* E_states, const double* J11,
                                               const double* J12, const double* J22, const double* lz, double* out_J11,
                                               double* out_J12, double* out_J22, double* out_logZ, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (K < 1 || K > 16) return -3;
  if (n < 1 || 3 * n * n + 1 > 1024) return -4;
  if (!E_states) return -5;
  if (!J11 || !J12 || !J22 || !lz) return -6;
  if (!out_J11 || !out_J12 || !out_J22 || !out_logZ) return -10;
  if (B == 0 || T == 1) return 0;
  const int threads = ((3 * n * n + 1) + 63) / 64 * 64;
  const int per_seq = (T - 1 + SVAE_MIXPAIR_U - 1) / SVAE_MIXPAIR_U;    // U steps per thread and trip
#ifndef SVAE_MIXPAIR_GX
#define SVAE_MIXPAIR_GX 8
#endif
  const unsigned gx = (unsigned)(per_seq < SVAE_MIXPAIR_GX ? per_seq : SVAE_MIXPAIR_GX);
  hipLaunchKernelGGL(svae::slds_mix_pair_kernel, dim3(gx, (unsigned)(B < 65535 ? B : 65535)), dim3(threads), 0, (hipStream_t)stream, B, T, K, n * n,
                     E_states, J11, J12, J22, lz, out_J11, out_J12, out_J22, out_logZ);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// One pass over the per-step pair statistics of the SLDS final pass (see slds_pair_contract_kernel): node_out (B,T,K) rows
// 1 .. T-1 and `blocks` partials (blocks, 8, 3 n^2) of the weighted sums.  n <= 10, K <= 8.
extern "C" int svae_slds_pair_contract_f64(int B, int T, int K, int n, const double* pair_stats, const double* P,
                                           const double* lz, const double* weights, double* node_out, double* gpart,
                                           int blocks, void* stream) {
  if (B < 0) return -1;
  if (T < 2) return -2;
  if (K < 1 || K > 8) return -3;
  if (n < 1 || n > 10) return -4;
  if (!pair_stats) return -5;
  if (!P || !lz) return -6;
  if (!weights) return -8;
  if (!node_out || !gpart) return -9;
  if (blocks < 1) return -11;
  const int E = 3 * n * n;
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) {
    (void)hipMemsetAsync(gpart, 0, sizeof(double) * (size_t)blocks * 8 * E, s);
    return 0;
  }
  if (blocks > B) {       // (workgroups without a sequence would leave their partial unwritten)
    (void)hipMemsetAsync(gpart + (size_t)B * 8 * E, 0, sizeof(double) * (size_t)(blocks - B) * 8 * E, s);
    blocks = B;
  }
  hipLaunchKernelGGL((svae::slds_pair_contract_kernel<5, 8>), dim3(blocks), dim3(256), 0, s, B, T, K, E, pair_stats, P, lz,
                     weights, node_out, gpart);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

static int hmm_dispatch(const svae::HmmArgs& a, void* stream) {
  const int K = a.K;
  hipStream_t s = (hipStream_t)stream;
  if (K > 16) return svae_hmm_wide_launch(&a, stream);
  switch (K) {
#define SVAE_CASE(KK) case KK: return svae::launch_hmm<KK>(a, s);
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10) SVAE_CASE(11) SVAE_CASE(12) SVAE_CASE(13)
    SVAE_CASE(14) SVAE_CASE(15) SVAE_CASE(16)
#undef SVAE_CASE
  }
  return -3;
}
