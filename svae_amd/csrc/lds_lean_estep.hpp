// lds_lean_estep.hpp -- E-step + backward sampler in ONE launch on LEAN per-step records (round 6).
//
// What it replaces (reference = mattjj/svae, /root/reference): the composite the model layer calls,
//   cython_natural_lds_inference_general   svae/lds/lds_inference.py:196-202
//     natural_filter_forward_general       svae/lds/cython_lds_inference.pyx:28-90
//     natural_smoother_general             svae/lds/cython_lds_inference.pyx:149-210
//     natural_sample_backward              svae/lds/cython_lds_inference.pyx:310-355
// for large batches of sequences sharing homogeneous pair parameters -- north_star's 4096 x T = 200 x n = 10 training
// step.  Same algorithm, mapping and arithmetic as lds_estep_kernel.hpp (one DPP row per sequence, four sequences per
// wavefront, Gauss-Jordan forward, moment-form backward); what changes is what travels through HBM between the two
// halves and on to the VJP sweeps.  The packed kernels at this batch size are one wavefront per SIMD and wait on memory
// (rocprof round 5: 4.8 GB per launch, VALU busy 25 %), so the record is what they cost:
//   * the forward half keeps per step ONLY U = chol(P)^-T (upper triangle, packed) and c = P^-1 h_filt: 66 doubles at
//     n = 10 instead of H = [P^-1 J12 | c] (120) + P^-1 (100) + the LDL' factor (110);
//   * the backward half rebuilds P^-1 = U U' (55 DPP multiply-adds) and (P^-1 J12)' = J12' P^-1 (100) per step;
//   * the backward SAMPLER runs inside the backward half (it walks the same records in the same order): x_t = c_t
//     - (P^-1 J12) x_{t+1} + U eps_t costs 20 multiply-adds per sample and no second pass over the records
//     (lds_sample_vec_kernel re-read H and the factor: 1.7 GB per launch at 4096 sequences).
// U eps IS the reference's noise map chol(P)^-T eps (dtrtrs 'L','T', cython_gaussian_grads.pxd:431-454): equal eps
// give equal samples.  Layout of the record: lds_args.hpp (lean_*).
#pragma once
#include "lds_estep_kernel.hpp"

namespace svae {

// SM: samples per sequence the instantiation holds registers for (0: no sampling; 1: the training step's S = 1; 2)
// INH: per-step pair parameters (T-1,n,n) / per-sequence (B,T-1,n,n) (a.pair_seq_stride), per-step pair statistics out --
// the final pass of the SLDS's run_inference (slds_svae.py:289-310) on the mixed parameters of the converged mean field:
// forward only (no cross-moment record: the VJP with statistics cotangents runs on the full records); J12_t is read a
// second time in the backward half instead of (P^-1 J12)_t being stored and re-read.
template <int N, int SM, bool INH = false>
__global__ __launch_bounds__(64) void lds_infer_lean_kernel(const LdsArgs a, const LeanSample ls) {
  static_assert(N >= 1 && N <= LEAN_MAX_N, "lean records: n <= 10");
  constexpr int IL = SVAE_IL;
  constexpr int HS = ws_h_stride(N);
  constexpr int TRI = lean_tri(N), LR = lean_rec_doubles(N), TRASH = lean_trash(N);
  constexpr bool SAMP = SM > 0;
  constexpr int SMAX = SAMP ? SM : 1;
  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int b0 = blockIdx.x * 4;            // (uniform) first sequence of the wavefront
  const int brow = b0 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;     // surplus rows repeat the last sequence (same values to the same addresses)
  const bool col = c < N;
  const bool st = valid && col;
  const bool sth = valid && c <= N;
  const int cc = col ? c : 0;
  const int T = a.T;
  [[maybe_unused]] const int S = ls.S;

  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EN = (c == N) ? 1.0 : 0.0;

  // ---- pair parameters (info form: J = -2 natJ, J12 = -natJ12), column c per lane: as lds_estep_kernel ---------------
  double NJ12T[N], Cc[N];                   // (info-form J12 rows: re-read from L1 per step -- 20 registers)
  const double* pJ11 = a.J11 + (long)b * a.pair_seq_stride;
  const double* pJ12 = a.J12 + (long)b * a.pair_seq_stride;
  const double* pJ22 = a.J22 + (long)b * a.pair_seq_stride;
  auto load_pair = [&](int t, bool with_next_J11) {   // pair t (and J11 of pair t+1): unconditional loads, selected afterwards
    const long o = INH ? (long)t * N * N : 0;
    const long o1 = INH ? (long)(t + 1) * N * N : 0;
    static_for<0, N>([&](auto i) {
      const double r12t = pJ12[o + cc * N + i], r22 = pJ22[o + i * N + cc];
      const double r11 = with_next_J11 ? pJ11[o1 + i * N + cc] : 0.0;
      NJ12T[i] = col ? r12t : 0.0;          // lane j of register k: nat J12[j][k] = -(info) J12[j][k]
      Cc[i] = col ? -2.0 * (r22 + r11) : 0.0;
    });
  };
  if (!INH && T > 1) { load_pair(0, true); dpp_fence(NJ12T); }
  if (T == 1) static_for<0, N>([&](auto i) { NJ12T[i] = 0.0; Cc[i] = 0.0; });

  double An[N];
  static_for<0, N>([&](auto i) {
    const double ij = a.init_J[i * N + cc], ih = a.init_h[i], j11 = T > 1 ? pJ11[i * N + cc] : 0.0;
    An[i] = col ? -2.0 * (ij + j11) : ((c == N) ? ih : 0.0);
  });

  const double* nJ = a.node_J + ((long)b * T) * N + cc;
  const double* nh = a.node_h + ((long)b * T) * N + cc;

  // lean records: uniform base (the wavefront's first sequence) + 32-bit per-lane byte offsets
  char* const recs = reinterpret_cast<char*>(a.ws + (long)b0 * T * LR);
  const unsigned rowoff = (unsigned)(b - b0) * (unsigned)T * (unsigned)LR;       // doubles
  unsigned uoff[N];                          // store of row k of U: lane c >= k -> its packed entry, the others -> trash
  static_for<0, N>([&](auto k) {
    uoff[k] = 8u * (rowoff + ((col && c >= k) ? (unsigned)(lean_row_off(N, k) + c - k) : (unsigned)TRASH));
  });
  const unsigned coff = 8u * (rowoff + (col ? (unsigned)(TRI + c) : (unsigned)TRASH));

  double qacc = 0.0, ldM = 1.0, pmin = 1.0;
  int ldE = 0;
  double Jo_n = nJ[0];
  double ho_n = nh[0];

  // Lanes N+1 .. 15 of every DPP row carry nothing: the two time loops run with them switched off (EXEC) -- 5 of 16
  // lanes of every fp64 operation not toggling; at these batch sizes every SIMD is busy and the chip is power-limited
  // (lds_estep_twoend_rpc.hpp, SVAE_RPC_LANEMASK: the shader clock sags with all 64 lanes live).  -DSVAE_LEAN_LANEMASK=0: A/B.
#ifndef SVAE_LEAN_LANEMASK
#define SVAE_LEAN_LANEMASK 1
#endif
  const bool live = !SVAE_LEAN_LANEMASK || c <= N;
  // ---- forward filter ------------------------------------------------------------------------------------------------
  if (live) {
  for (int t = 0; t < T; ++t) {
    const bool last = (t == T - 1);
    const double Jo = -2.0 * Jo_n;
    double ho = ho_n;
    if (!last) {
      Jo_n = nJ[(long)(t + 1) * N];
      ho_n = nh[(long)(t + 1) * N];
    }
    if (INH && !last) { load_pair(t, t + 1 < T - 1); dpp_fence(NJ12T); }
    double P[N], X[N];
    static_for<0, N>([&](auto i) { P[i] = __builtin_fma(Jo, E[i], An[i]); });
    if (last) {
      asm volatile("; last step: no pair potential, G = 0");   // keep this a branch
      static_for<0, N>([&](auto i) { X[i] = EN * An[i]; });
    } else {
      const long o = INH ? (long)t * N * N : 0;
      static_for<0, N>([&](auto i) { const double r = pJ12[o + i * N + cc]; X[i] = __builtin_fma(EN, An[i], col ? -r : 0.0); });
    }
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(X[i], ho, EN); });
    dpp_fence(P);

    // in-place Gauss-Jordan (exact form: the factor defines the sampler's map), the scaled pivot rows kept in registers:
    // lanes j > k of R[k] hold L[j][k] of P = L D L'
    double pv = 0.0, R[N];
    gauss_jordan<N, true>(P, X, E, qacc, pmin, ldM, ldE, pv, [&](auto kk, double r) { R[kk] = r; });

    // U = L^-T D^-1/2 by back substitution on the unit upper factor (rows of U: zero left of the diagonal, exactly)
    const double dis = rsqrt_nr(col ? pv : 1.0);               // lane k: d_k^-1/2
    double U[N];
    static_for<0, N>([&](auto k) { U[k] = E[k] * dis; });
    dpp_fence(R);
    static_for<1, N>([&](auto jj) {
      constexpr int j = N - jj;
      static_for<0, j>([&](auto k) { mac_bc<j, true>(U[k], R[k], U[j]); });
    });
    // c as a vector: lane i <- lane N of X[i]
    double cv = 0.0;
    dpp_fence(X);
    static_for<0, N>([&](auto i) { mac_bc<N>(cv, X[i], E[i]); });
    {
      char* w = recs + (long)t * (LR * 8);
      static_for<0, N>([&](auto k) {
        asm volatile("" : "+v"(uoff[k]));
        *reinterpret_cast<double*>(w + uoff[k]) = U[k];
      });
      *reinterpret_cast<double*>(w + coff) = cv;
    }

    if (!last) {
      const bool next_last = (t + 1 == T - 1);
      if (!INH && next_last) {
        asm volatile("; next step is the last: its pivot block has no J11 term");   // keep a branch
        static_for<0, N>([&](auto i) { const double r = pJ22[i * N + cc]; An[i] = col ? -2.0 * r : 0.0; });
      } else {
        static_for<0, N>([&](auto i) { An[i] = Cc[i]; });
      }
      asm volatile("s_nop 1");
      static_for<0, (N + IL - 1) / IL>([&](auto g) { rows_src_bcast<IL, g * IL, N, 0>(An, NJ12T, X); });
    }
  }
  }   // live

  // ---- log-normaliser ------------------------------------------------------------------------------------------------
  {
    double z = 0.0;
    if (a.node_logZ) {
      for (int t = c; t < T; t += 16) z += a.node_logZ[(long)b * T + t];
    }
    if (INH) {
      const double* lz = a.logZ_pair + (a.pair_seq_stride ? (long)b * (T - 1) : 0);
      for (int t = c; t < T - 1; t += 16) z += lz[t];
    }
    double total = row_sum16(__builtin_fma(0.5, qacc * EN, z));
    total += a.init_logZ[0];
    if (!INH && T > 1) total += (double)(T - 1) * a.logZ_pair[0];
    total -= 0.5 * (::log(ldM) + (double)ldE * 0.6931471805599453094);
    if (valid && c == 0) a.lognorm[b] = total;
    const bool bad = !(pmin > 0.0) || !(total == total);
    if (bad && valid && c == 0) {
      int old = *(volatile int32_t*)a.info;
      while (old == 0 || old > b + 1) {
        const int seen = atomicCAS(a.info, old, b + 1);
        if (seen == old) break;
        old = seen;
      }
    }
  }

  // ---- backward half: moment-form smoother + sampler on the lean records --------------------------------------------
  double S_[N + 1];
  static_for<0, N + 1>([&](auto i) { S_[i] = 0.0; });
  S_[N] = EN;
  dpp_fence(S_);
  double sumA[INH ? 1 : N], sumW[INH ? 1 : N];
  static_for<0, (INH ? 1 : N)>([&](auto i) { sumA[i] = 0.0; sumW[i] = 0.0; });
  double* const oPair = INH ? a.E_pair + ((long)b * (T - 1)) * 3 * N * N + cc : nullptr;
  double Xs[SMAX];                              // x_{t+1}[c] per sample (lane = vector component)
  static_for<0, SMAX>([&](auto s) { Xs[s] = 0.0; });
  double M[N];                                  // M[k][c] = (c <= k): zeros of U' (lane c of register k holds U[c][k])
  static_for<0, N>([&](auto k) { M[k] = (col && c <= k) ? 1.0 : 0.0; });

  double* oEx = a.E_node_x + ((long)b * T) * N + cc;
  const long dxx_delta = a.E_node_diagxx - a.E_node_x;               // (uniform: one per-lane pointer serves both outputs)

  // operands of a step, fetched one step ahead (raw: no arithmetic before their step; every load unconditional)
  struct Ops { double Ur[N], Ut[N], cv, ep[SMAX], Jt[INH ? N : 1]; };
  const double* jp = INH ? pJ12 + (long)(T > 1 ? T - 2 : 0) * N * N + (long)cc * N : nullptr;   // J12_t[c][.] of the step's pair (t = T-1: unused)
  int jleft = T - 2;                            // pair index the next load_ops fetches (clamped at 0)
  const unsigned lo_r = 8u * (rowoff + (unsigned)cc);                                 // row k of U: + 8 (row_off(k) - k)
  const unsigned lo_t = 8u * (rowoff + (unsigned)(lean_row_off(N, cc) - cc));         // U[c][k]:   + 8 k
  const char* rp = recs + (long)(T - 1) * (LR * 8);            // walking (uniform) record pointer: steps T-1, T-2, ..
  const double* epp = SAMP ? ls.eps + ((long)b * T + (T - 1)) * S * N + cc : nullptr;
  auto load_ops = [&](Ops& o, long more) {
    static_for<0, N>([&](auto k) {
      o.Ur[k] = *reinterpret_cast<const double*>(rp + lo_r + 8 * (lean_row_off(N, k) - k));
      o.Ut[k] = *reinterpret_cast<const double*>(rp + lo_t + 8 * k);
    });
    o.cv = *reinterpret_cast<const double*>(rp + lo_r + 8 * TRI);
    if constexpr (INH) {
      static_for<0, N>([&](auto k) { o.Jt[k] = jp[k]; });
    }
    if constexpr (SAMP) {
      static_for<0, SMAX>([&](auto s) { o.ep[s] = epp[(long)(s < S ? s : S - 1) * N]; });
      epp -= more * S * N;
    }
    rp -= more * (LR * 8);
  };
  // (INH: the first record fetched is step T-1, which has no pair: it re-reads pair T-2; from then on step t reads pair t)
  auto advance_pair = [&](int t_next) {         // called before the load_ops that fetches step t_next
    if constexpr (INH) {
      const int want = t_next < T - 1 ? t_next : T - 2;
      jp += (long)((want > 0 ? want : 0) - (jleft > 0 ? jleft : 0)) * N * N;
      jleft = want;
    }
  };

  auto step = [&](int t, Ops& cur, Ops& nxt) {
    advance_pair(t > 0 ? t - 1 : 0);
    load_ops(nxt, t > 1 ? 1 : 0);               // unconditional prefetch of step t-1 (t = 0: re-reads record 0, unused)
    double Ut[N];
    static_for<0, N>([&](auto k) { Ut[k] = cur.Ut[k] * M[k]; });
    const double cvm = cur.cv * M[N - 1];         // (M[N-1] = (c < N))
    // the recursion-free part of the samples: noise + c_t  (y = c + U eps)
    double Y[SMAX];
    if constexpr (SAMP) {
      dpp_fence(cur.ep);
      static_for<0, SMAX>([&](auto s) { Y[s] = cvm; });
      static_for<0, N>([&](auto k) {
        static_for<0, SMAX>([&](auto s) { mac_bc<k>(Y[s], cur.ep[s], Ut[k]); });
      });
    }
    // P^-1 = U U'
    double Pi[N];
    static_for<0, N>([&](auto i) { Pi[i] = 0.0; });
    dpp_fence(cur.Ur);
    static_for<0, N>([&](auto k) {
      constexpr int kk = decltype(k)::value;
      static_for<0, kk + 1>([&](auto i) { mac_bc<kk>(Pi[i], cur.Ur[i], Ut[kk]); });
    });
    // H[k][c] = (P^-1 J12)[c][k] = sum_j J12[j][k] P^-1[j][c]  (k < N);  H[N] = (c', 1)
    double H[N + 1];
    static_for<0, N>([&](auto k) { H[k] = 0.0; });
    if constexpr (INH) {
      double Jm[N];
      static_for<0, N>([&](auto k) { Jm[k] = cur.Jt[k] * M[N - 1]; });           // lanes >= N: 0
      dpp_fence(Jm);
      static_for<0, N>([&](auto j) {
        static_for<0, N>([&](auto k) { mac_bc<j, true>(H[k], Jm[k], Pi[j]); });
      });
    } else {
      static_for<0, N>([&](auto j) {
        static_for<0, N>([&](auto k) { mac_bc<j, true>(H[k], NJ12T[k], Pi[j]); });
      });
    }
    H[N] = cvm + EN;

    // W~ = S~_{t+1} G~'
    double W[N + 1];
    static_for<0, N + 1>([&](auto i) { W[i] = 0.0; });
    static_for<0, (N + 1 + IL - 1) / IL>([&](auto g) { rows_src_bcast<IL, g * IL, N + 1, N>(W, S_, H); });
    if (a.ws3) {   // VJP mode: keep W~_t (rows 0..N, lanes 0..N)
      double* w3 = a.ws3 + ((long)b * T + t) * (N + 1) * HS + c;
      if (sth) static_for<0, N + 1>([&](auto i) { w3[i * HS] = W[i]; });
    }
    // S~_t = G~ W~ + diag(P^-1, 0) through its transpose
    static_for<0, N>([&](auto i) { S_[i] = Pi[i]; });
    S_[N] = 0.0;
    asm volatile("s_nop 1");   // block entry behind the conditional stores: two wait states before the DPP reads (audit rule)
    static_for<0, (N + 1 + IL - 1) / IL>([&](auto g) { rows_lane_bcast<IL, g * IL, N + 1, N>(S_, W, H); });

    if constexpr (INH) {
      // per-step pair blocks [E x_t x_t' | E x_t x_{t+1}' | E x_{t+1} x_{t+1}'] of pair t: S~_t completes pair t (first
      // block) and pair t-1 (third block); W~ (t < T-1) is pair t, transposed by the addressing
      if (st) {
        if (t < T - 1) {
          double* o = oPair + (long)t * 3 * N * N;
          static_for<0, N>([&](auto i) { o[i * N] = S_[i]; });
          double* o2 = a.E_pair + (((long)b * (T - 1) + t) * 3 + 1) * N * N + (long)cc * N;
          static_for<0, N>([&](auto i) { o2[i] = W[i]; });
        }
        if (t > 0) {
          double* o = oPair + ((long)(t - 1) * 3 + 2) * N * N;
          static_for<0, N>([&](auto i) { o[i * N] = S_[i]; });
        }
      }
    } else if (t < T - 1) static_for<0, N>([&](auto i) { sumA[i] += S_[i]; sumW[i] += W[i]; });
    else if (st) {                                                     // S~_{T-1} waits in its output slot
      double* const ep3 = a.E_pair + (long)b * 3 * N * N + 2 * N * N;
      static_for<0, N>([&](auto i) { ep3[i * N + cc] = S_[i]; });
    }

    // diag E[x_t x_t'] = sum_i (c == i) S~[i] with (c == i) = M[i] - M[i-1] (summed by parts: the identity tile of the
    // forward half is not kept alive through this loop -- 20 registers)
    double dg = M[N - 1] * S_[N - 1], dg1 = 0.0;
    static_for<0, N - 1>([&](auto i) {
      if constexpr (i % 2 == 0) dg = __builtin_fma(M[i], S_[i] - S_[i + 1], dg); else dg1 = __builtin_fma(M[i], S_[i] - S_[i + 1], dg1);
    });
    if (st) {
      oEx[(long)t * N] = S_[N];
      oEx[(long)t * N + dxx_delta] = dg + dg1;
    }

    if constexpr (SAMP) {
      // x_t = y - (P^-1 J12) x_{t+1}
      dpp_fence(Xs);
      static_for<0, SMAX>([&](auto s) {
        if (SM == 1 || s < S) {
          double acc1 = 0.0;
          asm volatile("s_nop 1");   // block entry: two wait states before the first DPP read (audit rule)
          static_for<0, N>([&](auto j) {
            if constexpr (j % 2 == 0) mac_bc<j, true>(Y[s], Xs[s], H[j]); else mac_bc<j, true>(acc1, Xs[s], H[j]);
          });
          const double xt = Y[s] + acc1;
          if (st) ls.samples[(((long)b * T + t) * S + s) * N + c] = xt;
          Xs[s] = xt;
        }
      });
    }
  };

  if (live) {
    Ops Ra, Rb;
    load_ops(Ra, T > 1 ? 1 : 0);                // step T-1 (INH: with pair T-2, unused)
    int t = T - 1;
    for (; t >= 1; t -= 2) {          // two steps per trip: the prefetch buffers ping-pong
      step(t, Ra, Rb);
      step(t - 1, Rb, Ra);
    }
    if (t == 0) step(0, Ra, Rb);
  }

  // ---- global statistics ---------------------------------------------------------------------------------------------
  if (st) {
    double* ei = a.E_init + (long)b * (N * N + N);
    static_for<0, N>([&](auto i) { ei[i * N + cc] = S_[i]; });   // E[x0 x0']
    ei[N * N + cc] = S_[N];                                      // E[x0]
    double* ep = a.E_pair + (long)b * 3 * N * N;
    if constexpr (!INH) static_for<0, N>([&](auto i) {
      ep[i * N + cc] = sumA[i];                               // sum_{t<T-1} E[x_t x_t']
      ep[N * N + cc * N + i] = sumW[i];                       // sum_t E[x_t x_{t+1}'] = (sum_t W_t)'
      const double sl = ep[2 * N * N + i * N + cc];
      ep[2 * N * N + i * N + cc] = (sumA[i] - S_[i]) + sl;    // sum_{t>=1} E[x_t x_t']
    });
  }
}

template <int N>
static int launch_infer_lean(const LdsArgs& a, const LeanSample& ls, bool inhomog, hipStream_t stream) {
  if constexpr (N <= LEAN_MAX_N) {
    dim3 grid((a.B + 3) / 4), block(64);
    if (inhomog) {
      if (a.ws3) return -3;                     // (per-step parameters: forward only)
      if (ls.S > 1) hipLaunchKernelGGL((lds_infer_lean_kernel<N, LEAN_MAX_S, true>), grid, block, 0, stream, a, ls);
      else if (ls.S == 1) hipLaunchKernelGGL((lds_infer_lean_kernel<N, 1, true>), grid, block, 0, stream, a, ls);
      else hipLaunchKernelGGL((lds_infer_lean_kernel<N, 0, true>), grid, block, 0, stream, a, ls);
    } else if (ls.S > 1) hipLaunchKernelGGL((lds_infer_lean_kernel<N, LEAN_MAX_S>), grid, block, 0, stream, a, ls);
    else if (ls.S == 1) hipLaunchKernelGGL((lds_infer_lean_kernel<N, 1>), grid, block, 0, stream, a, ls);
    else hipLaunchKernelGGL((lds_infer_lean_kernel<N, 0>), grid, block, 0, stream, a, ls);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
