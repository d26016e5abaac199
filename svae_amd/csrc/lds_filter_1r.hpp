// lds_filter_1r.hpp -- one-directional forward filter with the ONE-REGISTER Gauss-Jordan of the two-ended kernel,
// for latent dimension n <= 10 and small batches (one sequence per wavefront).
//
// What it computes and leaves behind is exactly what lds_estep_split_kernel<.., CHOL, FILT> does
// (natural_filter_forward_general, svae/lds/cython_lds_inference.pyx:28-90, without the message outputs): the
// log-normaliser, the hand-off records [P^-1 J12 | c] / P^-1 per step in the one-directional workspace layout
// (lds_args.hpp) and the factor region (unit LDL' factor rows + pivots) that the backward sampler
// (svae_lds_sample_f64) and the reverse-mode sweeps (svae_lds_estep_vjp_f64) read.  Only the arithmetic layout is
// the two-ended kernel's (lds_estep_twoend.hpp): one register per matrix row holds [P row | right-hand-side columns
// | h] (columns split over the two DPP rows of a pair), so the elimination costs one DPP FMA per row update
// instead of three, and the scaled pivot rows ARE the factor rows the sampler needs.  It is the longer leg of the
// training step's forward pass (it runs next to the two-ended E-step kernel, see svae_lds_estep_f64): 0.36 -> ~0.2 ms
// at B = 512, T = 200, n = 10.
#pragma once
#include "lds_estep_twoend.hpp"

#ifndef SVAE_FILTER_LDS_SPLIT
#define SVAE_FILTER_LDS_SPLIT 1   // 1: re-replication of the next pivot block through LDS behind the stores; 0: shuffles (A/B)
#endif

namespace svae {

#ifndef SVAE_FILTER_DUAL_MIN_B
#define SVAE_FILTER_DUAL_MIN_B 512
#endif
constexpr int FILTER_1R_DUAL_MIN_B = SVAE_FILTER_DUAL_MIN_B;   // one-directional filter: two sequences per wavefront above this batch

// gauss_jordan_1r (lds_estep_twoend.hpp) with a hook that receives every scaled pivot row (lanes j > k: L[j][k])
template <int N, class StoreR>
__device__ __forceinline__ void gauss_jordan_1r_hook(double (&M)[N], const double (&E)[N], double& qacc, double& vfull,
                                                     StoreR&& store_r) {
  double p = bcast_fenced<0>(M[0]);
  double rinv = rcp_nr(p);
  static_for<0, N>([&](auto k) {
    const double mk0 = __builtin_fma(-p, E[k], M[k]);      // lane k -> exactly 0
    const double ru = mk0 * rinv;                          // scaled pivot row (lane k: 0)
    qacc = __builtin_fma(M[k], ru, qacc);
    vfull = __builtin_fma(-rinv, E[k], vfull);             // lane k <- -1/p_k
    store_r(k, ru);
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
      if constexpr (REM <= 0) static_for<0, 5>(chain);
      static_for<0, N>([&](auto i) {
        if constexpr (i != k && i != k + 1) {
          constexpr int pos = i - (i > k ? 1 : 0) - (i > k + 1 ? 1 : 0);
          update(i, std::false_type{});
          constexpr int lo = pos * 5 / (REM > 0 ? REM : 1), hi = (pos + 1) * 5 / (REM > 0 ? REM : 1);
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

// DUAL (batches beyond one wavefront per SIMD): the two DPP-row pairs of the wavefront, which otherwise carry the SAME
// sequence, run sequences 2 blk and 2 blk + 1 -- the instruction stream is unchanged, so a full launch issues half the
// instructions; the second pair's records are addressed as the first pair's plus a 32-bit per-lane offset (an odd
// batch's last wavefront repeats sequence B - 1 in its second pair: same values to the same addresses).
template <int N, bool INHOMOG, bool DUAL = false>
__device__ __forceinline__ void lds_filter_1r_body(const LdsArgs& a, const int blk) {
  static_assert(N >= 1 && N <= TE_MAX_N && N + ((N - 1) >> 1) <= 14, "right-hand-side columns in lanes N..14 of two DPP rows");
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N);
  constexpr int J = (N + 1) / 2;          // slots holding rows 0..N-1 (row i = 2j + gl)
  constexpr int HL = 15;                  // lane of the h column
  // one wavefront per SIMD: this kernel runs next to the two-ended E-step kernel (see lds_estep_split.hpp, FILT)
  if constexpr (!DUAL) asm volatile("v_accvgpr_write_b32 a63, 0" ::: "a63");
  __shared__ double split_tile[2 * (N + 1) * 16];      // [DPP row pair][row 0..N][16 lanes]: re-replication of the next pivot block

  const int lane = threadIdx.x;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int gl = g & 1;                   // DPP row within its pair; the pairs (0,1) and (2,3) carry the SAME work (DUAL: two sequences)
  const int b0 = DUAL ? 2 * blk : blk;    // (uniform) the wavefront's first sequence
  const int sq = DUAL ? ((b0 + (g >> 1) < a.B) ? (g >> 1) : 0) : 0;
  const int b = b0 + sq;                  // this DPP-row pair's sequence
  const bool lead = DUAL ? gl == 0 : g == 0;      // the DPP row of a sequence that reports its P^-1 rows, factor and scalars
  const bool col = c < N;
  const int cc = col ? c : 0;
  const int T = a.T;

  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EH = (c == HL) ? 1.0 : 0.0;

  // ---- pair parameters (as chain A of the two-ended kernel) -----------------------------------------------------
  const double* q11 = a.J11 + (long)b * a.pair_seq_stride;
  const double* q22 = a.J22 + (long)b * a.pair_seq_stride;
  const double* q12 = a.J12 + (long)b * a.pair_seq_stride;
  const int xq = 2 * (c - N) + gl;                    // right-hand-side column of this lane (c >= N)
  const bool xok = c >= N && c < HL && xq < N;
  const int xx = xok ? xq : 0;
  double EX[N], NJ12c[N], Cc[J];
  auto load_pair = [&](int l, bool with_next_J11) {
    const long o = INHOMOG ? (long)l * N * N : 0;
    const long o1 = INHOMOG ? (long)(l + 1) * N * N : 0;
    static_for<0, N>([&](auto i) {
      const double rx = q12[o + i * N + xx], rc = q12[o + i * N + cc];
      EX[i] = xok ? -rx : E[i];
      NJ12c[i] = col ? rc : 0.0;
    });
    static_for<0, J>([&](auto j) {
      const int i = 2 * j + gl;
      const int ii = i < N ? i : 0;
      const double r22 = q22[o + ii * N + cc];
      const double r11 = with_next_J11 ? q11[o1 + ii * N + cc] : 0.0;
      Cc[j] = (col && i < N) ? -2.0 * (r22 + r11) : 0.0;
    });
  };
  double CcLast[J];                                   // homogeneous: the last pivot block has no J11 term
  static_for<0, J>([&](auto j) { CcLast[j] = 0.0; });
  if (!INHOMOG && T > 1) {
    load_pair(0, false);
    static_for<0, J>([&](auto j) { CcLast[j] = Cc[j]; });
    load_pair(0, true);
  }

  // An (replicated): lanes < N = pivot block of the next node without its node potential, lane 15 = h_pred
  double An[N];
  static_for<0, N>([&](auto i) {
    const double ij = a.init_J[i * N + cc], ih = a.init_h[i], j11 = T > 1 ? q11[i * N + cc] : 0.0;
    An[i] = col ? -2.0 * (ij + j11) : ((c == HL) ? ih : 0.0);
  });

  const double* nJ = a.node_J + ((long)b * T) * N + cc;
  const double* nh = a.node_h + ((long)b * T) * N + cc;
  double* zpage = a.ws + (long)b * ws_seq_doubles(N, T);   // [e_N | 0] rows the backward kernels' idle lanes read
  double* wsb = a.ws + (long)b0 * ws_seq_doubles(N, T) + ws_zpage_doubles(N);     // (uniform: the first sequence's records)
  if (lead && c < HS) zpage[c] = (c == N) ? 1.0 : 0.0;
  if (lead && c < PS) zpage[HS + c] = 0.0;
  double* ws2b = a.ws2 + ((long)b * T) * (N * N + N);
  const unsigned sqoff = 8u * (unsigned)(sq * ws_seq_doubles(N, T));             // bytes from the first sequence's records
  // hand-off store of register i: ONE unconditional instruction per register.  DPP row 0: lane c < N -> P^-1[i][c],
  // lane 15 -> c_i; the right-hand-side lanes of DPP rows 0 and 1 -> X[i][x]; every other lane -> the record's pad
  // entry (the H rows have one when N is even, the P^-1 rows when N is odd).
  constexpr int TRASH = (N % 2 == 0) ? N + 1 : N * HS + N;
  unsigned off[N];                                    // bytes from the step's record (uniform base + 32-bit lane offset)
  static_for<0, N>([&](auto i) {
    off[i] = sqoff + 8u * (unsigned)((lead && col) ? N * HS + i * PS + c
                                      : (((DUAL || g < 2) && xok) ? i * HS + xx : ((lead && c == HL) ? i * HS + N : TRASH)));
  });
  const int foff = (lead && col) ? c : -1;            // factor rows / pivots: the sequence's leading DPP row, lanes < N

  double qacc = 0.0, ldM = 1.0, vworst = -1.0;
  int ldE = 0;
  double Jo_n = nJ[0], ho_n = nh[0];
  for (int t = 0; t < T; ++t) {
    const bool last = (t == T - 1);
    const double JoX = col ? -2.0 * Jo_n : 1.0;
    double ho = ho_n;
    {
      const long tn = last ? t : t + 1;
      Jo_n = nJ[tn * N];
      ho_n = nh[tn * N];
    }
    if (INHOMOG && !last) load_pair(t, t + 1 < T - 1);

    double M[N], Bt[N];
    if (last) {
      // (the operand registers are overwritten in this branch, taken once: selecting E / EX per step costs N moves)
      asm volatile("; last step: no pair potential ahead (G = 0)");
      static_for<0, N>([&](auto i) { EX[i] = E[i]; });
    }
    static_for<0, N>([&](auto i) { M[i] = __builtin_fma(JoX, EX[i], An[i]); });
    dpp_fence(ho);
    static_for<0, N>([&](auto i) { mac_bc<i>(M[i], ho, EH); });      // lane 15: h_filt = h_pred + h_node
    // (pinned ahead of the elimination: hipcc otherwise sinks these below it and copies M to keep the old values)
    static_for<0, N>([&](auto k) { Bt[k] = __builtin_fma(-EH, M[k], NJ12c[k]); asm volatile("" : "+v"(Bt[k])); });
    dpp_fence(M);

    double* w = wsb + (long)t * WS;                   // (uniform)
    double* w2 = ws2b + (long)t * (N * N + N);
    double* fbase = foff >= 0 ? w2 + foff : w + TRASH;
    const long fstride = foff >= 0 ? N : 0;
    double vfull = col ? 0.0 : 1.0;
    // the generated, hand-scheduled elimination (gj1r_gen.hpp) returning every scaled pivot row: lanes j > k of
    // row k are L[j][k], the unit LDL' factor the backward sampler reads
    double RU[N];
#if SVAE_GJ_GENERATED
    gauss_jordan_1r_asm_rows<N>(M, E, qacc, vfull, RU);
#else
    gauss_jordan_1r_hook<N>(M, E, qacc, vfull, [&](auto k, double ru) { RU[k] = ru; });
#endif
    // Schur stage FIRST (it needs only the eliminated block and Bt), its slot-layout result on its way through LDS
    // while the step's stores are issued: the re-replication of the next pivot block as in the two-ended kernel
    // (lds_estep_twoend.hpp, round 4) instead of v_permlane16_swap shuffles behind the stores.
    if (!last) {
      double AnD[J];
      const bool next_last = (t + 1 == T - 1);
      if (!INHOMOG && next_last) {
        asm volatile("; next step is the last: its pivot block has no J11 term");
        static_for<0, J>([&](auto j) { Cc[j] = CcLast[j]; });
      }
      static_for<0, J>([&](auto j) { AnD[j] = Cc[j]; });
      asm volatile("s_nop 1");
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<N + j>(AnD[j], M[k], Bt[k]); });
      });
#if SVAE_FILTER_LDS_SPLIT
      double* sp = split_tile + (g >> 1) * (N + 1) * 16;
      __builtin_amdgcn_wave_barrier();
      static_for<0, J>([&](auto j) { sp[(2 * j + gl) * 16 + c] = AnD[j]; });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
      dpp_fence(AnD);
      static_for<0, J>([&](auto j) {
        if constexpr (2 * j + 1 < N) pair_split(AnD[j], An[2 * j], An[2 * j + 1]);
        else { double dummy; pair_split(AnD[j], An[2 * j], dummy); }
      });
#endif
    }
    static_for<0, N>([&](auto k) { fbase[k * fstride] = RU[k]; });
    // pivots d_c = -1 / vfull_c
    {
      const double pvv = -rcp_nr(col ? vfull : -1.0);
      double* pq = foff >= 0 ? w2 + N * N + foff : w + TRASH;
      *pq = pvv;
    }
    vworst = asm_max(vworst, vfull);
    ldM *= vfull;
    if ((t & 3) == 3) {
      asm volatile("; renormalise the determinant product");      // (keeps this a branch: as selects it is 7 instructions per step)
      ldE += __builtin_amdgcn_frexp_exp(ldM);
      ldM = __builtin_amdgcn_frexp_mant(ldM);
    }
    {
      char* wb = reinterpret_cast<char*>(w);            // uniform base + 32-bit lane offset (cf. lds_estep_twoend.hpp hand_off)
      static_for<0, N>([&](auto i) {
        asm volatile("" : "+v"(off[i]));
        *reinterpret_cast<double*>(wb + off[i]) = M[i] * vfull;
      });
    }
#if SVAE_FILTER_LDS_SPLIT
    if (!last) {
      const double* sp = split_tile + (g >> 1) * (N + 1) * 16 + c;
      static_for<0, N>([&](auto i) { An[i] = sp[i * 16]; });
      __builtin_amdgcn_wave_barrier();
    }
#endif
  }

  // ---- log-normaliser --------------------------------------------------------------------------------------------
  {
    const int ex = __builtin_amdgcn_frexp_exp(ldM);
    const double mant = __builtin_amdgcn_frexp_mant(ldM);
    double part = col ? (::log(fabs(mant)) + (double)(ldE + ex) * 0.6931471805599453094) : 0.0;
    if (c == HL) part = qacc;
    double z = 0.0;
    if (a.node_logZ) {
      for (int t = c; t < T; t += 16) z += a.node_logZ[(long)b * T + t];
    }
    if (INHOMOG) {
      const double* lz = a.logZ_pair + (a.pair_seq_stride ? (long)b * (T - 1) : 0);
      for (int t = c; t < T - 1; t += 16) z += lz[t];
    }
    double total = row_sum16(z) + 0.5 * row_sum16(part) + a.init_logZ[0];
    if (!INHOMOG && T > 1) total += (double)(T - 1) * a.logZ_pair[0];
    const bool reporter = DUAL ? (lane & 31) == 0 : lane == 0;             // one lane per sequence
    if (reporter) a.lognorm[b] = total;
    const bool lane_bad = col && !(vworst < 0.0);
    const unsigned long long bal = __ballot(lane_bad);
    const bool bad = (DUAL ? ((bal >> (lane & 32)) & 0xffffffffull) != 0 : bal != 0) || !(total == total);
    if (bad && reporter) {
      int old = *(volatile int32_t*)a.info;
      while (old == 0 || old > b + 1) {
        const int seen = atomicCAS(a.info, old, b + 1);
        if (seen == old) break;
        old = seen;
      }
    }
  }
}

template <int N, bool INHOMOG, bool DUAL = false>
__global__ __launch_bounds__(64) void lds_filter_1r_kernel(const LdsArgs a) {
  lds_filter_1r_body<N, INHOMOG, DUAL>(a, (int)blockIdx.x);
}

// The forward pass of a training step at small batches in ONE launch: workgroups 0 .. B-1 run the one-directional
// filter (the longer leg: dispatched first), workgroups B .. 2B-1 the two-ended E-step that also leaves the cross
// moments (CROSS).  Rounds 2 - 3 ran the two as separate kernels on two streams, forked and joined by events inside
// svae_lds_estep_f64: 262 us and 200 us of kernels took 0.31 ms on the caller's stream -- ~45 us of event / stream
// traffic.  One wavefront per SIMD as before (both bodies touch a high AGPR); a workgroup takes one branch, whole.
// CROSS: the E-step body also stores the cross moments (e.ws3) -- the hand-off of the VJP; without it (factor kept for the
// sampler only) the plain two-ended body runs.
template <int N, bool INHOMOG, bool CROSS>
__global__ __launch_bounds__(64) void lds_forward_pair_kernel(const LdsArgs f, const LdsArgs e) {
  if ((int)blockIdx.x < f.B) lds_filter_1r_body<N, INHOMOG>(f, (int)blockIdx.x);
  else lds_estep_twoend_body<N, INHOMOG, !INHOMOG, false, CROSS, false>(e, (int)blockIdx.x - f.B);
}

template <int N>
static int launch_forward_pair(const LdsArgs& f, const LdsArgs& e, bool inhomog, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    dim3 grid(f.B + e.B), block(64);
    const bool cross = e.ws3 != nullptr;
    if (inhomog && cross) hipLaunchKernelGGL((lds_forward_pair_kernel<N, true, true>), grid, block, 0, stream, f, e);
    else if (inhomog) hipLaunchKernelGGL((lds_forward_pair_kernel<N, true, false>), grid, block, 0, stream, f, e);
    else if (cross) hipLaunchKernelGGL((lds_forward_pair_kernel<N, false, true>), grid, block, 0, stream, f, e);
    else hipLaunchKernelGGL((lds_forward_pair_kernel<N, false, false>), grid, block, 0, stream, f, e);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

template <int N>
static int launch_filter_1r(const LdsArgs& a, bool inhomog, hipStream_t stream) {
  if constexpr (N <= TE_MAX_N) {
    dim3 grid(a.B), grid2((a.B + 1) / 2), block(64);
    // beyond one wavefront per SIMD a second wavefront of the same sequence count only adds instructions: two sequences
    // per wavefront (the latency of a step is the same instruction stream either way)
    const bool dual = a.B > FILTER_1R_DUAL_MIN_B;
    if (inhomog && dual) hipLaunchKernelGGL((lds_filter_1r_kernel<N, true, true>), grid2, block, 0, stream, a);
    else if (dual) hipLaunchKernelGGL((lds_filter_1r_kernel<N, false, true>), grid2, block, 0, stream, a);
    else if (inhomog) hipLaunchKernelGGL((lds_filter_1r_kernel<N, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lds_filter_1r_kernel<N, false>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  } else {
    return -3;
  }
}

}  // namespace svae
