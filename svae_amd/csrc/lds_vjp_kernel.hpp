// lds_vjp_kernel.hpp -- reverse-mode (VJP) sweeps of the LDS E-step [+ backward sampler] on MI355X.
//
// What it replaces (reference = mattjj/svae, /root/reference), composed exactly as the reference wires
// its autograd primitives (svae/lds/lds_inference.py:26-39):
//   natural_filter_grad            svae/lds/cython_lds_inference.pyx:92-145   (+ pxd:84-120,146-162,190-204)
//   natural_smoother_general_grad  svae/lds/cython_lds_inference.pyx:236-306  (+ pxd:222-238,298-426)
//   natural_sample_backward_grad   svae/lds/cython_lds_inference.pyx:357-409  (+ pxd:456-487,510-528)
//   and the linear-algebra VJP helpers of svae/cython_linalg_grads.pxd:10-118.
// Not a translation: it differentiates THIS library's forward algorithm (lds_estep_kernel.hpp), whose
// per-step state is (P^-1, H = [P^-1 J12 | c]), the moment recursion S~_t = G~ S~_{t+1} G~' + diag(P^-1,0)
// and the sampler x_t = c - (P^-1 J12) x_{t+1} + chol(P)^-T eps.  Adjoint algebra (prototype with a
// finite-difference and reference check: tools/proto/vjp_proto.py), S^ = adjoint of S~ kept symmetric:
//   sweep 1, t = 0 .. T-1 (reverse of the smoother / sampler recursions):
//     S^ += direct cotangents (E[x_t] in row/column N, diag E[x x'] on the diagonal)
//     G^ = 2 S^ W~_t'                      -> Xbar_t = -G^[:n,:n],  cbar_t = G^[:n,n]
//     Pinvbar_t = S^[:n,:n];               S^ <- G~' S^ G~
//     xhat_t = g_x_t - X_{t-1}' xhat_{t-1}; cbar_t += sum_s xhat; Xbar_t -= sum_s xhat x_{t+1}'
//     Pbar_t(direct) = -U Lh U',  U = chol(P)^-T,  Lh = lower-half(sum_s eps_s (U' xhat_s)')
//     statistics cotangents (SLDS: hmm_vlb depends on E_init and on the per-step E_pair,
//     svae/models/slds_svae.py:131-155,300):  Q_t = d/dE[x_t x_t'] goes into S^[:n,:n] symmetrised;
//     Cb_t = d/dE[x_t x_{t+1}'] acts through W~_t = S~_{t+1} G~':  G^[:n] += Cb_t S~_{t+1}[:n],
//     S^_{t+1} += sym([Cb_t' 0; 0 0] G~)   (S~_{t+1} is read back from the forward outputs)
//   sweep 2, t = T-1 .. 0 (reverse of the filter recursion):
//     [Xbar | cbar]_t -= J12 [Abar | hbar]_{t+1}
//     Bbar = P^-1 [Xbar | cbar];   hfbar = Bbar[:,n] + g c
//     Pbar = -P^-1 Pinvbar P^-1 - Bbar H' - 1/2 g (c c' + P^-1) + Pbar(direct)
//     g_node_J_t = -2 diag(Pbar);  g_node_h_t = hfbar;  [Abar | hbar]_t = [Pbar | hfbar]
// Scratch between the sweeps (lds_args.hpp: vjp_step_doubles per sequence-step): G^ (n x (n+1)); in two-role launches
// the sampler role's share of G^ as its rank-S factors [xhat_s | x_{t+1,s}] (2 n doubles per sample instead of an
// n x (n+1) matrix -- sweep 2 forms the product, n DPP multiply-adds per sample); -P^-1 Pinvbar P^-1 as its lower
// triangle (symmetric); Pbar(direct).  The sweeps are bandwidth-bound at large batches, so the record is what they cost.
// Layout: as the packed E-step kernel (one DPP row per sequence, lane = column, fused DPP FMAs).
// Hazards: every product stage fences its DPP-read operand array once (mm_ab / mm_atb), the few
// stand-alone broadcasts fence theirs; `make audit` checks the ISA of every latent dimension.
// Variants (launch_vjp picks by batch size; all run the same recursion, tests/test_vjp_hip.py compares them):
//   packed          lds_vjp_sweep{1,2}_kernel: four sequences per wavefront, register prefetch; any batch, any S,
//                   statistics cotangents (STATC)
//   + producers     lds_vjp_sweep{1,2}_prod_kernel (B <= prod_max_b): producer wavefronts stream the records through an
//                   LDS ring; sweep 1 also has helper wavefronts for the work that does not feed the recursion
//   one sequence    lds_vjp_sweep2_s4_kernel, lds_vjp_sweep1_s4_kernel (B <= VJP_S4_MAX_B): one sequence per consumer
//   per wavefront   wavefront, product stages split over its four DPP rows
#pragma once
#ifndef SVAE_S1_PAIR
#define SVAE_S1_PAIR 0          // sweep 1, packed producer launch with samples: two role-0 consumer wavefronts in the row-pair layout (0: one, four whole sequences).  Measured, not kept: DESIGN 7.2
#endif
#ifndef SVAE_S1_NH1
#define SVAE_S1_NH1 2           // sweep 1, packed producer launch: helper wavefronts of the sampler-adjoint role (2 or 3).  3: measured, not kept: DESIGN 7.2
#endif
#ifndef SVAE_S2_PROD_PX
#define SVAE_S2_PROD_PX 1       // sweep 2, one sequence per consumer: the PRODUCER wavefront computes the recursion-free product P^-1 [-G^ | G^[:,n]] (0: consumer; A/B)
#endif
#ifndef SVAE_S2_LDS_GATHER
#define SVAE_S2_LDS_GATHER 1   // sweep 2, one sequence per consumer: the adjoint for the next step all-gathered through LDS (0: shuffles; A/B)
#endif
#include "lds_estep_kernel.hpp"
#include "lds_estep_twoend_s4.hpp"   // quad_gather

namespace svae {

// OUT[i] += sum_{k<KN} (+/-) bcast_k(A[i]) * B[k]   for i in [0, IN)      (OUT = A B)
template <int IN, int KN, bool NEG, int MA, int MB, int MO>
__device__ __forceinline__ void mm_ab(double (&out)[MO], const double (&A)[MA], const double (&Bt)[MB]) {
  dpp_fence(const_cast<double(&)[MA]>(A));     // A is the DPP-read operand: order its producers first
  static_for<0, KN>([&](auto k) {
    static_for<0, IN>([&](auto i) { mac_bc<k, NEG>(out[i], A[i], Bt[k]); });
  });
}
// OUT[i] += sum_{k<KN} (+/-) bcast_i(A[k]) * B[k]   for i in [0, IN)      (OUT = A' B)
template <int IN, int KN, bool NEG, int MA, int MB, int MO>
__device__ __forceinline__ void mm_atb(double (&out)[MO], const double (&A)[MA], const double (&Bt)[MB]) {
  dpp_fence(const_cast<double(&)[MA]>(A));
  static_for<0, KN>([&](auto k) {
    static_for<0, IN>([&](auto i) { mac_bc<i, NEG>(out[i], A[k], Bt[k]); });
  });
}

// transpose an N-row column-layout tile inside its DPP row through LDS (tab: 256 doubles per row)
template <int N>
__device__ __forceinline__ void transpose_tile(double* tab, int c, const double (&a)[N], double (&at)[N]) {
  __builtin_amdgcn_wave_barrier();
  static_for<0, N>([&](auto i) { tab[c * 16 + i] = a[i]; });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  static_for<0, N>([&](auto i) { at[i] = tab[i * 16 + c]; });
  __builtin_amdgcn_wave_barrier();
}

// Cache warming (sweep 2, small batches): the sweep is serial in t and each step's operands (a few KB per sequence written
// long ago by another kernel) come from HBM, ~2 us away; a one-step register prefetch cannot hide
// that behind ~1 us of arithmetic.  Each DPP row instead touches one 128-byte line per lane of the
// records of a step several iterations ahead; the values are folded into a dummy one iteration later
// (by then, vmcnt being in-order, they have arrived) and the real loads find the lines in L2.
template <int LINES>
__device__ __forceinline__ double touch_lines(const double* rec, int c, long doubles) {
  double s = 0.0;
  static_for<0, (LINES + 15) / 16>([&](auto j) {
    const long off = ((long)(c + 16 * j) * 16);
    s += rec[off < doubles ? off : 0];
  });
  return s;
}
constexpr int VJP_AHEAD = 4;
constexpr int SPRE = 2;      // samples whose x_{t+1} / eps rows are requested at the top of the sampler-adjoint role

// f(integral_constant<s>) for s = S0, S0+1, .. while s < S (S <= 16 at run time): leaves at the first s >= S
template <int S0, class F>
__device__ __forceinline__ void for_samples(int S, F&& f) {
  if constexpr (S0 < 16) {
    if (S0 < S) {
      f(std::integral_constant<int, S0>{});
      for_samples<S0 + 1>(S, f);
    }
  }
}

// ---- sweep 1: smoother + sampler adjoints, forward in time ----------------------------------------
// The two adjoint chains of this sweep are independent (the S^ recursion of the smoother; the xhat
// recursion + noise adjoint of the sampler): with samples they run as two ROLES in separate
// workgroups (blockIdx.x & 1), each writing its own share of G^ -- the sweep is latency-bound on one
// wavefront's instruction stream, and small batches leave most SIMDs idle.
// PROD (small batches, S <= 4): as in sweep 2 below, four more wavefronts of the workgroup are producers, one per
// sequence; each gathers the step's operands of the workgroup's role (role 0: E-step record, W~, direct cotangents;
// role 1: H, the LDL' factor, sample cotangents, x_{t+1}, eps) PD steps ahead and publishes them into a three-slot LDS
// ring; the consumer wavefront reads LDS only.  One s_barrier per step.
// The consumer keeps ONLY the recursions (role 0: S^ and G^; role 1: xhat and its share of G^).  What a step computes
// without feeding the next one -- role 0: -P^-1 Pinvbar P^-1; role 1: the whole noise adjoint -- goes to HELPER
// wavefronts of the workgroup (role 0: one; role 1: two, taking alternate steps, each job spread over two barrier
// intervals): the consumer leaves Pinvbar_t / xhat_t in an LDS mailbox, the helper picks it up one barrier later
// together with the step's record, which is still in the ring.
// Barriers are numbered 0 .. T and every wavefront of the workgroup executes each exactly once:
//   barrier t (t < T): step t's record is in ring slot t%3;  barrier t+1: the consumer's mailbox of step t is written.
constexpr int VJP_PROD_MAX_S = 4;
static_assert(VJP_PROD_MAX_S <= VJP_SPLIT_MAX_S, "the producer variants run as two roles");
// consumer, 4 producers, 2 helpers (+ 1 with the experimental SVAE_S1_PAIR / SVAE_S1_NH1 = 3 layouts: a second role-0
// consumer / a third role-1 helper).  Wavefronts w and w + 4 of a workgroup share a SIMD (tools/scratch/hwid.hip).
constexpr int VJP_S1_WAVES = (SVAE_S1_PAIR || SVAE_S1_NH1 == 3) ? 8 : 7;
template <int N> constexpr int vjp_s1_pieces() {
  constexpr int HS = ws_h_stride(N), WS = ws_step_doubles(N);
  constexpr int r0 = WS + (N + 1) * HS + 2 * N, r1 = N * HS + N * N + N + 3 * VJP_PROD_MAX_S * N;
  return ((r0 > r1 ? r0 : r1) + 63) / 64;
}
// ROLE: 0 = smoother adjoint, 1 = sampler adjoint (SPLIT: the workgroup's role, chosen by the caller -- each role is
// its own instantiation so that the register allocation is the larger of the two, not their union), 2 = both
// The smoother adjoint of sweep 1 with ONE SEQUENCE PER WORKGROUP (no sample cotangents, B <= VJP_S4_MAX_B): wavefront 0 = consumer,
// 1 = producer, 2 = helper; ring (3 slots of one record), mailbox (2 slots) and barrier protocol are those of
// lds_vjp_sweep1_body.  The consumer's and the helper's four DPP rows split every product stage by output row (row i
// of a tile in DPP row i & 3, slot i >> 2; cf. lds_vjp_sweep2_s4_kernel): 99 instead of 352 DPP multiply-adds per
// consumer step, 60 instead of 200 per helper step, one all-gather each.  The transposed product S^ <- G~' M needs
// column i of G~ per output row: row i of G~' is read from the record with the lane as the row index instead.
template <int N>
__device__ __forceinline__ void s1_role0_wg(const VjpArgs& a, double* ring, double* mail, const int seq) {
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N), AS = vjp_step_doubles(N);
  constexpr int W3 = (N + 1) * HS;
  constexpr int KT = vjp_s1_pieces<N>(), SLOT = KT * 64, MSLOT = N * 16, PD = 6;
  constexpr int J = (N + 3) / 4;             // slots holding rows 0..N-1 (row i = 4j + r)
  constexpr int J1 = (N + 4) / 4;            // slots holding rows 0..N
  constexpr int NS = N >> 2, NR = N & 3;     // row N: slot NS of DPP row NR
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, r = lane >> 4;
  const long b = seq;
  const int T = a.T;
  const bool col = c < N, colN = c <= N;
  const int cN = colN ? c : 0, ccl = col ? c : 0;
  if (wv >= 3) return;
  if (wv == 1) {
    // ---- producer: gathers the step's record [E-step record | W~ | g_x | g_diagxx] PD steps ahead -----------------
    const double* wsq = a.ws + b * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
    const double* base[KT];
    int stp[KT], off[KT];
    static_for<0, KT>([&](auto k) { base[k] = wsq; stp[k] = 0; off[k] = 0; });
    int start = 0;
    auto seg = [&](const double* p, int len, int stride) {
      static_for<0, KT>([&](auto k) {
        const int f = k * 64 + lane - start;
        if (f >= 0 && f < len) { base[k] = p + f; stp[k] = stride; }
      });
      start += len;
    };
    seg(wsq, WS, WS);
    seg(a.ws3 + b * T * W3, W3, W3);
    seg(a.g_x ? a.g_x + b * T * N : wsq, N, a.g_x ? N : 0);
    seg(a.g_diagxx ? a.g_diagxx + b * T * N : wsq, N, a.g_diagxx ? N : 0);
    GStage<KT> s0, s1, s2, s3, s4, s5;                  // stage of step t: t % 6
    static_assert(PD == 6, "six named stages");
    gather_issue<KT>(s0, base, stp, off, 0, T);
    gather_issue<KT>(s1, base, stp, off, 1, T);
    gather_issue<KT>(s2, base, stp, off, 2, T);
    gather_issue<KT>(s3, base, stp, off, 3, T);
    gather_issue<KT>(s4, base, stp, off, 4, T);
    gather_issue<KT>(s5, base, stp, off, 5, T);
    gather_publish<KT>(s0, ring, lane);
    gather_issue<KT>(s0, base, stp, off, 6, T);
    lds_barrier();                                     // barrier 0: step 0 is in slot 0
#define SVAE_PROD_STEP(sg, t)                                                 \
    {                                                                           \
      gather_publish<KT>(sg, ring + (((t) + 1) % 3) * SLOT, lane);              \
      gather_issue<KT>(sg, base, stp, off, (t) + 1 + PD, T);                    \
      lds_barrier();                                                            \
    }
    int t0 = 0;
    for (; t0 + 6 < T; t0 += PD) {                     // (no branch inside the steady-state loop)
      SVAE_PROD_STEP(s1, t0)
      SVAE_PROD_STEP(s2, t0 + 1)
      SVAE_PROD_STEP(s3, t0 + 2)
      SVAE_PROD_STEP(s4, t0 + 3)
      SVAE_PROD_STEP(s5, t0 + 4)
      SVAE_PROD_STEP(s0, t0 + 5)
    }
    if (t0 + 1 < T) SVAE_PROD_STEP(s1, t0)
    if (t0 + 2 < T) SVAE_PROD_STEP(s2, t0 + 1)
    if (t0 + 3 < T) SVAE_PROD_STEP(s3, t0 + 2)
    if (t0 + 4 < T) SVAE_PROD_STEP(s4, t0 + 3)
    if (t0 + 5 < T) SVAE_PROD_STEP(s5, t0 + 4)
#undef SVAE_PROD_STEP
    lds_barrier();                                     // barrier T
    return;
  }
  double rm[J];
  int ri[J1];
  static_for<0, J>([&](auto j) { rm[j] = (4 * j + r < N) ? 1.0 : 0.0; });
  static_for<0, J1>([&](auto j) { const int i = 4 * j + r; ri[j] = i <= N ? i : 0; });
  if (wv == 2) {
    // ---- helper: the part of Pbar_t that does not depend on sweep 2,  -P^-1 Pinvbar P^-1 -----------------------------
    lds_barrier();                                     // barrier 0
    for (int t = 0; t < T; ++t) {
      lds_barrier();                                   // barrier t+1: Pinvbar_t is in the mailbox
      const double* rec = ring + (t % 3) * SLOT;
      const double* mb = mail + (t & 1) * MSLOT;
      double PiR[N], PiD[J], PibD[J], T1[J], Pbp[J];
      static_for<0, N>([&](auto k) { PiR[k] = rec[N * HS + k * PS + ccl]; });                 // rows of P^-1, replicated
      static_for<0, J>([&](auto j) {
        const int i = 4 * j + r < N ? 4 * j + r : 0;
        PiD[j] = rec[N * HS + i * PS + ccl] * rm[j];                                          // my rows
        PibD[j] = mb[i * 16 + c] * rm[j];
        T1[j] = 0.0; Pbp[j] = 0.0;
      });
      dpp_fence(PibD);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k>(T1[j], PibD[j], PiR[k]); });
      });
      dpp_fence(T1);
      double T1R[4 * J];
      static_for<0, J>([&](auto j) { quad_gather(T1[j], T1R[4 * j], T1R[4 * j + 1], T1R[4 * j + 2], T1R[4 * j + 3]); });
      dpp_fence(PiD);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k, true>(Pbp[j], PiD[j], T1R[k]); });
      });
      double* ad = a.adj + (b * T + t) * AS;
      static_for<0, J>([&](auto j) {                                      // lower triangle (Pbp is symmetric)
        const int i = 4 * j + r;
        if (i < N && c <= i) ad[vjp_pbp_off(N) + i * (i + 1) / 2 + c] = Pbp[j];
      });
    }
    return;
  }
  // ---- consumer ------------------------------------------------------------------------------------------------------
  const double EN = (c == N) ? 1.0 : 0.0;
  const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);
  const double cm = col ? 1.0 : 0.0;
  const bool own_N = (r == NR);
  const bool has_gx = a.g_x != nullptr, has_gd = a.g_diagxx != nullptr;
  double ED[J], sgi[J1], dN[J1];
  static_for<0, J>([&](auto j) { ED[j] = (c == 4 * j + r && c < N) ? 1.0 : 0.0; });
  static_for<0, J1>([&](auto j) {
    const int i = 4 * j + r;
    sgi[j] = i < N ? -1.0 : (i == N ? 1.0 : 0.0);       // sign of column i of G~ (0: no such row)
    dN[j] = (i == N) ? EN : 0.0;                        // row N of G~ is e_N: G~[N][i] = (i == N), in lane N
  });
  double Sh[J1];
  static_for<0, J1>([&](auto j) { Sh[j] = 0.0; });
  for (int t = 0; t < T; ++t) {
    lds_barrier();                                            // barrier t: step t is in slot t % 3
    const double* rec = ring + (t % 3) * SLOT;
    double WT[N + 1], Gc[N + 1], GcT[J1], gxs[J];
    load_row<N + 1>(rec + WS + cN * HS, WT);                    // row c of W~: WT[k][c] = W~[c][k]
    static_for<0, N>([&](auto k) { Gc[k] = rec[k * HS + cN]; });
    static_for<0, J1>([&](auto j) { GcT[j] = rec[ccl * HS + ri[j]]; });        // lane k: H[k][i]
    static_for<0, J>([&](auto j) { gxs[j] = rec[WS + W3 + (4 * j + r < N ? 4 * j + r : 0)]; });
    const double gxl = rec[WS + W3 + ccl], gdl = rec[WS + W3 + N + ccl];
    static_for<0, N + 1>([&](auto k) { WT[k] = colN ? 2.0 * WT[k] : 0.0; });  // 2 W~'
    static_for<0, N>([&](auto k) { Gc[k] *= sg; });                            // G~ row k (replicated)
    Gc[N] = EN;
    static_for<0, J1>([&](auto j) { GcT[j] = __builtin_fma(cm * sgi[j], GcT[j], dN[j]); });   // row i of G~'
    // S^ += direct cotangents, symmetrised: S^[i][N] += g_x[i] / 2, S^[N][c] += g_x[c] / 2, S^[i][i] += g_diagxx[i]
    const double gx = (has_gx && col) ? 0.5 * gxl : 0.0, gd = (has_gd && col) ? gdl : 0.0;
    static_for<0, J>([&](auto j) {
      Sh[j] = __builtin_fma(EN * rm[j], has_gx ? 0.5 * gxs[j] : 0.0, Sh[j]);
      Sh[j] = __builtin_fma(gd, ED[j], Sh[j]);
    });
    Sh[NS] += own_N ? gx : 0.0;
    dpp_fence(Sh);
    // G^ rows i < N:  2 S^ W~'
    double Gb[J];
    static_for<0, J>([&](auto j) { Gb[j] = 0.0; });
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J>([&](auto j) { mac_bc<k>(Gb[j], Sh[j], WT[k]); });
    });
    // Pinvbar = S^[:n,:n] (before the propagation) for the helper wavefront
    {
      double* mb = mail + (t & 1) * MSLOT;
      static_for<0, J>([&](auto j) { if (4 * j + r < N) mb[(4 * j + r) * 16 + c] = Sh[j] * cm; });
    }
    // S^ <- G~' (S^ G~)
    double M[J1];
    static_for<0, J1>([&](auto j) { M[j] = 0.0; });
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k>(M[j], Sh[j], Gc[k]); });
    });
    dpp_fence(M);
    double MR[4 * J1];
    static_for<0, J1>([&](auto j) { quad_gather(M[j], MR[4 * j], MR[4 * j + 1], MR[4 * j + 2], MR[4 * j + 3]); });
    double Sn[J1];
    static_for<0, J1>([&](auto j) { Sn[j] = 0.0; });
    dpp_fence(GcT);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k>(Sn[j], GcT[j], MR[k]); });
    });
    static_for<0, J1>([&](auto j) { Sh[j] = Sn[j]; });
    double* ad = a.adj + (b * T + t) * AS;
    if (colN) static_for<0, J>([&](auto j) { if (4 * j + r < N) ad[(4 * j + r) * HS + c] = Gb[j]; });
  }
  lds_barrier();                                              // barrier T
}

// Consumer of the smoother adjoint for TWO of the workgroup's four sequences (packed producer launch with samples: the
// role-0 workgroup runs two of these, wavefronts 0 and 7): each sequence on a PAIR of DPP rows that split every product
// stage by output row (row i of a tile in DPP row i & 1 of the pair, slot i >> 1) -- 2 x 66 DPP multiply-adds and one
// row exchange (v_permlane16_swap) per step instead of the 252 of one wavefront holding four whole sequences, which was
// the longest instruction stream of the sweep (354 instructions per step).  Ring, mailbox and barrier protocol are
// those of lds_vjp_sweep1_body (the helpers and producers are unchanged: rows of S^ go to the mailbox of the sequence).
template <int N>
__device__ __forceinline__ void s1_role0_pair_consumer(const VjpArgs& a, const double* ring, double* mail, const int grp,
                                                       const int half) {
  constexpr int HS = ws_h_stride(N), WS = ws_step_doubles(N);
  constexpr int W3 = (N + 1) * HS;
  constexpr int KT = vjp_s1_pieces<N>(), REC = KT * 64, SLOT = 4 * REC, MSLOT = 4 * N * 16;
  constexpr int J = (N + 1) / 2;             // slots holding rows 0..N-1 (row i = 2j + q)
  constexpr int J1 = (N + 2) / 2;            // slots holding rows 0..N
  constexpr int NS = N >> 1, NR = N & 1;     // row N: slot NS of DPP row NR of the pair
  const int lane = threadIdx.x & 63, c = lane & 15, q = (lane >> 4) & 1, row = 2 * half + (lane >> 5);
  const int T = a.T;
  const bool col = c < N, colN = c <= N;
  const int cN = colN ? c : 0, ccl = col ? c : 0;
  const double* ringrow = ring + row * REC;
  double* mailrow = mail + row * N * 16;
  double rm[J];
  int ri[J1];
  static_for<0, J>([&](auto j) { rm[j] = (2 * j + q < N) ? 1.0 : 0.0; });
  static_for<0, J1>([&](auto j) { const int i = 2 * j + q; ri[j] = i <= N ? i : 0; });
  const double EN = (c == N) ? 1.0 : 0.0;
  const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);
  const double cm = col ? 1.0 : 0.0;
  const bool own_N = (q == NR);
  const bool has_gx = a.g_x != nullptr, has_gd = a.g_diagxx != nullptr;
  double ED[J], sgi[J1], dN[J1];
  static_for<0, J>([&](auto j) { ED[j] = (c == 2 * j + q && c < N) ? 1.0 : 0.0; });
  static_for<0, J1>([&](auto j) {
    const int i = 2 * j + q;
    sgi[j] = i < N ? -1.0 : (i == N ? 1.0 : 0.0);       // sign of column i of G~ (0: no such row)
    dN[j] = (i == N) ? EN : 0.0;                        // row N of G~ is e_N: G~[N][i] = (i == N), in lane N
  });
  double Sh[J1];
  static_for<0, J1>([&](auto j) { Sh[j] = 0.0; });
  for (int t = 0; t < T; ++t) {
    lds_barrier();                                            // barrier t: step t is in slot t % 3
    const double* rec = ringrow + (t % 3) * SLOT;
    double Gc[N + 1], GcT[J1], gxs[J];
    static_for<0, N>([&](auto k) { Gc[k] = rec[k * HS + cN]; });
    static_for<0, J1>([&](auto j) { GcT[j] = rec[ccl * HS + ri[j]]; });        // lane k: H[k][i]
    static_for<0, J>([&](auto j) { gxs[j] = rec[WS + W3 + (2 * j + q < N ? 2 * j + q : 0)]; });
    const double gxl = rec[WS + W3 + ccl], gdl = rec[WS + W3 + N + ccl];
    static_for<0, N>([&](auto k) { Gc[k] *= sg; });                            // G~ row k (replicated)
    Gc[N] = EN;
    static_for<0, J1>([&](auto j) { GcT[j] = __builtin_fma(cm * sgi[j], GcT[j], dN[j]); });   // row i of G~'
    // S^ += direct cotangents, symmetrised: S^[i][N] += g_x[i] / 2, S^[N][c] += g_x[c] / 2, S^[i][i] += g_diagxx[i]
    const double gx = (has_gx && col) ? 0.5 * gxl : 0.0, gd = (has_gd && col) ? gdl : 0.0;
    static_for<0, J>([&](auto j) {
      Sh[j] = __builtin_fma(EN * rm[j], has_gx ? 0.5 * gxs[j] : 0.0, Sh[j]);
      Sh[j] = __builtin_fma(gd, ED[j], Sh[j]);
    });
    Sh[NS] += own_N ? gx : 0.0;
    // rows 0..N-1 of S^ (before the propagation) for the helper wavefronts: Pinvbar = S^[:n,:n], G^ = 2 S^ W~'
    {
      double* mb = mailrow + (t & 1) * MSLOT;
      static_for<0, J>([&](auto j) { if (2 * j + q < N) mb[(2 * j + q) * 16 + c] = Sh[j]; });
    }
    // S^ <- G~' (S^ G~)
    double M[J1];
    static_for<0, J1>([&](auto j) { M[j] = 0.0; });
    dpp_fence(Sh);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k>(M[j], Sh[j], Gc[k]); });
    });
    dpp_fence(M);
    double MR[2 * J1];
    static_for<0, J1>([&](auto j) { pair_split(M[j], MR[2 * j], MR[2 * j + 1]); });
    double Sn[J1];
    static_for<0, J1>([&](auto j) { Sn[j] = 0.0; });
    dpp_fence(GcT);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J1>([&](auto j) { mac_bc<k>(Sn[j], GcT[j], MR[k]); });
    });
    static_for<0, J1>([&](auto j) { Sh[j] = Sn[j]; });
  }
  lds_barrier();                                              // barrier T
}

// grp: index of the workgroup's group of four sequences
template <int N, bool SAMP, bool STATC, bool SPLIT, bool PROD, int ROLE>
__device__ __forceinline__ void lds_vjp_sweep1_body(const VjpArgs& a, double* tabs, double* ring, double* mail,
                                                    const int grp) {
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N);
  constexpr int AS = vjp_step_doubles(N);
  constexpr int W3 = (N + 1) * HS, R1 = N * HS + N * N + N;
  constexpr int KT = vjp_s1_pieces<N>(), REC = KT * 64, SLOT = 4 * REC, PD = 6;
  static_assert(!PROD || (!STATC && (SPLIT || !SAMP)), "producers: one role per workgroup, no statistics cotangents");
  static_assert(ROLE == 2 ? !(SAMP && SPLIT) : (ROLE == 0 || (SAMP && SPLIT)), "role / split mismatch");
  constexpr int MSLOT = 4 * N * 16;            // mailbox slot: per DPP row, N registers of 16 lanes
  const int lane = threadIdx.x & 63;
  if constexpr (PROD) {
    const int wv = threadIdx.x >> 6;
    if constexpr (SAMP && ROLE == 0 && SVAE_S1_PAIR) {
      // two consumer wavefronts (0 and 7), two sequences each, in the row-pair layout
      if (wv == 0 || wv == 7) { s1_role0_pair_consumer<N>(a, ring, mail, grp, wv == 7 ? 1 : 0); return; }
    }
    if (wv >= 5) {
      // ---- helper wavefronts ---------------------------------------------------------------------------------------
      constexpr int NH = ROLE == 1 ? SVAE_S1_NH1 : 2;   // role 0: Pbar share | G^ rows;  role 1: the noise adjoint of every NH-th step
      const int h = wv - 5, T = a.T;
      if (h >= NH) return;
#ifdef SVAE_S1_SKIP_H
      if (ROLE != 1 && ((SVAE_S1_SKIP_H >> h) & 1)) return;        // (timing experiment)
#endif
      const int c = lane & 15, row = lane >> 4;
      const int brow = grp * 4 + row;
      const bool valid = brow < a.B;
      const long b = valid ? brow : a.B - 1;
      const bool col = c < N;
      const int ccl = col ? c : 0;
      const double* ringrow = ring + row * REC;
      const double* mailrow = mail + row * N * 16;
      if constexpr (ROLE != 1) {
        const double cm = col ? 1.0 : 0.0;
        const bool colN = c <= N;
        const int cN = colN ? c : 0;
        lds_barrier();                                                 // barrier 0
        for (int t = 0; t < T; ++t) {
          lds_barrier();                                               // barrier t+1: rows 0..N-1 of S^_t are in the mailbox
          const double* rec = ringrow + (t % 3) * SLOT;
          const double* mb = mailrow + (t & 1) * MSLOT;
          double* ad = a.adj + (b * T + t) * AS;
          if (h == 0) {
            // the part of Pbar_t that does not depend on sweep 2:  -P^-1 Pinvbar P^-1,  Pinvbar = S^[:n,:n]
            double Pi[N], Pib[N], T1[N], Pbp[N];
            static_for<0, N>([&](auto i) {
              Pi[i] = rec[N * HS + i * PS + ccl];
              Pib[i] = mb[i * 16 + c] * cm;
              T1[i] = 0.0; Pbp[i] = 0.0;
            });
            mm_ab<N, N, false>(T1, Pib, Pi);
            mm_ab<N, N, true>(Pbp, Pi, T1);
            if (valid) static_for<0, N>([&](auto i) { if (c <= i) ad[vjp_pbp_off(N) + i * (i + 1) / 2 + c] = Pbp[i]; });
          } else {
            // G^ rows i < N:  2 S^ W~'   (lanes 0..N; with samples: this role's share)
            double Sm[N], WT[N + 1], Gb[N];
            load_row<N + 1>(rec + WS + cN * HS, WT);
            static_for<0, N>([&](auto i) { Sm[i] = mb[i * 16 + c]; Gb[i] = 0.0; });
            static_for<0, N + 1>([&](auto k) { WT[k] = colN ? 2.0 * WT[k] : 0.0; });
            mm_ab<N, N + 1, false>(Gb, Sm, WT);
            if (valid && colN) static_for<0, N>([&](auto i) { ad[i * HS + c] = Gb[i]; });
          }
        }
      } else {
        // noise adjoint:  Pbar_t(direct) = -U (Lh U')  with  U = L^-T D^-1/2,  Lh from E' = sum_s eps_s z_s'
        const int S = a.S, SN = S * N;
        double* tab = tabs + (h * 4 + row) * 256;
        double E[N], mU[N];
        static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
        static_for<0, N>([&](auto j) { mU[j] = (c > j) ? 1.0 : ((c == j) ? 0.5 : 0.0); });
        for (int k = 0; k <= h && k <= T; ++k) lds_barrier();          // barriers 0 .. h
        for (int t = h; t < T; t += NH) {
          lds_barrier();                                               // barrier t+1: xhat_t is in the mailbox
          const double* rec = ringrow + (t % 3) * SLOT;
          const double* mb = mailrow + (t & 1) * MSLOT;
          double xh[N], R[N], epr[VJP_PROD_MAX_S];
          static_for<0, N>([&](auto k) { xh[k] = mb[k * 16 + c]; R[k] = rec[N * HS + k * N + ccl]; });
          const double pvv = rec[N * HS + N * N + ccl];
          static_for<0, VJP_PROD_MAX_S>([&](auto s) {
            const int sq = s < S ? (int)s : 0;
            epr[s] = rec[R1 + 2 * SN + sq * N + ccl];
          });
          double U[N];
          const double dis = col ? rsqrt_nr(pvv) : 0.0;
          static_for<0, N>([&](auto k) { U[k] = E[k] * dis; });
          dpp_fence(R);
          static_for<1, N>([&](auto jj) {
            constexpr int j = N - jj;
            static_for<0, j>([&](auto k) { mac_bc<j, true>(U[k], R[k], U[j]); });
          });
          double z[N], ET[N];
          static_for<0, N>([&](auto j) { z[j] = 0.0; ET[j] = 0.0; });
          dpp_fence(U);
          static_for<0, N>([&](auto i) {
            static_for<0, N>([&](auto j) { mac_bc<j>(z[j], U[i], xh[i]); });        // z = U' xhat
          });
          dpp_fence(z);
          if (NH == 3 && t + 2 <= T) lds_barrier();                    // barrier t+2 (the job spans NH intervals)
          static_for<0, VJP_PROD_MAX_S>([&](auto s) {
            const double ev = (col && s < S) ? epr[s] : 0.0;                         // (S <= 4: no branches)
            static_for<0, N>([&](auto j) { mac_bc<s>(ET[j], z[j], ev); });         // ET[j][c] = E[c][j]
          });
          if (NH == 2 && t + 2 <= T) lds_barrier();                    // barrier t+2
          double LhT[N], K[N], KT_[N], Pex[N];
          static_for<0, N>([&](auto j) { LhT[j] = ET[j] * mU[j]; K[j] = 0.0; Pex[j] = 0.0; });
          mm_ab<N, N, false>(K, U, LhT);              // K = U Lh'
          transpose_tile<N>(tab, c, K, KT_);          // KT = Lh U'
          if (NH == 3 && t + 3 <= T) lds_barrier();                    // barrier t+3
          mm_ab<N, N, true>(Pex, U, KT_);             // Pex = -U Lh U'
          double* ad = a.adj + (b * T + t) * AS;
          if (valid && col) static_for<0, N>([&](auto i) { ad[vjp_pex_off(N) + i * PS + c] = Pex[i]; });
        }
      }
      return;
    }
    if (wv >= 1) {
      // ---- producer wavefronts: wavefront 1 + r gathers the records of the workgroup's sequence r ----------------
      const int r = wv - 1, T = a.T, SN = a.S * N;
      constexpr bool role1 = ROLE == 1;
      const int br = grp * 4 + r;
      const long bb = br < a.B ? br : a.B - 1;
      const double* wsq = a.ws + bb * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
      const double* base[KT];
      int stp[KT], off[KT];
      static_for<0, KT>([&](auto k) { base[k] = wsq; stp[k] = 0; off[k] = 0; });
      int start = 0;
      auto seg = [&](const double* p, int len, int stride, int o) {   // next `len` elements of the record <- p[t * stride ..]
        static_for<0, KT>([&](auto k) {
          const int f = k * 64 + lane - start;
          if (f >= 0 && f < len) { base[k] = p + f; stp[k] = stride; off[k] = o; }
        });
        start += len;
      };
      if (!role1) {
        seg(wsq, WS, WS, 0);
        seg(a.ws3 + bb * T * W3, W3, W3, 0);
        seg(a.g_x ? a.g_x + bb * T * N : wsq, N, a.g_x ? N : 0, 0);
        seg(a.g_diagxx ? a.g_diagxx + bb * T * N : wsq, N, a.g_diagxx ? N : 0, 0);
      } else {
        seg(wsq, N * HS, WS, 0);
        seg(a.ws2 + bb * T * (N * N + N), N * N + N, N * N + N, 0);
        seg(a.g_samples + bb * T * SN, SN, SN, 0);
        seg(a.samples + bb * T * SN, SN, SN, 1);                    // x_{t+1}
        seg(a.eps + bb * T * SN, SN, SN, 0);
      }
      GStage<KT> s0, s1, s2, s3, s4, s5;                  // stage of step t: t % 6
      static_assert(PD == 6, "six named stages");
      double* slot0 = ring + r * REC;
      gather_issue<KT>(s0, base, stp, off, 0, T);
      gather_issue<KT>(s1, base, stp, off, 1, T);
      gather_issue<KT>(s2, base, stp, off, 2, T);
      gather_issue<KT>(s3, base, stp, off, 3, T);
      gather_issue<KT>(s4, base, stp, off, 4, T);
      gather_issue<KT>(s5, base, stp, off, 5, T);
      gather_publish<KT>(s0, slot0, lane);
      gather_issue<KT>(s0, base, stp, off, 6, T);
      lds_barrier();                                     // barrier 0: step 0 is in slot 0
      // consumer iteration t (between barriers t and t+1) reads slot t%3, the helpers slot (t-1)%3: publish step t+1
      // into the third
      // (the steady-state loop has no branch inside: across one, hipcc's wait-count bookkeeping degrades to
      // vmcnt(0) before every publish, i.e. a prefetch distance of one step)
#define SVAE_PROD_STEP(sg, t)                                                 \
      {                                                                         \
        gather_publish<KT>(sg, slot0 + (((t) + 1) % 3) * SLOT, lane);           \
        gather_issue<KT>(sg, base, stp, off, (t) + 1 + PD, T);                  \
        lds_barrier();                                                          \
      }
      int t0 = 0;
      for (; t0 + 6 < T; t0 += PD) {
        SVAE_PROD_STEP(s1, t0)
        SVAE_PROD_STEP(s2, t0 + 1)
        SVAE_PROD_STEP(s3, t0 + 2)
        SVAE_PROD_STEP(s4, t0 + 3)
        SVAE_PROD_STEP(s5, t0 + 4)
        SVAE_PROD_STEP(s0, t0 + 5)
      }
      if (t0 + 1 < T) SVAE_PROD_STEP(s1, t0)
      if (t0 + 2 < T) SVAE_PROD_STEP(s2, t0 + 1)
      if (t0 + 3 < T) SVAE_PROD_STEP(s3, t0 + 2)
      if (t0 + 4 < T) SVAE_PROD_STEP(s4, t0 + 3)
      if (t0 + 5 < T) SVAE_PROD_STEP(s5, t0 + 4)
#undef SVAE_PROD_STEP
      lds_barrier();                                     // barrier T
      return;
    }
  }
  const int c = lane & 15;
  double* tab = tabs + (lane >> 4) * 256;
  double* mailrow = mail + (PROD ? (lane >> 4) * N * 16 : 0);
  // SPLIT (small batches): role 0 = smoother adjoint, role 1 = sampler adjoint, in separate workgroups;
  // otherwise one workgroup runs both bodies back to back
  constexpr bool do0 = ROLE != 1;
  constexpr bool do1 = SAMP && ROLE != 0;
  const int brow = grp * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N, colN = c <= N;
  const int T = a.T, S = a.S;
  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EN = (c == N) ? 1.0 : 0.0;
  const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);        // G~ row k = H row k * sg
  double mU[SAMP ? N : 1];                                    // (c > j) + 1/2 (c == j)
  if constexpr (SAMP) static_for<0, N>([&](auto j) { mU[j] = (c > j) ? 1.0 : ((c == j) ? 0.5 : 0.0); });

  const double* wsb = a.ws + (long)b * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
  double Sh[N + 1];
  static_for<0, N + 1>([&](auto i) { Sh[i] = 0.0; });
  double HcPrev[SAMP ? N : 1], xh[SAMP ? N : 1];
  if constexpr (SAMP) static_for<0, N>([&](auto k) { HcPrev[k] = 0.0; xh[k] = 0.0; });
  const bool sv = c < S;                                      // this lane carries a sample
  const int ss = sv ? c : 0;

  const double cm = col ? 1.0 : 0.0;
  const double* ringrow = ring + (PROD ? (lane >> 4) * REC : 0);    // PROD: this DPP row's record inside a ring slot
  const int SN = S * N;
  // Memory schedule: the operands a step STARTS with (role 0: W~' and the direct cotangents; role 1: the
  // sample cotangents) are fetched into registers one step ahead; the others (H, P^-1, the LDL' factor) are
  // requested raw at the top of the step and first touched a product later.  Every load unconditional.
  const int cN = colN ? c : 0, ccl = col ? c : 0;
  const bool has_gx = a.g_x != nullptr, has_gd = a.g_diagxx != nullptr;
  double WTn[N + 1], gxn = 0.0, gdn = 0.0, gsn[SAMP ? N : 1];
  auto fetch_next = [&](int t) {
    const double* rec = ringrow + (t % 3) * SLOT;               // (PROD)
    if (do0) {
      if constexpr (!PROD) load_row<N + 1>(a.ws3 + ((long)b * T + t) * (N + 1) * HS + cN * HS, WTn);
      if constexpr (PROD) {
        const double x = rec[WS + W3 + ccl], d = rec[WS + W3 + N + ccl];
        gxn = has_gx ? x : 0.0;
        gdn = has_gd ? d : 0.0;
      } else {
        const long o = ((long)b * T + t) * N + ccl;
        gxn = has_gx ? a.g_x[o] : 0.0;
        gdn = has_gd ? a.g_diagxx[o] : 0.0;
      }
    }
    if constexpr (SAMP) {
      if (do1) {
        const double* gs = PROD ? rec + R1 + ss * N : a.g_samples + (((long)b * T + t) * S + ss) * N;
        static_for<0, N>([&](auto k) { gsn[k] = gs[k]; });
      }
    }
  };
  if constexpr (!PROD) fetch_next(0);
  for (int t = 0; t < T; ++t) {
    if constexpr (PROD) {
      lds_barrier();                                            // barrier t: step t is in slot t%3
      fetch_next(t);
    }
    const double* w = PROD ? ringrow + (t % 3) * SLOT : wsb + (long)t * WS;
    double* ad = a.adj + ((long)b * T + t) * AS;
    double Hcr[N];
    static_for<0, N>([&](auto k) { Hcr[k] = w[k * HS + cN]; });
    double Gb[N];                              // G^ rows i < N (lanes 0..N); one share per role if SPLIT
    static_for<0, N>([&](auto i) { Gb[i] = 0.0; });
    double WT[N + 1], Gc[N + 1], Pir[N], gx = 0.0, gd = 0.0, gsv[SAMP ? N : 1];
    if (do0) {
      if constexpr (!PROD) static_for<0, N + 1>([&](auto k) { WT[k] = colN ? 2.0 * WTn[k] : 0.0; });      // 2 W~'
      gx = col ? 0.5 * gxn : 0.0;              // direct cotangents, symmetrised
      gd = col ? gdn : 0.0;
      if constexpr (!PROD) static_for<0, N>([&](auto i) { Pir[i] = w[N * HS + i * PS + ccl]; });
    }
    if constexpr (SAMP) static_for<0, N>([&](auto k) { gsv[k] = gsn[k]; });
    if constexpr (!PROD) fetch_next(t + 1 < T ? t + 1 : t);
    double (&Hc)[N] = Hcr;      // raw: lanes > N are zeroed by sg / never broadcast
    if (do0) {
    static_for<0, N>([&](auto k) { Gc[k] = Hc[k] * sg; });
    Gc[N] = EN;
    Sh[N] += gx;
    dpp_fence(gx);
    static_for<0, N>([&](auto i) {
      mac_bc<i>(Sh[i], gx, EN);
      Sh[i] = __builtin_fma(gd, E[i], Sh[i]);
    });

    double Cb[STATC ? N : 1], CbT[STATC ? N : 1];
    bool cross = false;
    if constexpr (STATC) {
      // n x n cotangent of E[x_t x_t'] (first statistic of pair t, third of pair t-1, E_init at t = 0)
      const long nn = (long)N * N;
      const int cc = col ? c : 0;
      static_for<0, N>([&](auto i) {
        double q = 0.0;
        if (a.g_E_pair && t < T - 1) {
          const double* gp = a.g_E_pair + ((long)b * (T - 1) + t) * 3 * nn;
          q += gp[i * N + cc] + gp[cc * N + i];
        }
        if (a.g_E_pair && t > 0) {
          const double* gp = a.g_E_pair + ((long)b * (T - 1) + t - 1) * 3 * nn + 2 * nn;
          q += gp[i * N + cc] + gp[cc * N + i];
        }
        if (a.g_E_init && t == 0) {
          const double* gi = a.g_E_init + (long)b * (nn + N);
          q += gi[i * N + cc] + gi[cc * N + i];
        }
        Sh[i] = __builtin_fma(0.5, col ? q : 0.0, Sh[i]);
      });
      if (a.g_E_init && t == 0) {
        double qx = col ? 0.5 * a.g_E_init[(long)b * (nn + N) + nn + c] : 0.0;
        Sh[N] += qx;
        dpp_fence(qx);
        static_for<0, N>([&](auto i) { mac_bc<i>(Sh[i], qx, EN); });
      }
      cross = a.g_E_pair && t < T - 1;
      static_for<0, N>([&](auto i) { Cb[i] = 0.0; CbT[i] = 0.0; });
      if (cross) {
        const double* gc = a.g_E_pair + ((long)b * (T - 1) + t) * 3 * nn + nn;
        static_for<0, N>([&](auto i) {
          const double v = gc[i * N + cc], vt = gc[cc * N + i];
          Cb[i] = col ? v : 0.0;                      // row i of Cb
          CbT[i] = col ? vt : 0.0;                    // row i of Cb'
        });
      }
    }

    // G^ rows i < N:  2 S^ W~'   (lanes 0..N)      (PROD: the second helper wavefront's job)
    if constexpr (!PROD) mm_ab<N, N + 1, false>(Gb, Sh, WT);
    if constexpr (STATC) {
      if (cross) {
        // G^[i] += sum_k Cb[i][k] S~_{t+1}[k]:  S~_{t+1} rows from the forward outputs
        const long nn = (long)N * N;
        const double* e3 = a.E_pair + ((long)b * (T - 1) + t) * 3 * nn + 2 * nn;
        const double* ex = a.E_node_x + ((long)b * T + t + 1) * N;
        double Snx[N];
        static_for<0, N>([&](auto k) {
          const double v = e3[k * N + (col ? c : 0)], m = ex[k];
          Snx[k] = col ? v : (c == N ? m : 0.0);
        });
        dpp_fence(Cb);
        mm_ab<N, N, false>(Gb, Cb, Snx);
      }
    }
    if constexpr (PROD) {
      // rows 0..N-1 of S^ (before the propagation below) for the helper wavefronts: Pinvbar = S^[:n,:n], G^ = 2 S^ W~'
      double* mb = mailrow + (t & 1) * MSLOT;
      static_for<0, N>([&](auto i) { mb[i * 16 + c] = Sh[i]; });
    } else {
      // the part of Pbar_t that does not depend on the filter-adjoint recursion of sweep 2:
      //   -P^-1 Pinvbar P^-1,   Pinvbar = S^[:n,:n] (before the propagation below)
      double Pi[N], Pib[N], T1[N], Pbp[N];
      static_for<0, N>([&](auto i) {
        Pi[i] = Pir[i];
        Pib[i] = Sh[i] * cm;
        T1[i] = 0.0; Pbp[i] = 0.0;
      });
      mm_ab<N, N, false>(T1, Pib, Pi);
      mm_ab<N, N, true>(Pbp, Pi, T1);
      if (valid) static_for<0, N>([&](auto i) { if (c <= i) ad[vjp_pbp_off(N) + i * (i + 1) / 2 + c] = Pbp[i]; });
    }

    // S^ <- G~' (S^ G~)
    {
      double M[N + 1], Sn[N + 1];
      static_for<0, N + 1>([&](auto i) { M[i] = 0.0; Sn[i] = 0.0; });
      mm_ab<N + 1, N + 1, false>(M, Sh, Gc);
      mm_atb<N + 1, N + 1, false>(Sn, Gc, M);
      static_for<0, N + 1>([&](auto i) { Sh[i] = Sn[i]; });
    }
    if constexpr (STATC) {
      if (cross) {                                  // S^_{t+1} += sym([Cb' 0; 0 0] G~)
        double Mx[N], MxT[N + 1];
        static_for<0, N>([&](auto i) { Mx[i] = 0.0; });
        static_for<0, N + 1>([&](auto i) { MxT[i] = 0.0; });
        dpp_fence(CbT);
        mm_ab<N, N, false>(Mx, CbT, Gc);            // rows k < N of  Cb' G~[:n]
        mm_atb<N + 1, N, false>(MxT, Gc, Cb);       // its transpose
        static_for<0, N>([&](auto i) { Sh[i] = __builtin_fma(0.5, Mx[i], Sh[i]); });
        static_for<0, N + 1>([&](auto i) { Sh[i] = __builtin_fma(0.5, MxT[i], Sh[i]); });
      }
    }

    if (!do1 && !PROD) { if (valid && colN) static_for<0, N>([&](auto i) { ad[i * HS + c] = Gb[i]; }); }
    }   // smoother adjoint

    if constexpr (SAMP) {
      if (do1) {
      // xhat_t = g_samples_t - X_{t-1}' xhat_{t-1}   (register k, lane = sample)
      double xn[N];
      static_for<0, N>([&](auto k) { xn[k] = sv ? gsv[k] : 0.0; });
      // the LDL' factor of the step (needed two product stages further down)
      const double* w2 = PROD ? w + N * HS : a.ws2 + ((long)b * T + t) * (N * N + N);
      const double* x1rec = PROD ? w + R1 + SN : a.samples + ((long)b * T + (t + 1 < T ? t + 1 : t)) * SN;   // x_{t+1}, per sample
      const double* eprec = PROD ? w + R1 + 2 * SN : a.eps + ((long)b * T + t) * SN;
      double Rr[N], pvv = 1.0;
      if constexpr (!PROD) {
        static_for<0, N>([&](auto k) { Rr[k] = w2[k * N + ccl]; });
        pvv = w2[N * N + ccl];
      }
      double x1r[SPRE], epr[SPRE];
      static_for<0, SPRE>([&](auto s) {
        const int sq = s < S ? (int)s : 0;                      // (clamped: unconditional loads)
        x1r[s] = x1rec[sq * N + ccl];
        if constexpr (!PROD) epr[s] = eprec[sq * N + ccl];
      });
      dpp_fence(HcPrev);
      static_for<0, N>([&](auto j) {
        static_for<0, N>([&](auto k) { mac_bc<k, true>(xn[k], HcPrev[j], xh[j]); });
      });
      static_for<0, N>([&](auto k) { xh[k] = xn[k]; HcPrev[k] = Hc[k]; });
      if constexpr (PROD) {                        // xhat_t for the helper wavefronts (noise adjoint)
        double* mb = mailrow + (t & 1) * MSLOT;
        static_for<0, N>([&](auto k) { mb[k * 16 + c] = xh[k]; });
      }
      dpp_fence(xh);
      // cbar_t += sum_s xhat (lane N);  Xbar_t -= sum_s xhat x_{t+1}'  (i.e. G^[:, :n] += ...)
      // (x_{t+1} and eps of the first samples were requested at the top of the role: x1r, epr)
      if constexpr (SPLIT) {
        // two-role launch (S <= VJP_SPLIT_MAX_S): this role's share of G^ leaves as its factors, per sample
        // [xhat_s | x_{t+1,s}] -- sweep 2 forms the rank-S product
        double* vo = ad + vjp_vec_off(N);
        if (valid && sv) static_for<0, N>([&](auto k) { vo[c * 2 * N + k] = xh[k]; });     // lane s holds xhat_s
        for_samples<0>(S, [&](auto s) {
          const double x1 = (s < SPRE) ? x1r[s < SPRE ? (int)s : 0] : x1rec[s * N + ccl];
          if (valid && col) vo[s * 2 * N + N + c] = (t + 1 < T) ? x1 : 0.0;
        });
      } else {
      for_samples<0>(S, [&](auto s) {
        double v = EN;
        if (t + 1 < T) {
          const double x1 = (s < SPRE) ? x1r[s < SPRE ? (int)s : 0] : x1rec[s * N + ccl];
          v += col ? x1 : 0.0;
        }
        asm volatile("s_nop 1");   // block entry after the branch: two wait states before the DPP reads (audit rule)
        static_for<0, N>([&](auto i) { mac_bc<s>(Gb[i], xh[i], v); });
      });
      }
      if constexpr (!PROD) {
      // noise adjoint:  Pbar_t(direct) = -U (Lh U')  with  U = L^-T D^-1/2,  Lh from E' = sum_s eps_s z_s'
      double R[N], U[N];
      static_for<0, N>([&](auto k) { R[k] = Rr[k]; });
      const double dis = col ? rsqrt_nr(pvv) : 0.0;
      static_for<0, N>([&](auto k) { U[k] = E[k] * dis; });
      dpp_fence(R);
      static_for<1, N>([&](auto jj) {
        constexpr int j = N - jj;
        static_for<0, j>([&](auto k) { mac_bc<j, true>(U[k], R[k], U[j]); });
      });
      double z[N], ET[N];
      static_for<0, N>([&](auto j) { z[j] = 0.0; ET[j] = 0.0; });
      dpp_fence(U);
      static_for<0, N>([&](auto i) {
        static_for<0, N>([&](auto j) { mac_bc<j>(z[j], U[i], xh[i]); });        // z = U' xhat
      });
      dpp_fence(z);
      for_samples<0>(S, [&](auto s) {
        const double e1 = (s < SPRE) ? epr[s < SPRE ? (int)s : 0] : eprec[s * N + ccl];
        const double ev = col ? e1 : 0.0;
        asm volatile("s_nop 1");   // block entry (audit rule)
        static_for<0, N>([&](auto j) { mac_bc<s>(ET[j], z[j], ev); });         // ET[j][c] = E[c][j]
      });
      double LhT[N], K[N], KT[N], Pex[N];
      static_for<0, N>([&](auto j) { LhT[j] = ET[j] * mU[j]; K[j] = 0.0; Pex[j] = 0.0; });
      mm_ab<N, N, false>(K, U, LhT);              // K = U Lh'
      transpose_tile<N>(tab, c, K, KT);           // KT = Lh U'
      mm_ab<N, N, true>(Pex, U, KT);              // Pex = -U Lh U'
      if (valid && col) static_for<0, N>([&](auto i) { ad[vjp_pex_off(N) + i * PS + c] = Pex[i]; });
      }
      // G^ total when this workgroup ran both bodies
      if constexpr (!SPLIT) { if (valid && colN) static_for<0, N>([&](auto i) { ad[i * HS + c] = Gb[i]; }); }
      }   // sampler adjoint
    }
  }
  if constexpr (PROD) lds_barrier();               // barrier T
}

template <int N> constexpr int vjp_s1_ring_doubles() { return 3 * 4 * vjp_s1_pieces<N>() * 64; }
template <int N, bool SAMP, bool STATC, bool SPLIT>
__global__ __launch_bounds__(64) void lds_vjp_sweep1_kernel(const VjpArgs a) {
  __shared__ double tabs[4 * 256];
  if constexpr (SAMP && SPLIT) {
    if ((blockIdx.x & 1) == 0) lds_vjp_sweep1_body<N, SAMP, STATC, SPLIT, false, 0>(a, tabs, nullptr, nullptr, blockIdx.x >> 1);
    else lds_vjp_sweep1_body<N, SAMP, STATC, SPLIT, false, 1>(a, tabs, nullptr, nullptr, blockIdx.x >> 1);
  } else {
    lds_vjp_sweep1_body<N, SAMP, STATC, SPLIT, false, 2>(a, tabs, nullptr, nullptr, blockIdx.x);
  }
}
// consumer wavefront + four producer wavefronts + helper wavefronts
template <int N, bool SAMP>
__global__ __launch_bounds__(64 * VJP_S1_WAVES) void lds_vjp_sweep1_prod_kernel(const VjpArgs a) {
  __shared__ double tabs[(SVAE_S1_NH1 == 3 ? 3 : 2) * 4 * 256];
  __shared__ double ring[vjp_s1_ring_doubles<N>()];
  __shared__ double mail[2 * 4 * N * 16];
  if constexpr (SAMP) {
#ifdef SVAE_S1_SKIP_ROLE
    if ((blockIdx.x & 1) == SVAE_S1_SKIP_ROLE) return;     // (timing experiment: one role alone)
#endif
    if ((blockIdx.x & 1) == 0) lds_vjp_sweep1_body<N, true, false, true, true, 0>(a, tabs, ring, mail, blockIdx.x >> 1);
    else lds_vjp_sweep1_body<N, true, false, true, true, 1>(a, tabs, ring, mail, blockIdx.x >> 1);
  } else {
    lds_vjp_sweep1_body<N, false, false, false, true, 2>(a, tabs, ring, mail, blockIdx.x);
  }
}

// ---- sweep 2: filter adjoint, backward in time -----------------------------------------------------
// PROD (small batches): four more wavefronts of the workgroup are PRODUCERS, one per sequence.  The sweep is serial in t and every
// step's operands -- the E-step record and the sweep-1 record of step t, 5 KB per sequence, written long ago -- come
// from HBM, ~2 us away, against ~0.9 us of arithmetic; the consumer's registers cannot hold enough steps in flight.
// The producer can: it keeps PD steps of records in its own registers (it does nothing else), publishes the oldest into
// an LDS ring of two slots each step, and the consumer reads its operands from LDS.  One s_barrier per step.
template <int N, bool SAMP, bool SPLIT, bool PROD>
__device__ __forceinline__ void lds_vjp_sweep2_body(const VjpArgs& a) {
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N);
  constexpr int AS = vjp_step_doubles(N);
  constexpr int REC = WS + AS;                 // doubles per (sequence, step) in a ring slot
  constexpr int SLOT = 4 * REC;
  constexpr int PD = 4;                        // steps each producer keeps in flight (24 VGPRs per step: no spills -- scratch reloads would be in-order VMEM too)
  constexpr int KW = (WS / 2 + 63) / 64, KA = (AS / 2 + 63) / 64;   // 16-byte loads per lane, sequence and record
  static_assert(WS % 2 == 0 && AS % 2 == 0, "records are copied as 16-byte pairs");
  __shared__ double ring[PROD ? 2 * SLOT : 2];
  const int lane = threadIdx.x & 63;
  const int T = a.T;
  if constexpr (PROD) {
    const int wv = threadIdx.x >> 6;
    if (wv >= 1) {
      // ---- producer wavefronts: wavefront 1 + r streams the records of the workgroup's sequence r ---------------
      const int r = wv - 1;
      const int br = blockIdx.x * 4 + r;
      const int bb = br < a.B ? br : a.B - 1;
      const vd2* wrec = reinterpret_cast<const vd2*>(a.ws + (long)bb * ws_seq_doubles(N, T) + ws_zpage_doubles(N));
      const vd2* arec = reinterpret_cast<const vd2*>(a.adj + ((long)bb * T) * AS);
      Stage<KW, KA> s0, s1, s2, s3;                      // stage of step t: (T - 1 - t) % 4
      static_assert(PD == 4, "four named stages");
      vd2* slot0 = reinterpret_cast<vd2*>(ring) + r * (REC / 2);
      auto rec = [&](int t) { return t > 0 ? t : 0; };
      prod_issue<WS / 2, AS / 2>(s0, wrec, arec, rec(T - 1), lane);
      prod_issue<WS / 2, AS / 2>(s1, wrec, arec, rec(T - 2), lane);
      prod_issue<WS / 2, AS / 2>(s2, wrec, arec, rec(T - 3), lane);
      prod_issue<WS / 2, AS / 2>(s3, wrec, arec, rec(T - 4), lane);
      prod_publish<WS / 2, AS / 2>(s0, slot0 + ((T - 1) & 1) * (SLOT / 2), lane);
      prod_issue<WS / 2, AS / 2>(s0, wrec, arec, rec(T - 5), lane);
      lds_barrier();                                     // barrier 0: step T-1 is in its slot
      // consumer iteration t (between two barriers) reads slot t%2: publish step t-1 into the other slot, refill the
      // stage with step t-1-PD
      // (no branch inside the steady-state loop: see sweep 1)
#define SVAE_PROD_STEP(sg, t)                                                                         \
      {                                                                                                 \
        prod_publish<WS / 2, AS / 2>(sg, slot0 + (((t) - 1) & 1) * (SLOT / 2), lane);                   \
        prod_issue<WS / 2, AS / 2>(sg, wrec, arec, rec((t) - 1 - PD), lane);                            \
        lds_barrier();                                                                                  \
      }
      int t0 = T - 1;
      for (; t0 >= 4; t0 -= PD) {
        SVAE_PROD_STEP(s1, t0)
        SVAE_PROD_STEP(s2, t0 - 1)
        SVAE_PROD_STEP(s3, t0 - 2)
        SVAE_PROD_STEP(s0, t0 - 3)
      }
      if (t0 >= 1) SVAE_PROD_STEP(s1, t0)
      if (t0 >= 2) SVAE_PROD_STEP(s2, t0 - 1)
      if (t0 >= 3) SVAE_PROD_STEP(s3, t0 - 2)
#undef SVAE_PROD_STEP
      return;
    }
  }
  const int c = lane & 15;
  const int brow = blockIdx.x * 4 + (lane >> 4);
  const bool valid = brow < a.B;
  const int b = valid ? brow : a.B - 1;
  const bool col = c < N, colN = c <= N;
  double E[N];
  static_for<0, N>([&](auto i) { E[i] = (c == i) ? 1.0 : 0.0; });
  const double EN = (c == N) ? 1.0 : 0.0;
  const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);
  double J12c[N];                                             // info form: J12 = -natJ12
  const double* pJ12 = a.J12 + (long)b * a.pair_seq_stride;
  static_for<0, N>([&](auto i) { J12c[i] = 0.0; });
  if (T > 1 && a.pair_t_stride == 0)
    static_for<0, N>([&](auto i) { const double v = pJ12[i * N + (col ? c : 0)]; J12c[i] = col ? -v : 0.0; });
  const double g = a.g_lognorm[b];

  const double* wsb = a.ws + (long)b * ws_seq_doubles(N, T) + ws_zpage_doubles(N);
  const double* ringrow = ring + (PROD ? (lane >> 4) * REC : 0);   // PROD: this DPP row's record inside a ring slot
  double Ab[N];                                               // [Abar | hbar] of step t+1
  static_for<0, N>([&](auto i) { Ab[i] = 0.0; });

  // Memory schedule of a step (the sweep is serial in t, so nothing else hides a load):
  //   * G^ of step t-1 (the operand the step STARTS with) is fetched into registers one step ahead;
  //   * P^-1, H, H' of the step are requested at its top, raw, and first touched after the J12 product;
  //   * the sweep-1 shares of Pbar are requested before the P^-1 product that precedes their use;
  //   * (small batches) the records of a step VJP_AHEAD iterations ahead are touched so that all of these
  //     find their lines in L2 (they were written ~T steps ago by other kernels: HBM otherwise).
  const int cN = colN ? c : 0, cc = col ? c : 0;
  constexpr int VS = (SAMP && SPLIT) ? VJP_SPLIT_MAX_S : 1;    // two-role launches: the sampler share of G^ arrives as factors
  const int S = a.S;
  double gbn[N], xsn[VS], vsn[VS];
  auto fetch_g = [&](int t) {
    const double* ad = PROD ? ringrow + (t & 1) * SLOT + WS : a.adj + ((long)b * T + t) * AS;
    static_for<0, N>([&](auto i) { gbn[i] = ad[i * HS + cN]; });
    if constexpr (SAMP && SPLIT) {
      static_for<0, VS>([&](auto s) {
        const int sq = s < S ? (int)s : 0;                       // (clamped: unconditional loads)
        xsn[s] = ad[vjp_vec_off(N) + sq * 2 * N + cc];           // xhat_s
        vsn[s] = ad[vjp_vec_off(N) + sq * 2 * N + N + cc];       // x_{t+1,s}
      });
    }
  };
  int symo[N];                                                 // entry (i, c) of a symmetric matrix kept as its lower triangle
  static_for<0, N>([&](auto i) { symo[i] = cc <= i ? i * (i + 1) / 2 + cc : cc * (cc + 1) / 2 + i; });
  if constexpr (!PROD) fetch_g(T - 1);
  double warm = 0.0, sink = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    if constexpr (PROD) {
      lds_barrier();                                          // step t is in slot t%2
      fetch_g(t);                                             // (from LDS: no step-ahead register copy needed)
    }
    const double* w = PROD ? ringrow + (t & 1) * SLOT : wsb + (long)t * WS;
    const double* ad = PROD ? w + WS : a.adj + ((long)b * T + t) * AS;
    double Xc[N];
    static_for<0, N>([&](auto i) { Xc[i] = colN ? gbn[i] * sg : 0.0; });      // [Xbar | cbar] = [-G^ | G^[:,n]]
    if constexpr (SAMP && SPLIT) {
      // sampler share: G^ += sum_s xhat_s [x_{t+1,s}' | 1]  (sign of [Xbar | cbar] folded into the second factor)
      double xs[VS], vq[VS];
      static_for<0, VS>([&](auto s) { xs[s] = xsn[s]; vq[s] = col ? -vsn[s] : EN; });
      dpp_fence(xs);
      static_for<0, N>([&](auto i) { mac_bc<i>(Xc[i], xs[0], vq[0]); });      // (sample 0 without a branch)
      if (S > 1) {
        for_samples<1>(S, [&](auto s) {
          if constexpr (s < VS) {
            asm volatile("s_nop 1");   // block entry after the branch: two wait states before the DPP reads (audit rule)
            static_for<0, N>([&](auto i) { mac_bc<i>(Xc[i], xs[s], vq[s]); });
          }
        });
      }
    }
    // raw operands: lanes outside the tile read element 0 of the row (finite); they are never a DPP
    // broadcast source and only reach lanes of the results that are masked or not stored
    double Pi[N], Hc[N], HT[N + 1];
    static_for<0, N>([&](auto i) { Pi[i] = w[N * HS + i * PS + cc]; Hc[i] = w[i * HS + cN]; });
    load_row<N + 1>(w + cc * HS, HT);                          // H' (lane c: row c of H)
    if constexpr (!PROD) fetch_g(t > 0 ? t - 1 : 0);
    sink += warm;                                              // last iteration's touches have landed
    if (!PROD && a.B <= 2048) {                                // (large batches are bandwidth-bound: no extra traffic)
      const int tw = t >= VJP_AHEAD ? t - VJP_AHEAD : 0;
      warm = touch_lines<(WS + 15) / 16>(wsb + (long)tw * WS, c, WS)
           + touch_lines<(AS + 15) / 16>(a.adj + ((long)b * T + tw) * AS, c, AS);
    }
    // [Xbar | cbar] -= J12_t [Abar | hbar]_{t+1}
    if (t < T - 1) {
      if (a.pair_t_stride != 0) {
        const double* pj = pJ12 + (long)t * a.pair_t_stride;
        static_for<0, N>([&](auto i) { const double v = pj[i * N + (col ? c : 0)]; J12c[i] = col ? -v : 0.0; });
        dpp_fence(J12c);
      }
      mm_ab<N, N, true>(Xc, J12c, Ab);
    }
    // Bbar = P^-1 [Xbar | cbar]
    double Bb[N], Pb[N], Pxr[SAMP ? N : 1];
    static_for<0, N>([&](auto i) {                             // sweep-1 shares of Pbar: land during the product
      Pb[i] = ad[vjp_pbp_off(N) + symo[i]];
      if constexpr (SAMP) Pxr[i] = ad[vjp_pex_off(N) + i * PS + cc];
      Bb[i] = 0.0;
    });
    mm_ab<N, N, false>(Bb, Pi, Xc);
    // Pbar = [-P^-1 Pinvbar P^-1: from sweep 1] - Bbar H' - 1/2 g (c c' + P^-1) [+ direct]
    mm_ab<N, N + 1, true>(Pb, Bb, HT);
    double cvs = -0.5 * g * HT[N];
    dpp_fence(cvs);
    static_for<0, N>([&](auto i) {
      mac_bc<i>(Pb[i], cvs, HT[N]);
      Pb[i] = __builtin_fma(-0.5 * g, Pi[i], Pb[i]);
    });
    if constexpr (SAMP) {
      static_for<0, N>([&](auto i) { Pb[i] += Pxr[i]; });
    }
    // outputs and the adjoint handed to step t-1:  Ab = [Pbar | Bbar[:,n] + g c]
    double gJ = 0.0, gh = 0.0;
    static_for<0, N>([&](auto i) {
      const double hfb = __builtin_fma(g, Hc[i], Bb[i]);      // lane N: hfbar_i
      Ab[i] = __builtin_fma(EN, hfb - Pb[i], Pb[i]);
      gJ = __builtin_fma(E[i], Pb[i], gJ);
    });
    dpp_fence(Ab);
    static_for<0, N>([&](auto i) { mac_bc<N>(gh, Ab[i], E[i]); });
    if (sink == 1.2345e300) gJ += sink;                        // keeps the touches alive; never true
    if (valid && col) {
      a.g_node_J[((long)b * T + t) * N + c] = -2.0 * gJ;
      a.g_node_h[((long)b * T + t) * N + c] = gh;
    }
    if (a.g_P) {
      // dense node potentials (the reference's Python path, lds_inference.py:65-82): J_node,t enters P_t alone, so its
      // cotangent is the whole of -2 Pbar_t, not just the diagonal (the caller symmetrises)
      double* gp = a.g_P + (((long)b * T + t) * N) * N + c;
      if (valid && col) static_for<0, N>([&](auto i) { gp[i * N] = -2.0 * Pb[i]; });
    }
  }
}

// Sweep 1 WITHOUT sample cotangents for B <= VJP_S4_MAX_B: the smoother adjoint with ONE SEQUENCE PER WORKGROUP --
// consumer, producer and helper wavefront, all in the four-row split layout (s1_role0_wg) -- so that its heavy
// wavefronts spread over the chip's SIMDs instead of sharing the four of one CU (0.23 -> 0.15 ms at 512 x 200 x 10).
// With samples the packed two-role kernel stays: next to role 1's workgroups the split layout's extra wavefronts
// (1024 + 384 heavy ones on 1024 SIMDs) cost more than the shorter instruction stream gains (measured: 0.46 vs 0.43 ms
// for the whole VJP).
template <int N>
__global__ __launch_bounds__(192) void lds_vjp_sweep1_s4_kernel(const VjpArgs a) {
  __shared__ double ring[3 * vjp_s1_pieces<N>() * 64];
  __shared__ double mail[2 * N * 16];
  s1_role0_wg<N>(a, ring, mail, blockIdx.x);
}

// ---- sweep 2, ONE SEQUENCE PER WAVEFRONT (B <= VJP_S4_MAX_B: the packed sweep leaves 7/8 of the SIMDs idle) ------------
// The same recursion as lds_vjp_sweep2_body with the four DPP rows of the consumer wavefront SPLITTING every product
// stage by output row (row i of a tile in DPP row i & 3, slot i >> 2, as in lds_estep_twoend_s4.hpp): 96 instead of
// 320 DPP multiply-adds per step; the two operands a product needs replicated ([Xbar | cbar] for Bbar = P^-1 [..], and
// [Abar | hbar] for the J12 product of the next step) are all-gathered with v_permlane16_swap + v_permlane32_swap.
// Second wavefront of the workgroup: the producer of the sequence's records (VJP_S4_PD steps in flight, two-slot LDS
// ring, one barrier per step; T barriers per wavefront).
constexpr int VJP_S4_MAX_B = 512;
constexpr int VJP_S4_PD = 6;
template <int N, bool SAMP>
__global__ __launch_bounds__(128) void lds_vjp_sweep2_s4_kernel(const VjpArgs a) {
  constexpr int HS = ws_h_stride(N), PS = ws_p_stride(N), WS = ws_step_doubles(N);
  constexpr int AS = vjp_step_doubles(N);
  constexpr int REC = WS + AS, PD = VJP_S4_PD;
  constexpr int KW = (WS / 2 + 63) / 64, KA = (AS / 2 + 63) / 64;
  constexpr int J = (N + 3) / 4;                 // slots holding rows 0..N-1 (row i = 4j + r)
  static_assert(WS % 2 == 0 && AS % 2 == 0, "records are copied as 16-byte pairs");
  constexpr int PXO = REC;                        // slot: [E-step record | sweep-1 record | PX (J slots x 64 lanes)]
  constexpr int REC2 = SVAE_S2_PROD_PX ? REC + J * 64 : REC;
  __shared__ double ring[2 * REC2];
  constexpr int RSG = 4 * J;                      // row stride of the consumer's gather tile
  __shared__ double gtile[16 * RSG];             // [lane c][row 4j + r]: the adjoint handed to the next step, on its way to every DPP row
  const int lane = threadIdx.x & 63;
  const int T = a.T;
  const long b = blockIdx.x;
  if ((threadIdx.x >> 6) == 1) {
    // ---- producer wavefront ---------------------------------------------------------------------------------------
    const vd2* wrec = reinterpret_cast<const vd2*>(a.ws + b * ws_seq_doubles(N, T) + ws_zpage_doubles(N));
    const vd2* arec = reinterpret_cast<const vd2*>(a.adj + (b * T) * AS);
    Stage<KW, KA> s0, s1, s2, s3, s4, s5;          // stage of step t: (T - 1 - t) % 6
    static_assert(PD == 6, "six named stages");
    vd2* slot0 = reinterpret_cast<vd2*>(ring);
    auto rec = [&](int t) { return t > 0 ? t : 0; };
#if SVAE_S2_PROD_PX
    // PX_t = P^-1 [-G^ | G^[:,n]]_t (G^ with the sampler share), the part of Bbar_t that does not depend on the recursion:
    // computed here, from the record just published, in the consumer's slot layout (row i = 4j + r in DPP row r) -- the
    // consumer's serial chain per step is then  Bbar = PX - H Abar_{t+1}  ->  Pbar  ->  Abar_t  with ONE all-gather
    // (it was  Xbar = Xhat - J12 Abar -> all-gather -> Bbar = P^-1 Xbar -> Pbar -> Abar_t -> all-gather)
    const int pc = lane & 15, pr = lane >> 4;
    const bool pcol = pc < N;
    const int pcN = pc <= N ? pc : 0, pcc = pcol ? pc : 0;
    const double pEN = (pc == N) ? 1.0 : 0.0;
    const double psg = pcol ? -1.0 : (pc == N ? 1.0 : 0.0);
    int pri[J];
    double prm[J];
    static_for<0, J>([&](auto j) { const int i = 4 * j + pr; pri[j] = i < N ? i : 0; prm[j] = i < N ? 1.0 : 0.0; });
    const int pS = a.S;
    auto publish_px = [&](double* slot) {
      const double* w = slot;
      const double* ad = w + WS;
      double Pi[J], XR[N], PX[J];
      static_for<0, J>([&](auto j) { Pi[j] = w[N * HS + pri[j] * PS + pcc]; });
      static_for<0, N>([&](auto k) { XR[k] = ad[k * HS + pcN]; });                 // row k of G^ in every DPP row
      if constexpr (SAMP) {
        // sampler share of G^ (two-role launch: factors per sample):  G^ += sum_s xhat_s [x_{t+1,s}' | 1]
        auto add_sample = [&](auto s) {
          const double* vo = ad + vjp_vec_off(N) + s * 2 * N;
          const double x1 = vo[N + pcc];
          const double vv = pcol ? x1 : pEN;
          static_for<0, N>([&](auto k) { XR[k] = __builtin_fma(vo[k], vv, XR[k]); });
        };
        add_sample(std::integral_constant<int, 0>{});
        if (pS > 1) static_for<1, VJP_SPLIT_MAX_S>([&](auto s) { if (s < pS) add_sample(s); });
      }
      static_for<0, J>([&](auto j) { Pi[j] *= prm[j]; PX[j] = 0.0; });
      dpp_fence(Pi);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k>(PX[j], Pi[j], XR[k]); });
      });
      static_for<0, J>([&](auto j) { slot[PXO + j * 64 + lane] = PX[j] * psg; });
    };
#define SVAE_PUBLISH_PX(slotv) publish_px(reinterpret_cast<double*>(slotv));
#else
#define SVAE_PUBLISH_PX(slotv)
#endif
    prod_issue<WS / 2, AS / 2>(s0, wrec, arec, rec(T - 1), lane);
    prod_issue<WS / 2, AS / 2>(s1, wrec, arec, rec(T - 2), lane);
    prod_issue<WS / 2, AS / 2>(s2, wrec, arec, rec(T - 3), lane);
    prod_issue<WS / 2, AS / 2>(s3, wrec, arec, rec(T - 4), lane);
    prod_issue<WS / 2, AS / 2>(s4, wrec, arec, rec(T - 5), lane);
    prod_issue<WS / 2, AS / 2>(s5, wrec, arec, rec(T - 6), lane);
    prod_publish<WS / 2, AS / 2>(s0, slot0 + ((T - 1) & 1) * (REC2 / 2), lane);
    prod_issue<WS / 2, AS / 2>(s0, wrec, arec, rec(T - 7), lane);
    SVAE_PUBLISH_PX(slot0 + ((T - 1) & 1) * (REC2 / 2))
    lds_barrier();                                     // barrier 0: step T-1 is in its slot
#define SVAE_PROD_STEP(sg, t)                                                                         \
    {                                                                                                   \
      prod_publish<WS / 2, AS / 2>(sg, slot0 + (((t) - 1) & 1) * (REC2 / 2), lane);                     \
      prod_issue<WS / 2, AS / 2>(sg, wrec, arec, rec((t) - 1 - PD), lane);                              \
      SVAE_PUBLISH_PX(slot0 + (((t) - 1) & 1) * (REC2 / 2))                                             \
      lds_barrier();                                                                                    \
    }
    int t0 = T - 1;
    for (; t0 >= 6; t0 -= PD) {                        // (no branch inside the steady-state loop: see sweep 1)
      SVAE_PROD_STEP(s1, t0)
      SVAE_PROD_STEP(s2, t0 - 1)
      SVAE_PROD_STEP(s3, t0 - 2)
      SVAE_PROD_STEP(s4, t0 - 3)
      SVAE_PROD_STEP(s5, t0 - 4)
      SVAE_PROD_STEP(s0, t0 - 5)
    }
    if (t0 >= 1) SVAE_PROD_STEP(s1, t0)
    if (t0 >= 2) SVAE_PROD_STEP(s2, t0 - 1)
    if (t0 >= 3) SVAE_PROD_STEP(s3, t0 - 2)
    if (t0 >= 4) SVAE_PROD_STEP(s4, t0 - 3)
    if (t0 >= 5) SVAE_PROD_STEP(s5, t0 - 4)
#undef SVAE_PROD_STEP
#undef SVAE_PUBLISH_PX
    return;
  }
  // ---- consumer wavefront -------------------------------------------------------------------------------------------
  const int c = lane & 15, r = lane >> 4;
  const bool col = c < N, colN = c <= N;
  const int cN = colN ? c : 0, cc = col ? c : 0;
  const double EN = (c == N) ? 1.0 : 0.0;
  [[maybe_unused]] const double sg = col ? -1.0 : (c == N ? 1.0 : 0.0);
  double ED[J];                                     // ED[j][c] = (c == 4j + r): the diagonal in slot layout
  static_for<0, J>([&](auto j) { ED[j] = (c == 4 * j + r && c < N) ? 1.0 : 0.0; });
  int ri[J];                                        // this DPP row's tile rows (clamped) and their validity
  double rm[J];
  static_for<0, J>([&](auto j) { const int i = 4 * j + r; ri[j] = i < N ? i : 0; rm[j] = i < N ? 1.0 : 0.0; });
#if !SVAE_S2_PROD_PX
  double J12c[J];                                   // info form: J12 = -natJ12, my rows
  const double* pJ12 = a.J12 + b * a.pair_seq_stride;
  static_for<0, J>([&](auto j) { J12c[j] = 0.0; });
  if (T > 1 && a.pair_t_stride == 0)
    static_for<0, J>([&](auto j) { const double v = pJ12[ri[j] * N + cc]; J12c[j] = col ? -v * rm[j] : 0.0; });
#endif
  const double g = a.g_lognorm[b];
  double AbR[N];                                    // [Abar | hbar] of step t+1, replicated over the DPP rows
  static_for<0, N>([&](auto i) { AbR[i] = 0.0; });
  const bool olane = col && (c & 3) == r;           // lane i of DPP row i & 3 reports node i
  int so[J];                                        // entry (row, c) of a symmetric matrix kept as its lower triangle
  static_for<0, J>([&](auto j) { so[j] = cc <= ri[j] ? ri[j] * (ri[j] + 1) / 2 + cc : cc * (cc + 1) / 2 + ri[j]; });
  [[maybe_unused]] const int S = a.S;

  for (int t = T - 1; t >= 0; --t) {
    lds_barrier();                                  // step t is in slot t % 2
    const double* w = ring + (t & 1) * REC2;
    const double* ad = w + WS;
#if SVAE_S2_PROD_PX
    double Pi[J], Hc[J], Pb[J], HT[N + 1], Bb[J];
    static_for<0, J>([&](auto j) {
      Hc[j] = w[ri[j] * HS + cN];
      Bb[j] = w[PXO + j * 64 + lane];               // PX_t, from the producer
      Pi[j] = w[N * HS + ri[j] * PS + cc];
      Pb[j] = ad[vjp_pbp_off(N) + so[j]];
      if constexpr (SAMP) Pb[j] += ad[vjp_pex_off(N) + ri[j] * PS + cc];
    });
    load_row<N + 1>(w + cc * HS, HT);               // H' (lane c: row c of H), the same in every DPP row
    // Bbar = P^-1 ([-G^ | G^[:,n]] - J12_t [Abar | hbar]_{t+1}) = PX - H [Abar | hbar]_{t+1}      (my rows; H = P^-1 J12_t
    // is the forward record's, so the per-step pair parameters are not read again)
    if (t < T - 1) {
      double HcM[J];
      static_for<0, J>([&](auto j) { HcM[j] = Hc[j] * rm[j]; });
      dpp_fence(HcM);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k, true>(Bb[j], HcM[j], AbR[k]); });
      });
    }
    static_for<0, J>([&](auto j) { Pi[j] *= rm[j]; });
#else
    double gb[J], Pi[J], Hc[J], Pb[J], HT[N + 1];
    static_for<0, J>([&](auto j) {
      gb[j] = ad[ri[j] * HS + cN];
      Pi[j] = w[N * HS + ri[j] * PS + cc];
      Hc[j] = w[ri[j] * HS + cN];
      Pb[j] = ad[vjp_pbp_off(N) + so[j]];
      if constexpr (SAMP) Pb[j] += ad[vjp_pex_off(N) + ri[j] * PS + cc];
    });
    if constexpr (SAMP) {
      // sampler share of G^ (two-role launch: factors per sample):  G^ += sum_s xhat_s [x_{t+1,s}' | 1]
      // (sample 0 without a branch: its loads go out with the others of the step)
      auto add_sample = [&](auto s) {
        const double* vo = ad + vjp_vec_off(N) + s * 2 * N;
        const double x1 = vo[N + cc];
        double xr[J];
        static_for<0, J>([&](auto j) { xr[j] = vo[ri[j]]; });
        const double vv = col ? x1 : EN;
        static_for<0, J>([&](auto j) { gb[j] = __builtin_fma(xr[j], vv, gb[j]); });
      };
      add_sample(std::integral_constant<int, 0>{});
      if (S > 1) static_for<1, VJP_SPLIT_MAX_S>([&](auto s) { if (s < S) add_sample(s); });
    }
    load_row<N + 1>(w + cc * HS, HT);               // H' (lane c: row c of H), the same in every DPP row
    // [Xbar | cbar] = [-G^ | G^[:,n]] - J12_t [Abar | hbar]_{t+1}      (my rows)
    double Xc[J];
    static_for<0, J>([&](auto j) { Xc[j] = colN ? gb[j] * sg * rm[j] : 0.0; });
    if (t < T - 1) {
      if (a.pair_t_stride != 0) {
        const double* pj = pJ12 + (long)t * a.pair_t_stride;
        static_for<0, J>([&](auto j) { const double v = pj[ri[j] * N + cc]; J12c[j] = col ? -v * rm[j] : 0.0; });
      }
      dpp_fence(J12c);
      static_for<0, N>([&](auto k) {
        static_for<0, J>([&](auto j) { mac_bc<k, true>(Xc[j], J12c[j], AbR[k]); });
      });
    }
    dpp_fence(Xc);
    double XcR[4 * J];
    static_for<0, J>([&](auto j) { quad_gather(Xc[j], XcR[4 * j], XcR[4 * j + 1], XcR[4 * j + 2], XcR[4 * j + 3]); });
    // Bbar = P^-1 [Xbar | cbar]
    double Bb[J];
    static_for<0, J>([&](auto j) { Bb[j] = 0.0; Pi[j] *= rm[j]; });
    dpp_fence(Pi);
    static_for<0, N>([&](auto k) {
      static_for<0, J>([&](auto j) { mac_bc<k>(Bb[j], Pi[j], XcR[k]); });
    });
#endif
    // Pbar = [sweep-1 shares] - Bbar H' - 1/2 g (c c' + P^-1)
    dpp_fence(Bb);
    static_for<0, N + 1>([&](auto k) {
      static_for<0, J>([&](auto j) { mac_bc<k, true>(Pb[j], Bb[j], HT[k]); });
    });
    double cvs[J];
    static_for<0, J>([&](auto j) { cvs[j] = -0.5 * g * Hc[j] * rm[j]; });       // lane N: -1/2 g c_i
    dpp_fence(cvs);
    static_for<0, J>([&](auto j) {
      mac_bc<N>(Pb[j], cvs[j], HT[N]);
      Pb[j] = __builtin_fma(-0.5 * g, Pi[j], Pb[j]);
    });
    // outputs and the adjoint handed to step t-1:  Ab = [Pbar | Bbar[:,n] + g c]
    double AbD[J], gJ = 0.0, gh = 0.0;
    static_for<0, J>([&](auto j) {
      const double hfb = __builtin_fma(g, Hc[j], Bb[j]);                          // lane N: hfbar_i
      AbD[j] = __builtin_fma(EN, hfb - Pb[j], Pb[j]) * rm[j];
      gJ = __builtin_fma(ED[j], Pb[j], gJ);
    });
    dpp_fence(AbD);
    static_for<0, J>([&](auto j) { mac_bc<N>(gh, AbD[j], ED[j]); });
    // [Abar | hbar] of this step, all rows in every DPP row, for the next step: through LDS (store [c][row], read back
    // [c][0..N-1]) -- the round trip runs under the next step's barrier and ring reads, which do not depend on it; as
    // v_permlane16/32_swap shuffles (quad_gather) it was 36 instructions at the end of every step (round 4)
#if SVAE_S2_LDS_GATHER
    __builtin_amdgcn_wave_barrier();
    static_for<0, J>([&](auto j) { gtile[c * RSG + 4 * j + r] = AbD[j]; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    static_for<0, (N + 1) / 2>([&](auto q) {
      const double2 v = reinterpret_cast<const double2*>(gtile + c * RSG)[q];
      AbR[2 * q] = v.x;
      if constexpr (2 * q + 1 < N) AbR[2 * q + 1] = v.y;
    });
    __builtin_amdgcn_wave_barrier();
#else
    double AbG[4 * J];
    static_for<0, J>([&](auto j) { quad_gather(AbD[j], AbG[4 * j], AbG[4 * j + 1], AbG[4 * j + 2], AbG[4 * j + 3]); });
    static_for<0, N>([&](auto i) { AbR[i] = AbG[i]; });
#endif
    if (olane) {
      a.g_node_J[(b * T + t) * N + c] = -2.0 * gJ;
      a.g_node_h[(b * T + t) * N + c] = gh;
    }
  }
}

template <int N, bool SAMP, bool SPLIT>
__global__ __launch_bounds__(64) void lds_vjp_sweep2_kernel(const VjpArgs a) {
  lds_vjp_sweep2_body<N, SAMP, SPLIT, false>(a);
}
// consumer wavefront + four producer wavefronts
template <int N, bool SAMP, bool SPLIT>
__global__ __launch_bounds__(320) void lds_vjp_sweep2_prod_kernel(const VjpArgs a) {
  lds_vjp_sweep2_body<N, SAMP, SPLIT, true>(a);
}

constexpr int VJP_PROD2_MAX_N = 12;     // sweep 2 with producers: beyond, the ring exceeds 64 KB and the consumer spills
template <int N>
static int launch_vjp(const VjpArgs& a, hipStream_t stream) {
  dim3 grid((a.B + 3) / 4), grid2(2 * ((a.B + 3) / 4)), block(64);
  const bool statc = a.g_E_init || a.g_E_pair;
  if (a.g_P && a.prod_max_b > 0) return -1001;       // (the dispatcher asks for the packed sweeps: only they write g_P)
  // two roles while 2 wavefronts per 4 sequences still find idle SIMDs (the sampler role hands its share of G^ to
  // sweep 2 as per-sample factors: room for VJP_SPLIT_MAX_S of them in the scratch record)
  const bool split = a.B <= 2048 && a.S <= VJP_SPLIT_MAX_S;
  if (a.g_samples) {
    if (statc && split) hipLaunchKernelGGL((lds_vjp_sweep1_kernel<N, true, true, true>), grid2, block, 0, stream, a);
    else if (statc) hipLaunchKernelGGL((lds_vjp_sweep1_kernel<N, true, true, false>), grid, block, 0, stream, a);
    else if (split && a.B <= a.prod_max_b && a.S <= VJP_PROD_MAX_S)
      hipLaunchKernelGGL((lds_vjp_sweep1_prod_kernel<N, true>), grid2, dim3(64 * VJP_S1_WAVES), 0, stream, a);
    else if (split) hipLaunchKernelGGL((lds_vjp_sweep1_kernel<N, true, false, true>), grid2, block, 0, stream, a);
    else hipLaunchKernelGGL((lds_vjp_sweep1_kernel<N, true, false, false>), grid, block, 0, stream, a);
    bool done2 = false;
    if (split && a.B <= a.prod_max_b && a.B <= VJP_S4_MAX_B) {
      hipLaunchKernelGGL((lds_vjp_sweep2_s4_kernel<N, true>), dim3(a.B), dim3(128), 0, stream, a);
      done2 = true;
    }
    if constexpr (N <= VJP_PROD2_MAX_N) {
      if (!done2 && split && a.B <= a.prod_max_b) {
        hipLaunchKernelGGL((lds_vjp_sweep2_prod_kernel<N, true, true>), grid, dim3(320), 0, stream, a);
        done2 = true;
      }
    }
    if (done2) {}
    else if (split) hipLaunchKernelGGL((lds_vjp_sweep2_kernel<N, true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lds_vjp_sweep2_kernel<N, true, false>), grid, block, 0, stream, a);
  } else {
    const bool prod = a.B <= a.prod_max_b;       // small batches: producer wavefronts hide the HBM latency of the serial sweeps
    if (statc) hipLaunchKernelGGL((lds_vjp_sweep1_kernel<N, false, true, false>), grid, block, 0, stream, a);
    else if (prod && a.B <= VJP_S4_MAX_B)
      hipLaunchKernelGGL((lds_vjp_sweep1_s4_kernel<N>), dim3(a.B), dim3(192), 0, stream, a);
    else if (prod) hipLaunchKernelGGL((lds_vjp_sweep1_prod_kernel<N, false>), grid, dim3(64 * VJP_S1_WAVES), 0, stream, a);
    else hipLaunchKernelGGL((lds_vjp_sweep1_kernel<N, false, false, false>), grid, block, 0, stream, a);
    bool done2 = false;
    if (prod && a.B <= VJP_S4_MAX_B) {
      hipLaunchKernelGGL((lds_vjp_sweep2_s4_kernel<N, false>), dim3(a.B), dim3(128), 0, stream, a);
      done2 = true;
    }
    if constexpr (N <= VJP_PROD2_MAX_N) {
      if (!done2 && prod) {
        hipLaunchKernelGGL((lds_vjp_sweep2_prod_kernel<N, false, false>), grid, dim3(320), 0, stream, a);
        done2 = true;
      }
    }
    if (!done2) hipLaunchKernelGGL((lds_vjp_sweep2_kernel<N, false, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace svae
