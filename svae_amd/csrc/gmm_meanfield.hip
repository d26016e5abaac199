// gmm_meanfield.hip -- GMM-SVAE local mean field (responsibility updates) for MI355X (gfx950, fp64).
//
// What it replaces (reference = mattjj/svae, /root/reference):
//   local_meanfield        svae/models/gmm.py:62-88
//   meanfield_fixed_point  svae/models/gmm.py:90-110   (block coordinate ascent, <= max_iter sweeps)
//   gaussian_meanfield     svae/models/gmm.py:112-117  -> gaussian.expectedstats / logZ
//                                                         (svae/distributions/gaussian.py:11-25)
//   label_meanfield        svae/models/gmm.py:119-124  -> categorical softmax / logsumexp
//                                                         (svae/distributions/categorical.py:6-9)
//
// Mapping to the hardware.  A minibatch is a few hundred to a few thousand points with N ~ 2 and
// K ~ 5..15: < 1 MB of state, so the whole fixed point runs as ONE launch of ONE workgroup (no
// host round trip per sweep -- the reference's loop is a host loop): one point per lane, the N x N
// solve/inverse/Cholesky in registers, the K global potentials G_k fetched with wave-uniform
// (scalar) loads, the batch-total KL the stopping rule needs reduced through LDS in a fixed order
// (bit-reproducible run to run).  It is launch-latency bound by construction.
//
// Dense packing (gaussian.py:39-57): a (N+2)x(N+2) block holds A in [:N,:N], b in [:N,N], c at
// [N,N], d at [N+1,N+1]; every other entry is zero.  G_k is required to have that sparsity (it is
// the output of niw.expectedstats, niw.py:25), node potentials have A diagonal.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/svae_hip.h"

namespace svae {

struct GmmArgs {
  int T, K, max_iter;
  double tol;
  const double* __restrict__ label_global;      // (K)
  const double* __restrict__ gaussian_globals;  // (K, D, D), D = N + 2
  const double* __restrict__ node_J;            // (T, N)
  const double* __restrict__ node_h;            // (T, N)
  const double* __restrict__ label_init;        // (T, K)
  double* label_stats;                          // (T, K)   (iterated in place)
  double* label_fixed;                          // (T, K) or nullptr: the fixed point itself
  double* gaussian_stats;                       // (T, D, D)
  double* label_natparam;                       // (T, K)
  double* gaussian_natparam;                    // (T, D, D)
  double* dirichlet_stats;                      // (K)
  double* niw_stats;                            // (K, D, D)
  double* kl;
  int32_t* iters;
  int32_t* assign;
  int32_t* info;
};

// threads per (single) workgroup: more registers per lane for larger N
template <int N> constexpr int gmm_block() { return N <= 3 ? 1024 : (N <= 5 ? 512 : 256); }

// Fixed-order sums (the SAME arithmetic in every kernel of this file, so that the KL totals -- and with them the stopping
// decisions -- agree bit for bit between the one-workgroup kernel's batches, the per-sweep launches and the persistent
// kernel).  wave_sum64: xor butterfly, every lane returns the same total (a + b == b + a bitwise).
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// block total = ((w0 + w1) + w2) + ... over the wavefronts' butterfly sums; every thread returns it.  ONE barrier:
// `red` holds two sets of wavefront sums, callers alternate `parity` between consecutive calls.
template <int GMM_BLOCK>
__device__ __forceinline__ double block_sum(double v, double* red, int parity) {
  constexpr int W = GMM_BLOCK / 64;
  v = wave_sum64(v);
  if constexpr (W == 1) return v;
  double* r = red + (parity & 1) * W;
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
  __syncthreads();
  double out = r[0];
#pragma unroll
  for (int w = 1; w < W; ++w) out += r[w];
  return out;
}

// Per-point Gaussian factor update: eta = node + sum_k r_k G_k  ->  (Ex, ExxT, logZ, <node,stats>).
template <int N>
struct PointGauss {
  double A[N][N];   // -1/2 J block of eta
  double h[N];
  double ab;        // eta[N,N] + eta[N+1,N+1]
  double Ex[N];
  double ExxT[N][N];
  double logZ;
  bool ok;
};

// Elementary functions of the per-point update, written out: the fixed point is a chain of ~50 dependent sweeps of one
// wavefront per 64 points, i.e. bound by the instruction count of ONE sweep, of which the library's exp / log / sqrt /
// IEEE division expansions were two thirds.  All are accurate to a few ulp (the tests hold reals to 1e-9, labels
// bit-exact).
__device__ __forceinline__ double gmm_rcp(double x) {           // 1 / x, x normal: v_rcp_f64 + two Newton steps
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// exp(x) for x <= 0 (the max-subtracted label natural parameters): x = n ln2 + r, |r| <= ln2 / 2, Taylor to r^13
// (remainder < 4e-18), scaled by 2^n; underflows to 0 like exp().
__device__ __forceinline__ double gmm_exp_nonpos(double x) {
  x = fmax(x, -746.0);
  const double n = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                                   // 1/13!
  p = __builtin_fma(p, r, 2.08767569878680989792e-09);                 // 1/12!
  p = __builtin_fma(p, r, 2.50521083854417187751e-08);                 // 1/11!
  p = __builtin_fma(p, r, 2.75573192239858906526e-07);                 // 1/10!
  p = __builtin_fma(p, r, 2.75573192239858906526e-06);                 // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873015873e-05);                 // 1/8!
  p = __builtin_fma(p, r, 1.98412698412698412698e-04);                 // 1/7!
  p = __builtin_fma(p, r, 1.38888888888888888889e-03);                 // 1/6!
  p = __builtin_fma(p, r, 8.33333333333333333333e-03);                 // 1/5!
  p = __builtin_fma(p, r, 4.16666666666666666667e-02);                 // 1/4!
  p = __builtin_fma(p, r, 1.66666666666666666667e-01);                 // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// log(x), x > 0 normal: x = 2^e m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716,
// odd series to s^23 (remainder < 1e-19).
__device__ __forceinline__ double gmm_log(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);                           // [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * gmm_rcp(2.0 + f);
  const double z = s * s;
  double p = 1.0 / 23.0;
  p = __builtin_fma(p, z, 1.0 / 21.0);
  p = __builtin_fma(p, z, 1.0 / 19.0);
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  const double t = (s * z) * p;                                        // atanh(s) - s
  const double de = (double)e;
  // e ln2 + 2 s + 2 t, the small terms first
  return __builtin_fma(de, 6.93147180369123816490e-01,
                       __builtin_fma(2.0, s, __builtin_fma(2.0, t, de * 1.90821492927058770002e-10)));
}

template <int N>
__device__ __forceinline__ void gauss_update(PointGauss<N>& g) {
  // J = -2 A (SPD) = L D L' (unit lower L, pivots d_j > 0 <=> SPD); Sigma = J^-1 = L^-T D^-1 L^-1, Ex = Sigma h,
  // sum_j log chol(J)_jj = 1/2 log prod_j d_j   [gaussian.py:11-25].  No square root, one reciprocal per pivot.
  double L[N][N], d[N], dinv[N];
  g.ok = true;
  double det = 1.0;       // prod_j d_j (N <= 8 factors of O(1e-3 .. 1e3): no overflow)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double v[N];          // v_m = L_jm d_m
    double dj = -2.0 * g.A[j][j];
#pragma unroll
    for (int m = 0; m < j; ++m) {
      v[m] = L[j][m] * d[m];
      dj = __builtin_fma(-L[j][m], v[m], dj);
    }
    g.ok = g.ok && (dj > 0.0);
    d[j] = dj;
    dinv[j] = gmm_rcp(dj);
    det *= dj;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = -2.0 * g.A[i][j];
#pragma unroll
      for (int m = 0; m < j; ++m) s = __builtin_fma(-L[i][m], v[m], s);
      L[i][j] = s * dinv[j];
    }
  }
  // Li = L^-1 (unit lower, strict part)
  double Li[N][N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = -L[i][j];
#pragma unroll
      for (int m = j + 1; m < i; ++m) s = __builtin_fma(-L[i][m], Li[m][j], s);
      Li[i][j] = s;
    }
  }
  double u[N];            // u = D^-1 L^-1 h
  double vv = 0.0;        // h' Sigma h
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = g.h[i];
#pragma unroll
    for (int m = 0; m < i; ++m) s = __builtin_fma(Li[i][m], g.h[m], s);
    u[i] = s * dinv[i];
    vv = __builtin_fma(s, u[i], vv);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = u[i];
#pragma unroll
    for (int m = i + 1; m < N; ++m) s = __builtin_fma(Li[m][i], u[m], s);
    g.Ex[i] = s;
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = (j == i) ? dinv[i] : dinv[i] * Li[i][j];
#pragma unroll
      for (int m = i + 1; m < N; ++m) s = __builtin_fma(Li[m][i] * dinv[m], Li[m][j], s);
      s = __builtin_fma(g.Ex[i], g.Ex[j], s);
      g.ExxT[i][j] = s;
      g.ExxT[j][i] = s;
    }
  g.logZ = 0.5 * vv - 0.5 * gmm_log(det) + g.ab;
}

// The K global potentials as a table in LDS, staged once per launch: per state k the entries the (N+2) x (N+2)
// dense packing actually holds -- [A (N x N) | b (N) | c | d | E log pi_k] -- padded to an even count (16-byte reads,
// every lane reads the same address: a broadcast); states K .. KP-1 are zero.  (Read through the kernel-argument
// pointers hipcc emitted a VECTOR load per entry, point and sweep -- ~400 L2 round trips per sweep.)
template <int N> constexpr int gmm_tab_stride() { return (N * N + N + 3 + 1) & ~1; }
__host__ __device__ constexpr int gmm_tab_states(int K) { return K <= 8 ? 8 : (K <= 16 ? 16 : K); }
template <int N> constexpr size_t gmm_tab_bytes(int K) { return (size_t)gmm_tab_states(K) * gmm_tab_stride<N>() * sizeof(double); }

template <int N>
__device__ __forceinline__ void gmm_stage_table(const GmmArgs& a, double* tab) {
  constexpr int D = N + 2, TS = gmm_tab_stride<N>();
  const int KP = gmm_tab_states(a.K);
  for (int q = threadIdx.x; q < KP * TS; q += blockDim.x) {
    const int k = q / TS, e = q % TS;
    double v = 0.0;
    if (k < a.K) {
      const double* G = a.gaussian_globals + (long)k * D * D;
      if (e < N * N) v = G[(e / N) * D + (e % N)];
      else if (e < N * N + N) v = G[(e - N * N) * D + N];
      else if (e == N * N + N) v = G[N * D + N];
      else if (e == N * N + N + 1) v = G[(N + 1) * D + N + 1];
      else if (e == N * N + N + 2) v = a.label_global[k];
    }
    tab[q] = v;
  }
  __syncthreads();
}

// One point of one sweep: Gaussian factor update from the responsibilities r (K), label update, this point's KL
// term.  final_pass = the extra pass of gmm.py:74-77 (no linear correction term, writes every per-point output).
// Shared by the single-workgroup kernel, the per-sweep launches and the persistent kernel: per-point results are
// bit-identical across them.  nJ / nh: the point's node potentials; tab: gmm_stage_table's LDS table.
//   KR > 0 (K <= KR, KR = 8 / 16): r, the label natural parameters and exp(l - max) live in REGISTERS -- r[] is the
//          point's state, old on entry (zero beyond K), new on exit; nothing is written unless final_pass (the callers
//          store r where their protocol needs it).  One exp per state and sweep; the multiply-add loops run over all
//          KR table rows (zero rows beyond K) without a branch.
//   KR = 0 (any K <= 64): the same arithmetic with label_natparam (T,K) as scratch between the K-loops and
//          label_stats as the state (rin -> label_stats[t]).
struct GmmNoHook { __device__ __forceinline__ void operator()() const {} };
// (mid: called once in the middle of the point's arithmetic -- the persistent kernel requests its partners' KL partials
//  there, so that the memory round trip runs under the second half of the sweep)
template <int N, int KR, class Hook = GmmNoHook>
__device__ __forceinline__ double gmm_point(const GmmArgs& a, const double* tab, const int t, const double (&nJ)[N],
                                            const double (&nh)[N], double (&r)[KR > 0 ? KR : 1], const double* rin,
                                            bool final_pass, Hook&& mid = Hook()) {
  constexpr int D = N + 2, TS = gmm_tab_stride<N>();
  constexpr int KU = KR > 0 ? KR : 1;
  const int K = a.K;
  // The table reads are loop-invariant and hipcc hoists them out of the sweep loop into registers: welcome while the
  // table is small (KR = 8, N <= 3: it never leaves the registers), 688 bytes of scratch per lane at KR = 16 -- there
  // the address passes through an empty asm, so that every sweep reads the LDS table afresh.
  if constexpr (KR * TS > 96) asm volatile("" : "+v"(tab));
  PointGauss<N> g;
  // eta = pack_dense(node) + sum_k r_k G_k          [gmm.py:113-114]
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) g.A[i][j] = 0.0;
    g.h[i] = 0.0;
  }
  double cN = 0.0, dN = 0.0;   // eta[N,N], eta[N+1,N+1]
  auto eta_term = [&](const double* G, double rk) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) g.A[i][j] = __builtin_fma(rk, G[i * N + j], g.A[i][j]);
      g.h[i] = __builtin_fma(rk, G[N * N + i], g.h[i]);
    }
    cN = __builtin_fma(rk, G[N * N + N], cN);
    dN = __builtin_fma(rk, G[N * N + N + 1], dN);
  };
  if constexpr (KR > 0) {
#pragma unroll
    for (int k = 0; k < KU; ++k) eta_term(tab + k * TS, r[k]);
  } else {
    for (int k = 0; k < K; ++k) eta_term(tab + k * TS, rin[k]);
  }
  g.ab = cN + dN;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    g.A[i][i] += nJ[i];
    g.h[i] += nh[i];
  }
  gauss_update<N>(g);
  mid();
  if (!g.ok) {
    int old = *(volatile int32_t*)a.info;
    while (old == 0 || old > t + 1) {
      const int seen = atomicCAS(a.info, old, t + 1);
      if (seen == old) break;
      old = seen;
    }
  }
  // gaussian_kl_t = <node, stats> - logZ(eta)        [gmm.py:116]
  double nodedot = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) nodedot += nJ[i] * g.ExxT[i][i] + nh[i] * g.Ex[i];
  double klt = nodedot - g.logZ;

  // label update: l_k = <stats, G_k>, natparam = l + label_global, r = softmax   [gmm.py:119-124]
  auto label_term = [&](const double* G) -> double {
    double l = G[N * N + N] + G[N * N + N + 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) l = __builtin_fma(g.ExxT[i][j], G[i * N + j], l);
      l = __builtin_fma(g.Ex[i], G[N * N + i], l);
    }
    return l + G[N * N + N + 2];
  };
  double mx = -1.0 / 0.0;
  double lab = 0.0, lin = 0.0, se = 0.0;
  int best = 0;
  double bestv = -1.0;
  if constexpr (KR > 0) {
    double np_[KU], ex[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) np_[k] = label_term(tab + k * TS);
#pragma unroll
    for (int k = 0; k < KU; ++k) mx = (k < K && np_[k] > mx) ? np_[k] : mx;
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      ex[k] = 0.0;
      if (k < K) { ex[k] = gmm_exp_nonpos(np_[k] - mx); se += ex[k]; }
    }
    const double lse = mx + gmm_log(se);
    const double inv = gmm_rcp(se);
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const double l = np_[k] - tab[k * TS + N * N + N + 2];     // (zero rows beyond K: l = 0, r = rnew = 0)
      const double rnew = ex[k] * inv;
      lab = __builtin_fma(rnew, l, lab);
      lin = __builtin_fma(r[k] - rnew, l, lin);     // <eta - sum_k rnew_k G_k - node, stats>, gmm.py:99-102
      if (k < K && rnew > bestv) { bestv = rnew; best = k; }
      r[k] = rnew;
    }
    if (final_pass) {
#pragma unroll
      for (int k = 0; k < KU; ++k) if (k < K) {
        a.label_natparam[(long)t * K + k] = np_[k];
        a.label_stats[(long)t * K + k] = r[k];
      }
    }
    klt += lab - lse;
  } else {
    for (int k = 0; k < K; ++k) {
      const double np_ = label_term(tab + k * TS);
      a.label_natparam[(long)t * K + k] = np_;   // scratch between the two k-loops
      mx = np_ > mx ? np_ : mx;
    }
    for (int k = 0; k < K; ++k) se += gmm_exp_nonpos(a.label_natparam[(long)t * K + k] - mx);
    const double lse = mx + gmm_log(se);
    const double inv = gmm_rcp(se);
    for (int k = 0; k < K; ++k) {
      const double np_ = a.label_natparam[(long)t * K + k];
      const double l = np_ - tab[k * TS + N * N + N + 2];
      const double rnew = gmm_exp_nonpos(np_ - mx) * inv;
      const double rold = rin[k];
      lab = __builtin_fma(rnew, l, lab);
      lin = __builtin_fma(rold - rnew, l, lin);
      if (rnew > bestv) { bestv = rnew; best = k; }
      a.label_stats[(long)t * K + k] = rnew;
    }
    klt += lab - lse;
  }
  if (!final_pass) klt += lin;

  if (final_pass) {
    a.assign[t] = best;
    double* gs = a.gaussian_stats + (long)t * D * D;
    double* gn = a.gaussian_natparam + (long)t * D * D;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double sv = 0.0, nv = 0.0;
        if (i < N && j < N) { sv = g.ExxT[i < N ? i : 0][j < N ? j : 0]; nv = g.A[i < N ? i : 0][j < N ? j : 0]; }
        else if (i < N && j == N) { sv = g.Ex[i < N ? i : 0]; nv = g.h[i < N ? i : 0]; }
        else if (i == j) { sv = 1.0; }
        gs[i * D + j] = sv;
        gn[i * D + j] = nv;
      }
    gn[N * D + N] = cN;
    gn[(N + 1) * D + N + 1] = dN;
  }
  return klt;
}

// State of a thread's point(s) across the sweeps of a kernel that runs the whole fixed point.  With at most ONE point
// per thread and K <= KR the responsibilities never leave the registers (loaded from label_init, stored with the final
// pass); otherwise label_stats in global memory is the state, as in the per-sweep launches.
template <int N, int KR>
struct GmmSweeper {
  static constexpr int KU = KR > 0 ? KR : 1;
  const GmmArgs& a;
  const double* tab;            // gmm_stage_table's LDS table
  const int first, stride;      // this thread's points: first, first + stride, ... < T
  const bool resident;
  double r[KU];
  double nJ[N], nh[N];
  __device__ GmmSweeper(const GmmArgs& a_, const double* tab_, int first_, int stride_, bool whole_fixed_point)
      : a(a_), tab(tab_), first(first_), stride(stride_), resident(KR > 0 && whole_fixed_point && a_.T <= stride_) {
    if (resident && first < a.T) { load(a.label_init, first); node(first); }
  }
  __device__ __forceinline__ void node(int t) {
#pragma unroll
    for (int i = 0; i < N; ++i) { nJ[i] = a.node_J[(long)t * N + i]; nh[i] = a.node_h[(long)t * N + i]; }
  }
  __device__ __forceinline__ void load(const double* src, int t) {
    if constexpr (KR > 0) {
#pragma unroll
      for (int k = 0; k < KU; ++k) r[k] = k < a.K ? src[(long)t * a.K + k] : 0.0;
    }
  }
  __device__ __forceinline__ void store(double* dst, int t) {
    if constexpr (KR > 0) {
#pragma unroll
      for (int k = 0; k < KU; ++k) if (k < a.K) dst[(long)t * a.K + k] = r[k];
    }
  }
  // one fixed-point sweep (from_init: the responsibilities come from label_init) -> this thread's KL partial
  // (RES: the caller's compile-time copy of `resident` -- the sweep loops of the kernels that run the whole fixed point
  //  are written out once per case, so that the register allocation of the resident loop is not the union of both: the
  //  merged loop carried 120 register moves per sweep)
  template <bool RES = false, class Hook = GmmNoHook>
  __device__ __forceinline__ double sweep(bool from_init, Hook&& mid = Hook()) {
    double klpart = 0.0;
    if constexpr (RES) {
      if (first < a.T) klpart = gmm_point<N, KR>(a, tab, first, nJ, nh, r, nullptr, false, mid);
      else mid();
      return klpart;
    }
    for (int t = first; t < a.T; t += stride) {
      const double* rin = (from_init ? a.label_init : a.label_stats) + (long)t * a.K;
      node(t);
      if constexpr (KR > 0) {
        load(from_init ? a.label_init : a.label_stats, t);
        klpart += gmm_point<N, KR>(a, tab, t, nJ, nh, r, nullptr, false);
        store(a.label_stats, t);
      } else {
        klpart += gmm_point<N, KR>(a, tab, t, nJ, nh, r, rin, false);
      }
    }
    return klpart;
  }
  // the final pass of gmm.py:74-86 (label_fixed <- the fixed point it starts from) -> this thread's KL partial
  __device__ __forceinline__ double final_pass(bool from_init) {
    double klpart = 0.0;
    for (int t = first; t < a.T; t += stride) {
      const double* src = from_init ? a.label_init : a.label_stats;
      if (!resident) node(t);
      if constexpr (KR > 0) {
        if (!resident) load(src, t);
        if (a.label_fixed) store(a.label_fixed, t);
        klpart += gmm_point<N, KR>(a, tab, t, nJ, nh, r, nullptr, true);
      } else {
        const double* rin = src + (long)t * a.K;
        if (a.label_fixed)
          for (int k = 0; k < a.K; ++k) a.label_fixed[(long)t * a.K + k] = rin[k];
        if (from_init)     // gmm_point<.., 0> reads rin and overwrites label_stats
          for (int k = 0; k < a.K; ++k) a.label_stats[(long)t * a.K + k] = rin[k];
        klpart += gmm_point<N, KR>(a, tab, t, nJ, nh, r, from_init ? a.label_stats + (long)t * a.K : rin, true);
      }
    }
    return klpart;
  }
};

template <int N, int KR>
__global__ __launch_bounds__(gmm_block<N>()) void gmm_meanfield_kernel(const GmmArgs a) {
  constexpr int D = N + 2;
  constexpr int GMM_BLOCK = gmm_block<N>();
  __shared__ double red[2 * (GMM_BLOCK / 64)];
  const int tid = threadIdx.x;
  const int T = a.T, K = a.K;
  extern __shared__ __attribute__((aligned(16))) double gmm_tab[];
  gmm_stage_table<N>(a, gmm_tab);
  GmmSweeper<N, KR> sw(a, gmm_tab, tid, GMM_BLOCK, true);

  // ---- fixed point [gmm.py:90-110] -------------------------------------------------------------
  double kl_prev = 1.0 / 0.0;
  int it = 0;
  auto run = [&](auto res) {
    for (int i = 0; i < a.max_iter; ++i) {
      it = i + 1;
      const double kl = block_sum<GMM_BLOCK>(sw.template sweep<decltype(res)::value>(i == 0), red, i);
      const bool stop = fabs(kl - kl_prev) < a.tol;
      kl_prev = kl;
      if (stop) break;
    }
  };
  if (KR > 0 && sw.resident) run(std::true_type{}); else run(std::false_type{});
  // ---- final pass + outputs [gmm.py:74-86] -----------------------------------------------------
  const double kl = block_sum<GMM_BLOCK>(sw.final_pass(a.max_iter == 0), red, it);
  if (tid == 0) { a.kl[0] = kl; a.iters[0] = it; }
  __syncthreads();
  __threadfence_block();

  // ---- global statistics: dirichlet_stats = sum_t r_t ; niw_stats_k = sum_t r_tk stats_t -------
  for (int j = tid; j < K * (1 + D * D); j += GMM_BLOCK) {
    if (j < K) {
      double s = 0.0;
      for (int t = 0; t < T; ++t) s += a.label_stats[(long)t * K + j];
      a.dirichlet_stats[j] = s;
    } else {
      const int k = (j - K) / (D * D), e = (j - K) % (D * D);
      double s = 0.0;
      for (int t = 0; t < T; ++t)
        s = __builtin_fma(a.label_stats[(long)t * K + k], a.gaussian_stats[(long)t * D * D + e], s);
      a.niw_stats[(long)k * D * D + e] = s;
    }
  }
}

// ---- multi-workgroup (and multi-GPU) variant ---------------------------------------------------------
// The stopping rule is on the BATCH-TOTAL KL (gmm.py:104-105), so a sweep over points spread across
// workgroups -- or across GPUs -- needs one scalar reduction per sweep.  Here every sweep is its own launch
// (enqueued back to back, no host round trip):
//   * each workgroup block-reduces its points' KL terms in a fixed order and writes partials[wg];
//   * the LAST workgroup to arrive (a ticket counter per sweep) sums the partials in index order -> kl_hist[i]:
//     bit-reproducible, independent of the arrival order;
//   * under torch.distributed the caller all-reduces kl_hist[i] in place between two launches (every rank
//     then holds the same bits);
//   * sweep i+1 starts by scanning kl_hist[0..i] for the first j with |kl_hist[j] - kl_hist[j-1]| < tol: if
//     there is one the fixed point was reached at sweep j and the launch is a no-op.  The host therefore
//     enqueues max_iter launches blindly; those after convergence cost a few microseconds each.
// State (caller-owned, svae_gmm_mw_workspace_bytes): kl_hist (max_iter + 1) | partials (G) | statistics
// partials (GS x K (1 + D^2)) | counters (max_iter + 3, int32, zeroed by svae_gmm_mw_begin).
struct GmmMwArgs {
  GmmArgs g;
  int sweep;            // index of this sweep (mode 0), or max_iter (mode 1)
  int mode;             // 0: fixed-point sweep, 1: final pass
  double* kl_hist;
  double* partials;
  int32_t* counters;
};
constexpr int GMM_MW_BLOCK = 256;

// first sweep index j < upto with |kl_hist[j] - kl_hist[j-1]| < tol (kl_hist[-1] = inf), or -1
__device__ __forceinline__ int gmm_converged_at(const double* kl_hist, int upto, double tol) {
  double prev = 1.0 / 0.0;
  for (int j = 0; j < upto; ++j) {
    const double kl = kl_hist[j];
    if (fabs(kl - prev) < tol) return j;
    prev = kl;
  }
  return -1;
}

// sum of partials[0..count) in a fixed order -- lane l adds partials l, l + 64, .. in ascending order, then the wavefront
// butterfly: any ONE wavefront computes it alone, every lane returns it
__device__ __forceinline__ double gmm_fixed_order_sum(const double* partials, int count) {
  double v = 0.0;
  for (int j = threadIdx.x & 63; j < count; j += 64) v += partials[j];
  return wave_sum64(v);
}

template <int N, int KR>
__global__ __launch_bounds__(GMM_MW_BLOCK) void gmm_mw_sweep_kernel(const GmmMwArgs m) {
  const GmmArgs& a = m.g;
  __shared__ double red[2 * (GMM_MW_BLOCK / 64)];
  __shared__ int sh_flag;
  const int tid = threadIdx.x;
  const int conv = gmm_converged_at(m.kl_hist, m.sweep, a.tol);     // same answer in every thread / workgroup
  if (m.mode == 0 && conv >= 0) return;                             // fixed point already reached
  const bool final_pass = m.mode == 1;
  const bool from_init = final_pass ? (a.max_iter == 0) : (m.sweep == 0);
  extern __shared__ __attribute__((aligned(16))) double gmm_tab[];
  gmm_stage_table<N>(a, gmm_tab);
  GmmSweeper<N, KR> sw(a, gmm_tab, blockIdx.x * GMM_MW_BLOCK + tid, gridDim.x * GMM_MW_BLOCK, false);
  const double klpart = final_pass ? sw.final_pass(from_init) : sw.sweep(from_init);
  const double wgsum = block_sum<GMM_MW_BLOCK>(klpart, red, 0);
  if (tid == 0) {
    m.partials[blockIdx.x] = wgsum;
    __threadfence();
    const int ticket = atomicAdd(&m.counters[final_pass ? a.max_iter + 1 : m.sweep], 1);
    sh_flag = (ticket == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!sh_flag) return;
  __threadfence();
  const double total = gmm_fixed_order_sum(m.partials, gridDim.x);
  if (tid == 0) {
    if (final_pass) {
      a.kl[0] = total;                                   // (this rank's points; the caller sums over ranks)
      a.iters[0] = conv >= 0 ? conv + 1 : a.max_iter;
    } else {
      m.kl_hist[m.sweep] = total;
    }
  }
}

// The same sweeps in ONE launch (single GPU): a grid-wide exchange of the workgroups' KL partials per sweep instead of
// a launch per sweep (16 .. 21 us each, the whole cost of a sweep below ~100 k points).  The exchange is NOT a
// barrier + reduction but an all-to-all of data-tagged granules (MI355X: per-XCD L2s are not coherent, so every shared
// word is an agent-scope access): workgroup w publishes its partial as two 8-byte {tag = sweep + 1, half of the double}
// words with relaxed agent-scope atomic stores (write-through: no release fence, the data is the flag); thread j of
// every workgroup polls the two words of workgroup j with relaxed agent-scope loads until both tags match -- no
// atomic read-modify-write, no fence, no cooperative-groups grid sync (software on ROCm 7.2: ~26 us per sync) -- then
// every workgroup sums the partials in index order with the per-sweep kernels' own gmm_fixed_order_sum arithmetic: all
// reach the same total, bit for bit, and the same stopping decision.  The slots are double-buffered by sweep parity (a
// workgroup can be at most one exchange ahead of the slowest), a third set serves the final pass.  A plain launch:
// the grid (<= 32 workgroups in the window this path serves, capped by the occupancy query) is co-resident on any
// idle-enough device; every spin is bounded and reports through `info` (-77) instead of hanging.
// With at most one point per thread and K <= 16 the responsibilities stay in registers for the whole fixed point.
#ifndef SVAE_GMM_RUN_AHEAD
#define SVAE_GMM_RUN_AHEAD 1     // persistent kernel: compute the next sweep before collecting the previous exchange (0: A/B)
#endif
struct GmmSlot { unsigned long long lo, hi; };          // {tag << 32 | low half}, {tag << 32 | high half}
constexpr int GMM_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ void gmm_publish(GmmSlot* slot, unsigned tag, double v) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(&slot->lo, ((unsigned long long)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&slot->hi, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// total of the G workgroups' partials of exchange `tag`: thread 0 publishes this workgroup's, then EVERY wavefront polls
// all G slots (lane l: slots l, l + 64, ..) and sums them with gmm_fixed_order_sum's arithmetic -- no workgroup barrier
// behind the exchange; every thread returns the total.  G == 1: nothing to exchange.
// (the collecting half alone: every wavefront polls all G slots of exchange `tag`, this workgroup's own included)
__device__ __forceinline__ double gmm_collect(GmmSlot* slots, int G, unsigned tag, int32_t* info) {
  double v = 0.0;
  for (int j = threadIdx.x & 63; j < G; j += 64) {
    unsigned long long lo = 0, hi = 0;
    int spins = 0;
    for (;;) {
      lo = __hip_atomic_load(&slots[j].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hi = __hip_atomic_load(&slots[j].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
      if (++spins > GMM_SPIN_LIMIT) { atomicMin(info, -77); break; }     // never hang the device
      __builtin_amdgcn_s_sleep(1);
    }
    v += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
  }
  return wave_sum64(v);
}

__device__ __forceinline__ double gmm_exchange(GmmSlot* slots, int G, unsigned tag, double mine, int32_t* info) {
  if (G == 1) return wave_sum64((threadIdx.x & 63) == 0 ? mine : 0.0);
  if (threadIdx.x == 0) gmm_publish(slots + blockIdx.x, tag, mine);
  double v = 0.0;
  for (int j = threadIdx.x & 63; j < G; j += 64) {
    unsigned long long lo = 0, hi = 0;
    int spins = 0;
    for (;;) {
      lo = __hip_atomic_load(&slots[j].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hi = __hip_atomic_load(&slots[j].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
      if (++spins > GMM_SPIN_LIMIT) { atomicMin(info, -77); break; }     // never hang the device
      __builtin_amdgcn_s_sleep(1);
    }
    v += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
  }
  return wave_sum64(v);
}

template <int N, int KR>
__global__ __launch_bounds__(GMM_MW_BLOCK) void gmm_mw_persistent_kernel(const GmmMwArgs m) {
  const GmmArgs& a = m.g;
  __shared__ double red[2 * (GMM_MW_BLOCK / 64)];
  const int tid = threadIdx.x, G = gridDim.x;
  GmmSlot* slots = reinterpret_cast<GmmSlot*>(m.partials);
  extern __shared__ __attribute__((aligned(16))) double gmm_tab[];
  gmm_stage_table<N>(a, gmm_tab);
  GmmSweeper<N, KR> sw(a, gmm_tab, blockIdx.x * GMM_MW_BLOCK + tid, G * GMM_MW_BLOCK, true);
  double prev = 1.0 / 0.0;
  int iters = 0;
#ifdef SVAE_GMM_TIMING      // tools/gmm_phase_timing.py: shader cycles per phase, wall clock (100 MHz) of the loop
  long long tc[3] = {0, 0, 0}, t0_ = __builtin_readcyclecounter();
  const unsigned long long w0_ = __builtin_amdgcn_s_memrealtime();
#define GMM_TICK(k) { const long long n_ = __builtin_readcyclecounter(); tc[k] += n_ - t0_; t0_ = n_; }
#else
#define GMM_TICK(k)
#endif
  auto run = [&](auto res) {
    for (int i = 0; i < a.max_iter; ++i) {
      iters = i + 1;
      const double klpart = sw.template sweep<decltype(res)::value>(i == 0);
      GMM_TICK(0)
      const double wgsum = block_sum<GMM_MW_BLOCK>(klpart, red, i);
      GMM_TICK(1)
      const double total = gmm_exchange(slots + (i & 1) * G, G, (unsigned)(i + 1), wgsum, a.info);
      GMM_TICK(2)
      if (blockIdx.x == 0 && tid == 0) m.kl_hist[i] = total;
      const bool stop = fabs(total - prev) < a.tol;
      prev = total;
      if (stop) break;
    }
  };
  // Responsibilities in registers, several workgroups: the exchange of sweep i - 1 (~0.9 us: a store has to reach the
  // other XCDs) is collected AFTER sweep i has been computed -- speculatively: a sweep reads and writes only this thread's
  // registers, its result does not depend on the total, and when the total of sweep i - 1 says "converged" the
  // registers are put back.  Same sweeps, same totals, same stopping sweep; one discarded sweep of arithmetic at the end
  // against an exchange latency per sweep (round 4: 3.1 -> ~2.2 us per sweep at 1000 points, four workgroups).
  auto run_ahead = [&]() {
    if (a.max_iter <= 0) return;
    double wgsum = block_sum<GMM_MW_BLOCK>(sw.template sweep<true>(true), red, 0);
    if (tid == 0) gmm_publish(slots + blockIdx.x, 1u, wgsum);
    iters = 1;
    for (int i = 1;; ++i) {
      const bool more = i < a.max_iter;
      double keep[KR > 0 ? KR : 1], klnext = 0.0;
      // the partners' partials of exchange i - 1 are REQUESTED in the middle of sweep i (they were published ~1 us
      // ago: a store needs ~0.9 us to become visible across XCDs) and looked at behind it: the request's own round
      // trip (~0.35 us) runs under the second half of the sweep; a slot whose tag is still old is polled as before
      GmmSlot* ex = slots + ((i - 1) & 1) * G;
      const int myslot = (tid & 63) < G ? (tid & 63) : 0;
      unsigned long long elo = 0, ehi = 0;
      auto request = [&]() {
        elo = __hip_atomic_load(&ex[myslot].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ehi = __hip_atomic_load(&ex[myslot].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      if (more) {
#pragma unroll
        for (int k = 0; k < (KR > 0 ? KR : 1); ++k) keep[k] = sw.r[k];
        klnext = sw.template sweep<true>(false, request);
      } else {
        request();
      }
      GMM_TICK(0)
      double total;
      {
        const unsigned tag = (unsigned)i;
        const bool fresh = (unsigned)(elo >> 32) == tag && (unsigned)(ehi >> 32) == tag;
        if (G <= 64 && !__any(!fresh)) {
          const double v = (tid & 63) < G ? __longlong_as_double((long long)((ehi << 32) | (elo & 0xffffffffull))) : 0.0;
          total = wave_sum64(v);                        // (gmm_collect's arithmetic: lane j holds slot j, ascending butterfly)
        } else {
          total = gmm_collect(ex, G, tag, a.info);
        }
      }
      GMM_TICK(2)
      if (blockIdx.x == 0 && tid == 0) m.kl_hist[i - 1] = total;
      const bool stop = fabs(total - prev) < a.tol;
      prev = total;
      if (stop || !more) {
        if (more) {
#pragma unroll
          for (int k = 0; k < (KR > 0 ? KR : 1); ++k) sw.r[k] = keep[k];
        }
        break;
      }
      iters = i + 1;
      wgsum = block_sum<GMM_MW_BLOCK>(klnext, red, i);
      GMM_TICK(1)
      if (tid == 0) gmm_publish(slots + (i & 1) * G + blockIdx.x, (unsigned)(i + 1), wgsum);
    }
  };
  if (KR > 0 && sw.resident) { if (G > 1 && SVAE_GMM_RUN_AHEAD) run_ahead(); else run(std::true_type{}); }
  else run(std::false_type{});
#ifdef SVAE_GMM_TIMING
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == G - 1))
    printf("gmm persistent wg %d/%d: %d sweeps, cycles per sweep: point %lld  block_sum %lld  exchange %lld | loop wall %.2f us per sweep\n",
           (int)blockIdx.x, G, iters, tc[0] / iters, tc[1] / iters, tc[2] / iters,
           (double)(__builtin_amdgcn_s_memrealtime() - w0_) / 100.0 / iters);
#endif
  // final pass (gmm.py:74-86)
  const double wgsum = block_sum<GMM_MW_BLOCK>(sw.final_pass(a.max_iter == 0), red, iters);
  if (blockIdx.x != 0) {
    if (tid == 0) gmm_publish(slots + 2 * G + blockIdx.x, (unsigned)(a.max_iter + 2), wgsum);
    return;
  }
  const double total = gmm_exchange(slots + 2 * G, G, (unsigned)(a.max_iter + 2), wgsum, a.info);
  if (tid == 0) { a.kl[0] = total; a.iters[0] = iters; }
}

// global statistics  dirichlet_stats = sum_t r_t,  niw_stats_k = sum_t r_tk stats_t  over many workgroups:
// thread j of a workgroup accumulates output j over the workgroup's slice of points; the last workgroup to
// arrive sums the per-workgroup partials in index order.
__global__ __launch_bounds__(GMM_MW_BLOCK) void gmm_mw_stats_kernel(const GmmArgs a, int D, double* spart,
                                                                   int32_t* counter) {
  __shared__ int sh_flag;
  const int K = a.K, T = a.T, NO = K * (1 + D * D);
  const int per = (T + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(T, t0 + per);
  // (short slices -- 16 points per workgroup up to 16 k points -- and loads that do not depend on the running sum: the
  //  first version walked 250 points per thread with one dependent L2 round trip each, 59 us at 1000 points)
  for (int j = threadIdx.x; j < NO; j += GMM_MW_BLOCK) {
    double s = 0.0;
    if (j < K) {
#pragma unroll 8
      for (int t = t0; t < t1; ++t) s += a.label_stats[(long)t * K + j];
    } else {
      const int k = (j - K) / (D * D), e = (j - K) % (D * D);
#pragma unroll 8
      for (int t = t0; t < t1; ++t)
        s = __builtin_fma(a.label_stats[(long)t * K + k], a.gaussian_stats[(long)t * D * D + e], s);
    }
    spart[(long)blockIdx.x * NO + j] = s;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) sh_flag = (atomicAdd(counter, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!sh_flag) return;
  __threadfence();
  for (int j = threadIdx.x; j < NO; j += GMM_MW_BLOCK) {
    double s = 0.0;
#pragma unroll 8
    for (int w = 0; w < (int)gridDim.x; ++w) s += spart[(long)w * NO + j];
    if (j < K) a.dirichlet_stats[j] = s; else a.niw_stats[j - K] = s;
  }
}

// (N, K) -> the kernel instance: N = 1 .. 8 compile-time; KR = 8 / 16 (responsibilities in registers: N <= 4, where
// that form fits the register file without scratch) / 0 (K <= 64, any N)
template <bool REG_FORM, typename Fn>
static int gmm_dispatch(int N, int K, Fn&& fn) {
  auto byk = [&](auto n) -> int {
    if constexpr (REG_FORM && decltype(n)::value <= 4) {
      if (K <= 8) return fn(n, std::integral_constant<int, 8>{});
      if (K <= 16) return fn(n, std::integral_constant<int, 16>{});
    }
    return fn(n, std::integral_constant<int, 0>{});
  };
  switch (N) {
    case 1: return byk(std::integral_constant<int, 1>{});
    case 2: return byk(std::integral_constant<int, 2>{});
    case 3: return byk(std::integral_constant<int, 3>{});
    case 4: return byk(std::integral_constant<int, 4>{});
    case 5: return byk(std::integral_constant<int, 5>{});
    case 6: return byk(std::integral_constant<int, 6>{});
    case 7: return byk(std::integral_constant<int, 7>{});
    case 8: return byk(std::integral_constant<int, 8>{});
  }
  return -2;
}

}  // namespace svae

extern "C" int svae_gmm_meanfield_f64(int T, int N, int K,
                                      const double* label_global, const double* gaussian_globals,
                                      const double* node_J, const double* node_h,
                                      const double* label_init, double tol, int max_iter,
                                      double* label_stats, double* label_fixed,
                                      double* gaussian_stats,
                                      double* label_natparam, double* gaussian_natparam,
                                      double* dirichlet_stats, double* niw_stats,
                                      double* kl, int32_t* iters, int32_t* assign,
                                      int32_t* info, void* stream) {
  if (T < 0) return -1;
  if (N < 1 || N > 8) return -2;
  if (K < 1 || K > 64) return -3;
  if (!label_global) return -4;
  if (!gaussian_globals) return -5;
  if (T > 0 && !node_J) return -6;
  if (T > 0 && !node_h) return -7;
  if (T > 0 && !label_init) return -8;
  if (!(tol >= 0.0)) return -9;
  if (max_iter < 0) return -10;
  if (T > 0 && (!label_stats || !gaussian_stats || !label_natparam || !gaussian_natparam)) return -11;
  if (!dirichlet_stats || !niw_stats || !kl || !iters) return -15;
  if (T > 0 && !assign) return -19;
  if (!info) return -20;
  svae::GmmArgs a;
  a.T = T; a.K = K; a.max_iter = max_iter; a.tol = tol;
  a.label_global = label_global; a.gaussian_globals = gaussian_globals;
  a.node_J = node_J; a.node_h = node_h; a.label_init = label_init;
  a.label_stats = label_stats; a.label_fixed = label_fixed; a.gaussian_stats = gaussian_stats;
  a.label_natparam = label_natparam; a.gaussian_natparam = gaussian_natparam;
  a.dirichlet_stats = dirichlet_stats; a.niw_stats = niw_stats;
  a.kl = kl; a.iters = iters; a.assign = assign; a.info = info;
  hipStream_t s = (hipStream_t)stream;
  // (one workgroup of up to 1024 threads: 128 registers per lane -- the responsibilities-in-registers form does not fit)
  return svae::gmm_dispatch<false>(N, K, [&](auto n, auto kr) -> int {
    constexpr int NN = decltype(n)::value, KR = decltype(kr)::value;
    hipLaunchKernelGGL((svae::gmm_meanfield_kernel<NN, KR>), dim3(1), dim3(svae::gmm_block<NN>()),
                       svae::gmm_tab_bytes<NN>(K), s, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  });
}

// ---- multi-workgroup entry points (see the block comment above gmm_mw_sweep_kernel) ---------------------
namespace svae {
static int gmm_mw_grid(int T) {
  const int g = (T + GMM_MW_BLOCK - 1) / GMM_MW_BLOCK;
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}
static int gmm_mw_stats_grid(int T) {
  const int g = (T + 15) / 16;          // 16 points per workgroup, at most 1024 workgroups
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}
struct GmmMwLayout { double* kl_hist; double* partials; double* spart; int32_t* counters; };
static size_t gmm_mw_doubles(int T, int N, int K, int max_iter) {
  const int D = N + 2;
  // kl_hist (rounded up to an even count: the exchange slots are 16-byte words) | 3 sets of G 16-byte slots | statistics partials
  return (size_t)((max_iter + 2) & ~1) + 6 * gmm_mw_grid(T) + (size_t)gmm_mw_stats_grid(T) * K * (1 + D * D);
}
static GmmMwLayout gmm_mw_layout(void* ws, int T, int N, int K, int max_iter) {
  const int D = N + 2;
  GmmMwLayout l;
  l.kl_hist = (double*)ws;
  l.partials = l.kl_hist + ((max_iter + 2) & ~1);
  l.spart = l.partials + 6 * gmm_mw_grid(T);
  l.counters = (int32_t*)(l.spart + (size_t)gmm_mw_stats_grid(T) * K * (1 + D * D));
  return l;
}
}  // namespace svae

extern "C" size_t svae_gmm_mw_workspace_bytes(int T, int N, int K, int max_iter) {
  if (T < 0 || N < 1 || N > 8 || K < 1 || K > 64 || max_iter < 0) return 0;
  return svae::gmm_mw_doubles(T, N, K, max_iter) * sizeof(double) + (size_t)(max_iter + 3) * sizeof(int32_t);
}

extern "C" int svae_gmm_mw_begin(int T, int N, int K, int max_iter, void* workspace, size_t ws_bytes, void* stream) {
  if (T < 0) return -1;
  if (N < 1 || N > 8) return -2;
  if (K < 1 || K > 64) return -3;
  if (max_iter < 0) return -4;
  if (!workspace || ws_bytes < svae_gmm_mw_workspace_bytes(T, N, K, max_iter)) return -5;
  // kl_hist and the ticket counters start at zero
  if (hipMemsetAsync(workspace, 0, svae_gmm_mw_workspace_bytes(T, N, K, max_iter), (hipStream_t)stream) != hipSuccess)
    return -1000;
  return 0;
}

// phase: 0 = fixed-point sweep number `sweep` (0 <= sweep < max_iter), 1 = final pass, 2 = global statistics
extern "C" int svae_gmm_mw_step_f64(int phase, int sweep, int T, int N, int K,
                                    const double* label_global, const double* gaussian_globals,
                                    const double* node_J, const double* node_h,
                                    const double* label_init, double tol, int max_iter,
                                    double* label_stats, double* label_fixed, double* gaussian_stats,
                                    double* label_natparam, double* gaussian_natparam,
                                    double* dirichlet_stats, double* niw_stats,
                                    double* kl, int32_t* iters, int32_t* assign, int32_t* info,
                                    void* workspace, size_t ws_bytes, void* stream) {
  if (phase < 0 || phase > 2) return -1;
  if (phase == 0 && (sweep < 0 || sweep >= max_iter)) return -2;
  if (T < 0) return -3;
  if (N < 1 || N > 8) return -4;
  if (K < 1 || K > 64) return -5;
  if (!label_global) return -6;
  if (!gaussian_globals) return -7;
  if (T > 0 && (!node_J || !node_h || !label_init)) return -8;
  if (!(tol >= 0.0)) return -11;
  if (max_iter < 0) return -12;
  if (T > 0 && (!label_stats || !gaussian_stats || !label_natparam || !gaussian_natparam)) return -13;
  if (!dirichlet_stats || !niw_stats || !kl || !iters) return -18;
  if (T > 0 && !assign) return -22;
  if (!info) return -23;
  if (!workspace || ws_bytes < svae_gmm_mw_workspace_bytes(T, N, K, max_iter)) return -24;
  svae::GmmMwArgs m;
  svae::GmmArgs& a = m.g;
  a.T = T; a.K = K; a.max_iter = max_iter; a.tol = tol;
  a.label_global = label_global; a.gaussian_globals = gaussian_globals;
  a.node_J = node_J; a.node_h = node_h; a.label_init = label_init;
  a.label_stats = label_stats; a.label_fixed = label_fixed; a.gaussian_stats = gaussian_stats;
  a.label_natparam = label_natparam; a.gaussian_natparam = gaussian_natparam;
  a.dirichlet_stats = dirichlet_stats; a.niw_stats = niw_stats;
  a.kl = kl; a.iters = iters; a.assign = assign; a.info = info;
  const svae::GmmMwLayout l = svae::gmm_mw_layout(workspace, T, N, K, max_iter);
  m.kl_hist = l.kl_hist; m.partials = l.partials; m.counters = l.counters;
  hipStream_t s = (hipStream_t)stream;
  if (phase == 2) {
    hipLaunchKernelGGL(svae::gmm_mw_stats_kernel, dim3(svae::gmm_mw_stats_grid(T)), dim3(svae::GMM_MW_BLOCK), 0, s,
                       a, N + 2, l.spart, l.counters + max_iter + 2);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  }
  m.mode = phase;
  m.sweep = phase == 0 ? sweep : max_iter;
  return svae::gmm_dispatch<true>(N, K, [&](auto n, auto kr) -> int {
    constexpr int NN = decltype(n)::value, KR = decltype(kr)::value;
    hipLaunchKernelGGL((svae::gmm_mw_sweep_kernel<NN, KR>), dim3(svae::gmm_mw_grid(m.g.T)), dim3(svae::GMM_MW_BLOCK),
                       svae::gmm_tab_bytes<NN>(K), s, m);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  });
}

/* device address of kl_hist[0] inside the workspace (the scalar a multi-GPU caller all-reduces per sweep) */
extern "C" double* svae_gmm_mw_kl_hist(void* workspace) { return (double*)workspace; }

// The whole fixed point + final pass in ONE launch (an all-to-all of tagged KL partials per sweep, see
// gmm_mw_persistent_kernel), then the statistics launch: the single-GPU form of the svae_gmm_mw_* sweeps (same per-point
// code, same fixed-order KL reduction, same results).  Returns -50 only if the device cannot be queried (callers may
// then use svae_gmm_mw_step_f64 -- and must say so: the Python layer warns and records the path).
extern "C" int svae_gmm_mw_fixed_point_f64(int T, int N, int K,
                                           const double* label_global, const double* gaussian_globals,
                                           const double* node_J, const double* node_h,
                                           const double* label_init, double tol, int max_iter,
                                           double* label_stats, double* label_fixed, double* gaussian_stats,
                                           double* label_natparam, double* gaussian_natparam,
                                           double* dirichlet_stats, double* niw_stats,
                                           double* kl, int32_t* iters, int32_t* assign, int32_t* info,
                                           void* workspace, size_t ws_bytes, void* stream) {
  if (T < 0) return -3;
  if (N < 1 || N > 8) return -4;
  if (K < 1 || K > 64) return -5;
  if (!label_global) return -6;
  if (!gaussian_globals) return -7;
  if (T > 0 && (!node_J || !node_h || !label_init)) return -8;
  if (!(tol >= 0.0)) return -11;
  if (max_iter < 0) return -12;
  if (T > 0 && (!label_stats || !gaussian_stats || !label_natparam || !gaussian_natparam)) return -13;
  if (!dirichlet_stats || !niw_stats || !kl || !iters) return -18;
  if (T > 0 && !assign) return -22;
  if (!info) return -23;
  if (!workspace || ws_bytes < svae_gmm_mw_workspace_bytes(T, N, K, max_iter)) return -24;
  svae::GmmMwArgs m;
  svae::GmmArgs& a = m.g;
  a.T = T; a.K = K; a.max_iter = max_iter; a.tol = tol;
  a.label_global = label_global; a.gaussian_globals = gaussian_globals;
  a.node_J = node_J; a.node_h = node_h; a.label_init = label_init;
  a.label_stats = label_stats; a.label_fixed = label_fixed; a.gaussian_stats = gaussian_stats;
  a.label_natparam = label_natparam; a.gaussian_natparam = gaussian_natparam;
  a.dirichlet_stats = dirichlet_stats; a.niw_stats = niw_stats;
  a.kl = kl; a.iters = iters; a.assign = assign; a.info = info;
  const svae::GmmMwLayout l = svae::gmm_mw_layout(workspace, T, N, K, max_iter);
  m.kl_hist = l.kl_hist; m.partials = l.partials; m.counters = l.counters; m.sweep = 0; m.mode = 0;
  hipStream_t s = (hipStream_t)stream;
  // every polled word starts at zero (tags are sweep + 1 .. max_iter + 2); the counters section is the statistics
  // kernel's ticket
  if (hipMemsetAsync(workspace, 0, svae_gmm_mw_workspace_bytes(T, N, K, max_iter), s) != hipSuccess) return -1000;
  const int rc = svae::gmm_dispatch<true>(N, K, [&](auto n, auto kr) -> int {
    constexpr int NN = decltype(n)::value, KR = decltype(kr)::value;
    auto kern = svae::gmm_mw_persistent_kernel<NN, KR>;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();      // -50 = "use the per-sweep launches instead": leave no sticky error behind for them
      return -50;
    }
    // co-residency of the grid (the workgroups wait for each other): at most ONE 256-thread workgroup per CU -- far
    // inside what the device holds (no reliance on the occupancy query, which is one block per CU high at some SGPR counts)
    const int cap = cus;
    int G = svae::gmm_mw_grid(T);
    if (cap < 1) return -50;
    if (G > cap) G = cap;
    hipLaunchKernelGGL(kern, dim3(G), dim3(svae::GMM_MW_BLOCK), svae::gmm_tab_bytes<NN>(K), s, m);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  });
  if (rc != 0) return rc;
  hipLaunchKernelGGL(svae::gmm_mw_stats_kernel, dim3(svae::gmm_mw_stats_grid(T)), dim3(svae::GMM_MW_BLOCK), 0, s,
                     a, N + 2, l.spart, l.counters + max_iter + 2);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
