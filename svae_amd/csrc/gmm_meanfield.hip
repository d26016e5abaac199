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

// fixed-order block reduction (every thread returns the total)
template <int GMM_BLOCK>
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = GMM_BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
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

template <int N>
__device__ __forceinline__ void gauss_update(PointGauss<N>& g) {
  // J = -2 A (SPD); Cholesky J = L L', Sigma = J^-1, Ex = Sigma h   [gaussian.py:11-25]
  double L[N][N];
  g.ok = true;
  double sumlog = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = -2.0 * g.A[j][j];
#pragma unroll
    for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
    g.ok = g.ok && (d > 0.0);
    const double l = sqrt(d);
    L[j][j] = l;
    sumlog += log(l);
    const double inv = 1.0 / l;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = -2.0 * g.A[i][j];
#pragma unroll
      for (int m = 0; m < j; ++m) s -= L[i][m] * L[j][m];
      L[i][j] = s * inv;
    }
  }
  // Linv (lower), Sigma = Linv' Linv
  double Li[N][N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    Li[j][j] = 1.0 / L[j][j];
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int m = j; m < i; ++m) s -= L[i][m] * Li[m][j];
      Li[i][j] = s / L[i][i];
    }
  }
  double v[N];   // v = Linv h
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int m = 0; m <= i; ++m) s += Li[i][m] * g.h[m];
    v[i] = s;
  }
  double vv = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) vv += v[i] * v[i];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int m = i; m < N; ++m) s += Li[m][i] * v[m];
    g.Ex[i] = s;
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int m = i; m < N; ++m) s += Li[m][i] * Li[m][j];
      s += g.Ex[i] * g.Ex[j];
      g.ExxT[i][j] = s;
      g.ExxT[j][i] = s;
    }
  g.logZ = 0.5 * vv - sumlog + g.ab;
}

// One point of one sweep: Gaussian factor update from the responsibilities `rin` (K), label update,
// this point's KL term.  final_pass = the extra pass of gmm.py:74-77 (no linear correction term, writes
// every output).  Shared by the single-workgroup kernel and the multi-workgroup sweeps.
template <int N>
__device__ __forceinline__ double gmm_point(const GmmArgs& a, int t, const double* rin, bool final_pass) {
  constexpr int D = N + 2;
  const int K = a.K;
  PointGauss<N> g;
  // eta = pack_dense(node) + sum_k r_k G_k          [gmm.py:113-114]
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) g.A[i][j] = 0.0;
    g.h[i] = 0.0;
  }
  double cN = 0.0, dN = 0.0;   // eta[N,N], eta[N+1,N+1]
  for (int k = 0; k < K; ++k) {
    const double r = rin[k];
    const double* G = a.gaussian_globals + (long)k * D * D;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) g.A[i][j] = __builtin_fma(r, G[i * D + j], g.A[i][j]);
      g.h[i] = __builtin_fma(r, G[i * D + N], g.h[i]);
    }
    cN = __builtin_fma(r, G[N * D + N], cN);
    dN = __builtin_fma(r, G[(N + 1) * D + N + 1], dN);
  }
  g.ab = cN + dN;
  double nJ[N], nh[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    nJ[i] = a.node_J[(long)t * N + i];
    nh[i] = a.node_h[(long)t * N + i];
    g.A[i][i] += nJ[i];
    g.h[i] += nh[i];
  }
  gauss_update<N>(g);
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
  double mx = -1.0 / 0.0;
  for (int k = 0; k < K; ++k) {
    const double* G = a.gaussian_globals + (long)k * D * D;
    double l = G[N * D + N] + G[(N + 1) * D + N + 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) l = __builtin_fma(g.ExxT[i][j], G[i * D + j], l);
      l = __builtin_fma(g.Ex[i], G[i * D + N], l);
    }
    const double np_ = l + a.label_global[k];
    a.label_natparam[(long)t * K + k] = np_;   // scratch between the two k-loops
    mx = np_ > mx ? np_ : mx;
  }
  double se = 0.0;
  for (int k = 0; k < K; ++k) se += exp(a.label_natparam[(long)t * K + k] - mx);
  const double lse = mx + log(se);
  const double inv = 1.0 / se;
  double lab = 0.0, lin = 0.0;
  int best = 0;
  double bestv = -1.0;
  for (int k = 0; k < K; ++k) {
    const double np_ = a.label_natparam[(long)t * K + k];
    const double l = np_ - a.label_global[k];
    const double rnew = exp(np_ - mx) * inv;
    const double rold = rin[k];
    lab = __builtin_fma(rnew, l, lab);
    lin = __builtin_fma(rold - rnew, l, lin);     // <eta - sum_k rnew_k G_k - node, stats>, gmm.py:99-102
    if (rnew > bestv) { bestv = rnew; best = k; }
    a.label_stats[(long)t * K + k] = rnew;
  }
  klt += lab - lse;
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

template <int N>
__global__ __launch_bounds__(gmm_block<N>()) void gmm_meanfield_kernel(const GmmArgs a) {
  constexpr int D = N + 2;
  constexpr int GMM_BLOCK = gmm_block<N>();
  __shared__ double red[GMM_BLOCK];
  const int tid = threadIdx.x;
  const int T = a.T, K = a.K;

  // one sweep over this thread's points; returns the thread's KL partial
  auto sweep = [&](bool first, bool final_pass) -> double {
    double klpart = 0.0;
    for (int t = tid; t < T; t += GMM_BLOCK)
      klpart += gmm_point<N>(a, t, first ? a.label_init + (long)t * K : a.label_stats + (long)t * K, final_pass);
    return klpart;
  };

  // ---- fixed point [gmm.py:90-110] -------------------------------------------------------------
  double kl_prev = 1.0 / 0.0;
  int it = 0;
  for (int i = 0; i < a.max_iter; ++i) {
    it = i + 1;
    const double kl = block_sum<GMM_BLOCK>(sweep(i == 0, false), red);
    const bool stop = fabs(kl - kl_prev) < a.tol;
    kl_prev = kl;
    if (stop) break;
  }
  // label_stats now holds the fixed point (or label_init if max_iter == 0)
  if (a.max_iter == 0) {
    for (int j = tid; j < T * K; j += GMM_BLOCK) a.label_stats[j] = a.label_init[j];
    __syncthreads();
  }
  if (a.label_fixed) {   // the responsibilities the final pass starts from (gmm.py:71)
    for (int t = tid; t < T; t += GMM_BLOCK)
      for (int k = 0; k < K; ++k) a.label_fixed[(long)t * K + k] = a.label_stats[(long)t * K + k];
  }
  // ---- final pass + outputs [gmm.py:74-86] -----------------------------------------------------
  // (reads r from label_stats and overwrites it point by point, by the same thread)
  const double kl = block_sum<GMM_BLOCK>(sweep(false, true), red);
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

// sum of partials[0..count) in a fixed order (strided per-thread partial sums, then a tree): every thread
// of the block returns it
__device__ __forceinline__ double gmm_fixed_order_sum(const double* partials, int count, double* red) {
  double v = 0.0;
  for (int j = threadIdx.x; j < count; j += GMM_MW_BLOCK) v += partials[j];
  return block_sum<GMM_MW_BLOCK>(v, red);
}

template <int N>
__global__ __launch_bounds__(GMM_MW_BLOCK) void gmm_mw_sweep_kernel(const GmmMwArgs m) {
  const GmmArgs& a = m.g;
  __shared__ double red[GMM_MW_BLOCK];
  __shared__ int sh_flag;
  const int tid = threadIdx.x, K = a.K, T = a.T;
  const int conv = gmm_converged_at(m.kl_hist, m.sweep, a.tol);     // same answer in every thread / workgroup
  if (m.mode == 0 && conv >= 0) return;                             // fixed point already reached
  const bool final_pass = m.mode == 1;
  const bool from_init = final_pass ? (a.max_iter == 0) : (m.sweep == 0);
  double klpart = 0.0;
  for (int t = blockIdx.x * GMM_MW_BLOCK + tid; t < T; t += gridDim.x * GMM_MW_BLOCK) {
    const double* rin = (from_init ? a.label_init : a.label_stats) + (long)t * K;
    if (final_pass && a.label_fixed)
      for (int k = 0; k < K; ++k) a.label_fixed[(long)t * K + k] = rin[k];     // gmm.py:71
    if (final_pass && from_init)    // max_iter == 0: the "fixed point" is label_init; gmm_point reads rin, writes label_stats
      for (int k = 0; k < K; ++k) a.label_stats[(long)t * K + k] = rin[k];
    klpart += gmm_point<N>(a, t, rin, final_pass);
  }
  const double wgsum = block_sum<GMM_MW_BLOCK>(klpart, red);
  if (tid == 0) {
    m.partials[blockIdx.x] = wgsum;
    __threadfence();
    const int ticket = atomicAdd(&m.counters[final_pass ? a.max_iter + 1 : m.sweep], 1);
    sh_flag = (ticket == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!sh_flag) return;
  __threadfence();
  const double total = gmm_fixed_order_sum(m.partials, gridDim.x, red);
  if (tid == 0) {
    if (final_pass) {
      a.kl[0] = total;                                   // (this rank's points; the caller sums over ranks)
      a.iters[0] = conv >= 0 ? conv + 1 : a.max_iter;
    } else {
      m.kl_hist[m.sweep] = total;
    }
  }
}

// The same sweeps in ONE cooperative launch (single GPU): a grid barrier per sweep instead of a launch per sweep
// (16 .. 21 us each, the whole cost of a sweep below ~100 k points).  Every workgroup sums the per-workgroup KL
// partials in index order itself (all reach the same total and the same stopping decision); the partials are
// double-buffered by sweep parity so that one barrier per sweep suffices.
template <int N>
__global__ __launch_bounds__(GMM_MW_BLOCK) void gmm_mw_persistent_kernel(const GmmMwArgs m) {
  namespace cg = cooperative_groups;
  cg::grid_group grid = cg::this_grid();
  const GmmArgs& a = m.g;
  __shared__ double red[GMM_MW_BLOCK];
  const int tid = threadIdx.x, K = a.K, T = a.T, G = gridDim.x;
  double prev = 1.0 / 0.0;
  int iters = 0;
  for (int i = 0; i < a.max_iter; ++i) {
    iters = i + 1;
    double klpart = 0.0;
    for (int t = blockIdx.x * GMM_MW_BLOCK + tid; t < T; t += G * GMM_MW_BLOCK)
      klpart += gmm_point<N>(a, t, (i == 0 ? a.label_init : a.label_stats) + (long)t * K, false);
    const double wgsum = block_sum<GMM_MW_BLOCK>(klpart, red);
    double* part = m.partials + (i & 1) * G;
    if (tid == 0) part[blockIdx.x] = wgsum;
    __threadfence();
    grid.sync();
    const double total = gmm_fixed_order_sum(part, G, red);
    if (blockIdx.x == 0 && tid == 0) m.kl_hist[i] = total;
    const bool stop = fabs(total - prev) < a.tol;
    prev = total;
    if (stop) break;
  }
  // final pass (gmm.py:74-86)
  const bool from_init = a.max_iter == 0;
  double klpart = 0.0;
  for (int t = blockIdx.x * GMM_MW_BLOCK + tid; t < T; t += G * GMM_MW_BLOCK) {
    const double* rin = (from_init ? a.label_init : a.label_stats) + (long)t * K;
    if (a.label_fixed)
      for (int k = 0; k < K; ++k) a.label_fixed[(long)t * K + k] = rin[k];
    if (from_init)
      for (int k = 0; k < K; ++k) a.label_stats[(long)t * K + k] = rin[k];
    klpart += gmm_point<N>(a, t, rin, true);
  }
  const double wgsum = block_sum<GMM_MW_BLOCK>(klpart, red);
  double* part = m.partials + 2 * G;
  if (tid == 0) part[blockIdx.x] = wgsum;
  __threadfence();
  grid.sync();
  if (blockIdx.x == 0) {
    const double total = gmm_fixed_order_sum(part, G, red);
    if (tid == 0) { a.kl[0] = total; a.iters[0] = iters; }
  }
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
  for (int j = threadIdx.x; j < NO; j += GMM_MW_BLOCK) {
    double s = 0.0;
    if (j < K) {
      for (int t = t0; t < t1; ++t) s += a.label_stats[(long)t * K + j];
    } else {
      const int k = (j - K) / (D * D), e = (j - K) % (D * D);
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
    for (int w = 0; w < (int)gridDim.x; ++w) s += spart[(long)w * NO + j];
    if (j < K) a.dirichlet_stats[j] = s; else a.niw_stats[j - K] = s;
  }
}

template <int N>
static int launch_gmm(const GmmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((gmm_meanfield_kernel<N>), dim3(1), dim3(gmm_block<N>()), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
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
  switch (N) {
    case 1: return svae::launch_gmm<1>(a, s);
    case 2: return svae::launch_gmm<2>(a, s);
    case 3: return svae::launch_gmm<3>(a, s);
    case 4: return svae::launch_gmm<4>(a, s);
    case 5: return svae::launch_gmm<5>(a, s);
    case 6: return svae::launch_gmm<6>(a, s);
    case 7: return svae::launch_gmm<7>(a, s);
    case 8: return svae::launch_gmm<8>(a, s);
  }
  return -2;
}

// ---- multi-workgroup entry points (see the block comment above gmm_mw_sweep_kernel) ---------------------
namespace svae {
static int gmm_mw_grid(int T) {
  const int g = (T + GMM_MW_BLOCK - 1) / GMM_MW_BLOCK;
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}
static int gmm_mw_stats_grid(int T) {
  const int g = (T + 255) / 256;
  return g < 1 ? 1 : (g > 256 ? 256 : g);
}
struct GmmMwLayout { double* kl_hist; double* partials; double* spart; int32_t* counters; };
static size_t gmm_mw_doubles(int T, int N, int K, int max_iter) {
  const int D = N + 2;
  return (size_t)(max_iter + 1) + 3 * gmm_mw_grid(T) + (size_t)gmm_mw_stats_grid(T) * K * (1 + D * D);
}
static GmmMwLayout gmm_mw_layout(void* ws, int T, int N, int K, int max_iter) {
  const int D = N + 2;
  GmmMwLayout l;
  l.kl_hist = (double*)ws;
  l.partials = l.kl_hist + (max_iter + 1);
  l.spart = l.partials + 3 * gmm_mw_grid(T);
  l.counters = (int32_t*)(l.spart + (size_t)gmm_mw_stats_grid(T) * K * (1 + D * D));
  return l;
}
template <int N>
static int launch_gmm_mw(const GmmMwArgs& m, hipStream_t s) {
  hipLaunchKernelGGL((gmm_mw_sweep_kernel<N>), dim3(gmm_mw_grid(m.g.T)), dim3(GMM_MW_BLOCK), 0, s, m);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
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
  switch (N) {
    case 1: return svae::launch_gmm_mw<1>(m, s);
    case 2: return svae::launch_gmm_mw<2>(m, s);
    case 3: return svae::launch_gmm_mw<3>(m, s);
    case 4: return svae::launch_gmm_mw<4>(m, s);
    case 5: return svae::launch_gmm_mw<5>(m, s);
    case 6: return svae::launch_gmm_mw<6>(m, s);
    case 7: return svae::launch_gmm_mw<7>(m, s);
    case 8: return svae::launch_gmm_mw<8>(m, s);
  }
  return -4;
}

/* device address of kl_hist[0] inside the workspace (the scalar a multi-GPU caller all-reduces per sweep) */
extern "C" double* svae_gmm_mw_kl_hist(void* workspace) { return (double*)workspace; }

// The whole fixed point + final pass in ONE cooperative launch (grid barrier per sweep), then the statistics
// launch: the single-GPU form of the svae_gmm_mw_* sweeps (same per-point code, same fixed-order KL reduction, same
// results).  Returns -50 if the device cannot co-schedule the grid (fall back to svae_gmm_mw_step_f64).
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
  const void* kern = nullptr;
  switch (N) {
    case 1: kern = (const void*)svae::gmm_mw_persistent_kernel<1>; break;
    case 2: kern = (const void*)svae::gmm_mw_persistent_kernel<2>; break;
    case 3: kern = (const void*)svae::gmm_mw_persistent_kernel<3>; break;
    case 4: kern = (const void*)svae::gmm_mw_persistent_kernel<4>; break;
    case 5: kern = (const void*)svae::gmm_mw_persistent_kernel<5>; break;
    case 6: kern = (const void*)svae::gmm_mw_persistent_kernel<6>; break;
    case 7: kern = (const void*)svae::gmm_mw_persistent_kernel<7>; break;
    case 8: kern = (const void*)svae::gmm_mw_persistent_kernel<8>; break;
  }
  int per_cu = 0, dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, svae::GMM_MW_BLOCK, 0) != hipSuccess) {
    (void)hipGetLastError();      // -50 = "use the per-sweep launches instead": leave no sticky error behind for them
    return -50;
  }
  int G = svae::gmm_mw_grid(T);
  const int cap = per_cu * cus;
  if (cap < 1) return -50;
  if (G > cap) G = cap;
  // the counters section doubles as the statistics kernel's ticket: zero it (kl_hist needs no reset here)
  if (hipMemsetAsync(l.counters, 0, (size_t)(max_iter + 3) * sizeof(int32_t), s) != hipSuccess) return -1000;
  void* params[] = {(void*)&m};
  if (hipLaunchCooperativeKernel(kern, dim3(G), dim3(svae::GMM_MW_BLOCK), params, 0, s) != hipSuccess) {
    (void)hipGetLastError();
    return -50;
  }
  hipLaunchKernelGGL(svae::gmm_mw_stats_kernel, dim3(svae::gmm_mw_stats_grid(T)), dim3(svae::GMM_MW_BLOCK), 0, s,
                     a, N + 2, l.spart, l.counters + max_iter + 2);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
