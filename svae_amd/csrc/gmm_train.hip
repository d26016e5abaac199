// gmm_train.hip -- the differentiable tail of the GMM-SVAE local step for MI355X (gfx950, fp64): sampling from the
// per-point Gaussian factors and the reverse-mode pass of (samples, local KL) w.r.t. the recognition potentials.
//
// What it replaces (reference = mattjj/svae, /root/reference):
//   svae/distributions/gaussian.py:27-33   natural_sample:  x = J^-1 h + chol(J)^-T eps
//   svae/models/gmm.py:74-86               the ONE pass after the fixed point that stays on the autograd tape
//                                          (gaussian_meanfield + label_meanfield on the boxed node potentials), whose
//                                          value -- local_kl -- the fixed-point kernels already return (gmm_meanfield.hip)
//   autograd's reverse pass through both (svae/svae.py:21-30: vgrad of mc_elbo w.r.t. the recognition network)
//
// One point per lane; everything of a point (N <= 8) lives in registers.  The adjoint is derived, not traced:
// with r (the fixed point, a constant of this pass), eta_t = node_t + sum_k r_tk G_k, s_t = E[t(x)] under eta_t,
// l_tk = <s_t, G_k>, r'_t = softmax(l_t + E log pi):
//   d local_kl_t = < node_t + sum_k w_k G_k , d s_t >,   w_k = r'_k (l_k - sum_j r'_j l_j)
// (the terms through eta's own differential cancel: d logZ(eta) = <s, d eta>; d logsumexp = <r', d l>), and
//   s = (Sigma + mu mu', mu), Sigma = J^-1, mu = Sigma h, J = -2 (diag(node_J) + ..):
//   mu_bar = g + (G + G') mu + sum_s x_bar_s,  h_bar = Sigma mu_bar,  J_bar_ii = -[(Sigma G Sigma)_ii + h_bar_i mu_i]
// plus the Cholesky path of the noise, n_s = L^-T eps_s: L_bar = -tril(sum_s n_s (L^-1 x_bar_s)'),
// J_bar += L^-T Phi(L' L_bar) L^-1 (diagonal only: node_J is diagonal).  g_node_J = -2 diag(J_bar), g_node_h = h_bar.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae_hip.h"

namespace svae {

template <int N>
struct GmmFactor {          // J = -2 A = Lt D Lt' (unit lower Lt), Li = Lt^-1, Sigma = J^-1, mu = Sigma h
  double Li[N][N];          // strict lower part used, unit diagonal implied
  double dinv[N], dis[N];   // 1 / d_j, 1 / sqrt(d_j)
  double Sig[N][N], mu[N], h[N];
};

template <int N>
__device__ __forceinline__ void gmm_factor(const double* gn /* (N+2) x (N+2) dense-packed natparam */, GmmFactor<N>& f) {
  constexpr int D = N + 2;
  double L[N][N], d[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double v[N];
    double dj = -2.0 * gn[j * D + j];
#pragma unroll
    for (int m = 0; m < j; ++m) {
      v[m] = L[j][m] * d[m];
      dj = __builtin_fma(-L[j][m], v[m], dj);
    }
    d[j] = dj;
    f.dinv[j] = 1.0 / dj;
    f.dis[j] = 1.0 / sqrt(dj);
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = -2.0 * gn[i * D + j];
#pragma unroll
      for (int m = 0; m < j; ++m) s = __builtin_fma(-L[i][m], v[m], s);
      L[i][j] = s * f.dinv[j];
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    f.Li[j][j] = 1.0;
#pragma unroll
    for (int i = 0; i < j; ++i) f.Li[i][j] = 0.0;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double s = -L[i][j];
#pragma unroll
      for (int m = j + 1; m < i; ++m) s = __builtin_fma(-L[i][m], f.Li[m][j], s);
      f.Li[i][j] = s;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) f.h[i] = gn[i * D + N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int m = i; m < N; ++m) s = __builtin_fma(f.Li[m][i] * f.dinv[m], f.Li[m][j], s);
      f.Sig[i][j] = s;
      f.Sig[j][i] = s;
    }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) s = __builtin_fma(f.Sig[i][j], f.h[j], s);
    f.mu[i] = s;
  }
}

// n = chol(J)^-T e = Lt^-T D^-1/2 e
template <int N>
__device__ __forceinline__ void gmm_noise(const GmmFactor<N>& f, const double (&e)[N], double (&n)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int m = i; m < N; ++m) s = __builtin_fma(f.Li[m][i], f.dis[m] * e[m], s);
    n[i] = s;
  }
}

template <int N>
__global__ __launch_bounds__(256) void gmm_sample_kernel(int T, int S, const double* __restrict__ natparam,
                                                         const double* __restrict__ eps, double* __restrict__ samples) {
  constexpr int D = N + 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  GmmFactor<N> f;
  gmm_factor<N>(natparam + (long)t * D * D, f);
  for (int s = 0; s < S; ++s) {
    double e[N], n[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = eps[((long)t * S + s) * N + i];
    gmm_noise<N>(f, e, n);
#pragma unroll
    for (int i = 0; i < N; ++i) samples[((long)t * S + s) * N + i] = f.mu[i] + n[i];
  }
}

template <int N>
__global__ __launch_bounds__(256) void gmm_local_vjp_kernel(
    int T, int K, int S, const double* __restrict__ label_global, const double* __restrict__ gaussian_globals,
    const double* __restrict__ node_J, const double* __restrict__ node_h, const double* __restrict__ natparam,
    const double* __restrict__ label_natparam, const double* __restrict__ g_kl, const double* __restrict__ eps,
    const double* __restrict__ g_samples, double* __restrict__ g_node_J, double* __restrict__ g_node_h) {
  constexpr int D = N + 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  GmmFactor<N> f;
  gmm_factor<N>(natparam + (long)t * D * D, f);
  const double gk = g_kl ? g_kl[0] : 0.0;

  // ---- cotangent of the statistics: gk * (node + sum_k w_k G_k) --------------------------------------------------
  double GA[N][N], gb[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) GA[i][j] = 0.0;
    GA[i][i] = node_J[(long)t * N + i];
    gb[i] = node_h[(long)t * N + i];
  }
  if (gk != 0.0) {
    const double* np_ = label_natparam + (long)t * K;
    double mx = -1.0 / 0.0;
    for (int k = 0; k < K; ++k) mx = np_[k] > mx ? np_[k] : mx;
    double se = 0.0, lbar = 0.0;
    for (int k = 0; k < K; ++k) {
      const double e = exp(np_[k] - mx);
      se += e;
      lbar = __builtin_fma(e, np_[k] - label_global[k], lbar);
    }
    const double inv = 1.0 / se;
    lbar *= inv;                                            // sum_j r'_j l_j
    for (int k = 0; k < K; ++k) {
      const double l = np_[k] - label_global[k];
      const double w = exp(np_[k] - mx) * inv * (l - lbar);
      const double* G = gaussian_globals + (long)k * D * D;
#pragma unroll
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) GA[i][j] = __builtin_fma(w, G[i * D + j], GA[i][j]);
        gb[i] = __builtin_fma(w, G[i * D + N], gb[i]);
      }
    }
  }
  // mu_bar = gk (gb + (GA + GA') mu) + sum_s x_bar_s
  double mub[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = gb[i];
#pragma unroll
    for (int j = 0; j < N; ++j) s = __builtin_fma(GA[i][j] + GA[j][i], f.mu[j], s);
    mub[i] = gk * s;
  }
  // ---- Cholesky path of the noise: Lbar = -tril(sum_s n_s z_s'), z_s = L^-1 x_bar_s ---------------------------------
  double Lb[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) Lb[i][j] = 0.0;
  if (g_samples) {
    for (int s = 0; s < S; ++s) {
      double e[N], xb[N], n[N], z[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        e[i] = eps[((long)t * S + s) * N + i];
        xb[i] = g_samples[((long)t * S + s) * N + i];
        mub[i] += xb[i];
      }
      gmm_noise<N>(f, e, n);
#pragma unroll
      for (int i = 0; i < N; ++i) {                         // z = D^-1/2 Lt^-1 x_bar
        double a = 0.0;
#pragma unroll
        for (int m = 0; m <= i; ++m) a = __builtin_fma(f.Li[i][m], xb[m], a);
        z[i] = f.dis[i] * a;
      }
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Lb[i][j] = __builtin_fma(-n[i], z[j], Lb[i][j]);
    }
  }
  // h_bar = Sigma mu_bar
  double hb[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) s = __builtin_fma(f.Sig[i][j], mub[j], s);
    hb[i] = s;
  }
  // diag(Sigma (gk GA) Sigma)
  double Jb[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < N; ++b) s = __builtin_fma(GA[a][b], f.Sig[b][i], s);
      acc = __builtin_fma(f.Sig[i][a], s, acc);
    }
    Jb[i] = -(gk * acc + hb[i] * f.mu[i]);
  }
  if (g_samples) {
    // L = Lt D^1/2 (Lt = Li^-1): M = L' Lbar; Phi(M) = tril(M), diagonal halved; P = L^-T Phi(M) L^-1; J_bar_ii += P_ii.
    // L^-1 = D^-1/2 Li.  L itself from Li: Lt = Li^-1 (unit lower), by forward substitution.
    double Lt[N][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int i = 0; i < N; ++i) Lt[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int i = j + 1; i < N; ++i) {
        double s = -f.Li[i][j];
#pragma unroll
        for (int m = j + 1; m < i; ++m) s = __builtin_fma(-f.Li[i][m], Lt[m][j], s);
        Lt[i][j] = s;
      }
    }
    double Ph[N][N];        // Phi(L' Lbar), lower
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        double s = 0.0;     // (L' Lbar)_ij = sum_{m >= i} L_mi Lbar_mj,  L_mi = Lt_mi sqrt(d_i) = Lt_mi / dis_i
#pragma unroll
        for (int m = i; m < N; ++m) s = __builtin_fma(Lt[m][i], Lb[m][j], s);
        s = s / f.dis[i];
        Ph[i][j] = (i == j) ? 0.5 * s : s;
      }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      // P_ii = sum_{m >= i} (L^-1)_mi Q_mi,  Q = Ph L^-1,  (L^-1)_ab = dis_a Li_ab
      double acc = 0.0;
#pragma unroll
      for (int m = i; m < N; ++m) {
        double q = 0.0;     // Q_mi = sum_{i <= a <= m} Ph_ma (L^-1)_ai
#pragma unroll
        for (int a = i; a <= m; ++a) q = __builtin_fma(Ph[m][a], f.dis[a] * f.Li[a][i], q);
        acc = __builtin_fma(f.dis[m] * f.Li[m][i], q, acc);
      }
      Jb[i] += acc;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    g_node_J[(long)t * N + i] = -2.0 * Jb[i];
    g_node_h[(long)t * N + i] = hb[i];
  }
}


// ---- the global side of a GMM step in ONE launch -------------------------------------------------------------------
// dirichlet.expectedstats (dirichlet.py:5-7) and niw.expectedstats of the K components (niw.py:15-25, fudge 1e-8)
// -> the potentials the fixed-point kernels take, and the prior KL of gmm.py:54-58 (with dirichlet.logZ dirichlet.py:9-11,
// niw.logZ niw.py:27-31).  One lane per component (K <= 64, N <= 8: a component's matrices live in registers); the sums
// over components are wavefront butterflies.  In torch the same maps are ~60 small launches (1.4 ms of a 2.2 ms step).
__device__ inline double gmm_digamma_pos(double x) {
  double acc = 0.0;
  while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
  const double r = 1.0 / x, r2 = r * r;
  const double s = -1.0 / 12.0 + r2 * (1.0 / 120.0 + r2 * (-1.0 / 252.0 + r2 * (1.0 / 240.0 + r2 * (-1.0 / 132.0
                   + r2 * (691.0 / 32760.0 + r2 * (-1.0 / 12.0))))));
  return acc + log(x) - 0.5 * r + r2 * s;
}

__device__ __forceinline__ double gmm_wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// NIW natural parameter (dense (N+2)x(N+2)) -> standard (S, m, kappa, nu); S^-1 and log|S| by Gauss-Jordan without
// pivoting (S is SPD for a valid parameter; a non-positive pivot clears *ok)
template <int N>
struct NiwStd { double Sinv[N][N], m[N], kappa, nu, logdet; bool ok; };

template <int N>
__device__ __forceinline__ void niw_standard(const double* nat, NiwStd<N>& q, bool want_inverse) {
  constexpr int D = N + 2;
  q.kappa = nat[N * D + N];
  q.nu = nat[(N + 1) * D + N + 1];
  double S[N][N], b[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { b[i] = nat[i * D + N]; q.m[i] = b[i] / q.kappa; }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) S[i][j] = nat[i * D + j] - b[i] * q.m[j];
  // in-place Gauss-Jordan: S -> S^-1, pivots -> log det
  q.ok = true;
  double ld = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double piv = S[k][k];
    q.ok = q.ok && (piv > 0.0);
    ld += log(piv);
    if (want_inverse) {
      const double r = 1.0 / piv;
#pragma unroll
      for (int j = 0; j < N; ++j) S[k][j] = (j == k) ? r : S[k][j] * r;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (i == k) continue;
        const double f = S[i][k];
#pragma unroll
        for (int j = 0; j < N; ++j) S[i][j] = (j == k) ? -f * r : __builtin_fma(-f, S[k][j], S[i][j]);
      }
    } else {
      const double r = 1.0 / piv;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const double f = S[i][k] * r;
#pragma unroll
        for (int j = k + 1; j < N; ++j) S[i][j] = __builtin_fma(-f, S[k][j], S[i][j]);
      }
    }
  }
  q.logdet = ld;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) q.Sinv[i][j] = S[i][j];
}

// niw.logZ of one component (niw.py:27-31): d nu/2 log 2 + multigammaln(nu/2, d) - nu/2 log|S| - d/2 log kappa
template <int N>
__device__ __forceinline__ double niw_logZ_one(const NiwStd<N>& q) {
  double mg = 0.25 * N * (N - 1) * 1.1447298858494001741;       // log(pi)
#pragma unroll
  for (int j = 0; j < N; ++j) mg += lgamma(0.5 * q.nu - 0.5 * j);
  return 0.5 * N * q.nu * 0.6931471805599453094 + mg - 0.5 * q.nu * q.logdet - 0.5 * N * log(q.kappa);
}

template <int N>
__global__ __launch_bounds__(64) void gmm_global_step_kernel(int K, const double* __restrict__ dir_nat,
                                                             const double* __restrict__ niw_nat,
                                                             const double* __restrict__ prior_dir,
                                                             const double* __restrict__ prior_niw,
                                                             double* __restrict__ label_global,
                                                             double* __restrict__ gaussian_globals,
                                                             double* __restrict__ kl, int32_t* __restrict__ info) {
  constexpr int D = N + 2;
  const int k = threadIdx.x;
  const bool on = k < K;
  const int kk = on ? k : 0;
  // ---- Dirichlet factor ----
  const double alpha = dir_nat[kk] + 1.0;
  const double asum = gmm_wave_sum(on ? alpha : 0.0);
  const double es_dir = gmm_digamma_pos(alpha) - gmm_digamma_pos(asum);
  if (on) label_global[k] = es_dir;
  // ---- NIW factors ----
  NiwStd<N> q;
  niw_standard<N>(niw_nat + (long)kk * D * D, q, true);
  double EJ[N][N], Eh[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) EJ[i][j] = q.nu * (0.5 * (q.Sinv[i][j] + q.Sinv[j][i])) + ((i == j) ? 1e-8 : 0.0);
  double mEh = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double sacc = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) sacc = __builtin_fma(EJ[i][j], q.m[j], sacc);
    Eh[i] = sacc;
    mEh = __builtin_fma(q.m[i], sacc, mEh);
  }
  const double E_hJh = (double)N / q.kappa + mEh;
  double dg = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) dg += gmm_digamma_pos(0.5 * (q.nu - i));
  const double E_logdet = dg + N * 0.6931471805599453094 - q.logdet;
  if (on) {
    double* G = gaussian_globals + (long)k * D * D;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double v = 0.0;
        if (i < N && j < N) v = -0.5 * EJ[i < N ? i : 0][j < N ? j : 0];
        else if (i < N && j == N) v = Eh[i < N ? i : 0];
        else if (i == N && j == N) v = -0.5 * E_hJh;
        else if (i == N + 1 && j == N + 1) v = 0.5 * E_logdet;
        G[i * D + j] = v;
      }
  }
  const bool bad = __ballot(on && !q.ok) != 0;
  if (bad && k == 0) atomicMax(info, 1);
  if (!kl) return;
  // ---- prior KL (gmm.py:54-58): <eta_q - eta_p, E_q t> - (logZ(q) - logZ(p)) ----
  NiwStd<N> pq;
  niw_standard<N>(prior_niw + (long)kk * D * D, pq, false);
  double contr = (dir_nat[kk] - prior_dir[kk]) * es_dir;
  {
    const double* g = niw_nat + (long)kk * D * D;
    const double* p = prior_niw + (long)kk * D * D;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) contr = __builtin_fma(g[i * D + j] - p[i * D + j], -0.5 * EJ[i][j], contr);
      contr = __builtin_fma(g[i * D + N] - p[i * D + N], Eh[i], contr);
    }
    contr = __builtin_fma(g[N * D + N] - p[N * D + N], -0.5 * E_hJh, contr);
    contr = __builtin_fma(g[(N + 1) * D + N + 1] - p[(N + 1) * D + N + 1], 0.5 * E_logdet, contr);
  }
  const double palpha = prior_dir[kk] + 1.0;
  const double psum = gmm_wave_sum(on ? palpha : 0.0);
  const double lz = (lgamma(alpha) - lgamma(palpha)) + (niw_logZ_one<N>(q) - niw_logZ_one<N>(pq));
  const double mlz = gmm_wave_sum(on ? -lz : 0.0) + (lgamma(asum) - lgamma(psum));   // -(logZ(q) - logZ(p))
  const double total = gmm_wave_sum(on ? contr : 0.0) + mlz;
  if (k == 0) {
    kl[0] = total;
    // kl[1]: the value the reference AS SHIPPED returns -- gmm.py:55-56 flattens with util.py:39 `flat`, which
    // (util.py:166 rebinds `flatten`) keeps the FIRST scalar of the nested structure: the contraction is its first term
    kl[1] = (dir_nat[0] - prior_dir[0]) * es_dir + mlz;
  }
  const bool pbad = __ballot(on && !pq.ok) != 0;
  if (pbad && k == 0) atomicMax(info, 1);
}

template <typename Fn>
static int gmm_train_dispatch(int N, Fn&& fn) {
  switch (N) {
    case 1: return fn(std::integral_constant<int, 1>{});
    case 2: return fn(std::integral_constant<int, 2>{});
    case 3: return fn(std::integral_constant<int, 3>{});
    case 4: return fn(std::integral_constant<int, 4>{});
    case 5: return fn(std::integral_constant<int, 5>{});
    case 6: return fn(std::integral_constant<int, 6>{});
    case 7: return fn(std::integral_constant<int, 7>{});
    case 8: return fn(std::integral_constant<int, 8>{});
  }
  return -2;
}

}  // namespace svae

extern "C" int svae_gmm_sample_f64(int T, int N, int S, const double* gaussian_natparam, const double* eps,
                                   double* samples, void* stream) {
  if (T < 0) return -1;
  if (N < 1 || N > 8) return -2;
  if (S < 0) return -3;
  if (T > 0 && !gaussian_natparam) return -4;
  if (T > 0 && S > 0 && (!eps || !samples)) return -5;
  if (T == 0 || S == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  return svae::gmm_train_dispatch(N, [&](auto n) -> int {
    hipLaunchKernelGGL((svae::gmm_sample_kernel<decltype(n)::value>), dim3((T + 255) / 256), dim3(256), 0, s, T, S,
                       gaussian_natparam, eps, samples);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  });
}

extern "C" int svae_gmm_local_vjp_f64(int T, int N, int K, int S, const double* label_global,
                                      const double* gaussian_globals, const double* node_J, const double* node_h,
                                      const double* gaussian_natparam, const double* label_natparam,
                                      const double* g_kl, const double* eps, const double* g_samples,
                                      double* g_node_J, double* g_node_h, void* stream) {
  if (T < 0) return -1;
  if (N < 1 || N > 8) return -2;
  if (K < 1 || K > 64) return -3;
  if (S < 0) return -4;
  if (!label_global) return -5;
  if (!gaussian_globals) return -6;
  if (T > 0 && (!node_J || !node_h)) return -7;
  if (T > 0 && (!gaussian_natparam || !label_natparam)) return -9;
  if (g_samples && S > 0 && !eps) return -12;
  if (T > 0 && (!g_node_J || !g_node_h)) return -14;
  if (T == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const double* gs = (S > 0) ? g_samples : nullptr;
  return svae::gmm_train_dispatch(N, [&](auto n) -> int {
    hipLaunchKernelGGL((svae::gmm_local_vjp_kernel<decltype(n)::value>), dim3((T + 255) / 256), dim3(256), 0, s, T, K, S,
                       label_global, gaussian_globals, node_J, node_h, gaussian_natparam, label_natparam, g_kl, eps, gs,
                       g_node_J, g_node_h);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  });
}

extern "C" int svae_gmm_global_step_f64(int K, int N, const double* dirichlet_natparam, const double* niw_natparam,
                                        const double* prior_dirichlet, const double* prior_niw,
                                        double* label_global, double* gaussian_globals, double* kl, int32_t* info,
                                        void* stream) {
  if (K < 1 || K > 64) return -1;
  if (N < 1 || N > 8) return -2;
  if (!dirichlet_natparam) return -3;
  if (!niw_natparam) return -4;
  if (kl && (!prior_dirichlet || !prior_niw)) return -5;
  if (!label_global) return -7;
  if (!gaussian_globals) return -8;
  if (!info) return -10;
  hipStream_t s = (hipStream_t)stream;
  return svae::gmm_train_dispatch(N, [&](auto n) -> int {
    hipLaunchKernelGGL((svae::gmm_global_step_kernel<decltype(n)::value>), dim3(1), dim3(64), 0, s, K, dirichlet_natparam,
                       niw_natparam, prior_dirichlet, prior_niw, label_global, gaussian_globals, kl, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
  });
}
