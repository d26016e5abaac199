// lds_global.hip -- the once-per-step GLOBAL side of the LDS-SVAE on the device:
//   * global -> local maps: niw.expectedstats      /root/reference/svae/distributions/niw.py:15-25
//                           mniw.expectedstats     /root/reference/svae/distributions/mniw.py:19-20, 33-55
//     (the LDS init potential is the unpacked NIW expected statistic, the pair potential the MNIW one:
//      svae/models/lds.py:23-25), and the prior KL of svae/models/lds.py:16-20 with niw.logZ (niw.py:27-31) and
//     mniw.logZ (mniw.py:13-17) -- ONE launch of one workgroup instead of ~150 tiny library launches
//     (two LU inversions, three slogdets, digamma / multigammaln chains per factor);
//   * the natural-gradient expression of svae/svae.py:33-34 on the packed statistics buffer the E-step's batch
//     reduction (svae_lds_reduce_stats_f64, after the all-reduce) leaves: one launch.
// n <= 64; everything float64.  Inversions are in-LDS Gauss-Jordan eliminations without pivoting (the matrices
// are SPD whenever the natural parameters are valid; a non-positive pivot is reported through `info`).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/svae_hip.h"
#include "per_device.hpp"

namespace svae {

constexpr int GL_BLOCK = 256;
constexpr int GL_MAX_N = 64;
constexpr double GL_FUDGE = 1e-8;      // niw.py:18, mniw.py:43-44

// psi(x), x > 0: recurrence up to x >= 10, then the asymptotic series (error < 1e-16 there)
__device__ inline double digamma_pos(double x) {
  double acc = 0.0;
  while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
  const double r = 1.0 / x, r2 = r * r;
  double s = -1.0 / 12.0 + r2 * (1.0 / 120.0 + r2 * (-1.0 / 252.0 + r2 * (1.0 / 240.0 + r2 * (-1.0 / 132.0
             + r2 * (691.0 / 32760.0 + r2 * (-1.0 / 12.0))))));
  return acc + log(x) - 0.5 * r + r2 * s;
}

// sum_{i<n} psi((nu - i)/2)  and  multigammaln(nu/2, n)  (scipy.special.multigammaln): every thread returns both
__device__ inline void wishart_terms(double nu, int n, double* red, double& psi_sum, double& mgl) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    a += digamma_pos(0.5 * (nu - i));
    b += lgamma(0.5 * (nu - i));
  }
  red[threadIdx.x] = a;
  red[GL_BLOCK + threadIdx.x] = b;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) { red[threadIdx.x] += red[threadIdx.x + s]; red[GL_BLOCK + threadIdx.x] += red[GL_BLOCK + threadIdx.x + s]; }
    __syncthreads();
  }
  psi_sum = red[0];
  mgl = red[GL_BLOCK] + 0.25 * n * (n - 1) * 1.1447298858494001741;   // log(pi)
  __syncthreads();
}

// In place: M (n x n, row stride ld, SPD) -> M^-1; returns log det M (every thread); *bad set on a non-positive pivot.
// Gauss-Jordan: per pivot k the scaled pivot row and the multiplier column are staged, then every thread updates
// its elements.
__device__ inline double spd_inverse(double* M, int n, int ld, double* rowk, double* colk, int* bad, double* alt) {
  double logdet = 0.0;
  if (n * n <= GL_BLOCK && ld == n) {
    // small matrices (n <= 16): thread e owns element (i, j) in a register for the whole elimination; per pivot it
    // publishes the element into one of two LDS copies (ping-pong: ONE barrier per pivot) and reads M[i][k], M[k][j]
    // and the pivot back.  No index arithmetic inside the loop.
    const int e = threadIdx.x, i = e / n, j = e - i * n;
    const bool own = e < n * n;
    double v = own ? M[e] : 0.0;
    __syncthreads();                       // (M is one of the two copies: everyone has read its element)
    for (int k = 0; k < n; ++k) {
      double* cur = (k & 1) ? alt : M;
      if (own) cur[e] = v;
      __syncthreads();
      const double p = cur[k * n + k];
      if (!(p > 0.0) && threadIdx.x == 0) *bad = 1;
      logdet += log(p);
      if (own) {
        const double pinv = 1.0 / p;
        const double mik = cur[i * n + k], mkj = cur[k * n + j];
        if (i == k) v = (j == k) ? pinv : mkj * pinv;
        else if (j == k) v = -mik * pinv;
        else v = __builtin_fma(-mik, mkj * pinv, v);
      }
    }
    __syncthreads();                       // the last copy has been read
    if (own) M[e] = v;
    __syncthreads();
    return logdet;
  }
  for (int k = 0; k < n; ++k) {
    const double p = M[k * ld + k];
    __syncthreads();
    if (!(p > 0.0) && threadIdx.x == 0) *bad = 1;
    logdet += log(p);
    const double pinv = 1.0 / p;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      rowk[j] = (j == k) ? pinv : M[k * ld + j] * pinv;
      colk[j] = (j == k) ? 0.0 : M[j * ld + k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e % n;
      double v;
      if (i == k) v = rowk[j];
      else if (j == k) v = -colk[i] * pinv;
      else v = M[i * ld + j] - colk[i] * rowk[j];
      M[i * ld + j] = v;
    }
    __syncthreads();
  }
  return logdet;
}

__device__ inline void symmetrize(double* M, int n, int ld) {
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e % n;
    if (i < j) { const double v = 0.5 * (M[i * ld + j] + M[j * ld + i]); M[i * ld + j] = v; M[j * ld + i] = v; }
  }
  __syncthreads();
}

struct GlobalArgs {
  int n;
  const double* niw;        // (n+2, n+2) dense-packed natural parameter
  const double* mA; const double* mB; const double* mC; const double* md;   // MNIW natural parameter (n,n) x3, (1)
  const double* p_niw;      // prior (same shapes) or nullptr: no KL
  const double* p_mA; const double* p_mB; const double* p_mC; const double* p_md;
  double* init_J; double* init_h; double* init_logZ;      // LDS init potential (-1/2 E[J], E[h], sum of the two constants)
  double* J11; double* J12; double* J22; double* logZ_pair;
  double* niw_es;           // (n+2, n+2) expected statistics, dense-packed, or nullptr
  double* global_kl;        // (1) or nullptr
  int32_t* info;
};

// One factor pair (NIW, MNIW): expected statistics (optional outputs) and log-normalisers.
// Work arrays in LDS: W0, W1, W2 (n x n each), vectors.
struct FactorOut { double niw_logZ, mniw_logZ; };

__device__ __forceinline__ void lds_global_body(const GlobalArgs& a, const int pass);

__global__ __launch_bounds__(GL_BLOCK) void lds_global_kernel(const GlobalArgs a) { lds_global_body(a, (int)blockIdx.x); }

// K parameter sets in ONE launch (the SLDS global -> local maps: K launches of 36 us each were a fifth of the start-up of
// an ascent): workgroup k runs set k.  No priors / KL here (pass 0 only).
constexpr int GL_MULTI_MAX = 16;
struct GlobalArgsMulti { GlobalArgs s[GL_MULTI_MAX]; };
__global__ __launch_bounds__(GL_BLOCK) void lds_global_multi_kernel(const GlobalArgsMulti m) {
  lds_global_body(m.s[blockIdx.x], 0);
}

__device__ __forceinline__ void lds_global_body(const GlobalArgs& a, const int pass) {
  extern __shared__ double sm[];
  const int n = a.n, D = n + 2, tid = threadIdx.x;
  double* W0 = sm;                  // n*n
  double* W1 = W0 + n * n;
  double* W2 = W1 + n * n;
  double* W3 = W2 + n * n;
  double* v0 = W3 + n * n;          // n
  double* v1 = v0 + n;
  double* rowk = v1 + n;
  double* colk = rowk + n;
  double* red = colk + n;           // 2 * GL_BLOCK
  double* alt = red + 2 * GL_BLOCK; // GL_BLOCK: second copy of a small matrix during its inversion
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();

  double kl = 0.0;                  // accumulated by thread 0 only where noted
  // two passes, one WORKGROUP each (they are independent): q = 0 the global factors (outputs written), q = 1 the
  // prior (log-normalisers only).  Each adds its share of the KL to *global_kl (zeroed by the launcher): two terms,
  // so the sum does not depend on the order of the two atomic additions.
  double es_dot = 0.0;              // <prior - global, E_global[t]>  partial of this thread
  double logZ_g = 0.0, logZ_p = 0.0;
  {
    const int q = pass;
    const double* niw = q ? a.p_niw : a.niw;
    const double* mA = q ? a.p_mA : a.mA;
    const double* mB = q ? a.p_mB : a.mB;
    const double* mC = q ? a.p_mC : a.mC;
    const double nu2 = q ? a.p_md[0] : a.md[0];
    // ---- NIW: S = A - b m', m = b / kappa -----------------------------------------------------------
    const double kappa = niw[n * D + n], nu = niw[(n + 1) * D + n + 1];
    for (int j = tid; j < n; j += blockDim.x) v0[j] = niw[j * D + n] / kappa;          // m
    __syncthreads();
    for (int e = tid; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e % n;
      W0[e] = niw[i * D + j] - niw[i * D + n] * v0[j];
    }
    __syncthreads();
    const double logdetS = spd_inverse(W0, n, n, rowk, colk, &bad, alt);
    symmetrize(W0, n, n);
    double psi, mgl;
    wishart_terms(nu, n, red, psi, mgl);
    const double niw_logZ = 0.5 * n * nu * 0.69314718055994530942 + mgl - 0.5 * nu * logdetS - 0.5 * n * log(kappa);
    if (q == 0) {
      // E_J = nu S^-1 + fudge I;  E_h = E_J m;  E_hTJinvh = n / kappa + m' E_h;  E_logdetJ = psi + n log 2 - logdet S
      for (int e = tid; e < n * n; e += blockDim.x) W0[e] = nu * W0[e] + ((e / n == e % n) ? GL_FUDGE : 0.0);
      __syncthreads();
      for (int i = tid; i < n; i += blockDim.x) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s = __builtin_fma(W0[i * n + j], v0[j], s);
        v1[i] = s;                                                                   // E_h
      }
      __syncthreads();
      double mEh = 0.0;
      for (int j = 0; j < n; ++j) mEh = __builtin_fma(v0[j], v1[j], mEh);
      const double es_c = -0.5 * ((double)n / kappa + mEh);                           // -1/2 E[h' J^-1 h]
      const double es_d = 0.5 * (psi + n * 0.69314718055994530942 - logdetS);         // 1/2 E[log |J|]
      for (int e = tid; e < n * n; e += blockDim.x) {
        const double val = -0.5 * W0[e];
        a.init_J[e] = val;
        if (a.niw_es) a.niw_es[(e / n) * D + (e % n)] = val;
        if (a.p_niw) es_dot = __builtin_fma(a.p_niw[(e / n) * D + (e % n)] - a.niw[(e / n) * D + (e % n)], val, es_dot);
      }
      for (int j = tid; j < n; j += blockDim.x) {
        a.init_h[j] = v1[j];
        if (a.niw_es) a.niw_es[j * D + n] = v1[j];
        if (a.p_niw) es_dot = __builtin_fma(a.p_niw[j * D + n] - a.niw[j * D + n], v1[j], es_dot);
      }
      if (tid == 0) {
        a.init_logZ[0] = es_c + es_d;
        if (a.niw_es) { a.niw_es[n * D + n] = es_c; a.niw_es[(n + 1) * D + n + 1] = es_d; }
        if (a.p_niw) es_dot += (a.p_niw[n * D + n] - kappa) * es_c + (a.p_niw[(n + 1) * D + n + 1] - nu) * es_d;
      }
      if (a.niw_es) {   // the structurally zero entries of the dense packing
        for (int e = tid; e < D * D; e += blockDim.x) {
          const int i = e / D, j = e % D;
          const bool used = (i < n && j <= n) || (i == j);
          if (!used) a.niw_es[e] = 0.0;
        }
      }
      __syncthreads();
    }
    // ---- MNIW (A, B, C, d): K = sym(A^-1), M = (K B)', S = C - M B --------------------------------------
    for (int e = tid; e < n * n; e += blockDim.x) W1[e] = mA[e];
    __syncthreads();
    const double logdetA = spd_inverse(W1, n, n, rowk, colk, &bad, alt);                   // W1 = K
    symmetrize(W1, n, n);
    for (int e = tid; e < n * n; e += blockDim.x) {                                     // W2 = M' = K B   (M[i][j] = W2[j][i])
      const int i = e / n, j = e % n;
      double s = 0.0;
      for (int k = 0; k < n; ++k) s = __builtin_fma(W1[i * n + k], mB[k * n + j], s);
      W2[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += blockDim.x) {                                     // W3 = S = C - M B
      const int i = e / n, j = e % n;
      double s = mC[e];
      for (int k = 0; k < n; ++k) s = __builtin_fma(-W2[k * n + i], mB[k * n + j], s);
      W3[e] = s;
    }
    __syncthreads();
    const double logdetS2 = spd_inverse(W3, n, n, rowk, colk, &bad, alt);                  // W3 = S^-1
    symmetrize(W3, n, n);
    double psi2, mgl2;
    wishart_terms(nu2, n, red, psi2, mgl2);
    const double mniw_logZ = 0.5 * n * nu2 * 0.69314718055994530942 + mgl2 - 0.5 * nu2 * logdetS2 - 0.5 * n * logdetA;
    if (q == 0) {
      logZ_g = niw_logZ + mniw_logZ;
      // SinvM = S^-1 M  -> W0 ;  E_Sigmainv_A = nu SinvM ;  E_AT_Sigmainv_A = n K + nu sym(M' SinvM) + fudge I
      for (int e = tid; e < n * n; e += blockDim.x) {
        const int i = e / n, j = e % n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s = __builtin_fma(W3[i * n + k], W2[j * n + k], s);   // M[k][j] = W2[j][k]
        W0[e] = s;
      }
      __syncthreads();
      for (int e = tid; e < n * n; e += blockDim.x) {
        const int i = e / n, j = e % n;
        double sij = 0.0, sji = 0.0;                                                   // (M' SinvM)[i][j] and [j][i]
        for (int k = 0; k < n; ++k) {
          sij = __builtin_fma(W2[i * n + k], W0[k * n + j], sij);
          sji = __builtin_fma(W2[j * n + k], W0[k * n + i], sji);
        }
        const double j11 = -0.5 * ((double)n * W1[e] + nu2 * 0.5 * (sij + sji) + ((i == j) ? GL_FUDGE : 0.0));
        const double j12 = nu2 * W0[j * n + i];                                        // (nu SinvM)'
        const double j22 = -0.5 * (nu2 * W3[e] + ((i == j) ? GL_FUDGE : 0.0));
        a.J11[e] = j11; a.J12[e] = j12; a.J22[e] = j22;
        if (a.p_niw) {
          es_dot = __builtin_fma(a.p_mA[e] - a.mA[e], j11, es_dot);
          es_dot = __builtin_fma(a.p_mB[e] - a.mB[e], j12, es_dot);
          es_dot = __builtin_fma(a.p_mC[e] - a.mC[e], j22, es_dot);
        }
      }
      const double lz = 0.5 * (psi2 + n * 0.69314718055994530942 - logdetS2);
      if (tid == 0) {
        a.logZ_pair[0] = lz;
        if (a.p_niw) es_dot += (a.p_md[0] - a.md[0]) * lz;
      }
      __syncthreads();
    } else {
      logZ_p = niw_logZ + mniw_logZ;
    }
  }
  if (a.global_kl && a.p_niw) {
    red[tid] = es_dot;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) atomicAdd(a.global_kl, pass == 0 ? -red[0] - logZ_g + kl : logZ_p);   // lds.py:16-20
  }
  if (tid == 0 && bad) atomicMax(a.info, 1);
}

// svae.py:33-34 on the packed statistics of svae_lds_reduce_stats_f64 ([sum E_init (n^2+n) | sum E_pair (3n^2) |
// . | count]):  natgrad = -scale * (prior + num_batches * stats - params)  over the flat global parameter
// [NIW dense (n+2)^2 | A | B | C | d], stats = (pack_dense(sum ExxT0, sum Ex0, count, count), (E_pair sums, count (T-1))).
__global__ __launch_bounds__(256) void lds_natgrad_kernel(int n, int T, const double* packed, const double* prior,
                                                          const double* params, double num_batches, double scale,
                                                          double* out) {
  const int D = n + 2, nn = n * n, tot = D * D + 3 * nn + 1;
  const double cnt = packed[4 * nn + n + 1];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
    double st;
    if (e < D * D) {
      const int i = e / D, j = e % D;
      if (i < n && j < n) st = packed[i * n + j];
      else if (i < n && j == n) st = packed[nn + i];
      else if (i == j) st = cnt;
      else st = 0.0;
    } else if (e < D * D + 3 * nn) {
      st = packed[nn + n + (e - D * D)];
    } else {
      st = cnt * (double)(T - 1);
    }
    out[e] = -scale * (prior[e] + num_batches * st - params[e]);
  }
}

}  // namespace svae

extern "C" int svae_lds_global_step_f64(int n, const double* niw, const double* mniw_A, const double* mniw_B,
                                        const double* mniw_C, const double* mniw_d,
                                        const double* prior_niw, const double* prior_A, const double* prior_B,
                                        const double* prior_C, const double* prior_d,
                                        double* init_J, double* init_h, double* init_logZ,
                                        double* J11, double* J12, double* J22, double* logZ_pair,
                                        double* niw_expectedstats, double* global_kl, int32_t* info, void* stream) {
  if (n < 1 || n > svae::GL_MAX_N) return -1;
  if (!niw) return -2;
  if (!mniw_A || !mniw_B || !mniw_C || !mniw_d) return -3;
  if (prior_niw && (!prior_A || !prior_B || !prior_C || !prior_d)) return -7;
  if (!init_J || !init_h || !init_logZ) return -12;
  if (!J11 || !J12 || !J22 || !logZ_pair) return -15;
  if (global_kl && !prior_niw) return -20;
  if (!info) return -21;
  svae::GlobalArgs a;
  a.n = n; a.niw = niw; a.mA = mniw_A; a.mB = mniw_B; a.mC = mniw_C; a.md = mniw_d;
  a.p_niw = prior_niw; a.p_mA = prior_A; a.p_mB = prior_B; a.p_mC = prior_C; a.p_md = prior_d;
  a.init_J = init_J; a.init_h = init_h; a.init_logZ = init_logZ;
  a.J11 = J11; a.J12 = J12; a.J22 = J22; a.logZ_pair = logZ_pair;
  a.niw_es = niw_expectedstats; a.global_kl = global_kl; a.info = info;
  const size_t lds = (size_t)(4 * n * n + 4 * n + 3 * svae::GL_BLOCK) * sizeof(double);
  static svae::LdsGrant grant;
  if (!grant.ensure(reinterpret_cast<const void*>(svae::lds_global_kernel),
                    (long)((4 * svae::GL_MAX_N * svae::GL_MAX_N + 4 * svae::GL_MAX_N + 3 * svae::GL_BLOCK) * sizeof(double))))
    return -1001;
  // (n = 10: 80 us with staged pivot rows and both passes in one workgroup; 62 us with register-resident elements
  //  and one barrier per pivot; one workgroup per pass: see DESIGN.md 4c.  It does not depend on the minibatch.)
  const bool with_kl = global_kl && prior_niw;
  if (with_kl && hipMemsetAsync(global_kl, 0, sizeof(double), (hipStream_t)stream) != hipSuccess) return -1000;
  hipLaunchKernelGGL(svae::lds_global_kernel, dim3(prior_niw ? 2 : 1), dim3(svae::GL_BLOCK), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

extern "C" int svae_lds_global_step_multi_f64(int K, int n, const double* const* niw, const double* const* mniw_A,
                                              const double* const* mniw_B, const double* const* mniw_C,
                                              const double* const* mniw_d,
                                              double* init_J, double* init_h, double* init_logZ,
                                              double* J11, double* J12, double* J22, double* logZ_pair,
                                              double* niw_expectedstats, int32_t* info, void* stream) {
  if (K < 1 || K > svae::GL_MULTI_MAX) return -1;
  if (n < 1 || n > svae::GL_MAX_N) return -2;
  if (!niw || !mniw_A || !mniw_B || !mniw_C || !mniw_d) return -3;
  if (!init_J || !init_h || !init_logZ) return -8;
  if (!J11 || !J12 || !J22 || !logZ_pair) return -11;
  if (!info) return -16;
  svae::GlobalArgsMulti m;
  const long nn = (long)n * n, D = n + 2;
  for (int k = 0; k < svae::GL_MULTI_MAX; ++k) {
    const int q = k < K ? k : K - 1;                 // (unused entries repeat the last set: never launched)
    if (!niw[q] || !mniw_A[q] || !mniw_B[q] || !mniw_C[q] || !mniw_d[q]) return -3;
    svae::GlobalArgs& a = m.s[k];
    a.n = n; a.niw = niw[q]; a.mA = mniw_A[q]; a.mB = mniw_B[q]; a.mC = mniw_C[q]; a.md = mniw_d[q];
    a.p_niw = nullptr; a.p_mA = a.p_mB = a.p_mC = a.p_md = nullptr;
    a.init_J = init_J + q * nn; a.init_h = init_h + q * n; a.init_logZ = init_logZ + q;
    a.J11 = J11 + q * nn; a.J12 = J12 + q * nn; a.J22 = J22 + q * nn; a.logZ_pair = logZ_pair + q;
    a.niw_es = niw_expectedstats ? niw_expectedstats + q * D * D : nullptr;
    a.global_kl = nullptr; a.info = info;
  }
  const size_t lds = (size_t)(4 * n * n + 4 * n + 3 * svae::GL_BLOCK) * sizeof(double);
  static svae::LdsGrant grant;
  if (!grant.ensure(reinterpret_cast<const void*>(svae::lds_global_multi_kernel),
                    (long)((4 * svae::GL_MAX_N * svae::GL_MAX_N + 4 * svae::GL_MAX_N + 3 * svae::GL_BLOCK) * sizeof(double))))
    return -1001;
  hipLaunchKernelGGL(svae::lds_global_multi_kernel, dim3(K), dim3(svae::GL_BLOCK), lds, (hipStream_t)stream, m);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

extern "C" int svae_lds_natgrad_f64(int n, int T, const double* packed_stats, const double* prior_flat,
                                    const double* params_flat, double num_batches, double scale,
                                    double* natgrad_flat, void* stream) {
  if (n < 1 || n > svae::GL_MAX_N) return -1;
  if (T < 1) return -2;
  if (!packed_stats) return -3;
  if (!prior_flat) return -4;
  if (!params_flat) return -5;
  if (!natgrad_flat) return -8;
  const int tot = (n + 2) * (n + 2) + 3 * n * n + 1;
  hipLaunchKernelGGL(svae::lds_natgrad_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     n, T, packed_stats, prior_flat, params_flat, num_batches, scale, natgrad_flat);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
