// hmm_estep_wide.hip -- batched HMM E-step for 17 <= K <= 64 discrete states on MI355X (gfx950): round 6, lifting the
// K <= 16 limit of hmm_estep.hip (one 16-lane DPP row per sequence) for the state counts the reference's SLDS was used
// with (its compiled kernels take any K: svae/hmm/cython_hmm_inference.pyx:93-166).
//
// What it replaces (reference = mattjj/svae, /root/reference), as hmm_estep.hip:
//   hmm_logZ        svae/hmm/cython_hmm_inference.pyx:93-121   (forward pass)
//   hmm_logZ_grad   svae/hmm/cython_hmm_inference.pyx:126-166  (at g = 1 the E-step: E[z_0], sum_t E[z_t z_{t+1}'], E[z_t])
//
// Mapping: ONE WAVEFRONT PER SEQUENCE, lane j = discrete state j (KP = 32 or 64 padded states; padding carries zero
// mass).  The transition matrix lives in registers -- its column j in the forward pass (KP doubles per lane), its row i
// in the backward pass (the two phases do not overlap) --, the vector that multiplies it (alpha^_t, then e o beta^ / c)
// is published through a KP-double LDS line and read back as wave-uniform (broadcast) loads: a step is KP multiply-adds
// per lane, the backward one 2 KP (the products P[i][j] w[j] also feed the K x K transition counts, KP accumulators per
// lane).  Scaled recursions as hmm_estep.hip: node log-potentials shifted by their maximum and exponentiated once per
// lane, alpha^ renormalised to sum 1 per step, log Z = sum of the log scales.
// Dynamic range: a sequence one of whose steps has a normaliser below 1e-200 (every path into the states the next
// observation allows ~460 nats below the transition matrix' maximum) is flagged and REDONE IN LOG SPACE -- the
// reference's own arithmetic, K log-sum-exps of K terms per step (LOGSPACE instantiation, second launch: workgroups of
// unflagged sequences leave at once).
// Bound: LDS broadcast reads + fp64 issue of one wavefront per sequence; the kernel is the general-K path, not a
// tuned one: K = 64, T = 500, 2048 sequences measured in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae_hip.h"
#include "dpp.hpp"
#include "hmm_args.hpp"

namespace svae {

__device__ __forceinline__ double wave_sum64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ double wave_max64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
  return x;
}

// publish one double per lane, then read all KP back as wave-uniform loads
template <int KP>
__device__ __forceinline__ void publish(double* line, int lane, double x) {
  __builtin_amdgcn_wave_barrier();
  line[lane] = x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KP, bool LOGSPACE>
__global__ __launch_bounds__(64) void hmm_estep_wide_kernel(const HmmArgs a) {
  constexpr int REC = hmm_wide_rec(KP);
  constexpr double NEG_BIG = -1.0e300;
  __shared__ double line[64];
  const int lane = threadIdx.x;
  const int K = a.K, T = a.T;
  const int braw = a.seq_index ? a.seq_index[blockIdx.x] : (int)blockIdx.x;
  if (braw < 0) return;                       // (indexed launches: unused slot)
  const long b = braw;
  const bool st = lane < K;
  const int cc = st ? lane : 0;
  double* wsb = a.ws + b * T * REC;
  if constexpr (LOGSPACE) {
    if (wsb[KP + 1] == 0.0) return;           // not flagged by the scaled pass
  }
  const double* pp = a.pair_params + b * a.pair_stride;
  const double* nd = a.node_params + (b * T) * K;

  // ---- forward pass: column `lane` of the transition matrix in registers ------------------------------------------
  double pmax = NEG_BIG;
  double Pc[KP];
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const double v = pp[(i < K ? i : 0) * K + cc];
    Pc[i] = (st && i < K) ? v : NEG_BIG;
    pmax = fmax(pmax, Pc[i]);
  }
  pmax = wave_max64(pmax);
  if constexpr (!LOGSPACE) {
#pragma unroll
    for (int i = 0; i < KP; ++i) Pc[i] = (st && i < K) ? exp(Pc[i] - pmax) : 0.0;
  }
  double logZ = 0.0, al;                      // al: alpha^_t[lane] (scaled) or log alpha_t[lane] (log space)
  bool flagged = false;
  {
    const double x = st ? a.init_params[cc] + nd[cc] : NEG_BIG;
    if constexpr (LOGSPACE) {
      al = x;
    } else {
      const double m = wave_max64(x);
      const double u = st ? exp(x - m) : 0.0;
      const double s = wave_sum64(u);
      al = u / s;
      logZ = m + ::log(s);
      if (lane == 0) wsb[KP] = s;
    }
    if (lane < KP) wsb[lane] = al;
  }
  for (int t = 1; t < T; ++t) {
    const double x = st ? nd[(long)t * K + cc] : NEG_BIG;
    publish<KP>(line, lane, al);
    if constexpr (LOGSPACE) {
      double m = NEG_BIG;
#pragma unroll
      for (int i = 0; i < KP; ++i) m = fmax(m, line[i] + Pc[i]);
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < KP; ++i) s += exp(line[i] + Pc[i] - m);
      al = st ? m + ::log(s) + x : NEG_BIG;
    } else {
      const double m = wave_max64(x);
      const double e = st ? exp(x - m) : 0.0;
      double v0 = 0.0, v1 = 0.0;
#pragma unroll
      for (int i = 0; i < KP; i += 2) {
        v0 = __builtin_fma(line[i], Pc[i], v0);
        v1 = __builtin_fma(line[i + 1], Pc[i + 1], v1);
      }
      const double u = (v0 + v1) * e;
      const double c = wave_sum64(u);
      flagged = flagged || !(c > HMM_WIDE_TINY);
      al = u / c;
      logZ += ::log(c) + m + pmax;
      if (lane == 0) wsb[(long)t * REC + KP] = c;
    }
    if (lane < KP) wsb[(long)t * REC + lane] = al;
  }
  if constexpr (LOGSPACE) {
    const double m = wave_max64(al);
    logZ = m + ::log(wave_sum64(st ? exp(al - m) : 0.0));
  } else {
    if (lane == 0) wsb[KP + 1] = flagged ? 1.0 : 0.0;          // the sequence's REDO flag (first record)
    if (flagged) return;                                       // (wave-uniform) the log-space launch takes it
  }
  if (lane == 0) a.logZ[b] = logZ;

  // ---- backward pass: row `lane` of the transition matrix in registers; transition counts row `lane` ---------------
  double Pr[KP], acc[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    const double v = pp[cc * K + (j < K ? j : 0)];
    if constexpr (LOGSPACE) Pr[j] = (st && j < K) ? v : NEG_BIG;
    else Pr[j] = (st && j < K) ? exp(v - pmax) : 0.0;
    acc[j] = 0.0;
  }
  double bt = LOGSPACE ? 0.0 : (st ? 1.0 : 0.0);              // beta^_{T-1} = 1  (log beta = 0)
  {
    const double g = LOGSPACE ? (st ? exp(al - logZ) : 0.0) : al;
    if (st) a.E_states[(b * T + (T - 1)) * K + lane] = g;
    if (T == 1 && st) a.E_init[b * K + lane] = g;
  }
  for (int t = T - 2; t >= 0; --t) {
    const double x = st ? nd[(long)(t + 1) * K + cc] : NEG_BIG;
    const double alt = wsb[(long)t * REC + (lane < KP ? lane : 0)];
    double g;
    if constexpr (LOGSPACE) {
      publish<KP>(line, lane, st ? x + bt : NEG_BIG);          // log (e o beta)_{t+1}
      double m = NEG_BIG;
#pragma unroll
      for (int j = 0; j < KP; ++j) m = fmax(m, Pr[j] + line[j]);
      double s = 0.0;
      const double base = alt - logZ;                          // xi_t[lane][j] = exp(log alpha_t + log P + log(e beta) - logZ)
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        const double q = Pr[j] + line[j];
        s += exp(q - m);
        acc[j] += st ? exp(base + q) : 0.0;
      }
      bt = st ? m + ::log(s) : NEG_BIG;
      g = st ? exp(alt + bt - logZ) : 0.0;
    } else {
      const double m = wave_max64(x);
      const double e = st ? exp(x - m) : 0.0;
      const double c = wsb[(long)(t + 1) * REC + KP];
      publish<KP>(line, lane, e * bt / c);                     // w_{t+1}
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int j = 0; j < KP; j += 2) {
        const double p0 = Pr[j] * line[j], p1 = Pr[j + 1] * line[j + 1];
        s0 += p0; s1 += p1;
        acc[j] = __builtin_fma(alt, p0, acc[j]);
        acc[j + 1] = __builtin_fma(alt, p1, acc[j + 1]);
      }
      bt = s0 + s1;
      g = alt * bt;
    }
    if (st) a.E_states[(b * T + t) * K + lane] = g;
    if (t == 0 && st) a.E_init[b * K + lane] = g;
  }
  if (st) {
#pragma unroll
    for (int j = 0; j < KP; ++j) if (j < K) a.E_trans[(b * K + lane) * K + j] = acc[j];
  }
}

}  // namespace svae

extern "C" int svae_hmm_wide_launch(const svae::HmmArgs* a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a->B), block(64);
  if (a->K <= 32) {
    hipLaunchKernelGGL((svae::hmm_estep_wide_kernel<32, false>), grid, block, 0, s, *a);
    hipLaunchKernelGGL((svae::hmm_estep_wide_kernel<32, true>), grid, block, 0, s, *a);
  } else {
    hipLaunchKernelGGL((svae::hmm_estep_wide_kernel<64, false>), grid, block, 0, s, *a);
    hipLaunchKernelGGL((svae::hmm_estep_wide_kernel<64, true>), grid, block, 0, s, *a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
