// lds_estep.hip -- C ABI (include/svae_hip.h) of the batched LDS E-step: argument checks, dispatch
// on the latent dimension to the per-n kernels (lds_estep_n.hip), and the deterministic batch
// reduction of the global statistics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/svae_hip.h"
#include "lds_args.hpp"
#include "per_device.hpp"

extern "C" {
#define SVAE_DECL_(NN) int svae_lds_launch_n##NN(const svae::LdsArgs*, int, void*); \
  int svae_lds_launch_split_n##NN(const svae::LdsArgs*, int, void*);           \
  int svae_lds_launch_twoend_n##NN(const svae::LdsArgs*, int, int, int, void*);          \
  int svae_lds_launch_twoend_mix_n##NN(const svae::LdsArgs*, void*);          \
  int svae_lds_launch_slds_rpc_n##NN(const svae::LdsArgs*, int, int, void*);  \
  int svae_lds_launch_filter_n##NN(const svae::LdsArgs*, int, void*);          \
  int svae_lds_launch_filter_split_n##NN(const svae::LdsArgs*, int, void*);    \
  int svae_lds_launch_filter_1r_n##NN(const svae::LdsArgs*, int, void*);       \
  int svae_lds_launch_forward_pair_n##NN(const svae::LdsArgs*, const svae::LdsArgs*, int, void*); \
  int svae_lds_sample_n##NN(const svae::SampleArgs*, void*);                   \
  int svae_lds_vjp_n##NN(const svae::VjpArgs*, void*);                         \
  int svae_lds_infer_lean_n##NN(const svae::LdsArgs*, const svae::LeanSample*, int, void*); \
  int svae_lds_vjp_lean_n##NN(const svae::VjpArgs*, void*);
#define SVAE_DECL(NN) SVAE_DECL_(NN)
#ifdef SVAE_ONLY_N   /* experimental single-n builds (tools/build_variant.sh) */
SVAE_DECL(SVAE_ONLY_N)
#else
SVAE_DECL(1) SVAE_DECL(2) SVAE_DECL(3) SVAE_DECL(4) SVAE_DECL(5) SVAE_DECL(6) SVAE_DECL(7)
SVAE_DECL(8) SVAE_DECL(9) SVAE_DECL(10) SVAE_DECL(11) SVAE_DECL(12) SVAE_DECL(13) SVAE_DECL(14)
SVAE_DECL(15)
#endif
#undef SVAE_DECL
/* 16 <= n <= SVAE_LDS_TILE_MAX_N: LDS-tiled MFMA path (lds_estep_tile.hip) */
int svae_lds_launch_tile(const svae::LdsArgs*, int n, int inhomog, void* stream);
size_t svae_lds_tile_step_doubles(int n);
size_t svae_lds_tile_packed_doubles(int B, int T, int n, int inhomog, int pair_batched);
}

namespace svae {

// ---- deterministic batch reduction of the global statistics ------------------------------------
// out = [sum_b E_init (n^2+n) | sum_b E_pair (3 n^2) | sum_b lognorm | B].  Fixed summation order => bit-reproducible.
// A workgroup takes 8 consecutive statistics: thread (jj, ch) = (tid & 7, tid >> 3) sums the sequences b = ch, ch + 32,
// ... of statistic j0 + jj in four interleaved partial sums, sixteen requests in flight (the 8 lanes of a chunk read one
// 64-byte sector of a sequence's block: the first version read one statistic with a stride of n^2 + n doubles across
// the lanes -- every lane its own sector, 13 MB of traffic for 1.7 MB of data at 512 sequences), then the 32 chunks
// meet in LDS.
__global__ __launch_bounds__(256) void lds_reduce_stats_kernel(int B, int n, const double* __restrict__ E_init,
                                                               const double* __restrict__ E_pair,
                                                               const double* __restrict__ lognorm, double* out) {
  const int ni = n * n + n, np = 3 * n * n, tot = ni + np + 1;
  __shared__ double red[256], red2[64];
  const int jj = threadIdx.x & 7, ch = threadIdx.x >> 3;
  const int j = blockIdx.x * 8 + jj;
  const double* src;
  long stride;
  if (j < ni) { src = E_init + j; stride = ni; }
  else if (j < ni + np) { src = E_pair + (j - ni); stride = np; }
  else { src = lognorm; stride = 1; }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (j < tot) {
    int b = ch;
    for (; b + 31 * 32 < B; b += 32 * 32) {       // (large batches: 32 requests in flight)
      double v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = src[(long)(b + 32 * u) * stride];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc[u & 3] += v[u];
    }
    for (; b + 15 * 32 < B; b += 16 * 32) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(long)(b + 32 * u) * stride];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
    }
    for (int u = 0; b < B; b += 32, ++u) acc[u & 3] += src[(long)b * stride];
  }
  red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (threadIdx.x < 64) {                       // (jj, q): four chunk groups of eight per statistic, then a 4-lane sum
    const int q = threadIdx.x >> 3;             // 0 .. 7
    double v = ((red[8 * (4 * q) + jj] + red[8 * (4 * q + 1) + jj]) + (red[8 * (4 * q + 2) + jj] + red[8 * (4 * q + 3) + jj]));
    red2[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8 && j < tot) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += red2[8 * q + jj];
    out[j] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[tot] = (double)B;
}

}  // namespace svae

#ifndef SVAE_FILTER_WIDE_MAX_B
#define SVAE_FILTER_WIDE_MAX_B 6144     // filter-only launches without message outputs: the one-register filter (two sequences per wavefront beyond 512) up to this batch; measured 4096: 0.89 vs 0.96 ms packed, 8192: equal
#endif

extern "C" {

// Kernel selection (include/svae_hip.h, SVAE_OPT_*): keep == 0, n <= 10, T >= 4 -> the two-ended kernel
// (lds_estep_twoend.hpp: one sequence per wavefront, both elimination chains in one instruction stream).
// Otherwise batches up to 1023 run the one-directional latency variant (one sequence per wavefront, product
// stages split across the DPP rows: lds_estep_split.hpp), larger ones the packed kernel (four sequences per
// wavefront).  Measured crossover split/packed on MI355X (T=200, n=10): B ~ 1024 (one wavefront per SIMD).
// The selection is a function of the call's arguments only: the library holds no process-global state.
struct Selection { int twoend; bool split; int layout; int prod_max_b; };
static bool decode_options(unsigned options, int B, Selection* s) {
  if ((options & SVAE_OPT_LEAN_ON) && (options & SVAE_OPT_LEAN_OFF)) return false;
  options &= ~(SVAE_OPT_LEAN_ON | SVAE_OPT_LEAN_OFF | SVAE_OPT_INFER_RECORDS);   /* (record format: lean_applies) */
  if (options & ~SVAE_OPT_ALL) return false;
  if ((options & SVAE_OPT_TWOEND_OFF) && (options & SVAE_OPT_TWOEND_FULL)) return false;
  if ((options & SVAE_OPT_LAYOUT_SPLIT) && (options & SVAE_OPT_LAYOUT_PACKED)) return false;
  if ((options & SVAE_OPT_PRODUCERS_ON) && (options & SVAE_OPT_PRODUCERS_OFF)) return false;
  s->twoend = (options & SVAE_OPT_TWOEND_OFF) ? 0 : (options & SVAE_OPT_TWOEND_FULL) ? 2 : 1;
  s->layout = (options & SVAE_OPT_LAYOUT_SPLIT) ? 1 : (options & SVAE_OPT_LAYOUT_PACKED) ? 2 : 0;
  s->split = s->layout == 1 || (s->layout == 0 && B <= 1023);
  s->prod_max_b = (options & SVAE_OPT_PRODUCERS_ON) ? 0x7fffffff : (options & SVAE_OPT_PRODUCERS_OFF) ? 0 : 1024;
  return true;
}

int svae_hip_abi_version(void) { return SVAE_HIP_ABI_VERSION; }

// Record format of svae_lds_inference_f64 (and of the svae_lds_estep_vjp_ex_f64 call that reads its workspace with
// SVAE_OPT_INFER_RECORDS): a pure function of the arguments both calls share -- the library keeps no state.
static bool lean_applies(int B, int T, int n, int S, int inhomog, int keep_vjp, unsigned options) {
  if (n > svae::LEAN_MAX_N || T < 2 || S < 0 || S > svae::LEAN_MAX_S) return false;
  if (inhomog && keep_vjp) return false;          // per-step pair parameters: forward only (the VJP with statistics cotangents reads the full records)
  if ((long)T * svae::lean_rec_doubles(n) * 8 * 4 >= (1l << 31)) return false;     // 32-bit lane offsets inside a wavefront's records
  if (options & SVAE_OPT_LEAN_OFF) return false;
  if (options & SVAE_OPT_LEAN_ON) return true;
  if (options & (SVAE_OPT_LAYOUT_SPLIT | SVAE_OPT_PRODUCERS_ON)) return false;
  return B >= svae::LEAN_MIN_B;
}
int svae_lds_inference_is_lean(int B, int T, int n, int S, int inhomog, int keep_vjp, unsigned options) {
  return lean_applies(B, T, n, S, inhomog, keep_vjp, options) ? 1 : 0;
}

// main region: the one-directional layout (lds_args.hpp), then -- n <= 10 -- the two-ended one (a launch that
// keeps the sampler / VJP hand-off runs BOTH kernels, see svae_lds_estep_f64), then B doubles of scratch
static size_t one_ws_doubles(int B, int T, int n) { return (size_t)B * (size_t)svae::ws_seq_doubles(n, T); }
static size_t main_ws_doubles(int B, int T, int n) {
  const long two = n <= svae::TE_MAX_N ? svae::te_seq_doubles(n, T) : 0;
  return one_ws_doubles(B, T, n) + (size_t)B * (size_t)two + (size_t)B;
}
static size_t factor_ws_doubles(int B, int T, int n) { return (size_t)B * T * (n * n + n); }
static size_t cross_ws_doubles(int B, int T, int n) { return (size_t)B * T * (n + 1) * svae::ws_h_stride(n); }

size_t svae_lds_tile_sigma_offset_bytes(int B, int T, int n, int inhomog, int pair_batched) {
  if (n <= SVAE_LDS_MAX_N || n > SVAE_LDS_TILE_MAX_N) return 0;
  return (svae_lds_workspace_bytes_ex(B, T, n, inhomog, pair_batched) + 255) / 256 * 256;
}

size_t svae_lds_workspace_bytes_ex(int B, int T, int n, int inhomog, int pair_batched) {
  if (B <= 0 || T <= 0 || n <= 0 || n > SVAE_LDS_TILE_MAX_N) return 0;
  if (n <= SVAE_LDS_MAX_N) return svae_lds_workspace_bytes(B, T, n);
  return ((size_t)B * T * svae_lds_tile_step_doubles(n) +
          svae_lds_tile_packed_doubles(B, T, n, inhomog, pair_batched)) * sizeof(double);
}

size_t svae_lds_workspace_bytes(int B, int T, int n) {
  if (B <= 0 || T <= 0 || n <= 0 || n > SVAE_LDS_TILE_MAX_N) return 0;
  if (n > SVAE_LDS_MAX_N) return svae_lds_workspace_bytes_ex(B, T, n, 0, 0);
  return (main_ws_doubles(B, T, n) + factor_ws_doubles(B, T, n) + cross_ws_doubles(B, T, n)) * sizeof(double);
}

int svae_lds_estep_f64(int B, int T, int n, int inhomog, int pair_batched, int keep, unsigned options,
                       const double* init_J, const double* init_h, const double* init_logZ,
                       const double* J11, const double* J12, const double* J22,
                       const double* logZ_pair,
                       const double* node_J, const double* node_h, const double* node_logZ,
                       double* lognorm, double* E_init, double* E_pair,
                       double* E_node_diagxx, double* E_node_x,
                       int32_t* info, void* workspace, size_t ws_bytes, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (n < 1 || n > SVAE_LDS_TILE_MAX_N) return -3;
  if (n > SVAE_LDS_MAX_N ? (keep & ~SVAE_KEEP_SIGMA) != 0 : (keep & ~3) != 0) return -23;   /* factor / cross regions: register path; Sigma: tiled path */
  if (pair_batched && !inhomog) return -5;
  if (!init_J) return -6;
  if (!init_h) return -7;
  if (!init_logZ) return -8;
  if (T > 1 && (!J11 || !J12 || !J22 || !logZ_pair)) return -9;
  if (!node_J) return -13;
  if (!node_h) return -14;
  if (!lognorm) return -16;
  if (!E_init) return -17;
  if (!E_pair) return -18;
  if (!E_node_diagxx) return -19;
  if (!E_node_x) return -20;
  if (!info) return -21;
  if (!workspace || ws_bytes < svae_lds_workspace_bytes_ex(B, T, n, inhomog, pair_batched)) return -22;
  Selection sel;
  const unsigned tile_bits = options & (SVAE_OPT_TILE_FORWARD | SVAE_OPT_TILE_BACKWARD);
  if (tile_bits == (SVAE_OPT_TILE_FORWARD | SVAE_OPT_TILE_BACKWARD) || (tile_bits && n <= SVAE_LDS_MAX_N)) return -24;
  if (!decode_options(options & ~tile_bits, B, &sel)) return -24;
  if (B == 0) return 0;
  const int twoend = sel.twoend;

  svae::LdsArgs a;
  a.tile_half = (tile_bits & SVAE_OPT_TILE_FORWARD) ? 1 : (tile_bits & SVAE_OPT_TILE_BACKWARD) ? 2 : 0;
  a.B = B; a.T = T;
  a.init_J = init_J; a.init_h = init_h; a.init_logZ = init_logZ;
  a.J11 = J11; a.J12 = J12; a.J22 = J22; a.logZ_pair = logZ_pair;
  a.node_J = node_J; a.node_h = node_h; a.node_logZ = node_logZ;
  a.lognorm = lognorm; a.E_init = E_init; a.E_pair = E_pair;
  a.E_node_diagxx = E_node_diagxx; a.E_node_x = E_node_x;
  a.info = info; a.ws = (double*)workspace;
  a.ws2 = (keep & 1) ? (double*)workspace + main_ws_doubles(B, T, n) : nullptr;
  a.ws3 = (keep & 2) ? (double*)workspace + main_ws_doubles(B, T, n) + factor_ws_doubles(B, T, n) : nullptr;
  a.pair_seq_stride = pair_batched ? (long)(T - 1) * n * n : 0;
  a.msg_Jp = a.msg_hp = a.msg_Jf = a.msg_hf = nullptr;
  a.mix_w = nullptr; a.mix_out = nullptr; a.seq_index = nullptr; a.mix_K = 0; a.lds_keep = 0;
  a.sig_out = nullptr;
  if (n > SVAE_LDS_MAX_N) {
    a.ws2 = a.ws3 = nullptr;
    if (keep & SVAE_KEEP_SIGMA) {
      const size_t off = svae_lds_tile_sigma_offset_bytes(B, T, n, inhomog, pair_batched);
      if (ws_bytes < off + (size_t)B * T * n * n * sizeof(double)) return -22;
      a.sig_out = (double*)((char*)workspace + off);
    }
    return svae_lds_launch_tile(&a, n, inhomog, stream);
  }
  const bool split = sel.split;
  if (twoend && keep && split && n <= svae::TE_MAX_N && T >= svae::TE_MIN_T) {
    // Small batch, hand-off kept: the statistics (and the cross moments the VJP reads: posterior moments, the same
    // whichever way the chain is eliminated) come from the two-ended kernel, while the one-directional FILTER --
    // whose factorisation defines the sampler's eps -> sample map and the records the sweeps differentiate --
    // runs NEXT to it on the SIMDs the small batch leaves idle (512 + 512 wavefronts on 1024 SIMDs at B = 512).
    // Round 4: both in ONE launch (lds_forward_pair_kernel: the first B workgroups take the filter's body, the next
    // B the E-step's) instead of two kernels on two streams forked and joined by events (0.31 -> 0.27 ms at B = 512).
    svae::LdsArgs f = a;                       // the filter: one-directional layout at the start of the workspace
    f.ws2 = (double*)workspace + main_ws_doubles(B, T, n);
    f.ws3 = nullptr;
    f.lognorm = (double*)workspace + main_ws_doubles(B, T, n) - B;      // scratch (the E-step's lognorm is the other kernel's)
    f.E_init = nullptr; f.E_pair = nullptr; f.E_node_diagxx = nullptr; f.E_node_x = nullptr;
    svae::LdsArgs e = a;                       // the E-step: two-ended records behind the one-directional ones
    e.ws = (double*)workspace + one_ws_doubles(B, T, n);
    e.ws2 = nullptr;
    switch (n) {
#define SVAE_CASE_(NN) case NN: return svae_lds_launch_forward_pair_n##NN(&f, &e, inhomog, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
      SVAE_CASE(SVAE_ONLY_N)
#else
      SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
      SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
    }
    return -3;
  }
  if (twoend && !keep && n <= svae::TE_MAX_N && T >= svae::TE_MIN_T) {
    switch (n) {
#define SVAE_CASE_(NN) case NN: return svae_lds_launch_twoend_n##NN(&a, inhomog, twoend == 1 && !inhomog, sel.layout, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
      SVAE_CASE(SVAE_ONLY_N)
#else
      SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
      SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
    }
  }
  switch (n) {
#define SVAE_CASE_(NN) case NN: return split ? svae_lds_launch_split_n##NN(&a, inhomog, stream) \
                                             : svae_lds_launch_n##NN(&a, inhomog, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
    SVAE_CASE(SVAE_ONLY_N)
#else
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10) SVAE_CASE(11) SVAE_CASE(12) SVAE_CASE(13)
    SVAE_CASE(14) SVAE_CASE(15)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
  }
  return -3;
}

int svae_lds_filter_f64(int B, int T, int n, int inhomog, int pair_batched, unsigned options,
                        const double* init_J, const double* init_h, const double* init_logZ,
                        const double* J11, const double* J12, const double* J22, const double* logZ_pair,
                        const double* node_J, const double* node_h, const double* node_logZ,
                        double* lognorm, double* J_pred, double* h_pred, double* J_filt, double* h_filt,
                        int32_t* info, void* workspace, size_t ws_bytes, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (n < 1 || n > SVAE_LDS_MAX_N) return -3;
  if (pair_batched && !inhomog) return -5;
  if (!init_J) return -6;
  if (!init_h) return -7;
  if (!init_logZ) return -8;
  if (T > 1 && (!J11 || !J12 || !J22 || !logZ_pair)) return -9;
  if (!node_J) return -13;
  if (!node_h) return -14;
  if (!lognorm) return -16;
  if (!info) return -21;
  if (!workspace || ws_bytes < svae_lds_workspace_bytes(B, T, n)) return -22;
  Selection sel;
  if (!decode_options(options, B, &sel)) return -24;
  if (B == 0) return 0;
  const int twoend = sel.twoend;
  svae::LdsArgs a;
  a.B = B; a.T = T;
  a.init_J = init_J; a.init_h = init_h; a.init_logZ = init_logZ;
  a.J11 = J11; a.J12 = J12; a.J22 = J22; a.logZ_pair = logZ_pair;
  a.node_J = node_J; a.node_h = node_h; a.node_logZ = node_logZ;
  a.lognorm = lognorm; a.E_init = nullptr; a.E_pair = nullptr;
  a.E_node_diagxx = nullptr; a.E_node_x = nullptr;
  a.info = info; a.ws = (double*)workspace;
  a.ws2 = (double*)workspace + main_ws_doubles(B, T, n);      // factor region: the sampler may follow
  a.ws3 = nullptr;
  a.pair_seq_stride = pair_batched ? (long)(T - 1) * n * n : 0;
  a.msg_Jp = J_pred; a.msg_hp = h_pred; a.msg_Jf = J_filt; a.msg_hf = h_filt;
  a.mix_w = nullptr; a.mix_out = nullptr; a.seq_index = nullptr; a.mix_K = 0; a.lds_keep = 0; a.tile_half = 0; a.sig_out = nullptr;
  // small batches without message outputs: one sequence per wavefront (0.62 -> 0.24 ms at B = 512, T = 200, n = 10).
  // The one-register filter (n <= 10) stays ahead of the packed kernel until ~3 wavefronts per SIMD (filter + sampler,
  // T = 500: 1024 sequences 0.86 vs 1.81 ms, 2048: 1.85 vs 2.53; T = 200, 4096: 1.43 vs 1.27)
  const bool wide = sel.layout == 1 || (sel.layout == 0 && B <= (n <= svae::TE_MAX_N && twoend ? SVAE_FILTER_WIDE_MAX_B : 1023));
  const bool fsplit = wide && !J_pred && !h_pred && !J_filt && !h_filt;
  switch (n) {
#define SVAE_CASE_(NN) case NN: return !fsplit ? svae_lds_launch_filter_n##NN(&a, inhomog, stream)          \
                                       : (NN <= svae::TE_MAX_N && twoend) ? svae_lds_launch_filter_1r_n##NN(&a, inhomog, stream) \
                                       : svae_lds_launch_filter_split_n##NN(&a, inhomog, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
    SVAE_CASE(SVAE_ONLY_N)
#else
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10) SVAE_CASE(11) SVAE_CASE(12) SVAE_CASE(13)
    SVAE_CASE(14) SVAE_CASE(15)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
  }
  return -3;
}

size_t svae_slds_lds_meanfield_workspace_bytes(int rows, int T, int n) {
  if (rows <= 0 || T < svae::TE_MIN_T || n < 1 || n > svae::TE_MAX_N) return 0;
  return (size_t)rows * (size_t)svae::te_seq_doubles(n, T) * sizeof(double);     // the two-ended records only
}

size_t svae_slds_lds_meanfield_lds_bytes(int n, int K) {
  if (n < 1 || n > svae::TE_MAX_N || K < 1 || K > svae::TE_MIX_MAX_K) return 0;
  return (size_t)svae::te_mix_lds_bytes(n, K);
}

int svae_slds_lds_meanfield_f64(int B, int rows, int T, int n, int K,
                                const double* init_J, const double* init_h,
                                const double* J11, const double* J12, const double* J22,
                                const double* weights,
                                const double* node_J, const double* node_h, const double* node_logZ,
                                const int32_t* seq_index,
                                double* lognorm, double* E_init, double* E_node_diagxx, double* E_node_x,
                                double* pair_contr, int32_t* info,
                                void* workspace, size_t ws_bytes, unsigned options, void* stream) {
  if (B < 0 || B > rows) return -1;
  if (T < svae::TE_MIN_T) return -2;
  if (n < 1 || n > svae::TE_MAX_N) return -3;
  if (K < 1 || K > svae::TE_MIX_MAX_K || svae::te_mix_lds_bytes(n, K) > svae::TE_MIX_MAX_LDS) return -4;
  if (!init_J) return -5;
  if (!init_h) return -6;
  if (!J11 || !J12 || !J22) return -7;
  if (!weights) return -10;
  if (!node_J) return -11;
  if (!node_h) return -12;
  if (!lognorm) return -15;
  if (!E_init) return -16;
  if (!E_node_diagxx) return -17;
  if (!E_node_x) return -18;
  if (!pair_contr) return -19;
  if (!info) return -20;
  if (!workspace || ws_bytes < svae_slds_lds_meanfield_workspace_bytes(rows, T, n)) return -21;
  // kernel selection (a pure function of the arguments): SVAE_OPT_LAYOUT_SPLIT = one sequence per wavefront, the K
  // parameter sets as LDS tables (rounds 2 - 4); SVAE_OPT_LAYOUT_PACKED = row-per-chain consumers + MFMA producer
  // wavefronts (round 5; K <= 8, workspace below 4 GiB: 32-bit lane offsets); 0 = the latter where it applies.
  // SVAE_OPT_PRODUCERS_OFF with the packed layout: reference producers (plain loops, no MFMA: test infrastructure).
  if (options & ~(SVAE_OPT_LAYOUT_SPLIT | SVAE_OPT_LAYOUT_PACKED | SVAE_OPT_PRODUCERS_OFF)) return -24;
  if ((options & SVAE_OPT_LAYOUT_SPLIT) && (options & SVAE_OPT_LAYOUT_PACKED)) return -24;
  const bool rpc_ok = K <= 8 && svae_slds_lds_meanfield_workspace_bytes(rows, T, n) < (1ull << 32);
  if ((options & SVAE_OPT_LAYOUT_PACKED) && !rpc_ok) return -24;
  const bool rpc = rpc_ok && !(options & SVAE_OPT_LAYOUT_SPLIT);
  if (B == 0) return 0;
  svae::LdsArgs a;
  a.B = B; a.T = T;
  a.init_J = init_J; a.init_h = init_h; a.init_logZ = nullptr;
  a.J11 = J11; a.J12 = J12; a.J22 = J22; a.logZ_pair = nullptr;
  a.node_J = node_J; a.node_h = node_h; a.node_logZ = node_logZ;
  a.lognorm = lognorm; a.E_init = E_init; a.E_pair = nullptr;
  a.E_node_diagxx = E_node_diagxx; a.E_node_x = E_node_x;
  a.info = info; a.ws = (double*)workspace; a.ws2 = nullptr; a.ws3 = nullptr;
  a.pair_seq_stride = 0;
  a.msg_Jp = a.msg_hp = a.msg_Jf = a.msg_hf = nullptr;
  a.mix_w = weights; a.mix_out = pair_contr; a.seq_index = seq_index; a.mix_K = K; a.lds_keep = 0; a.tile_half = 0; a.sig_out = nullptr;
  switch (n) {
#define SVAE_CASE_(NN) case NN: return rpc ? svae_lds_launch_slds_rpc_n##NN(&a, (options & SVAE_OPT_PRODUCERS_OFF) ? 1 : 0, \
                                                                              (options & SVAE_OPT_LAYOUT_PACKED) ? 0 : 1, stream) \
                                        : svae_lds_launch_twoend_mix_n##NN(&a, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
    SVAE_CASE(SVAE_ONLY_N)
#else
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
  }
  return -3;
}

int svae_lds_reduce_stats_f64(int B, int n, const double* E_init, const double* E_pair,
                              const double* lognorm, double* out, void* stream) {
  if (B < 0) return -1;
  if (n < 1 || n > SVAE_LDS_TILE_MAX_N) return -2;
  if (!E_init) return -3;
  if (!E_pair) return -4;
  if (!lognorm) return -5;
  if (!out) return -6;
  const int tot = 4 * n * n + n + 1;
  hipLaunchKernelGGL(svae::lds_reduce_stats_kernel, dim3((tot + 7) / 8), dim3(256), 0,
                     (hipStream_t)stream, B, n, E_init, E_pair, lognorm, out);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // extern "C"

extern "C" int svae_lds_sample_f64(int B, int T, int n, int S, unsigned options, const double* eps, double* samples,
                                   const void* workspace, size_t ws_bytes, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (n < 1 || n > SVAE_LDS_MAX_N) return -3;
  if (S < 1) return -4;
  if (!eps) return -5;
  if (!samples) return -6;
  if (!workspace || ws_bytes < svae_lds_workspace_bytes(B, T, n)) return -7;
  Selection sel;
  if (!decode_options(options, B, &sel)) return -24;
  if (B == 0) return 0;
  svae::SampleArgs a;
  a.B = B; a.T = T; a.S = S; a.eps = eps; a.samples = samples;
  a.prod_max_b = sel.prod_max_b;
  a.ws = (const double*)workspace;
  a.ws2 = (const double*)workspace + main_ws_doubles(B, T, n);
  switch (n) {
#define SVAE_CASE_(NN) case NN: return svae_lds_sample_n##NN(&a, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
    SVAE_CASE(SVAE_ONLY_N)
#else
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10) SVAE_CASE(11) SVAE_CASE(12) SVAE_CASE(13)
    SVAE_CASE(14) SVAE_CASE(15)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
  }
  return -3;
}

extern "C" int svae_lds_inference_f64(int B, int T, int n, int S, int inhomog, int pair_batched, int keep_vjp, unsigned options,
                                      const double* init_J, const double* init_h, const double* init_logZ,
                                      const double* J11, const double* J12, const double* J22, const double* logZ_pair,
                                      const double* node_J, const double* node_h, const double* node_logZ,
                                      const double* eps, double* samples,
                                      double* lognorm, double* E_init, double* E_pair,
                                      double* E_node_diagxx, double* E_node_x,
                                      int32_t* info, void* workspace, size_t ws_bytes, void* stream) {
  if (n < 1 || n > SVAE_LDS_MAX_N) return -3;
  if (S < 0 || (S > 0 && (!eps || !samples))) return -4;
  if (!lean_applies(B, T, n, S, inhomog, keep_vjp, options)) {
    // the general path: E-step keeping the hand-off, the factor and (keep_vjp) the cross moments, then the sampler
    const unsigned o = options & SVAE_OPT_ALL;
    const int rc = svae_lds_estep_f64(B, T, n, inhomog, pair_batched, keep_vjp ? 3 : (S > 0 ? 1 : 0), o, init_J, init_h, init_logZ, J11, J12, J22,
                                      logZ_pair, node_J, node_h, node_logZ, lognorm, E_init, E_pair, E_node_diagxx,
                                      E_node_x, info, workspace, ws_bytes, stream);
    if (rc != 0 || S == 0 || B == 0) return rc;
    const int rs = svae_lds_sample_f64(B, T, n, S, o, eps, samples, workspace, ws_bytes, stream);
    return rs == 0 ? 0 : -100 + rs;
  }
  if (B < 0) return -1;
  if (!init_J) return -6;
  if (!init_h) return -7;
  if (!init_logZ) return -8;
  if (!J11 || !J12 || !J22 || !logZ_pair) return -9;
  if (pair_batched && !inhomog) return -5;
  if (!node_J) return -13;
  if (!node_h) return -14;
  if (!lognorm) return -16;
  if (!E_init) return -17;
  if (!E_pair) return -18;
  if (!E_node_diagxx) return -19;
  if (!E_node_x) return -20;
  if (!info) return -21;
  if (!workspace || ws_bytes < svae_lds_workspace_bytes(B, T, n)) return -22;
  Selection sel;
  if (!decode_options(options, B, &sel)) return -24;
  if (B == 0) return 0;
  svae::LdsArgs a;
  a.tile_half = 0;
  a.B = B; a.T = T;
  a.init_J = init_J; a.init_h = init_h; a.init_logZ = init_logZ;
  a.J11 = J11; a.J12 = J12; a.J22 = J22; a.logZ_pair = logZ_pair;
  a.node_J = node_J; a.node_h = node_h; a.node_logZ = node_logZ;
  a.lognorm = lognorm; a.E_init = E_init; a.E_pair = E_pair;
  a.E_node_diagxx = E_node_diagxx; a.E_node_x = E_node_x;
  a.info = info; a.ws = (double*)workspace;        // lean records at the start of the main region
  a.ws2 = nullptr;
  a.ws3 = keep_vjp ? (double*)workspace + main_ws_doubles(B, T, n) + factor_ws_doubles(B, T, n) : nullptr;   // cross moments: where the VJP looks
  a.pair_seq_stride = pair_batched ? (long)(T - 1) * n * n : 0;
  a.msg_Jp = a.msg_hp = a.msg_Jf = a.msg_hf = nullptr;
  a.mix_w = nullptr; a.mix_out = nullptr; a.seq_index = nullptr; a.mix_K = 0; a.lds_keep = 0;
  a.sig_out = nullptr;
  svae::LeanSample ls;
  ls.S = S; ls.eps = eps; ls.samples = samples;
  switch (n) {
#define SVAE_CASE_(NN) case NN: return svae_lds_infer_lean_n##NN(&a, &ls, inhomog, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
    SVAE_CASE(SVAE_ONLY_N)
#else
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
  }
  return -3;
}

extern "C" size_t svae_lds_vjp_workspace_bytes(int B, int T, int n) {
  if (B <= 0 || T <= 0 || n <= 0 || n > SVAE_LDS_MAX_N) return 0;
  return (size_t)B * T * svae::vjp_step_doubles(n) * sizeof(double);
}

static int vjp_impl(int B, int T, int n, int S, int inhomog, int pair_batched, unsigned options,
                    const double* J12, const double* g_lognorm,
                    const double* g_E_node_diagxx, const double* g_E_node_x,
                    const double* g_E_init, const double* g_E_pair,
                    const double* g_samples, const double* eps,
                    const double* samples, const double* E_pair,
                    const double* E_node_x, double* g_node_J, double* g_node_h, double* g_node_J_dense,
                    const void* workspace, size_t ws_bytes,
                    void* vjp_workspace, size_t vjp_ws_bytes, void* stream) {
  if (B < 0) return -1;
  if (T < 1) return -2;
  if (n < 1 || n > SVAE_LDS_MAX_N) return -3;
  if (g_samples && (S < 1 || S > 16)) return -4;
  if (T > 1 && !J12) return -5;
  if (!g_lognorm) return -6;
  if (pair_batched && !inhomog) return -7;
  if (g_E_pair && (!inhomog || !E_pair || !E_node_x)) return -8;   /* per-step statistics only */
  if (g_samples && (!eps || !samples)) return -10;
  if (!g_node_J) return -12;
  if (!g_node_h) return -13;
  if (!workspace || ws_bytes < svae_lds_workspace_bytes(B, T, n)) return -14;
  if (!vjp_workspace || vjp_ws_bytes < svae_lds_vjp_workspace_bytes(B, T, n)) return -16;
  Selection sel;
  if (!decode_options(options, B, &sel)) return -24;
  if (B == 0) return 0;
  svae::VjpArgs a;
  a.B = B; a.T = T; a.S = g_samples ? S : 0;
  a.prod_max_b = sel.prod_max_b;
  a.J12 = J12; a.g_lognorm = g_lognorm; a.g_diagxx = g_E_node_diagxx; a.g_x = g_E_node_x;
  a.pair_t_stride = inhomog ? (long)n * n : 0;
  a.pair_seq_stride = pair_batched ? (long)(T - 1) * n * n : 0;
  a.g_E_init = g_E_init; a.g_E_pair = g_E_pair; a.E_pair = E_pair; a.E_node_x = E_node_x;
  a.g_samples = g_samples; a.eps = eps; a.samples = samples;
  a.g_node_J = g_node_J; a.g_node_h = g_node_h;
  a.ws = (const double*)workspace;
  a.ws2 = a.ws + main_ws_doubles(B, T, n);
  a.ws3 = a.ws2 + factor_ws_doubles(B, T, n);
  a.adj = (double*)vjp_workspace;
  a.g_P = g_node_J_dense;
  if (g_node_J_dense) a.prod_max_b = 0;          /* the packed sweeps write it */
  if ((options & SVAE_OPT_INFER_RECORDS) && lean_applies(B, T, n, S, inhomog, 1, options)) {
    if (g_E_init || g_E_pair || g_node_J_dense) return -8;        /* lean records: cotangents of the node statistics, lognorm and samples */
    switch (n) {
#define SVAE_CASE_(NN) case NN: return svae_lds_vjp_lean_n##NN(&a, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
      SVAE_CASE(SVAE_ONLY_N)
#else
      SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
      SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
    }
    return -3;
  }
  switch (n) {
#define SVAE_CASE_(NN) case NN: return svae_lds_vjp_n##NN(&a, stream);
#define SVAE_CASE(NN) SVAE_CASE_(NN)
#ifdef SVAE_ONLY_N
    SVAE_CASE(SVAE_ONLY_N)
#else
    SVAE_CASE(1) SVAE_CASE(2) SVAE_CASE(3) SVAE_CASE(4) SVAE_CASE(5) SVAE_CASE(6) SVAE_CASE(7)
    SVAE_CASE(8) SVAE_CASE(9) SVAE_CASE(10) SVAE_CASE(11) SVAE_CASE(12) SVAE_CASE(13)
    SVAE_CASE(14) SVAE_CASE(15)
#endif
#undef SVAE_CASE
#undef SVAE_CASE_
  }
  return -3;
}

extern "C" int svae_lds_estep_vjp_ex_f64(int B, int T, int n, int S, int inhomog, int pair_batched, unsigned options,
                                         const double* J12, const double* g_lognorm,
                                         const double* g_E_node_diagxx, const double* g_E_node_x,
                                         const double* g_E_init, const double* g_E_pair,
                                         const double* g_samples, const double* eps,
                                         const double* samples, const double* E_pair,
                                         const double* E_node_x, double* g_node_J, double* g_node_h,
                                         const void* workspace, size_t ws_bytes,
                                         void* vjp_workspace, size_t vjp_ws_bytes, void* stream) {
  return vjp_impl(B, T, n, S, inhomog, pair_batched, options, J12, g_lognorm, g_E_node_diagxx, g_E_node_x, g_E_init, g_E_pair,
                  g_samples, eps, samples, E_pair, E_node_x, g_node_J, g_node_h, nullptr, workspace, ws_bytes,
                  vjp_workspace, vjp_ws_bytes, stream);
}

extern "C" int svae_lds_estep_vjp_dense_f64(int B, int T, int n, int S, int inhomog, int pair_batched, unsigned options,
                                            const double* J12, const double* g_lognorm,
                                            const double* g_E_node_diagxx, const double* g_E_node_x,
                                            const double* g_E_init, const double* g_E_pair,
                                            const double* g_samples, const double* eps,
                                            const double* samples, const double* E_pair,
                                            const double* E_node_x, double* g_node_J, double* g_node_h,
                                            double* g_node_J_dense,
                                            const void* workspace, size_t ws_bytes,
                                            void* vjp_workspace, size_t vjp_ws_bytes, void* stream) {
  if (!g_node_J_dense) return -27;
  return vjp_impl(B, T, n, S, inhomog, pair_batched, options, J12, g_lognorm, g_E_node_diagxx, g_E_node_x, g_E_init, g_E_pair,
                  g_samples, eps, samples, E_pair, E_node_x, g_node_J, g_node_h, g_node_J_dense, workspace, ws_bytes,
                  vjp_workspace, vjp_ws_bytes, stream);
}

extern "C" int svae_lds_estep_vjp_f64(int B, int T, int n, int S, const double* J12,
                                      const double* g_lognorm, const double* g_E_node_diagxx,
                                      const double* g_E_node_x, const double* g_samples,
                                      const double* eps, const double* samples,
                                      double* g_node_J, double* g_node_h,
                                      const void* workspace, size_t ws_bytes,
                                      void* vjp_workspace, size_t vjp_ws_bytes, void* stream) {
  const int rc = svae_lds_estep_vjp_ex_f64(B, T, n, S, 0, 0, SVAE_OPT_DEFAULT, J12, g_lognorm, g_E_node_diagxx, g_E_node_x,
                                           nullptr, nullptr, g_samples, eps, samples, nullptr, nullptr,
                                           g_node_J, g_node_h, workspace, ws_bytes, vjp_workspace,
                                           vjp_ws_bytes, stream);
  return rc == -12 ? -12 : rc;
}
