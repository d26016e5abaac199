// ipc_allreduce.hip -- one-shot all-reduce (sum, fp64) of a SMALL buffer over IPC-mapped peer mailboxes, for MI355X
// nodes (gfx950, xGMI).
//
// What it is for: the exchange step of the structured E-step under data parallelism
// (/root/reference/svae/svae.py:33-34 consumes the batch-summed statistics): ONE all-reduce of the packed 4n^2+n+2
// doubles (3.3 KB at n = 10) per natural-gradient step, next to a 0.14 ms kernel.  RCCL serves it with its small-message
// protocol (a launch + a ring / tree of flag-carrying stores, 15-30 us); this kernel is the latency-minimal form of the
// same idea with the reduction's arithmetic fused in: every rank STORES its buffer into every peer's mailbox (one hop
// over xGMI, point to point -- no ring), then sums what arrived in ITS mailbox in rank order.  One launch, one hop, no
// host involvement, bit-identical results on every rank (same order of summation).
//
// Protocol (the data-tagged granules of gmm_meanfield.hip, at system scope):
//   * mailbox of rank r (fine-grained device memory, IPC-mapped into every peer): [parity 2][source rank W][2 cap] words
//     of 8 bytes, word = {tag = epoch (32 bits) | one half of a double (32 bits)} -- the data is the flag; `cap` is the
//     mailbox's CAPACITY in doubles, fixed at its creation: the layout does not move with the length n <= cap of a
//     call (round 4 laid the sets out with the call's n: two consecutive calls of different lengths put their parity
//     sets on overlapping words, and a rank one call ahead could overwrite words a slower peer had not read yet);
//   * rank r writes word (parity, r, g) of EVERY mailbox with a relaxed system-scope atomic store (write-through);
//   * then polls the W x 2n words of its own mailbox until every tag equals the epoch, adds the values in rank order;
//   * epochs count calls (never 0); the parity double-buffers: a rank can be at most one call ahead of the slowest
//     (it needs that rank's words of the current epoch to finish it).
// Every spin is bounded (spin_limit polls per word, each a few hundred ns; 0 = 2^28, minutes of rank skew); an element
// whose words never arrive sets *info = -78 and comes out as NaN: a timeout cannot pass for a sum (the
// reduction is in place over the statistics a natural-gradient step consumes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svae_hip.h"

namespace svae {

constexpr int IPC_MAX_WORLD = 16;
constexpr unsigned IPC_SPIN_LIMIT = 1u << 28;

struct IpcArgs {
  int n, cap, rank, world;
  unsigned epoch, spin_limit;
  const double* in;
  double* out;
  unsigned long long* box[IPC_MAX_WORLD];     // mailbox of every rank as mapped into THIS process
  int32_t* info;
};

__global__ __launch_bounds__(256) void ipc_allreduce_kernel(const IpcArgs a) {
  const int n = a.n, W = a.world;
  const size_t slot = (size_t)2 * a.cap;                       // words per source rank: the mailbox's capacity, not n
  const size_t set = (size_t)(a.epoch & 1u) * W * slot;        // parity set
  const unsigned long long tag = (unsigned long long)a.epoch << 32;
  // ---- publish: my 2n words into slot `rank` of every mailbox ------------------------------------------------------
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(a.in[i]);
    const unsigned long long lo = tag | (bits & 0xffffffffull), hi = tag | (bits >> 32);
    for (int q = 0; q < W; ++q) {
      unsigned long long* w = a.box[q] + set + (size_t)a.rank * slot + 2 * i;
      __hip_atomic_store(w, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(w + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // ---- collect: my mailbox, rank order ---------------------------------------------------------------------------------
  const unsigned long long* mine = a.box[a.rank] + set;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int q = 0; q < W; ++q) {
      const unsigned long long* w = mine + (size_t)q * slot + 2 * i;
      unsigned long long lo = 0, hi = 0;
      unsigned spins = 0;
      bool arrived = true;
      for (;;) {
        lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        hi = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((lo >> 32) == a.epoch && (hi >> 32) == a.epoch) break;
        if (++spins > a.spin_limit) { atomicMin(a.info, -78); arrived = false; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      s += arrived ? __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull))) : __builtin_nan("");
    }
    a.out[i] = s;                                              // NaN if any peer's words never arrived
  }
}

}  // namespace svae

extern "C" size_t svae_ipc_mailbox_bytes(int capacity, int world) {
  if (capacity < 1 || world < 1 || world > svae::IPC_MAX_WORLD) return 0;
  return (size_t)2 * world * 2 * capacity * sizeof(unsigned long long);
}

extern "C" int svae_ipc_allreduce_f64(int n, int capacity, int rank, int world, unsigned epoch, unsigned spin_limit,
                                      const double* in, double* out, void* const* mailboxes, int32_t* info,
                                      void* stream) {
  if (n < 1) return -1;
  if (capacity < n) return -2;
  if (world < 1 || world > svae::IPC_MAX_WORLD) return -4;
  if (rank < 0 || rank >= world) return -3;
  if (epoch == 0) return -5;
  if (!in) return -7;
  if (!out) return -8;
  if (!mailboxes) return -9;
  if (!info) return -10;
  svae::IpcArgs a;
  a.n = n; a.cap = capacity; a.rank = rank; a.world = world; a.epoch = epoch; a.in = in; a.out = out; a.info = info;
  a.spin_limit = spin_limit ? spin_limit : svae::IPC_SPIN_LIMIT;
  for (int q = 0; q < svae::IPC_MAX_WORLD; ++q) a.box[q] = q < world ? (unsigned long long*)mailboxes[q] : nullptr;
  for (int q = 0; q < world; ++q) if (!a.box[q]) return -9;
  const int grid = (n + 255) / 256 > 8 ? 8 : (n + 255) / 256;      // a few workgroups: the buffer is small by contract
  hipLaunchKernelGGL(svae::ipc_allreduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
