// per_device.hpp -- the little host-side state the library keeps, keyed so that it is safe with several
// devices, several caller streams and several host threads in one process:
//   * LdsGrant: "largest dynamic-LDS size granted to this kernel" per DEVICE (hipFuncSetAttribute acts on the
//     current device's copy of the function);
//   * fork_join_for(stream): the helper stream + event pair of svae_lds_estep_f64's two-kernel forward pass,
//     one per (device, caller stream).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>

namespace svae {

constexpr int MAX_DEVICES = 64;

struct LdsGrant {
  std::atomic<long> granted[MAX_DEVICES];   // static storage: zero-initialised
  std::mutex mu;
  // make sure `kern` may be launched with `bytes` of dynamic LDS on the CURRENT device; true on success
  bool ensure(const void* kern, long bytes) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) return false;
    if (bytes <= granted[d].load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lock(mu);
    if (bytes <= granted[d].load(std::memory_order_relaxed)) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    granted[d].store(bytes, std::memory_order_release);
    return true;
  }
};

struct ForkJoin {
  hipStream_t aux;
  hipEvent_t fork, join;
};

// the helper stream / events for work forked off `caller` on the current device (created on first use; the
// table only grows -- a process uses a handful of streams)
inline bool fork_join_for(hipStream_t caller, ForkJoin* out) {
  struct Entry { int device; hipStream_t caller; ForkJoin fj; };
  static std::mutex mu;
  static std::vector<Entry> table;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  for (const Entry& e : table)
    if (e.device == d && e.caller == caller) { *out = e.fj; return true; }
  ForkJoin fj{nullptr, nullptr, nullptr};
  if (hipStreamCreateWithFlags(&fj.aux, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&fj.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&fj.join, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  table.push_back(Entry{d, caller, fj});
  *out = fj;
  return true;
}

}  // namespace svae
