// Micro-benchmark: issue cost of v_mfma_f64_16x16x4_f64 (cycles per instruction per wave) for
// dependent (same accumulator) and independent chains, and with 1..8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64 && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k(long long* out, double* sink, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001;
  d4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = d4{0.1 * i, 0.2, 0.3, 0.4};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int ch = 0; ch < CHAINS; ++ch) c[ch] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[ch], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < CHAINS; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS>
void run(const char* name) {
  const int iters = 200;
  for (int waves : {1, 4, 8, 16}) {
    const int blocks = 256, threads = 64 * waves;
    long long* out; double* sink;
    (void)hipMalloc(&out, sizeof(long long) * blocks * waves);
    (void)hipMalloc(&sink, sizeof(double) * blocks * threads);
    k<CHAINS><<<blocks, threads>>>(out, sink, iters);
    k<CHAINS><<<blocks, threads>>>(out, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * waves);
    (void)hipMemcpy(h.data(), out, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-22s waves/CU=%2d : %.1f cycles per MFMA per wave (median)\n", name, waves,
           (double)h[h.size() / 2] / (iters * 16.0 * CHAINS));
    (void)hipFree(out); (void)hipFree(sink);
  }
}

int main() {
  run<1>("mfma_f64_16x16x4 dep");
  run<2>("mfma_f64_16x16x4 x2");
  run<4>("mfma_f64_16x16x4 x4");
  return 0;
}
