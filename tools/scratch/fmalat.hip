// Dependent-chain latency of v_fmac_f64_dpp / v_fma_f64: cycles per instruction with A independent accumulators
// (tuning aid: tools/scratch).  One wavefront per SIMD (grid = 1 block of 64 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int A, bool DPP>
__global__ void chain(double* out, long long* cyc, int iters) {
  double acc[A];
  for (int i = 0; i < A; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double b = 1.0000001, s = 0.5 + threadIdx.x * 1e-6;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / A; ++r) {
#pragma unroll
      for (int i = 0; i < A; ++i) {
        if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(s), "v"(b));
        else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(s), "v"(b));
      }
    }
  }
  long long t1 = clock64();
  double r = 0; for (int i = 0; i < A; ++i) r += acc[i];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int A, bool DPP> void run(double* d, long long* c) {
  int iters = 2000;
  hipLaunchKernelGGL((chain<A, DPP>), dim3(1), dim3(64), 0, 0, d, c, iters);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%s A=%2d: %.2f clock64 ticks per instruction\n", DPP ? "dpp " : "fmac", A, (double)h / (iters * 24.0));
}
int main() {
  double* d; long long* c;
  (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 8);
  run<1, true>(d, c); run<2, true>(d, c); run<3, true>(d, c); run<4, true>(d, c); run<6, true>(d, c); run<12, true>(d, c);
  run<1, false>(d, c); run<2, false>(d, c); run<3, false>(d, c); run<4, false>(d, c); run<6, false>(d, c);
  // clock64 frequency vs shader clock: time a known spin
  return 0;
}
