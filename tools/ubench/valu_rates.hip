// Micro-benchmark: per-wave issue cost (cycles/instruction, via s_memtime) of the instructions the
// E-step kernel is made of, and how it scales with the number of co-resident waves.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void k(long long* out, double* sink, int iters, int active_lanes) {
  if ((int)threadIdx.x % 64 >= active_lanes) return;
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
         a6 = a0 + 6, a7 = a0 + 7;
  double b = 1.0000001, c = 1e-9;
  int m = threadIdx.x & 1;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {  // independent v_fma_f64 (8 chains)
      REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (OP == 1) {  // dependent v_fma_f64 chain
      REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
    } else if (OP == 2) {  // independent v_fmac_f64_dpp row_newbcast (8 chains)
      REP8(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                        "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                        "v_fmac_f64_dpp %4, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                        "v_fmac_f64_dpp %6, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (OP == 3) {  // dependent v_fmac_f64_dpp chain (same accumulator)
      REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b), "v"(c));)
    } else if (OP == 4) {  // v_cndmask_b32 (independent)
      float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
      REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                        "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(m) : "vcc");)
      a0 += f0 + f1 + f2 + f3;
    } else if (OP == 5) {  // v_mov_b64
      REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %4\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 6) {  // s_nop 0
      REP64(asm volatile("s_nop 0");)
    } else if (OP == 7) {  // v_add_f64 independent
      REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                        "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (OP == 8) {  // v_fma_f32 independent
      float f0 = a0, f1 = a1, f2 = a2, f3 = a3, g = 1.0001f, h = 1e-6f;
      REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                        "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h));)
      a0 += f0 + f1 + f2 + f3;
    } else if (OP == 9) {  // mov_b64_dpp + fma pair (the compiler-only form)
      double t;
      REP8(asm volatile("v_mov_b64_dpp %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, %8, %10, %0\n"
                        "v_mov_b64_dpp %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fma_f64 %1, %8, %10, %1\n"
                        "v_mov_b64_dpp %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fma_f64 %2, %8, %10, %2\n"
                        "v_mov_b64_dpp %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fma_f64 %3, %8, %10, %3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t) : "v"(b), "v"(c));)
    } else if (OP == 10) {  // v_rcp_f64 independent
      REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name, int blocks, int threads, int active) {
  const int iters = 200;
  long long* d; double* s;
  hipMalloc(&d, sizeof(long long) * blocks * (threads / 64));
  hipMalloc(&s, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(d, s, 10, active);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(d, s, iters, active);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  double per = 64.0 * iters;
  printf("%-22s blocks=%5d thr=%4d lanes=%2d : memtime/instr med %.2f max %.2f | wall %.1f us -> %.2f ns/instr/wave\n",
         name, blocks, threads, active, h[h.size() / 2] / per, h.back() / per, ms * 1e3, ms * 1e6 / per);
  hipFree(d); hipFree(s);
}

#define RUNALL(OP, name)                                                          \
  run<OP>(name, 1, 64, 64); run<OP>(name, 1, 64, 16); run<OP>(name, 1, 256, 64);   \
  run<OP>(name, 128, 64, 64); run<OP>(name, 256, 64, 64); run<OP>(name, 512, 64, 64); \
  run<OP>(name, 512, 64, 16); run<OP>(name, 1024, 64, 64); run<OP>(name, 2048, 64, 64); run<OP>(name, 256, 256, 64);

int main() {
  RUNALL(0, "fma_f64 indep")
  RUNALL(1, "fma_f64 dep")
  RUNALL(2, "fmac_f64_dpp indep")
  RUNALL(3, "fmac_f64_dpp dep")
  RUNALL(9, "mov_dpp+fma pair(x2)")
  RUNALL(4, "cndmask_b32")
  RUNALL(5, "mov_b64")
  RUNALL(7, "add_f64 indep")
  RUNALL(8, "fma_f32 indep")
  RUNALL(6, "s_nop 0")
  RUNALL(10, "rcp_f64")
  return 0;
}
