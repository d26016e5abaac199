// Which SIMD does wavefront w of a workgroup land on?  (tuning aid: tools/scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  __shared__ double pad[7000];                       // ~56 KB like the sweep-1 kernel
  pad[threadIdx.x] = 1.0;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = id;
    out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc;
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (pad[threadIdx.x] == 2.0) out[0] = 0;
}
int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 512;
  int waves = threads / 64;
  unsigned* d;
  hipMalloc(&d, blocks * waves * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * waves * 2);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  for (int b = 0; b < blocks; ++b) {
    if (b >= 6 && b % 37 != 0) continue;
    printf("block %3d:", b);
    for (int w = 0; w < waves; ++w) {
      unsigned id = h[(b * waves + w) * 2], x = h[(b * waves + w) * 2 + 1];
      printf(" [x%u cu%2u s%u w%u]", x & 0xf, (id >> 8) & 0xf, (id >> 4) & 0x3, id & 0xf);
    }
    printf("  se%u sh%u\n", (h[b * waves * 2] >> 13) & 0x7, (h[b * waves * 2] >> 12) & 1);
  }
  // how many distinct (xcc, se, sh, cu) host more than one block
  std::vector<int> cnt(1 << 16, 0);
  for (int b = 0; b < blocks; ++b) { unsigned id = h[b * waves * 2], x = h[b * waves * 2 + 1] & 0xf; cnt[(x << 12) | ((id >> 8) & 0xff) << 0 | (((id >> 12) & 0xf) << 8)]++; }
  int hist[8] = {0};
  for (int v : cnt) if (v) hist[v < 7 ? v : 7]++;
  printf("CUs hosting 1,2,3,4 blocks: %d %d %d %d\n", hist[1], hist[2], hist[3], hist[4]);
  return 0;
}
