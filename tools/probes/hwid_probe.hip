// hwid_probe.hip -- where do the wavefronts of co-resident workgroups land?  512 workgroups of 256 threads with 72 KB of
// LDS each (two per CU); every wavefront records HW_ID and XCC_ID.   hipcc --offload-arch=gfx950 -O2 -o hwid_probe hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  extern __shared__ double sm[];
  const int wave = threadIdx.x >> 6;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double acc = sm[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 1e-9;     // stay resident while the rest launches
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + wave) * 2] = hw;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc + (acc == 0.5 ? 1 : 0);
  }
}
int main() {
  const int B = 512;
  unsigned* d;
  hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipLaunchKernelGGL(probe, dim3(B), dim3(256), 72 * 1024, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(B * 4 * 2);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  // key: (xcc, se, sh, cu) -> list of (block, wave, simd, tg, waveid)
  std::map<unsigned, std::vector<std::vector<int>>> cus;
  int simd_eq_wave = 0;
  for (int b = 0; b < B; ++b)
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 15,
                wid = hw & 15;
      simd_eq_wave += simd == w;
      cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({b, w, simd, tg, wid});
    }
  printf("wavefronts with simd == wave index: %d of %d; distinct CUs: %zu\n", simd_eq_wave, B * 4, cus.size());
  int shown = 0;
  for (auto& kv : cus) {
    if (shown++ >= 6) break;
    printf("cu key %05x:", kv.first);
    for (auto& e : kv.second) printf("  [b%d w%d simd%d tg%d id%d]", e[0], e[1], e[2], e[3], e[4]);
    printf("\n");
  }
  // histogram: per CU, number of workgroups, and whether the two workgroups' wave 0 share a SIMD
  int two = 0, same0 = 0, tgdiff = 0;
  for (auto& kv : cus) {
    std::map<int, std::vector<int>> byb;
    for (auto& e : kv.second) if (e[1] == 0) byb[e[0]] = e;
    if (byb.size() == 2) {
      ++two;
      auto it = byb.begin(); auto a = it->second; ++it; auto c = it->second;
      same0 += a[2] == c[2];
      tgdiff += a[3] != c[3];
    }
  }
  printf("CUs with two workgroups: %d; wave 0 of both on the same SIMD: %d; TG_ID differs: %d\n", two, same0, tgdiff);
  // block-index relation of co-resident pairs
  std::map<int, int> delta;
  for (auto& kv : cus) {
    std::vector<int> bs;
    for (auto& e : kv.second) if (e[1] == 0) bs.push_back(e[0]);
    if (bs.size() == 2) delta[abs(bs[0] - bs[1])]++;
  }
  for (auto& kv : delta) printf("  |b1 - b2| = %d : %d CUs\n", kv.first, kv.second);
  return 0;
}
