// rcp_seed.hip -- accuracy of the v_rcp_f64 seed and of the refinements built on it (dpp.hpp: rcp_nr).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/rcp_seed.hip -o /tmp/rcp_seed && /tmp/rcp_seed
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = x[i];
  const double r0 = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r0, 1.0);
  const double r_newton1 = __builtin_fma(r0, e, r0);
  double e2 = __builtin_fma(-p, r_newton1, 1.0);
  const double r_newton2 = __builtin_fma(r_newton1, e2, r_newton1);
  const double ec = __builtin_fma(e, e, e);
  const double r_cubic = __builtin_fma(r0, ec, r0);
  out[4 * i] = r0; out[4 * i + 1] = r_newton1; out[4 * i + 2] = r_newton2; out[4 * i + 3] = r_cubic;
}

int main() {
  const int n = 1 << 22;
  double* hx = (double*)malloc(n * sizeof(double));
  srand(1);
  for (int i = 0; i < n; ++i) {
    const double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9;
    hx[i] = ldexp(m, (rand() % 200) - 100) * ((i & 1) ? -1.0 : 1.0);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * sizeof(double)); hipMalloc(&dout, 4 * n * sizeof(double));
  hipMemcpy(dx, hx, n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  double* ho = (double*)malloc(4 * n * sizeof(double));
  hipMemcpy(ho, dout, 4 * n * sizeof(double), hipMemcpyDeviceToHost);
  double worst[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)hx[i];
    for (int j = 0; j < 4; ++j) {
      const double rel = (double)fabsl(((long double)ho[4 * i + j] - t) / t);
      if (rel > worst[j]) worst[j] = rel;
    }
  }
  printf("max relative error over %d inputs: seed %.3e (2^%.1f)  one Newton step %.3e  two Newton steps %.3e (%.2f ulp)  "
         "one cubic step %.3e (%.2f ulp)\n", n, worst[0], log2(worst[0]), worst[1], worst[2], worst[2] / 1.1102230246251565e-16,
         worst[3], worst[3] / 1.1102230246251565e-16);
  return 0;
}
