// accuracy of v_rcp_f64 / v_rsq_f64 on this device, and after one / two Newton steps (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], t;
  asm volatile("v_rcp_f64 %0, %1" : "=v"(t) : "v"(v));
  r0[i] = t;
  double e = __builtin_fma(-v, t, 1.0);
  t = __builtin_fma(t, e, t);
  r1[i] = t;
  e = __builtin_fma(-v, t, 1.0);
  t = __builtin_fma(t, e, t);
  r2[i] = t;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
  double *x, *r[3];
  (void)hipMalloc(&x, n * 8); for (auto& p : r) (void)hipMalloc(&p, n * 8);
  (void)hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r[0], r[1], r[2], n);
  (void)hipDeviceSynchronize();
  std::vector<double> o(n);
  for (int j = 0; j < 3; ++j) {
    (void)hipMemcpy(o.data(), r[j], n * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) { double rel = std::fabs((double)((long double)o[i] * (long double)h[i] - 1.0L)); if (rel > worst) worst = rel; }
    printf("v_rcp_f64 + %d Newton steps: max |x * r - 1| = %.3e\n", j, worst);
  }
  return 0;
}
