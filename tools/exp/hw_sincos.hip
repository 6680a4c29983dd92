// accuracy probe: v_sin_f32 / v_cos_f32 (input in revolutions) after a 2-fma reduction, against fp64
#include "../../nif_amd/csrc/nif_internal.h"
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* a, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nif_sincosf(a[i], &s[i], &c[i]);
}
int main() {
  const int n = 1 << 22;
  for (float range : {0.5f, 3.2f, 30.f, 100.f, 1000.f, 1e5f, 1e6f, 4e6f}) {
    std::vector<float> h(n), hs(n), hc(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = range * (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    float *a, *s, *c;
    hipMalloc(&a, n * 4); hipMalloc(&s, n * 4); hipMalloc(&c, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, s, c, n);
    hipMemcpy(hs.data(), s, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    double ms = 0, mc = 0, rs = 0, rc = 0;
    for (int i = 0; i < n; ++i) {
      double es = fabs((double)hs[i] - sin((double)h[i])), ec = fabs((double)hc[i] - cos((double)h[i]));
      if (es > ms) ms = es; if (ec > mc) mc = ec; rs += es * es; rc += ec * ec;
    }
    printf("range %-7g max|err| sin %.3e cos %.3e   rms sin %.3e cos %.3e\n", range, ms, mc, sqrt(rs / n), sqrt(rc / n));
    hipFree(a); hipFree(s); hipFree(c);
  }
  return 0;
}
