// read_bw.hip -- what a pure read stream reaches on this GPU, in the access pattern of the weight-gradient kernels (a workgroup
// reads whole 8 KB stash tiles, tiles strided by the grid) and as one linear stream per workgroup.
// hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// tile = 2048 floats (8 KB); 256 threads read it as 2 x 16 B each
__global__ __launch_bounds__(256) void k_tiles(const float* __restrict__ a, long ntiles, float* __restrict__ out) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const f32x4* p = reinterpret_cast<const f32x4*>(a + t * 2048);
    s += p[threadIdx.x]; s += p[threadIdx.x + 256];
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ a, long n4, long span, float* __restrict__ out) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const f32x4* p = reinterpret_cast<const f32x4*>(a);
  const long u0 = (long)blockIdx.x * span, u1 = u0 + span < n4 ? u0 + span : n4;
  for (long u = u0 + threadIdx.x; u < u1; u += 1024) {
    s += p[u];
    if (u + 256 < u1) s += p[u + 256];
    if (u + 512 < u1) s += p[u + 512];
    if (u + 768 < u1) s += p[u + 768];
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
int main() {
  const long n = 1L << 29;      // 2 GiB of floats
  float *a, *o; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&o, 4)); CK(hipMemset(a, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* nm, auto f) {
    f(); CK(hipDeviceSynchronize()); float best = 1e9f;
    for (int i = 0; i < 5; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-34s %7.3f ms %8.1f GB/s\n", nm, best, n * 4.0 / 1e9 / (best * 1e-3));
  };
  for (int g : {256, 512, 1024, 2048, 4096}) { char nm[64]; snprintf(nm, 64, "tiles strided by grid=%d", g); run(nm, [&] { hipLaunchKernelGGL(k_tiles, dim3(g), dim3(256), 0, 0, a, n / 2048, o); }); }
  for (int g : {1024, 4096, 16384}) { const long n4 = n / 4; const long span = ((n4 + g - 1) / g + 1023) / 1024 * 1024; char nm[64]; snprintf(nm, 64, "linear per workgroup, grid=%d", g); run(nm, [&] { hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, a, n4, span, o); }); }
  return 0;
}
