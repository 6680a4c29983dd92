// coissue_probe.hip (r6) -- does a SIMD of gfx950 overlap one wave's VALU stream with another wave's MFMA stream?
// One 1024-thread workgroup per CU (4 waves per SIMD, 128 registers each, as k_snet6): waves 0-7 run a vector block like the
// producers' (FMA chains + v_sin + f16 splits), waves 8-15 a chunk step like the matrix phases (ds_read_b128 A operands from LDS +
// three-deep v_mfma_f32_16x16x32_f16 chains on four accumulators).  Modes: V alone, M alone, both, and V on all 16 waves / M on all 16.
// build: hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip ; run: ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float vblock(float x, int iters) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {      // per element: bias + scale + fract + sin (quarter rate) + 4 FMAs: ~ the producers' mix
      float t = a[i] * 0.15915494f + 0.25f;
      t = __builtin_amdgcn_fractf(t);
      float s = __builtin_amdgcn_sinf(t);
      s = fmaf(s, 1.0001f, 0.5f); s = fmaf(s, 0.999f, -0.5f); s = fmaf(s, a[(i + 1) & 15], 0.1f); s = fmaf(s, 0.5f, a[i]);
      a[i] = s;
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i];
  return r;
}
__device__ __forceinline__ float mblock(const f16x8* lds, int lane, int iters) {
  f32x4 T[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f16x8 b0, b1;
#pragma unroll
  for (int t = 0; t < 8; ++t) { b0[t] = (_Float16)(0.001f * lane); b1[t] = (_Float16)(0.002f * t); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        const f16x8 a0 = lds[((ks * 4 + ob) * 2 + 0) * 64 + lane], a1 = lds[((ks * 4 + ob) * 2 + 1) * 64 + lane];
        T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, T[ob], 0, 0, 0);
        T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, T[ob], 0, 0, 0);
        T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, T[ob], 0, 0, 0);
      }
    }
  }
  return T[0][0] + T[1][1] + T[2][2] + T[3][3];
}
// vmask / mmask: bit (w >> 2) of the wave's slot on its SIMD (w = 0..15: slot 0..3) says whether the wave runs that role
__global__ __launch_bounds__(1024) void probe(float* out, int vmask, int mmask, int vit, int mit, int prio) {
  extern __shared__ f16x8 lds[];
  for (int i = threadIdx.x; i < 16 * 64; i += 1024) {
    f16x8 q;
#pragma unroll
    for (int t = 0; t < 8; ++t) q[t] = (_Float16)(0.01f * ((i + t) & 31));
    lds[i] = q;
  }
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = w >> 2;
  float r = 0.f;
  if ((vmask >> slot) & 1) {
    if (prio == 2) __builtin_amdgcn_s_setprio(3);
    r = vblock(0.001f * threadIdx.x, vit);
  }
  else if ((mmask >> slot) & 1) {
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    r = mblock(lds, lane, mit);
  }
  if (r == 123.456f) out[blockIdx.x * 1024 + threadIdx.x] = r;
}
int main() {
  float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int vit = 400, mit = 1100;
  struct { const char* name; int vm, mm; } modes[] = {
      {"V on 2 waves / SIMD", 0x3, 0x0}, {"M on 2 waves / SIMD", 0x0, 0xC}, {"V on 2 + M on 2", 0x3, 0xC},
      {"V on 1", 0x1, 0x0}, {"M on 1", 0x0, 0x4}, {"V on 1 + M on 1", 0x1, 0x4}, {"V on 1 + M on 2", 0x1, 0xC}, {"V on 2 + M on 1", 0x3, 0x4},
      {"M on slots 0,1 + V on 2,3", 0xC, 0x3}, {"V on 4", 0xF, 0x0}, {"M on 4", 0x0, 0xF}};
  for (int prio = 0; prio < 3; ++prio)
  for (auto& m : modes) {
    if (prio && !(m.vm && m.mm)) continue;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 16 * 1024, 0, out, m.vm, m.mm, vit, mit, prio);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-24s prio %d (1: M waves s_setprio 3, 2: V waves) %8.3f ms\n", m.name, prio, best);
  }
  return 0;
}
