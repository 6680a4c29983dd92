// l2w_probe.hip -- write-stream experiments for model_lr_to_w (w[a][s] = sum_k z[a][k] Wh[k][s] + bh[s], po odd):
//   A  pure aligned-window stores in the shipped kernel's pattern (4 KB per workgroup per row, rows 4*po bytes apart)
//   B  flat: the output as ONE array of 16-byte units, a workgroup streams a contiguous span; (a, s) by division; Wh/bh by
//      dword loads (L2 / L1)
//   C  flat with Wh / bh staged in LDS (r = 1)
//   M  hipMemsetAsync of the same bytes
// build: hipcc --offload-arch=gfx950 -O3 -o l2w_probe l2w_probe.hip ; run: ./l2w_probe [B] [po]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void kA(float* __restrict__ w, long po, long B) {
  const long t4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (t4 - 3 >= po) return;
  for (long a = blockIdx.y; a < B; a += gridDim.y) {
    const int m = (int)((a * po) & 3);
    const long s0 = t4 - m;
    float* dst = w + a * po + s0;
    f32x4 v = {1.f, 2.f, 3.f, (float)a};
    if (s0 >= 0 && s0 + 4 <= po) *reinterpret_cast<f32x4*>(dst) = v;
  }
}
// flat: unit u covers floats 4u..4u+3 of the [B*po] array
template <int UNR>
__global__ __launch_bounds__(256) void kB(const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ z,
                                         float* __restrict__ w, long po, long nunits, long span) {
  const long u0 = (long)blockIdx.x * span;
  const long u1 = u0 + span < nunits ? u0 + span : nunits;
  for (long ub = u0 + threadIdx.x; ub < u1; ub += 256 * UNR) {
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
      const long u = ub + 256 * i;
      if (u < u1) {
        const long e = 4 * u;
        long a = e / po; long s = e - a * po;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = fmaf(z[a], Wh[s], bh[s]);
          if (++s == po) { s = 0; ++a; }
        }
        *reinterpret_cast<f32x4*>(w + e) = v;
      }
    }
  }
}
template <int UNR>
__global__ __launch_bounds__(1024) void kC(const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ z,
                                          float* __restrict__ w, int po, long nunits, long span) {
  extern __shared__ float sm[];
  float* sW = sm; float* sB = sm + po;
  for (int i = threadIdx.x; i < po; i += 1024) { sW[i] = Wh[i]; sB[i] = bh[i]; }
  __syncthreads();
  const long u0 = (long)blockIdx.x * span;
  const long u1 = u0 + span < nunits ? u0 + span : nunits;
  for (long ub = u0 + threadIdx.x; ub < u1; ub += 1024 * UNR) {
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
      const long u = ub + 1024 * i;
      if (u < u1) {
        const long e = 4 * u;
        long a = e / po; int s = (int)(e - a * po);
        float za = z[a];
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = fmaf(za, sW[s], sB[s]);
          if (++s == po) { s = 0; ++a; za = z[a < 0 ? 0 : a]; }
        }
        *reinterpret_cast<f32x4*>(w + e) = v;
      }
    }
  }
}

int main(int argc, char** argv) {
  const long B = argc > 1 ? atol(argv[1]) : 131072;
  const long po = argc > 2 ? atol(argv[2]) : 16833;
  const long n = B * po;
  float *w, *Wh, *bh, *z;
  CK(hipMalloc(&w, sizeof(float) * (size_t)(n + 64)));
  CK(hipMalloc(&Wh, sizeof(float) * po)); CK(hipMalloc(&bh, sizeof(float) * po)); CK(hipMalloc(&z, sizeof(float) * (B + 1)));
  CK(hipMemset(Wh, 0, sizeof(float) * po)); CK(hipMemset(bh, 0, sizeof(float) * po)); CK(hipMemset(z, 0, sizeof(float) * (B + 1)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double gb = (double)n * 4 / 1e9;
  auto timeit = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, best, gb / (best * 1e-3));
  };
  timeit("M hipMemsetAsync", [&] { CK(hipMemsetAsync(w, 0, sizeof(float) * (size_t)n)); });
  for (int gy : {512, 2048, 8192}) {
    char nm[64]; snprintf(nm, 64, "A window stores gy=%d", gy);
    timeit(nm, [&] { hipLaunchKernelGGL(kA, dim3((unsigned)((po + 3 + 1023) / 1024), gy), dim3(256), 0, 0, w, po, B); });
  }
  const long nunits = n / 4;
  for (int nb : {2048, 8192, 32768}) {
    const long span = ((nunits + nb - 1) / nb + 255) / 256 * 256;
    char nm[64];
    snprintf(nm, 64, "B flat L2 loads nb=%d u1", nb);
    timeit(nm, [&] { hipLaunchKernelGGL((kB<1>), dim3(nb), dim3(256), 0, 0, Wh, bh, z, w, po, nunits, span); });
    snprintf(nm, 64, "B flat L2 loads nb=%d u4", nb);
    timeit(nm, [&] { hipLaunchKernelGGL((kB<4>), dim3(nb), dim3(256), 0, 0, Wh, bh, z, w, po, nunits, span); });
  }
  const size_t shm = sizeof(float) * 2 * po;
  CK(hipFuncSetAttribute((const void*)kC<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  CK(hipFuncSetAttribute((const void*)kC<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int nb : {256, 1024, 4096}) {
    const long span = ((nunits + nb - 1) / nb + 1023) / 1024 * 1024;
    char nm[64];
    snprintf(nm, 64, "C flat LDS nb=%d u1", nb);
    timeit(nm, [&] { hipLaunchKernelGGL((kC<1>), dim3(nb), dim3(1024), shm, 0, Wh, bh, z, w, (int)po, nunits, span); });
    snprintf(nm, 64, "C flat LDS nb=%d u4", nb);
    timeit(nm, [&] { hipLaunchKernelGGL((kC<4>), dim3(nb), dim3(1024), shm, 0, Wh, bh, z, w, (int)po, nunits, span); });
  }
  return 0;
}
