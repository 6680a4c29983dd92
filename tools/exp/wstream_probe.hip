// wstream_probe.hip (r6) -- pure 16-byte store streams of 8.8 GB in the shapes k_latent_to_w_flat can take (1024-thread workgroups,
// 135 KB of LDS = one workgroup per CU, no arithmetic):
//   S  span: workgroup b writes ONE contiguous span (the r2-r6 kernel)
//   I  interleaved: workgroup b writes the 16 KB pieces b, b + nblk, b + 2 nblk, ... (all CUs sweep one window, as a grid-stride fill does)
//   M  hipMemsetAsync of the same bytes
// build: hipcc --offload-arch=gfx950 -O3 -o wstream_probe wstream_probe.hip ; run: ./wstream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// U: span form with UNR 16-byte stores per thread and iteration: UNR x 16 KB per workgroup pass (PIECE = true: each THREAD's UNR stores are
// adjacent = 64 contiguous bytes per lane; false: UNR passes of 16 KB each)
template <int UNR, bool PIECE>
__global__ __launch_bounds__(1024) void kU(float* __restrict__ w, long nunits, long span) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0) sm[0] = 1.f;
  __syncthreads();
  const f32x4 v = {sm[0], 2.f, 3.f, 4.f};
  const long u0 = (long)blockIdx.x * span, u1 = u0 + span < nunits ? u0 + span : nunits;
  for (long ub = u0; ub < u1; ub += 1024 * UNR) {
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
      const long u = PIECE ? ub + threadIdx.x * UNR + i : ub + 1024 * i + threadIdx.x;
      if (u < u1) *reinterpret_cast<f32x4*>(w + 4 * u) = v;
    }
  }
}
template <bool INTER>
__global__ __launch_bounds__(1024) void kS(float* __restrict__ w, long nunits, long span, long nblk) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0) sm[0] = 1.f;
  __syncthreads();
  const f32x4 v = {sm[0], 2.f, 3.f, 4.f};
  if (INTER) {
    for (long u = (long)blockIdx.x * 1024 + threadIdx.x; u < nunits; u += nblk * 1024) *reinterpret_cast<f32x4*>(w + 4 * u) = v;
  } else {
    const long u0 = (long)blockIdx.x * span, u1 = u0 + span < nunits ? u0 + span : nunits;
    for (long u = u0 + threadIdx.x; u < u1; u += 1024) *reinterpret_cast<f32x4*>(w + 4 * u) = v;
  }
}
int main() {
  const long B = 1 << 17, po = 16833, n = B * po, nunits = n / 4;
  float* w; CK(hipMalloc(&w, n * 4 + 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t shm = 135 * 1024;
  CK(hipFuncSetAttribute((const void*)kS<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  CK(hipFuncSetAttribute((const void*)kS<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int mode = 0; mode < 7; ++mode) {
    const long nblks[] = {4096, 256, 4096, 256, 1024, 512, 0};
    const long nblk = nblks[mode];
    const long span = nblk ? ((nunits + nblk - 1) / nblk + 1023) / 1024 * 1024 : 0;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) {
        if (mode == 6) CK(hipMemsetAsync(w, 0, n * 4));
        else if (mode < 2) hipLaunchKernelGGL(kS<false>, dim3((nunits + span - 1) / span), dim3(1024), shm, 0, w, nunits, span, nblk);
        else hipLaunchKernelGGL(kS<true>, dim3(nblk), dim3(1024), shm, 0, w, nunits, span, nblk);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      if (rep && ms < best) best = ms;
    }
    printf("%s nblk %ld: %.3f ms  %.0f GB/s\n", mode == 6 ? "memset" : (mode < 2 ? "span" : "interleaved"), nblk, best, n * 4.0 / best / 1e6);
  }
  {
    const long nblk = 4096, span = ((nunits + nblk - 1) / nblk + 4095) / 4096 * 4096, nb = (nunits + span - 1) / span;
#define RUNU(UNR_, PIECE_, SHM_, NAME_)                                                                                   \
    {                                                                                                                     \
      CK(hipFuncSetAttribute((const void*)kU<UNR_, PIECE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));      \
      float best = 1e9f;                                                                                                  \
      for (int rep = 0; rep < 4; ++rep) {                                                                                 \
        CK(hipEventRecord(e0));                                                                                           \
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((kU<UNR_, PIECE_>), dim3(nb), dim3(1024), SHM_, 0, w, nunits, span); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                              \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;                                                          \
        if (rep && ms < best) best = ms;                                                                                  \
      }                                                                                                                   \
      printf("%s: %.3f ms  %.0f GB/s\n", NAME_, best, n * 4.0 / best / 1e6);                                              \
    }
    RUNU(1, false, shm, "span unr 1, 135 KB LDS")
    RUNU(2, false, shm, "span unr 2 (2 passes), 135 KB LDS")
    RUNU(4, false, shm, "span unr 4 (4 passes), 135 KB LDS")
    RUNU(2, true, shm, "span 32 B per lane, 135 KB LDS")
    RUNU(4, true, shm, "span 64 B per lane, 135 KB LDS")
    RUNU(1, false, 64, "span unr 1, no LDS (2 workgroups per CU)")
    RUNU(4, false, 64, "span unr 4, no LDS")
    RUNU(4, true, 64, "span 64 B per lane, no LDS")
  }
  return 0;
}
