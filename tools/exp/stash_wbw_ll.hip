// stash_wbw_ll.hip -- k_snet4<8, .., LL>'s stash traffic alone at configs[3]'s size (r4): 2 M points, 128-row stash
// tiles, 14 slots (7 x h, 7 x dL/da) of 1 GiB each = 15 GB written and 3.8 GB re-read per launch, no arithmetic.
//   mode 0: stores only, the production tile walk (tile group = blockIdx + k * gridDim)     mode 1: + the adjoint's re-reads of the 7 h slots
//   mode 2: as 1 with a contiguous tile walk per workgroup                                   mode 3: as 1 at 1/8 of the batch (footprint)
// build: hipcc --offload-arch=gfx950 -O3 -o stash_wbw_ll stash_wbw_ll.hip ; run: ./stash_wbw_ll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool RD, bool CONTIG>
__global__ __launch_bounds__(256, 2) void k_w(float* __restrict__ stash, long slot_stride, long B, int nh, float* __restrict__ sink) {
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const long nt16 = B / 16, ngroups = nt16 / 4;
  const long per = (ngroups + gridDim.x - 1) / gridDim.x;
  float acc = 0.f;
  for (long it = 0;; ++it) {
    const long tg = CONTIG ? (long)blockIdx.x * per + it : (long)blockIdx.x + it * gridDim.x;
    if (CONTIG ? (it >= per || tg >= ngroups) : tg >= ngroups) break;
    const long t16 = tg * 4 + wid;
    const float val = (float)(t16 & 1023) + 0.001f * lane;
    const long row0 = (t16 >> 1) * 128 * 32 + 16 * (t16 & 1) + p;
    for (int s = 0; s <= nh; ++s) {           // forward: h_0 .. h_nh
      float* slot = stash + (long)s * slot_stride;
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) slot[row0 + (long)(16 * b + 4 * g + v) * 32] = val + b + v;
    }
    for (int s = nh; s >= 0; --s) {           // adjoint: re-read h_s, write dL/da_s
      float x = val;
      if (RD) {
        const float* slot = stash + (long)s * slot_stride;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) x += slot[row0 + (long)(16 * b + 4 * g + v) * 32];
        acc += x;
      }
      float* slot = stash + (long)(nh + 1 + s) * slot_stride;
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) slot[row0 + (long)(16 * b + 4 * g + v) * 32] = x + b + v;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int nh = 6, nslots = 2 * (nh + 1);
  const long Bmax = 1 << 21;
  float *d, *sink; CK(hipMalloc(&d, sizeof(float) * Bmax * 128 * nslots)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode)
    for (int grid : {512, 256, 1024}) {
      const long B = mode == 3 ? Bmax / 8 : Bmax;
      const long slot_stride = B * 128;
      const double wbytes = (double)B * 128 * 4 * nslots, rbytes = mode ? (double)B * 128 * 4 * (nh + 1) : 0.0;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL((k_w<false, false>), dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nh, sink); break;
          case 2: hipLaunchKernelGGL((k_w<true, true>), dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nh, sink); break;
          default: hipLaunchKernelGGL((k_w<true, false>), dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nh, sink); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("mode %d grid %4d: %.3f ms  written %.2f GB read %.2f GB  %.2f TB/s\n", mode, grid, best, wbytes / 1e9, rbytes / 1e9,
             (wbytes + rbytes) / best / 1e9);
    }
  return 0;
}
