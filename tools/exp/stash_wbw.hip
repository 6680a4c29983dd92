// stash_wbw.hip -- what does the memory system do with k_snet4's stash WRITE pattern?  (round 3)
// 768 workgroups x 4 waves, every wave walks its 16-point tiles and writes the ten stash slots of the 4x64 benchmark net
// (2.68 GB per launch) with no arithmetic, in several layouts:
//   0  production: [tile32][feature][32 points]; a wave's store instruction = 4 x 64 B segments, the other half of every
//      128-B line belongs to the sibling wave
//   1  [tile16][feature][16 points]: the wave's tile is 4 KB contiguous, a 128-B line = two feature rows of the SAME wave
//   2  [tile16][feature/4][16 points][4 features]: one 16-byte store per lane, a store instruction = 1 KiB contiguous
//   3  as 0 but slot-major order of the stores is kept while the tile walk is contiguous per workgroup
// build: hipcc --offload-arch=gfx950 -O3 -o stash_wbw stash_wbw.hip ; run: ./stash_wbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 3) void k_w(float* __restrict__ stash, long slot_stride, long B, int nslots) {
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const long nt16 = B / 16, ngroups = nt16 / 4;
  const long per = (ngroups + gridDim.x - 1) / gridDim.x;
  for (long it = 0;; ++it) {
    const long tg = PAT == 3 ? (long)blockIdx.x * per + it : (long)blockIdx.x + it * gridDim.x;
    if (PAT == 3 ? (it >= per || tg >= ngroups) : tg >= ngroups) break;
    const long t16 = tg * 4 + wid;
    const float val = (float)(t16 & 1023) + 0.001f * lane;
    for (int s = 0; s < nslots; ++s) {
      float* slot = stash + (long)s * slot_stride;
      if (PAT == 0 || PAT == 3) {
        const long row0 = (t16 >> 1) * 64 * 32 + 16 * (t16 & 1) + p;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) slot[row0 + (long)(16 * b + 4 * g + v) * 32] = val + b + v;
      } else if (PAT == 1) {
        const long row0 = t16 * 64 * 16 + p;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) slot[row0 + (long)(16 * b + 4 * g + v) * 16] = val + b + v;
      } else {
        // feature group q = 4b + g (16 groups of 4 features), [q][point][4]
        f32x4* t = reinterpret_cast<f32x4*>(slot + t16 * 64 * 16);
#pragma unroll
        for (int b = 0; b < 4; ++b) { f32x4 x = {val + b, val + b + 1, val + b + 2, val + b + 3}; t[(4 * b + g) * 16 + p] = x; }
      }
    }
  }
}

int main() {
  const long B = 1 << 20; const int nslots = 10;
  const long slot_stride = B * 64;
  float* d; CK(hipMalloc(&d, sizeof(float) * slot_stride * nslots));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)B * 64 * 4 * nslots;
  for (int pat = 0; pat < 4; ++pat)
    for (int grid : {768, 512, 1024}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        switch (pat) {
          case 0: hipLaunchKernelGGL(k_w<0>, dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nslots); break;
          case 1: hipLaunchKernelGGL(k_w<1>, dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nslots); break;
          case 2: hipLaunchKernelGGL(k_w<2>, dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nslots); break;
          default: hipLaunchKernelGGL(k_w<3>, dim3(grid), dim3(256), 0, 0, d, slot_stride, B, nslots); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("pattern %d grid %4d: %.3f ms  %.2f TB/s\n", pat, grid, best, bytes / best / 1e9);
    }
  return 0;
}
