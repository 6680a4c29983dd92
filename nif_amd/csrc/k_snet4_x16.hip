// k_snet4_x16.hip -- the instantiations of k_snet4 (k_snet4_dev.h) with PR = 3 (r5): fp32-exact products of the hidden n x n layers on
// HALF pairs -- planes (hi, lo) x operand (hi, lo), three v_mfma_f32_16x16x32_f16 per pair forward and in the data adjoint (k_pack16b
// mode 3, k_plane_scales; tools/exp/f16_split_mfma.hip) -- for every SIREN form (NIFMultiScale plain / resblock, the last-layer class,
// training and inference).  They replace the six- / three-product bf16 forms of r1 - r4 for these nets: half the forward matrix work
// and two thirds of its chunk bytes, 22 significand bits in the data adjoint.  A translation unit of its own so that it compiles
// next to k_snet4.hip.
#include "k_snet4_dev.h"

void launch_snet4_x16(const SNetArgs& a, bool train, int nblk, size_t shm, hipStream_t st) {
  const int NBL = snet3_nbl(a.n);
  dim3 grid(nblk), block(256);
#define S4L(NBL_, TR_, ACT_, MODE_, SGN_, LL_)                                                                     \
  {                                                                                                              \
    if (shm > 48 * 1024)                                                                                         \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_, 3>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                           \
    hipLaunchKernelGGL((k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_, 3>), grid, block, shm, st, a);                \
  }
#define S4M(NBL_)                                                                                                \
  if (a.res) { if (train) S4L(NBL_, true, ACT_SINE, 1, true, false) else S4L(NBL_, false, ACT_SINE, 1, false, false) } \
  else if (train) S4L(NBL_, true, ACT_SINE, 0, true, false)                                                      \
  else S4L(NBL_, false, ACT_SINE, 0, false, false)
#define S4N(NBL_)   /* last-layer class: no NIF skip form */                                                     \
  if (a.res) { if (train) S4L(NBL_, true, ACT_SINE, 1, true, true) else S4L(NBL_, false, ACT_SINE, 1, false, true) } \
  else if (train) S4L(NBL_, true, ACT_SINE, 0, true, true)                                                       \
  else S4L(NBL_, false, ACT_SINE, 0, false, true)
#define S4(NBL_) if (a.ll) { S4N(NBL_) } else { S4M(NBL_) }
  switch (NBL) {
    case 2: S4(2) break;
    case 4: S4(4) break;
    case 6: S4(6) break;
    default: S4(8) break;
  }
#undef S4
#undef S4N
#undef S4M
#undef S4L
}
