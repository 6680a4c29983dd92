// k_snet4_f16.hip -- the mixed_float16 instantiations of k_snet4 (k_snet4_dev.h, PR = 2): Keras' `mixed_float16` policy
// (reference nif/model.py:101-105 hands the name to tf.keras.mixed_precision) restated like the build's mixed_bfloat16 --
// variables fp32; the operands of the ShapeNet's hidden n x n products (activations and omega_0-folded planes forward, dL/da
// and planes in the data adjoint) rounded to half precision, ONE v_mfma_f32_16x16x32_f16 per operand pair, fp32 accumulation;
// biases, activations, first / last layer, loss, ParameterNet and every weight-gradient sum stay fp32.  dL/da is rounded under a
// loss scale and the chain scaled back (k_snet4_dev.h): where Keras' compile() wraps the optimizer in a LossScaleOptimizer with
// one dynamic scale per step, here every point's dL/da vector takes its own power of two (largest entry in [2^14, 2^15)): no
// overflow, no skipped step, no history.  A translation unit of its own so that it compiles next to k_snet4.hip.
#include "k_snet4_dev.h"

void launch_snet4_f16(const SNetArgs& a, bool train, int nblk, size_t shm, hipStream_t st) {
  const int NBL = snet3_nbl(a.n);
  dim3 grid(nblk), block(256);
#define S4L(NBL_, TR_, ACT_, MODE_, SGN_, LL_)                                                                     \
  {                                                                                                              \
    if (shm > 48 * 1024)                                                                                         \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_, 2>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                           \
    hipLaunchKernelGGL((k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_, 2>), grid, block, shm, st, a);                \
  }
#define S4M(NBL_)                                                                                                \
  if (a.nif_skip) { if (train) S4L(NBL_, true, -1, 2, false, false) else S4L(NBL_, false, -1, 2, false, false) } \
  else if (a.res) { if (train) S4L(NBL_, true, ACT_SINE, 1, true, false) else S4L(NBL_, false, ACT_SINE, 1, false, false) } \
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
