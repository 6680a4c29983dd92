// k_sob_ll.hip -- the Sobolev step kernel (k_sob_dev.h) instantiated for the last-layer-parameterised class (LL): the shared
// dense SIREN ShapeNet x -> phi with forward tangents w.r.t. coordinate columns and their adjoint, u = Dot(phi, a) + bias
// (reference nif/model.py:1219-1269 under JacobianLayer, nif/layers/gradient.py:36-49).  General form only (3 streams,
// act'(a) ring, 1 workgroup per CU); units <= 128 (bf16-split planes where k_snet4<LL> has packed them, f32-input planes else).
#include "k_sob_dev.h"

void launch_sob_ll(const SobArgs& J, bool train, bool bf, int nblk, size_t shm, hipStream_t st) {
  const SNetArgs& a = J.s;
  const int NBL = snet3_nbl(a.n);
  dim3 grid(nblk), block(256);
#define SLL(NBL_, MODE_, TR_, BF_)                                                                                     \
  {                                                                                                                    \
    if (shm > 48 * 1024)                                                                                               \
      (void)hipFuncSetAttribute((const void*)k_sob<NBL_, MODE_, TR_, BF_, false, NIF_SOB_MAXSEED, false, true>,        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                 \
    hipLaunchKernelGGL((k_sob<NBL_, MODE_, TR_, BF_, false, NIF_SOB_MAXSEED, false, true>), grid, block, shm, st, J);   \
  }
#define SLK(NBL_, BF_)                                                                  \
  if (a.res) { if (train) SLL(NBL_, 1, true, BF_) else SLL(NBL_, 1, false, BF_) }        \
  else { if (train) SLL(NBL_, 0, true, BF_) else SLL(NBL_, 0, false, BF_) }
  switch (NBL) {
    case 1: SLK(1, 0) break;
    case 3: SLK(3, 0) break;
    case 2: if (bf) { SLK(2, 1) } else { SLK(2, 0) } break;
    case 4: if (bf) { SLK(4, 1) } else { SLK(4, 0) } break;
    case 6: if (bf) { SLK(6, 1) } else { SLK(6, 0) } break;
    default: SLK(8, 0) break;
  }
#undef SLK
#undef SLL
}
bool sob_ll_supported(const SNetArgs& a) {
  const int NBL = snet3_nbl(a.n);
  (void)NBL;
  return a.ll && a.n <= 128 && a.nh >= 1 && a.so <= 64;
}
