// k_sob_par.hip -- the Sobolev step kernel (k_sob_dev.h) instantiated with parameter seeds (PAR): x_index of JacobianLayer
// addressing ParameterNet inputs (reference nif/layers/gradient.py:207-231 takes any input column).  A rare configuration:
// one general form per width (3 streams, act'(a) ring, 1 workgroup per CU); the bf16 policy runs on the exact-split products.
#include "k_sob_dev.h"

void launch_sob_par(const SobArgs& J, bool train, bool bf, int nblk, size_t shm, hipStream_t st) {
  const SNetArgs& a = J.s;
  const int NBL = snet3_nbl(a.n);
  dim3 grid(nblk), block(256);
#define SPL(NBL_, MODE_, TR_, BF_)                                                                                     \
  {                                                                                                                    \
    if (shm > 48 * 1024)                                                                                               \
      (void)hipFuncSetAttribute((const void*)k_sob<NBL_, MODE_, TR_, BF_, false, NIF_SOB_MAXSEED, true>,               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                 \
    hipLaunchKernelGGL((k_sob<NBL_, MODE_, TR_, BF_, false, NIF_SOB_MAXSEED, true>), grid, block, shm, st, J);          \
  }
#define SPK(NBL_, BF_)                                                                  \
  if (a.nif_skip) { if (train) SPL(NBL_, 2, true, BF_) else SPL(NBL_, 2, false, BF_) }   \
  else if (a.res) { if (train) SPL(NBL_, 1, true, BF_) else SPL(NBL_, 1, false, BF_) }   \
  else { if (train) SPL(NBL_, 0, true, BF_) else SPL(NBL_, 0, false, BF_) }
  switch (NBL) {
    case 1: SPK(1, 0) break;
    case 2: if (bf) { SPK(2, 1) } else { SPK(2, 0) } break;
    case 3: SPK(3, 0) break;
    case 4: if (bf) { SPK(4, 1) } else { SPK(4, 0) } break;
    case 6: if (bf) { SPK(6, 1) } else { SPK(6, 0) } break;
    default: SPK(8, 0) break;
  }
#undef SPK
#undef SPL
}
