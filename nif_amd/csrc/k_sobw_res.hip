// k_sobw_res.hip -- SIREN_ResNet (MODE 1) instantiations of the streams-on-waves Sobolev kernel (k_sobw_dev.h)
#include "k_sobw_dev.h"
void launch_sobw_res(const SobArgs& J, int nblk, hipStream_t st, bool train) { launch_sobw_mode<1>(J, nblk, st, train); }
