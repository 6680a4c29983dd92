// k_sobw_nif.hip -- class NIF (MODE 2: any activation, skip connections) instantiations of the streams-on-waves Sobolev kernel (k_sobw_dev.h)
#include "k_sobw_dev.h"
void launch_sobw_nif(const SobArgs& J, int nblk, hipStream_t st, bool train) { launch_sobw_mode<2>(J, nblk, st, train); }
