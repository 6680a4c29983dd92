// k_sobw.hip -- host side of the streams-on-waves Sobolev kernel (k_sobw_dev.h) and the plain-SIREN (MODE 0) instantiations; the
// SIREN_ResNet (MODE 1) and class-NIF (MODE 2) ones live in k_sobw_res.hip / k_sobw_nif.hip
#include "k_sobw_dev.h"

// ---- host side ---------------------------------------------------------------------------------
size_t sobw_shmem(const SNetArgs& a, int NBL, int ns) {
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + ((a.r + 3) & ~3) + ((a.so + 3) & ~3) + 4) * 16;
  const size_t pw = 2 * a.r * 64 + 2 * ni;
  const size_t W = NIF_SOBW_WMAX(NBL);      // waves = tiles per group x streams (an upper bound over the seed counts)
  return (size_t)2 * NBL * 3 * 64 * 16 + (sm_tot + W * pw + W * NBL * 256 + W * a.r * 16 + 16) * sizeof(float);
}
// coordinate seeds of a plain SIREN NIFMultiScale net on the packed bf16 planes, n <= 64, training
bool sobw_supported(const SNetArgs& a, int ns, bool any_par) {
  static const int on = [] { const char* e = getenv("NIF_SOBW"); return e ? atoi(e) : 1; }();
  const int NBL = snet3_nbl(a.n);
  if (!on || ns < 1 || ns > 3 || any_par || a.ll || !a.WF4 || !a.WB4) return false;
  if (a.res && (a.nh & 1)) return false;
  if ((NBL & 1) || NBL > 8 || a.nh < 1 || a.r < 1) return false;
  return sobw_shmem(a, NBL, ns) <= 160u * 1024u;
}
int sobw_tiles_per_group(int n, int ns) { return NIF_SOBW_TPG(snet3_nbl(n), ns); }
int sobw_grid_cap() { return 256; }       // one workgroup per CU
void launch_sobw_plain(const SobArgs& J, int nblk, hipStream_t st, bool train) { launch_sobw_mode<0>(J, nblk, st, train); }
void launch_sobw(const SobArgs& J, int nblk, hipStream_t st, bool train) {
  if (J.s.nif_skip) launch_sobw_nif(J, nblk, st, train);
  else if (J.s.res) launch_sobw_res(J, nblk, st, train);
  else launch_sobw_plain(J, nblk, st, train);
}
