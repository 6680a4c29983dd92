// k_sob.hip -- launcher of the Sobolev step kernel (k_sob_dev.h) for coordinate seeds; the instantiations that also carry
// parameter seeds live in k_sob_par.hip (a translation unit of its own: the two compile side by side)
#include "k_sob_dev.h"

long sob_ring_floats_per_wave(int n, int nh) { return (long)(nh + 1) * (2 + NIF_SOB_MAXSEED) * snet3_nbl(n) * 256; }

// bf16 dL/da stash rows are written by the streams-on-waves form (k_sobw<PR>) alone; k_sob always writes fp32 rows
bool sob_writes_da_bf16(const SNetArgs& a, int ns, bool any_par) { return a.da_bf16 != 0 && a.prec == 1 && sobw_supported(a, ns, any_par); }
int launch_sob(const SNetArgs& a, bool train, int ns, const int* seeds, const float* gt, float wj, float* ring, float* ju,
               bool query_only, hipStream_t st, const SobPar* par) {
  bool any_par = false;
  if (par) for (int d = 0; d < ns; ++d) any_par = any_par || par->par[d] >= 0;
  const int NBL = snet3_nbl(a.n);
  const long nt16 = 2 * ((a.B + 31) / 32);
  long ngroups = (nt16 + 3) / 4;
  const bool bf_ = a.WF4 && a.WB4 && !(NBL & 1) && NBL <= 6;
  // r4: the 1- / 2-seed "slim" instantiations (two workgroups per CU; r2's configs[4] kernels) are gone -- k_sobw takes those shapes,
  // and what still falls back to k_sob (NIF_SOBW=0, exchange tiles beyond the LDS) runs the general 3-stream form
  const bool slim = false, two = false;
  (void)bf_;
  const bool wav = sobw_supported(a, ns, any_par);      // k_sobw.hip: one 12-wave (65..128 units: 8-wave) workgroup per CU; r4: predict() too
  if (wav) ngroups = (nt16 + sobw_tiles_per_group(a.n, ns) - 1) / sobw_tiles_per_group(a.n, ns);
  const long cap = wav ? sobw_grid_cap() : (two ? 512 : (NBL <= 4 ? 256 * NIF_SOB_OCC : 256));
  const int nblk = (int)(ngroups < cap ? ngroups : cap);
  int one_buf = 0;
  {   // LDS the launch below will ask for (same arithmetic).  Two plane buffers when they fit, else ONE (r3: the derivative layers
      // must not refuse a shape the plain step trains); -1 = even that does not fit one CU, the caller refuses the shape
    const bool bfq = a.WF4 && a.WB4 && !(NBL & 1) && NBL <= 6;
    const size_t planeq = bfq ? (size_t)(NBL / 2) * NBL * 3 * 64 * 4 : (size_t)NBL * NBL * 256;
    const size_t smq = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
    size_t npwq = 1;
    if (par && !a.ll) for (int d = 0; d < ns; ++d) npwq += par->par[d] >= 0;
    const size_t restq = (smq + 4 * npwq * (size_t)(a.r * 64 + a.r * 16) + 8) * sizeof(float);
    const size_t llwq = a.ll ? (size_t)(2 * a.rl + (1 + NIF_SOB_MAXSEED) * (a.so + a.so_u) + NIF_SOB_MAXSEED * (2 * a.rl + a.so_u)) * 16 : 0;
    size_t need = 2 * planeq * sizeof(float) + restq;
    if (a.ll && need + 4 * llwq * sizeof(float) > 160u * 1024u && 4 * llwq > planeq) need += 4 * llwq * sizeof(float);
    if (need > 160u * 1024u) {
      one_buf = 1;      // (the last-layer epilogue's scratch then needs LDS of its own: there is no idle plane buffer)
      if (planeq * sizeof(float) + restq + 4 * llwq * sizeof(float) > 160u * 1024u) return -1;
    }
  }
  if (query_only) return nblk;
  SobArgs J;
  J.s = a; J.ns = ns; J.gt = gt; J.wj = wj; J.ring = ring; J.JU = ju;
  for (int d = 0; d < NIF_SOB_MAXSEED; ++d) {
    J.seed[d] = d < ns ? seeds[d] : 0;
    J.gcol[d] = (par && d < ns) ? par->gcol[d] : d;
    J.par[d] = (par && d < ns) ? par->par[d] : -1;
  }
  J.ZT = par ? par->ZT : nullptr; J.DZT = par ? par->DZT : nullptr;
  J.npar = 0; J.nx_tot = ns; J.DAT = nullptr; J.ZTL = nullptr; J.zl_rows = 0; J.one_buf = one_buf;
  for (int d = 0; d < NIF_SOB_MAXSEED; ++d) { J.parc[d] = 0; J.pcol[d] = 0; }
  if (a.ll && par) {       // last-layer class: parameter columns are heads of the epilogue, not streams
    J.npar = par->npar; J.nx_tot = ns + par->npar; J.DAT = par->DAT; J.ZTL = par->ZTL; J.zl_rows = par->zl_rows;
    for (int e = 0; e < par->npar; ++e) { J.parc[e] = par->parc[e]; J.pcol[e] = par->pcol[e]; }
    for (int d = 0; d < NIF_SOB_MAXSEED; ++d) J.par[d] = -1;
    any_par = false;
  }
  {   // the derivative term's bookkeeping (SobArgs wu / wjn / ymask / gstride)
    const int so_out = a.ll ? a.so_u : a.so;
    const int nx_all = (par && par->nx_all > 0) ? par->nx_all : J.nx_tot;
    const int ny = (par && par->ny > 0) ? par->ny : so_out;
    J.gstride = (par && par->gstride > 0) ? par->gstride : J.nx_tot;
    J.ymask = (par && par->ymask) ? par->ymask : 0xFFFFFFFFu;
    J.wu = (par && par->no_primal) ? 0.0f : 1.0f;
    J.wjn = wj / ((float)ny * (float)nx_all);
  }
  if (wav) { launch_sobw(J, nblk, st, train); return nblk; }
  dim3 grid(nblk), block(256);
  const bool bf = a.WF4 && a.WB4 && !(NBL & 1) && NBL <= 6;      // whole bf16 planes in LDS: up to n = 96
  const bool sgn = !a.res && !a.nif_skip && (long)(a.nh + 1) * 4 * NBL <= 128;  // sign bits fit the 128-bit shift register (SIREN only)
  const size_t plane = bf ? (size_t)(NBL / 2) * NBL * 3 * 64 * 4 : (size_t)NBL * NBL * 256;
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  size_t npw = 1;
  if (any_par) for (int d = 0; d < ns; ++d) npw += J.par[d] >= 0;
  const size_t shm = ((one_buf ? 1 : 2) * plane + sm_tot + 4 * npw * (size_t)(a.r * 64 + a.r * 16) + 8) * sizeof(float);
  J.ll_plane = 0;
  if (a.ll) {
    const size_t llw = (size_t)(2 * a.rl + (1 + NIF_SOB_MAXSEED) * (a.so + a.so_u) + NIF_SOB_MAXSEED * (2 * a.rl + a.so_u)) * 16;
    size_t shm_ll = shm + 4 * llw * sizeof(float);
    if (!one_buf && shm_ll > 160u * 1024u && 4 * llw <= plane) { J.ll_plane = 1; shm_ll = shm; }   // scratch in the idle plane buffer
    launch_sob_ll(J, train, bf, nblk, shm_ll, st);
    return nblk;
  }
  if (any_par) { launch_sob_par(J, train, bf, nblk, shm, st); return nblk; }
#define SBL(NBL_, MODE_, TR_, BF_, SGN_)                                                                            \
  {                                                                                                             \
    if (shm > 48 * 1024)                                                                                        \
      (void)hipFuncSetAttribute((const void*)k_sob<NBL_, MODE_, TR_, BF_, SGN_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)shm);                                                                      \
    hipLaunchKernelGGL((k_sob<NBL_, MODE_, TR_, BF_, SGN_>), grid, block, shm, st, J);                                \
  }
#define SBK(NBL_, BF_)                                                               \
  if (a.nif_skip) { if (train) SBL(NBL_, 2, true, BF_, false) else SBL(NBL_, 2, false, BF_, false) }   \
  else if (a.res) { if (train) SBL(NBL_, 1, true, BF_, false) else SBL(NBL_, 1, false, BF_, false) }   \
  else if (train) { if (sgn) SBL(NBL_, 0, true, BF_, true) else SBL(NBL_, 0, true, BF_, false) } \
  else SBL(NBL_, 0, false, BF_, false)
  switch (NBL) {
    case 1: SBK(1, 0) break;
    case 2: if (bf && a.prec == 1) { SBK(2, 2) } else if (bf) { SBK(2, 1) } else { SBK(2, 0) } break;
    case 3: SBK(3, 0) break;
    case 4: if (bf && a.prec == 1) { SBK(4, 2) } else if (bf) { SBK(4, 1) } else { SBK(4, 0) } break;
    case 6: if (bf && a.prec == 1) { SBK(6, 2) } else if (bf) { SBK(6, 1) } else { SBK(6, 0) } break;
    default: SBK(8, 0) break;
  }
#undef SBK
#undef SBL
  return nblk;
}
