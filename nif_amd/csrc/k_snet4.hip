// k_snet4.hip -- k_snet3 with every n x n product on the BF16 matrix cores at fp32 accuracy.
//
// gfx950 runs v_mfma_f32_16x16x32_bf16 at 16x the rate of the f32-input MFMAs (1024 vs 64 FLOP/clk/SIMD).
// An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand bits, same exponent range):
//     x = x0 + x1 + x2 ,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// and bf16 x bf16 products are exact in the fp32 accumulator, so
//     a.b  =  a0b0 + (a0b1 + a1b0) + (a0b2 + a2b0 + a1b1) + O(2^-24 |a||b|)
// Six MFMAs (small terms first) reproduce the fp32 product to 4.2e-8 rms / 4.3e-7 max of sum|a_k b_k| at K = 64 --
// slightly BETTER than v_mfma_f32_16x16x4_f32 itself (6.1e-8 / 5.9e-7; tools/exp/bf16_split_mfma.hip, measured
// on MI355X) -- in 6 x 16 = 96 matrix-pipe cycles per 16x16x32 block instead of 8 x 32 = 256.  The forward pass
// (predictions and loss, 1e-5 bar) uses the 6-product form; the data adjoint, whose bar is the gradient
// tolerance, the 3-product form (a0b0 + a0b1 + a1b0, 1.9e-6 rms) in 48 cycles.
//
// Weight planes are pre-split at pack time (k_pack16b) into bf16 A operands; the activation tile (B operand) is
// split once per layer in registers and shared by all r+1 planes: the per-point latent factor zt_k is applied
// to the PRODUCT (column scaling commutes with the GEMM).  Planes stream L2 -> LDS by DMA in K-step chunks
// (32 input features x all outputs: NBL x 3 KiB forward, NBL x 2 KiB adjoint), double buffered, one barrier per
// chunk -> 24 KiB of plane LDS per workgroup for the 64-wide net.
//
// Everything else (tiles, stashes, ring, first / last layer, loss, modes) is k_snet3's; see there.
#include "k_snet4_dev.h"

// ---- packing ------------------------------------------------------------------------------------
// K-slot (g, t), g = lane >> 4, t = 0..7 of K-step ks  <->  feature 16*(2ks + (t >> 2)) + 4g + (t & 3): exactly what a
// lane of the previous layer's C/D tile holds in blocks 2ks, 2ks+1 -- the activation tile is the B operand as is.
//   fwd chunk (plane, ks): unit ((ob*3 + s)*64 + lane), 8 bf16: split s of M[in = slot(ks,g,t)][out = 16ob + (lane&15)]
//   bwd chunk (plane, ks): unit ((ib*2 + s)*64 + lane), 8 bf16: split s of M[in = 16ib + (lane&15)][out = slot(ks,g,t)]
// blockIdx.y = matrix j of a batch of equally shaped matrices `mstride` slots apart (the hidden hyper-matrices)
// scale: omega_0 of the SIREN layer, folded into the planes (r3) so that no consumer multiplies by it per element
// mode 0: the split groups above.  mode 1 / 2 (the policies' COMPACT plane set, late r4): one 16-bit value per element -- bf16(x)
// (mixed_bfloat16) or half(x) (mixed_float16; RNE, saturated) -- unit (ob * 64 + lane) of a chunk, a third / half of the split
// planes' bytes: k_snet4<.., PR> / k_snet6<.., PR> stream these (CP in mfma_x6 / mfma_x3), every other kernel the split groups
// mode 3 (r5): the EXACT-PRODUCT HALF planes of k_snet6 -- x = hi + lo with hi = half(x), lo = half(x - hi) (11 + 11 significand bits:
// |x - hi - lo| <= 2^-24 |x|), x = s w0 M with s the plane's power of two (k_plane_scales: max |s w0 M| in [2^13, 2^14), so that lo
// keeps its bits inside half's exponent range; the kernels scale the products back, exactly).  Three half products hi.hi + hi.lo + lo.hi
// then carry an fp32 product: measured on MI355X 2.8e-8 rms / 2.1e-7 max of sum|a_k b_k| at K = 64 -- the bf16 6-product form's
// 2.7e-8 / 2.9e-7, the f32-input MFMA's 3.9e-8 / 4.0e-7 (tools/exp/f16_split_mfma.hip, profiles/r05_f16_probe.txt) -- at HALF the
// matrix work and two thirds of the plane bytes.  Both directions use the adjoint geometry: unit ((blk*2 + s)*64 + lane)
__global__ void k_pack16b(const float* __restrict__ theta, MatRef m, long mstride, int NBL, __bf16* __restrict__ WF,
                          __bf16* __restrict__ WB, long fstride, long bstride, float scale, int mode, const float* __restrict__ pscale) {
  m.base_k += (long)blockIdx.y * mstride; m.base_last += (long)blockIdx.y * mstride;
  WF += (long)blockIdx.y * fstride; WB += (long)blockIdx.y * bstride;
  const int NCH = NBL / 2;
  const int nsf = mode == 3 ? 2 : (mode ? 1 : 3), nsb = (mode == 1 || mode == 2) ? 1 : 2;
  const long fwd_plane = (long)NCH * NBL * nsf * 64 * 8, bwd_plane = (long)NCH * NBL * nsb * 64 * 8;
  const long total_f = fwd_plane * (m.r + 1), total_b = bwd_plane * (m.r + 1);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_f + total_b; idx += (long)gridDim.x * blockDim.x) {
    const bool fwd = idx < total_f;
    const long e = fwd ? idx : idx - total_f;
    const int ns = fwd ? nsf : nsb;
    const long per_plane = fwd ? fwd_plane : bwd_plane;
    const int k = (int)(e / per_plane);
    long rem = e - (long)k * per_plane;
    const int t = rem & 7; rem >>= 3;
    const int lane = rem & 63; rem >>= 6;
    const int s = (int)(rem % ns); rem /= ns;
    const int blk = (int)(rem % NBL);
    const int ks = (int)(rem / NBL);
    const int slot = 16 * (2 * ks + (t >> 2)) + 4 * (lane >> 4) + (t & 3);
    const int row = 16 * blk + (lane & 15);
    const int in = fwd ? slot : row, out = fwd ? row : slot;
    float x = (in < m.nin && out < m.nout) ? scale * theta[matref_index(m, k, in, out)] : 0.f;
    if (mode == 3) {
      x *= pscale[(blockIdx.y * (m.r + 1) + k) * 2];
      const _Float16 h0 = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
      const _Float16 h1 = (_Float16)(x - (float)h0);
      reinterpret_cast<_Float16*>(fwd ? WF : WB)[e] = s == 0 ? h0 : h1;
      continue;
    }
    if (mode == 2) {
      reinterpret_cast<_Float16*>(fwd ? WF : WB)[e] = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
      continue;
    }
    const __bf16 x0 = (__bf16)x;
    const float r1 = x - (float)x0;
    const __bf16 x1 = (__bf16)r1;
    const __bf16 x2 = (__bf16)(r1 - (float)x1);
    (fwd ? WF : WB)[e] = s == 0 ? x0 : (s == 1 ? x1 : x2);
  }
}
// power-of-two scale of plane (matrix blockIdx.y, k = blockIdx.x) for the half planes (mode 3): max |scale M| 2^e in [2^13, 2^14)
__global__ void k_plane_scales(const float* __restrict__ theta, MatRef m, long mstride, float scale, float* __restrict__ pscale) {
  m.base_k += (long)blockIdx.y * mstride; m.base_last += (long)blockIdx.y * mstride;
  const int k = blockIdx.x;
  float mx = 0.f;
  for (int idx = threadIdx.x; idx < m.nin * m.nout; idx += blockDim.x) {
    const int in = idx / m.nout, out = idx - in * m.nout;
    mx = fmaxf(mx, fabsf(scale * theta[matref_index(m, k, in, out)]));
  }
  __shared__ float red[256];
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    mx = red[0];
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &ex); else ex = 14;     // mx = f 2^ex, f in [0.5, 1)
    int sh = 14 - ex;
    sh = sh > 60 ? 60 : (sh < -60 ? -60 : sh);       // (the scaled constants of sine16_tag_sc stay normal numbers)
    pscale[(blockIdx.y * (m.r + 1) + k) * 2] = ldexpf(1.0f, sh);          // [s | 1 / s]
    pscale[(blockIdx.y * (m.r + 1) + k) * 2 + 1] = ldexpf(1.0f, -sh);
  }
}
// r5: the split groups (mode 0) AND the half planes (mode 3) of a batch of matrices in ONE launch, the planes' powers of two found in
// the kernel (every workgroup of a half plane takes the maximum over its plane itself -- <= 16 K cached words -- instead of waiting
// for a k_plane_scales launch; all of them arrive at the same power of two, the workgroup holding the plane's first forward unit
// writes it for the consumers): three launches of ~5 us per training step -> one.  Index space [mode-0 units | mode-3 units], both
// multiples of the workgroup's 256 elements, so a workgroup's elements of one loop trip lie in ONE part and ONE plane.
__global__ __launch_bounds__(256) void k_pack16b_dual(const float* __restrict__ theta, MatRef m, long mstride, int NBL,
                                                     __bf16* __restrict__ WF, __bf16* __restrict__ WB, long fstride, long bstride,
                                                     _Float16* __restrict__ WFx, _Float16* __restrict__ WBx, long fxstride, long bxstride,
                                                     float scale, float* __restrict__ pscale) {
  m.base_k += (long)blockIdx.y * mstride; m.base_last += (long)blockIdx.y * mstride;
  WF += (long)blockIdx.y * fstride; WB += (long)blockIdx.y * bstride;
  WFx += (long)blockIdx.y * fxstride; WBx += (long)blockIdx.y * bxstride;
  const int NCH = NBL / 2;
  const long fwd0 = (long)NCH * NBL * 3 * 64 * 8, bwd0 = (long)NCH * NBL * 2 * 64 * 8, px = bwd0;   // units per plane: mode 0 fwd / bwd, mode 3 (both directions)
  const long tot0 = (fwd0 + bwd0) * (m.r + 1), totx = 2 * px * (m.r + 1);
  __shared__ float red[256];
  int k_have = -1;
  float s_have = 1.0f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < tot0 + totx; idx += (long)gridDim.x * blockDim.x) {
    const bool half = idx >= tot0;
    const long i2 = half ? idx - tot0 : idx;
    const long tf = half ? px * (m.r + 1) : fwd0 * (m.r + 1);
    const bool fwd = i2 < tf;
    const long e = fwd ? i2 : i2 - tf;
    const int ns = half ? 2 : (fwd ? 3 : 2);
    const long per_plane = half ? px : (fwd ? fwd0 : bwd0);
    const int k = (int)(e / per_plane);
    long rem = e - (long)k * per_plane;
    const int t = rem & 7; rem >>= 3;
    const int lane = rem & 63; rem >>= 6;
    const int s = (int)(rem % ns); rem /= ns;
    const int blk = (int)(rem % NBL);
    const int ks = (int)(rem / NBL);
    const int slot = 16 * (2 * ks + (t >> 2)) + 4 * (lane >> 4) + (t & 3);
    const int row = 16 * blk + (lane & 15);
    const int in = fwd ? slot : row, out = fwd ? row : slot;
    float x = (in < m.nin && out < m.nout) ? scale * theta[matref_index(m, k, in, out)] : 0.f;
    if (half) {
      if (k != k_have) {        // (uniform over the workgroup: see above)
        float mx = 0.f;
        for (int q = threadIdx.x; q < m.nin * m.nout; q += 256) {
          const int qi = q / m.nout, qo = q - qi * m.nout;
          mx = fmaxf(mx, fabsf(scale * theta[matref_index(m, k, qi, qo)]));
        }
        __syncthreads();
        red[threadIdx.x] = mx;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
          if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
          __syncthreads();
        }
        mx = red[0];
        int ex = 0;
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &ex); else ex = 14;     // (k_plane_scales' rule)
        int sh = 14 - ex;
        sh = sh > 60 ? 60 : (sh < -60 ? -60 : sh);
        s_have = ldexpf(1.0f, sh); k_have = k;
        if (fwd && e == (long)k * per_plane && pscale) {       // the thread holding the plane's first forward unit
          pscale[(blockIdx.y * (m.r + 1) + k) * 2] = s_have;
          pscale[(blockIdx.y * (m.r + 1) + k) * 2 + 1] = ldexpf(1.0f, -sh);
        }
      }
      x *= s_have;
      const _Float16 h0 = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
      const _Float16 h1 = (_Float16)(x - (float)h0);
      (fwd ? WFx : WBx)[e] = s == 0 ? h0 : h1;
      continue;
    }
    const __bf16 x0 = (__bf16)x;
    const float r1 = x - (float)x0;
    const __bf16 x1 = (__bf16)r1;
    const __bf16 x2 = (__bf16)(r1 - (float)x1);
    (fwd ? WF : WB)[e] = s == 0 ? x0 : (s == 1 ? x1 : x2);
  }
}
// split groups into (WF, WB) and half planes into (WFx, WBx) + their scales into pscale[(matrix, plane)][s | 1 / s], one launch
void launch_pack16b_dual(const float* theta, const MatRef& m0, long mstride, int nmat, int NBL, void* WF, void* WB, long fstride_elems,
                         long bstride_elems, void* WFx, void* WBx, long fxstride_elems, long bxstride_elems, float scale, float* pscale,
                         hipStream_t st) {
  const long total = (long)(NBL / 2) * NBL * (5 + 4) * 64 * 8 * (m0.r + 1);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_pack16b_dual, dim3(grid, nmat), dim3(256), 0, st, theta, m0, mstride, NBL, (__bf16*)WF, (__bf16*)WB, fstride_elems,
                     bstride_elems, (_Float16*)WFx, (_Float16*)WBx, fxstride_elems, bxstride_elems, scale, pscale);
}
void launch_pack16b(const float* theta, const MatRef& m, int NBL, void* WF, void* WB, float scale, hipStream_t st, int mode) {
  launch_pack16b_batch(theta, m, 0, 1, NBL, WF, WB, 0, 0, scale, st, mode);
}
void launch_pack16b_batch(const float* theta, const MatRef& m0, long mstride, int nmat, int NBL, void* WF, void* WB,
                          long fstride_elems, long bstride_elems, float scale, hipStream_t st, int mode, float* pscale) {
  const long total = (long)(NBL / 2) * NBL * 5 * 64 * 8 * (m0.r + 1);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (mode == 3) hipLaunchKernelGGL(k_plane_scales, dim3(m0.r + 1, nmat), dim3(256), 0, st, theta, m0, mstride, scale, pscale);
  hipLaunchKernelGGL(k_pack16b, dim3(grid, nmat), dim3(256), 0, st, theta, m0, mstride, NBL, (__bf16*)WF, (__bf16*)WB,
                     fstride_elems, bstride_elems, scale, mode, pscale);
}

// phi layer of the last-layer class: dense W[n][sop], sop <= 32 (two 16-output blocks)
__global__ void k_pack_phi(const float* __restrict__ theta, long w_off, int n, int sop, int NBL, __bf16* __restrict__ WPF,
                           __bf16* __restrict__ WPB) {
  const int NCH = NBL / 2;
  const long total_f = (long)NCH * 2 * 3 * 64 * 8, total_b = (long)NBL * 2 * 64 * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_f + total_b; idx += (long)gridDim.x * blockDim.x) {
    const bool fwd = idx < total_f;
    long rem = fwd ? idx : idx - total_f;
    const int t = rem & 7; rem >>= 3;
    const int lane = rem & 63; rem >>= 6;
    const int ns = fwd ? 3 : 2;
    const int sp = (int)(rem % ns); rem /= ns;
    int in, out;
    if (fwd) {
      const int ob = (int)(rem % 2), ks = (int)(rem / 2);
      in = 16 * (2 * ks + (t >> 2)) + 4 * (lane >> 4) + (t & 3);
      out = 16 * ob + (lane & 15);
    } else {
      const int ib = (int)rem;
      in = 16 * ib + (lane & 15);
      out = 16 * (t >> 2) + 4 * (lane >> 4) + (t & 3);
    }
    const float x = (in < n && out < sop) ? theta[w_off + (long)in * sop + out] : 0.f;
    const __bf16 x0 = (__bf16)x;
    const float r1 = x - (float)x0;
    const __bf16 x1 = (__bf16)r1;
    const __bf16 x2 = (__bf16)(r1 - (float)x1);
    (fwd ? WPF : WPB)[fwd ? idx : idx - total_f] = sp == 0 ? x0 : (sp == 1 ? x1 : x2);
  }
}
long snet4_phi_fwd_elems(int n) { return (long)(snet3_nbl(n) / 2) * 2 * 3 * 64 * 8; }
long snet4_phi_bwd_elems(int n) { return (long)snet3_nbl(n) * 2 * 64 * 8; }
void launch_pack_phi(const float* theta, long w_off, int n, int sop, void* WPF, void* WPB, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_phi, dim3(64), dim3(256), 0, st, theta, w_off, n, sop, snet3_nbl(n), (__bf16*)WPF, (__bf16*)WPB);
}

// ---- host side ---------------------------------------------------------------------------------
// floats per k of the LDS small-vector image (last-layer class: + last_layer_bias and the rl x rl map)
int snet4_nsm_ll(int si, int sop, int nh, int n, int sou, int rl) {
  return snet3_nsm(si, sop, nh, n) + ((sou + 3) & ~3) + ((rl * rl + 3) & ~3);
}
__global__ void k_ll_slots(const float* __restrict__ theta, LLSlotMap m, float* __restrict__ slots) {
  for (int sgi = blockIdx.y; sgi < m.nseg; sgi += gridDim.y)
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < m.seg[sgi].len; e += (long)gridDim.x * blockDim.x)
      slots[m.seg[sgi].dst + e] = theta[m.seg[sgi].src + e];
}
void launch_ll_slots(const float* theta, const LLSlotMap& m, float* slots, hipStream_t st) {
  hipLaunchKernelGGL(k_ll_slots, dim3(16, m.nseg), dim3(256), 0, st, theta, m, slots);
}
bool snet4_supported(const SNetArgs& a) {
  const int NBL = snet3_nbl(a.n);
  if (a.n > 128 || (NBL & 1) || a.nh < 1) return false;
  if (a.ll && a.so > 32) return false;          // the phi layer runs as two 16-output MFMA blocks
  return snet4_shmem(a, NBL) <= 160u * 1024u;
}
// bf16 elements of the packed forward / adjoint planes of ONE hidden hyper-matrix (all r+1 planes)
long snet4_fwd_elems(int n, int r) { const int NBL = snet3_nbl(n); return (long)(NBL / 2) * NBL * 3 * 64 * 8 * (r + 1); }
long snet4_bwd_elems(int n, int r) { const int NBL = snet3_nbl(n); return (long)(NBL / 2) * NBL * 2 * 64 * 8 * (r + 1); }

// plain SIREN: the act'(a) ring is not needed (the cosine's sign rides in the stashed sine, any depth / width)
bool snet4_sign_ring(const SNetArgs& a) { return !a.nif_skip; }
// does the training launch of `a` write its hidden-layer dL/da stash rows in bf16?  The kernel's own condition (`PR && NBL != 6 &&
// A.da_bf16` at the store), stated once next to it: the host derives the consumers' GwArgs.da_bf16 from THIS, not from a second copy
// of the dispatch predicate
bool snet4_writes_da_bf16(const SNetArgs& a) { return a.da_bf16 != 0 && a.prec == 1 && snet3_nbl(a.n) != 6; }
// ... and the hidden matrices' INPUT rows as 16-bit phases: the kernel's PHC condition (plain tagged-sine training form, PR = 1, NBL = 8)
bool snet4_writes_h_ph16(const SNetArgs& a) { return a.h_ph16 != 0 && a.prec == 1 && snet3_nbl(a.n) == 8 && !a.res && !a.nif_skip; }
int launch_snet4(const SNetArgs& a, bool train, bool query_only, hipStream_t st) {
  const int NBL = snet3_nbl(a.n);
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 3) / 4;
  long cap = NBL <= 4 ? 256 * NIF_S4_OCC : 256 * NIF_S4_OCC_WIDE;
  if (NBL == 8 && a.ll && train && !a.res) cap = 256 * NIF_S4_OCC_LL8;
  if (a.wg_cap > 0 && a.wg_cap < cap) cap = a.wg_cap;
  const int nblk = (int)(ngroups < cap ? ngroups : cap);
  if (query_only) return nblk;
  dim3 grid(nblk), block(256);
  const size_t shm = snet4_shmem(a, NBL);
  SNetArgs h = a;
  const int pr = snet4_pr(a);
  if (a.prec != 0) { h.WF4 = a.WF4h; h.WB4 = a.WB4h; }      // the policies' compact plane set (k_pack16b mode 1 / 2)
  if (pr == 2) {      // mixed_float16: the PR = 2 instantiations (k_snet4_f16.hip)
    launch_snet4_f16(h, train, nblk, shm, st);
    return nblk;
  }
  if (pr == 3) {      // r5: fp32-exact products on half pairs -- every SIREN form (k_snet4_x16.hip); the bf16-split forms below are class NIF's
    h.WF4 = a.WF4x; h.WB4 = a.WB4x;
    launch_snet4_x16(h, train, nblk, shm, st);
    return nblk;
  }
#define S4L(NBL_, TR_, ACT_, MODE_, SGN_, LL_, PR_)                                                                    \
  {                                                                                                                 \
    if (shm > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_, PR_>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                              \
    hipLaunchKernelGGL((k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_, PR_>), grid, block, shm, st, h);                      \
  }
// PR_: the mixed_bfloat16 policy (ONE bf16 product per n x n operand pair); plain SIREN training always takes the tagged-sine form
#define S4M(NBL_, LL_, PR_)                                                                                         \
  if (a.nif_skip) { if (train) S4L(NBL_, true, -1, 2, false, LL_, PR_) else S4L(NBL_, false, -1, 2, false, LL_, PR_) }    \
  else if (a.res) { if (train) S4L(NBL_, true, ACT_SINE, 1, true, LL_, PR_) else S4L(NBL_, false, ACT_SINE, 1, false, LL_, PR_) } \
  else if (train) S4L(NBL_, true, ACT_SINE, 0, true, LL_, PR_)                                                      \
  else S4L(NBL_, false, ACT_SINE, 0, false, LL_, PR_)
#define S4N(NBL_, LL_, PR_)   /* last-layer class: no NIF skip form */                                              \
  if (a.res) { if (train) S4L(NBL_, true, ACT_SINE, 1, true, LL_, PR_) else S4L(NBL_, false, ACT_SINE, 1, false, LL_, PR_) } \
  else if (train) S4L(NBL_, true, ACT_SINE, 0, true, LL_, PR_)                                                      \
  else S4L(NBL_, false, ACT_SINE, 0, false, LL_, PR_)
// (PR = 0, the exact bf16 splits: class NIF alone since r5 -- its activations are not bounded, the half pairs of the SIREN forms need |h| <= 1;
//  a SIREN net reaches this point with PR = 0 only when the half planes are not packed, which nif_api never does: refused below)
#define S4(NBL_)                                                            \
  if (a.ll && a.prec == 1) { S4N(NBL_, true, true) }                        \
  else if (a.prec == 1) { S4M(NBL_, false, true) }                          \
  else if (a.nif_skip) { if (train) S4L(NBL_, true, -1, 2, false, false, false) else S4L(NBL_, false, -1, 2, false, false, false) } \
  else return -1;
#ifdef NIF_S4_DEV      // ISA work (tools/isa_hist.py): only the benchmark shape's training / inference instantiations
  if (train) S4L(4, true, ACT_SINE, 0, true, false, false) else S4L(4, false, ACT_SINE, 0, false, false, false)
#else
  switch (NBL) {
    case 2: S4(2) break;
    case 4: S4(4) break;
    case 6: S4(6) break;
    default: S4(8) break;
  }
#endif
#undef S4
#undef S4N
#undef S4M
#undef S4L
  return nblk;
}
