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
#include "k_snet3_dev.h"

#ifndef NIF_S4_OCC
#define NIF_S4_OCC 3
#endif
#ifndef NIF_S4_OCC_WIDE
#define NIF_S4_OCC_WIDE 2   // workgroups per CU of the 96- and 128-wide instantiations: 2 x 256 registers with ~200 spilled beat 1 x 512 (cfg-3: 3.45 -> 2.79 ms)
#endif

// ---- packing ------------------------------------------------------------------------------------
// K-slot (g, t), g = lane >> 4, t = 0..7 of K-step ks  <->  feature 16*(2ks + (t >> 2)) + 4g + (t & 3): exactly what a
// lane of the previous layer's C/D tile holds in blocks 2ks, 2ks+1 -- the activation tile is the B operand as is.
//   fwd chunk (plane, ks): unit ((ob*3 + s)*64 + lane), 8 bf16: split s of M[in = slot(ks,g,t)][out = 16ob + (lane&15)]
//   bwd chunk (plane, ks): unit ((ib*2 + s)*64 + lane), 8 bf16: split s of M[in = 16ib + (lane&15)][out = slot(ks,g,t)]
// blockIdx.y = matrix j of a batch of equally shaped matrices `mstride` slots apart (the hidden hyper-matrices)
__global__ void k_pack16b(const float* __restrict__ theta, MatRef m, long mstride, int NBL, __bf16* __restrict__ WF,
                          __bf16* __restrict__ WB, long fstride, long bstride) {
  m.base_k += (long)blockIdx.y * mstride; m.base_last += (long)blockIdx.y * mstride;
  WF += (long)blockIdx.y * fstride; WB += (long)blockIdx.y * bstride;
  const int NCH = NBL / 2;
  const long fwd_plane = (long)NCH * NBL * 3 * 64 * 8, bwd_plane = (long)NCH * NBL * 2 * 64 * 8;
  const long total_f = fwd_plane * (m.r + 1), total_b = bwd_plane * (m.r + 1);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_f + total_b; idx += (long)gridDim.x * blockDim.x) {
    const bool fwd = idx < total_f;
    const long e = fwd ? idx : idx - total_f;
    const int ns = fwd ? 3 : 2;
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
    const float x = (in < m.nin && out < m.nout) ? theta[matref_index(m, k, in, out)] : 0.f;
    const __bf16 x0 = (__bf16)x;
    const float r1 = x - (float)x0;
    const __bf16 x1 = (__bf16)r1;
    const __bf16 x2 = (__bf16)(r1 - (float)x1);
    (fwd ? WF : WB)[e] = s == 0 ? x0 : (s == 1 ? x1 : x2);
  }
}
void launch_pack16b(const float* theta, const MatRef& m, int NBL, void* WF, void* WB, hipStream_t st) {
  launch_pack16b_batch(theta, m, 0, 1, NBL, WF, WB, 0, 0, st);
}
void launch_pack16b_batch(const float* theta, const MatRef& m0, long mstride, int nmat, int NBL, void* WF, void* WB,
                          long fstride_elems, long bstride_elems, hipStream_t st) {
  const long total = (long)(NBL / 2) * NBL * 5 * 64 * 8 * (m0.r + 1);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_pack16b, dim3(grid, nmat), dim3(256), 0, st, theta, m0, mstride, NBL, (__bf16*)WF, (__bf16*)WB,
                     fstride_elems, bstride_elems);
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

// sum over the 16 lanes of a DPP row (= the 16 points of a tile for one feature group); result in lane 15 of the row
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}
// Edge-gradient accumulation  E[m][f] += sum_{p<16} W[m][p] * T[p][f]  for one 16-point tile:
//   T   the wave's register tile (lane (p,g): features 16b+4g+v) -- transposed through 2 KB of private LDS, 32 features at
//       a time, so that a lane (f&15, kk) reads 4 consecutive points of one feature: the B operand of v_mfma_f32_16x16x4_f32
//       with the K order  step t <-> point 4*kk + t;
//   W   at most 16 per-point weight rows m (built by the caller in the SAME K order): the A operand;
//   E   rows m < nrows of the D tile (lanes with 4*(lane>>4)+v == m) added into the wave's LDS accumulators [m][NP].
// ~45 instructions per 32 features instead of 8 DPP row reductions per value.
template <int NBL>
__device__ __forceinline__ void edge_accum(const f32x4 (&T)[NBL], const f32x4 wA /*A operand: rows m, k-steps t=0..3*/, int nrows,
                                           float* __restrict__ eacc_rows, float* __restrict__ tT, int lane) {
  const int p = lane & 15, g = lane >> 4;
#pragma unroll
  for (int half = 0; half < NBL; half += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) tT[(16 * b + 4 * g + v) * 16 + p] = T[half + b][v];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(tT + (16 * b + p) * 16 + 4 * g);   // feature 16b + (lane&15), points 4g..4g+3
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[t], bq[t], d, 0, 0, 0);
      // d[v] = E[m = 4g + v][f = 16(half+b) + p]
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (4 * g + v < nrows) eacc_rows[(4 * g + v) * (16 * NBL) + 16 * (half + b) + p] += d[v];
    }
  }
}
#define ZERO4_(x) { (x)[0] = 0.f; (x)[1] = 0.f; (x)[2] = 0.f; (x)[3] = 0.f; }
#define ZERO_T(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

// MODE: 0 = plain (NIFMultiScale without resblock), 1 = SIREN resblock, 2 = NIF skip connection
// SGN (plain SIREN only): no act'(a) ring.  The next layer's stashed input IS sin(a), so cos(a) = +-sqrt(1 - sin^2):
// only the SIGN of cos(a) is kept -- 4*NBL bits per layer in a 128-bit shift register (4 VGPRs), pushed forward,
// popped in the adjoint.  Removes 2 x 4n bytes/point/layer of write + re-read traffic.  |error| of the rebuilt
// cosine <= 2.4e-4 in the measure-zero neighbourhood of cos = 0, ~1e-7 typically: gradient-path only.
// LL: last-layer-parameterised class (model.py:1044-1068, :1219-1269): the ShapeNet is a shared-weight dense SIREN
// (r = 0, one plane per layer) whose last layer emits phi [so_u x rl]; u = Dot(phi, a) + bias with the ParameterNet
// output a; the adjoint starts from dphi = du (x) a and also yields dL/da (and dL/dlatent through the rl x rl map).
// EDGE: the first-/last-layer weight gradients are accumulated in the kernel (edge_accum) instead of being stashed for
// k_gw_first / k_gw_out.  Measured on cfg-2: -0.15 ms in the gradient kernels, +0.07 ms here (72 spilled registers at
// the 168-register budget): no net gain yet, so it is opt-in (NIF_FUSE_EDGE=1) and a separate instantiation.
template <int NBL, bool TRAIN, int ACT, int MODE, bool SGN, bool LL, bool EDGE = false, bool PR = false>
__global__ __launch_bounds__(256, (NBL <= 4 ? NIF_S4_OCC : NIF_S4_OCC_WIDE)) void k_snet4(SNetArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 256, WAVES = 4;
  constexpr int NCH = NBL / 2;                      // K-step chunks per plane
  constexpr int CF = NBL * 3 * 64, CB = NBL * 2 * 64;   // 16-byte units per forward / adjoint chunk
  constexpr int QF = (CF + NT - 1) / NT;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const int FP = stash_fp(n);                             // feature rows per stash tile (nif_internal.h)
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  bf16x8* chunks = reinterpret_cast<bf16x8*>(smem);            // 2 x CF units
  float* sm = smem + 2 * CF * 4;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  const int rl = LL ? A.rl : 0, sou = LL ? A.so_u : so;
  // per-wave input rows [column][16 points] of the tile: coordinates, latent (ParameterNet output), targets, sample
  // weight -- each group padded to 4 columns = one 64-lane LDS-DMA instruction; two sets: the NEXT tile's inputs are
  // fetched by DMA while the current tile computes, so no global-load latency sits in the tile's critical path
  const int nz = LL ? rl : r;
  const int CX = (si + 3) & ~3, CZ = (nz + 3) & ~3, CY = (sou + 3) & ~3;
  const int NI = (CX + CZ + CY + 4) * 16;
  constexpr bool edge = EDGE && TRAIN && !LL;            // first/last-layer weight gradients accumulated here
  const int NE = edge ? A.edge_ne : 0;
  const int pw = 2 * r * 64 + (LL ? (rl + so + sou) * 16 : 0) + 2 * NI + NE + (edge ? 512 + so * 16 : 0);   // per-wave LDS floats
  float* dzs = sm + sm_tot + (long)wid * pw;
  float* sks = dzs + r * 64;
  float* phis = sks + r * 64;       // LL: phi / dphi [so][16], dL/da [rl][16], du [sou][16]
  float* das = phis + (LL ? so * 16 : 0);
  float* dul = das + rl * 16;
  float* inp = dul + (LL ? sou * 16 : 0);
  float* eacc = inp + 2 * NI;        // edge-gradient accumulators of this wave (whole kernel)
  float* tT = eacc + NE;             // 32 x 16 transposition scratch
  float* dus = tT + 512;             // dL/du of the tile [so][16]
  if (edge)
    for (int e = lane; e < NE; e += 64) eacc[e] = 0.f;
  const int e_l = (r + 1) * (si + 1) * (16 * NBL);          // start of the last-layer part
  const int e_b = e_l + (r + 1) * so * (16 * NBL);          // start of the last-layer bias part
  float* lsum = sm + sm_tot + (long)WAVES * pw;
  const int o_llb = LL ? ((nsm - ((sou + 3) & ~3) - ((rl * rl + 3) & ~3))) : 0;   // LL extras sit at the end of sm
  const int o_lw = o_llb + ((sou + 3) & ~3);
  constexpr int NP = 16 * NBL;
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  const int NPL = nh * (r + 1);
  const int nfwd = NPL * NCH;
  constexpr int PHF = 2 * 3 * 64;                       // units of a phi-layer forward chunk (LL)
  const int nphi = LL ? NCH + (TRAIN ? 1 : 0) : 0;      // LL: phi forward chunks, then its adjoint chunk, sit between the sweeps
  const int nchunks = (TRAIN ? 2 * nfwd : nfwd) + nphi;
  const bf16x8* WF = reinterpret_cast<const bf16x8*>(A.WF4);
  const bf16x8* WB = reinterpret_cast<const bf16x8*>(A.WB4);
  auto chunk_units = [&](int i) -> int { return i < nfwd ? CF : ((LL && i < nfwd + NCH) ? PHF : CB); };
  auto chunk_src = [&](int i) -> const bf16x8* {
    if (i < nfwd) return WF + (long)i * CF;
    if (LL && i < nfwd + NCH) return reinterpret_cast<const bf16x8*>(A.WPF) + (long)(i - nfwd) * PHF;
    if (LL && i == nfwd + NCH) return reinterpret_cast<const bf16x8*>(A.WPB);
    const int ii = i - nfwd - nphi;
    const int pp = ii / NCH, ks = ii - pp * NCH;
    const int j = nh - 1 - pp / (r + 1), k = pp % (r + 1);
    return WB + (((long)j * (r + 1) + k) * NCH + ks) * CB;
  };
  auto dma = [&](int i, int buf) {
    const bf16x8* src = chunk_src(i);
    bf16x8* dst = chunks + buf * CF;
    const int nun = chunk_units(i);
#pragma unroll
    for (int q = 0; q < QF; ++q)
      if (wid * 64 + NT * q < nun)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid + NT * q),
                                         (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);
  };
  auto prefetch_inputs = [&](long tgn, int set) {
    long t16n = tgn * WAVES + wid;
    if (t16n >= nt16) t16n = nt16 - 1;
    const long tile32n = t16n >> 1;
    const int poffn = 16 * (int)(t16n & 1) + p;
    long ptn = t16n * 16 + p;
    if (ptn >= A.B) ptn = A.B - 1;
    float* dst = inp + set * NI;
    for (int i0 = 0; i0 < CX; i0 += 4) {
      const int c = i0 + g < si ? i0 + g : si - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.xin + ptn * A.ncol + A.col0 + c),
                                       (__attribute__((address_space(3))) void*)(dst + i0 * 16), 4, 0, 0);
    }
    for (int i0 = 0; i0 < CZ; i0 += 4) {
      const int c = i0 + g < nz ? i0 + g : nz - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + (tile32n * nz + c) * 32 + poffn),
                                       (__attribute__((address_space(3))) void*)(dst + (CX + i0) * 16), 4, 0, 0);
    }
    if (TRAIN) {
      for (int i0 = 0; i0 < CY; i0 += 4) {
        const int c = i0 + g < sou ? i0 + g : sou - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.y + ptn * sou + c),
                                         (__attribute__((address_space(3))) void*)(dst + (CX + CZ + i0) * 16), 4, 0, 0);
      }
      const float* swp = A.sw ? A.sw + ptn : A.y + ptn * sou;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)swp,
                                       (__attribute__((address_space(3))) void*)(dst + (CX + CZ + CY) * 16), 4, 0, 0);
    }
  };
  {
    const long s_wl = (long)si * n + (long)nh * n * n;
    const long s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
    prefetch_inputs(blockIdx.x, 0);
    for (int idx = tid; idx < (r + 1) * nsm; idx += NT) {
      const int k = idx / nsm, e = idx - k * nsm;
      float v = 0.f;
      if (e < o_wl) { const int dd = e / NP, f = e - dd * NP; if (f < n) v = hyp3(A, k, (long)dd * n + f); }
      else if (e < o_b1) { const int o = (e - o_wl) / NP, f = (e - o_wl) - o * NP; if (f < n) v = hyp3(A, k, s_wl + (long)f * so + o); }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = hyp3(A, k, s_b1 + f); }
      else if (e < o_bl) { const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP; if (f < n) v = hyp3(A, k, s_bh + (long)j * n + f); }
      else if (e < o_bl + so) v = hyp3(A, k, s_bl + (e - o_bl));
      else if (LL && e >= o_llb && e < o_llb + sou) v = hyp3(A, k, s_bl + so + (e - o_llb));
      else if (LL && e >= o_lw && e < o_lw + rl * rl) v = hyp3(A, k, s_bl + so + sou + (e - o_lw));
      sm[idx] = v;
    }
    if (nchunks > 0) dma(0, 0);
  }
  __syncthreads();
  int gpar = 0;
  int tlc = 0; (void)tlc;
#ifdef NIF_TIMELINE
#define NIF_TL(id) do { if (A.tl && blockIdx.x == 0 && tid == 0 && tlc < 250) { A.tl[2 * tlc] = (id); A.tl[2 * tlc + 1] = (long long)__builtin_amdgcn_s_memtime(); ++tlc; } } while (0)
#else
#define NIF_TL(id) do { } while (0)
#endif
  float loss_lane = 0.f;
  float* dring = (TRAIN && !SGN) ? A.dring + ((long)blockIdx.x * WAVES + wid) * (long)(nh + 1) * (NBL * 256) : nullptr;
  constexpr int SW = 4 * NBL;   // sign bits per layer
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * A.slot_stride;

// one chunk step: start the DMA of the next chunk into the other buffer, compute on the current one, barrier
// (hipcc drains the DMA with s_waitcnt vmcnt(0) in front of the barrier).  Tried and not kept (cfg-2, 1.11 ms): whole
// planes per step (half the barriers, 54 KB LDS, still 3 workgroups/CU): 1.40 ms; 12-wave workgroups (a third of
// the L2->LDS plane traffic): 1.13 ms.  Ablations: no DMA -8 %, no barrier -3 %, no MFMA/LDS reads -13 %, no stash
// stores -10 %, no activation -5 %: a latency chain with no dominant term (3 waves/SIMD at 168 VGPRs)
#define NIF_CHUNK(...)                                                        \
  {                                                                           \
    if ((cc + 1 < nchunks) || !last_group) dma(cc + 1 < nchunks ? cc + 1 : 0, (gpar + 1) & 1); \
    const bf16x8* cur = chunks + (gpar & 1) * CF;                             \
    __VA_ARGS__                                                               \
    __syncthreads();                                                          \
    ++gpar; ++cc;                                                             \
  }

  int iset = 0;
  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++iset) {
    const bool last_group = tg + gridDim.x >= ngroups;
    const long t16_raw = tg * WAVES + wid;
    const bool active = t16_raw < nt16;
    const long t16 = active ? t16_raw : nt16 - 1;
    const long tile32 = t16 >> 1;
    const int poff = 16 * (int)(t16 & 1) + p;
    const long pt = t16 * 16 + p;
    const bool valid = active && pt < A.B;
    const float* xs = inp + (iset & 1) * NI + p;        // x_d = xs[d*16], fetched during the previous tile
    const float* zs = inp + (iset & 1) * NI + CX * 16;  // latent rows [k][16]
    const float* zl = zs;
    const float* ys = zs + CZ * 16 + p;                 // y_o = ys[o*16]
    const float* wsp = zs + (CZ + CY) * 16 + p;
    const float* zt_base = zs + p;
    const long row0 = tile32 * (long)FP * 32 + poff;
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    NIF_TL(1);
    f32x4 h[NBL], acc[NBL];
    unsigned long long sg_lo = 0ull, sg_hi = 0ull;
    // ---- first layer ---------------------------------------------------------------------------
    ZERO_T(acc)
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 16] : 1.0f;
      const float* s0 = sm + k * nsm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        acc[b] += zt * (A.omega * s + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b));
      }
    }
    {
      f32x4 d[NBL];
      if (TRAIN && SGN) sine16_sign<NBL>(acc, h, d);    // only the sign of cos(a) is kept: no v_cos_f32
      else act16<NBL, ACT>(A.act, acc, h, d, n, g);
      if (TRAIN && SGN) sgn_push(sg_lo, sg_hi, sgn_pack<NBL>(d), SW);
      if (TRAIN && !SGN) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) reinterpret_cast<f32x4*>(dring)[b * 64 + lane] = d[b];
      }
    }
    NIF_TL(2);
    prefetch_inputs(tg + gridDim.x, (iset + 1) & 1);   // lands behind the hidden-layer barriers of THIS tile
    // ---- hidden hyper-matrices -------------------------------------------------------------------
    int cc = 0;
    f32x4 ublk[MODE == 1 ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
      if (TRAIN && active) st_store16<NBL>(IN0 + (long)j * A.slot_stride, row0, h, g);
      bf16x8 b0[NCH], b1[NCH], b2[NCH];
      split3<NBL>(h, b0, b1, b2);
      NIF_TL(10 + j);
      ZERO_T(acc)
      for (int k = 0; k <= r; ++k) {
        if (k < r) {
          f32x4 T[NBL];
          ZERO_T(T)
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) NIF_CHUNK({ mfma_x6<NBL, PR>(cur, b0[ks], b1[ks], b2[ks], T, lane); })
          const float zt = zt_base[k * 16];
#pragma unroll
          for (int b = 0; b < NBL; ++b) acc[b] += zt * T[b];
        } else {
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) NIF_CHUNK({ mfma_x6<NBL, PR>(cur, b0[ks], b1[ks], b2[ks], acc, lane); })
        }
      }
      NIF_TL(30 + j);
#pragma unroll
      for (int b = 0; b < NBL; ++b) acc[b] *= A.omega;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
      {
        f32x4 d[NBL];
        if (TRAIN && SGN) sine16_sign<NBL>(acc, acc, d);
        else act16<NBL, ACT>(A.act, acc, acc, d, n, g);
        if (TRAIN && SGN) sgn_push(sg_lo, sg_hi, sgn_pack<NBL>(d), SW);
        if (TRAIN && !SGN) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) reinterpret_cast<f32x4*>(dring)[((j + 1) * NBL + b) * 64 + lane] = d[b];
        }
      }
      if (MODE == 0) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) h[b] = acc[b];
      } else if (MODE == 2) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) h[b] += acc[b];
      } else {
        if (!(j & 1)) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { ublk[b] = h[b]; h[b] = acc[b]; }
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) h[b] = 0.5f * (ublk[b] + acc[b]);
        }
      }
    }
    NIF_TL(3);
    // ---- last layer (n -> so, linear), MSE, start of the adjoint ---------------------------------
    if (TRAIN && active && !edge) st_store16<NBL>(IN0 + (long)nh * A.slot_stride, row0, h, g);
    f32x4 gh[NBL];
    ZERO_T(gh)
    const float wsamp = (valid ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
    float se = 0.f;
    if (LL) {
      // phi[o] = <h, Wl[:, o]> + bl[o] into the wave's LDS row, then per point u = Dot(phi, a) + bias
      // phi = Wl^T h on the matrix cores (two 16-output blocks, the 6-product form), into the wave's LDS rows
      {
        bf16x8 b0[NCH], b1[NCH], b2[NCH];
        split3<NBL>(h, b0, b1, b2);
        f32x4 T2[2];
        ZERO4_(T2[0]) ZERO4_(T2[1])
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) NIF_CHUNK({ mfma_x6<2>(cur, b0[ks], b1[ks], b2[ks], T2, lane); })
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int o = 16 * b + 4 * g + v;
            if (o < so) phis[o * 16 + p] = T2[b][v] + sm[o_bl + o];
          }
      }
      for (int s_ = 0; s_ < sou; ++s_) {
        float uo = sm[o_llb + s_];
        for (int j = 0; j < rl; ++j) uo = fmaf(phis[(s_ * rl + j) * 16 + p], zl[j * 16 + p], uo);
        if (valid && g == 0 && A.u_out) A.u_out[pt * sou + s_] = uo;
        if (TRAIN) {
          const float e = uo - ys[s_ * 16];
          se = fmaf(e, e, se);
          const float du = 2.0f * wsamp * e * A.inv_bg / (float)sou;
          if (g == 0) {
            dul[s_ * 16 + p] = du;
            if (active) A.DU[(tile32 * sou + s_) * 32 + poff] = du;
          }
        }
      }
      if (TRAIN) {
        // dL/da[j] = sum_s du[s] phi[s][j]  (lane group g takes j = g, g+4, ..), then dL/dlatent through the rl x rl map
        for (int j = g; j < rl; j += 4) {
          float da = 0.f;
          for (int s_ = 0; s_ < sou; ++s_) da = fmaf(dul[s_ * 16 + p], phis[(s_ * rl + j) * 16 + p], da);
          das[j * 16 + p] = da;
          if (active) A.DA_ll[(tile32 * rl + j) * 32 + poff] = da;
        }
        for (int k = g; k < rl; k += 4) {
          float dz = 0.f;
          for (int c = 0; c < rl; ++c) dz = fmaf(das[c * 16 + p], sm[o_lw + k * rl + c], dz);
          if (active) A.DZL[(tile32 * rl + k) * 32 + poff] = dz;
        }
        // dphi[o] = du[s] a[j] replaces phi in LDS; it is also the "dL/dout" stash of the phi layer's weight gradient
        for (int s_ = 0; s_ < sou; ++s_) {
          const float du = dul[s_ * 16 + p];
          for (int j = g; j < rl; j += 4) {
            const int o = s_ * rl + j;
            const float dq = du * zl[j * 16 + p];
            phis[o * 16 + p] = dq;
            if (active) A.DPHI[(tile32 * so + o) * 32 + poff] = dq;
          }
        }
        // dL/dh = Wl dphi: one adjoint chunk (K = the 32 padded outputs), 3-product form
        {
          f32x4 dq2[2];
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int o = 16 * b + 4 * g + v;
              dq2[b][v] = o < so ? phis[o * 16 + p] : 0.f;
            }
          bf16x8 d0[1], d1[1];
          split2<2>(dq2, d0, d1);
          NIF_CHUNK({ mfma_x3<NBL>(cur, d0[0], d1[0], gh, lane); })
        }
      }
    } else
    for (int o = 0; o < so; ++o) {
      f32x4 wg[NBL];
      ZERO_T(wg)
      float part = 0.f, bias = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* s0 = sm + k * nsm;
        float sk = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
          sk += (h[b][0] * w[0] + h[b][1] * w[1]) + (h[b][2] * w[2] + h[b][3] * w[3]);
          if (TRAIN) wg[b] += zt * w;
        }
        part = fmaf(zt, sk, part);
        bias = fmaf(zt, s0[o_bl + o], bias);
        if (TRAIN && k < r) sks[k * 64 + lane] = sk;
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      const float uo = part + bias;
      if (valid && g == 0 && A.u_out) A.u_out[pt * so + o] = uo;
      if (TRAIN) {
        const float e = uo - ys[o * 16];
        se = fmaf(e, e, se);
        const float du = 2.0f * wsamp * e * A.inv_bg / (float)so;
        if (!edge && active && g == 0) A.DU[(tile32 * so + o) * 32 + poff] = du;
        if (edge && g == 0) dus[o * 16 + p] = du;     // dL/du of this tile, for the edge accumulation below
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] += du * wg[b];
        for (int k = 0; k < r; ++k) {
          float t = du * sks[k * 64 + lane];
          if (g == 0) t = fmaf(du, sm[k * nsm + o_bl + o], t);
          dzs[k * 64 + lane] += t;
        }
      }
    }
    if (edge) {
      // dL/dWl^(k)[f][o] += sum_p zt_k du_o h[f] ,  dL/dbl^(k)[o] += sum_p zt_k du_o :  rows m = k*so + o
      const int nrow = (r + 1) * so;
      const int m = p, kk_ = g;                      // A operand: lane (row m, k-group): points 4kk..4kk+3
      f32x4 wA = {0.f, 0.f, 0.f, 0.f};
      if (m < nrow) {
        const int k = m / so, o = m - k * so;
#pragma unroll
        for (int t = 0; t < 4; ++t) wA[t] = (k < r ? zs[k * 16 + 4 * kk_ + t] : 1.0f) * dus[o * 16 + 4 * kk_ + t];
      }
      edge_accum<NBL>(h, wA, nrow, eacc + e_l, tT, lane);
      float sb_ = (wA[0] + wA[1]) + (wA[2] + wA[3]);
      sb_ += __shfl_xor(sb_, 16);
      sb_ += __shfl_xor(sb_, 32);
      if (g == 0 && m < nrow) eacc[e_b + m] += sb_;
    }
    if (TRAIN) {
      if (g == 0) loss_lane += wsamp * se / (float)sou * A.inv_bg;
      NIF_TL(4);
      // ---- adjoint through the hidden hyper-matrices --------------------------------------------
      f32x4 skip[MODE == 0 ? 1 : NBL];
      f32x4 dnext[NBL], hin[NBL];
      if (SGN) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) hin[b] = h[b];     // sin(a) of the top hidden layer is the last layer's input
      }
      for (int j = nh - 1; j >= 0; --j) {
        f32x4 ga[NBL];
        if (SGN) {
          sgn_cos<NBL>(hin, sgn_pop(sg_lo, sg_hi, SW), dnext);
          st_load16<NBL>(IN0 + (long)j * A.slot_stride, row0, hin, g);   // h_j: dz dot product now, sin(a) of layer j-1 next
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[((j + 1) * NBL + b) * 64 + lane];
        }
        if (MODE == 1 && (j & 1)) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { skip[b] = 0.5f * gh[b]; ga[b] = dnext[b] * skip[b]; }
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
          if (MODE == 2) {
#pragma unroll
            for (int b = 0; b < NBL; ++b) skip[b] = gh[b];
          }
        }
        if (active) st_store16<NBL>(DA0 + (long)(j + 1) * A.slot_stride, row0, ga, g);
        // <dL/da, b^(k)> now, so that dL/da is dead once it is split and stashed (16 registers less across the planes)
        for (int k = 0; k < r; ++k) {
          const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
          float sbv = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
            sbv += (ga[b][0] * bb[0] + ga[b][1] * bb[1]) + (ga[b][2] * bb[2] + ga[b][3] * bb[3]);
          }
          dzs[k * 64 + lane] += sbv;
        }
        bf16x8 b0[NCH], b1[NCH];
        split2<NBL>(ga, b0, b1);
        NIF_TL(50 + j);
        ZERO_T(gh)
        for (int k = 0; k <= r; ++k) {
          if (k < r) {
            f32x4 U[NBL];
            ZERO_T(U)
#pragma unroll
            for (int ks = 0; ks < NCH; ++ks)
              NIF_CHUNK({
                mfma_x3<NBL, PR>(cur, b0[ks], b1[ks], U, lane);
                if (!SGN && ks == 0) st_load16<NBL>(IN0 + (long)j * A.slot_stride, row0, hin, g);
              })
            const float zt = zt_base[k * 16];
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < NBL; ++b) {
              gh[b] += zt * U[b];
#pragma unroll
              for (int v = 0; v < 4; ++v) s = fmaf(hin[b][v], U[b][v], s);
            }
            dzs[k * 64 + lane] = fmaf(A.omega, s, dzs[k * 64 + lane]);
          } else {
#pragma unroll
            for (int ks = 0; ks < NCH; ++ks) NIF_CHUNK({ mfma_x3<NBL, PR>(cur, b0[ks], b1[ks], gh, lane); })
          }
        }
        NIF_TL(70 + j);
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          gh[b] *= A.omega;
          if (MODE == 2 || (MODE == 1 && !(j & 1))) gh[b] += skip[b];
        }
      }
      NIF_TL(5);
      // ---- first layer ---------------------------------------------------------------------------
      {
        f32x4 ga[NBL];
        if (SGN) {
          sgn_cos<NBL>(hin, sgn_pop(sg_lo, sg_hi, SW), dnext);
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[b * 64 + lane];
        }
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
        if (active && !edge) st_store16<NBL>(DA0, row0, ga, g);
        if (edge) {
          // dL/dW1^(k)[d][f] += sum_p zt_k x_d da0[f] (w0 is applied in k_reduce_edge); row d = si is the bias: m = k*(si+1) + d
          const int nrow = (r + 1) * (si + 1);
          const int m = p, kk_ = g;
          f32x4 wA = {0.f, 0.f, 0.f, 0.f};
          if (m < nrow && active) {
            const int k = m / (si + 1), dd = m - k * (si + 1);
            const float* xr = inp + (iset & 1) * NI;
#pragma unroll
            for (int t = 0; t < 4; ++t)
              wA[t] = (k < r ? zs[k * 16 + 4 * kk_ + t] : 1.0f) * (dd < si ? xr[dd * 16 + 4 * kk_ + t] : 1.0f);
          }
          edge_accum<NBL>(ga, wA, nrow, eacc, tT, lane);
        }
        for (int k = 0; k < r; ++k) {
          const float* s0 = sm + k * nsm + 4 * g;
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            f32x4 xw = {0.f, 0.f, 0.f, 0.f};
            for (int dd = 0; dd < si; ++dd) xw += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
            const f32x4 t = A.omega * xw + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
            s += (ga[b][0] * t[0] + ga[b][1] * t[1]) + (ga[b][2] * t[2] + ga[b][3] * t[3]);
          }
          float tot = dzs[k * 64 + lane] + s;
          tot += __shfl_xor(tot, 16);
          tot += __shfl_xor(tot, 32);
          if (active && g == 0) A.DZ[(tile32 * r + k) * 32 + poff] = tot;
        }
      }
    }
  }
#undef NIF_CHUNK
  if (TRAIN) {
    for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
    if (lane == 0) lsum[wid] = loss_lane;
    __syncthreads();
    if (edge) {   // the four waves' accumulators -> this workgroup's compact partial (fixed order)
      const float* e0 = sm + sm_tot + (pw - NE - 512 - so * 16);
      for (int e = tid; e < NE; e += NT)
        A.EDGE[(long)blockIdx.x * NE + e] = (e0[e] + e0[pw + e]) + (e0[2 * pw + e] + e0[3 * pw + e]);
    }
    if (tid == 0) A.loss_partial[blockIdx.x] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  }
}

// ---- host side ---------------------------------------------------------------------------------
static size_t snet4_shmem(const SNetArgs& a, int NBL) {
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const int nz = a.ll ? a.rl : a.r, sou = a.ll ? a.so_u : a.so;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + ((nz + 3) & ~3) + ((sou + 3) & ~3) + 4) * 16;
  const size_t pw = 2 * a.r * 64 + (a.ll ? (size_t)(a.rl + a.so + a.so_u) * 16 : 0) + 2 * ni + (a.EDGE ? a.edge_ne + 512 + a.so * 16 : 0);
  return (size_t)2 * NBL * 3 * 64 * 16 + (sm_tot + 4 * pw + 8) * sizeof(float);
}
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

bool snet4_sign_ring(const SNetArgs& a);
int snet4_edge_ne(const SNetArgs& a) {
  if (a.ll || !snet4_sign_ring(a)) return 0;     // built for the plain-SIREN training instantiation
  const int NP = 16 * snet3_nbl(a.n);
  const long ne = (long)(a.r + 1) * (a.si + 1) * NP + (long)(a.r + 1) * a.so * NP + (long)(a.r + 1) * a.so;
  const long ne4 = (ne + 3) & ~3L;
  if ((a.r + 1) * (a.si + 1) > 16 || (a.r + 1) * a.so > 16) return 0;   // rows of one 16-row MFMA operand
  return ne4 <= 1024 ? (int)ne4 : 0;       // 4 waves x 4 KB of LDS at most
}
// edge partials [nblk][ne] -> the first-/last-layer entries of the flat gradient (fixed summation order)
__global__ __launch_bounds__(256) void k_reduce_edge(SNetArgs A, const float* __restrict__ edge, int nblk, int ne, int NP,
                                                     float* __restrict__ g) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  float s = 0.f;
  if (e < ne)
    for (int b = rg; b < nblk; b += 4) s += edge[(long)b * ne + e];
  red[rg][col] = s;
  __syncthreads();
  if (rg != 0 || e >= ne) return;
  const float v = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
  const int r = A.r, si = A.si, so = A.so, n = A.n;
  const long s_wl = (long)si * n + (long)A.nh * n * n, s_b1 = s_wl + (long)n * so, s_bl = s_b1 + n + (long)A.nh * n;
  const int e_l = (r + 1) * (si + 1) * NP, e_b = e_l + (r + 1) * so * NP;
  auto base = [&](int k) -> long { return k < r ? A.off_Wh + (long)k * A.po : A.off_bh; };
  if (e < e_l) {
    const int f = e % NP, kd = e / NP, k = kd / (si + 1), dd = kd % (si + 1);
    if (f < n) {
      if (dd < si) g[base(k) + (long)dd * n + f] = A.omega * v;
      else g[base(k) + s_b1 + f] = v;
    }
  } else if (e < e_b) {
    const int e2 = e - e_l, f = e2 % NP, ko = e2 / NP, k = ko / so, o = ko % so;
    if (f < n) g[base(k) + s_wl + (long)f * so + o] = v;
  } else if (e < e_b + (r + 1) * so) {
    const int ko = e - e_b, k = ko / so, o = ko % so;
    g[base(k) + s_bl + o] = v;
  }
}
void launch_reduce_edge(const SNetArgs& a, const float* edge, int nblk, float* grad, hipStream_t st) {
  const int ne = a.edge_ne;
  hipLaunchKernelGGL(k_reduce_edge, dim3((ne + 63) / 64), dim3(256), 0, st, a, edge, nblk, ne, 16 * snet3_nbl(a.n), grad);
}
// plain SIREN whose sign bits fit the 128-bit shift register: the act'(a) ring is not needed
bool snet4_sign_ring(const SNetArgs& a) {
  return !a.nif_skip && !a.res && (long)(a.nh + 1) * 4 * snet3_nbl(a.n) <= 128;
}
int launch_snet4(const SNetArgs& a, bool train, bool query_only, hipStream_t st) {
  const int NBL = snet3_nbl(a.n);
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 3) / 4;
  long cap = NBL <= 4 ? 256 * NIF_S4_OCC : 256 * NIF_S4_OCC_WIDE;
  if (a.wg_cap > 0 && a.wg_cap < cap) cap = a.wg_cap;
  const int nblk = (int)(ngroups < cap ? ngroups : cap);
  if (query_only) return nblk;
  dim3 grid(nblk), block(256);
  const size_t shm = snet4_shmem(a, NBL);
#define S4L(NBL_, TR_, ACT_, MODE_, SGN_, LL_)                                                                         \
  {                                                                                                                 \
    if (shm > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_>,                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                              \
    hipLaunchKernelGGL((k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, LL_>), grid, block, shm, st, a);                           \
  }
#define S4P(NBL_, TR_, ACT_, MODE_, SGN_)   /* mixed_bfloat16 policy: single bf16 product per n x n operand pair */        \
  {                                                                                                                 \
    if (shm > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, false, false, true>,                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                              \
    hipLaunchKernelGGL((k_snet4<NBL_, TR_, ACT_, MODE_, SGN_, false, false, true>), grid, block, shm, st, a);            \
  }
#define S4LP(NBL_, TR_, MODE_, SGN_)   /* last-layer class under the policy: the shared n x n products as ONE bf16 product */ \
  {                                                                                                                 \
    if (shm > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, TR_, ACT_SINE, MODE_, SGN_, true, false, true>,               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                              \
    hipLaunchKernelGGL((k_snet4<NBL_, TR_, ACT_SINE, MODE_, SGN_, true, false, true>), grid, block, shm, st, a);         \
  }
#define S4LE(NBL_)                                                                                                  \
  {                                                                                                                 \
    if (shm > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_snet4<NBL_, true, ACT_SINE, 0, true, false, true>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                              \
    hipLaunchKernelGGL((k_snet4<NBL_, true, ACT_SINE, 0, true, false, true>), grid, block, shm, st, a);             \
  }
#define S4(NBL_)                                                            \
  if (a.ll && a.prec == 1) {                                                \
    if (a.res) { if (train) S4LP(NBL_, true, 1, false) else S4LP(NBL_, false, 1, false) } \
    else if (train) { if (snet4_sign_ring(a)) S4LP(NBL_, true, 0, true) else S4LP(NBL_, true, 0, false) } \
    else S4LP(NBL_, false, 0, false)                                        \
  } else if (a.ll) {                                                        \
    if (a.res) { if (train) S4L(NBL_, true, ACT_SINE, 1, false, true) else S4L(NBL_, false, ACT_SINE, 1, false, true) } \
    else if (train) { if (snet4_sign_ring(a)) S4L(NBL_, true, ACT_SINE, 0, true, true) else S4L(NBL_, true, ACT_SINE, 0, false, true) } \
    else S4L(NBL_, false, ACT_SINE, 0, false, true)                         \
  } else if (a.prec == 1) {                                                 \
    if (a.nif_skip) { if (train) S4P(NBL_, true, -1, 2, false) else S4P(NBL_, false, -1, 2, false) } \
    else if (a.res) { if (train) S4P(NBL_, true, ACT_SINE, 1, false) else S4P(NBL_, false, ACT_SINE, 1, false) } \
    else if (train) { if (snet4_sign_ring(a)) S4P(NBL_, true, ACT_SINE, 0, true) else S4P(NBL_, true, ACT_SINE, 0, false) } \
    else S4P(NBL_, false, ACT_SINE, 0, false)                               \
  } else if (a.nif_skip) {                                                  \
    if (train) S4L(NBL_, true, -1, 2, false, false) else S4L(NBL_, false, -1, 2, false, false) \
  } else if (a.res) {                                                       \
    if (train) S4L(NBL_, true, ACT_SINE, 1, false, false) else S4L(NBL_, false, ACT_SINE, 1, false, false) \
  } else if (train) {                                                       \
    if (snet4_sign_ring(a)) { if (a.EDGE) S4LE(NBL_) else S4L(NBL_, true, ACT_SINE, 0, true, false) } \
    else S4L(NBL_, true, ACT_SINE, 0, false, false) \
  } else {                                                                  \
    S4L(NBL_, false, ACT_SINE, 0, false, false)                             \
  }
  switch (NBL) {
    case 2: S4(2) break;
    case 4: S4(4) break;
    case 6: S4(6) break;
    default: S4(8) break;
  }
#undef S4
#undef S4LE
#undef S4LP
#undef S4P
#undef S4L
  return nblk;
}
