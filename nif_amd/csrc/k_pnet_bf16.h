// k_pnet_bf16.h -- the 32x32 dense products of a narrow ParameterNet (nst <= 32) on the bf16 matrix cores.
//
// Same exact-split scheme as the ShapeNet (k_snet3_dev.h), on v_mfma_f32_32x32x16_bf16 and the 32-point
// register tile of k_pnet / k_pnet_bwg (lane (p, hf): register e = feature fmap(e, hf) of point p):
//   forward  T[o]  = sum_f W[f][o] h[f]    three-way splits of h and W, six products (small terms first)
//   adjoint  U[f]  = sum_o W[f][o] ga[o]   two-way splits, three products
// A C/D tile is directly the next B operand: MFMA step m (K = 16) takes registers 8m..8m+7, i.e. K slot (hf, t) of
// step m is feature 16 m + 8 (t >> 2) + 4 hf + (t & 3).  The A operands (weights) are split ONCE per workgroup into
// LDS planes [term][m][64 lanes] of bf16x8 by pbf_build (f32-input MFMAs were ~half of k_pnet_bwg's matrix-pipe
// time: 16 x 64 cycles per product instead of 12 x 32 / 6 x 32).
#pragma once
#include "nif_internal.h"

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
#define PBF_FWD_U4 (3 * 2 * 64)   // 16-byte units of a forward plane set (hi, mid, lo)
#define PBF_BWD_U4 (2 * 2 * 64)   // adjoint plane set (hi, lo)

__device__ __forceinline__ int pbf_feat(int m, int hf, int t) { return 16 * m + 8 * (t >> 2) + 4 * hf + (t & 3); }

// planes of ONE nst x nst matrix W[in][out] (row-major at theta + w_off): fwd [3][2][64], bwd [2][2][64] bf16x8 (or null)
__device__ __forceinline__ void pbf_build(pbf16x8* fwd, pbf16x8* bwd, const float* __restrict__ theta, long w_off, int nst,
                                          int tid, int nthreads) {
  __bf16* f16 = reinterpret_cast<__bf16*>(fwd);
  __bf16* b16 = reinterpret_cast<__bf16*>(bwd);
  for (int e = tid; e < 2 * 64 * 8; e += nthreads) {
    const int t = e & 7, lane = (e >> 3) & 63, m = e >> 9;
    const int row = lane & 31, kf = pbf_feat(m, lane >> 5, t);
    {  // forward: A[o = row][k = in feature kf] = W[kf][row]
      const float w = (kf < nst && row < nst) ? theta[w_off + (long)kf * nst + row] : 0.f;
      const __bf16 w0 = (__bf16)w;
      const float r1 = w - (float)w0;
      const __bf16 w1 = (__bf16)r1;
      f16[((0 * 2 + m) * 64 + lane) * 8 + t] = w0;
      f16[((1 * 2 + m) * 64 + lane) * 8 + t] = w1;
      f16[((2 * 2 + m) * 64 + lane) * 8 + t] = (__bf16)(r1 - (float)w1);
    }
    if (bwd) {  // adjoint: A[f = row][k = out feature kf] = W[row][kf]
      const float w = (kf < nst && row < nst) ? theta[w_off + (long)row * nst + kf] : 0.f;
      const __bf16 w0 = (__bf16)w;
      b16[((0 * 2 + m) * 64 + lane) * 8 + t] = w0;
      b16[((1 * 2 + m) * 64 + lane) * 8 + t] = (__bf16)(w - (float)w0);
    }
  }
}

// T = W^T h (forward), 6 products; fwd = the matrix's [3][2][64] plane set in LDS
__device__ __forceinline__ void pbf_dense_fwd(const pbf16x8* fwd, const f32x16& h, f32x16& T, int lane) {
  pbf16x8 x0[2], x1[2], x2[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x = h[8 * m + t];
      const __bf16 a = (__bf16)x;
      const float r1 = x - (float)a;
      const __bf16 b = (__bf16)r1;
      x0[m][t] = a; x1[m][t] = b; x2[m][t] = (__bf16)(r1 - (float)b);
    }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const pbf16x8 w0a = fwd[(0 * 2 + 0) * 64 + lane], w0b = fwd[(0 * 2 + 1) * 64 + lane];
  const pbf16x8 w1a = fwd[(1 * 2 + 0) * 64 + lane], w1b = fwd[(1 * 2 + 1) * 64 + lane];
  const pbf16x8 w2a = fwd[(2 * 2 + 0) * 64 + lane], w2b = fwd[(2 * 2 + 1) * 64 + lane];
  // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1a, x1[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1b, x1[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2a, x0[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2b, x0[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x2[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x2[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1a, x0[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1b, x0[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x1[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x1[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x0[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x0[1], acc, 0, 0, 0);
  T = acc;
}

// U = W ga (adjoint), 3 products; bwd = the matrix's [2][2][64] plane set in LDS
__device__ __forceinline__ void pbf_dense_bwd(const pbf16x8* bwd, const f32x16& ga, f32x16& U, int lane) {
  pbf16x8 x0[2], x1[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x = ga[8 * m + t];
      const __bf16 a = (__bf16)x;
      x0[m][t] = a; x1[m][t] = (__bf16)(x - (float)a);
    }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const pbf16x8 w0a = bwd[(0 * 2 + 0) * 64 + lane], w0b = bwd[(0 * 2 + 1) * 64 + lane];
  const pbf16x8 w1a = bwd[(1 * 2 + 0) * 64 + lane], w1b = bwd[(1 * 2 + 1) * 64 + lane];
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1a, x0[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1b, x0[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x1[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x1[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x0[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x0[1], acc, 0, 0, 0);
  U = acc;
}

// ---- wider ParameterNets (nst <= 32 NB, r3): the same scheme per pair of 32-feature blocks.  fwd: [ob][ib] plane sets of
// PBF_FWD_U4 units.  An activation block is split once and used against every output block.
template <int NB>
__device__ __forceinline__ void pbfn_build_fwd(pbf16x8* fwd, const float* __restrict__ theta, long w_off, int nst, int tid, int nthreads) {
  __bf16* f16 = reinterpret_cast<__bf16*>(fwd);
  for (int e = tid; e < NB * NB * 1024; e += nthreads) {
    const int t = e & 7, lane = (e >> 3) & 63, m = (e >> 9) & 1, blk = e >> 10;
    const int ob = blk / NB, ib = blk - ob * NB;
    const int row = 32 * ob + (lane & 31), kf = 32 * ib + pbf_feat(m, lane >> 5, t);
    const float w = (kf < nst && row < nst) ? theta[w_off + (long)kf * nst + row] : 0.f;
    const __bf16 w0 = (__bf16)w;
    const float r1 = w - (float)w0;
    const __bf16 w1 = (__bf16)r1;
    __bf16* q = f16 + (long)blk * PBF_FWD_U4 * 8;
    q[((0 * 2 + m) * 64 + lane) * 8 + t] = w0;
    q[((1 * 2 + m) * 64 + lane) * 8 + t] = w1;
    q[((2 * 2 + m) * 64 + lane) * 8 + t] = (__bf16)(r1 - (float)w1);
  }
}
template <int NB>
__device__ __forceinline__ void pbfn_dense_fwd(const pbf16x8* fwd, const f32x16 (&h)[NB], f32x16 (&T)[NB], int lane) {
  pbf16x8 x0[NB][2], x1[NB][2], x2[NB][2];
#pragma unroll
  for (int ib = 0; ib < NB; ++ib)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float x = h[ib][8 * m + t];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        x0[ib][m][t] = a; x1[ib][m][t] = b; x2[ib][m][t] = (__bf16)(r1 - (float)b);
      }
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      const pbf16x8* q = fwd + (long)(ob * NB + ib) * PBF_FWD_U4;
      const pbf16x8 w0a = q[(0 * 2 + 0) * 64 + lane], w0b = q[(0 * 2 + 1) * 64 + lane];
      const pbf16x8 w1a = q[(1 * 2 + 0) * 64 + lane], w1b = q[(1 * 2 + 1) * 64 + lane];
      const pbf16x8 w2a = q[(2 * 2 + 0) * 64 + lane], w2b = q[(2 * 2 + 1) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1a, x1[ib][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1b, x1[ib][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2a, x0[ib][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2b, x0[ib][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x2[ib][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x2[ib][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1a, x0[ib][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1b, x0[ib][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x1[ib][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x1[ib][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0a, x0[ib][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b, x0[ib][1], acc, 0, 0, 0);
    }
    T[ob] = acc;
  }
}
