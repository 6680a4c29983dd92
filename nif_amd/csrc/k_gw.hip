// k_gw.hip -- weight-gradient reductions over the point batch (gfx950).
//
//   C[k][in][out] = scale * sum_p zt_k[p] * IN[p][in] * DA[p][out]        zt = (z_1..z_r, 1)
//
// is the gradient of every hypernetwork slice (k < r -> row k of the hyper kernel, k = r -> the
// hyper bias) and, with r = 0, of every shared-weight dense layer.  The sum over points is the K
// dimension of v_mfma_f32_32x32x16_bf16 (both operands split into bf16 hi + lo, three products); the
// operands come from the [tile][feature][32 points] stashes: lane (i, hf) holds 16 consecutive points of
// feature i, so the K order is "hf picks the half-tile" for A and B alike -- no transpose.  The default
// kernels (k_gw_lds, k_gw_first_lds, k_gw_out_lds) bring the tiles in by LDS-DMA, one contiguous KiB per
// load instruction; the k_gw_mfma / k_gw_first* / k_gw_out* register-load forms remain for the shapes the
// DMA forms do not cover and for A/B runs (NIF_GW_LDS=0).  Each workgroup reduces a strided subset of
// tiles and writes one row of the partial buffer; k_reduce sums the rows in a fixed order (deterministic,
// no atomics).
//
// This replaces what GradientTape does for  tf.einsum('ai,aij->aj') / Dense / SIREN  weights
// (nif/layers/mlp.py:219, nif/model.py:253-300 StridedSliceGrad + AddN; SURVEY a-10).
#include "nif_internal.h"

#ifndef NIF_GW_BF16
#define NIF_GW_BF16 1
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// sum a per-lane value over the 4 waves of the block (deterministic order), result valid on wave 0
__device__ __forceinline__ float block_sum4(float v, float* red /*[3*64]*/, int wid, int lane) {
  if (wid > 0) red[(wid - 1) * 64 + lane] = v;
  __syncthreads();
  if (wid == 0) v = ((v + red[lane]) + red[64 + lane]) + red[128 + lane];
  __syncthreads();
  return v;
}

// ------------------------------------------------------------------------------------------
// n x n blocks on the matrix cores.  grid = (rows, ceil((r+1)/KC), NBO/OBC)
// ------------------------------------------------------------------------------------------
// sum a per-lane value over the WV waves of the block (fixed order), result valid on wave 0
template <int WV>
__device__ __forceinline__ float block_sum(float v, float* red /*[(WV-1)*64]*/, int wid, int lane) {
  if (wid > 0) red[(wid - 1) * 64 + lane] = v;
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int w = 0; w < WV - 1; ++w) v += red[w * 64 + lane];
  }
  __syncthreads();
  return v;
}

// the same for a whole 16-register accumulator block: 2 barriers per 16 values
template <int WV>
__device__ __forceinline__ f32x16 block_sum16(f32x16 v, float* red16 /*[(WV-1)*16*64]*/, int wid, int lane) {
  if (wid > 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red16[((wid - 1) * 16 + e) * 64 + lane] = v[e];
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int w = 0; w < WV - 1; ++w)
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += red16[(w * 16 + e) * 64 + lane];
  }
  __syncthreads();
  return v;
}

// WV waves per workgroup: with the bf16 path the kernel is a stream of loads + VALU splits + few MFMAs, and two
// 256-register waves per SIMD overlap one wave's splitting with the other's loads
template <int NBI, int OBC, int KC, int WV, bool DBUF>
__global__ __launch_bounds__(64 * WV) void k_gw_mfma(GwArgs A, int NBO) {
  __shared__ float red[(WV - 1) * 64];
  __shared__ float red16[(WV - 1) * 16 * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const int k0 = blockIdx.y * KC;
  const int ob0 = blockIdx.z * OBC;
  const long nwaves = (long)gridDim.x * WV;
  const long FI = (long)NBI * 32 * 32, FO = (long)NBO * 32 * 32;

  f32x16 acc[KC][NBI][OBC];
  float bacc[KC][OBC];
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) {
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int ob = 0; ob < OBC; ++ob)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kk][ib][ob][e] = 0.f;
#pragma unroll
    for (int ob = 0; ob < OBC; ++ob) bacc[kk][ob] = 0.f;
  }

  // register double buffering: the next tile's operands are in flight while the current tile feeds the
  // matrix cores (this kernel runs one wave per SIMD, so nothing else would hide the HBM latency)
  const int zt_mod = (int)A.zt_mod, nt_all = (int)A.ntiles;
  auto zmod = [&](long t) -> int {
    const int ti = __builtin_amdgcn_readfirstlane((int)t);
    return zt_mod >= nt_all ? ti : ti % zt_mod;
  };
  auto load_tile = [&](long t, f32x4 (&af)[NBI][4], f32x4 (&bf)[OBC][4], f32x4 (&zq)[KC][4]) {
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[ib][q] = ld4(A.IN + t * FI + (long)(32 * ib + i) * 32 + 16 * hf + 4 * q);
#pragma unroll
    for (int ob = 0; ob < OBC; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[ob][q] = ld4(A.DA + t * FO + (long)(32 * (ob0 + ob) + i) * 32 + 16 * hf + 4 * q);
    // unconditional loads (static s_waitcnt counts keep the next tile's prefetch in flight); k >= r reads a valid
    // dummy row and is replaced by ones.  Tile index of the latent: t mod zt_mod in 32-bit scalar arithmetic
    const int tz = zmod(t);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const int k = k0 + kk;
      const int kz = k < A.r ? k : (A.r > 0 ? A.r - 1 : 0);
      const float* zrow = A.Z + ((long)tz * A.r + kz) * 32 + 16 * hf;
#pragma unroll
      for (int q = 0; q < 4; ++q) zq[kk][q] = ld4(zrow + 4 * q);   // k >= r: replaced by ones at use (compute_tile)
    }
  };
#if NIF_GW_BF16
  // the K = batch GEMM on the bf16 matrix cores: both operands split into bf16 hi + lo, three products
  // (hi*hi + hi*lo + lo*hi, 1.9e-6 rms of sum|a b| -- well inside the fp32 re-association noise of a sum over
  // 10^6 points); v_mfma_f32_32x32x16_bf16 takes 8 consecutive points per lane: the two halves of a lane's 16
  auto compute_tile = [&](long t, const f32x4 (&af)[NBI][4], const f32x4 (&bf)[OBC][4], const f32x4 (&zq)[KC][4]) {
    const bool wbias = t < A.bias_ntiles;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      bf16x8 bh[OBC], bl[OBC];
#pragma unroll
      for (int ob = 0; ob < OBC; ++ob)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = bf[ob][2 * hh + (e >> 2)][e & 3];
          const __bf16 x0 = (__bf16)x;
          bh[ob][e] = x0; bl[ob][e] = (__bf16)(x - (float)x0);
        }
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        if (k0 + kk > A.r) break;
#pragma unroll
        for (int ib = 0; ib < NBI; ++ib) {
          bf16x8 ah, al;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = af[ib][2 * hh + (e >> 2)][e & 3] * (k0 + kk < A.r ? zq[kk][2 * hh + (e >> 2)][e & 3] : 1.0f);
            const __bf16 x0 = (__bf16)x;
            ah[e] = x0; al[e] = (__bf16)(x - (float)x0);
          }
#pragma unroll
          for (int ob = 0; ob < OBC; ++ob) {
            acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ob], acc[kk][ib][ob], 0, 0, 0);
            acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ob], acc[kk][ib][ob], 0, 0, 0);
            acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ob], acc[kk][ib][ob], 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 2 * hh; q < 2 * hh + 2; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ob = 0; ob < OBC; ++ob)
              bacc[kk][ob] = fmaf(wbias ? (k0 + kk < A.r ? zq[kk][q][c] : 1.0f) : 0.f, bf[ob][q][c], bacc[kk][ob]);
      }
    }
  };
#else
  auto compute_tile = [&](long t, const f32x4 (&af)[NBI][4], const f32x4 (&bf)[OBC][4], const f32x4 (&zq)[KC][4]) {
    const bool wbias = t < A.bias_ntiles;   // tangent pseudo-tiles (Sobolev) carry no bias gradient
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      if (k0 + kk > A.r) break;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float zt = k0 + kk < A.r ? zq[kk][q][c] : 1.0f;
#pragma unroll
          for (int ib = 0; ib < NBI; ++ib) {
            const float a = af[ib][q][c] * zt;
#pragma unroll
            for (int ob = 0; ob < OBC; ++ob)
              acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[ob][q][c], acc[kk][ib][ob], 0, 0, 0);
          }
#pragma unroll
          for (int ob = 0; ob < OBC; ++ob) bacc[kk][ob] = fmaf(wbias ? zt : 0.f, bf[ob][q][c], bacc[kk][ob]);
        }
    }
  };
#endif
  if (!DBUF) {
    // one register set (wide variants whose accumulators already fill the file)
    f32x4 af0[NBI][4], bf0[OBC][4], zq0[KC][4];
    for (long t = (long)blockIdx.x * WV + wid; t < A.ntiles; t += nwaves) {
      load_tile(t, af0, bf0, zq0);
      compute_tile(t, af0, bf0, zq0);
    }
  } else {
    // every load_tile is unconditional (beyond the end it re-reads the last tile): hipcc can then count the
    // outstanding loads statically and the s_waitcnt in front of a tile's compute leaves the NEXT tile's 24 loads
    // in flight -- with a conditional prefetch it has to drain them
    f32x4 af0[NBI][4], bf0[OBC][4], zq0[KC][4], af1[NBI][4], bf1[OBC][4], zq1[KC][4];
    const long last = A.ntiles - 1;
    long t = (long)blockIdx.x * WV + wid;
    load_tile(t < last ? t : last, af0, bf0, zq0);
    while (t < A.ntiles) {
      const long t1 = t + nwaves;
      load_tile(t1 < last ? t1 : last, af1, bf1, zq1);
      compute_tile(t, af0, bf0, zq0);
      if (t1 >= A.ntiles) break;
      const long t2 = t1 + nwaves;
      load_tile(t2 < last ? t2 : last, af0, bf0, zq0);
      compute_tile(t1, af1, bf1, zq1);
      t = t2;
    }
  }

  // block reduction (4 waves -> wave 0) and write of this block's partial row
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) {
    const int k = k0 + kk;
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int ob = 0; ob < OBC; ++ob) {
        // a whole 32x32 accumulator block per round: 2 barriers per 16 values instead of per value
        f32x16 v = acc[kk][ib][ob];
        if (wid > 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) red16[((wid - 1) * 16 + e) * 64 + lane] = v[e];
        }
        __syncthreads();
        if (wid == 0) {
#pragma unroll
          for (int w = 0; w < WV - 1; ++w)
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += red16[(w * 16 + e) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int in = 32 * ib + fmap(e, hf), out = 32 * (ob0 + ob) + i;
            if (k <= A.r && in < A.W.nin && out < A.W.nout) prow[matref_index(A.W, k, in, out)] = A.scale * v[e];
          }
        }
        __syncthreads();
      }
#pragma unroll
    for (int ob = 0; ob < OBC; ++ob) {
      float v = bacc[kk][ob];
      v += __shfl_xor(v, 32);
      v = block_sum<WV>(v, red, wid, lane);
      const int out = 32 * (ob0 + ob) + i;
      if (A.has_bias && wid == 0 && hf == 0 && k <= A.r && out < A.Bv.nout) prow[matref_index(A.Bv, k, 0, out)] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------
// k_gw_lds: the same K = batch GEMM with the stash tiles brought in by LDS-DMA.
//
// k_gw_mfma's register loads give lane (i, hf) 16 bytes of row i: one load instruction touches 32 cache lines for
// 32 bytes each, and the address path (not HBM) bounds the kernel at ~4 TB/s.  Here every load instruction is a
// global_load_lds of ONE contiguous KiB (8 stash rows) and the (feature, points) operand shape is recovered by
// ds_read_b128.  The DMA writes LDS lane-major, but WHICH 16-byte piece of the KiB a lane fetches is free: piece
// c of row r of chunk j goes to slot 8 r + (c ^ r ^ (j & 1)), which makes the 16 lanes of a ds_read_b128 pass hit 16
// distinct 4-bank groups.  Per tile: read tile t from LDS into registers, issue the DMA of tile t+1 into the other
// buffer, then split + MFMA tile t while that DMA is in flight.  Wave-private buffers, no barriers in the loop.
// ------------------------------------------------------------------------------------------
// Measured and not kept: one buffer per wave + two workgroups per CU (256 registers: 51 spilled, 0.27 ms instead of 0.11);
// two tiles in flight per wave (refill the buffer just read with tile t+2, s_waitcnt vmcnt(17)): 0.106 ms either way --
// at 5 TB/s the kernel sits at the read bandwidth this access pattern reaches (k_gw_out_lds, no arithmetic: 5.3 TB/s).
// NBUF = 1 (128-wide layers: 256 accumulator registers and a 24-KiB tile per wave): one buffer, refilled as soon as the
// tile sits in registers -- the DMA of tile t+1 still overlaps the whole split + MFMA phase of tile t.
// DAB (mixed_bfloat16, r3): the dL/da stash holds bf16 rows [tile][feature][32 points] of 64 B (k_snet4<PR> / k_sobw<PR> write them:
// half the bytes of this operand) and the products are the policy's: ONE bf16 product per operand pair, bf16(zt_k h_in) x dL/da.
// A 1-KiB DMA chunk is then 16 rows of 4 pieces; piece c of row r goes to slot 4 r + (c ^ (r >> 2)) -- the 16 lanes of a
// ds_read_b128 pass (rows 0..15, same piece) hit 16 distinct 4-bank groups.
template <int NBI, int OBC, int NBUF, bool DAB = false>   // KC = 2, 4 waves; grid = (rows, planes / 2, NBO / OBC)
__global__ __launch_bounds__(256) void k_gw_lds(GwArgs A, int NBO) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  constexpr int KC = 2, WV = 4;
  constexpr int TFI = NBI * 1024, TFB = OBC * (DAB ? 512 : 1024);   // floats of the operand tiles held by this workgroup
  constexpr int BUF = TFI + TFB + 64;                 // IN | DA | Z rows (2 x 32)
  const int ob0 = blockIdx.z * OBC;
  const long TFO = (long)NBO * (DAB ? 512 : 1024);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const int k0 = blockIdx.y * KC;
  const long nwaves = (long)gridDim.x * WV;
  float* wbuf = gsm + (long)wid * NBUF * BUF;
  // bf16 dL/da rows: DMA source of this lane inside a chunk (row lane >> 2, piece (lane & 3) ^ (row >> 2)), reader offsets of
  // lane (i, hf) for the two K halves (row i, piece 2 hf + hh = points 16 hf + 8 hh ..)
  const int bsrc = (lane >> 2) * 16 + (((lane & 3) ^ (lane >> 4)) & 3) * 4;
  int boff[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) boff[hh] = (i >> 4) * 256 + ((i & 15) * 4 + (((2 * hf + hh) ^ ((i & 15) >> 2)) & 3)) * 4;

  f32x16 acc[KC][NBI][OBC];
  float bacc[KC][OBC];
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) {
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int ob = 0; ob < OBC; ++ob)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kk][ib][ob][e] = 0.f;
#pragma unroll
    for (int ob = 0; ob < OBC; ++ob) bacc[kk][ob] = 0.f;
  }
  const int zt_mod = (int)A.zt_mod, nt_all = (int)A.ntiles;
  // DMA source of this lane inside a 1-KiB chunk j: row r = lane>>3, piece c = (lane&7) ^ r ^ (j&1)
  const int dr = lane >> 3, dx = lane & 7;
  const int src0 = dr * 32 + ((dx ^ dr) & 7) * 4, src1 = dr * 32 + ((dx ^ dr ^ 1) & 7) * 4;
  // latent rows k0, k0+1 (clamped to a valid row; k >= r is replaced by ones at use)
  const int kz = (k0 + hf < A.r) ? k0 + hf : (A.r > 0 ? A.r - 1 : 0);
  auto dma_tile = [&](long t, int set) {
    float* dst = wbuf + set * BUF;
    const float* in = A.IN + t * TFI;
    const float* da = A.DA + t * TFO + (long)ob0 * (DAB ? 512 : 1024);
#pragma unroll
    for (int j = 0; j < NBI * 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + j * 256 + ((j & 1) ? src1 : src0)),
                                       (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
    if constexpr (DAB) {
#pragma unroll
      for (int j = 0; j < OBC * 2; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + j * 256 + bsrc),
                                         (__attribute__((address_space(3))) void*)(dst + TFI + j * 256), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < OBC * 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + j * 256 + ((j & 1) ? src1 : src0)),
                                         (__attribute__((address_space(3))) void*)(dst + TFI + j * 256), 16, 0, 0);
    }
    const int ti = __builtin_amdgcn_readfirstlane((int)t);
    const int tz = zt_mod >= nt_all ? ti : ti % zt_mod;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + ((long)tz * A.r + kz) * 32 + i),
                                     (__attribute__((address_space(3))) void*)(dst + TFI + TFB), 4, 0, 0);
  };
  // reader offset of lane (i, hf): chunk i>>3 of the block, row i&7, piece 4 hf + q
  const int jr = i >> 3, rr = i & 7;
  int roff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) roff[q] = jr * 256 + (rr * 8 + (((4 * hf + q) ^ rr ^ (jr & 1)) & 7)) * 4;

  const long last = A.ntiles - 1;
  long t = (long)blockIdx.x * WV + wid;
  int set = 0;
  if (t < A.ntiles) dma_tile(t, 0);
  for (; t < A.ntiles; t += nwaves, set ^= (NBUF - 1)) {
    const float* buf = wbuf + set * BUF;
    f32x4 af[NBI][4], bf[DAB ? 1 : OBC][4], zq[KC][4];
    bf16x8 bq[DAB ? OBC : 1][2];       // DAB: the tile's dL/da operands as they are
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // operands of the tile's quads [q0, q1) from LDS into registers
    auto read_quads = [&](int q0, int q1) {
#pragma unroll
      for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
        for (int q = q0; q < q1; ++q) af[ib][q] = *reinterpret_cast<const f32x4*>(buf + ib * 1024 + roff[q]);
      if constexpr (DAB) {
#pragma unroll
        for (int ob = 0; ob < OBC; ++ob)
#pragma unroll
          for (int hh = q0 / 2; hh < q1 / 2; ++hh) bq[ob][hh] = *reinterpret_cast<const bf16x8*>(buf + TFI + ob * 512 + boff[hh]);
      } else {
#pragma unroll
        for (int ob = 0; ob < OBC; ++ob)
#pragma unroll
          for (int q = q0; q < q1; ++q) bf[ob][q] = *reinterpret_cast<const f32x4*>(buf + TFI + ob * 1024 + roff[q]);
      }
#pragma unroll
      for (int kk = 0; kk < KC; ++kk)
#pragma unroll
        for (int q = q0; q < q1; ++q) zq[kk][q] = *reinterpret_cast<const f32x4*>(buf + TFI + TFB + kk * 32 + 16 * hf + 4 * q);
    };
    const long t1 = t + nwaves;
    if (NBUF == 2) {
      read_quads(0, 4);
      dma_tile(t1 < last ? t1 : last, set ^ 1);   // unconditional (re-reads the last tile at the end)
    }
    const bool wbias = t < A.bias_ntiles;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (NBUF == 1) {
        // one buffer (256 accumulator registers): only half a tile of operands is live at a time (the whole tile in
        // registers spilled 65 of them, and scratch reloads share the in-order vmcnt queue with the DMA); the buffer is
        // refilled once its second half has been read, so the DMA overlaps the second half's split + MFMA phase
        read_quads(2 * hh, 2 * hh + 2);
        if (hh == 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_tile(t1 < last ? t1 : last, set);
        }
      }
      bf16x8 bh[OBC], bl[DAB ? 1 : OBC];
#pragma unroll
      for (int ob = 0; ob < OBC; ++ob) {
        if constexpr (DAB) bh[ob] = bq[ob][hh];
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = bf[ob][2 * hh + (e >> 2)][e & 3];
            const __bf16 x0 = (__bf16)x;
            bh[ob][e] = x0; bl[ob][e] = (__bf16)(x - (float)x0);
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        if (k0 + kk > A.r) break;
#pragma unroll
        for (int ib = 0; ib < NBI; ++ib) {
          bf16x8 ah, al;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = af[ib][2 * hh + (e >> 2)][e & 3] * (k0 + kk < A.r ? zq[kk][2 * hh + (e >> 2)][e & 3] : 1.0f);
            const __bf16 x0 = (__bf16)x;
            ah[e] = x0;
            if (!DAB) al[e] = (__bf16)(x - (float)x0);
          }
#pragma unroll
          for (int ob = 0; ob < OBC; ++ob) {
            if constexpr (!DAB) {
              acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ob], acc[kk][ib][ob], 0, 0, 0);
              acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ob], acc[kk][ib][ob], 0, 0, 0);
            }
            acc[kk][ib][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ob], acc[kk][ib][ob], 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 2 * hh; q < 2 * hh + 2; ++q)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ob = 0; ob < OBC; ++ob)
              bacc[kk][ob] = fmaf(wbias ? (k0 + kk < A.r ? zq[kk][q][c] : 1.0f) : 0.f,
                                  DAB ? (float)bh[ob][4 * (q - 2 * hh) + c] : bf[DAB ? 0 : ob][q][c], bacc[kk][ob]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // the tile buffers become the reduction scratch

  float* red = gsm;
  float* red16 = gsm + (WV - 1) * 64;
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) {
    const int k = k0 + kk;
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int ob = 0; ob < OBC; ++ob) {
        f32x16 v = block_sum16<WV>(acc[kk][ib][ob], red16, wid, lane);
        if (wid == 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int in = 32 * ib + fmap(e, hf), out = 32 * (ob0 + ob) + i;
            if (k <= A.r && in < A.W.nin && out < A.W.nout) prow[matref_index(A.W, k, in, out)] = A.scale * v[e];
          }
        }
      }
#pragma unroll
    for (int ob = 0; ob < OBC; ++ob) {
      float v = bacc[kk][ob];
      v += __shfl_xor(v, 32);
      v = block_sum<WV>(v, red, wid, lane);
      const int out = 32 * (ob0 + ob) + i;
      if (A.has_bias && wid == 0 && hf == 0 && k <= A.r && out < A.Bv.nout) prow[matref_index(A.Bv, k, 0, out)] = v;
    }
  }
}

static GwArgs gw_fix(const GwArgs& in) {
  GwArgs a = in;
  if (a.zt_mod <= 0) a.zt_mod = a.ntiles > 0 ? a.ntiles : 1;
  if (a.bias_ntiles <= 0) a.bias_ntiles = a.ntiles;
  if (!a.Z) a.Z = a.IN ? a.IN : a.DA;   // r == 0: the (ignored) latent loads of k_gw_mfma need a readable address
  return a;
}

#ifndef NIF_GW_WIDE_DBUF
#define NIF_GW_WIDE_DBUF false
#endif
#ifndef NIF_GW_WIDE_KC2
#define NIF_GW_WIDE_KC2 1
#endif
#ifndef NIF_GW_WAVES
#define NIF_GW_WAVES 4   // 8 = two waves per SIMD, single-buffered: spills at 256 registers, slower
#endif
bool gw8_supported(const GwArgs& a, int NBI, int NBO);
static bool gw_use_lds() { static const bool v = [] { const char* e = getenv("NIF_GW_LDS"); return !(e && e[0] == '0'); }(); return v; }
// may the producers of this context write their hidden-layer dL/da stash rows in bf16 (mixed_bfloat16)?  k_gw_lds / k_gw8 read them
bool gw_da_bf16_ok(int NBI, int NBO, int r) {
  static const bool on = [] { const char* e = getenv("NIF_DA_BF16"); return !(e && e[0] == '0'); }();
  static const bool use_gw8 = [] { const char* e = getenv("NIF_GW8"); return !(e && e[0] == '0'); }();
  if (!on || NBI != NBO) return false;
  if (NBI == 4) return use_gw8 && (r == 0 || r == 1);      // k_gw8<R, DAB>
  return gw_use_lds() && NBI <= 2;                         // k_gw_lds<.., DAB>
}
// ... and their hidden matrices' INPUT rows as 16-bit phases (r5)?  One reader: k_gw8<R, true, true>, next to bf16 dL/da rows
bool gw_in_ph16_ok(int NBI, int NBO, int r) {
  static const bool on = [] { const char* e = getenv("NIF_H_PH16"); return !(e && e[0] == '0'); }();
  return on && NBI == 4 && gw_da_bf16_ok(NBI, NBO, r);
}
void launch_gw8(const GwArgs& a, int rows, hipStream_t st);
int launch_gw_mfma(const GwArgs& a_, int NBI, int NBO, int rows, hipStream_t st) {
  const GwArgs a = gw_fix(a_);
  if (a.in_ph16 && !(a.da_bf16 && gw_in_ph16_ok(NBI, NBO, a.r))) return -1;
  // bf16 dL/da rows have readers of two forms only (k_gw8<R, true>, k_gw_lds<1|2, .., true>): anything else would read them as fp32
  if (a.da_bf16 && !gw_da_bf16_ok(NBI, NBO, a.r)) return -1;
  constexpr int WV = NIF_GW_WAVES;
  dim3 block(64 * WV);
  const bool use_lds = gw_use_lds();
  // 128-wide layers: the 8-wave shared-tile kernel reads every stash tile once (k_gw8.hip); NIF_GW8=0: the r1 kernels (A/B)
  static const bool use_gw8 = [] { const char* e = getenv("NIF_GW8"); return !(e && e[0] == '0'); }();
  if (use_gw8 && gw8_supported(a, NBI, NBO)) { launch_gw8(a, rows, st); return 0; }
  if (use_lds && NBI == NBO && (NBI <= 2 || NBI == 4)) {
#define NIF_GWL(NBI_, OBC_, NBUF_, DAB_)                                                                                   \
  do {                                                                                                                     \
    const size_t shm = sizeof(float) * (size_t)(4 * NBUF_ * ((NBI_ + OBC_) * 1024 + 64));                                  \
    dim3 grid(rows, (a.r + 1 + 1) / 2, NBI_ / OBC_);                                                                       \
    (void)hipFuncSetAttribute((const void*)k_gw_lds<NBI_, OBC_, NBUF_, DAB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((k_gw_lds<NBI_, OBC_, NBUF_, DAB_>), grid, dim3(256), shm, st, a, NBO);                             \
  } while (0)
    if (a.da_bf16) { if (NBI == 1) NIF_GWL(1, 1, 2, true); else NIF_GWL(2, 2, 2, true); }      // (gw_da_bf16_ok: NBI <= 2)
    else if (NBI == 1) NIF_GWL(1, 1, 2, false);
    else if (NBI == 2) NIF_GWL(2, 2, 2, false);
    else NIF_GWL(4, 2, 1, false);
#undef NIF_GWL
    return 0;
  }
  if (NBI == 1 && NBO == 1) {
    dim3 grid(rows, (a.r + 1 + 1) / 2, 1);
    hipLaunchKernelGGL((k_gw_mfma<1, 1, 2, WV, true>), grid, block, 0, st, a, NBO);
  } else if (NBI == 2 && NBO == 2) {
    dim3 grid(rows, (a.r + 1 + 1) / 2, 1);
    hipLaunchKernelGGL((k_gw_mfma<2, 2, 2, WV, true>), grid, block, 0, st, a, NBO);
  } else if (NBI == 4 && NBO == 4) {
#if NIF_GW_WIDE_KC2
    // 128-wide: both planes of a pair in one workgroup (256 accumulator registers, single-buffered): the layer-input
    // stash is read 2x and dL/da 1x instead of 4x and 2x
    dim3 grid(rows, (a.r + 1 + 1) / 2, 2);
    hipLaunchKernelGGL((k_gw_mfma<4, 2, 2, WV, NIF_GW_WIDE_DBUF>), grid, block, 0, st, a, NBO);
#else
    dim3 grid(rows, a.r + 1, 2);
    hipLaunchKernelGGL((k_gw_mfma<4, 2, 1, WV, true>), grid, block, 0, st, a, NBO);
#endif
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// first layers (K = pi or si input columns, tiny): VALU.  grid = (rows, r+1)
//   gW[k][d][f] = scale * sum_p zt_k x_d da[p][f],  gb[k][f] = sum_p zt_k da[p][f]
// ------------------------------------------------------------------------------------------
template <int NBO>
__global__ __launch_bounds__(256) void k_gw_first(GwArgs A) {
  extern __shared__ float sm[];  // red[3*64] then per-wave acc[(nd+1)*NBO][64]
  float* red = sm;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const int k = blockIdx.y;
  const long nwaves = (long)gridDim.x * 4;
  const long FO = (long)NBO * 32 * 32;
  const int nacc = (A.nd + 1) * NBO;
  float* lacc = sm + 192 + (long)wid * nacc * 64;
  for (int s = 0; s < nacc; ++s) lacc[s * 64 + lane] = 0.f;

  for (long t = (long)blockIdx.x * 4 + wid; t < A.ntiles; t += nwaves) {
    f32x4 bf[NBO][4], zq[4];
    const int tz = A.zt_mod >= A.ntiles ? (int)t : (int)t % (int)A.zt_mod;
#pragma unroll
    for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[ob][q] = ld4(A.DA + t * FO + (long)(32 * ob + i) * 32 + 16 * hf + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (k < A.r) zq[q] = ld4(A.Z + ((long)tz * A.r + k) * 32 + 16 * hf + 4 * q);
      else { zq[q][0] = 1.f; zq[q][1] = 1.f; zq[q][2] = 1.f; zq[q][3] = 1.f; }
    }
    // Sobolev pseudo-tiles (t >= bias_ntiles): the "input" of tangent stream d is the one-hot e_seed[d]
    const int pseudo = t < A.bias_ntiles ? -1 : A.seed[(int)t / (int)A.zt_mod - 1];
    for (int dd = 0; dd <= A.nd; ++dd) {  // dd == nd : the bias (x = 1)
      if (pseudo >= 0 && dd != pseudo) continue;
      float s[NBO];
#pragma unroll
      for (int ob = 0; ob < NBO; ++ob) s[ob] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          long pt = t * 32 + 16 * hf + 4 * q + c;
          if (pt >= A.B) pt = A.B - 1;
          const float xv = (dd < A.nd && pseudo < 0) ? A.xin[pt * A.ncol + A.col0 + dd] : 1.0f;
          const float w = xv * zq[q][c];
#pragma unroll
          for (int ob = 0; ob < NBO; ++ob) s[ob] = fmaf(w, bf[ob][q][c], s[ob]);
        }
#pragma unroll
      for (int ob = 0; ob < NBO; ++ob) lacc[(dd * NBO + ob) * 64 + lane] += s[ob];
    }
  }
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
  for (int dd = 0; dd <= A.nd; ++dd)
    for (int ob = 0; ob < NBO; ++ob) {
      float v = lacc[(dd * NBO + ob) * 64 + lane];
      v += __shfl_xor(v, 32);
      v = block_sum4(v, red, wid, lane);
      const int f = 32 * ob + i;
      if (wid == 0 && hf == 0) {
        if (dd < A.nd) { if (f < A.W.nout) prow[matref_index(A.W, k, dd, f)] = A.scale * v; }
        else if (A.has_bias && f < A.Bv.nout) prow[matref_index(A.Bv, k, 0, f)] = v;
      }
    }
}

// MFMA form of k_gw_first for (r+1)(nd+1) <= 32: the rows of ONE 32-row A operand are the planes x inputs
//   A[(k, d)][p] = zt_k[p] * x_d[p]   (d = nd: the bias row, x = 1),   B = dL/da tile,   K = the tile's 32 points
// so a tile costs 16 MFMAs per 32 output features instead of (r+1)(nd+1) VALU passes with LDS accumulators.
template <int NBO, int WV>
__global__ __launch_bounds__(64 * WV) void k_gw_first_mfma(GwArgs A) {
  __shared__ float red16[(WV - 1) * 16 * 64];
  extern __shared__ __attribute__((aligned(16))) float xs_all[];   // per wave: the tile's input columns [nd][32]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const long nwaves = (long)gridDim.x * WV;
  const long FO = (long)NBO * 32 * 32;
  const int nd1 = A.nd + 1;
  const int k = i / nd1, d = i - k * nd1;
  const bool row_ok = i < (A.r + 1) * nd1;
  f32x16 acc[NBO];
#pragma unroll
  for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ob][e] = 0.f;

  for (long t = (long)blockIdx.x * WV + wid; t < A.ntiles; t += nwaves) {
    f32x4 bf[NBO][4], a[4];
#pragma unroll
    for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[ob][q] = ld4(A.DA + t * FO + (long)(32 * ob + i) * 32 + 16 * hf + 4 * q);
    const int tz = A.zt_mod >= A.ntiles ? (int)t : (int)t % (int)A.zt_mod;
    const int pseudo = t < A.bias_ntiles ? -1 : A.seed[(int)t / (int)A.zt_mod - 1];
    // the tile's input columns, transposed through the wave's private LDS rows (one coalesced pass instead of 16
    // scalar loads per lane)
    float* xs = xs_all + (long)wid * A.nd * 32;
    if (pseudo < 0)
      for (int e2 = lane; e2 < 32 * A.nd; e2 += 64) {
        const int pp = e2 / A.nd, dd = e2 - pp * A.nd;
        long pt = (long)tz * 32 + pp;
        if (pt >= A.B) pt = A.B - 1;
        xs[dd * 32 + pp] = A.xin[pt * A.ncol + A.col0 + dd];
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 z = {1.f, 1.f, 1.f, 1.f};
      if (row_ok && k < A.r) z = ld4(A.Z + ((long)tz * A.r + k) * 32 + 16 * hf + 4 * q);
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      if (row_ok) {
        if (pseudo >= 0) { const float o = d == pseudo ? 1.0f : 0.0f; xv[0] = o; xv[1] = o; xv[2] = o; xv[3] = o; }   // one-hot, no bias row
        else if (d == A.nd) { xv[0] = 1.f; xv[1] = 1.f; xv[2] = 1.f; xv[3] = 1.f; }
        else xv = *reinterpret_cast<const f32x4*>(xs + d * 32 + 16 * hf + 4 * q);
      }
      a[q] = xv * z;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ob = 0; ob < NBO; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], bf[ob][q][c], acc[ob], 0, 0, 0);
  }
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
#pragma unroll
  for (int ob = 0; ob < NBO; ++ob) {
    const f32x16 vs = block_sum16<WV>(acc[ob], red16, wid, lane);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = vs[e];
      const int m = fmap(e, hf), f = 32 * ob + i;
      const int kk = m / nd1, dd = m - kk * nd1;
      if (wid == 0 && m < (A.r + 1) * nd1) {
        if (dd < A.nd) { if (f < A.W.nout) prow[matref_index(A.W, kk, dd, f)] = A.scale * v; }
        else if (A.has_bias && f < A.Bv.nout) prow[matref_index(A.Bv, kk, 0, f)] = v;
      }
    }
  }
}

// MFMA form of k_gw_out for (r+1) nc <= 32: the columns of ONE 32-column B operand are the planes x outputs
//   A = layer-input tile,   B[p][(k, c)] = zt_k[p] * dL/dout_c[p];   bias gradient = column sums of B (real tiles only)
template <int NBI, int WV>
__global__ __launch_bounds__(64 * WV) void k_gw_out_mfma(GwArgs A) {
  __shared__ float red[(WV - 1) * 64];
  __shared__ float red16[(WV - 1) * 16 * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const long nwaves = (long)gridDim.x * WV;
  const long FI = (long)NBI * 32 * 32;
  const int k = i / A.nc, c = i - k * A.nc;
  const bool col_ok = i < (A.r + 1) * A.nc;
  f32x16 acc[NBI];
  float bsum = 0.f;
#pragma unroll
  for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ib][e] = 0.f;
  const bool need_w = A.W.nin > 0;

  for (long t = (long)blockIdx.x * WV + wid; t < A.ntiles; t += nwaves) {
    f32x4 af[NBI][4], b[4];
    if (need_w) {
#pragma unroll
      for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
        for (int q = 0; q < 4; ++q) af[ib][q] = ld4(A.IN + t * FI + (long)(32 * ib + i) * 32 + 16 * hf + 4 * q);
    }
    const int tz = A.zt_mod >= A.ntiles ? (int)t : (int)t % (int)A.zt_mod;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (col_ok) {
        v = ld4(A.SM + (t * A.nc + c) * 32 + 16 * hf + 4 * q);
        if (k < A.r) v *= ld4(A.Z + ((long)tz * A.r + k) * 32 + 16 * hf + 4 * q);
      }
      b[q] = v;
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    if (t < A.bias_ntiles) bsum += s;
    if (need_w) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ib = 0; ib < NBI; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ib][q][e], b[q][e], acc[ib], 0, 0, 0);
    }
  }
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
  if (need_w) {
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib) {
      const f32x16 vs = block_sum16<WV>(acc[ib], red16, wid, lane);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = 32 * ib + fmap(e, hf);
        if (wid == 0 && col_ok && f < A.W.nin) prow[matref_index(A.W, k, f, c)] = A.scale * vs[e];
      }
    }
  }
  {
    float v = bsum;
    v += __shfl_xor(v, 32);
    v = block_sum<WV>(v, red, wid, lane);
    if (A.has_bias && wid == 0 && hf == 0 && col_ok) prow[matref_index(A.Bv, k, 0, c)] = v;
  }
}

#ifndef NIF_GW_EDGE_MFMA
#define NIF_GW_EDGE_MFMA 1
#endif
#ifndef NIF_GW_EDGE_WAVES
#define NIF_GW_EDGE_WAVES 16
#endif
// LDS-DMA form of k_gw_first_mfma (see k_gw_lds): the dL/da tile arrives as contiguous KiB chunks, the tile's input
// rows [32 points][ncol] and latent rows by 4-byte DMA; the K = 32 points product runs as three bf16 products
// (hi*lo + lo*hi + hi*hi, like the hidden layers) on v_mfma_f32_32x32x16_bf16.  grid = (rows)
template <int NBO, int WV>
__global__ __launch_bounds__(64 * WV) void k_gw_first_lds(GwArgs A) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  constexpr int TF = NBO * 1024;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const int r = A.r, ncol = A.ncol;
  const int nzf = r * 32, nxf = ncol * 32;
  const int ZO = TF, XO = TF + ((nzf + 63) & ~63);
  const int BUF = XO + ((nxf + 63) & ~63);
  const long nwaves = (long)gridDim.x * WV;
  float* wbuf = gsm + (long)wid * 2 * BUF;
  const int nd1 = A.nd + 1;
  const int k = i / nd1, d = i - k * nd1;
  const bool row_ok = i < (r + 1) * nd1;
  f32x16 acc[NBO];
#pragma unroll
  for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ob][e] = 0.f;
  const int zt_mod = (int)A.zt_mod, nt_all = (int)A.ntiles;
  const long xmax = A.B * ncol - 1;
  const int dr = lane >> 3, dx = lane & 7;
  const int src0 = dr * 32 + ((dx ^ dr) & 7) * 4, src1 = dr * 32 + ((dx ^ dr ^ 1) & 7) * 4;
  auto dma_tile = [&](long t, int set) {
    float* dst = wbuf + set * BUF;
    const float* da = A.DA + t * TF;
#pragma unroll
    for (int j = 0; j < NBO * 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + j * 256 + ((j & 1) ? src1 : src0)),
                                       (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
    const int ti = __builtin_amdgcn_readfirstlane((int)t);
    const int tz = zt_mod >= nt_all ? ti : ti % zt_mod;
    for (int m = 0; m < nzf; m += 64) {
      const int e = m + lane < nzf ? m + lane : nzf - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + (long)tz * nzf + e),
                                       (__attribute__((address_space(3))) void*)(dst + ZO + m), 4, 0, 0);
    }
    for (int m = 0; m < nxf; m += 64) {
      long e = (long)tz * nxf + m + lane;     // rows beyond the batch: any valid element (their dL/da rows are zero)
      e = e < xmax ? e : xmax;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.xin + e),
                                       (__attribute__((address_space(3))) void*)(dst + XO + m), 4, 0, 0);
    }
  };
  const int jr = i >> 3, rr = i & 7;
  int roff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) roff[q] = jr * 256 + (rr * 8 + (((4 * hf + q) ^ rr ^ (jr & 1)) & 7)) * 4;

  const long last = A.ntiles - 1;
  long t = (long)blockIdx.x * WV + wid;
  int set = 0;
  if (t < A.ntiles) dma_tile(t, 0);
  for (; t < A.ntiles; t += nwaves, set ^= 1) {
    const float* buf = wbuf + set * BUF;
    f32x4 bf[NBO][4], a[4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {   // refill the OTHER buffer first (its reads ended with the previous tile): the whole tile's work overlaps the DMA
      const long t1 = t + nwaves;
      dma_tile(t1 < last ? t1 : last, set ^ 1);
    }
#pragma unroll
    for (int ob = 0; ob < NBO; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[ob][q] = *reinterpret_cast<const f32x4*>(buf + ob * 1024 + roff[q]);
    // wave-uniform, and the seed picked by static index (a dynamic index into the kernel arguments is a vector load
    // whose s_waitcnt vmcnt(0) would also drain the DMA just issued)
    const int tu = __builtin_amdgcn_readfirstlane((int)t);
    const int sidx = tu < (int)A.bias_ntiles ? -1 : tu / zt_mod - 1;
    const int pseudo = sidx < 0 ? -1 : (sidx == 0 ? A.seed[0] : (sidx == 1 ? A.seed[1] : A.seed[2]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 z = {1.f, 1.f, 1.f, 1.f};
      if (row_ok && k < r) z = *reinterpret_cast<const f32x4*>(buf + ZO + k * 32 + 16 * hf + 4 * q);
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      if (row_ok) {
        if (pseudo >= 0) { const float o = d == pseudo ? 1.0f : 0.0f; xv[0] = o; xv[1] = o; xv[2] = o; xv[3] = o; }   // one-hot, no bias row
        else if (d == A.nd) { xv[0] = 1.f; xv[1] = 1.f; xv[2] = 1.f; xv[3] = 1.f; }
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) xv[c] = buf[XO + (16 * hf + 4 * q + c) * ncol + A.col0 + d];
        }
      }
      a[q] = xv * z;
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      bf16x8 ah, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = a[2 * hh + (e >> 2)][e & 3];
        const __bf16 x0 = (__bf16)x;
        ah[e] = x0; al[e] = (__bf16)(x - (float)x0);
      }
#pragma unroll
      for (int ob = 0; ob < NBO; ++ob) {
        bf16x8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = bf[ob][2 * hh + (e >> 2)][e & 3];
          const __bf16 x0 = (__bf16)x;
          bh[e] = x0; bl[e] = (__bf16)(x - (float)x0);
        }
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ob], 0, 0, 0);
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ob], 0, 0, 0);
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ob], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red16 = gsm;
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
#pragma unroll
  for (int ob = 0; ob < NBO; ++ob) {
    const f32x16 vs = block_sum16<WV>(acc[ob], red16, wid, lane);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = vs[e];
      const int m = fmap(e, hf), f = 32 * ob + i;
      const int kk = m / nd1, dd = m - kk * nd1;
      if (wid == 0 && m < (r + 1) * nd1) {
        if (dd < A.nd) { if (f < A.W.nout) prow[matref_index(A.W, kk, dd, f)] = A.scale * v; }
        else if (A.has_bias && f < A.Bv.nout) prow[matref_index(A.Bv, kk, 0, f)] = v;
      }
    }
  }
}

void launch_gw_first(const GwArgs& a_, int NBO, int rows, hipStream_t st) {
  const GwArgs a = gw_fix(a_);
  static const bool use_lds = [] { const char* e = getenv("NIF_GW_LDS"); return !(e && e[0] == '0'); }();
  if (use_lds && NIF_GW_EDGE_MFMA && (a.r + 1) * (a.nd + 1) <= 32 && (NBO <= 2 || NBO == 4) && a.ncol <= 16) {
    const int buf = NBO * 1024 + ((a.r * 32 + 63) & ~63) + ((a.ncol * 32 + 63) & ~63);
    constexpr int WVL = 4;   // one wave per SIMD (two: 0.089 instead of 0.080 ms)
    size_t shl = sizeof(float) * (size_t)(WVL * 2 * buf);
    const size_t shr = sizeof(float) * (size_t)((WVL - 1) * 16 * 64);   // the block reduction reuses the tile buffers
    if (shl < shr) shl = shr;
    if (NBO == 1) {
      (void)hipFuncSetAttribute((const void*)k_gw_first_lds<1, WVL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl);
      hipLaunchKernelGGL((k_gw_first_lds<1, WVL>), dim3(rows), dim3(64 * WVL), shl, st, a);
    } else if (NBO == 2) {
      (void)hipFuncSetAttribute((const void*)k_gw_first_lds<2, WVL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl);
      hipLaunchKernelGGL((k_gw_first_lds<2, WVL>), dim3(rows), dim3(64 * WVL), shl, st, a);
    } else {
      (void)hipFuncSetAttribute((const void*)k_gw_first_lds<4, WVL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl);
      hipLaunchKernelGGL((k_gw_first_lds<4, WVL>), dim3(rows), dim3(64 * WVL), shl, st, a);
    }
    return;
  }
  if (NIF_GW_EDGE_MFMA && (a.r + 1) * (a.nd + 1) <= 32) {
    constexpr int WV = NIF_GW_EDGE_WAVES;   // one workgroup per partial row: many waves hide the single-buffered loads
    dim3 grid(rows);
    const size_t shx = (size_t)WV * (a.nd > 0 ? a.nd : 1) * 32 * sizeof(float);
    if (NBO == 1) hipLaunchKernelGGL((k_gw_first_mfma<1, WV>), grid, dim3(64 * WV), shx, st, a);
    else if (NBO == 2) hipLaunchKernelGGL((k_gw_first_mfma<2, WV>), grid, dim3(64 * WV), shx, st, a);
    else hipLaunchKernelGGL((k_gw_first_mfma<4, 4>), grid, dim3(256), shx, st, a);   // 128+ registers: 4 waves
    return;
  }
  dim3 grid(rows, a.r + 1), block(256);
  const size_t shm = (size_t)(192 + 4 * (a.nd + 1) * NBO * 64) * sizeof(float);
  if (NBO == 1) hipLaunchKernelGGL((k_gw_first<1>), grid, block, shm, st, a);
  else if (NBO == 2) hipLaunchKernelGGL((k_gw_first<2>), grid, block, shm, st, a);
  else hipLaunchKernelGGL((k_gw_first<4>), grid, block, shm, st, a);
}

// ------------------------------------------------------------------------------------------
// narrow output layers (n -> so, nst -> r): VALU.  grid = (rows, r+1)
//   gW[k][f][c] = scale * sum_p zt_k h[p][f] dout[p][c],  gb[k][c] = sum_p zt_k dout[p][c]
// ------------------------------------------------------------------------------------------
template <int NBI>
__global__ __launch_bounds__(256) void k_gw_out(GwArgs A) {
  extern __shared__ float sm[];  // red[3*64] then per-wave acc[nc*(NBI+1)][64]
  float* red = sm;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const int k = blockIdx.y;
  const long nwaves = (long)gridDim.x * 4;
  const long FI = (long)NBI * 32 * 32;
  const int nacc = A.nc * (NBI + 1);
  float* lacc = sm + 192 + (long)wid * nacc * 64;
  for (int s = 0; s < nacc; ++s) lacc[s * 64 + lane] = 0.f;

  for (long t = (long)blockIdx.x * 4 + wid; t < A.ntiles; t += nwaves) {
    f32x4 af[NBI][4], zq[4];
    const int tz = A.zt_mod >= A.ntiles ? (int)t : (int)t % (int)A.zt_mod;
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[ib][q] = ld4(A.IN + t * FI + (long)(32 * ib + i) * 32 + 16 * hf + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (k < A.r) zq[q] = ld4(A.Z + ((long)tz * A.r + k) * 32 + 16 * hf + 4 * q);
      else { zq[q][0] = 1.f; zq[q][1] = 1.f; zq[q][2] = 1.f; zq[q][3] = 1.f; }
    }
    const bool wbias = t < A.bias_ntiles;
    for (int c = 0; c < A.nc; ++c) {
      float s[NBI + 1];
#pragma unroll
      for (int ib = 0; ib <= NBI; ++ib) s[ib] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 dq = ld4(A.SM + (t * A.nc + c) * 32 + 16 * hf + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float w = dq[e] * zq[q][e];
#pragma unroll
          for (int ib = 0; ib < NBI; ++ib) s[ib] = fmaf(w, af[ib][q][e], s[ib]);
          s[NBI] += wbias ? w : 0.f;
        }
      }
#pragma unroll
      for (int ib = 0; ib <= NBI; ++ib) lacc[(c * (NBI + 1) + ib) * 64 + lane] += s[ib];
    }
  }
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
  for (int c = 0; c < A.nc; ++c)
    for (int ib = 0; ib <= NBI; ++ib) {
      float v = lacc[(c * (NBI + 1) + ib) * 64 + lane];
      v += __shfl_xor(v, 32);
      v = block_sum4(v, red, wid, lane);
      if (wid == 0 && hf == 0) {
        const int f = 32 * ib + i;
        if (ib < NBI) { if (f < A.W.nin) prow[matref_index(A.W, k, f, c)] = A.scale * v; }
        else if (A.has_bias && i == 0) prow[matref_index(A.Bv, k, 0, c)] = v;
      }
    }
}

// LDS-DMA form of k_gw_out for (r+1) nc <= 8 (see k_gw_lds): the layer-input tile arrives as contiguous KiB chunks,
// all r+1 planes are reduced by the same workgroup (the stash is read once, not r+1 times) and the (plane, column)
// accumulators stay in registers.  grid = (rows)
template <int NBI>
__global__ __launch_bounds__(256) void k_gw_out_lds(GwArgs A) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  constexpr int WV = 4, TF = NBI * 1024, MAXKC = 8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, hf = lane >> 5;
  const int r = A.r, nc = A.nc, nkc = (r + 1) * nc;
  const int nzf = r * 32, nsf = nc * 32;                 // floats of the tile's latent / dL/dout rows
  const int ZO = TF, SO = TF + ((nzf + 63) & ~63);       // buffer: IN | Z rows | SM rows
  const int BUF = SO + ((nsf + 63) & ~63);
  const long nwaves = (long)gridDim.x * WV;
  float* wbuf = gsm + (long)wid * 2 * BUF;
  float acc[MAXKC][NBI + 1];
#pragma unroll
  for (int kc = 0; kc < MAXKC; ++kc)
#pragma unroll
    for (int ib = 0; ib <= NBI; ++ib) acc[kc][ib] = 0.f;
  const int zt_mod = (int)A.zt_mod, nt_all = (int)A.ntiles;
  const int dr = lane >> 3, dx = lane & 7;
  const int src0 = dr * 32 + ((dx ^ dr) & 7) * 4, src1 = dr * 32 + ((dx ^ dr ^ 1) & 7) * 4;
  auto dma_tile = [&](long t, int set) {
    float* dst = wbuf + set * BUF;
    const float* in = A.IN + t * TF;
#pragma unroll
    for (int j = 0; j < NBI * 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + j * 256 + ((j & 1) ? src1 : src0)),
                                       (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
    const int ti = __builtin_amdgcn_readfirstlane((int)t);
    const int tz = zt_mod >= nt_all ? ti : ti % zt_mod;
    for (int m = 0; m < nzf; m += 64) {
      const int e = m + lane < nzf ? m + lane : nzf - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + (long)tz * nzf + e),
                                       (__attribute__((address_space(3))) void*)(dst + ZO + m), 4, 0, 0);
    }
    for (int m = 0; m < nsf; m += 64) {
      const int e = m + lane < nsf ? m + lane : nsf - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.SM + t * nsf + e),
                                       (__attribute__((address_space(3))) void*)(dst + SO + m), 4, 0, 0);
    }
  };
  const int jr = i >> 3, rr = i & 7;
  int roff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) roff[q] = jr * 256 + (rr * 8 + (((4 * hf + q) ^ rr ^ (jr & 1)) & 7)) * 4;

  const long last = A.ntiles - 1;
  long t = (long)blockIdx.x * WV + wid;
  int set = 0;
  if (t < A.ntiles) dma_tile(t, 0);
  for (; t < A.ntiles; t += nwaves, set ^= 1) {
    const float* buf = wbuf + set * BUF;
    f32x4 af[NBI][4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[ib][q] = *reinterpret_cast<const f32x4*>(buf + ib * 1024 + roff[q]);
    const long t1 = t + nwaves;
    const bool wbias = t < A.bias_ntiles;
    // the small rows are read per (plane, column) below, so the refill of the other buffer can start right away
    dma_tile(t1 < last ? t1 : last, set ^ 1);
#pragma unroll
    for (int kc = 0; kc < MAXKC; ++kc) {
      if (kc < nkc) {
        const int k = kc / nc, c = kc - k * nc;
        float sb = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 w = *reinterpret_cast<const f32x4*>(buf + SO + c * 32 + 16 * hf + 4 * q);
          if (k < r) w *= *reinterpret_cast<const f32x4*>(buf + ZO + k * 32 + 16 * hf + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int ib = 0; ib < NBI; ++ib) acc[kc][ib] = fmaf(w[e], af[ib][q][e], acc[kc][ib]);
            sb += w[e];
          }
        }
        acc[kc][NBI] += wbias ? sb : 0.f;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red = gsm;
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
#pragma unroll
  for (int kc = 0; kc < MAXKC; ++kc) {
    if (kc < nkc) {
      const int k = kc / nc, c = kc - k * nc;
#pragma unroll
      for (int ib = 0; ib <= NBI; ++ib) {
        float v = acc[kc][ib];
        v += __shfl_xor(v, 32);
        v = block_sum4(v, red, wid, lane);
        if (wid == 0 && hf == 0) {
          const int f = 32 * ib + i;
          if (ib < NBI) { if (f < A.W.nin) prow[matref_index(A.W, k, f, c)] = A.scale * v; }
          else if (A.has_bias && i == 0) prow[matref_index(A.Bv, k, 0, c)] = v;
        }
      }
    }
  }
}

// bias only (W.nin = 0, r = 0: last_layer_bias of the last-layer class): column sums of the small per-point vectors
// SM [tiles][nc][32] -- 4 nc bytes per point instead of a whole layer-input tile (r3: the call went through k_gw_out_lds, which
// streamed the 512 B / point of the 128-wide stash tile only to ignore it: 0.195 ms of cfg-4's step).  grid = (rows)
__global__ __launch_bounds__(256) void k_gw_bias(GwArgs A) {
  __shared__ float red[8 * 8];
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;       // 8 tiles in flight per workgroup, lane = point of the tile
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
  for (int c0 = 0; c0 < A.nc; c0 += 8) {                           // eight columns at a time: their loads are independent
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (long t = (long)blockIdx.x * 8 + sub; t < A.bias_ntiles; t += (long)gridDim.x * 8) {
      const float* q = A.SM + (t * A.nc + c0) * 32 + lane;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < A.nc) s[e] += q[e * 32];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      for (int off = 16; off > 0; off >>= 1) s[e] += __shfl_down(s[e], off, 32);
      if (lane == 0) red[sub * 8 + e] = s[e];
    }
    __syncthreads();
    if (threadIdx.x < 8 && c0 + (int)threadIdx.x < A.nc) {
      float v = 0.f;
      for (int w = 0; w < 8; ++w) v += red[w * 8 + threadIdx.x];
      if (c0 + (int)threadIdx.x < A.Bv.nout) prow[matref_index(A.Bv, 0, 0, c0 + threadIdx.x)] = v;
    }
    __syncthreads();
  }
}

void launch_gw_out(const GwArgs& a_, int NBI, int rows, hipStream_t st) {
  const GwArgs a = gw_fix(a_);
  if (a.W.nin == 0 && a.r == 0 && a.has_bias) { hipLaunchKernelGGL(k_gw_bias, dim3(rows), dim3(256), 0, st, a); return; }
  // few columns (e.g. so = 1, r = 1): the VALU kernel below is as fast; from 8 columns on the MFMA form wins big
  if (NIF_GW_EDGE_MFMA && (a.r + 1) * a.nc <= 32 && (a.r + 1) * a.nc >= 8) {
    dim3 grid(rows), block(256);
    if (NBI == 1) hipLaunchKernelGGL((k_gw_out_mfma<1, 4>), grid, block, 0, st, a);
    else if (NBI == 2) hipLaunchKernelGGL((k_gw_out_mfma<2, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_gw_out_mfma<4, 4>), grid, block, 0, st, a);
    return;
  }
  static const bool use_lds = [] { const char* e = getenv("NIF_GW_LDS"); return !(e && e[0] == '0'); }();
  if (use_lds && (a.r + 1) * a.nc <= 8 && (NBI <= 2 || NBI == 4)) {
    const int buf = NBI * 1024 + ((a.r * 32 + 63) & ~63) + ((a.nc * 32 + 63) & ~63);
    const size_t shl = sizeof(float) * (size_t)(4 * 2 * buf);
    if (NBI == 1) {
      (void)hipFuncSetAttribute((const void*)k_gw_out_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl);
      hipLaunchKernelGGL((k_gw_out_lds<1>), dim3(rows), dim3(256), shl, st, a);
    } else if (NBI == 2) {
      (void)hipFuncSetAttribute((const void*)k_gw_out_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl);
      hipLaunchKernelGGL((k_gw_out_lds<2>), dim3(rows), dim3(256), shl, st, a);
    } else {
      (void)hipFuncSetAttribute((const void*)k_gw_out_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl);
      hipLaunchKernelGGL((k_gw_out_lds<4>), dim3(rows), dim3(256), shl, st, a);
    }
    return;
  }
  dim3 grid(rows, a.r + 1), block(256);
  const size_t shm = (size_t)(192 + 4 * a.nc * (NBI + 1) * 64) * sizeof(float);
  if (NBI == 1) hipLaunchKernelGGL((k_gw_out<1>), grid, block, shm, st, a);
  else if (NBI == 2) hipLaunchKernelGGL((k_gw_out<2>), grid, block, shm, st, a);
  else hipLaunchKernelGGL((k_gw_out<4>), grid, block, shm, st, a);
}
