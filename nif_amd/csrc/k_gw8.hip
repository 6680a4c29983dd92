// k_gw8.hip -- weight-gradient GEMM of a 128 x 128 hyper-matrix (K = batch), ONE read of each stash tile.
//
// k_gw_lds<4,2,1> needs 256 accumulator registers per wave for a 128 x 128 x (r+1 = 2) gradient, so its workgroups each
// own half of the output blocks and the layer-input stash is read twice (3 TB/s).  Here the
// accumulators are spread over the 8 waves of a workgroup instead (one workgroup per CU):
//   * the workgroup walks the 32-point tiles; a tile's IN and DA rows ([128 features][32 points] fp32 each) are loaded ONCE
//     by its 512 threads (coalesced 16-byte loads, the next tile's loads in flight during the products);
//   * every thread splits its 4-point pieces into bf16 hi / lo, the IN pieces once per plane (scaled by the latent factor
//     zt_k of the point for k < r), and stores them in LDS as MFMA operands [feature][32 points] (hi and lo planes, the
//     16-byte columns XOR-swizzled by the row so that the operand reads are bank-conflict free);
//   * after one barrier, wave w = (plane k, input block bi) multiplies its 32 input features with all four 32-feature output
//     blocks: hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, two K halves of 16 points: 24 MFMAs per tile, 64
//     accumulator registers; the packed operands are double buffered: one barrier per tile;
//   * bias gradients: every thread sums its own dL/da pieces (x zt_k) over the tiles it loads; one LDS reduction at the end;
//   * the workgroup writes its partial-gradient row like every other gradient kernel (k_reduce sums the rows).
// r = 1 (hypernetwork ShapeNets, 2 planes) or r = 0 (the dense ShapeNet of the last-layer class: the 8 waves split the 16
// blocks as (input block, output-block pair)).  Tangent pseudo-tiles of the Sobolev step (zt_mod / bias_ntiles) as in k_gw_lds.
#include "nif_internal.h"

#ifndef NIF_G8_ABL
#define NIF_G8_ABL 0      // measurement builds: bit 0 = no MFMAs, bit 1 = operand splits without the lo planes (results wrong).
                          // r4, cfg-3 (six 128 x 128 x 2-plane gradients, 512 k points): all gradient kernels 1.09 ms; without MFMAs 0.98, without
                          // lo-plane stores 1.04, neither 1.00 -- neither the matrix pipe nor the LDS writes bound k_gw8<1>, its load stream does
#endif
typedef __bf16 g8_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g8_bf16x4 __attribute__((ext_vector_type(4)));
typedef short g8_s16x4 __attribute__((ext_vector_type(4)));
typedef float g8_f32x16 __attribute__((ext_vector_type(16)));

// byte offset of points [p0, p0+4) of feature f inside a [128][32] bf16 plane (64-byte rows, 16-byte columns swizzled)
// swizzle key (f >> 2) & 3: rows f and f + 4 start at the same bank (64-byte rows, 256 bytes of banks), a 16-lane phase of a
// ds_read_b128 covers 16 consecutive rows = 4 rows per bank base, which the key sends to 4 different 16-byte columns (with the
// (f >> 1) key of the first version rows f and f + 8 collided: 1.0e7 conflict cycles next to 7.0e6 LDS issue cycles per launch;
// now 8e3 -- the kernel's time did not move, 1.12 -> 1.09 ms on cfg-3: it is not LDS-bound either)
__device__ __forceinline__ int g8_off(int f, int p0) { return f * 64 + ((((p0 >> 3) ^ (f >> 2)) & 3) << 4) + ((p0 & 7) << 1); }

// DAB (mixed_bfloat16, r3): the dL/da stash holds bf16 rows (k_snet4<8, .., PR> writes them) and the sums are the policy's: one
// product bf16(zt_k h_in) x dL/da per operand pair -- the dL/da pieces go to the operand plane as they are, no lo planes
// PH (r5, with DAB): the IN stash holds 16-bit PHASE rows (k_snet3_dev.h: q = rint(65536 f), f the reduced argument of the layer's
// sine in revolutions) -- 8 bytes per piece instead of 16, h = v_sin_f32(q / 65536) rebuilt here (4 sines per piece: the kernel is
// bound by its load stream, not by the VALU)
template <int R, bool DAB = false, bool PH = false>   // R = 1: planes k = 0 (x zt), 1; R = 0: one plane
__global__ __launch_bounds__(512, 1) void k_gw8(GwArgs A) {
  extern __shared__ __attribute__((aligned(16))) char g8sm[];
  constexpr int NPL = R + 1;
  constexpr int PLANE = 128 * 64;                          // bytes of one bf16 operand plane
  constexpr int SET = (2 * NPL + 2) * PLANE;               // A hi/lo per plane + B hi/lo
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i32 = lane & 31, kg = lane >> 5;
  // this wave's blocks
  const int uk = R ? wid >> 2 : 0;
  const int ubi = R ? (wid & 3) : (wid >> 1);
  const int ob0 = R ? 0 : 2 * (wid & 1);
  constexpr int NOB = R ? 4 : 2;
  g8_f32x16 acc[NOB];
#pragma unroll
  for (int o = 0; o < NOB; ++o)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
  // loader role: chunk c = tid + 512 j (j = 0, 1) of the tile's 1024 16-byte pieces: feature c >> 3, points 4 (c & 7) ..
  const int f0 = tid >> 3, f1 = f0 + 64, p0 = 4 * (tid & 7);
  float bsum[2][NPL];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < NPL; ++k) bsum[j][k] = 0.f;
  const int zt_mod = (int)A.zt_mod;
  const long nt = A.ntiles;
  // two register sets: while tile t is split and multiplied, tiles t+1 and t+2 (64 KB per CU) are in flight -- one tile
  // (32 KB) does not cover the HBM latency at 5 TB/s (measured 3.8 TB/s)
  // r4: the one-plane fp32 form takes three sets (cfg-4 128 x 6: 0.49 -> 0.47 ms per matrix); the two-plane forms lose with a
  // third set (registers: 1.12 -> 2.1 ms on cfg-3), the bf16-row forms do not move.  Also measured and not kept: the split of
  // tile t + 1 sharing the barrier interval with the products of tile t, the two waves of a SIMD in opposite order (+10 %)
  constexpr int NS = (R == 0 && (!DAB || PH)) ? 3 : 2;      // (r5: the one-plane phase-row form too: 1.735 -> 1.68 ms on configs[3] bf16; the two-plane one spills 42 registers with a third set: 0.755 -> 1.08 ms on cfg-3)
  f32x4 rin[PH ? 1 : NS][2], rda[DAB ? 1 : NS][2], rz[NS];
  g8_bf16x4 rdb[DAB ? NS : 1][2];
  g8_s16x4 rph[PH ? NS : 1][2];
#define G8_LOAD(SET_, T_)                                                                          \
  {                                                                                                \
    const float* in_ = A.IN + (T_) * (128 * 32);                                                   \
    const float* da_ = A.DA + (T_) * (128 * 32);                                                   \
    if (PH) {                                                                                      \
      const short* ph_ = reinterpret_cast<const short*>(A.IN) + (T_) * (128 * 32);                 \
      rph[PH ? SET_ : 0][0] = *reinterpret_cast<const g8_s16x4*>(ph_ + f0 * 32 + p0);              \
      rph[PH ? SET_ : 0][1] = *reinterpret_cast<const g8_s16x4*>(ph_ + f1 * 32 + p0);              \
    } else {                                                                                       \
      rin[PH ? 0 : SET_][0] = *reinterpret_cast<const f32x4*>(in_ + f0 * 32 + p0);                 \
      rin[PH ? 0 : SET_][1] = *reinterpret_cast<const f32x4*>(in_ + f1 * 32 + p0);                 \
    }                                                                                              \
    if (DAB) {                                                                                     \
      const __bf16* db_ = reinterpret_cast<const __bf16*>(A.DA) + (T_) * (128 * 32);               \
      rdb[DAB ? SET_ : 0][0] = *reinterpret_cast<const g8_bf16x4*>(db_ + f0 * 32 + p0);            \
      rdb[DAB ? SET_ : 0][1] = *reinterpret_cast<const g8_bf16x4*>(db_ + f1 * 32 + p0);            \
    } else {                                                                                       \
      rda[DAB ? 0 : SET_][0] = *reinterpret_cast<const f32x4*>(da_ + f0 * 32 + p0);                \
      rda[DAB ? 0 : SET_][1] = *reinterpret_cast<const f32x4*>(da_ + f1 * 32 + p0);                \
    }                                                                                              \
    if (R) {                                                                                       \
      const long tz_ = zt_mod >= nt ? (T_) : (T_) % zt_mod;                                        \
      rz[SET_] = *reinterpret_cast<const f32x4*>(A.Z + tz_ * 32 + p0);   /* r = 1: one latent row per tile */ \
    }                                                                                              \
  }
  auto split_store = [&](char* plane_hi, char* plane_lo, int f, const f32x4& v) {
    g8_bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const __bf16 h = (__bf16)v[e]; hi[e] = h; lo[e] = (__bf16)(v[e] - (float)h); }
    const int o = g8_off(f, p0);
    *reinterpret_cast<g8_bf16x4*>(plane_hi + o) = hi;
    if (!DAB && !(NIF_G8_ABL & 2)) *reinterpret_cast<g8_bf16x4*>(plane_lo + o) = lo;
  };
  // one tile: split register set RS into operand buffer `set`, barrier, refill RS with tile t + 2 grid, products
#define G8_TILE(RS)                                                                                                    \
  {                                                                                                                    \
    char* S = g8sm + set * SET;                                                                                        \
    const bool wbias = t < A.bias_ntiles;                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                    \
      const int f = j ? f1 : f0;                                                                                       \
      f32x4 dav;                                                                                                       \
      if (DAB) { _Pragma("unroll") for (int e = 0; e < 4; ++e) dav[e] = (float)rdb[DAB ? RS : 0][j][e]; }              \
      else dav = rda[DAB ? 0 : RS][j];                                                                                 \
      f32x4 hv;                                                                                                        \
      if (PH) { _Pragma("unroll") for (int e = 0; e < 4; ++e) hv[e] = __builtin_amdgcn_sinf((float)(int)rph[PH ? RS : 0][j][e] * (1.0f / 65536.0f)); } \
      else hv = rin[PH ? 0 : RS][j];                                                                                   \
      if (R) {                                                                                                         \
        const f32x4 zin = {hv[0] * rz[RS][0], hv[1] * rz[RS][1], hv[2] * rz[RS][2], hv[3] * rz[RS][3]};                \
        split_store(S + 0 * PLANE, S + 1 * PLANE, f, zin);              /* plane 0: zt h */                            \
        split_store(S + 2 * PLANE, S + 3 * PLANE, f, hv);               /* plane 1: h */                               \
        if (wbias) {                                                                                                   \
          bsum[j][0] += (dav[0] * rz[RS][0] + dav[1] * rz[RS][1]) + (dav[2] * rz[RS][2] + dav[3] * rz[RS][3]);         \
          bsum[j][1] += (dav[0] + dav[1]) + (dav[2] + dav[3]);                                                         \
        }                                                                                                              \
      } else {                                                                                                         \
        split_store(S + 0 * PLANE, S + 1 * PLANE, f, hv);                                                              \
        if (wbias) bsum[j][0] += (dav[0] + dav[1]) + (dav[2] + dav[3]);                                                \
      }                                                                                                                \
      if (DAB) *reinterpret_cast<g8_bf16x4*>(S + 2 * NPL * PLANE + g8_off(f, p0)) = rdb[DAB ? RS : 0][j];              \
      else split_store(S + 2 * NPL * PLANE, S + (2 * NPL + 1) * PLANE, f, dav);                                        \
    }                                                                                                                  \
    const long tn = t + NS * (long)gridDim.x;   /* the set's registers are dead once split: refill them BEFORE the barrier */ \
    if (tn < nt) G8_LOAD(RS, tn)                                                                                       \
    __syncthreads();   /* planes of `set` complete; the other buffer's readers finished before the previous barrier */ \
    const char* Ahi = S + (2 * uk) * PLANE, *Alo = Ahi + PLANE;                                                        \
    const char* Bhi = S + 2 * NPL * PLANE, *Blo = Bhi + PLANE;                                                         \
    const int fa = 32 * ubi + i32;                                                                                     \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                                 \
      const int ca = ((((2 * hh + kg) ^ (fa >> 2)) & 3) << 4);                                                         \
      const g8_bf16x8 ah = *reinterpret_cast<const g8_bf16x8*>(Ahi + fa * 64 + ca);                                    \
      const g8_bf16x8 al = *reinterpret_cast<const g8_bf16x8*>(Alo + fa * 64 + ca);                                    \
      _Pragma("unroll") for (int o = 0; o < NOB; ++o) {                                                                \
        const int fb = 32 * (ob0 + o) + i32;                                                                           \
        const int cb = ((((2 * hh + kg) ^ (fb >> 2)) & 3) << 4);                                                       \
        const g8_bf16x8 bh = *reinterpret_cast<const g8_bf16x8*>(Bhi + fb * 64 + cb);                                  \
        const g8_bf16x8 bl = *reinterpret_cast<const g8_bf16x8*>(Blo + fb * 64 + cb);                                  \
        if (NIF_G8_ABL & 1) { acc[o][0] += (float)ah[0] * (float)bl[1] + (float)al[2] * (float)bh[3]; }                \
        else {                                                                                                         \
        if (!DAB) {                                                                                                    \
          acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[o], 0, 0, 0);                                   \
          acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[o], 0, 0, 0);                                   \
        }                                                                                                              \
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[o], 0, 0, 0);                                     \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    t += gridDim.x; set ^= 1;                                                                                          \
  }
  long t = blockIdx.x;
  if (t < nt) G8_LOAD(0, t)
  if (t + gridDim.x < nt) G8_LOAD(1, t + gridDim.x)
  if constexpr (NS > 2) { if (t + 2 * (long)gridDim.x < nt) G8_LOAD(2, t + 2 * (long)gridDim.x) }
  int set = 0;
  while (t < nt) {
    G8_TILE(0)
    if (t >= nt) break;
    G8_TILE(1)
    if constexpr (NS > 2) {
      if (t >= nt) break;
      G8_TILE(2)
    }
  }
#undef G8_TILE
#undef G8_LOAD
  // ---- epilogue: this workgroup's partial row ----
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
#pragma unroll
  for (int o = 0; o < NOB; ++o) {
    const int out = 32 * (ob0 + o) + i32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int in = 32 * ubi + fmap(e, kg);
      if (in < A.W.nin && out < A.W.nout) prow[matref_index(A.W, uk, in, out)] = A.scale * acc[o][e];
    }
  }
  if (A.has_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(g8sm);            // [2 chunks][NPL][512]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < NPL; ++k) red[(j * NPL + k) * 512 + tid] = bsum[j][k];
    __syncthreads();
    // feature f = 64 j + (tid >> 3): the 8 threads of a feature are consecutive
    for (int e = tid; e < 2 * NPL * 64; e += 512) {
      const int j = e / (NPL * 64), k = (e / 64) % NPL, fl = e % 64;
      const float* q = red + (j * NPL + k) * 512 + fl * 8;
      const float v = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
      const int out = 64 * j + fl;
      if (out < A.Bv.nout) prow[matref_index(A.Bv, k, 0, out)] = v;
    }
  }
}


bool gw8_supported(const GwArgs& a, int NBI, int NBO) {
  return NBI == 4 && NBO == 4 && (a.r == 0 || a.r == 1);
}
void launch_gw8(const GwArgs& a, int rows, hipStream_t st) {
  if (a.da_bf16) {
    const size_t shm = (size_t)2 * (2 * (a.r + 1) + 2) * 128 * 64;
    if (a.in_ph16) {
      if (a.r == 1) {
        (void)hipFuncSetAttribute((const void*)k_gw8<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL((k_gw8<1, true, true>), dim3(rows), dim3(512), shm, st, a);
      } else {
        (void)hipFuncSetAttribute((const void*)k_gw8<0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL((k_gw8<0, true, true>), dim3(rows), dim3(512), shm, st, a);
      }
      return;
    }
    if (a.r == 1) {
      (void)hipFuncSetAttribute((const void*)k_gw8<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_gw8<1, true>), dim3(rows), dim3(512), shm, st, a);
    } else {
      (void)hipFuncSetAttribute((const void*)k_gw8<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      hipLaunchKernelGGL((k_gw8<0, true>), dim3(rows), dim3(512), shm, st, a);
    }
    return;
  }
  if (a.r == 1) {
    const size_t shm = (size_t)2 * (2 * 2 + 2) * 128 * 64;
    (void)hipFuncSetAttribute((const void*)k_gw8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_gw8<1>), dim3(rows), dim3(512), shm, st, a);
  } else {
    const size_t shm = (size_t)2 * (2 * 1 + 2) * 128 * 64;
    (void)hipFuncSetAttribute((const void*)k_gw8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_gw8<0>), dim3(rows), dim3(512), shm, st, a);
  }
}
