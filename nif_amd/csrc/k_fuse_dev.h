// k_fuse_dev.h -- the LDS exchange of the fused weight-gradient path (k_snet6.hip; tools/exp/tr_probe.hip checks it alone).
//
// A wave of the fused training kernel owns one 16-point tile with POINTS ON LANES (lane = (p, g): point p = lane & 15, features
// 16 b + 4 g + v of block b in registers).  The weight-gradient GEMM dM = sum_p h_in[., p] (x) dL/da[., p] has the points as its
// K dimension, so both operands are needed with FEATURES ON LANES.  The transposition goes through LDS:
//   * deposit: the bf16 (hi, lo) splits the kernel holds anyway as MFMA B operands (bf16x8 per K-step: 4 features of block 2ks,
//     4 of block 2ks+1) are written as they are, one ds_write_b64 per (point, feature quad) UNIT of 8 bytes;
//   * consume: ds_read_b64_tr_b16 -- every 16-lane group fetches a [4 points][16 features] block and the hardware hands lane i
//     feature i of the four points: two reads give the 8 K values of a v_mfma_f32_32x32x16_bf16 operand row.
// Unit (p, q) (q = feature quad 0..15) of a (tile, plane) image of 256 units = 2 KB sits at
//     ((p >> 2) * 2 + (q >> 3)) * 32 + (((p & 3) + 4 (q & 7) + 4 (p >> 2)) & 31)
// i.e. the 32 units one 32-lane half of a transpose read touches are one contiguous, 256-byte aligned group (all 64 banks once),
// rotated by the point group so that the 16 lanes of a ds_write_b64 pass (fixed quad, points 0..15) hit 16 distinct bank pairs.
#pragma once
#include "k_snet3_dev.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define FUSE_PLANE_UNITS 256                    // 8-byte units of one (tile, plane) image: 16 points x 16 feature quads
#define FUSE_PLANE_BYTES 2048

__host__ __device__ inline int fuse_unit(int p, int q) {
  return (((p >> 2) * 2 + (q >> 3)) << 5) + (((p & 3) + 4 * (q & 7) + 4 * (p >> 2)) & 31);
}

// producer lane (p, g): byte offsets of its units of the even / odd feature blocks (block b -> quad 4 b + g); blocks 2, 3 are
// 256 bytes further (q >> 3 = 1)
struct FuseDep { int u0, u1; };
__device__ __forceinline__ FuseDep fuse_dep_addr(int p, int g) {
  FuseDep d;
  d.u0 = fuse_unit(p, g) * 8;
  d.u1 = fuse_unit(p, 4 + g) * 8;
  return d;
}
// deposit one operand (NBL = 4: two K-steps of 8 features) of this lane into the plane image at `img` (LDS byte pointer)
__device__ __forceinline__ void fuse_deposit4(char* img, const FuseDep& d, const bf16x8 (&s)[2]) {
  typedef unsigned long long u64;
#ifdef NIF_S6_NODEP      // measurement builds (results are wrong)
  return;
#endif
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const u64 lo = __builtin_bit_cast(u64, __builtin_shufflevector(s[ks], s[ks], 0, 1, 2, 3));
    const u64 hi = __builtin_bit_cast(u64, __builtin_shufflevector(s[ks], s[ks], 4, 5, 6, 7));
    *reinterpret_cast<u64*>(img + d.u0 + 256 * ks) = lo;
    *reinterpret_cast<u64*>(img + d.u1 + 256 * ks) = hi;
  }
}

// consumer lane l: byte offsets of its two transpose reads (points 8 (l >> 5) + 4 rd .. + 3) inside the 32-feature block 0 of a
// plane image; block 1 is 256 bytes further
struct FuseRd { int a0, a1; };
__device__ __forceinline__ FuseRd fuse_rd_addr(int lane) {
  const int grp = lane >> 4, s = lane & 15;
  FuseRd r;
  const int q = 4 * (grp & 1) + (s & 3);
  r.a0 = fuse_unit(8 * (grp >> 1) + (s >> 2), q) * 8;
  r.a1 = fuse_unit(8 * (grp >> 1) + 4 + (s >> 2), q) * 8;
  return r;
}
// the MFMA operand (32 features of block `blk` on lanes & 31, the 16 points as K) of a plane image
__device__ __forceinline__ bf16x8 fuse_read_op(const char* img, const FuseRd& r, int blk) {
  const bf16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + r.a0 + 256 * blk));
  const bf16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + r.a1 + 256 * blk));
  return __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// sum over the lane's 8 points of (hi + lo) of an operand: the bias gradient's share of this lane (other K half on lane ^ 32)
__device__ __forceinline__ float fuse_sum8(const bf16x8 hi, const bf16x8 lo, float acc) {
  const __bf16 one = (__bf16)1.0f;
  const bf16x2 ones = {one, one};
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const bf16x2 a = {hi[e], hi[e + 1]}, b = {lo[e], lo[e + 1]};
    acc = __builtin_amdgcn_fdot2_f32_bf16(a, ones, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(b, ones, acc, false);
  }
  return acc;
}
// sum over the lane's 8 points of x (hi + lo) . w (whi + wlo), three bf16 products (the lo . lo term is dropped)
__device__ __forceinline__ float fuse_dot8(const bf16x8 hi, const bf16x8 lo, const bf16x8 whi, const bf16x8 wlo, float acc) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const bf16x2 a = {hi[e], hi[e + 1]}, b = {lo[e], lo[e + 1]}, c = {whi[e], whi[e + 1]}, d = {wlo[e], wlo[e + 1]};
    acc = __builtin_amdgcn_fdot2_f32_bf16(a, d, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(b, c, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(a, c, acc, false);
  }
  return acc;
}
