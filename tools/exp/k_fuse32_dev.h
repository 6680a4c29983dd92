// ---- r5: fp32 deposits ----------------------------------------------------------------------------------------------------------
// The producers deposit their tile's h and dL/da rows AS THEY HOLD THEM (fp32; lane (p, g): the 4 features 16 b + 4 g .. of point p are
// one 16-byte piece) and the CONSUMER waves, which idle most of a chunk step, form the bf16 (hi, lo) operand pairs (and zt h): r4's
// producers spent ~180 vector instructions per layer and tile on those splits -- a third of the vector work that the s_memtime
// timeline (profiles/r05_timeline_2buf.txt) shows as the blocks in which the matrix pipe idles.  Plane image = [16 points][64
// features] fp32 with 16 bytes of padding per point row (272 B): the producer's ds_write_b128 (8 consecutive points of one lane group
// per LDS cycle) and the consumer's ds_read_b32 (32 consecutive features of one point per cycle) are both conflict free, and the
// consumer's 8 reads are immediate offsets of ONE address.
#define FUSE32_ROW 272
#define FUSE32_PLANE (16 * FUSE32_ROW)
__device__ __forceinline__ void fuse32_deposit(char* img, int p, int g, const f32x4 (&v)[4]) {
#ifdef NIF_S6_NODEP      // measurement builds (results are wrong)
  return;
#endif
  f32x4* q = reinterpret_cast<f32x4*>(img + p * FUSE32_ROW + 16 * g);
#pragma unroll
  for (int b = 0; b < 4; ++b) q[4 * b] = v[b];
}
// consumer lane (i, hf) = (lane & 31, lane >> 5): byte offset of feature 32 blk + i, point 8 hf inside a plane image
__device__ __forceinline__ int fuse32_rd_off(int lane, int blk) { return (32 * blk + (lane & 31)) * 4 + (lane >> 5) * 8 * FUSE32_ROW; }
__device__ __forceinline__ void fuse32_read8(const char* img, int off, float (&x)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = *reinterpret_cast<const float*>(img + off + t * FUSE32_ROW);
}
// the lane's 8 points of a per-tile fp32 vector [16 points] (zt, du_o, x_c)
__device__ __forceinline__ void fuse32_vec8(const char* row, int lane, float (&x)[8]) {
  const f32x4* q = reinterpret_cast<const f32x4*>(row + (lane >> 5) * 32);
  const f32x4 a = q[0], b = q[1];
  x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
}
// bf16 (hi, lo) pair of 8 values: the MFMA operand of v_mfma_f32_32x32x16_bf16 (K = the lane's 8 points)
__device__ __forceinline__ void fuse32_split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const __bf16 x0 = (__bf16)x[t];
    hi[t] = x0; lo[t] = (__bf16)(x[t] - (float)x0);
  }
}
