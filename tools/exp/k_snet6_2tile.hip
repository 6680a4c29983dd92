// EXPERIMENT (r5, not in the build): k_snet6 with TWO tiles per producer wave -- 4 producer + 4 consumer waves at 256 registers
// (230 used, 0 spilled), every A operand read from LDS feeds two MFMAs.  Parity green (439 GPU tests), but SLOWER than the r4 form:
// fused kernel 1.198 -> 1.373 ms, step 1.318 -> 1.527 ms (same box class, gpurun_out/r05_a.json).  Reading: with ONE producer wave per
// SIMD nothing overlaps that wave's VALU blocks (sine, splits, deposits) with matrix work -- the halved chunk reads do not buy back
// what the second producer wave of a SIMD hid.  The lever is overlap between waves of a SIMD, not LDS bytes per MFMA.

// k_snet6.hip -- the plain-SIREN training kernel (k_snet4<NBL, TRAIN, SINE, 0, SGN>) with EVERY ShapeNet weight gradient fused in:
// no dL/da stash, no weight-gradient launches (k_gw_first_lds, 4 x k_gw_lds, k_gw_out_lds), one partial-gradient row per workgroup.
//
// Why: k_snet4 writes 2.5 KB/point of h / dL/da rows that exist only so that the K = batch reductions
//     dL/dM_j^(k)[in][out] = w0 sum_p zt_k(p) h_j[in][p] dL/da_{j+1}[out][p]
// can run as separate HBM-bound kernels.  Here they are accumulated where both operands are live.
//
// r5 -- TWO TILES PER PRODUCER WAVE.  r4's form (8 producer + 8 consumer waves at 128 registers, one 16-point
// tile per producer) kept the matrix pipe 27 % busy: every 1 KB A operand (weights) a producer read from LDS fed ONE
// v_mfma_f32_16x16x32 (16 cycles on its SIMD) -- 4 SIMDs x 1 KB / 16 cycles = 256 B/clk, the LDS's whole bandwidth, so the
// chunk reads of the 8 lock-stepped producers and their MFMAs could only alternate, and the 32 spilled registers of the
// 128-register budget sat in the same loop.  Now:
//   * ONE workgroup of 8 waves per CU at 256 registers (2 per SIMD): 4 PRODUCER waves, each running k_snet4's tile program on TWO
//     16-point tiles -- every A operand read feeds two MFMAs (half the chunk reads per point, four independent accumulator chains
//     per output-block pair) -- and 4 CONSUMER waves that own the accumulators of all hidden matrices and planes
//     (nh (r+1) n^2 = 128 KB at 4 x 64, r = 1): wave (k, I) holds the 32 x 32 blocks (plane k, input block I, output blocks 0 and 1)
//     of EVERY hidden matrix, 2 x 16 accumulator registers per matrix, and reads each A operand (h / zt h block I) once for both;
//   * at the end of adjoint layer j a producer DEPOSITS its tiles' operands in LDS as bf16 (hi, lo) planes in the form it holds
//     MFMA B operands anyway (24 ds_write_b64 per layer and tile); during the chunk steps of layer j-1 the consumers run their
//     blocks over the 8 deposited tiles: ds_read_b64_tr_b16 hands the operands over with features on lanes (k_fuse_dev.h) --
//     12 transpose reads + 6 v_mfma_f32_32x32x16_bf16 (hi.lo + lo.hi + hi.hi per block, K = the tile's 16 points) per tile;
//     the chunk barriers that exist anyway order deposit and consumption (two extra barriers per tile round around the first layer);
//   * biases, the first layer (K = si) and the last layer (N = so) are v_dot2_f32_bf16 sums of the same transposed operands against
//     per-tile weight vectors (zt, 1, x_c, zt x_c, du_o as bf16 hi | lo rows of 16 points);
//   * what is left of the stash: the layer inputs h_1 .. h_{nh-1} of a tile (forward -> adjoint, re-read by the same wave) in a
//     private ring; h_0 is recomputed from the tile's inputs.
// Built for: NIFMultiScale without resblocks, fp32 results, 49..64 units (NBL = 4), latent_dim 1, 1..4 hidden matrices, si, so <= 3.
// Everything else keeps k_snet4 + k_gw_*.  nif_set_option("fuse_gw", 0) / NIF_FUSE_GW=0 switches back (A/B, tests).
#include "k_fuse_dev.h"

#define ZERO_T6(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

// private ring of a tile slot: [matrix j][feature][16 points]
template <int NBL>
__device__ __forceinline__ void ring_store16(float* __restrict__ slot, const f32x4 (&h)[NBL], int g, int p) {
#ifdef NIF_ABL_NOSTORE
  if (h[0][0] != 12345.678f) return;
#endif
  float* q = slot + 4 * g * 16 + p;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) q[(16 * b + v) * 16] = h[b][v];
}
template <int NBL>
__device__ __forceinline__ void ring_load16(const float* __restrict__ slot, f32x4 (&h)[NBL], int g, int p) {
#ifdef NIF_ABL_NOLOAD
  if (p != -12345) {
#pragma unroll
    for (int b = 0; b < NBL; ++b) { h[b][0] = 0.5f; h[b][1] = 0.25f; h[b][2] = 0.125f; h[b][3] = 0.75f; }
    return;
  }
#endif
  const float* q = slot + 4 * g * 16 + p;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) h[b][v] = q[(16 * b + v) * 16];
}

struct S6Args {
  SNetArgs s;
  float* partial; long pstride;     // partial-gradient rows [gridDim.x][pstride] (the ShapeNet = hypernetwork columns of them)
};

#ifndef NIF_S6_RECOMP0
#define NIF_S6_RECOMP0 1     // the first layer's output (input of hidden matrix 0) is recomputed in the adjoint from the tile's inputs
                             // (si FMAs + a sine per element) instead of going through the ring: a quarter of the ring traffic less
#endif
#ifndef NIF_S6_CONS_PRIO
#define NIF_S6_CONS_PRIO 0     // s_setprio of the consumer waves
#endif

// one K-step chunk of a forward plane for the TWO tiles of a producer wave: T[tt][ob] += sum over the chunk's 32 features, the
// 6-product fp32-exact form of mfma_x6 (k_snet3_dev.h) -- every A operand (weights, from LDS) is read once and multiplies both
// tiles' B operands; the four chains of an output-block pair interleave.  PR / CP as in mfma_x6
template <int NBL, int PR, bool CP>
__device__ __forceinline__ void mfma_x6_2(const bf16x8* cur, const bf16x8 (&b0)[2], const bf16x8 (&b1)[2], const bf16x8 (&b2)[2],
                                           f32x4 (&T)[2][NBL], int lane) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ob = 0; ob < NBL; ob += 2) {
    if (PR != 0) {
      const bf16x8 a0 = cur[(CP ? ob : ob * 3) * 64 + lane], c0 = cur[(CP ? ob + 1 : ob * 3 + 3) * 64 + lane];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (PR == 2) { T[tt][ob] = mfma_f16(a0, b0[tt], T[tt][ob]); T[tt][ob + 1] = mfma_f16(c0, b0[tt], T[tt][ob + 1]); }
        else {
          T[tt][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[tt], T[tt][ob], 0, 0, 0);
          T[tt][ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0[tt], T[tt][ob + 1], 0, 0, 0);
        }
      }
      continue;
    }
    const bf16x8 a0 = cur[(ob * 3 + 0) * 64 + lane], a1 = cur[(ob * 3 + 1) * 64 + lane], a2 = cur[(ob * 3 + 2) * 64 + lane];
    const bf16x8 c0 = cur[(ob * 3 + 3) * 64 + lane], c1 = cur[(ob * 3 + 4) * 64 + lane], c2 = cur[(ob * 3 + 5) * 64 + lane];
#define S6_P4(A_, C_, B_)                                                                             \
    T[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_[0], T[0][ob], 0, 0, 0);                 \
    T[0][ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(C_, B_[0], T[0][ob + 1], 0, 0, 0);         \
    T[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_[1], T[1][ob], 0, 0, 0);                 \
    T[1][ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(C_, B_[1], T[1][ob + 1], 0, 0, 0);
    S6_P4(a1, c1, b1) S6_P4(a0, c0, b2) S6_P4(a2, c2, b0) S6_P4(a0, c0, b1) S6_P4(a1, c1, b0) S6_P4(a0, c0, b0)
#undef S6_P4
  }
  __builtin_amdgcn_s_setprio(0);
}
// one K-step chunk of an adjoint plane for two tiles, 3-product form (mfma_x3); ZI: the chains start from zero
template <int NBL, int PR, bool ZI, bool CP>
__device__ __forceinline__ void mfma_x3_2(const bf16x8* cur, const bf16x8 (&b0)[2], const bf16x8 (&b1)[2], f32x4 (&T)[2][NBL], int lane) {
  __builtin_amdgcn_s_setprio(1);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ib = 0; ib < NBL; ib += 2) {
    if (PR != 0) {
      const bf16x8 a0 = cur[(CP ? ib : ib * 2) * 64 + lane], c0 = cur[(CP ? ib + 1 : ib * 2 + 2) * 64 + lane];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (PR == 2) { T[tt][ib] = mfma_f16(a0, b0[tt], ZI ? z4 : T[tt][ib]); T[tt][ib + 1] = mfma_f16(c0, b0[tt], ZI ? z4 : T[tt][ib + 1]); }
        else {
          T[tt][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[tt], ZI ? z4 : T[tt][ib], 0, 0, 0);
          T[tt][ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0[tt], ZI ? z4 : T[tt][ib + 1], 0, 0, 0);
        }
      }
      continue;
    }
    const bf16x8 a0 = cur[(ib * 2 + 0) * 64 + lane], a1 = cur[(ib * 2 + 1) * 64 + lane];
    const bf16x8 c0 = cur[(ib * 2 + 2) * 64 + lane], c1 = cur[(ib * 2 + 3) * 64 + lane];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      T[tt][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1[tt], ZI ? z4 : T[tt][ib], 0, 0, 0);
      T[tt][ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b1[tt], ZI ? z4 : T[tt][ib + 1], 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      T[tt][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0[tt], T[tt][ib], 0, 0, 0);
      T[tt][ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, b0[tt], T[tt][ib + 1], 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      T[tt][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[tt], T[tt][ib], 0, 0, 0);
      T[tt][ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0[tt], T[tt][ib + 1], 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// PR: the producers' hidden n x n products under a Keras policy -- 1 = mixed_bfloat16 (ONE bf16 product per operand pair),
// 2 = mixed_float16 (half operands, per-point loss scale on dL/da; k_snet4_dev.h) -- as in k_snet4<.., PR>.  The CONSUMER side is
// untouched: the deposits stay bf16 (hi, lo) pairs of the fp32 rows and the weight-gradient sums three products, i.e. the policy's
// weight gradients here are those of fp32 stash rows (the tests emulate it with stash_bf16 = False).
template <int NBL, int PR = 0>
__global__ __launch_bounds__(512, 2) void k_snet6(S6Args F) {
  extern __shared__ __attribute__((aligned(256))) char smem6[];
  const SNetArgs& A = F.s;
  constexpr int PWV = 4, TPW = 2, TILES = PWV * TPW, NT = 64 * PWV, r = 1;   // producer waves, tiles per producer wave, tiles per round, producer threads
  constexpr int NCH = NBL / 2;
  constexpr int CF = NBL * 3 * 64, CB = NBL * 2 * 64;   // 16-byte units per forward / adjoint chunk
  constexpr bool CP = PR != 0;                           // the policies' compact plane set (k_snet4_dev.h): one plane per block
  constexpr int CFH = CP ? NBL * 64 : CF, CBH = CP ? NBL * 64 : CB;
  constexpr int QF = (CF + NT - 1) / NT;
  constexpr int NBUF = 2;
  constexpr int NPL = 6;                                // planes per tile: h (hi, lo), zt h (hi, lo), dL/da (hi, lo)
  constexpr int EXT = NPL * FUSE_PLANE_BYTES;
  // per-tile weight vectors [hi 16 | lo 16] bf16 = 64 B.  Last layer (WVL): du_o (o < 3), zt, ones.  First layer (WVF), per plane k:
  // k * 4 + c = (zt | 1) x_c, k * 4 + 3 = (zt | 1)
  constexpr int NVL = 5, NVF = 8, WVLT = NVL * 64, WVFT = NVF * 64;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + TILES - 1) / TILES;

  char* EX = smem6;                                     // [tile 8][plane 6][2 KB]
  char* WVL = EX + TILES * EXT;
  char* WVF = WVL + TILES * WVLT;
  bf16x8* chunks = reinterpret_cast<bf16x8*>(WVF + TILES * WVFT);
  float* sm = reinterpret_cast<float*>(chunks + NBUF * CF);
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  const int CX = (si + 3) & ~3, CZ = (r + 3) & ~3, CY = (so + 3) & ~3;
  const int NI = (CX + CZ + CY + 4) * 16;
  const int pw = 2 * r * 64 + 2 * NI;                   // per-tile-slot LDS floats (producers)
  float* lsum = sm + sm_tot + (long)TILES * pw;
  constexpr int NP = 16 * NBL;
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  {   // prologue, all 8 waves: LDS image of the small hyper-vectors; the exchange images start as zeros (the first tile round
      // consumes a first-layer deposit that nobody made)
    const long s_wl = (long)si * n + (long)nh * n * n;
    const long s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
    for (int idx = tid; idx < (r + 1) * nsm; idx += 512) {
      const int k = idx / nsm, e = idx - k * nsm;
      float v = 0.f;
      if (e < o_wl) { const int dd = e / NP, f = e - dd * NP; if (f < n) v = A.omega * hyp3(A, k, (long)dd * n + f); }
      else if (e < o_b1) { const int o = (e - o_wl) / NP, f = (e - o_wl) - o * NP; if (f < n) v = hyp3(A, k, s_wl + (long)f * so + o); }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = hyp3(A, k, s_b1 + f); }
      else if (e < o_bl) { const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP; if (f < n) v = hyp3(A, k, s_bh + (long)j * n + f); }
      else if (e < o_bl + so) v = hyp3(A, k, s_bl + (e - o_bl));
      sm[idx] = v;
    }
    for (int idx = tid; idx < (TILES * (EXT + WVLT + WVFT)) / 16; idx += 512) reinterpret_cast<f32x4*>(EX)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (wid >= PWV) {
    // =====================================================================================================================
    // consumer wave (plane kk, input block bI): the 32 x 32 blocks (kk, bI, J = 0 / 1) of every hidden matrix; the bI = 1 waves
    // also the hidden biases of plane kk (sums of the B operands they hold anyway) and the last layer's bias, the bI = 0 waves
    // the first layer of plane kk, every wave rows 32 bI .. of the last layer
    // =====================================================================================================================
    const int cw = wid - PWV, kk = cw >> 1, bI = cw & 1;
    __syncthreads();
    if (tid - NT < TILES * 16) {     // the constant "ones" vectors (hi = 1, lo = 0) of every tile
      const int t = (tid - NT) >> 4, q = (tid - NT) & 15;
      reinterpret_cast<__bf16*>(WVL + t * WVLT)[4 * 32 + q] = (__bf16)1.0f;
      reinterpret_cast<__bf16*>(WVF + t * WVFT)[7 * 32 + q] = (__bf16)1.0f;
    }
    __builtin_amdgcn_s_setprio(NIF_S6_CONS_PRIO);
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][J][e] = 0.f;
    float bacc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // hidden biases (kk, J) -- bI = 1 waves
    float facc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, fbacc[2] = {0.f, 0.f};   // first layer (kk, columns 32 J ..) -- bI = 0 waves
    float lacc[3] = {0.f, 0.f, 0.f}, blacc[3] = {0.f, 0.f, 0.f};   // last layer (kk, rows 32 bI ..); its bias -- wave (kk, 1)
    FuseRd rdA = fuse_rd_addr(lane), rdB = rdA;
    rdA.a0 += (2 - 2 * kk) * FUSE_PLANE_BYTES + 256 * bI; rdA.a1 += (2 - 2 * kk) * FUSE_PLANE_BYTES + 256 * bI;   // plane 0: zt h, plane 1 (= r): h
    rdB.a0 += 4 * FUSE_PLANE_BYTES; rdB.a1 += 4 * FUSE_PLANE_BYTES;                                               // dL/da (block J: + 256 J)
    const int wofs = 16 * (lane >> 5);                  // this lane's 8 points inside a weight vector (bytes)

#define S6_CBAR()                                                             \
  {                                                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);        /* lgkmcnt(0): the transpose reads are back */ \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
  }
    // hidden matrix J_: this wave's two blocks over the deposited tiles [T0_, T1_).  One tile's operands ahead of the MFMAs (the
    // transpose reads of tile t + 1 are in flight while tile t multiplies)
#define S6_HID_LOAD(T_, AH_, AL_, BH_, BL_, CH_, CL_)                                                       \
  {                                                                                                         \
    const char* img_ = EX + (T_) * EXT;                                                                     \
    AH_ = fuse_read_op(img_, rdA, 0); AL_ = fuse_read_op(img_ + FUSE_PLANE_BYTES, rdA, 0);                  \
    BH_ = fuse_read_op(img_, rdB, 0); BL_ = fuse_read_op(img_ + FUSE_PLANE_BYTES, rdB, 0);                  \
    CH_ = fuse_read_op(img_, rdB, 1); CL_ = fuse_read_op(img_ + FUSE_PLANE_BYTES, rdB, 1);                  \
  }
#define S6_HID_TILES(J_, T0_, T1_)                                                                          \
  {                                                                                                         \
    bf16x8 ah_, al_, bh_, bl_, ch_, cl_, ah2_, al2_, bh2_, bl2_, ch2_, cl2_;                                \
    S6_HID_LOAD(T0_, ah_, al_, bh_, bl_, ch_, cl_)                                                          \
    _Pragma("unroll") for (int t_ = T0_; t_ < T1_; ++t_) {                                                  \
      if (t_ + 1 < T1_) S6_HID_LOAD(t_ + 1, ah2_, al2_, bh2_, bl2_, ch2_, cl2_)                             \
      acc[J_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_, acc[J_][0], 0, 0, 0);                  \
      acc[J_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, cl_, acc[J_][1], 0, 0, 0);                  \
      acc[J_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_, acc[J_][0], 0, 0, 0);                  \
      acc[J_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, ch_, acc[J_][1], 0, 0, 0);                  \
      acc[J_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_, acc[J_][0], 0, 0, 0);                  \
      acc[J_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, ch_, acc[J_][1], 0, 0, 0);                  \
      if (bI == 1) {                                                                                        \
        const char* w_ = WVL + t_ * WVLT + (3 + kk) * 64 + wofs;                                            \
        const bf16x8 wh_ = *reinterpret_cast<const bf16x8*>(w_), wl_ = *reinterpret_cast<const bf16x8*>(w_ + 32); \
        bacc[J_][0] = fuse_dot8(bh_, bl_, wh_, wl_, bacc[J_][0]);                                           \
        bacc[J_][1] = fuse_dot8(ch_, cl_, wh_, wl_, bacc[J_][1]);                                           \
      }                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      ah_ = ah2_; al_ = al2_; bh_ = bh2_; bl_ = bl2_; ch_ = ch2_; cl_ = cl2_;                               \
    }                                                                                                       \
  }
    // the four chunk steps of an adjoint layer with the consumption of hidden deposit DJ_ (a compile-time index: the accumulators
    // are never selected at run time -- a switch over them made hipcc copy and spill whole accumulators around every call)
#define S6_HID_LAYER(DJ_)                                                                                   \
  if (DJ_ < nh) {                                                                                           \
    S6_CBAR()                                                                                               \
    S6_DO(S6_HID_TILES(DJ_, 0, 3))                                                                          \
    S6_CBAR()                                                                                               \
    S6_DO(S6_HID_TILES(DJ_, 3, 6))                                                                          \
    S6_CBAR()                                                                                               \
    S6_DO(S6_HID_TILES(DJ_, 6, 8))                                                                          \
    S6_CBAR()                                                                                               \
  }
    // last layer (h_nh, zt h_nh deposited as the A planes, du_o as vectors).  The skinny sums run as ROLLED loops over the tiles:
    // unrolled, hipcc fetched the weight vectors of all tiles first and spilled the accumulators to make room
    auto consume_last = [&](int t0, int t1) __attribute__((always_inline)) {
#pragma clang loop unroll(disable)
      for (int t = t0; t < t1; ++t) {
        const char* img = EX + t * EXT;
        const bf16x8 ah = fuse_read_op(img, rdA, 0), al = fuse_read_op(img + FUSE_PLANE_BYTES, rdA, 0);
        const char* w = WVL + t * WVLT + wofs;
        lacc[0] = fuse_dot8(ah, al, *reinterpret_cast<const bf16x8*>(w), *reinterpret_cast<const bf16x8*>(w + 32), lacc[0]);
        if (so > 1) lacc[1] = fuse_dot8(ah, al, *reinterpret_cast<const bf16x8*>(w + 64), *reinterpret_cast<const bf16x8*>(w + 96), lacc[1]);
        if (so > 2) lacc[2] = fuse_dot8(ah, al, *reinterpret_cast<const bf16x8*>(w + 128), *reinterpret_cast<const bf16x8*>(w + 160), lacc[2]);
        if (bI == 1) {
          const char* z = w + (3 + kk) * 64;
          const bf16x8 zhi = *reinterpret_cast<const bf16x8*>(z), zlo = *reinterpret_cast<const bf16x8*>(z + 32);
          blacc[0] = fuse_dot8(*reinterpret_cast<const bf16x8*>(w), *reinterpret_cast<const bf16x8*>(w + 32), zhi, zlo, blacc[0]);
          if (so > 1) blacc[1] = fuse_dot8(*reinterpret_cast<const bf16x8*>(w + 64), *reinterpret_cast<const bf16x8*>(w + 96), zhi, zlo, blacc[1]);
          if (so > 2) blacc[2] = fuse_dot8(*reinterpret_cast<const bf16x8*>(w + 128), *reinterpret_cast<const bf16x8*>(w + 160), zhi, zlo, blacc[2]);
        }
      }
    };
    // first layer (dL/da_0 deposited as the B planes, (zt | 1) x_c and (zt | 1) as vectors)
    auto consume_first = [&](int t0, int t1) __attribute__((always_inline)) {
      if (bI == 0) {
#pragma clang loop unroll(disable)
        for (int t = t0; t < t1; ++t) {
          const char* img = EX + t * EXT;
          const char* w = WVF + t * WVFT + kk * 256 + wofs;
#pragma unroll
          for (int J = 0; J < 2; ++J) {
            const bf16x8 bh = fuse_read_op(img, rdB, J), bl = fuse_read_op(img + FUSE_PLANE_BYTES, rdB, J);
            fbacc[J] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w + 192), *reinterpret_cast<const bf16x8*>(w + 224), fbacc[J]);
            facc[0][J] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w), *reinterpret_cast<const bf16x8*>(w + 32), facc[0][J]);
            if (si > 1) facc[1][J] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w + 64), *reinterpret_cast<const bf16x8*>(w + 96), facc[1][J]);
            if (si > 2) facc[2][J] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w + 128), *reinterpret_cast<const bf16x8*>(w + 160), facc[2][J]);
          }
        }
      }
    };
#ifdef NIF_S6_NOCONS
#define S6_DO(...)
#else
#define S6_DO(...) __VA_ARGS__
#endif
    // the barrier sequence of the producers' tile program, with this wave's share of the products between the barriers
    for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
      for (int j = 0; j < nh; ++j) {        // forward: the previous round's first-layer deposit next to hidden matrix 0
        S6_CBAR()
        if (j == 0) { S6_DO(consume_first(0, 4);) }
        S6_CBAR()
        if (j == 0) { S6_DO(consume_first(4, 8);) }
        S6_CBAR()
        S6_CBAR()
      }
      // adjoint: the last layer's deposit next to the steps of layer nh - 1, then deposit j + 1 next to layer j
      S6_CBAR()
      S6_DO(consume_last(0, 3);)
      S6_CBAR()
      S6_DO(consume_last(3, 6);)
      S6_CBAR()
      S6_DO(consume_last(6, 8);)
      S6_CBAR()
      S6_HID_LAYER(3) S6_HID_LAYER(2) S6_HID_LAYER(1)
      S6_CBAR()                              // deposit 0 next to the first layer's adjoint
      S6_DO(S6_HID_TILES(0, 0, 8))
      S6_CBAR()
    }
    __syncthreads();
    S6_DO(consume_first(0, 8);)
    __syncthreads();
#undef S6_DO
#undef S6_HID_LAYER
#undef S6_HID_TILES
#undef S6_HID_LOAD
#undef S6_CBAR
    // ---- this wave's entries of the workgroup's partial-gradient row (no reduction: every entry belongs to one wave) ---------------
    float* prow = F.partial + (long)blockIdx.x * F.pstride;
    const int i = lane & 31, hf = lane >> 5;
    const float om = A.omega;
    auto gidx = [&](long slot) -> long { return (kk < r ? A.off_Wh + (long)kk * A.po : A.off_bh) + slot; };
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nh) {
        const long ws = slot_wh(A, j);
#pragma unroll
        for (int J = 0; J < 2; ++J) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int in = 32 * bI + fmap(e, hf), out = 32 * J + i;
            if (in < n && out < n) prow[gidx(ws + (long)in * n + out)] = om * acc[j][J][e];
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (hipcc would form all 64 addresses first)
          }
          float v = bacc[j][J];
          v += __shfl_xor(v, 32);
          if (bI == 1 && hf == 0 && 32 * J + i < n) prow[gidx(slot_bh(A, j) + 32 * J + i)] = v;
        }
      }
#pragma unroll
    for (int J = 0; J < 2; ++J) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = facc[c][J];
        v += __shfl_xor(v, 32);
        if (c < si && bI == 0 && hf == 0 && 32 * J + i < n) prow[gidx((long)c * n + 32 * J + i)] = om * v;
      }
      float v = fbacc[J];
      v += __shfl_xor(v, 32);
      if (bI == 0 && hf == 0 && 32 * J + i < n) prow[gidx(slot_b1(A) + 32 * J + i)] = v;
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = lacc[o], w = blacc[o];
      v += __shfl_xor(v, 32);
      w += __shfl_xor(w, 32);
      if (o < so && hf == 0 && 32 * bI + i < n) prow[gidx(slot_wl(A) + (long)(32 * bI + i) * so + o)] = v;
      if (o < so && bI == 1 && lane == 0) prow[gidx(slot_bl(A) + o)] = w;
    }
    __syncthreads();          // (the producers' loss reduction)
    return;
  }

  // =======================================================================================================================
  // producer wave = two 16-point tiles per round (tile slots 2 wid, 2 wid + 1): k_snet4's tile program + the deposits
  // =======================================================================================================================
  const int ts0 = TPW * wid;
  float* dzs[TPW]; float* sks[TPW]; float* inp[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    dzs[tt] = sm + sm_tot + (long)(ts0 + tt) * pw;
    sks[tt] = dzs[tt] + r * 64;
    inp[tt] = sks[tt] + r * 64;
  }
  // ---- the chunk stream (k_snet4): forward planes of all hidden matrices, then the adjoint planes of matrix nh-1 .. 0 -----------
  const int NPC = (r + 1) * NCH;
  const bf16x8* cs_src = reinterpret_cast<const bf16x8*>(A.WF4);
  int cs_units = CFH, cs_left = nh * NPC, cs_phase = 0;
  long cs_groups = (ngroups - 1 - (long)blockIdx.x) / gridDim.x;
  auto cs_phase_step = [&]() {
    ++cs_phase;
    if (cs_phase < 1 + nh) {
      cs_src = reinterpret_cast<const bf16x8*>(A.WB4) + (long)(nh - 1 - (cs_phase - 1)) * NPC * CBH; cs_units = CBH; cs_left = NPC; return;
    }
    if (cs_groups <= 0) { cs_left = -1; return; }
    --cs_groups; cs_phase = 0;
    cs_src = reinterpret_cast<const bf16x8*>(A.WF4); cs_units = CFH; cs_left = nh * NPC;
  };
  auto cs_next = [&](int buf) {
    if (cs_left < 0) return;
    bf16x8* dst = chunks + buf * CF;
#pragma unroll
    for (int q = 0; q < QF; ++q)
      if (wid * 64 + NT * q < cs_units)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cs_src + tid + NT * q),
                                         (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);
    asm volatile("" ::: "memory");
    cs_src += cs_units;
    if (--cs_left == 0) cs_phase_step();
  };
  auto prefetch_inputs = [&](long tgn, int set) {
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      long t16n = tgn * TILES + ts0 + tt;
      if (t16n >= nt16) t16n = nt16 - 1;
      const long tile32n = t16n >> 1;
      const int poffn = 16 * (int)(t16n & 1) + p;
      long ptn = t16n * 16 + p;
      if (ptn >= A.B) ptn = A.B - 1;
      float* dst = inp[tt] + set * NI;
      for (int i0 = 0; i0 < CX; i0 += 4) {
        const int c = i0 + g < si ? i0 + g : si - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.xin + ptn * A.ncol + A.col0 + c),
                                         (__attribute__((address_space(3))) void*)(dst + i0 * 16), 4, 0, 0);
      }
      for (int i0 = 0; i0 < CZ; i0 += 4) {
        const int c = i0 + g < r ? i0 + g : r - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + (tile32n * r + c) * 32 + poffn),
                                         (__attribute__((address_space(3))) void*)(dst + (CX + i0) * 16), 4, 0, 0);
      }
      for (int i0 = 0; i0 < CY; i0 += 4) {
        const int c = i0 + g < so ? i0 + g : so - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.y + ptn * so + c),
                                         (__attribute__((address_space(3))) void*)(dst + (CX + CZ + i0) * 16), 4, 0, 0);
      }
      const float* swp = A.sw ? A.sw + ptn : A.y + ptn * so;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)swp,
                                       (__attribute__((address_space(3))) void*)(dst + (CX + CZ + CY) * 16), 4, 0, 0);
    }
  };
  prefetch_inputs(blockIdx.x, 0);
  if (cs_left <= 0) cs_left = -1;
  cs_next(0);
  __syncthreads();
  int cbuf = 0, nbuf = 1;
  float loss_lane = 0.f;
  float* ring0 = A.stash + ((long)blockIdx.x * TILES + ts0) * (long)nh * (NP * 16);    // [tile slot][matrix][NP features][16 points]
  const long ring_ts = (long)nh * (NP * 16);
  const FuseDep dep = fuse_dep_addr(p, g);
  char* exw0 = EX + ts0 * EXT;                           // this wave's tile images

#define S6_CHUNK(...)                                                         \
  {                                                                           \
    cs_next(nbuf);                                                            \
    const bf16x8* cur = chunks + cbuf * CF;                                   \
    __VA_ARGS__                                                               \
    __builtin_amdgcn_s_waitcnt(0x0070);        /* vmcnt(0) lgkmcnt(0): the chunk DMA has landed, the deposits are visible */ \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    cbuf ^= 1; nbuf ^= 1;                                                     \
  }

  int iset = 0;
  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++iset) {
    bool active[TPW], valid[TPW];
    long tile32[TPW]; int poff[TPW];
    const float *xs[TPW], *ys[TPW], *wsp[TPW], *zt_base[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      const long t16_raw = tg * TILES + ts0 + tt;
      active[tt] = t16_raw < nt16;
      const long t16 = active[tt] ? t16_raw : nt16 - 1;
      tile32[tt] = t16 >> 1;
      poff[tt] = 16 * (int)(t16 & 1) + p;
      const long pt = t16 * 16 + p;
      valid[tt] = active[tt] && pt < A.B;
      const float* zs = inp[tt] + (iset & 1) * NI + CX * 16;
      xs[tt] = inp[tt] + (iset & 1) * NI + p;
      ys[tt] = zs + CZ * 16 + p;
      wsp[tt] = zs + (CZ + CY) * 16 + p;
      zt_base[tt] = zs + p;
      dzs[tt][lane] = 0.f;
    }

    f32x4 h[TPW][NBL], acc[TPW][NBL];
    // ---- first layer ----------------------------------------------------------------------------------------------------
    auto first_layer = [&](int tt, f32x4 (&out)[NBL]) __attribute__((always_inline)) {
      f32x4 a_[NBL];
      {
        const float* s0 = sm + r * nsm + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          for (int dd = 0; dd < si; ++dd) s += xs[tt][dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          a_[b] = s;
        }
      }
      {
        const float zt = zt_base[tt][0];
        const float* s0 = sm + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          for (int dd = 0; dd < si; ++dd) s += xs[tt][dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          a_[b] += zt * s;
        }
      }
      sine16_tag<NBL>(a_, out);
    };
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) first_layer(tt, h[tt]);
    prefetch_inputs(tg + gridDim.x, (iset + 1) & 1);
    // ---- hidden hyper-matrices, forward ---------------------------------------------------------------------------------------
    for (int j = 0; j < nh; ++j) {
      bf16x8 b0[NCH][TPW], b1[NCH][TPW], b2[NCH][TPW];       // [K-step][tile]
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        bf16x8 s0[NCH], s1[NCH], s2[NCH];
        split3p<NBL, PR>(h[tt], s0, s1, s2);
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) { b0[ks][tt] = s0[ks]; b1[ks][tt] = s1[ks]; b2[ks][tt] = s2[ks]; }
      }
      f32x4 T[TPW][NBL];
      {
        const float* sb = sm + r * nsm + o_bh + j * NP + 4 * g;
        const float* sc = sm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + 16 * b), v0 = *reinterpret_cast<const f32x4*>(sc + 16 * b);
#pragma unroll
          for (int tt = 0; tt < TPW; ++tt) { acc[tt][b] = v1; T[tt][b] = v0; }
        }
      }
      if (!NIF_S6_RECOMP0 || j > 0) {
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) ring_store16<NBL>(ring0 + tt * ring_ts + j * (NP * 16), h[tt], g, p);
      }
      S6_CHUNK({ mfma_x6_2<NBL, PR, CP>(cur, b0[0], b1[0], b2[0], T, lane); })
      S6_CHUNK({ mfma_x6_2<NBL, PR, CP>(cur, b0[1], b1[1], b2[1], T, lane); })
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        const float zt = zt_base[tt][0];
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[tt][b] += zt * T[tt][b];
      }
      S6_CHUNK({ mfma_x6_2<NBL, PR, CP>(cur, b0[0], b1[0], b2[0], acc, lane); })
      S6_CHUNK({ mfma_x6_2<NBL, PR, CP>(cur, b0[1], b1[1], b2[1], acc, lane); })
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) sine16_tag<NBL>(acc[tt], h[tt]);
    }
    // ---- last layer (n -> so, linear), MSE, start of the adjoint ---------------------------------------------------------------
    f32x4 gh[TPW][NBL];
    float zt0[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      ZERO_T6(gh[tt])
      const float wsamp = (valid[tt] ? (A.sw ? wsp[tt][0] : 1.0f) : 0.0f);
      zt0[tt] = zt_base[tt][0];
      float se = 0.f;
      for (int o = 0; o < so; ++o) {
        f32x4 wg[NBL];
        ZERO_T6(wg)
        float part = 0.f, bias = 0.f;
#pragma unroll
        for (int k = 0; k <= r; ++k) {
          const float zt = k < r ? zt0[tt] : 1.0f;
          const float* s0 = sm + k * nsm;
          float sk = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
            sk += (h[tt][b][0] * w[0] + h[tt][b][1] * w[1]) + (h[tt][b][2] * w[2] + h[tt][b][3] * w[3]);
            wg[b] += zt * w;
          }
          part = fmaf(zt, sk, part);
          bias = fmaf(zt, s0[o_bl + o], bias);
          if (k < r) sks[tt][k * 64 + lane] = sk;
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float uo = part + bias;
        const float e = uo - ys[tt][o * 16];
        NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
        const float du = dfac * wsamp * A.inv_bg / (float)so;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[tt][b] += du * wg[b];
        {
          float t = du * sks[tt][lane];
          if (g == 0) t = fmaf(du, sm[o_bl + o], t);
          dzs[tt][lane] += t;
        }
        if (g == 0 && o < 3) {     // du_o of the tile's 16 points as a bf16 (hi | lo) row: the last layer's weight-gradient vector
          __bf16* wv = reinterpret_cast<__bf16*>(WVL + (ts0 + tt) * WVLT);
          const __bf16 d0 = (__bf16)du;
          wv[o * 32 + p] = d0; wv[o * 32 + 16 + p] = (__bf16)(du - (float)d0);
        }
      }
      if (g == 1) {     // zt of the tile (the hidden layers' plane-0 bias sums use it too)
        __bf16* wv = reinterpret_cast<__bf16*>(WVL + (ts0 + tt) * WVLT);
        const __bf16 z0 = (__bf16)zt0[tt];
        wv[3 * 32 + p] = z0; wv[3 * 32 + 16 + p] = (__bf16)(zt0[tt] - (float)z0);
      }
      {   // deposit "nh": the last layer's input h_nh (and zt h_nh) as the A planes
        char* exw = exw0 + tt * EXT;
        bf16x8 a0[NCH], a1[NCH];
        split2<NBL>(h[tt], a0, a1);
        fuse_deposit4(exw, dep, a0);
        fuse_deposit4(exw + FUSE_PLANE_BYTES, dep, a1);
        f32x4 zh[NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) zh[b] = zt0[tt] * h[tt][b];
        split2<NBL>(zh, a0, a1);
        fuse_deposit4(exw + 2 * FUSE_PLANE_BYTES, dep, a0);
        fuse_deposit4(exw + 3 * FUSE_PLANE_BYTES, dep, a1);
      }
      if (g == 0) loss_lane += wsamp * se / (float)so * A.inv_bg;
    }
    // ---- adjoint through the hidden hyper-matrices ---------------------------------------------------------------------------
    // hin = the (tagged) sine of the layer above = h[tt] at first
    for (int j = nh - 1; j >= 0; --j) {
      bf16x8 q0[NCH][TPW], b1[NCH][TPW];        // the products' operands [K-step][tile]
      bf16x8 d0[TPW][NCH], d1[TPW][NCH];        // the deposit's (hi, lo) pair of dL/da
      float ils[TPW];
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        f32x4 ga[NBL], dnext[NBL];
        tag_cos<NBL>(h[tt], dnext);
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[tt][b];
        {
          const float* sb = sm + o_bh + j * NP + 4 * g;
          float sbv = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
            sbv += (ga[b][0] * bb[0] + ga[b][1] * bb[1]) + (ga[b][2] * bb[2] + ga[b][3] * bb[3]);
          }
          dzs[tt][lane] += sbv;
        }
        split2<NBL>(ga, d0[tt], d1[tt]);        // the deposit's (hi, lo) pair; d0 is also the bf16 policy's operand
        ils[tt] = 1.0f;
        if (PR == 2) {      // mixed_float16: half(s dL/da), s a power of two per point
          float mx = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ga[b][0]), fabsf(ga[b][1]))), fmaxf(fabsf(ga[b][2]), fabsf(ga[b][3])));
          mx = fmaxf(mx, __shfl_xor(mx, 16));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const unsigned ef = (__float_as_uint(mx) >> 23) & 0xFFu;
          const unsigned sf = 268u - ef < 227u ? 268u - ef : 227u;
          ils[tt] = __uint_as_float((254u - sf) << 23);
          bf16x8 qq[NCH];
          cast_f16<NBL>(ga, qq, __uint_as_float(sf << 23));
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) q0[ks][tt] = qq[ks];
        } else {
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) q0[ks][tt] = d0[tt][ks];
        }
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) b1[ks][tt] = d1[tt][ks];
      }
      // h_j (dz dot product, this layer's A planes, the cosine of the layer below): from the ring, or recomputed (j = 0)
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        if (NIF_S6_RECOMP0 && j == 0) first_layer(tt, h[tt]);
        else ring_load16<NBL>(ring0 + tt * ring_ts + j * (NP * 16), h[tt], g, p);
      }
      {
        f32x4 U[TPW][NBL];
        S6_CHUNK({ mfma_x3_2<NBL, PR, true, CP>(cur, q0[0], b1[0], U, lane); })
        S6_CHUNK({ mfma_x3_2<NBL, PR, false, CP>(cur, q0[1], b1[1], U, lane); })
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) s = fmaf(h[tt][b][v], U[tt][b][v], s);
#pragma unroll
          for (int b = 0; b < NBL; ++b) gh[tt][b] = zt0[tt] * U[tt][b];
          dzs[tt][lane] += PR == 2 ? ils[tt] * s : s;
        }
      }
      S6_CHUNK({ mfma_x3_2<NBL, PR, false, CP>(cur, q0[0], b1[0], gh, lane); })
      S6_CHUNK({ mfma_x3_2<NBL, PR, false, CP>(cur, q0[1], b1[1], gh, lane); })
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        if (PR == 2) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) gh[tt][b] *= ils[tt];
        }
        // deposit j: (h_j ; zt h_j ; dL/da) of this tile -- the consumer waves take it during the chunk steps of layer j - 1
        char* exw = exw0 + tt * EXT;
        fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, d0[tt]);
        fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, d1[tt]);
        bf16x8 a0[NCH], a1[NCH];
        split2<NBL>(h[tt], a0, a1);
        fuse_deposit4(exw, dep, a0);
        fuse_deposit4(exw + FUSE_PLANE_BYTES, dep, a1);
        f32x4 zh[NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) zh[b] = zt0[tt] * h[tt][b];
        split2<NBL>(zh, a0, a1);
        fuse_deposit4(exw + 2 * FUSE_PLANE_BYTES, dep, a0);
        fuse_deposit4(exw + 3 * FUSE_PLANE_BYTES, dep, a1);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the deposits have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- first layer (the consumer waves take deposit 0 meanwhile) ------------------------------------------------------------
    {
      bf16x8 d0[TPW][NCH], d1[TPW][NCH];
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        f32x4 ga[NBL], dnext[NBL];
        tag_cos<NBL>(h[tt], dnext);
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[tt][b];
        {
          const float* s0 = sm + 4 * g;
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            f32x4 t = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
            for (int dd = 0; dd < si; ++dd) t += xs[tt][dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
            s += (ga[b][0] * t[0] + ga[b][1] * t[1]) + (ga[b][2] * t[2] + ga[b][3] * t[3]);
          }
          float tot = dzs[tt][lane] + s;
          tot += __shfl_xor(tot, 16);
          tot += __shfl_xor(tot, 32);
          if (active[tt] && g == 0) A.DZ[(tile32[tt] * r) * 32 + poff[tt]] = tot;
        }
        split2<NBL>(ga, d0[tt], d1[tt]);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();          // deposit 0 has been consumed
      asm volatile("" ::: "memory");
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        char* exw = exw0 + tt * EXT;
        fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, d0[tt]);
        fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, d1[tt]);
        {      // lane group g < si: x_g and zt x_g of the tile's 16 points as bf16 (hi | lo) rows; group 3: zt
          __bf16* wv = reinterpret_cast<__bf16*>(WVF + (ts0 + tt) * WVFT);
          const float x = g < si ? xs[tt][g * 16] : 1.0f;
          const float zx = zt0[tt] * x;
          const __bf16 x0 = (__bf16)x, z0 = (__bf16)zx;
          if (g < si || g == 3) { wv[g * 32 + p] = z0; wv[g * 32 + 16 + p] = (__bf16)(zx - (float)z0); }
          if (g < si && g < 3) { wv[(4 + g) * 32 + p] = x0; wv[(4 + g) * 32 + 16 + p] = (__bf16)(x - (float)x0); }
        }
      }
    }
  }
#undef S6_CHUNK
  __syncthreads();          // the last round's first-layer deposit is visible ...
  __syncthreads();          // ... and consumed
  for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
  if (lane == 0) lsum[wid] = loss_lane;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < PWV; ++w) s += lsum[w];
    A.loss_partial[blockIdx.x] = s;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static size_t snet6_shmem(const SNetArgs& a, int NBL) {
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + ((a.r + 3) & ~3) + ((a.so + 3) & ~3) + 4) * 16;
  const size_t pw = 2 * a.r * 64 + 2 * ni;
  return 8 * (6 * FUSE_PLANE_BYTES + (5 + 8) * 64) + 2 * (size_t)NBL * 3 * 64 * 16 + (sm_tot + 8 * pw + 16) * sizeof(float);
}
// the fused-gradient kernel takes this training step (plain NIFMultiScale, fp32 results)
bool snet6_supported(const SNetArgs& a) {
  if (a.ll || a.res || a.nif_skip) return false;
  if (a.prec != 0) {     // the policy forms: NIF_S6_POLICY=0 keeps the r3 policy step (k_snet4<PR> + bf16 dL/da stash + k_gw_lds<DAB>) for A/B
    static const bool pol = [] { const char* e = getenv("NIF_S6_POLICY"); return !(e && e[0] == '0'); }();
    if (!pol) return false;
  }
  if (snet3_nbl(a.n) != 4 || a.r != 1 || a.nh < 1 || a.nh > 4 || a.si > 3 || a.so > 3) return false;
  return snet6_shmem(a, 4) <= 160u * 1024u;
}
// workgroups = partial-gradient rows = loss partials of the launch
int snet6_rows(const SNetArgs& a) {
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 7) / 8;
  return (int)(ngroups < 256 ? ngroups : 256);
}
int launch_snet6(const SNetArgs& a, float* partial, long pstride, hipStream_t st) {
  const int nblk = snet6_rows(a);
  S6Args f; f.s = a; f.partial = partial; f.pstride = pstride;
  const size_t shm = snet6_shmem(a, 4);
#define S6L(PR_)                                                                                                  \
  {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)k_snet6<4, PR_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((k_snet6<4, PR_>), dim3(nblk), dim3(512), shm, st, f);                                     \
  }
  if (a.prec == 2) { f.s.WF4 = a.WF4h; f.s.WB4 = a.WB4h; S6L(2) }      // the policy's compact plane set (k_pack16b mode 2 / 1)
  else if (a.prec == 1) { f.s.WF4 = a.WF4h; f.s.WB4 = a.WB4h; S6L(1) }
  else S6L(0)
#undef S6L
  return nblk;
}
