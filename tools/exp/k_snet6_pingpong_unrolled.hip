// k_snet6.hip -- the plain-SIREN training kernel (k_snet4<NBL, TRAIN, SINE, 0, SGN>) with EVERY ShapeNet weight gradient fused in:
// no dL/da stash, no weight-gradient launches (k_gw_first_lds, 4 x k_gw_lds, k_gw_out_lds), one partial-gradient row per workgroup.
//
// Why: k_snet4 writes 2.5 KB/point of h / dL/da rows that exist only so that the K = batch reductions
//     dL/dM_j^(k)[in][out] = w0 sum_p zt_k(p) h_j[in][p] dL/da_{j+1}[out][p]
// can run as separate HBM-bound kernels.  Here they are accumulated where both operands are live:
//   * ONE workgroup of 16 waves per CU at 128 registers: 8 PRODUCER waves run k_snet4's tile program (one 16-point tile per wave and
//     round) and DEPOSIT, at the end of adjoint layer j, their tile's operands in LDS as bf16 (hi, lo) planes in the layout they
//     hold MFMA B operands in anyway -- h_j, zt h_j, dL/da_{j+1}: 6 planes x 2 KB per tile, 24 ds_write_b64; 8 CONSUMER waves own the
//     accumulators (nh (r+1) n^2 = 128 KB at 4 x 64, r = 1): wave (k, I, J) holds the 32 x 32 block (plane k, input block I, output
//     block J) of EVERY hidden matrix, reads the deposited tiles with ds_read_b64_tr_b16 (features on lanes, k_fuse_dev.h) and runs
//     3 v_mfma_f32_32x32x16_bf16 per tile (hi.lo + lo.hi + hi.hi, K = the tile's 16 points);
//   * biases, the first layer (K = si) and the last layer (N = so) are v_dot2_f32_bf16 sums of the same transposed operands against
//     per-tile weight vectors (zt, 1, x_c, zt x_c, du_o as bf16 hi | lo rows of 16 points);
//   * what is left of the stash: the layer inputs h_1 .. h_{nh-1} of the wave's own tile between its forward and adjoint sweep, in a
//     private ring [matrix][feature][16 points]; h_0 is recomputed from the tile's inputs.
//
// r5: PING-PONG.  r4 ran all 8 producers in lock step -- one barrier per K-step chunk, every wave reading its
// A operands, multiplying and then doing its activation / split arithmetic at the same time as all the others, so that LDS, matrix
// pipe and VALU took turns (27 % / 36 % / 43 % busy, adding up to the whole kernel).  Now the producers are two GROUPS of four (one
// wave of each group per SIMD) that run the same program ONE BARRIER INTERVAL APART, and the program alternates strictly between
//     M items: the two K-step chunks of one plane (24 MFMAs per wave, nothing else), and
//     V items: everything between two planes' products (the latent combine; or sine + operand splits + ring traffic + the next
//              layer's bias loads; or the adjoint's cosine, dL/da, loss scale, splits, deposits),
// so that in every interval one wave of a SIMD multiplies while its partner does vector work.  The chunk stream runs in PAIRS (a
// plane = two chunks) through two pair buffers: pair q is multiplied by group 0 in interval 2q and by group 1 in interval 2q + 1,
// pair q + 1 is DMA'd during those two intervals (every producer wave issues its slice in interval 2q and waits for it in front of
// the barrier that ends interval 2q + 1).  The consumers walk the same barrier sequence, two deposited tiles per interval.
// Products: fp32-exact on HALF pairs (below).
// Built for: NIFMultiScale without resblocks, fp32 results, 49..64 units (NBL = 4), latent_dim 1, 1..4 hidden matrices, si, so <= 3.
// Everything else keeps k_snet4 + k_gw_*.  nif_set_option("fuse_gw", 0) / NIF_FUSE_GW=0 switches back (A/B, tests).
#include "k_fuse_dev.h"

#define ZERO_T6(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

// private ring of a wave: [matrix j][feature][16 points]
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

// PR = 0 (r5): fp32-exact products on HALF pairs -- planes (hi, lo) x operand (hi, lo), three v_mfma_f32_16x16x32_f16 per pair in both
// directions (k_pack16b mode 3, split2h; forward: half of r4's six bf16 products and two thirds of its chunk bytes; adjoint: 22
// significand bits where r4's bf16 pairs carried 16).  The planes carry a power of two s_jk, the sines 2^12, dL/da a power of two per
// point: all of it is scaled back exactly (biases pre-scaled in the LDS image, the combine factor zt s1 / s0, the sine's constants).
// PR = 1 / 2: the producers' hidden n x n products under a Keras policy -- mixed_bfloat16 (ONE bf16 product per operand pair) /
// mixed_float16 (half operands, per-point loss scale on dL/da; k_snet4_dev.h) -- from the policy's compact plane set.  The CONSUMER side
// is the same for all three: bf16 (hi, lo) deposits of the fp32 rows, three-product weight-gradient sums.
template <int NBL, int PR = 0>
__global__ __launch_bounds__(1024, 4) void k_snet6(S6Args F) {
  extern __shared__ __attribute__((aligned(256))) char smem6[];
  const SNetArgs& A = F.s;
  constexpr int NT = 512, WAVES = 8, r = 1;             // producer threads / waves (= tiles per round); 8 consumer waves behind them
  constexpr int NCH = NBL / 2;
  constexpr bool X16 = PR == 0;
  constexpr bool CP = PR != 0;                           // the policies' compact plane set (k_snet4_dev.h): one plane per block
  constexpr int CU = X16 ? NBL * 2 * 64 : NBL * 64;      // 16-byte units per chunk (forward and adjoint alike), = the LDS stride of a chunk
  constexpr int PB = X16 ? 3 : PR;                       // product form of mfma_x3
  constexpr int QP = (2 * CU) / NT;                      // DMA instructions per thread and chunk PAIR (exactly: 2 CU is a multiple of NT)
  static_assert((2 * CU) % NT == 0, "a chunk pair is a whole number of DMA instructions per producer thread");
  constexpr int NPL = 6;                                // planes per tile: h (hi, lo), zt h (hi, lo), dL/da (hi, lo)
  constexpr int EXT = NPL * FUSE_PLANE_BYTES;
  // per-tile weight vectors [hi 16 | lo 16] bf16 = 64 B.  Last layer (WVL): du_o (o < 3), zt, ones.  First layer (WVF), per plane k:
  // k * 4 + c = (zt | 1) x_c, k * 4 + 3 = (zt | 1)
  constexpr int NVL = 5, NVF = 8, WVLT = NVL * 64, WVFT = NVF * 64;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + WAVES - 1) / WAVES;
  const long nrounds = ngroups > (long)blockIdx.x ? (ngroups - 1 - (long)blockIdx.x) / gridDim.x + 1 : 0;   // tile rounds of this workgroup

  char* EX = smem6;                                     // [tile 8][plane 6][2 KB]
  char* WVL = EX + WAVES * EXT;
  char* WVF = WVL + WAVES * WVLT;
  bf16x8* chunks = reinterpret_cast<bf16x8*>(WVF + WAVES * WVFT);      // two pair buffers of two chunks
  float* sm = reinterpret_cast<float*>(chunks + 4 * CU);
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  // per-tile input rows [column][16 points], two sets: coordinates (padded to 4 columns), (latent, sample weight, -, -) and the targets
  const int CX = (si + 3) & ~3, CY = (so + 3) & ~3;
  const int NI = (CX + 4 + CY) * 16;
  const int pw = 2 * r * 64 + 2 * NI;                   // per-wave LDS floats (producers)
  float* lsum = sm + sm_tot + (long)WAVES * pw;
  float* scl = lsum + 16;                               // X16: [matrix][plane][s | 1 / s] of the half planes
  constexpr int NP = 16 * NBL;
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  {   // prologue, all 16 waves: LDS image of the small hyper-vectors; the exchange images start as zeros (the first tile round
      // consumes deposits that nobody made)
    const long s_wl = (long)si * n + (long)nh * n * n;
    const long s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
    for (int idx = tid; idx < (r + 1) * nsm; idx += 1024) {
      const int k = idx / nsm, e = idx - k * nsm;
      float v = 0.f;
      if (e < o_wl) { const int dd = e / NP, f = e - dd * NP; if (f < n) v = A.omega * hyp3(A, k, (long)dd * n + f); }
      else if (e < o_b1) { const int o = (e - o_wl) / NP, f = (e - o_wl) - o * NP; if (f < n) v = hyp3(A, k, s_wl + (long)f * so + o); }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = hyp3(A, k, s_b1 + f); }
      else if (e < o_bl) {
        const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP;
        if (f < n) v = hyp3(A, k, s_bh + (long)j * n + f);
        if (X16) v *= 4096.0f * A.wscale[(j * (r + 1) + k) * 2];      // the hidden biases start the scaled MFMA chains
      }
      else if (e < o_bl + so) v = hyp3(A, k, s_bl + (e - o_bl));
      sm[idx] = v;
    }
    if (X16 && tid < nh * (r + 1) * 2) scl[tid] = A.wscale[tid];
    for (int idx = tid; idx < (WAVES * (EXT + WVLT + WVFT)) / 16; idx += 1024) reinterpret_cast<f32x4*>(EX)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // The barrier intervals of a round (8 nh of them), numbered by group 0's items: forward layer j = items 4j .. 4j + 3 (M plane 0, V,
  // M plane 1, V), adjoint layer j = items 4 nh + 4 (nh - 1 - j) .. + 3; group 1 runs item x in interval x + 1.  Deposits (group 0 /
  // group 1 write during the interval): last layer 4 nh - 1 / 4 nh; hidden matrix j >= 1 at D_j = 4 nh + 4 (nh - 1 - j) + 3 / + 1;
  // matrix 0 at 8 nh - 1 / the next round's 0; the first layer's dL/da planes at the next round's 1 / 2.  The consumers read tiles
  // 0 .. 3 (group 0) before 4 .. 7 (group 1), never a tile in the interval its producer writes it.
  if (wid >= WAVES) {
    // =====================================================================================================================
    // consumer wave (plane kk, input block bI, output block bJ): the 32 x 32 block (kk, bI, bJ) of every hidden matrix; the
    // bI = 1 waves also the hidden biases (kk, bJ) (sums of the B operands they hold anyway), the bI = 0 waves columns 32 bJ .. of
    // the first layer, the bJ = 0 waves rows 32 bI .. of the last layer, wave (kk, 1, 1) the last layer's bias
    // =====================================================================================================================
    const int cw = wid - WAVES, kk = cw >> 2, bI = (cw >> 1) & 1, bJ = cw & 1;
    __syncthreads();
    if (tid - NT < WAVES * 16) {     // the constant "ones" vectors (hi = 1, lo = 0) of every tile
      const int t = (tid - NT) >> 4, q = (tid - NT) & 15;
      reinterpret_cast<__bf16*>(WVL + t * WVLT)[4 * 32 + q] = (__bf16)1.0f;
      reinterpret_cast<__bf16*>(WVF + t * WVFT)[7 * 32 + q] = (__bf16)1.0f;
    }
    __builtin_amdgcn_s_setprio(NIF_S6_CONS_PRIO);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float bacc[4] = {0.f, 0.f, 0.f, 0.f};                         // hidden biases (kk, bJ) -- bI = 1 waves
    float facc[3] = {0.f, 0.f, 0.f}, fbacc = 0.f;                  // first layer (kk, columns 32 bJ ..) -- bI = 0 waves
    float lacc[3] = {0.f, 0.f, 0.f}, blacc[3] = {0.f, 0.f, 0.f};   // last layer (kk, rows 32 bI ..) -- bJ = 0 waves; its bias -- wave (kk, 1, 1)
    FuseRd rdA = fuse_rd_addr(lane), rdB = rdA;
    rdA.a0 += (2 - 2 * kk) * FUSE_PLANE_BYTES + 256 * bI; rdA.a1 += (2 - 2 * kk) * FUSE_PLANE_BYTES + 256 * bI;   // plane 0: zt h, plane 1 (= r): h
    rdB.a0 += 4 * FUSE_PLANE_BYTES + 256 * bJ; rdB.a1 += 4 * FUSE_PLANE_BYTES + 256 * bJ;                         // dL/da
    const int wofs = 16 * (lane >> 5);                  // this lane's 8 points inside a weight vector (bytes)

#define S6_CBAR()                                                             \
  {                                                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);        /* lgkmcnt(0): the transpose reads are back */ \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
  }
    // hidden matrix J_: this wave's block over the deposited tiles [T0_, T1_).  One tile's operands ahead of the MFMAs (the
    // transpose reads of tile t + 1 are in flight while tile t multiplies), never more: 64 accumulator + 2 x 16 operand registers
#define S6_HID_LOAD(T_, AH_, AL_, BH_, BL_)                                                                 \
  {                                                                                                         \
    const char* img_ = EX + (T_) * EXT;                                                                     \
    AH_ = fuse_read_op(img_, rdA, 0); AL_ = fuse_read_op(img_ + FUSE_PLANE_BYTES, rdA, 0);                  \
    BH_ = fuse_read_op(img_, rdB, 0); BL_ = fuse_read_op(img_ + FUSE_PLANE_BYTES, rdB, 0);                  \
  }
#define S6_HID_TILES(J_, T0_, T1_)                                                                          \
  {                                                                                                         \
    bf16x8 ah_, al_, bh_, bl_, ah2_, al2_, bh2_, bl2_;                                                      \
    S6_HID_LOAD(T0_, ah_, al_, bh_, bl_)                                                                    \
    _Pragma("unroll") for (int t_ = T0_; t_ < T1_; ++t_) {                                                  \
      if (t_ + 1 < T1_) S6_HID_LOAD(t_ + 1, ah2_, al2_, bh2_, bl2_)                                         \
      acc[J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_, acc[J_], 0, 0, 0);                        \
      acc[J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_, acc[J_], 0, 0, 0);                        \
      acc[J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_, acc[J_], 0, 0, 0);                        \
      if (bI == 1) {                                                                                        \
        const char* w_ = WVL + t_ * WVLT + (3 + kk) * 64 + wofs;                                            \
        bacc[J_] = fuse_dot8(bh_, bl_, *reinterpret_cast<const bf16x8*>(w_), *reinterpret_cast<const bf16x8*>(w_ + 32), bacc[J_]); \
      }                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      ah_ = ah2_; al_ = al2_; bh_ = bh2_; bl_ = bl2_;                                                       \
    }                                                                                                       \
  }
    // the deposit of hidden matrix DJ_ (DJ_ >= 1; a compile-time index: the accumulators are never selected at run time -- a switch
    // over them made hipcc copy and spill whole accumulators around every call) in the four intervals behind D_j, two tiles each
#define S6_HID_STEP(DJ_, M_)                                                                                \
  if (DJ_ < nh && iv == 4 * nh + 4 * (nh - 1 - DJ_) + 4 + (M_)) { S6_DO(S6_HID_TILES(DJ_, 2 * (M_), 2 * (M_) + 2)) }
#define S6_HID_LAYER(DJ_) S6_HID_STEP(DJ_, 0) S6_HID_STEP(DJ_, 1) S6_HID_STEP(DJ_, 2) S6_HID_STEP(DJ_, 3)
    // last layer (h_nh, zt h_nh deposited as the A planes, du_o as vectors).  The skinny sums run as ROLLED loops over the tiles:
    // unrolled, hipcc fetched the weight vectors of all tiles first and spilled the accumulators to make room
    auto consume_last = [&](int t0, int t1) __attribute__((always_inline)) {
      if (bJ == 0) {
#pragma clang loop unroll(disable)
        for (int t = t0; t < t1; ++t) {
          const char* img = EX + t * EXT;
          const bf16x8 ah = fuse_read_op(img, rdA, 0), al = fuse_read_op(img + FUSE_PLANE_BYTES, rdA, 0);
          const char* w = WVL + t * WVLT + wofs;
          lacc[0] = fuse_dot8(ah, al, *reinterpret_cast<const bf16x8*>(w), *reinterpret_cast<const bf16x8*>(w + 32), lacc[0]);
          if (so > 1) lacc[1] = fuse_dot8(ah, al, *reinterpret_cast<const bf16x8*>(w + 64), *reinterpret_cast<const bf16x8*>(w + 96), lacc[1]);
          if (so > 2) lacc[2] = fuse_dot8(ah, al, *reinterpret_cast<const bf16x8*>(w + 128), *reinterpret_cast<const bf16x8*>(w + 160), lacc[2]);
        }
      } else if (bI == 1) {
#pragma clang loop unroll(disable)
        for (int t = t0; t < t1; ++t) {
          const char* w = WVL + t * WVLT + wofs;
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
          const bf16x8 bh = fuse_read_op(img, rdB, 0), bl = fuse_read_op(img + FUSE_PLANE_BYTES, rdB, 0);
          const char* w = WVF + t * WVFT + kk * 256 + wofs;
          fbacc = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w + 192), *reinterpret_cast<const bf16x8*>(w + 224), fbacc);
          facc[0] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w), *reinterpret_cast<const bf16x8*>(w + 32), facc[0]);
          if (si > 1) facc[1] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w + 64), *reinterpret_cast<const bf16x8*>(w + 96), facc[1]);
          if (si > 2) facc[2] = fuse_dot8(bh, bl, *reinterpret_cast<const bf16x8*>(w + 128), *reinterpret_cast<const bf16x8*>(w + 160), facc[2]);
        }
      }
    };
#ifdef NIF_S6_NOCONS
#define S6_DO(...)
#else
#define S6_DO(...) __VA_ARGS__
#endif
    // the barrier sequence of the producers' tile program, with this wave's share of the products between the barriers
    for (long rd = 0; rd < nrounds; ++rd) {
      for (int iv = 0; iv < 8 * nh; ++iv) {
        // the previous round's deposit of matrix 0: group 0's tiles, then group 1's
        if (iv == 0) { S6_DO(S6_HID_TILES(0, 0, 4)) }
        if (iv == 1) { S6_DO(S6_HID_TILES(0, 4, 8)) }
        // the previous round's first-layer deposit (written in intervals 1 / 2)
        if (iv >= 2 && iv < 6) { S6_DO(consume_first(2 * (iv - 2), 2 * (iv - 2) + 2);) }
        // the last layer's deposit (4 nh - 1 / 4 nh)
        if (iv >= 4 * nh && iv < 4 * nh + 4) { S6_DO(consume_last(2 * (iv - 4 * nh), 2 * (iv - 4 * nh) + 2);) }
        S6_HID_LAYER(3) S6_HID_LAYER(2) S6_HID_LAYER(1)
        S6_CBAR()
      }
    }
    if (nrounds > 0) S6_CBAR()    // (group 1 is one interval behind: its last item)
    __syncthreads();          // every deposit of the last round is visible
    S6_DO(S6_HID_TILES(0, 0, 8))
    __syncthreads();          // ... consumed: the producers write the last round's first-layer deposit
    __syncthreads();
    S6_DO(consume_first(0, 8);)
#undef S6_DO
#undef S6_HID_LAYER
#undef S6_HID_STEP
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
        for (int e = 0; e < 16; ++e) {
          const int in = 32 * bI + fmap(e, hf), out = 32 * bJ + i;
          if (in < n && out < n) prow[gidx(ws + (long)in * n + out)] = om * acc[j][e];
          if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (hipcc would form all 64 addresses first: 128 registers next to the accumulators)
        }
        float v = bacc[j];
        v += __shfl_xor(v, 32);
        if (bI == 1 && hf == 0 && 32 * bJ + i < n) prow[gidx(slot_bh(A, j) + 32 * bJ + i)] = v;
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = facc[c];
      v += __shfl_xor(v, 32);
      if (c < si && bI == 0 && hf == 0 && 32 * bJ + i < n) prow[gidx((long)c * n + 32 * bJ + i)] = om * v;
    }
    {
      float v = fbacc;
      v += __shfl_xor(v, 32);
      if (bI == 0 && hf == 0 && 32 * bJ + i < n) prow[gidx(slot_b1(A) + 32 * bJ + i)] = v;
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = lacc[o], w = blacc[o];
      v += __shfl_xor(v, 32);
      w += __shfl_xor(w, 32);
      if (o < so && bJ == 0 && hf == 0 && 32 * bI + i < n) prow[gidx(slot_wl(A) + (long)(32 * bI + i) * so + o)] = v;
      if (o < so && bJ == 1 && bI == 1 && lane == 0) prow[gidx(slot_bl(A) + o)] = w;
    }
    __syncthreads();          // (the producers' loss reduction)
    return;
  }

  // =======================================================================================================================
  // producer wave = one 16-point tile per round: k_snet4's tile program as alternating M / V items + the deposits
  // =======================================================================================================================
  const int grp = wid >> 2;                             // waves w and w + 4 share a SIMD: one of each group
  float* dzs = sm + sm_tot + (long)wid * pw;
  float* sks = dzs + r * 64;
  float* inp = sks + r * 64;
  // ---- the chunk stream (k_snet4), in pairs: forward planes of all hidden matrices, then the adjoint planes of matrix nh-1 .. 0 --
  const int NPP = r + 1;                                // pairs of one hidden matrix (a plane = its two K-step chunks)
  const bf16x8* cs_src = reinterpret_cast<const bf16x8*>(A.WF4);
  int cs_left = nh * NPP, cs_phase = 0, cs_pb = 0;
  long cs_groups = nrounds - 1;
  auto cs_phase_step = [&]() {
    ++cs_phase;
    if (cs_phase < 1 + nh) {
      cs_src = reinterpret_cast<const bf16x8*>(A.WB4) + (long)(nh - 1 - (cs_phase - 1)) * NPP * 2 * CU; cs_left = NPP; return;
    }
    if (cs_groups <= 0) { cs_left = -1; return; }
    --cs_groups; cs_phase = 0;
    cs_src = reinterpret_cast<const bf16x8*>(A.WF4); cs_left = nh * NPP;
  };
  auto cs_next_pair = [&]() {      // this wave's slice of the next pair into the pair buffer that the pair before last has left
    if (cs_left < 0) return;
    bf16x8* dst = chunks + cs_pb * 2 * CU;
#pragma unroll
    for (int q = 0; q < QP; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cs_src + tid + NT * q),
                                       (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);
    asm volatile("" ::: "memory");
    cs_src += 2 * CU;
    cs_pb ^= 1;
    if (--cs_left == 0) cs_phase_step();
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
    {   // column 0: the latent, column 1: the sample weight (or a target, unused), columns 2, 3: the latent again
      const float* src = (g == 1) ? (A.sw ? A.sw + ptn : A.y + ptn * so) : A.Z + (tile32n * r) * 32 + poffn;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + CX * 16), 4, 0, 0);
    }
    for (int i0 = 0; i0 < CY; i0 += 4) {
      const int c = i0 + g < so ? i0 + g : so - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.y + ptn * so + c),
                                       (__attribute__((address_space(3))) void*)(dst + (CX + 4 + i0) * 16), 4, 0, 0);
    }
  };
  prefetch_inputs(blockIdx.x, 0);
  if (cs_left <= 0 || nrounds <= 0) cs_left = -1;
  cs_next_pair();                                        // pair 0
  __syncthreads();
  int mq = 0;                                            // M items done = the pair this wave multiplies next
  float loss_lane = 0.f;
  float* ring = A.stash + ((long)blockIdx.x * WAVES + wid) * (long)nh * (NP * 16);    // [matrix][NP features][16 points]
  const FuseDep dep = fuse_dep_addr(p, g);
  char* exw = EX + wid * EXT;                            // this wave's tile images

  // s_waitcnt vmcnt(N) lgkmcnt(0); vmcnt counts in issue order, so "at most QP outstanding" behind a DMA slice that was issued LAST
  // means: everything older has landed, the slice may stay in flight for one more interval
#define S6_WAIT(N_) __builtin_amdgcn_s_waitcnt(0x0070 | ((N_) & 15) | (((N_) >> 4) << 14))
#define S6_SYNC()                                                             \
  {                                                                           \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
  }
  // The next pair's DMA goes out in the interval in which group 0 multiplies the current one: group 0 issues its slice at the start
  // of its M item (no other vector-memory traffic in an M item: vmcnt(QP) at its end leaves exactly the slice in flight, vmcnt(0)
  // at the end of the V item behind it waits for it); group 1 at the END of the V item in front of its M item (behind that item's
  // ring / input traffic: vmcnt(QP) again), vmcnt(0) at the end of the M item.
  // end of a V item
#define S6_V_END()                                                            \
  {                                                                           \
    if (grp == 1) { cs_next_pair(); S6_WAIT(QP); } else { S6_WAIT(0); }       \
    S6_SYNC()                                                                 \
  }
  // an M item: the two K-step chunks of pair mq
#define S6_M_ITEM(KS0_, KS1_)                                                 \
  {                                                                           \
    if (grp == 0) cs_next_pair();                                             \
    { const bf16x8* cur = chunks + ((mq & 1) * 2) * CU; KS0_ }               \
    asm volatile("" ::: "memory");    /* (the second chunk's operand reads stay behind the first chunk's products: registers) */ \
    { const bf16x8* cur = chunks + ((mq & 1) * 2 + 1) * CU; KS1_ }           \
    ++mq;                                                                     \
    if (grp == 0) { S6_WAIT(QP); } else { S6_WAIT(0); }                       \
    S6_SYNC()                                                                 \
  }
#define S6_BWD(KS_, T_, ZI_) { mfma_x3<NBL, PB, ZI_, NBL, 0, CP>(cur, q0[KS_], q1[KS_], T_, lane); }
#define S6_FWD(KS_, T_) { if (X16) mfma_x3<NBL, 3, false, NBL, 0, false>(cur, b0[KS_], b1[KS_], T_, lane); else mfma_x6<NBL, PR, false, NBL, 0, CP>(cur, b0[KS_], b1[KS_], b2[KS_], T_, lane); }

  // ---- state of the current round's tile --------------------------------------------------------------------------------
  bool active = false, valid = false;
  long tile32 = 0; int poff = 0;
  const float *xs = inp, *ys = inp, *wsp = inp, *zt_base = inp;
  auto setup_round = [&](long tg, int iset) {
    const long t16_raw = tg * WAVES + wid;
    active = t16_raw < nt16;
    const long t16 = active ? t16_raw : nt16 - 1;
    tile32 = t16 >> 1;
    poff = 16 * (int)(t16 & 1) + p;
    const long pt = t16 * 16 + p;
    valid = active && pt < A.B;
    const float* zs = inp + (iset & 1) * NI + CX * 16;
    xs = inp + (iset & 1) * NI + p;
    zt_base = zs + p;
    wsp = zs + 16 + p;
    ys = zs + 4 * 16 + p;
  };
  f32x4 h[NBL], acc[NBL], T[NBL], gh[NBL];
  bf16x8 b0[NCH], b1[NCH], b2[NCH];                     // forward operand splits of h
  bf16x8 e0[NCH], e1[NCH];                              // the first layer's dL/da (hi, lo) of the previous round, until its deposit
#pragma unroll
  for (int ks = 0; ks < NCH; ++ks)
#pragma unroll
    for (int t = 0; t < 8; ++t) { e0[ks][t] = (__bf16)0.f; e1[ks][t] = (__bf16)0.f; b2[ks][t] = (__bf16)0.f; }
  auto first_layer = [&](f32x4 (&out)[NBL]) __attribute__((always_inline)) {
    f32x4 a_[NBL];
    {
      const float* s0 = sm + r * nsm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
        for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        a_[b] = s;
      }
    }
    {
      const float zt = zt_base[0];
      const float* s0 = sm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
        for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        a_[b] += zt * s;
      }
    }
    sine16_tag<NBL>(a_, out);
  };
  // what the V item in front of forward layer j's first product leaves behind: the operand splits of h, the accumulators started
  // with the (scaled) biases, h_j on its way to the ring
  auto prep_fwd = [&](int j) __attribute__((always_inline)) {
    if (!NIF_S6_RECOMP0 || j > 0) ring_store16<NBL>(ring + j * (NP * 16), h, g, p);
    if (X16) split2h<NBL>(h, 4096.0f, b0, b1);
    else split3p<NBL, PR>(h, b0, b1, b2);
    const float* sb = sm + r * nsm + o_bh + j * NP + 4 * g;
    const float* sc = sm + o_bh + j * NP + 4 * g;
#pragma unroll
    for (int b = 0; b < NBL; ++b) { acc[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b); T[b] = *reinterpret_cast<const f32x4*>(sc + 16 * b); }
  };

  // ---- item -1 (V): the first round's first layer.  Group 1 spends interval 0 on it (group 0 multiplies pair 0 meanwhile) ------
  int iset = 0;
  setup_round(blockIdx.x, 0);
  if (nrounds > 0) {
    dzs[lane] = 0.f;
    first_layer(h);
    prefetch_inputs((long)blockIdx.x + gridDim.x, 1);
    prep_fwd(0);
    if (grp == 1) S6_V_END()
  }
  for (long rd = 0; rd < nrounds; ++rd, ++iset) {
    const long tg = (long)blockIdx.x + rd * gridDim.x;
    float zt0 = 0.f;
    bf16x8 d0[NCH], d1[NCH];                // the deposit's (hi, lo) pair of dL/da
    bf16x8 q0[NCH], q1[NCH];                // the adjoint products' operand
    float ils = 1.0f;
    // what the V item in front of adjoint layer j's first product does: dL/da_{j+1} = cos(a_{j+1}) dL/dh_{j+1} (h holds the tagged
    // sine h_{j+1}), its share of dL/dz, its operand forms; then h_j comes back (ring, or recomputed)
    auto prep_bwd = [&](int j) __attribute__((always_inline)) {
      f32x4 ga[NBL];
      {
        f32x4 dnext[NBL];
        tag_cos<NBL>(h, dnext);
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
      }
      if (NIF_S6_RECOMP0 && j == 0) first_layer(h);
      else ring_load16<NBL>(ring + j * (NP * 16), h, g, p);
      {
        const float* sb = sm + o_bh + j * NP + 4 * g;
        float sbv = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
          sbv += (ga[b][0] * bb[0] + ga[b][1] * bb[1]) + (ga[b][2] * bb[2] + ga[b][3] * bb[3]);
        }
        dzs[lane] += X16 ? sbv * (scl[j * 4 + 1] * (1.0f / 4096.0f)) : sbv;     // (the LDS image holds 4096 s0 b^(0))
      }
      split2<NBL>(ga, d0, d1);
      ils = 1.0f;
      if (PR == 2 || X16) {      // half operands: s dL/da with s a power of two per point (the point's largest |dL/da| into [2^14, 2^15))
        float mx = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ga[b][0]), fabsf(ga[b][1]))), fmaxf(fabsf(ga[b][2]), fabsf(ga[b][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const unsigned ef = (__float_as_uint(mx) >> 23) & 0xFFu;
        const unsigned sf = 268u - ef < 227u ? 268u - ef : 227u;
        ils = __uint_as_float((254u - sf) << 23);
        if (X16) split2h<NBL>(ga, __uint_as_float(sf << 23), q0, q1);
        else cast_f16<NBL>(ga, q0, __uint_as_float(sf << 23));
      } else {
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) q0[ks] = d0[ks];
      }
      if (!X16) {
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) q1[ks] = d1[ks];
      }
    };

    // the first three items of forward layer j: M (plane 0), V (the latent combine: plane 0's chain carries s0, the sum s1; in layer 0
    // also the previous round's first-layer deposit), M (plane 1)
    auto fwd_head = [&](int j, auto dep_e) __attribute__((always_inline)) {
      const float s1_ = X16 ? scl[j * 4 + 2] : 1.0f, is0_ = X16 ? scl[j * 4 + 1] : 1.0f;
      S6_M_ITEM(S6_FWD(0, T), S6_FWD(1, T))
      {
        const float zt = X16 ? zt_base[0] * (s1_ * is0_) : zt_base[0];
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * T[b];
        if (decltype(dep_e)::value) {      // (layer 0 only -- a compile-time flag: e0 / e1 are dead behind it)
          fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, e0);
          fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, e1);
        }
        S6_V_END()
      }
      S6_M_ITEM(S6_FWD(0, acc), S6_FWD(1, acc))
      if (X16) sine16_tag_sc<NBL>(acc, h, scl[j * 4 + 3] * (1.0f / 4096.0f));
      else sine16_tag<NBL>(acc, h);
    };
    // the first three items of adjoint layer j: M (plane 0), V (<h_j, M^(0) dL/da> for dL/dz; the second plane's chain starts from zt
    // times the first one's result), M (plane 1); then dL/dh_j is scaled back and deposit j written: (h_j ; zt h_j ; dL/da_{j+1})
    auto bwd_head = [&](int j) __attribute__((always_inline)) {
      const float s1_ = X16 ? scl[j * 4 + 2] : 1.0f, is0_ = X16 ? scl[j * 4 + 1] : 1.0f, is1_ = X16 ? scl[j * 4 + 3] : 1.0f;
      S6_M_ITEM(S6_BWD(0, T, true), S6_BWD(1, T, false))
      {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) s = fmaf(h[b][v], T[b][v], s);
        const float ztc = X16 ? zt0 * (s1_ * is0_) : zt0;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] = ztc * T[b];
        dzs[lane] += X16 ? (ils * is0_) * s : (PR == 2 ? ils * s : s);
        S6_V_END()
      }
      S6_M_ITEM(S6_BWD(0, gh, false), S6_BWD(1, gh, false))
      if (PR == 2 || X16) {
        const float f_ = X16 ? ils * is1_ : ils;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] *= f_;
      }
      fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, d0);
      fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, d1);
      {
        bf16x8 a0[NCH], a1[NCH];
        split2<NBL>(h, a0, a1);
        fuse_deposit4(exw, dep, a0);
        fuse_deposit4(exw + FUSE_PLANE_BYTES, dep, a1);
        f32x4 zh[NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) zh[b] = zt0 * h[b];
        split2<NBL>(zh, a0, a1);
        fuse_deposit4(exw + 2 * FUSE_PLANE_BYTES, dep, a0);
        fuse_deposit4(exw + 3 * FUSE_PLANE_BYTES, dep, a1);
      }
    };

    // ---- hidden hyper-matrices, forward: the fourth item of a layer is the activation + the next layer's operands ... ------------
    if (nh > 1) {
      fwd_head(0, std::true_type());
      prep_fwd(1);
      S6_V_END()
      for (int j = 1; j + 1 < nh; ++j) {
        fwd_head(j, std::false_type());
        prep_fwd(j + 1);
        S6_V_END()
      }
    }
    {   // ... or, behind the last matrix, the last layer (n -> so, linear), the loss and the start of the adjoint
      if (nh > 1) fwd_head(nh - 1, std::false_type());
      else fwd_head(0, std::true_type());
      ZERO_T6(gh)
      const float wsamp = (valid ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
      zt0 = zt_base[0];
      float se = 0.f;
      for (int o = 0; o < so; ++o) {
        f32x4 wg[NBL];
        ZERO_T6(wg)
        float part = 0.f, bias = 0.f;
#pragma unroll
        for (int k = 0; k <= r; ++k) {
          const float zt = k < r ? zt0 : 1.0f;
          const float* s0 = sm + k * nsm;
          float sk = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
            sk += (h[b][0] * w[0] + h[b][1] * w[1]) + (h[b][2] * w[2] + h[b][3] * w[3]);
            wg[b] += zt * w;
          }
          part = fmaf(zt, sk, part);
          bias = fmaf(zt, s0[o_bl + o], bias);
          if (k < r) sks[k * 64 + lane] = sk;
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float uo = part + bias;
        const float e = uo - ys[o * 16];
        NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
        const float du = dfac * wsamp * A.inv_bg / (float)so;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] += du * wg[b];
        {
          float t = du * sks[lane];
          if (g == 0) t = fmaf(du, sm[o_bl + o], t);
          dzs[lane] += t;
        }
        if (g == 0 && o < 3) {     // du_o of the tile's 16 points as a bf16 (hi | lo) row: the last layer's weight-gradient vector
          __bf16* wv = reinterpret_cast<__bf16*>(WVL + wid * WVLT);
          const __bf16 v0 = (__bf16)du;
          wv[o * 32 + p] = v0; wv[o * 32 + 16 + p] = (__bf16)(du - (float)v0);
        }
      }
      if (g == 1) {     // zt of the tile (the hidden layers' plane-0 bias sums use it too)
        __bf16* wv = reinterpret_cast<__bf16*>(WVL + wid * WVLT);
        const __bf16 z0 = (__bf16)zt0;
        wv[3 * 32 + p] = z0; wv[3 * 32 + 16 + p] = (__bf16)(zt0 - (float)z0);
      }
      {   // deposit "nh": the last layer's input h_nh (and zt h_nh) as the A planes
        bf16x8 a0[NCH], a1[NCH];
        split2<NBL>(h, a0, a1);
        fuse_deposit4(exw, dep, a0);
        fuse_deposit4(exw + FUSE_PLANE_BYTES, dep, a1);
        f32x4 zh[NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) zh[b] = zt0 * h[b];
        split2<NBL>(zh, a0, a1);
        fuse_deposit4(exw + 2 * FUSE_PLANE_BYTES, dep, a0);
        fuse_deposit4(exw + 3 * FUSE_PLANE_BYTES, dep, a1);
      }
      if (g == 0) loss_lane += wsamp * se / (float)so * A.inv_bg;
      prep_bwd(nh - 1);
      S6_V_END()
    }
    // ---- adjoint through the hidden hyper-matrices: the fourth item ends with the layer below ... ---------------------------------
    for (int j = nh - 1; j > 0; --j) {
      bwd_head(j);
      prep_bwd(j - 1);
      S6_V_END()
    }
    {   // ... or with the first layer: dL/da_0, the tile's dL/dz, the weight-gradient vectors (its dL/da planes wait in e0 / e1 until
        // the consumers have taken deposit 0: the next round's first V item, or the tail), and the next round's first layer
      bwd_head(0);
      f32x4 ga[NBL];
      {
        f32x4 dnext[NBL];
        tag_cos<NBL>(h, dnext);
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
      }
      {
        const float* s0 = sm + 4 * g;
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 t = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          for (int dd = 0; dd < si; ++dd) t += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          s += (ga[b][0] * t[0] + ga[b][1] * t[1]) + (ga[b][2] * t[2] + ga[b][3] * t[3]);
        }
        float tot = dzs[lane] + s;
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        if (active && g == 0) A.DZ[(tile32 * r) * 32 + poff] = tot;
      }
      split2<NBL>(ga, e0, e1);
      {      // lane group g < si: x_g and zt x_g of the tile's 16 points as bf16 (hi | lo) rows; group 3: zt.  (The consumers read
             // the previous round's vectors in intervals 2 .. 5 of a round: long done)
        __bf16* wv = reinterpret_cast<__bf16*>(WVF + wid * WVFT);
        const float x = g < si ? xs[g * 16] : 1.0f;
        const float zx = zt0 * x;
        const __bf16 x0 = (__bf16)x, z0 = (__bf16)zx;
        if (g < si || g == 3) { wv[g * 32 + p] = z0; wv[g * 32 + 16 + p] = (__bf16)(zx - (float)z0); }
        if (g < si && g < 3) { wv[(4 + g) * 32 + p] = x0; wv[(4 + g) * 32 + 16 + p] = (__bf16)(x - (float)x0); }
      }
      if (rd + 1 < nrounds) {      // the next round's first layer (its inputs arrived during this round)
        setup_round(tg + gridDim.x, iset + 1);
        dzs[lane] = 0.f;
        first_layer(h);
        prefetch_inputs(tg + 2 * (long)gridDim.x, iset & 1);
        prep_fwd(0);
      }
      S6_V_END()
    }
  }
  if (nrounds > 0 && grp == 0) { S6_WAIT(0); S6_SYNC() }      // (group 1's last item)
#undef S6_FWD
#undef S6_BWD
#undef S6_M_ITEM
#undef S6_V_END
#undef S6_SYNC
#undef S6_WAIT
  __syncthreads();          // every deposit of the last round is visible ...
  __syncthreads();          // ... and deposit 0 consumed
  fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, e0);
  fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, e1);
  __syncthreads();          // the last round's first-layer deposit is visible
  for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
  if (lane == 0) lsum[wid] = loss_lane;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < WAVES; ++w) s += lsum[w];
    A.loss_partial[blockIdx.x] = s;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static size_t snet6_shmem(const SNetArgs& a, int NBL) {
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + 4 + ((a.so + 3) & ~3)) * 16;
  const size_t pw = 2 * a.r * 64 + 2 * ni;
  return 8 * (6 * FUSE_PLANE_BYTES + (5 + 8) * 64) + 4 * (size_t)NBL * (a.prec == 0 ? 2 : 1) * 64 * 16 + (sm_tot + 8 * pw + 16 + 16) * sizeof(float);
}
// the fused-gradient kernel takes this training step (plain NIFMultiScale, fp32 results)
bool snet6_supported(const SNetArgs& a) {
  if (a.ll || a.res || a.nif_skip) return false;
  if (a.prec != 0) {     // the policy forms: NIF_S6_POLICY=0 keeps the r3 policy step (k_snet4<PR> + bf16 dL/da stash + k_gw_lds<DAB>) for A/B
    static const bool pol = [] { const char* e = getenv("NIF_S6_POLICY"); return !(e && e[0] == '0'); }();
    if (!pol) return false;
  }
  if (snet3_nbl(a.n) != 4 || a.r != 1 || a.nh < 1 || a.nh > 4 || a.si > 3 || a.so > 3) return false;
  if (a.prec == 0 && (!a.WF4x || !a.WB4x || !a.wscale)) return false;
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
    hipLaunchKernelGGL((k_snet6<4, PR_>), dim3(nblk), dim3(1024), shm, st, f);                                    \
  }
  if (a.prec == 2) { f.s.WF4 = a.WF4h; f.s.WB4 = a.WB4h; S6L(2) }      // the policy's compact plane set (k_pack16b mode 2 / 1)
  else if (a.prec == 1) { f.s.WF4 = a.WF4h; f.s.WB4 = a.WB4h; S6L(1) }
  else { f.s.WF4 = a.WF4x; f.s.WB4 = a.WB4x; S6L(0) }      // the exact-product half planes (k_pack16b mode 3)
#undef S6L
  return nblk;
}
