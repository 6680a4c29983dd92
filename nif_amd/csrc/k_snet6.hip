// k_snet6.hip -- the plain-SIREN training kernel (k_snet4<NBL, TRAIN, SINE, 0, SGN>) with EVERY ShapeNet weight gradient fused in:
// no dL/da stash, no weight-gradient launches (k_gw_first_lds, 4 x k_gw_lds, k_gw_out_lds), one partial-gradient row per workgroup.
//
// Why: k_snet4 writes 2.5 KB/point of h / dL/da rows that exist only so that the K = batch reductions
//     dL/dM_j^(k)[in][out] = w0 sum_p zt_k(p) h_j[in][p] dL/da_{j+1}[out][p]
// can run as separate HBM-bound kernels (0.54 ms of the 1.76 ms step, next to a 1.02 ms kernel whose arithmetic side is 0.71 ms).
// Here they are accumulated where both operands are live:
//   * ONE workgroup of 8 waves per CU (2 per SIMD, 256 registers), wave = one 16-point tile as in k_snet4;
//   * the accumulators of all hidden matrices and planes (nh (r+1) n^2 = 128 KB at 4 x 64, r = 1) live in the 8 waves' registers:
//     wave (k, I, J) = (wid >> 2, (wid >> 1) & 1, wid & 1) owns the 32 x 32 block (plane k, input block I, output block J) of EVERY
//     hidden matrix: 16 accumulator registers per matrix;
//   * at the end of adjoint layer j every wave DEPOSITS its tile's operands in LDS as bf16 (hi, lo) planes in the form it holds
//     MFMA B operands anyway (dL/da: the split of the data adjoint's own product; h_j and zt h_j: split when the row comes back from
//     the stash -- nothing is held across the layer for it): 24 ds_write_b64 per layer and tile, no packing arithmetic beyond the
//     splits; during the chunk steps of layer j-1 every wave runs
//     ITS block over the 8 deposited tiles: ds_read_b64_tr_b16 hands the operands over with features on lanes (k_fuse_dev.h) --
//     8 transpose reads + 3 v_mfma_f32_32x32x16_bf16 (hi.lo + lo.hi + hi.hi, K = the tile's 16 points) per tile, no VALU work;
//     the chunk barriers that exist anyway order deposit and consumption (two extra barriers per tile round around the first layer);
//   * biases, the first layer (K = si) and the last layer (N = so) are v_dot2_f32_bf16 sums of the same transposed operands against
//     per-tile weight vectors (zt, 1, x_c, zt x_c, du_o as bf16 hi | lo rows of 16 points): the lane already holds 8 points of its
//     feature -- 12 VALU instructions per tile, vector and wave role;
//   * what is left of the stash: the layer inputs h_0 .. h_{nh-1} of the wave's own tile (forward -> adjoint, re-read by the same wave).
// Built for: NIFMultiScale without resblocks, fp32 results, 49..64 units (NBL = 4), latent_dim 1, 1..4 hidden matrices, si, so <= 3.
// Everything else keeps k_snet4 + k_gw_*.  nif_set_option("fuse_gw", 0) / NIF_FUSE_GW=0 switches back (A/B, tests).
#include "k_fuse_dev.h"

#define ZERO_T6(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

#ifndef NIF_S6_RING
#define NIF_S6_RING 1       // 1: the h_j rows of a wave's tile in a private ring [matrix j][feature][16 points] that stays cache resident; 0: the [tile32][feature][32] stash
#endif
// private ring of a wave, point-major: the 4 features of (block b, lane group g) of point p are ONE 16-byte piece at
// p * NP + 16 b + 4 g -- 4 store / load instructions per layer instead of 16 (NIF_S6_RING_V4 = 0: feature-major rows f * 16 + p)
#ifndef NIF_S6_RING_NT
#define NIF_S6_RING_NT 0      // bit 0: non-temporal ring stores, bit 1: non-temporal ring loads (measured: see DESIGN)
#endif
#ifndef NIF_S6_RING_V4
#define NIF_S6_RING_V4 0
#endif
#ifndef NIF_S6_VMRING
#define NIF_S6_VMRING 0       // 1: ring stores / loads behind the chunk DMA, counted out of the chunk wait (S6_CHUNK_RING) -- measured r4: 1.327 / 1.331 vs 1.332 / 1.330 ms per step, no gain: nothing waits for the ring
#endif
template <int NBL>
__device__ __forceinline__ void ring_store16(float* __restrict__ slot, const f32x4 (&h)[NBL], int g, int p) {
#ifdef NIF_ABL_NOSTORE
  if (h[0][0] != 12345.678f) return;
#endif
#if NIF_S6_RING_V4
  f32x4* q = reinterpret_cast<f32x4*>(slot + p * (16 * NBL) + 4 * g);
#pragma unroll
  for (int b = 0; b < NBL; ++b) q[4 * b] = h[b];
#else
  float* q = slot + 4 * g * 16 + p;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#if NIF_S6_RING_NT & 1
      __builtin_nontemporal_store(h[b][v], q + (16 * b + v) * 16);
#else
      q[(16 * b + v) * 16] = h[b][v];
#endif
    }
#endif
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
#if NIF_S6_RING_V4
  const f32x4* q = reinterpret_cast<const f32x4*>(slot + p * (16 * NBL) + 4 * g);
#pragma unroll
  for (int b = 0; b < NBL; ++b) h[b] = q[4 * b];
#else
  const float* q = slot + 4 * g * 16 + p;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#if NIF_S6_RING_NT & 2
      h[b][v] = __builtin_nontemporal_load(q + (16 * b + v) * 16);
#else
      h[b][v] = q[(16 * b + v) * 16];
#endif
    }
#endif
}

#ifdef NIF_TIMELINE      // measurement builds: s_memtime stamps of producer wave 0 and consumer wave 8 of block 0, third round -- kept in LDS
                         // (a stamp in global memory is a vector-memory instruction of its own: it changes what the vmcnt waits wait for) and
                         // copied out at the end: entries 0 .. of A.tl = wave 0, 1024 .. = wave 8
#define S6_TL(id) do { if (blockIdx.x == 0 && (tid == 0 || tid == 512) && tlr == 2 && tlc < 380) { tlb[tlc] = ((long long)(id) << 48) | ((long long)__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFFll); ++tlc; } } while (0)
#define S6_TL_FLUSH() do { if (A.tl && blockIdx.x == 0 && (tid == 0 || tid == 512)) for (int q_ = 0; q_ < tlc; ++q_) { A.tl[(tid ? 2048 : 0) + 2 * q_] = tlb[q_] >> 48; A.tl[(tid ? 2048 : 0) + 2 * q_ + 1] = tlb[q_] & 0xFFFFFFFFFFFFll; } } while (0)
#else
#define S6_TL(id) do { } while (0)
#define S6_TL_FLUSH() do { } while (0)
#endif
struct S6Args {
  SNetArgs s;
  float* partial; long pstride;     // partial-gradient rows [gridDim.x][pstride] (the ShapeNet = hypernetwork columns of them)
};

#ifndef NIF_S6_RECOMP0
#define NIF_S6_RECOMP0 1     // the first layer's output (input of hidden matrix 0) is recomputed in the adjoint from the tile's inputs
                             // (si FMAs + a sine per element) instead of going through the ring: a quarter of the ring traffic less
#endif
#ifndef NIF_S6_EARLYDEP
#define NIF_S6_EARLYDEP 0    // 1 (r5, measured and NOT kept): deposit j is written INSIDE the last chunk step of adjoint layer j (its operands
                             // are complete after the first one) so that it is visible at that step's barrier and the consumer waves run 5
                             // of its 8 tiles during the producers' long vector interval of the next layer -- where they idle -- and 3 in
                             // the chunk step behind it (default: 3 + 3 + 2 in three chunk steps); one barrier per round less.  Parity
                             // green, 1.268 vs 1.254 ms per step (three same-box pairs): the s_memtime timeline
                             // (profiles/r05_timeline_earlydep.txt) shows the deposit's ~150 vector instructions costing 1.1 k ticks in
                             // the chunk step and saving 0.25-0.4 k in the vector block -- both producer waves of a SIMD run them at the
                             // same moment wherever they stand, and behind the step's matrix instructions they do not overlap its LDS wait
#endif
#ifndef NIF_S6_PF
#define NIF_S6_PF 0
#endif
#ifndef NIF_S6_BIGCHUNK
#define NIF_S6_BIGCHUNK 1    // 1 (r6, exact-product form; brings the NIF_S6_DBAR schedule with it; 0: the r5 form, 2: + the pipelined plane step, measured slower): chunks of 16 KB = a whole plane of a hidden matrix (both K steps), TWO chunk
                             // steps per layer and direction instead of four: half the per-step fixed costs (DMA issue 170-380 ticks, s_waitcnt
                             // 140, barrier >= 140, loop glue 120 of a ~1 250-tick forward step: r6 timeline)
#endif
#ifndef NIF_S6_EARLYDMA
#define NIF_S6_EARLYDMA 1    // 1 (r6, with NIF_S6_BIGCHUNK; the product form: 1.154-1.166 vs 1.169-1.185 ms on same-box triples): the chunk DMA of an adjoint layer's second step is issued at the TOP of the
                             // layer's vector block instead of inside its first chunk step (an LDS-DMA piece costs 25-60 cycles to issue in a
                             // VALU-only stretch, 100-185 inside a phase with matrix and LDS traffic: MI355X_MICROARCH.md)
#endif
#ifndef NIF_S6_DBAR_T0
#define NIF_S6_DBAR_T0 5     // tiles of the deposit taken under the vector block + the first chunk step (timeline r6: 240 ticks per tile there,
                             // 650 once the producers' matrix instructions compete)
#endif
#ifndef NIF_S6_DBAR
#define NIF_S6_DBAR 0        // 1 (r6): a barrier right BEHIND every hidden deposit (not only deposit 0's): the consumer waves take deposit j
                             // during the producers' vector block of layer j - 1 (the matrix pipe idles there) and its first chunk step
                             // (5 + 3 tiles) instead of in chunk steps 1-3 (3 + 3 + 2), where they were the last at every barrier
#endif
#ifndef NIF_S6_CONS_PRIO
#define NIF_S6_CONS_PRIO 0     // s_setprio of the consumer waves
#endif

// PR (late r4): the producers' hidden n x n products under a Keras policy -- 1 = mixed_bfloat16 (ONE bf16 product per operand pair),
// 2 = mixed_float16 (half operands, per-point loss scale on dL/da; k_snet4_dev.h) -- as in k_snet4<.., PR>.  The CONSUMER side is
// untouched: the deposits stay bf16 (hi, lo) pairs of the fp32 rows and the weight-gradient sums three products, i.e. the policy's
// weight gradients here are those of fp32 stash rows (the tests emulate it with stash_bf16 = False).  (r4's first policy form also cut the consumers
// to one MFMA per tile and hipcc answered with 327 spilled registers; with the consumer code unchanged the allocation holds.)
template <int NBL, int PR = 0>
__global__ __launch_bounds__(1024, 4) void k_snet6(S6Args F) {
  extern __shared__ __attribute__((aligned(256))) char smem6[];
  const SNetArgs& A = F.s;
  constexpr int NT = 512, WAVES = 8, r = 1;             // producer threads / waves (= tiles per round); 8 consumer waves behind them
  constexpr int NCH = NBL / 2;
  // PR = 0 (r5): fp32-exact products on HALF pairs -- planes (hi, lo) x operand (hi, lo), three v_mfma_f32_16x16x32_f16 per pair in both
  // directions (k_pack16b mode 3, split2h; forward: half of r4's six bf16 products and two thirds of its chunk bytes; adjoint: 22
  // significand bits where r4's bf16 pairs carried 16).  The planes carry a power of two s_jk, the sines 2^12, dL/da a power of two per
  // point: all of it is scaled back exactly (biases pre-scaled in the LDS image, the combine factor zt s1 / s0, the sine's constants)
  constexpr bool X16 = PR == 0;
  constexpr int CF = X16 ? NBL * 2 * 64 : NBL * 3 * 64, CB = NBL * 2 * 64;   // 16-byte units per forward / adjoint chunk
  constexpr bool CP = PR != 0;                           // the policies' compact plane set (k_snet4_dev.h): one plane per block
  constexpr int CFH = CP ? NBL * 64 : CF, CBH = CP ? NBL * 64 : CB;
  constexpr bool BIG = X16 && NIF_S6_BIGCHUNK;
  static_assert(!NIF_S6_BIGCHUNK || (!NIF_S6_PF && !NIF_S6_EARLYDEP), "NIF_S6_BIGCHUNK: without NIF_S6_PF / NIF_S6_EARLYDEP");
  constexpr bool DBAR = BIG || (NIF_S6_DBAR != 0);      // a barrier behind every hidden deposit: always with the big chunks (the consumers' only
                                                        // other window would be ONE chunk step per layer), optional on the 8 KB forms (the policies)
  constexpr int CFB = BIG ? 2 * CF : CF;                 // units of one chunk BUFFER
  constexpr int QF = (CFB + NT - 1) / NT;
  // (r5: three buffers with the DMA two chunk steps ahead measured no gain -- 1.185 vs 1.15-1.19 ms -- although the s_memtime timeline
  // shows ~300 ticks of every step in front of the barrier's s_waitcnt: tools/exp/k_snet6_3buf.hip, profiles/r05_timeline_*.txt)
  // r6 (NIF_S6_PF = 1, exact-product form): THREE buffers, the DMA two steps ahead, so that a step can read the first operands of the next
  // chunk behind its own first products (mfma_x3_pf)
  constexpr bool PF = X16 && NIF_S6_PF;
  constexpr int NBUF = PF ? 3 : 2;
  constexpr int NPL = 6;                                // planes per tile: h (hi, lo), zt h (hi, lo), dL/da (hi, lo)
#if defined(NIF_ABL_NOSTORE) || defined(NIF_ABL_NOLOAD) || !NIF_S6_RING
  constexpr int NRING = 0;
#else
  constexpr int NRING = NIF_S6_RING_V4 ? NBL : 4 * NBL; // vector-memory instructions of one ring_store16 / ring_load16
#endif
  constexpr int EXT = NPL * FUSE_PLANE_BYTES;
  // per-tile weight vectors [hi 16 | lo 16] bf16 = 64 B.  Last layer (WVL): du_o (o < 3), zt, ones.  First layer (WVF), per plane k:
  // k * 4 + c = (zt | 1) x_c, k * 4 + 3 = (zt | 1)
  constexpr int NVL = 5, NVF = 8, WVLT = NVL * 64, WVFT = NVF * 64;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, nh = A.nh, si = A.si, so = A.so;
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  char* EX = smem6;                                     // [tile 8][plane 6][2 KB]
  char* WVL = EX + WAVES * EXT;
  char* WVF = WVL + WAVES * WVLT;
  bf16x8* chunks = reinterpret_cast<bf16x8*>(WVF + WAVES * WVFT);
  float* sm = reinterpret_cast<float*>(chunks + NBUF * CFB);
  constexpr int NP = 16 * NBL;
  // r5: the LDS image of the small hyper-vectors has a FIXED layout -- three first-layer rows, three last-layer rows, the first bias, four
  // hidden biases, the last bias (the shape's unused rows are zeros): every offset into it is a compile-time constant that folds into the
  // ds_read's immediate.  With offsets made of si / so / nh the tile program spent ~300 v_add_u32 per tile on LDS addresses (a tenth of
  // the producers' vector instructions, all of them inside the vector blocks)
  constexpr int o_w1 = 0, o_wl = 3 * NP, o_b1 = 6 * NP, o_bh = 7 * NP, o_bl = 11 * NP, nsm = 11 * NP + 4;
  constexpr int sm_tot = ((r + 1) * nsm + 3) & ~3;
  // r6: the input set of a tile is EIGHT rows of 16 points -- x_0..x_2 | z | y_0..y_2 | sample weight (si, so <= 3, r = 1: snet6_supported) --
  // fetched by TWO LDS-DMA instructions whose lane groups point at different arrays (r5: sixteen rows, four instructions, most of them
  // duplicates): 8 KB of LDS back per workgroup -- what the 16 KB chunk buffers of NIF_S6_BIGCHUNK need
  constexpr int CX = 3, CZ = 1, CY = 3;
  constexpr int NI = (CX + CZ + CY + 1) * 16;
  constexpr int pw = 2 * r * 64 + 2 * NI;               // per-wave LDS floats (producers)
  float* lsum = sm + sm_tot + (long)WAVES * pw;
  float* scl = lsum + 16;                               // X16: [matrix][plane][s | 1 / s] of the half planes
  long long* tlb = reinterpret_cast<long long*>(scl + 16) + (tid ? 380 : 0); (void)tlb;      // (NIF_TIMELINE builds: 2 x 380 stamps)

  {   // prologue, all 12 waves: LDS image of the small hyper-vectors; the exchange images start as zeros (the first tile round
      // consumes a first-layer deposit that nobody made)
    const long s_wl = (long)si * n + (long)nh * n * n;
    const long s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
    for (int idx = tid; idx < (r + 1) * nsm; idx += 1024) {
      const int k = idx / nsm, e = idx - k * nsm;
      float v = 0.f;
      if (e < o_wl) { const int dd = e / NP, f = e - dd * NP; if (f < n && dd < si) v = A.omega * hyp3(A, k, (long)dd * n + f); }
      else if (e < o_b1) { const int o = (e - o_wl) / NP, f = (e - o_wl) - o * NP; if (f < n && o < so) v = hyp3(A, k, s_wl + (long)f * so + o); }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = hyp3(A, k, s_b1 + f); }
      else if (e < o_bl) {
        const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP;
        if (f < n && j < nh) {
          v = hyp3(A, k, s_bh + (long)j * n + f);
          if (X16) v *= 4096.0f * A.wscale[(j * (r + 1) + k) * 2];    // the hidden biases start the scaled MFMA chains (rows j >= nh stay zero: wscale holds nh matrices only)
        }
      }
      else if (e < o_bl + so) v = hyp3(A, k, s_bl + (e - o_bl));
      sm[idx] = v;
    }
    if (X16 && tid < nh * (r + 1) * 2) scl[tid] = A.wscale[tid];
    for (int idx = tid; idx < (WAVES * (EXT + WVLT + WVFT)) / 16; idx += 1024) reinterpret_cast<f32x4*>(EX)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- the chunk stream (k_snet4): forward planes of all hidden matrices, then the adjoint planes of matrix nh-1 .. 0 -----------
  const int NPC = (r + 1) * NCH;
  const bf16x8* cs_src = reinterpret_cast<const bf16x8*>(A.WF4);
  const int NPCS = BIG ? (r + 1) : NPC;                  // chunks per matrix as the stream hands them out (BIG: one per plane)
  int cs_units = BIG ? 2 * CFH : CFH, cs_left = nh * NPCS, cs_phase = 0;
  long cs_groups = (ngroups - 1 - (long)blockIdx.x) / gridDim.x;
  auto cs_phase_step = [&]() {
    ++cs_phase;
    if (cs_phase < 1 + nh) {
      cs_src = reinterpret_cast<const bf16x8*>(A.WB4) + (long)(nh - 1 - (cs_phase - 1)) * NPC * CBH; cs_units = BIG ? 2 * CBH : CBH; cs_left = NPCS; return;
    }
    if (cs_groups <= 0) { cs_left = -1; return; }
    --cs_groups; cs_phase = 0;
    cs_src = reinterpret_cast<const bf16x8*>(A.WF4); cs_units = BIG ? 2 * CFH : CFH; cs_left = nh * NPCS;
  };
  // r5: BOTH roles walk the stream; the chunk of step c + 1 is issued during step c by the CONSUMER waves in the forward steps (they
  // idle there: the s_memtime timeline shows ~250 ticks of every producer step going into the DMA issue) and by the producers in
  // the adjoint steps (where the consumers carry the weight-gradient products)
  const int ctid = tid & (NT - 1), cwid = wid & (WAVES - 1);
  auto cs_next = [&](int buf, bool issue) {
    if (cs_left < 0) return;
    if (issue) {
      bf16x8* dst = chunks + buf * CFB;
#pragma unroll
      for (int q = 0; q < QF; ++q)
        if (cwid * 64 + NT * q < cs_units)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cs_src + ctid + NT * q),
                                           (__attribute__((address_space(3))) void*)(dst + cwid * 64 + NT * q), 16, 0, 0);
    }
    asm volatile("" ::: "memory");
    cs_src += cs_units;
    if (--cs_left == 0) cs_phase_step();
  };
  if (wid >= WAVES) {
    // =====================================================================================================================
    // consumer wave (plane kk, input block bI, output block bJ): the 32 x 32 block (kk, bI, bJ) of every hidden matrix; the
    // bI = 1 waves also the hidden biases (kk, bJ) (sums of the B operands they hold anyway), the bI = 0 waves columns 32 bJ .. of
    // the first layer, the bJ = 0 waves rows 32 bI .. of the last layer, wave (kk, 1, 1) the last layer's bias
    // =====================================================================================================================
    const int cw = wid - WAVES, kk = cw >> 2, bI = (cw >> 1) & 1, bJ = cw & 1;
    int tlc = 0, tlr = 0; (void)tlc; (void)tlr;
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
    // the skinny sums by wave role, seven registers where the four roles' arrays would take fourteen (r5):
    //   xacc[0 .. 3]: bI = 1 waves -- hidden biases (kk, bJ) of matrix j;  bI = 0 waves -- first layer (kk, columns 32 bJ ..): x_c (c < 3) | its bias
    //   xacc[4 .. 6]: bJ = 0 waves -- last layer (kk, rows 32 bI ..), output o;  wave (kk, 1, 1) -- its bias
    float xacc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define bacc xacc
#define facc xacc
#define fbacc xacc[3]
#define lacc (xacc + 4)
#define blacc (xacc + 4)
    FuseRd rdA = fuse_rd_addr(lane), rdB = rdA;
    rdA.a0 += (2 - 2 * kk) * FUSE_PLANE_BYTES + 256 * bI; rdA.a1 += (2 - 2 * kk) * FUSE_PLANE_BYTES + 256 * bI;   // plane 0: zt h, plane 1 (= r): h
    rdB.a0 += 4 * FUSE_PLANE_BYTES + 256 * bJ; rdB.a1 += 4 * FUSE_PLANE_BYTES + 256 * bJ;                         // dL/da
    const int wofs = 16 * (lane >> 5);                  // this lane's 8 points inside a weight vector (bytes)

#define S6_CBAR()                                                             \
  {                                                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);        /* lgkmcnt(0): the transpose reads are back */ \
    S6_TL(400);                                                               \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    S6_TL(500);                                                               \
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
    // the four chunk steps of an adjoint layer with the consumption of hidden deposit DJ_ (a compile-time index: the accumulators
    // are never selected at run time -- a switch over them made hipcc copy and spill whole accumulators around every call)
#if NIF_S6_EARLYDEP
#define S6_HID_LAYER(DJ_)      /* entered behind the barrier of the step that deposited DJ_ */              \
  if (DJ_ < nh) {                                                                                           \
    S6_DO(S6_HID_TILES(DJ_, 0, 5))                                                                          \
    S6_CBAR()                                                                                               \
    S6_DO(S6_HID_TILES(DJ_, 5, 8))                                                                          \
    S6_CBAR()                                                                                               \
    S6_CBAR()                                                                                               \
    S6_CBAR()                                                                                               \
  }
#else
#define S6_HID_LAYER(DJ_)      /* DBAR: entered in front of the barrier behind deposit DJ_; else in front of the first chunk barrier of layer DJ_ - 1 */ \
  if (DJ_ < nh) {                                                                                           \
    if (DBAR) {                                                                                             \
      S6_CBAR()                                                                                             \
      S6_DO(S6_HID_TILES(DJ_, 0, NIF_S6_DBAR_T0))                                                           \
      S6_CBAR()                                                                                             \
      S6_DO(S6_HID_TILES(DJ_, NIF_S6_DBAR_T0, 8))                                                           \
      S6_CBAR()                                                                                             \
      if (!BIG) {                                                                                           \
        S6_CBAR()                                                                                           \
        S6_CBAR()                                                                                           \
      }                                                                                                     \
    } else {                                                                                                \
      S6_CBAR()                                                                                             \
      S6_DO(S6_HID_TILES(DJ_, 0, 3))                                                                        \
      S6_CBAR()                                                                                             \
      S6_DO(S6_HID_TILES(DJ_, 3, 6))                                                                        \
      S6_CBAR()                                                                                             \
      S6_DO(S6_HID_TILES(DJ_, 6, 8))                                                                        \
      S6_CBAR()                                                                                             \
    }                                                                                                       \
  }
#endif
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
    if (cs_left <= 0) cs_left = -1;
    cs_next(0, false);                      // (chunk 0: issued by the producers in their prologue)
    int nb_c = 1;
    if (PF) { cs_next(1, false); nb_c = 2; }     // (PF: chunk 1 too; a step issues the chunk two steps ahead)
#define S6_ROTC() { if (PF) nb_c = nb_c == 2 ? 0 : nb_c + 1; else nb_c ^= 1; }
    // a forward interval: this wave's slice of the NEXT step's chunk goes out first and has landed in front of the barrier
#define S6_CFWD(...)                                                          \
  {                                                                           \
    cs_next(nb_c, true); S6_ROTC()                                            \
    __VA_ARGS__                                                               \
    __builtin_amdgcn_s_waitcnt(0x0070);        /* vmcnt(0) lgkmcnt(0) */      \
    S6_TL(400);                                                               \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    S6_TL(500);                                                               \
  }
    for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++tlr) {
      if (BIG) {
        // two intervals per layer; the previous round's first-layer deposit (visible behind the first barrier) over the next three
        // (r6 timeline: at 3 + 3 + 2 tiles over three intervals the consumers were the LAST at those barriers by 0.5-1.1 k ticks -- the
        // skinny sums are rolled, latency-bound loops; spread over all 2 nh - 1 intervals behind the first they hide under the producers' steps)
        const int nq = 2 * nh - 1;
        for (int j = 0; j < nh; ++j) {
          S6_CFWD(if (j > 0) { const int q_ = 2 * j - 1; S6_DO(consume_first(q_ * 8 / nq, (q_ + 1) * 8 / nq);) })
          S6_CFWD({ const int q_ = 2 * j; S6_DO(consume_first(q_ * 8 / nq, (q_ + 1) * 8 / nq);) })
        }
        for (int q = 0; q < (r + 1) * nh; ++q) { cs_next(nb_c, false); S6_ROTC() }
        S6_CBAR()                              // behind the last layer's deposit
        S6_DO(consume_last(0, 5);)
        S6_CBAR()                              // first chunk step of adjoint layer nh - 1
        S6_DO(consume_last(5, 8);)
        S6_CBAR()                              // second
        S6_HID_LAYER(3) S6_HID_LAYER(2) S6_HID_LAYER(1)
        S6_CBAR()                              // deposit 0 next to the first layer's adjoint
        S6_DO(S6_HID_TILES(0, 0, 8))
        S6_CBAR()
        continue;
      }
      for (int j = 0; j < nh; ++j) {        // forward: the previous round's first-layer deposit next to hidden matrix 0
        S6_CFWD()
        S6_CFWD(if (j == 0) { S6_DO(consume_first(0, 3);) })      // (three intervals: at four tiles the consumers were the last at these barriers, r5 timeline)
        S6_CFWD(if (j == 0) { S6_DO(consume_first(3, 6);) })
        S6_CFWD(if (j == 0) { S6_DO(consume_first(6, 8);) })
      }
      for (int q = 0; q < 4 * nh; ++q) { cs_next(nb_c, false); S6_ROTC() }      // (the adjoint steps' chunks: the producers issue them)
      // adjoint: the last layer's deposit next to the steps of layer nh - 1, then deposit j + 1 next to layer j
#if NIF_S6_EARLYDEP
      S6_CBAR()
      S6_DO(consume_last(0, 4);)
      S6_CBAR()
      S6_DO(consume_last(4, 8);)
      S6_CBAR()                              // (the producers write deposit nh - 1 over the last layer's during this step)
      S6_CBAR()
      S6_HID_LAYER(3) S6_HID_LAYER(2) S6_HID_LAYER(1)
      S6_DO(S6_HID_TILES(0, 0, 8))           // deposit 0 (visible since the last chunk barrier) next to the first layer's adjoint
      S6_CBAR()
#else
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
#endif
    }
    __syncthreads();
    S6_DO(consume_first(0, 8);)
    __syncthreads();
#undef S6_CFWD
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
    S6_TL_FLUSH();
    __syncthreads();          // (the producers' loss reduction)
    return;
  }
#undef bacc
#undef facc
#undef fbacc
#undef lacc
#undef blacc

  // =======================================================================================================================
  // producer wave = one 16-point tile per round: k_snet4's tile program + the deposits
  // =======================================================================================================================
  float* dzs = sm + sm_tot + (long)wid * pw;
  float* sks = dzs + r * 64;
  float* inp = sks + r * 64;
  auto prefetch_inputs = [&](long tgn, int set) {
    long t16n = tgn * WAVES + wid;
    if (t16n >= nt16) t16n = nt16 - 1;
    const long tile32n = t16n >> 1;
    const int poffn = 16 * (int)(t16n & 1) + p;
    long ptn = t16n * 16 + p;
    if (ptn >= A.B) ptn = A.B - 1;
    float* dst = inp + set * NI;
    {     // rows 0..3: x_g (g < 3, clamped to the net's coordinates) | z
      const int c = g < si ? g : si - 1;
      const float* src = g < 3 ? A.xin + ptn * A.ncol + A.col0 + c : A.Z + (tile32n * r) * 32 + poffn;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
    }
    {     // rows 4..7: y_g (g < 3, clamped) | sample weight (or a target where there is none: never read then)
      const int c = g < so ? g : so - 1;
      const float* src = (g < 3 || !A.sw) ? A.y + ptn * so + c : A.sw + ptn;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + 4 * 16), 4, 0, 0);
    }
  };
  prefetch_inputs(blockIdx.x, 0);
  if (cs_left <= 0) cs_left = -1;
  cs_next(0, true);
  if (PF) cs_next(1, true);
  __syncthreads();
  bool dma_mine = false;                                 // forward steps: the consumer waves issue the chunk DMA
  bool dma_early = false;                                // NIF_S6_EARLYDMA: the next S6_CHUNK's DMA went out already
  int cbuf = 0, nbuf = PF ? 2 : 1;
  bf16x8 pf[4]; (void)pf;                                // PF: the next chunk's first block pair (live inside a layer's four steps only)
#define S6_ROT() { if (PF) { cbuf = cbuf == 2 ? 0 : cbuf + 1; nbuf = nbuf == 2 ? 0 : nbuf + 1; } else { cbuf ^= 1; nbuf ^= 1; } }
  int tlc = 0, tlr = 0; (void)tlc; (void)tlr;
  float loss_lane = 0.f;
  const long sstride = A.slot_stride, tstride = (long)stash_fp(n) * 32;
  float* IN0 = A.stash;
  float* ring = A.stash + ((long)blockIdx.x * WAVES + wid) * (long)nh * (NP * 16);    // NIF_S6_RING: [matrix][NP features][16 points]
  (void)ring; (void)IN0; (void)sstride; (void)tstride;
  const FuseDep dep = fuse_dep_addr(p, g);
  char* exw = EX + wid * EXT;                            // this wave's tile images

#define S6_CHUNK(...)                                                         \
  {                                                                           \
    S6_TL(100);                                                               \
    if (!dma_early) cs_next(nbuf, dma_mine);                                  \
    dma_early = false;                                                        \
    S6_TL(200);                                                               \
    const bf16x8* cur = chunks + cbuf * CFB;                                  \
    __VA_ARGS__                                                               \
    S6_TL(300);                                                               \
    __builtin_amdgcn_s_waitcnt(0x0070);        /* vmcnt(0) lgkmcnt(0): the chunk DMA has landed, the deposits are visible */ \
    S6_TL(400);                                                               \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    S6_TL(500);                                                               \
    S6_ROT()                                                                  \
  }
// r6 (NIF_S6_PRE): the exact-product steps read the first block pair's A operands IN FRONT of the next chunk's DMA issue -- the issue
// (60-180 cycles per piece, MI355X_MICROARCH.md) then overlaps the LDS latency of the reads the step's first MFMAs wait for
#ifndef NIF_S6_PRE
#define NIF_S6_PRE 0
#endif
#define S6_CHUNKP(ZI_, B0_, B1_, T_)                                          \
  {                                                                           \
    S6_TL(100);                                                               \
    const bf16x8* cur = chunks + cbuf * CFB;                                  \
    const bf16x8 pa_[4] = {cur[lane], cur[64 + lane], cur[128 + lane], cur[192 + lane]};   \
    __builtin_amdgcn_sched_barrier(0);                                        \
    cs_next(nbuf, dma_mine);                                                  \
    S6_TL(200);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                        \
    mfma_x3_pre<NBL, ZI_>(cur, pa_, B0_, B1_, T_, lane);                      \
    S6_TL(300);                                                               \
    __builtin_amdgcn_s_waitcnt(0x0070);                                       \
    S6_TL(400);                                                               \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    S6_TL(500);                                                               \
    S6_ROT()                                                                  \
  }
// BIG: one step = a whole plane (two K-step halves of the 16 KB chunk)
#define S6_CHUNK2(ZI_, Q0_, Q1_, T_)                                          \
  S6_CHUNK({ if (NIF_S6_BIGCHUNK == 2) mfma_x3_plane<NBL, ZI_, CF>(cur, Q0_, Q1_, T_, lane); else { mfma_x3<NBL, 3, ZI_, NBL, 0, false>(cur, Q0_[0], Q1_[0], T_, lane); mfma_x3<NBL, 3, false, NBL, 0, false>(cur + CF, Q0_[1], Q1_[1], T_, lane); } })
#define S6_CHUNKF(USE_, MAKE_, ZI_, B0_, B1_, T_)                             \
  {                                                                           \
    S6_TL(100);                                                               \
    cs_next(nbuf, dma_mine);                                                  \
    S6_TL(200);                                                               \
    const bf16x8* cur = chunks + cbuf * CFB;                                  \
    const bf16x8* nxt = chunks + (cbuf == 2 ? 0 : cbuf + 1) * CF;             \
    mfma_x3_pf<NBL, ZI_, USE_, MAKE_>(cur, nxt, pf, B0_, B1_, T_, lane);      \
    S6_TL(300);                                                               \
    __builtin_amdgcn_s_waitcnt(0x0070);                                       \
    S6_TL(400);                                                               \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    S6_TL(500);                                                               \
    S6_ROT()                                                                  \
  }
// the chunk step that carries the layer's ring traffic: the NRING ring instructions are issued BEHIND the next chunk's DMA, so the
// wait at the end of the step may leave exactly them in flight (vmcnt counts in issue order: "at most NRING outstanding" = every
// DMA instruction has landed) -- their latency gets the following chunk step as well instead of sitting in front of this barrier
#if NIF_S6_VMRING
#define S6_CHUNK_RING(PRE_, ...)                                              \
  {                                                                           \
    cs_next(nbuf, dma_mine);                                                  \
    PRE_                                                                      \
    asm volatile("" ::: "memory");                                            \
    const bf16x8* cur = chunks + cbuf * CFB;                                  \
    __VA_ARGS__                                                               \
    __builtin_amdgcn_s_waitcnt(0x0070 | (NRING & 15) | ((NRING >> 4) << 14));   /* vmcnt(NRING) lgkmcnt(0) */ \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    S6_ROT()                                                                  \
  }
#else
#define S6_CHUNK_RING(PRE_, ...) { PRE_ S6_CHUNK(__VA_ARGS__) }
#endif

  int iset = 0;
  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++iset, ++tlr) {
    S6_TL(1);
    const long t16_raw = tg * WAVES + wid;
    const bool active = t16_raw < nt16;
    const long t16 = active ? t16_raw : nt16 - 1;
    const long tile32 = t16 >> 1;
    const int poff = 16 * (int)(t16 & 1) + p;
    const long pt = t16 * 16 + p;
    const bool valid = active && pt < A.B;
    const float* xs = inp + (iset & 1) * NI + p;
    const float* zs = inp + (iset & 1) * NI + CX * 16;
    const float* ys = zs + CZ * 16 + p;
    const float* wsp = zs + (CZ + CY) * 16 + p;
    const float* zt_base = zs + p;
    const long row0 = tile32 * tstride + poff;
    (void)row0;
    dzs[lane] = 0.f;

    f32x4 h[NBL], acc[NBL];
    // ---- first layer ----------------------------------------------------------------------------------------------------
    auto first_layer = [&](f32x4 (&out)[NBL]) __attribute__((always_inline)) {
      f32x4 a_[NBL];
      {
        const float* s0 = sm + r * nsm + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          _Pragma("unroll") for (int dd = 0; dd < 3; ++dd) if (dd < si) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          a_[b] = s;
        }
      }
      {
        const float zt = zt_base[0];
        const float* s0 = sm + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          _Pragma("unroll") for (int dd = 0; dd < 3; ++dd) if (dd < si) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          a_[b] += zt * s;
        }
      }
      sine16_tag<NBL>(a_, out);
    };
    first_layer(h);
    prefetch_inputs(tg + gridDim.x, (iset + 1) & 1);
    // ---- hidden hyper-matrices, forward ---------------------------------------------------------------------------------------
    dma_mine = false;
    for (int j = 0; j < nh; ++j) {
#if !NIF_S6_RING
      if (active) st_store16<NBL>(IN0 + (long)j * sstride, row0, h, g);
#endif
      bf16x8 b0[NCH], b1[NCH], b2[NCH];
      if (X16) split2h<NBL>(h, 4096.0f, b0, b1);
      else split3p<NBL, PR>(h, b0, b1, b2);
      const float s1_ = X16 ? scl[j * 4 + 2] : 1.0f, is0_ = X16 ? scl[j * 4 + 1] : 1.0f, is1_ = X16 ? scl[j * 4 + 3] : 1.0f;
      {
        const float* sb = sm + r * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
      {
        f32x4 T[NBL];
        const float* sb = sm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) T[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
#if NIF_S6_RING
#define S6_FWD(KS_, T_) { if (X16) mfma_x3<NBL, 3, false, NBL, 0, false>(cur, b0[KS_], b1[KS_], T_, lane); else mfma_x6<NBL, PR, false, NBL, 0, CP>(cur, b0[KS_], b1[KS_], b2[KS_], T_, lane); }
        if (BIG) {
          if (!NIF_S6_RECOMP0 || j > 0) ring_store16<NBL>(ring + j * (NP * 16), h, g, p);
          S6_CHUNK2(false, b0, b1, T)
        } else if (PF) {
          if (!NIF_S6_RECOMP0 || j > 0) ring_store16<NBL>(ring + j * (NP * 16), h, g, p);
          S6_CHUNKF(false, true, false, b0[0], b1[0], T)
        } else S6_CHUNK_RING({ if (!NIF_S6_RECOMP0 || j > 0) ring_store16<NBL>(ring + j * (NP * 16), h, g, p); }, S6_FWD(0, T))
#else
        S6_CHUNK(S6_FWD(0, T))
#endif
        if (BIG) { }
        else if (PF) S6_CHUNKF(true, true, false, b0[1], b1[1], T)
        else S6_CHUNK(S6_FWD(1, T))
        const float zt = X16 ? zt_base[0] * (s1_ * is0_) : zt_base[0];      // (plane 0's chain carries s0, the sum s1)
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * T[b];
      }
      if (BIG) S6_CHUNK2(false, b0, b1, acc)
      else if (PF) S6_CHUNKF(true, true, false, b0[0], b1[0], acc)
      else S6_CHUNK(S6_FWD(0, acc))
      if (BIG) { }
      else if (PF) S6_CHUNKF(true, false, false, b0[1], b1[1], acc)
      else S6_CHUNK(S6_FWD(1, acc))
#undef S6_FWD
      if (X16) sine16_tag_sc<NBL>(acc, acc, is1_ * (1.0f / 4096.0f));
      else sine16_tag<NBL>(acc, acc);
#pragma unroll
      for (int b = 0; b < NBL; ++b) h[b] = acc[b];
    }
    // ---- last layer (n -> so, linear), MSE, start of the adjoint ---------------------------------------------------------------
    f32x4 gh[NBL];
    ZERO_T6(gh)
    const float wsamp = (valid ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
    const float zt0 = zt_base[0];
    float se = 0.f;
    for (int o = 0; o < so; ++o) {
      f32x4 wg[NBL];
      ZERO_T6(wg)
      float part = 0.f, bias = 0.f;
#pragma unroll
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt0 : 1.0f;
        const float* s0 = sm + k * nsm;
        float sk = 0.f, sk2 = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
          sk = fmaf(h[b][0], w[0], sk); sk2 = fmaf(h[b][1], w[1], sk2); sk = fmaf(h[b][2], w[2], sk); sk2 = fmaf(h[b][3], w[3], sk2);      // (FMA chains, r5: 16 instructions per sum instead of 16 mul + 16 add)
          wg[b] += zt * w;
        }
        sk += sk2;
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
        const __bf16 d0 = (__bf16)du;
        wv[o * 32 + p] = d0; wv[o * 32 + 16 + p] = (__bf16)(du - (float)d0);
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
    if (BIG) {      // the last layer's deposit is visible NOW: its eight tiles of skinny sums (3.1 k ticks of rolled, latency-bound loops) spread over
                    // this wave's vector block and the first chunk step instead of filling the one step behind the first barrier (r6 timeline: the
                    // producers waited 1.9 k ticks for them there)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
      S6_TL(600);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      S6_TL(700);
    }
    // ---- adjoint through the hidden hyper-matrices ---------------------------------------------------------------------------
    f32x4 dnext[NBL], hin[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) hin[b] = h[b];
    dma_mine = true;
    for (int j = nh - 1; j >= 0; --j) {
      f32x4 ga[NBL];
      if (BIG && NIF_S6_EARLYDMA) { cs_next(nbuf, true); dma_early = true; }
      tag_cos<NBL>(hin, dnext);
#if !NIF_S6_RING
      st_load16<NBL>(IN0 + (long)j * sstride, row0, hin, g);
#endif
#pragma unroll
      for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
      {
        const float* sb = sm + o_bh + j * NP + 4 * g;
        float sbv = 0.f, sbv2 = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
          sbv = fmaf(ga[b][0], bb[0], sbv); sbv2 = fmaf(ga[b][1], bb[1], sbv2); sbv = fmaf(ga[b][2], bb[2], sbv); sbv2 = fmaf(ga[b][3], bb[3], sbv2);
        }
        sbv += sbv2;
        dzs[lane] += X16 ? sbv * (scl[j * 4 + 1] * (1.0f / 4096.0f)) : sbv;     // (the LDS image holds 4096 s0 b^(0))
      }
      bf16x8 b0[NCH], b1[NCH];
      constexpr bool LATE_SPLIT = X16 && NIF_S6_EARLYDEP;   // the deposit's pair is formed where it is deposited (dL/da stays live instead of it: same registers)
      if (!LATE_SPLIT) split2<NBL>(ga, b0, b1);             // the deposit's (hi, lo) pair; b0 is also the bf16 policy's operand
      bf16x8 q0[NCH], q1[NCH];                // the products' operand: b0, or half(s dL/da), s per point (mixed_float16: hi alone; X16: (hi, lo))
      const float s1_ = X16 ? scl[j * 4 + 2] : 1.0f, is0_ = X16 ? scl[j * 4 + 1] : 1.0f, is1_ = X16 ? scl[j * 4 + 3] : 1.0f;
      float ils = 1.0f;
      if (PR == 2 || X16) {
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
        for (int ks = 0; ks < NCH; ++ks) q0[ks] = b0[ks];
      }
      if (!X16) {
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) q1[ks] = b1[ks];
      }
      constexpr int PB = X16 ? 3 : PR;
      {
        f32x4 U[NBL];
#if NIF_S6_RING     // h_j (dz dot product, this layer's A planes, the cosine of the layer below) -- dnext was taken from hin above
        if (BIG) {
          if (NIF_S6_RECOMP0 && j == 0) first_layer(hin); else ring_load16<NBL>(ring + j * (NP * 16), hin, g, p);
          S6_CHUNK2(true, q0, q1, U)
        } else if (PF) {
          if (NIF_S6_RECOMP0 && j == 0) first_layer(hin); else ring_load16<NBL>(ring + j * (NP * 16), hin, g, p);
          S6_CHUNKF(false, true, true, q0[0], q1[0], U)
        } else S6_CHUNK_RING({ if (NIF_S6_RECOMP0 && j == 0) first_layer(hin); else ring_load16<NBL>(ring + j * (NP * 16), hin, g, p); }, { mfma_x3<NBL, PB, true, NBL, 0, CP>(cur, q0[0], q1[0], U, lane); })
#else
        S6_CHUNK({ mfma_x3<NBL, PB, true, NBL, 0, CP>(cur, q0[0], q1[0], U, lane); })
#endif
        if (BIG) { }
        else if (PF) S6_CHUNKF(true, true, false, q0[1], q1[1], U)
        else if (X16 && NIF_S6_PRE) S6_CHUNKP(false, q0[1], q1[1], U)
        else S6_CHUNK({ mfma_x3<NBL, PB, false, NBL, 0, CP>(cur, q0[1], q1[1], U, lane); })
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) s = fmaf(hin[b][v], U[b][v], s);
        const float ztc = X16 ? zt0 * (s1_ * is0_) : zt0;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] = ztc * U[b];
        dzs[lane] += X16 ? (ils * is0_) * s : (PR == 2 ? ils * s : s);
      }
      // deposit j: (h_j ; zt h_j ; dL/da) of this tile -- the consumer waves take it during the steps of layer j - 1
#define S6_DEPOSIT()                                                          \
      {                                                                       \
        if (LATE_SPLIT) split2<NBL>(ga, b0, b1);                              \
        fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, b0);                   \
        fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, b1);                   \
        bf16x8 a0[NCH], a1[NCH];                                              \
        split2<NBL>(hin, a0, a1);                                             \
        fuse_deposit4(exw, dep, a0);                                          \
        fuse_deposit4(exw + FUSE_PLANE_BYTES, dep, a1);                       \
        f32x4 zh[NBL];                                                        \
        _Pragma("unroll") for (int b = 0; b < NBL; ++b) zh[b] = zt0 * hin[b]; \
        split2<NBL>(zh, a0, a1);                                              \
        fuse_deposit4(exw + 2 * FUSE_PLANE_BYTES, dep, a0);                   \
        fuse_deposit4(exw + 3 * FUSE_PLANE_BYTES, dep, a1);                   \
      }
      if (BIG) S6_CHUNK2(false, q0, q1, gh)
      else if (PF) S6_CHUNKF(true, true, false, q0[0], q1[0], gh)
      else if (X16 && NIF_S6_PRE) S6_CHUNKP(false, q0[0], q1[0], gh)
      else S6_CHUNK({ mfma_x3<NBL, PB, false, NBL, 0, CP>(cur, q0[0], q1[0], gh, lane); })
#if NIF_S6_EARLYDEP
      // (the slot's previous deposit was consumed two barriers ago; the splits run behind this step's matrix instructions)
      S6_CHUNK({ mfma_x3<NBL, PB, false, NBL, 0, CP>(cur, q0[1], q1[1], gh, lane); S6_DEPOSIT() })
#else
      if (BIG) { }
      else if (PF) S6_CHUNKF(true, false, false, q0[1], q1[1], gh)
      else if (X16 && NIF_S6_PRE) S6_CHUNKP(false, q0[1], q1[1], gh)
      else S6_CHUNK({ mfma_x3<NBL, PB, false, NBL, 0, CP>(cur, q0[1], q1[1], gh, lane); })
#endif
      if (PR == 2 || X16) {
        const float f_ = X16 ? ils * is1_ : ils;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] *= f_;
      }
#if !NIF_S6_EARLYDEP
      S6_DEPOSIT()
#endif
#undef S6_DEPOSIT
#if !NIF_S6_EARLYDEP
      if (DBAR && j > 0) {      // deposit j is visible NOW: the consumers start on it under this wave's next vector block
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        S6_TL(600);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        S6_TL(700);
      }
#endif
    }
#if !NIF_S6_EARLYDEP
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the deposits have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
    // ---- first layer (the consumer waves take deposit 0 meanwhile) ------------------------------------------------------------
    {
      f32x4 ga[NBL];
      tag_cos<NBL>(hin, dnext);
#pragma unroll
      for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
      {
        const float* s0 = sm + 4 * g;
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 t = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          _Pragma("unroll") for (int dd = 0; dd < 3; ++dd) if (dd < si) t += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          s = fmaf(ga[b][0], t[0], s); s2 = fmaf(ga[b][1], t[1], s2); s = fmaf(ga[b][2], t[2], s); s2 = fmaf(ga[b][3], t[3], s2);
        }
        float tot = dzs[lane] + (s + s2);
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        if (active && g == 0) A.DZ[(tile32 * r) * 32 + poff] = tot;
      }
      bf16x8 b0[NCH], b1[NCH];
      split2<NBL>(ga, b0, b1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();          // deposit 0 has been consumed
      asm volatile("" ::: "memory");
      fuse_deposit4(exw + 4 * FUSE_PLANE_BYTES, dep, b0);
      fuse_deposit4(exw + 5 * FUSE_PLANE_BYTES, dep, b1);
      {      // lane group g < si: x_g and zt x_g of the tile's 16 points as bf16 (hi | lo) rows; group 3: zt
        __bf16* wv = reinterpret_cast<__bf16*>(WVF + wid * WVFT);
        const float x = g < si ? xs[g * 16] : 1.0f;
        const float zx = zt0 * x;
        const __bf16 x0 = (__bf16)x, z0 = (__bf16)zx;
        if (g < si || g == 3) { wv[g * 32 + p] = z0; wv[g * 32 + 16 + p] = (__bf16)(zx - (float)z0); }
        if (g < si && g < 3) { wv[(4 + g) * 32 + p] = x0; wv[(4 + g) * 32 + 16 + p] = (__bf16)(x - (float)x0); }
      }
    }
  }
#undef S6_CHUNK
  __syncthreads();          // the last round's first-layer deposit is visible ...
  __syncthreads();          // ... and consumed
  S6_TL_FLUSH();
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
  const size_t sm_tot = (((size_t)(a.r + 1) * (11 * 16 * NBL + 4)) + 3) & ~(size_t)3;      // (the kernel's fixed small-vector layout)
  const size_t ni = 8 * 16;      // (the kernel's input set: x_0..x_2 | z | y_0..y_2 | sample weight, 16 points each)
  const size_t pw = 2 * a.r * 64 + 2 * ni;
  return 8 * (6 * FUSE_PLANE_BYTES + (5 + 8) * 64) + (size_t)((a.prec == 0 && NIF_S6_PF) ? 3 : 2) * ((a.prec == 0 && NIF_S6_BIGCHUNK) ? 2 : 1) * NBL * (a.prec == 0 ? 2 : 3) * 64 * 16 + (sm_tot + 8 * pw + 16 + 16) * sizeof(float)
#ifdef NIF_TIMELINE
         + 2 * 380 * 8
#endif
      ;
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
