// k_sobw_dev.h -- the Sobolev training step of the plain SIREN ShapeNet with one to three coordinate seeds (BASELINE configs[4]:
// u, du/dx, du/dy; reference nif/layers/gradient.py:36-49 + the two-output mse of the Keras model), the streams of a tile on
// separate WAVES.
//
// k_sob (k_sob_dev.h) carries the primal and all tangent streams of a 16-point tile in ONE wave: 370 registers with two seeds, one
// wave per SIMD, every dependent latency exposed -- 3.99 ms of arithmetic where three k_snet4 passes take 2.1.
// Here a workgroup is twelve waves = 12 / (1 + ns) tiles x (primal, tangent 0, ..); each wave is a k_snet4 wave (same chunk stream,
// same bf16-split MFMA forms, same stash layout -- stream q of tile32 t is pseudo-tile q * nt32 + t, as k_sob writes them), three
// per SIMD, and the streams meet in LDS twice per layer:
//   forward :  primal  -> c = cos(a)                      -> tangents: h' = c a'
//   adjoint :  primal  -> c ;  tangent d -> w_d = mu_d a'_d   -> primal: da = lambda c - sin(a) sum_d w_d ; tangents: nu_d = mu_d c
// (formulas: k_sob_dev.h).  The tangent pre-activations a'_d wait in a global ring (one tile per hidden layer and tangent
// wave; the first layer's is recomputed), the cosine of the adjoint is rebuilt from the tagged sine (k_snet4).
// PR: the mixed_bfloat16 policy -- one bf16 product per operand pair, the stream's tile rounded once per layer, the latent
// factor applied to the product (k_snet4<PR>'s cast points, k_sob<BF = 2>'s too); with SNetArgs.da_bf16 the hidden layers'
// dL/da stash rows are bf16 (k_gw_lds<.., DAB> reads them).
#pragma once
#include "k_sob_dev.h"

template <int NBL>
__device__ __forceinline__ void sine16_tagc(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&c)[NBL]) {
  if (sine16_big<NBL>(a)) {
    f32x4 s[NBL];
    sine16_slow<NBL>(a, s, c);
#pragma unroll
    for (int b = 0; b < NBL; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) h[b][v] = __uint_as_float((__float_as_uint(s[b][v]) & ~1u) | (__float_as_uint(c[b][v]) >> 31));
    return;
  }
  const f32x2 C = {0.15915493667125702f, 0.15915493667125702f}, CL = {6.420638326565253e-09f, 6.420638326565253e-09f};
  const f32x2 M = {12582912.0f, 12582912.0f}, IP = {0.318309886183790672f, 0.318309886183790672f};
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; v += 2) {
      const f32x2 x = {a[b][v], a[b][v + 1]};
      const f32x2 k = __builtin_elementwise_fma(x, C, M) - M;
      f32x2 f = __builtin_elementwise_fma(x, C, -k);
      f = __builtin_elementwise_fma(x, CL, f);
      const f32x2 t = __builtin_elementwise_fma(x, IP, M);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float sv = __builtin_amdgcn_sinf(f[e]);
        c[b][v + e] = __builtin_amdgcn_cosf(f[e]);
        unsigned o;
        asm("s_nop 0\n\tv_bfi_b32 %0, 1, %1, %2" : "=v"(o) : "v"(__float_as_uint(t[e])), "v"(__float_as_uint(sv)));
        h[b][v + e] = __uint_as_float(o);
      }
    }
}

#ifndef NIF_SOBW_OCC
#define NIF_SOBW_OCC 3      // waves per SIMD the register budget allows (hipcc: the second __launch_bounds__ argument is waves per EU)
#endif
// tiles per workgroup: 12 waves / (1 + seeds) streams = one 12-wave workgroup per CU (two 6-wave workgroups of 2 tiles with two
// seeds: 4.6 instead of 3.4 ms).  r4: nets of 65..128 units (six / eight 16-feature blocks) need k_snet4<8>'s 256 registers per
// wave: two waves per SIMD, 8 / (1 + seeds) tiles (two seeds: 2 tiles = 6 waves)
#define NIF_SOBW_WMAX(NBL_) ((NBL_) <= 4 ? 12 : 8)
#define NIF_SOBW_TPG(NBL_, NS_) (NIF_SOBW_WMAX(NBL_) / (1 + (NS_)))
#define ZERO_T(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

// TRAIN = false (r4): the two-output model's predict() -- primal and tangent streams forward only (no stash, no ring, no targets)
// MODE = 1 (r4): SIREN_ResNet blocks (siren.py:381-410) -- h_out = 0.5 (u + sin(a2)), a2 = w0 t W2 + b2, t = sin(a1), a1 = w0 u W1 + b1:
// every stream carries the skip on its own (tangent: h'_out = 0.5 (u' + cos(a2) a2')); the block input u comes back from the stream's
// stash row of matrix j - 1 (training) or waits in registers (predict); the block output carries the tag of cos(a2) and the adjoint
// rebuilds sin(a2) = 2 h - u (k_snet4's resblock form)
// MODE = 2 (r4): class NIF (model.py:233-324) -- h_out = h_in + f(a) with any Keras activation f: the primal wave publishes c = f'(a)
// like the cosine and keeps (c, -f''(a)) of every layer in a ring of its own (the SIREN forms rebuild both from the tagged sine);
// nu = mu c, da = lambda c - (-f'') sum_d mu_d a'_d hold unchanged (k_sob_dev.h sob_act)
template <int NBL, bool PR, int NS, bool TRAIN = true, int MODE = 0>
__global__ __launch_bounds__(64 * NIF_SOBW_WMAX(NBL), (NBL <= 4 ? NIF_SOBW_OCC : 2)) void k_sobw(SobArgs J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SNetArgs& A = J.s;
  constexpr int NQ = 1 + NS, TPG = NIF_SOBW_TPG(NBL, NS), WAVES = TPG * NQ, NT = 64 * WAVES;
  constexpr int NCH = NBL / 2;
  constexpr int CF = NBL * 3 * 64, CB = NBL * 2 * 64;
  constexpr bool CP = PR;                                // the policy's compact plane set (k_snet4_dev.h): one bf16 plane per block
  constexpr int CFH = CP ? NBL * 64 : CF, CBH = CP ? NBL * 64 : CB;
  constexpr int QF = (CF + NT - 1) / NT;
  constexpr int NBUF = 2;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wid / TPG, tl = wid % TPG;               // stream (0 = primal, 1 + d = tangent d) and tile of the group
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const int FP = stash_fp(n);
  const long nt32 = (A.B + 31) / 32;
  const long nt16 = 2 * nt32;
  const long ngroups = (nt16 + TPG - 1) / TPG;
  const int seed = q ? J.seed[q - 1] : 0, gcol = q ? J.gcol[q - 1] : 0;

  bf16x8* chunks = reinterpret_cast<bf16x8*>(smem);
  float* sm = smem + NBUF * CF * 4;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  const int CX = (si + 3) & ~3, CZ = (r + 3) & ~3, CY = (so + 3) & ~3;
  const int NI = (CX + CZ + CY + 4) * 16;
  const int pw = 2 * r * 64 + 2 * NI;
  float* dzs = sm + sm_tot + (long)wid * pw;
  float* sks = dzs + r * 64;
  float* inp = sks + r * 64;
  f32x4* xch = reinterpret_cast<f32x4*>(sm + sm_tot + (long)WAVES * pw) + tl * (NQ * NBL * 64);   // this tile's c | w_0 | w_1
  f32x4* xc = xch + lane;
  f32x4* xw = xch + (1 + (q ? q - 1 : 0)) * NBL * 64 + lane;
  float* dzx = sm + sm_tot + (long)WAVES * pw + TPG * NQ * NBL * 256;       // [WAVES][r][16]
  float* lsum = dzx + WAVES * r * 16;
  constexpr int NP = 16 * NBL;
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  // the chunk stream (k_snet4): forward planes, then the adjoint planes of hidden matrix nh-1 .. 0, then the next tile group
  const int NPC = (r + 1) * NCH;
  const bf16x8* cs_src = reinterpret_cast<const bf16x8*>(A.WF4);
  int cs_units = CFH, cs_left = nh * NPC, cs_phase = 0;
  long cs_groups = (ngroups - 1 - (long)blockIdx.x) / gridDim.x;
  auto cs_phase_step = [&]() {
    ++cs_phase;
    if (TRAIN && cs_phase < 1 + nh) {
      cs_src = reinterpret_cast<const bf16x8*>(A.WB4) + (long)(nh - cs_phase) * NPC * CBH; cs_units = CBH; cs_left = NPC;
    } else {
      if (cs_groups <= 0) { cs_left = -1; return; }
      --cs_groups; cs_phase = 0;
      cs_src = reinterpret_cast<const bf16x8*>(A.WF4); cs_units = CFH; cs_left = nh * NPC;
    }
  };
  auto cs_next = [&](int buf) {
    if (cs_left < 0) return;
    bf16x8* dst = chunks + buf * CF;
#pragma unroll
    for (int qq = 0; qq < QF; ++qq)
      if (wid * 64 + NT * qq < cs_units)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cs_src + tid + NT * qq),
                                         (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * qq), 16, 0, 0);
    asm volatile("" ::: "memory");
    cs_src += cs_units;
    if (--cs_left == 0) cs_phase_step();
  };
  // the tile's input rows by DMA, one tile ahead (k_snet4); a tangent wave takes its column of the target derivatives as "y"
  auto prefetch_inputs = [&](long tgn, int set) {
    long t16n = tgn * TPG + tl;
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
      const int c = i0 + g < r ? i0 + g : r - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + (tile32n * r + c) * 32 + poffn),
                                       (__attribute__((address_space(3))) void*)(dst + (CX + i0) * 16), 4, 0, 0);
    }
    if (TRAIN) {
      for (int i0 = 0; i0 < CY; i0 += 4) {
        const int c = i0 + g < so ? i0 + g : so - 1;
        const float* src = q ? J.gt + (ptn * so + c) * J.gstride + gcol : A.y + ptn * so + c;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + (CX + CZ + i0) * 16), 4, 0, 0);
      }
      const float* swp = A.sw ? A.sw + ptn : A.y + ptn * so;
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
      if (e < o_wl) { const int dd = e / NP, f = e - dd * NP; if (f < n) v = A.omega * hyp3(A, k, (long)dd * n + f); }
      else if (e < o_b1) { const int o = (e - o_wl) / NP, f = (e - o_wl) - o * NP; if (f < n) v = hyp3(A, k, s_wl + (long)f * so + o); }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = hyp3(A, k, s_b1 + f); }
      else if (e < o_bl) { const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP; if (f < n) v = hyp3(A, k, s_bh + (long)j * n + f); }
      else if (e < o_bl + so) v = hyp3(A, k, s_bl + (e - o_bl));
      sm[idx] = v;
    }
    if (cs_left <= 0) cs_left = -1;
    cs_next(0);
  }
  __syncthreads();
  int cbuf = 0, nbuf = 1;
  float loss_lane = 0.f;
  // a'_d of hidden layer j (tangent waves): ring tile j of this wave
#ifdef NIF_ABL_NORING      // measurement builds: every workgroup on the same ring tiles (cache resident; results are wrong)
  const long ring_wg = 0;
#else
  const long ring_wg = blockIdx.x;
#endif
  f32x4* ring = reinterpret_cast<f32x4*>(J.ring) + (ring_wg * (WAVES - TPG) + (wid >= TPG ? wid - TPG : 0)) * (long)nh * (NBL * 64) + lane;
  // MODE 2: (c, -f'') of layer l (0 = first) of this primal wave's tile: tiles ((l * 2 + which) * NBL + b) * 64, behind every
  // workgroup's tangent rings (the buffer holds 20 (nh + 1) tiles per workgroup: (WAVES - TPG) nh + 2 TPG (nh + 1) are used)
  f32x4* pring = reinterpret_cast<f32x4*>(J.ring) + ((long)gridDim.x * (WAVES - TPG) * nh + (ring_wg * TPG + tl) * 2L * (nh + 1)) * (NBL * 64) + lane;
  (void)pring;
  const long sstride = A.slot_stride, tstride = (long)FP * 32;
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * sstride;

// the streams of a tile meet: LDS traffic only (lgkmcnt(0)); stash loads / stores stay in flight across the barrier
#define SW_MEET()                                                             \
  {                                                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                       \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
  }
// one chunk step (k_snet4's NIF_CHUNK with two buffers): the DMA of the next chunk is drained in front of the barrier.  Starting
// it a whole step earlier and leaving the layer's stash stores in flight across the barrier (counted vmcnt waits) measured the
// same (3.28 / 3.30 ms), so the simple form stays
#define SW_CHUNK(...)                                                         \
  {                                                                           \
    cs_next(nbuf);                                                            \
    const bf16x8* cur = chunks + cbuf * CF;                                   \
    __VA_ARGS__                                                               \
    __builtin_amdgcn_s_waitcnt(0x0F70);                                       \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    cbuf ^= 1; nbuf ^= 1;                                                     \
  }

  int iset = 0;
  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++iset) {
    const long t16_raw = tg * TPG + tl;
    const bool active = t16_raw < nt16;
    const long t16 = active ? t16_raw : nt16 - 1;
    const long tile32 = t16 >> 1;
    const int poff = 16 * (int)(t16 & 1) + p;
    const long pt = t16 * 16 + p;
    const bool valid = active && pt < A.B;
    const float* xs = inp + (iset & 1) * NI + p;
    const float* zs = inp + (iset & 1) * NI + CX * 16;
    const float* ys = zs + CZ * 16 + p;                 // primal: y_o ; tangent d: the target of du_o/dx_d
    const float* wsp = zs + (CZ + CY) * 16 + p;
    const float* zt_base = zs + p;
    const long row0 = ((long)q * nt32 + tile32) * tstride + poff;      // stream q's pseudo-tile
    for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    f32x4 h[NBL], acc[NBL];
    // ---- first layer: a = sum_k zt_k (x . (w0 W1^(k)) + b1^(k)) ;  a'_d = sum_k zt_k (w0 W1^(k))[seed_d, :] -----------------
    auto first_tangent = [&](f32x4 (&t)[NBL]) {
      const float* s0 = sm + r * nsm + 4 * g + o_w1 + seed * NP;
#pragma unroll
      for (int b = 0; b < NBL; ++b) t[b] = *reinterpret_cast<const f32x4*>(s0 + 16 * b);
      for (int k = 0; k < r; ++k) {
        const float zt = zt_base[k * 16];
        const float* sk = sm + k * nsm + 4 * g + o_w1 + seed * NP;
#pragma unroll
        for (int b = 0; b < NBL; ++b) t[b] += zt * *reinterpret_cast<const f32x4*>(sk + 16 * b);
      }
    };
    if (q == 0) {
      {
        const float* s0 = sm + r * nsm + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          acc[b] = s;
        }
      }
      for (int k = 0; k < r; ++k) {
        const float zt = zt_base[k * 16];
        const float* s0 = sm + k * nsm + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          acc[b] += zt * s;
        }
      }
      f32x4 c[NBL];
      if (MODE == 2) {
        f32x4 snr[NBL];
        sob_act<NBL, 2>(A.act, acc, h, c, snr, n, g);
        if (TRAIN) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { pring[(0 * NBL + b) * 64] = c[b]; pring[(1 * NBL + b) * 64] = snr[b]; }
        }
      } else sine16_tagc<NBL>(acc, h, c);
#pragma unroll
      for (int b = 0; b < NBL; ++b) xc[b * 64] = c[b];
    } else {
      first_tangent(acc);
    }
    SW_MEET()
    if (q) {
#pragma unroll
      for (int b = 0; b < NBL; ++b) h[b] = xc[b * 64] * acc[b];
    }
    prefetch_inputs(tg + gridDim.x, (iset + 1) & 1);
    // ---- hidden hyper-matrices ------------------------------------------------------------------------------------------
    f32x4 ublk[(MODE == 1 && !TRAIN) ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
      if (TRAIN && active) { st_store16<NBL>(IN0 + (long)j * sstride, row0, h, g); }
      if (MODE == 1 && !TRAIN && !(j & 1)) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) ublk[b] = h[b];
      }
      bf16x8 b0[NCH], b1[NCH], b2[NCH];
      split3<NBL>(h, b0, b1, b2);
      if (q == 0) {
        const float* sb = sm + r * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
      } else ZERO_T(acc)
      for (int k = 0; k < r; ++k) {
        f32x4 T[NBL];
        if (q == 0) {
          const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
          for (int b = 0; b < NBL; ++b) T[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
        } else ZERO_T(T)
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) SW_CHUNK({ mfma_x6<NBL, PR, false, NBL, 0, CP>(cur, b0[ks], b1[ks], b2[ks], T, lane); })
        const float zt = zt_base[k * 16];
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * T[b];
      }
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) SW_CHUNK({ mfma_x6<NBL, PR, false, NBL, 0, CP>(cur, b0[ks], b1[ks], b2[ks], acc, lane); })
      const bool blk_end = (MODE == 1 && (j & 1)) || MODE == 2;      // the layer's result meets the block / layer input
      if (q == 0) {
        f32x4 c[NBL];
        if (MODE == 2) {
          f32x4 snr[NBL];
          sob_act<NBL, 2>(A.act, acc, acc, c, snr, n, g);
          if (TRAIN) {
#pragma unroll
            for (int b = 0; b < NBL; ++b) { pring[(((j + 1) * 2 + 0) * NBL + b) * 64] = c[b]; pring[(((j + 1) * 2 + 1) * NBL + b) * 64] = snr[b]; }
          }
        } else if (blk_end) sine16_tagc<NBL>(acc, acc, c);
        else sine16_tagc<NBL>(acc, h, c);
#pragma unroll
        for (int b = 0; b < NBL; ++b) xc[b * 64] = c[b];
      } else if (TRAIN) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) ring[((long)j * NBL + b) * 64] = acc[b];
      }
      SW_MEET()
      if (blk_end) {
        if (q) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) acc[b] = xc[b * 64] * acc[b];
        }
        // acc: sin(a2) (primal, tagged) / cos(a2) a2' (tangent); class NIF: f(a) / f'(a) a'
        if (MODE == 2) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) h[b] += acc[b];
          continue;
        }
        f32x4 u[NBL];
        if (TRAIN) st_load16<NBL>(IN0 + (long)(j - 1) * sstride, row0, u, g);
        else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) u[b] = ublk[b];
        }
#pragma unroll
        for (int b = 0; b < NBL; ++b) h[b] = 0.5f * (u[b] + acc[b]);
        if (TRAIN && q == 0) {      // the block output carries the tag of cos(a2)
#pragma unroll
          for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) h[b][v] = __uint_as_float((__float_as_uint(h[b][v]) & ~1u) | (__float_as_uint(acc[b][v]) & 1u));
        }
      } else if (q) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) h[b] = xc[b * 64] * acc[b];
      }
    }
    // ---- last layer (n -> so, linear), the stream's loss term, start of the adjoint ---------------------------------------
    if (!TRAIN) {      // predict: u (primal wave) / du/dx_d (tangent waves), nothing else
      for (int o = 0; o < so; ++o) {
        float part = 0.f, bias = 0.f;
        for (int k = 0; k <= r; ++k) {
          const float zt = k < r ? zt_base[k * 16] : 1.0f;
          const float* s0 = sm + k * nsm;
          float sk = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
            sk += (h[b][0] * w[0] + h[b][1] * w[1]) + (h[b][2] * w[2] + h[b][3] * w[3]);
          }
          part = fmaf(zt, sk, part);
          bias = fmaf(zt, s0[o_bl + o], bias);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float uo = q ? part : part + bias;
        if (valid && g == 0) {
          if (q == 0) { if (A.u_out) A.u_out[pt * so + o] = uo; }
          else if (J.JU) J.JU[(pt * so + o) * J.gstride + gcol] = uo;
        }
      }
      continue;
    }
    if (active) { st_store16<NBL>(IN0 + (long)nh * sstride, row0, h, g); }
    f32x4 gh[NBL];
    ZERO_T(gh)
    const float wsamp = (valid ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
    const float lw = q ? J.wjn : J.wu / (float)so;      // weight of this stream's squared errors (r4: SobArgs wu / wjn / ymask)
    float se = 0.f;
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
          wg[b] += zt * w;
        }
        part = fmaf(zt, sk, part);
        bias = fmaf(zt, s0[o_bl + o], bias);
        if (k < r) sks[k * 64 + lane] = sk;
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      const float uo = q ? part : part + bias;
      if (valid && g == 0) {
        if (q == 0) { if (A.u_out) A.u_out[pt * so + o] = uo; }
        else if (J.JU) J.JU[(pt * so + o) * J.gstride + gcol] = uo;
      }
      const float e = (q && !((J.ymask >> o) & 1u)) ? 0.0f : uo - ys[o * 16];       // tangent streams: the outputs of y_index only
      NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
      const float du = dfac * lw * wsamp * A.inv_bg;
      if (active && g == 0) A.DU[(((long)q * nt32 + tile32) * so + o) * 32 + poff] = du;
#pragma unroll
      for (int b = 0; b < NBL; ++b) gh[b] += du * wg[b];
      for (int k = 0; k < r; ++k) {
        float t = du * sks[k * 64 + lane];
        if (q == 0 && g == 0) t = fmaf(du, sm[k * nsm + o_bl + o], t);
        dzs[k * 64 + lane] += t;
      }
    }
    if (g == 0) loss_lane += wsamp * A.inv_bg * lw * se;
    // ---- adjoint through the hidden hyper-matrices ------------------------------------------------------------------------
    // ex: the primal wave's cos(a) | the tangent wave's a'_d of the layer -- ONE array, so that the two roles share its registers
    f32x4 hin[NBL], ex[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) hin[b] = h[b];          // primal: the tagged sin(a) of the top hidden layer
    if (q) {
#pragma unroll
      for (int b = 0; b < NBL; ++b) ex[b] = ring[((long)(nh - 1) * NBL + b) * 64];
    }
    f32x4 skip[MODE != 0 ? NBL : 1];
    for (int j = nh - 1; j >= 0; --j) {
      f32x4 ga[NBL];
      if (MODE == 2) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) skip[b] = gh[b];
      }
      if (MODE == 1 && (j & 1)) {      // second matrix of a block: half of the incoming adjoint passes the block by
#pragma unroll
        for (int b = 0; b < NBL; ++b) { skip[b] = 0.5f * gh[b]; gh[b] = skip[b]; }
        if (q == 0) {                  // hin = the block output with the tag of cos(a2): sin(a2) = 2 h - u, u = the block input
          f32x4 ub[NBL];
          st_load16<NBL>(IN0 + (long)(j - 1) * sstride, row0, ub, g);
#pragma unroll
          for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const float t = fmaf(2.0f, hin[b][v], -ub[b][v]);
              hin[b][v] = __uint_as_float((__float_as_uint(t) & ~1u) | (__float_as_uint(hin[b][v]) & 1u));
            }
        }
      }
      if (q == 0) {
        if (MODE == 2) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { ex[b] = pring[(((j + 1) * 2 + 0) * NBL + b) * 64]; hin[b] = pring[(((j + 1) * 2 + 1) * NBL + b) * 64]; }
        } else tag_cos<NBL>(hin, ex);
#pragma unroll
        for (int b = 0; b < NBL; ++b) xc[b * 64] = ex[b];
      } else {
#pragma unroll
        for (int b = 0; b < NBL; ++b) xw[b * 64] = gh[b] * ex[b];
      }
      SW_MEET()
      if (q == 0) {      // hin still holds sin(a) (class NIF: -f''(a))
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 w = xch[(1 * NBL + b) * 64 + lane];
#pragma unroll
          for (int d = 1; d < NS; ++d) w += xch[((1 + d) * NBL + b) * 64 + lane];
          ga[b] = gh[b] * ex[b] - hin[b] * w;
        }
      } else {
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = gh[b] * xc[b * 64];
      }
      st_load16<NBL>(IN0 + (long)j * sstride, row0, hin, g);     // the stream's input of this layer (dz) -- primal: also sin(a) of layer j-1
      if (active) {
        if (PR && NBL != 6 && A.da_bf16) st_store16_bf<NBL>(DA0 + (long)(j + 1) * sstride, row0, ga, g);
        else st_store16<NBL>(DA0 + (long)(j + 1) * sstride, row0, ga, g);
      }
      if (q == 0)
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
      for (int k = 0; k < r; ++k) {
        f32x4 U[NBL];
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) {
          if (ks == 0) SW_CHUNK({ mfma_x3<NBL, PR, true, NBL, 0, CP>(cur, b0[0], b1[0], U, lane); })
          else SW_CHUNK({ mfma_x3<NBL, PR, false, NBL, 0, CP>(cur, b0[ks], b1[ks], U, lane); })
        }
        const float zt = zt_base[k * 16];
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) s = fmaf(hin[b][v], U[b][v], s);
        if (k == 0) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) gh[b] = zt * U[b];
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) gh[b] += zt * U[b];
        }
        dzs[k * 64 + lane] += s;
      }
      if (q && j > 0) {      // a'_d of the layer below: in flight behind the constant plane's chunks (U's registers are free now)
#pragma unroll
        for (int b = 0; b < NBL; ++b) ex[b] = ring[((long)(j - 1) * NBL + b) * 64];
      }
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) SW_CHUNK({ mfma_x3<NBL, PR, false, NBL, 0, CP>(cur, b0[ks], b1[ks], gh, lane); })
      if (MODE == 2 || (MODE == 1 && !(j & 1))) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] += skip[b];
      }
    }
    // ---- first layer ------------------------------------------------------------------------------------------------------
    {
      f32x4 ga[NBL];
      if (q == 0) {
        if (MODE == 2) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { ex[b] = pring[(0 * NBL + b) * 64]; hin[b] = pring[(1 * NBL + b) * 64]; }
        } else tag_cos<NBL>(hin, ex);
#pragma unroll
        for (int b = 0; b < NBL; ++b) xc[b * 64] = ex[b];
      } else {
        first_tangent(ex);                                  // a'_d of the first layer, recomputed
#pragma unroll
        for (int b = 0; b < NBL; ++b) xw[b * 64] = gh[b] * ex[b];
      }
      SW_MEET()
      if (q == 0) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 w = xch[(1 * NBL + b) * 64 + lane];
#pragma unroll
          for (int d = 1; d < NS; ++d) w += xch[((1 + d) * NBL + b) * 64 + lane];
          ga[b] = gh[b] * ex[b] - hin[b] * w;
        }
      } else {
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = gh[b] * xc[b * 64];
      }
      if (active) { st_store16<NBL>(DA0, row0, ga, g); }
      for (int k = 0; k < r; ++k) {
        const float* s0 = sm + k * nsm + 4 * g;
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 t;
          if (q == 0) {
            t = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
            for (int dd = 0; dd < si; ++dd) t += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          } else t = *reinterpret_cast<const f32x4*>(s0 + o_w1 + seed * NP + 16 * b);
          s += (ga[b][0] * t[0] + ga[b][1] * t[1]) + (ga[b][2] * t[2] + ga[b][3] * t[3]);
        }
        float tot = dzs[k * 64 + lane] + s;
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        if (g == 0) dzx[(wid * r + k) * 16 + p] = tot;
      }
      SW_MEET()
      if (q == 0 && active && g == 0)
        for (int k = 0; k < r; ++k)
        {
          float t = dzx[(tl * r + k) * 16 + p];
#pragma unroll
          for (int d = 1; d < NQ; ++d) t += dzx[((d * TPG + tl) * r + k) * 16 + p];
          A.DZ[(tile32 * r + k) * 32 + poff] = t;
        }
    }
  }
#undef SW_CHUNK
#undef SW_MEET
  if (!TRAIN) return;
  for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
  if (lane == 0) lsum[wid] = loss_lane;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < WAVES; ++w) t += lsum[w];
    A.loss_partial[blockIdx.x] = t;
  }
}


// ---- launch of ONE mode's instantiations (each mode is a translation unit of its own: k_sobw.hip, k_sobw_res.hip, k_sobw_nif.hip --
// 48 kernels apiece compile side by side) -------------------------------------------------------------------------------------------
size_t sobw_shmem(const SNetArgs& a, int NBL, int ns);
template <int MODE>
static void launch_sobw_mode(const SobArgs& J_, int nblk, hipStream_t st, bool train) {
  SobArgs J = J_;
  if (J.s.prec == 1) { J.s.WF4 = J.s.WF4h; J.s.WB4 = J.s.WB4h; }      // k_sobw<.., PR> streams the policy's compact plane set
  const SNetArgs& a = J.s;
  const int NBL = snet3_nbl(a.n);
  const size_t shm = sobw_shmem(a, NBL, J.ns);
  dim3 grid(nblk), block(64 * sobw_tiles_per_group(a.n, J.ns) * (1 + J.ns));
#define SWM(NBL_, PR_, NS_, TR_)                                                                                               \
  {                                                                                                                      \
    (void)hipFuncSetAttribute((const void*)k_sobw<NBL_, PR_, NS_, TR_, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((k_sobw<NBL_, PR_, NS_, TR_, MODE>), grid, block, shm, st, J);                                                \
  }
#define SWL(NBL_, PR_, NS_) { if (train) SWM(NBL_, PR_, NS_, true) else SWM(NBL_, PR_, NS_, false) }
#define SWN(NBL_, PR_) { if (J.ns == 1) SWL(NBL_, PR_, 1) else if (J.ns == 2) SWL(NBL_, PR_, 2) else SWL(NBL_, PR_, 3) }
  if (NBL == 8) { if (a.prec == 1) SWN(8, true) else SWN(8, false) }
  else if (NBL == 6) { if (a.prec == 1) SWN(6, true) else SWN(6, false) }
  else if (NBL == 4) { if (a.prec == 1) SWN(4, true) else SWN(4, false) }
  else { if (a.prec == 1) SWN(2, true) else SWN(2, false) }
#undef SWN
#undef SWL
#undef SWM
}
void launch_sobw_plain(const SobArgs& J, int nblk, hipStream_t st, bool train);
void launch_sobw_res(const SobArgs& J, int nblk, hipStream_t st, bool train);
void launch_sobw_nif(const SobArgs& J, int nblk, hipStream_t st, bool train);
