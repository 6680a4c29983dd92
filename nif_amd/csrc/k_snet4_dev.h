// k_snet4_dev.h -- the kernel template of k_snet4.hip (see there for the scheme), in a header so that the instantiations of the
// mixed_float16 policy (k_snet4_f16.hip) compile next to the others.
#pragma once
#include "k_snet3_dev.h"


#ifndef NIF_S4_OCC
#define NIF_S4_OCC 3
#endif
#ifndef NIF_S4_NBUF
#define NIF_S4_NBUF 2     // LDS chunk buffers of the <= 64-wide instantiations (the DMA runs NBUF - 1 chunk steps ahead)
#endif
#ifndef NIF_S4_SPLIT8
#define NIF_S4_SPLIT8 2     // 128-wide nets of the last-layer class: a K-step's chunk streams in this many pieces (output-block halves),
#endif                      // 2 x 12 KB of LDS per workgroup instead of 2 x 24: cfg-4 3.67 -> 2.64 ms (128 x 6: 7.80 -> 6.42); the
                            // hypernetwork classes (r + 1 planes per layer) lose with it: cfg-3 2.25 -> 2.38 ms, resblocks 3.01 -> 4.17
#ifndef NIF_S4_SPLIT4
#define NIF_S4_SPLIT4 1
#endif
#ifndef NIF_S4_NBUF_LL8
#define NIF_S4_NBUF_LL8 2
#endif
#ifndef NIF_S4_OCC_LL8
#define NIF_S4_OCC_LL8 2    // workgroups per CU of the 128-wide last-layer-class training kernel (168 registers: 3 would fit)
#endif
#ifndef NIF_S4_OCC_WIDE
#define NIF_S4_OCC_WIDE 2   // workgroups per CU of the 96- and 128-wide instantiations: 2 x 256 registers with ~200 spilled beat 1 x 512 (cfg-3: 3.45 -> 2.79 ms)
#endif

#define ZERO4_(x) { (x)[0] = 0.f; (x)[1] = 0.f; (x)[2] = 0.f; (x)[3] = 0.f; }
#define ZERO_T(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

// MODE: 0 = plain (NIFMultiScale without resblock), 1 = SIREN resblock, 2 = NIF skip connection
// SGN (SIREN nets, with or without resblocks): no act'(a) ring.  The next layer's stashed input IS sin(a), so cos(a) = +-sqrt(1 - sin^2): only the
// SIGN of cos(a) is kept -- as the least significant mantissa bit of the stashed sine itself (sine16_tag / tag_cos in
// k_snet3_dev.h; r3 -- r2 kept a 128-bit shift register per lane, which cost 3 pack instructions per element and limited the
// form to (nh + 1) * 4 * NBL <= 128 bits).  |error| of the rebuilt cosine <= 2.4e-4 in the measure-zero neighbourhood of
// cos = 0, ~1e-7 typically: gradient-path only; the forward activations move by at most one ulp.
// Resblocks (MODE 1): the first sine of a block is the second matrix's stashed input (as above); the block output
// 0.5 (u + sin(a2)) takes the tag of cos(a2), and the adjoint rebuilds sin(a2) = 2 h - u from the two stash rows (|error| ~2e-7).
// LL: last-layer-parameterised class (model.py:1044-1068, :1219-1269): the ShapeNet is a shared-weight dense SIREN
// (r = 0, one plane per layer) whose last layer emits phi [so_u x rl]; u = Dot(phi, a) + bias with the ParameterNet
// output a; the adjoint starts from dphi = du (x) a and also yields dL/da (and dL/dlatent through the rl x rl map).
//
// r3 (VALU diet): omega_0 is folded into the packed planes and into the LDS image of the first layer, the
// per-plane biases start the MFMA accumulators (no bias FMAs, no zeroing: acc = b^(r) + sum_k zt_k (b^(k) + h (w0 M^(k)))),
// zero-started chains take the inline constant as C, the chunk stream is a running pointer with a phase counter.
template <int NBL, bool TRAIN, int ACT, int MODE, bool SGN, bool LL, int PR = 0>
__global__ __launch_bounds__(256, (NBL <= 4 ? NIF_S4_OCC : ((NBL == 8 && LL && TRAIN && MODE == 0) ? NIF_S4_OCC_LL8 : NIF_S4_OCC_WIDE))) void k_snet4(SNetArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 256, WAVES = 4;
  constexpr int NCH = NBL / 2;                      // K-step chunks per plane
  constexpr int SPL = (NBL == 8 && LL) ? NIF_S4_SPLIT8 : (NBL == 4 ? NIF_S4_SPLIT4 : 1), NBS = NBL / SPL;   // chunk pieces per K-step, output (input) blocks per piece
  constexpr int CF = NBS * 3 * 64, CB = NBS * 2 * 64;   // 16-byte units per forward / adjoint chunk
  // the policies' compact plane set (late r4): ONE 16-bit plane per block -- a third / half of the chunk DMA (LDS buffers keep the
  // stride CF: the phi layer's chunks of the last-layer class stay split groups)
  constexpr bool CP = PR == 1 || PR == 2;
  // PR = 3 (r5): fp32-exact products on HALF pairs, as in k_snet6 -- planes (hi, lo) x operand (hi, lo), three v_mfma_f32_16x16x32_f16 per
  // pair in BOTH directions (k_pack16b mode 3: forward chunks in the adjoint geometry; split2h; tools/exp/f16_split_mfma.hip: as
  // accurate as the six bf16 products, better than the f32-input MFMA): half the forward matrix work, two thirds of its chunk bytes,
  // 22 significand bits in the data adjoint where the bf16 pairs carried 16.  SIREN nets only (|h| <= 1: sines scaled by 2^12 stay in
  // half's range; class NIF keeps the bf16 splits): the planes carry a power of two s_jk per (matrix, plane), dL/da one per point,
  // scaled back exactly (biases pre-scaled in the LDS image, s_r / s_k in the latent combine, 1 / (4096 s_r) in the sine's constants)
  constexpr bool X16 = PR == 3;
  // 16-bit phase stash of the layer inputs (r5, k_snet3_dev.h): plain SIREN training under mixed_bfloat16 on the 128-wide kernels,
  // where the reader of the rows is k_gw8<R, true, true>; SNetArgs.h_ph16 switches it (nif_api: producer and readers agree)
  constexpr bool PHC = TRAIN && SGN && MODE == 0 && PR == 1 && NBL == 8;
  const bool ph16 = PHC && A.h_ph16 != 0; (void)ph16;
  constexpr int CFH = CP ? NBS * 64 : (X16 ? CB : CF), CBH = CP ? NBS * 64 : CB;   // units per HIDDEN-matrix chunk
  constexpr int QF = (CF + NT - 1) / NT;
  // LDS ring of the chunk stream: NBUF buffers, the DMA runs DIST = NBUF - 1 chunk steps ahead of the MFMAs.  r2 had two buffers
  // and drained vmcnt(0) in front of every barrier: the L2 -> LDS latency of a chunk (~1.5-2 k cycles) had to hide behind ONE
  // chunk's 24 MFMAs (384 cycles) -- the s_memtime timeline showed ~1.3 k ticks per chunk step, i.e. the kernel was bound by
  // that latency, not by VALU issue (r3: the VALU diet alone moved it 1.10 -> 1.03 ms).  Three buffers for the <= 64-wide nets
  // (36 KB), two for the 96/128-wide ones (their chunks carry 4x the MFMAs and 2 x 24 KB is what fits twice per CU)
  constexpr int NBUF = NBL <= 4 ? NIF_S4_NBUF : ((NBL == 8 && LL) ? NIF_S4_NBUF_LL8 : 2), DIST = NBUF - 1;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = LL ? 0 : A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const int FP = stash_fp(n);                             // feature rows per stash tile (nif_internal.h)
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  bf16x8* chunks = reinterpret_cast<bf16x8*>(smem);            // NBUF x CF units
  float* sm = smem + NBUF * CF * 4;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  const int rl = LL ? A.rl : 0, sou = LL ? A.so_u : so;
  // per-wave input rows [column][16 points] of the tile: coordinates, latent (ParameterNet output), targets, sample
  // weight -- each group padded to 4 columns = one 64-lane LDS-DMA instruction; two sets: the NEXT tile's inputs are
  // fetched by DMA while the current tile computes, so no global-load latency sits in the tile's critical path
  const int nz = LL ? rl : r;
  const int CX = (si + 3) & ~3, CZ = (nz + 3) & ~3, CY = (sou + 3) & ~3;
  const int NI = (CX + CZ + CY + 4) * 16;
  const int pw = 2 * r * 64 + (LL ? (rl + so + sou) * 16 : 0) + 2 * NI;   // per-wave LDS floats
  float* dzs = sm + sm_tot + (long)wid * pw;
  float* sks = dzs + r * 64;
  float* phis = sks + r * 64;       // LL: phi / dphi [so][16], dL/da [rl][16], du [sou][16]
  float* das = phis + (LL ? so * 16 : 0);
  float* dul = das + rl * 16;
  float* inp = dul + (LL ? sou * 16 : 0);
  float* lsum = sm + sm_tot + (long)WAVES * pw;
  float* scl = lsum + 8;                              // X16: [matrix][plane][s | 1 / s] of the half planes
  const int o_llb = LL ? ((nsm - ((sou + 3) & ~3) - ((rl * rl + 3) & ~3))) : 0;   // LL extras sit at the end of sm
  const int o_lw = o_llb + ((sou + 3) & ~3);
  constexpr int NP = 16 * NBL;
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  // ---- the chunk stream: a running source pointer and a phase counter --------------------------------------------------
  //   phase 0: the forward planes (nh (r+1) NCH chunks of CF units, contiguous in WF);  1: the phi layer's forward chunks (LL);
  //   2: its adjoint chunk (LL, TRAIN);  3 + q: the adjoint planes of hidden matrix nh-1-q ((r+1) NCH chunks of CB units).
  // cs_next() issues the DMA of the next chunk of the stream into LDS buffer `buf` and steps the state; behind the last
  // phase the stream wraps to the next tile group's phase 0, or ends (cs_left < 0) when this workgroup has no further group.
  constexpr int PHF = 2 * 3 * 64;                       // units of a phi-layer forward chunk (LL)
  const int NPC = (r + 1) * NCH * SPL;                  // chunks of one hidden matrix
  const bf16x8* cs_src = reinterpret_cast<const bf16x8*>(A.WF4);
  int cs_units = CFH, cs_left = nh * NPC, cs_phase = 0;
  long cs_groups = (ngroups - 1 - (long)blockIdx.x) / gridDim.x;      // tile groups of this workgroup behind the current one
  auto cs_phase_step = [&]() {          // the current phase has been issued completely: find the next one that has chunks
    for (;;) {
      ++cs_phase;
      if (cs_phase == 1) { if (LL) { cs_src = reinterpret_cast<const bf16x8*>(A.WPF); cs_units = PHF; cs_left = NCH; return; } }
      else if (cs_phase == 2) { if (LL && TRAIN) { cs_src = reinterpret_cast<const bf16x8*>(A.WPB); cs_units = CB; cs_left = SPL; return; } }
      else if (TRAIN && cs_phase < 3 + nh) {
        cs_src = reinterpret_cast<const bf16x8*>(A.WB4) + (long)(nh - 1 - (cs_phase - 3)) * NPC * CBH; cs_units = CBH; cs_left = NPC; return;
      } else {
        if (cs_groups <= 0) { cs_left = -1; return; }
        --cs_groups; cs_phase = 0;
        cs_src = reinterpret_cast<const bf16x8*>(A.WF4); cs_units = CFH; cs_left = nh * NPC; return;
      }
    }
  };
  // vector-memory bookkeeping for the chunk waits (vmcnt decrements in issue order): vm_y1 = instructions this wave has issued
  // AFTER the DMA of the chunk the next wait is for, vm_y2 = after the newest DMA.  Stash stores / loads that are noted (vm_note)
  // may then still be in flight when the wait returns -- in r2 every chunk barrier also waited for the 16 stash stores of the
  // layer, i.e. for the HBM write path (the kernel without stash traffic: 0.71 ms, with: 1.04).  Un-noted instructions only make
  // a wait stricter; a noted one must really be issued, and after the DMA it is counted against (the fence in cs_next)
  int vm_y1 = 0, vm_y2 = 0;
  auto vm_note = [&](int n) { vm_y1 += n; vm_y2 += n; };
  auto cs_next = [&](int buf) -> int {        // returns the number of DMA instructions this WAVE issued
    if (cs_left < 0) return 0;
    bf16x8* dst = chunks + buf * CF;
    int nis = 0;
#pragma unroll
    for (int q = 0; q < QF; ++q)
      if (wid * 64 + NT * q < cs_units) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cs_src + tid + NT * q),
                                         (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);
        ++nis;
      }
    asm volatile("" ::: "memory");      // nothing that is counted as younger than this DMA may be hoisted above it
    cs_src += cs_units;
    if (--cs_left == 0) cs_phase_step();
    vm_y1 += nis; vm_y2 = 0;
    return nis;
  };
  // s_waitcnt vmcnt(n) alone (lgkmcnt / expcnt untouched; gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4] = 7,
  // lgkmcnt [11:8] = 15).  The immediate must be a constant: the allowance is rounded DOWN (which only waits for more) to
  // {0, 1, 2} stash tiles' worth of instructions plus 0..3 DMA instructions -- the values that occur in the steady state
#define NIF_VMW(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))
#define NIF_VMW4(B_, R_) { if ((R_) >= 3) NIF_VMW((B_) + 3); else if ((R_) == 2) NIF_VMW((B_) + 2); else if ((R_) == 1) NIF_VMW((B_) + 1); else NIF_VMW(B_); }
  auto wait_vm = [&](int n) {
    constexpr int S = 4 * NBL;
    if (DIST == 1) { NIF_VMW(0); return; }
    if (n >= 2 * S) NIF_VMW4(2 * S, n - 2 * S)
    else if (n >= S) NIF_VMW4(S, n - S)
    else NIF_VMW4(0, n)
  };
#undef NIF_VMW4
#undef NIF_VMW
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
    // LDS image of the small hyper-vectors, per plane k: first-layer rows (times omega_0), last-layer columns, biases
    for (int idx = tid; idx < (r + 1) * nsm; idx += NT) {
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
      else if (LL && e >= o_llb && e < o_llb + sou) v = hyp3(A, k, s_bl + so + (e - o_llb));
      else if (LL && e >= o_lw && e < o_lw + rl * rl) v = hyp3(A, k, s_bl + so + sou + (e - o_lw));
      sm[idx] = v;
    }
    if (X16) for (int idx = tid; idx < nh * (r + 1) * 2; idx += NT) scl[idx] = A.wscale[idx];
    if (cs_left <= 0) cs_left = -1;
#pragma unroll
    for (int d = 0; d < DIST; ++d) cs_next(d);       // the first DIST chunks
  }
  __syncthreads();
  vm_y1 = 0; vm_y2 = 0;                              // (the fence in front of the barrier drained vmcnt(0))
  int cbuf = 0, nbuf = DIST;                         // ring positions of the chunk being multiplied / the chunk being fetched
  int tlc = 0; (void)tlc;
#ifdef NIF_TIMELINE
#define NIF_TL(id) do { if (A.tl && blockIdx.x == 0 && tid == 0 && tlc < 250) { A.tl[2 * tlc] = (id); A.tl[2 * tlc + 1] = (long long)__builtin_amdgcn_s_memtime(); ++tlc; } } while (0)
#else
#define NIF_TL(id) do { } while (0)
#endif
  float loss_lane = 0.f;
  float* dring = (TRAIN && !SGN) ? A.dring + ((long)blockIdx.x * WAVES + wid) * (long)(nh + 1) * (NBL * 256) : nullptr;
#ifdef NIF_ABL_INTERLEAVE     // measurement builds: the ten slots of a tile next to each other (the consumers then read garbage)
  const long sstride = (long)FP * 32, tstride = (long)(2 * (nh + 1)) * FP * 32;
#else
  const long sstride = A.slot_stride, tstride = (long)FP * 32;
#endif
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * sstride;

// one chunk step: start the DMA of chunk c + DIST into the buffer that chunk c - 1 left, multiply chunk c, then wait until
// chunk c + 1 has landed -- i.e. until at most the vm_y1 instructions issued after ITS DMA are outstanding -- and meet the
// other three waves.  A raw s_barrier: __syncthreads() would make
// hipcc drain vmcnt(0) in front of it; the LDS traffic that has to be ordered here are the DMA writes (vmcnt, waited above) and
// the ds_reads of the chunk (their data is consumed by the MFMAs above); the empty asm statements keep the compiler from moving
// LDS accesses across.  Tried and not kept in r1/r2 (cfg-2, 1.11 ms): whole planes per step (half the barriers): 1.40 ms;
// 12-wave workgroups: 1.13 ms.
#define NIF_CHUNK(...)                                                        \
  {                                                                           \
    cs_next(nbuf);                                                            \
    const bf16x8* cur = chunks + cbuf * CF;                                   \
    __VA_ARGS__                                                               \
    wait_vm(vm_y1);                                                           \
    vm_y1 = vm_y2;                                                            \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    cbuf = cbuf == NBUF - 1 ? 0 : cbuf + 1;                                   \
    nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;                                   \
  }

// the SPL pieces of K-step ks_: forward (6-product) into T_, adjoint (3-product) into U_; ZI_: the chains start from zero
#define NIF_FWD_MM(OB0_, KS_, T_)                                                                                      \
  if (X16) mfma_x3<NBS, 3, false, NBL, OB0_, false>(cur, b0[KS_], b1[KS_], T_, lane);                                   \
  else mfma_x6<NBS, PR, false, NBL, OB0_, CP>(cur, b0[KS_], b1[KS_], b2[KS_], T_, lane);
#define NIF_FWD_STEP(KS_, T_)                                                                                          \
  _Pragma("unroll") for (int sp_ = 0; sp_ < SPL; ++sp_) {                                                              \
    if (sp_ == 0) NIF_CHUNK({ NIF_FWD_MM(0, KS_, T_) })                                                                 \
    else if (sp_ == 1) NIF_CHUNK({ NIF_FWD_MM((SPL > 1 ? NBS : 0), KS_, T_) })                                          \
    else if (sp_ == 2) NIF_CHUNK({ NIF_FWD_MM((SPL > 2 ? 2 * NBS : 0), KS_, T_) })                                      \
    else NIF_CHUNK({ NIF_FWD_MM((SPL > 3 ? 3 * NBS : 0), KS_, T_) })                                                    \
  }
#define NIF_BWD_STEP(B0_, B1_, U_, ZI_, PR_, ...)                                                                      \
  _Pragma("unroll") for (int sp_ = 0; sp_ < SPL; ++sp_) {                                                              \
    if (sp_ == 0) NIF_CHUNK({ mfma_x3<NBS, PR_, ZI_, NBL, 0, (PR_ == 1 || PR_ == 2)>(cur, B0_, B1_, U_, lane); __VA_ARGS__ })                  \
    else if (sp_ == 1) NIF_CHUNK({ mfma_x3<NBS, PR_, ZI_, NBL, (SPL > 1 ? NBS : 0), (PR_ == 1 || PR_ == 2)>(cur, B0_, B1_, U_, lane); })       \
    else if (sp_ == 2) NIF_CHUNK({ mfma_x3<NBS, PR_, ZI_, NBL, (SPL > 2 ? 2 * NBS : 0), (PR_ == 1 || PR_ == 2)>(cur, B0_, B1_, U_, lane); })   \
    else NIF_CHUNK({ mfma_x3<NBS, PR_, ZI_, NBL, (SPL > 3 ? 3 * NBS : 0), (PR_ == 1 || PR_ == 2)>(cur, B0_, B1_, U_, lane); })                 \
  }

  int iset = 0;
  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++iset) {
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
#ifdef NIF_ABL_STASHMASK      // measurement builds: the stash traffic folded onto a cache-resident window (results are wrong)
    const long row0 = (tile32 & NIF_ABL_STASHMASK) * (long)FP * 32 + poff;
#else
    const long row0 = tile32 * tstride + poff;
#endif
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    NIF_TL(1);
    f32x4 h[NBL], acc[NBL];
    unsigned phs[PHC ? 2 * NBL : 1]; (void)phs;      // 16-bit phase stash: the phases of h (k_snet3_dev.h, sine16_tag_ph)
    // ---- first layer: a = sum_k zt_k (x . (w0 W1^(k)) + b1^(k)) ---------------------------------------------------
    {
      const float* s0 = sm + r * nsm + 4 * g;           // the constant plane starts the sum
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
    if (TRAIN && SGN) {                                 // h = sin(a), the cosine's sign in its last mantissa bit
      if constexpr (PHC) { if (ph16) sine16_tag_ph<NBL>(acc, h, phs); else sine16_tag<NBL>(acc, h); }
      else sine16_tag<NBL>(acc, h);
    }
    else {
      f32x4 d[NBL];
      act16<NBL, ACT>(A.act, acc, h, d, n, g);
      if (TRAIN) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) reinterpret_cast<f32x4*>(dring)[b * 64 + lane] = d[b];
      }
    }
    NIF_TL(2);
    prefetch_inputs(tg + gridDim.x, (iset + 1) & 1);   // lands behind the hidden-layer barriers of THIS tile
    // ---- hidden hyper-matrices: a = b^(r) + h (w0 M^(r)) + sum_{k<r} zt_k (b^(k) + h (w0 M^(k))) -------------------
    f32x4 ublk[MODE == 1 ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
      if (TRAIN && active) {
        if constexpr (PHC) { if (ph16) st_store16_ph<NBL>(IN0 + (long)j * sstride, row0, phs, g); else st_store16<NBL>(IN0 + (long)j * sstride, row0, h, g); }
        else st_store16<NBL>(IN0 + (long)j * sstride, row0, h, g);
        vm_note(4 * NBL);
      }
      bf16x8 b0[NCH], b1[NCH], b2[NCH];
      if (X16) split2h<NBL>(h, 4096.0f, b0, b1);
      else split3p<NBL, PR>(h, b0, b1, b2);
      NIF_TL(10 + j);
      {
        const float* sb = sm + r * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
      for (int k = 0; k < r; ++k) {
        f32x4 T[NBL];
        const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) T[b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) NIF_FWD_STEP(ks, T)
        const float zt = X16 ? zt_base[k * 16] * (scl[(j * (r + 1) + r) * 2] * scl[(j * (r + 1) + k) * 2 + 1]) : zt_base[k * 16];   // (plane k's chain carries s_k, the sum s_r)
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * T[b];
      }
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) NIF_FWD_STEP(ks, acc)
      NIF_TL(30 + j);
      if (TRAIN && SGN && X16) sine16_tag_sc<NBL>(acc, acc, scl[(j * (r + 1) + r) * 2 + 1] * (1.0f / 4096.0f));
      else if (TRAIN && SGN) {
        if constexpr (PHC) { if (ph16) sine16_tag_ph<NBL>(acc, acc, phs); else sine16_tag<NBL>(acc, acc); }
        else sine16_tag<NBL>(acc, acc);
      }
      else {
        if (X16) {
          const float inv = scl[(j * (r + 1) + r) * 2 + 1] * (1.0f / 4096.0f);
#pragma unroll
          for (int b = 0; b < NBL; ++b) acc[b] *= inv;
        }
        f32x4 d[NBL];
        act16<NBL, ACT>(A.act, acc, acc, d, n, g);
        if (TRAIN) {
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
          if (TRAIN && SGN) {     // the block output carries the sign tag of cos(a2): sin(a2) itself is rebuilt as 2 h - u in the adjoint
#pragma unroll
            for (int b = 0; b < NBL; ++b)
#pragma unroll
              for (int v = 0; v < 4; ++v)
                h[b][v] = __uint_as_float((__float_as_uint(h[b][v]) & ~1u) | (__float_as_uint(acc[b][v]) & 1u));
          }
        }
      }
    }
    NIF_TL(3);
    // ---- last layer (n -> so, linear), MSE, start of the adjoint ---------------------------------
    if (TRAIN && active) { st_store16<NBL>(IN0 + (long)nh * sstride, row0, h, g); vm_note(4 * NBL); }
    f32x4 gh[NBL];
    ZERO_T(gh)
    const float wsamp = (valid ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
    float se = 0.f;
    if (LL) {
      // phi = Wl^T h + bl on the matrix cores (two 16-output blocks, the 6-product form), into the wave's LDS rows,
      // then per point u = Dot(phi, a) + bias
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
          NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
          const float du = dfac * wsamp * A.inv_bg / (float)sou;
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
          NIF_BWD_STEP(d0[0], d1[0], gh, false, false)
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
        NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
        const float du = dfac * wsamp * A.inv_bg / (float)so;
        if (active && g == 0) A.DU[(tile32 * so + o) * 32 + poff] = du;
#pragma unroll
        for (int b = 0; b < NBL; ++b) gh[b] += du * wg[b];
        for (int k = 0; k < r; ++k) {
          float t = du * sks[k * 64 + lane];
          if (g == 0) t = fmaf(du, sm[k * nsm + o_bl + o], t);
          dzs[k * 64 + lane] += t;
        }
      }
    }
    if (TRAIN) {
      if (g == 0) loss_lane += wsamp * se / (float)sou * A.inv_bg;
      NIF_TL(4);
      // ---- adjoint through the hidden hyper-matrices: dL/dh_in = sum_k zt_k (w0 M^(k)) dL/da -------------------------
      f32x4 skip[MODE == 0 ? 1 : NBL];
      f32x4 dnext[NBL], hin[NBL];
      if (SGN) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) hin[b] = h[b];     // (tagged) sin(a) of the top hidden layer is the last layer's input
      }
      for (int j = nh - 1; j >= 0; --j) {
        f32x4 ga[NBL];
        if (SGN) {
          if (MODE == 1 && (j & 1)) {
            // second matrix of a resblock: hin is the block output 0.5 (u + sin(a2)) with the tag of cos(a2); u = the block input
            f32x4 ub[NBL];
            st_load16<NBL>(IN0 + (long)(j - 1) * sstride, row0, ub, g);
#pragma unroll
            for (int b = 0; b < NBL; ++b)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const float t = fmaf(2.0f, hin[b][v], -ub[b][v]);
                ub[b][v] = __uint_as_float((__float_as_uint(t) & ~1u) | (__float_as_uint(hin[b][v]) & 1u));
              }
            tag_cos<NBL>(ub, dnext);
          } else if (PHC && ph16 && j < nh - 1) ph_cos<NBL>(hin, dnext);      // (hin holds the PHASE of a_j since the load below)
          else tag_cos<NBL>(hin, dnext);
          if (PHC && ph16) st_load16_ph<NBL>(IN0 + (long)j * sstride, row0, hin, g);
          else st_load16<NBL>(IN0 + (long)j * sstride, row0, hin, g);   // h_j: dz dot product now, sin(a) of layer j-1 next
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
        if (active) {
          if (PR == 1 && NBL != 6 && A.da_bf16) st_store16_bf<NBL>(DA0 + (long)(j + 1) * sstride, row0, ga, g);
          else st_store16<NBL>(DA0 + (long)(j + 1) * sstride, row0, ga, g);
          vm_note(4 * NBL);
        }
        // <dL/da, b^(k)> now, so that dL/da is dead once it is split and stashed (16 registers less across the planes)
        for (int k = 0; k < r; ++k) {
          const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
          float sbv = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
            sbv += (ga[b][0] * bb[0] + ga[b][1] * bb[1]) + (ga[b][2] * bb[2] + ga[b][3] * bb[3]);
          }
          dzs[k * 64 + lane] += X16 ? sbv * (scl[(j * (r + 1) + k) * 2 + 1] * (1.0f / 4096.0f)) : sbv;     // (the LDS image holds 4096 s_k b^(k))
        }
        bf16x8 b0[NCH], b1[NCH];
        // mixed_float16: dL/da enters the products as half(s dL/da) with a loss scale s PER POINT -- the power of two that brings
        // the point's largest |dL/da| into [2^14, 2^15) -- and the chains below carry s x their value until `gh` is scaled back.
        // Keras' LossScaleOptimizer holds ONE dynamic scale for the whole step (2^15 at first) and skips the steps that overflow;
        // a per-point power of two needs no history, cannot overflow and keeps the smallest adjoints half can represent
        // (exact scaling: only the rounding sees it).  A lane's point is lane & 15 in the B operand and in the C / D tile alike
        float ls = 1.0f, ils = 1.0f;
        if (PR >= 2) {
          float mx = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ga[b][0]), fabsf(ga[b][1]))), fmaxf(fabsf(ga[b][2]), fabsf(ga[b][3])));
          mx = fmaxf(mx, __shfl_xor(mx, 16));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const unsigned ef = (__float_as_uint(mx) >> 23) & 0xFFu;           // |dL/da|_max = 2^(ef - 127) x [1, 2)
          const unsigned sf = 268u - ef < 227u ? 268u - ef : 227u;           // s = 2^(141 - ef), at most 2^100 (all-zero rows)
          ls = __uint_as_float(sf << 23);
          ils = __uint_as_float((254u - sf) << 23);
        }
        if (X16) split2h<NBL>(ga, ls, b0, b1);
        else split2p<NBL, PR>(ga, b0, b1, ls);
        NIF_TL(50 + j);
        for (int k = 0; k < r; ++k) {
          f32x4 U[NBL];
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) {
            if (ks == 0) { NIF_BWD_STEP(b0[0], b1[0], U, true, PR, if (!SGN) st_load16<NBL>(IN0 + (long)j * sstride, row0, hin, g);) }
            else { NIF_BWD_STEP(b0[ks], b1[ks], U, false, PR) }
          }
          const float zt = X16 ? zt_base[k * 16] * (scl[(j * (r + 1) + r) * 2] * scl[(j * (r + 1) + k) * 2 + 1]) : zt_base[k * 16];
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) s = fmaf((PHC && ph16) ? __builtin_amdgcn_sinf(hin[b][v]) : hin[b][v], U[b][v], s);
          if (k == 0) {
#pragma unroll
            for (int b = 0; b < NBL; ++b) gh[b] = zt * U[b];
          } else {
#pragma unroll
            for (int b = 0; b < NBL; ++b) gh[b] += zt * U[b];
          }
          dzs[k * 64 + lane] += X16 ? (ils * scl[(j * (r + 1) + k) * 2 + 1]) * s : (PR == 2 ? ils * s : s);
        }
#pragma unroll
        for (int ks = 0; ks < NCH; ++ks) {
          if (LL && ks == 0) { NIF_BWD_STEP(b0[0], b1[0], gh, true, PR) }   // r = 0: the chain starts here
          else { NIF_BWD_STEP(b0[ks], b1[ks], gh, false, PR) }
        }
        if (PR >= 2) {
          const float f_ = X16 ? ils * scl[(j * (r + 1) + r) * 2 + 1] : ils;
#pragma unroll
          for (int b = 0; b < NBL; ++b) gh[b] *= f_;
        }
        NIF_TL(70 + j);
        if (MODE == 2 || (MODE == 1 && !(j & 1))) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) gh[b] += skip[b];
        }
      }
      NIF_TL(5);
      // ---- first layer ---------------------------------------------------------------------------
      {
        f32x4 ga[NBL];
        if (SGN) {
          if (PHC && ph16) ph_cos<NBL>(hin, dnext);
          else tag_cos<NBL>(hin, dnext);
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[b * 64 + lane];
        }
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
        if (active) { st_store16<NBL>(DA0, row0, ga, g); vm_note(4 * NBL); }
        for (int k = 0; k < r; ++k) {
          const float* s0 = sm + k * nsm + 4 * g;
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            f32x4 t = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
            for (int dd = 0; dd < si; ++dd) t += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
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
#undef NIF_BWD_STEP
#undef NIF_FWD_STEP
#undef NIF_FWD_MM
#undef NIF_CHUNK
  if (TRAIN) {
    for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
    if (lane == 0) lsum[wid] = loss_lane;
    __syncthreads();
    if (tid == 0) A.loss_partial[blockIdx.x] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  }
}

// ---- host side: LDS bytes of a launch ---------------------------------------------------------
// the product form (template parameter PR) of a launch: the policies, else the half pairs for SIREN nets whose planes are packed
static inline int snet4_pr(const SNetArgs& a) {
  if (a.prec == 1 || a.prec == 2) return a.prec;
  return (!a.nif_skip && a.WF4x && a.WB4x && a.wscale) ? 3 : 0;
}
static inline size_t snet4_shmem(const SNetArgs& a, int NBL) {
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const int nz = a.ll ? a.rl : a.r, sou = a.ll ? a.so_u : a.so;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + ((nz + 3) & ~3) + ((sou + 3) & ~3) + 4) * 16;
  const size_t pw = 2 * a.r * 64 + (a.ll ? (size_t)(a.rl + a.so + a.so_u) * 16 : 0) + 2 * ni;
  return (size_t)(NBL <= 4 ? NIF_S4_NBUF : ((NBL == 8 && a.ll) ? NIF_S4_NBUF_LL8 : 2)) * ((NBL == 8 && a.ll) ? NBL / NIF_S4_SPLIT8 : (NBL == 4 ? NBL / NIF_S4_SPLIT4 : NBL)) * 3 * 64 * 16 + (sm_tot + 4 * pw + 8 + 2 * (size_t)(a.nh > 0 ? a.nh : 1) * (a.r + 1)) * sizeof(float);   // NBUF chunk buffers (+ the plane scales of the half form)
}
