// tools/exp/k_llg.hip -- EXPERIMENT, not part of the build (r4; DESIGN 5.4 "configs[3]: what bounds it").
//
// Hypothesis: k_snet4<8, .., LL> (configs[3], 128 x 6 last-layer class, 2 M points: 6.5 ms fp32 / 5.0 ms mixed_bfloat16) sits at the
// latency of its weight stream (one 16-point tile per wave and stream pass).  Three kernels were built on that hypothesis, all
// parity-green against tests/test_gpu_parity.py -k "ll_ or last_layer" (125 cases) when hooked into launch_snet4:
//   (1) PT = 2 / 4 tiles per wave through the same stream (every A operand read feeds PT MFMAs, 1 / PT of the passes):
//       fp32 7.4-7.7 ms, policy 7.3 (PT 4) / 5.8 (PT 2) -- SLOWER;
//   (2) + the adjoint's stash re-reads issued one layer ahead of their use: policy 5.1, fp32 7.4;
//   (3) + THIS FILE: a loader wave that owns every chunk DMA, so that no compute wave's in-order vmcnt ever couples its stash stores
//       to a DMA wait: fp32 6.5, policy 5.3, 64-wide 3.2 (k_snet4: 6.6 / 5.0 / 2.6).
// Ablations of (1): without the stash stores 4.6 (policy) / 5.4 (fp32); without stores and re-reads 2.4 / 4.4.
// tools/exp/stash_wbw_ll.hip then measured the stash traffic ALONE (no arithmetic, same layout, same sizes): 15 GB of stores 3.0-3.4 ms,
// stores + re-reads 5.3-6.0 ms.  The kernel IS its stash traffic; no restructuring of the compute side can move it.  What moves it
// is fewer stash bytes: the weight gradients accumulated in the kernel (k_snet6's scheme for shared weights), next round.
//
// k_llg.hip -- training kernel of the last-layer-parameterised class (model.py:1044-1068, :1219-1269; siren.py:272-280) that USES
// the fact that its ShapeNet weights are shared by every point (r4).
//
// k_snet4<.., LL> runs this class through the hypernetwork kernel's tile program: one 16-point tile per wave and round, the packed
// weight planes streamed L2 -> LDS once per round.  For a hypernetwork that stream IS the work (r + 1 planes per layer); for a
// shared-weight SIREN it is overhead: a 128-wide layer's chunk piece (12 KB) feeds 4 (policy) .. 24 MFMAs per wave before the next
// barrier, and the kernel sits at the chunk latency (profiles/r04: 4.97 ms per 2 M points under the policy = 6 % of the pipe).
// Here a wave carries PT tiles (64 points under the policy, 32 for fp32 results) through the SAME stream: every A operand read from
// LDS feeds PT MFMAs, a chunk step carries PT times the matrix work, and the batch needs 1 / PT of the stream passes.  Everything
// else is k_snet4's: packed planes (k_pack16b / k_pack_phi), stash rows and their formats (k_gw8 / k_gw_lds consume them), tagged
// sine, the phi-layer epilogue (exact products under either policy), the loss kinds, the outputs.
#include "k_snet3_dev.h"

#ifndef NIF_LLG_NBUF
#define NIF_LLG_NBUF 3      // LDS chunk buffers (the DMA runs NBUF - 1 chunk steps ahead)
#endif
#ifndef NIF_LLG_SPLIT8
#define NIF_LLG_SPLIT8 2    // chunk pieces per K-step of a 128-wide layer (12 KB each)
#endif
#ifndef NIF_LLG_PT_PR
#define NIF_LLG_PT_PR 2     // 16-point tiles per wave: mixed_bfloat16 (one operand split) ...
#endif
#ifndef NIF_LLG_PT_F32
#define NIF_LLG_PT_F32 2    // ... and fp32 results (three forward / two adjoint splits of the activations live in registers)
#endif

#define LLG_Z4(x) { (x)[0] = 0.f; (x)[1] = 0.f; (x)[2] = 0.f; (x)[3] = 0.f; }

// one chunk piece of a forward K-step for PT tiles: every A operand (NBS output blocks x 1 or 3 splits) is read once
template <int NBS, int NBL, int OB0, bool PR, int PT>
__device__ __forceinline__ void llg_fwd(const bf16x8* cur, const bf16x8 (&b0)[PT][NBL / 2], const bf16x8 (&b1)[PT][NBL / 2],
                                        const bf16x8 (&b2)[PT][NBL / 2], int ks, f32x4 (&T)[PT][NBL], int lane) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ob = 0; ob < NBS; ++ob) {
    const bf16x8 a0 = cur[(ob * 3 + 0) * 64 + lane];
    if (PR) {
#pragma unroll
      for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[t][ks], T[t][OB0 + ob], 0, 0, 0);
      continue;
    }
    const bf16x8 a1 = cur[(ob * 3 + 1) * 64 + lane], a2 = cur[(ob * 3 + 2) * 64 + lane];
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[t][ks], T[t][OB0 + ob], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2[t][ks], T[t][OB0 + ob], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0[t][ks], T[t][OB0 + ob], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1[t][ks], T[t][OB0 + ob], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0[t][ks], T[t][OB0 + ob], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][OB0 + ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[t][ks], T[t][OB0 + ob], 0, 0, 0);
  }
  __builtin_amdgcn_s_setprio(0);
}
// one chunk piece of an adjoint K-step (K = the outputs of slot ks): NBS input blocks x 1 or 2 splits; ZI: the chains start here
template <int NBS, int NBL, int IB0, bool PR, bool ZI, int PT>
__device__ __forceinline__ void llg_bwd(const bf16x8* cur, const bf16x8 (&b0)[PT][NBL / 2], const bf16x8 (&b1)[PT][NBL / 2], int ks,
                                        f32x4 (&T)[PT][NBL], int lane) {
  __builtin_amdgcn_s_setprio(1);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ib = 0; ib < NBS; ++ib) {
    const bf16x8 a0 = cur[(ib * 2 + 0) * 64 + lane];
    if (PR) {
#pragma unroll
      for (int t = 0; t < PT; ++t) T[t][IB0 + ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[t][ks], ZI ? z4 : T[t][IB0 + ib], 0, 0, 0);
      continue;
    }
    const bf16x8 a1 = cur[(ib * 2 + 1) * 64 + lane];
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][IB0 + ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1[t][ks], ZI ? z4 : T[t][IB0 + ib], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][IB0 + ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0[t][ks], T[t][IB0 + ib], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < PT; ++t) T[t][IB0 + ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[t][ks], T[t][IB0 + ib], 0, 0, 0);
  }
  __builtin_amdgcn_s_setprio(0);
}

// adjoint pre-phase of one tile: X <- cos(a) * X with cos(a) rebuilt from the tagged sine this wave stashed in the forward pass
// (slot `hs`), the product stored as the layer's dL/da stash rows (`ds`; BF: bf16 rows), block by block (8 registers of sine live)
template <int NBL, bool BF>
__device__ __forceinline__ void llg_adj_tile(const float* __restrict__ hs, float* __restrict__ ds, long row0, f32x4 (&X)[NBL], bool active, int g) {
  if (!active) return;
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
    const float* q = hs + (row0 + (long)(16 * b + 4 * g) * 32);
    f32x4 sn;
#pragma unroll
    for (int v = 0; v < 4; ++v) sn[v] = q[v * 32];
#ifdef NIF_ABL_NOLOAD      // measurement builds (results are wrong)
    if (row0 != -12345) { sn[0] = 0.5f; sn[1] = 0.25f; sn[2] = 0.125f; sn[3] = 0.75f; }
#endif
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float s = sn[v];
      const float c = __builtin_amdgcn_sqrtf(__builtin_amdgcn_fmed3f(fmaf(-s, s, 1.0f), 0.0f, 1.0f));
      X[b][v] *= __uint_as_float(__float_as_uint(c) | (__float_as_uint(s) << 31));
    }
#ifdef NIF_ABL_NOSTORE
    if (X[b][0] != 12345.678f) continue;
#endif
    if (BF) {
      __bf16* o = reinterpret_cast<__bf16*>(ds) + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v * 32] = (__bf16)X[b][v];
    } else {
      float* o = ds + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v * 32] = X[b][v];
    }
  }
}

// the same with the tagged sine in registers (H, prefetched a layer ahead)
template <int NBL, bool BF>
__device__ __forceinline__ void llg_adj_regs(const f32x4 (&H)[NBL], float* __restrict__ ds, long row0, f32x4 (&X)[NBL], bool active, int g) {
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float s = H[b][v];
      const float c = __builtin_amdgcn_sqrtf(__builtin_amdgcn_fmed3f(fmaf(-s, s, 1.0f), 0.0f, 1.0f));
      X[b][v] *= __uint_as_float(__float_as_uint(c) | (__float_as_uint(s) << 31));
    }
  }
  if (!active) return;
#ifdef NIF_ABL_NOSTORE
  if (X[0][0] != 12345.678f) return;
#endif
  if (BF) st_store16_bf<NBL>(ds, row0, X, g);
  else st_store16<NBL>(ds, row0, X, g);
}

template <int NBL, bool PR, int PT>
__global__ __launch_bounds__(512, 1) void k_llg(SNetArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // 8 waves: CW = 7 compute waves (PT tiles each) and ONE loader wave that issues every chunk DMA and is the only wave that waits
  // for one.  vmcnt completes in issue order, so a compute wave that waited for "its" part of the next chunk also waited for every
  // stash store it had issued before that DMA -- one chunk step of slack for 32 KB of stores per layer, then the HBM write latency
  // in the critical path (measured: the 128 x 6 step without the stores 4.6 instead of 7.3 ms, without stores and re-reads 2.4).
  // Now a compute wave's counter only ever holds its own stores and its stash re-reads (issued a layer ahead of their use).
  constexpr int NT = 512, CW = 7, NCH = NBL / 2;
  constexpr int SPL = NBL == 8 ? NIF_LLG_SPLIT8 : 1, NBS = NBL / SPL;
  constexpr int CF = NBS * 3 * 64, CB = NBS * 2 * 64;      // 16-byte units per forward / adjoint chunk piece
  constexpr int NBUF = NIF_LLG_NBUF, DIST = NBUF - 1;
  constexpr int PHF = 2 * 3 * 64;                          // units of a phi-layer forward chunk
  constexpr int NP = 16 * NBL, GT = CW * PT;
  static_assert(PHF <= CF, "a phi chunk must fit a chunk buffer");
  static_assert(DIST >= 1 && DIST <= 3 && (PT & 1) == 0, "k_llg shape");
  constexpr bool PF = PT <= 2;      // registers for the adjoint's stash re-reads one layer ahead
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, nh = A.nh, si = A.si, so = A.so, rl = A.rl, sou = A.so_u;
  const int FP = stash_fp(n);
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + GT - 1) / GT;
  const long my_groups = (ngroups - 1 - (long)blockIdx.x) / gridDim.x + 1;      // tile groups of this workgroup

  bf16x8* chunks = reinterpret_cast<bf16x8*>(smem);            // NBUF x CF units
  float* sm = smem + NBUF * CF * 4;
  // LDS image of the small vectors: first-layer rows (times omega_0), biases, phi bias, last_layer_bias, the rl x rl map
  const int o_w1 = 0, o_b1 = si * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP, o_llb = o_bl + ((so + 3) & ~3);
  const int o_lw = o_llb + ((sou + 3) & ~3), sm_tot = (o_lw + rl * rl + 3) & ~3;
  const int CX = (si + 3) & ~3, CZ = (rl + 3) & ~3, CY = (sou + 3) & ~3;
  const int NI = (CX + CZ + CY + 4) * 16;                      // input rows [column][16 points] of one tile
  const int pw = (so + rl + sou) * 16 + PT * NI;               // per-wave LDS floats: the epilogue's scratch + PT input sets
  float* phis = sm + sm_tot + (long)(wid < CW ? wid : 0) * pw; // phi / dphi [so][16]
  float* das = phis + so * 16;                                 // dL/da [rl][16]
  float* dul = das + rl * 16;                                  // du [sou][16]
  float* inp = dul + sou * 16;
  float* lsum = sm + sm_tot + (long)CW * pw;
  constexpr int NPC = NCH * SPL;                               // chunk pieces of one hidden matrix
  const int steps_per_group = 2 * nh * NPC + NCH + SPL;

#define LLG_VMW(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))
  auto wait_vm = [&](int k) {      // s_waitcnt vmcnt(<= k): the allowance rounded down to a constant
    if (k >= 60) LLG_VMW(60); else if (k >= 48) LLG_VMW(48); else if (k >= 40) LLG_VMW(40); else if (k >= 36) LLG_VMW(36);
    else if (k >= 32) LLG_VMW(32); else if (k >= 28) LLG_VMW(28); else if (k >= 24) LLG_VMW(24); else if (k >= 20) LLG_VMW(20);
    else if (k >= 18) LLG_VMW(18); else if (k >= 16) LLG_VMW(16); else if (k >= 14) LLG_VMW(14); else if (k >= 12) LLG_VMW(12);
    else if (k >= 8) LLG_VMW(8); else if (k >= 6) LLG_VMW(6); else if (k >= 4) LLG_VMW(4); else LLG_VMW(0);
  };
  // the inputs of the wave's PT tiles of tile group tgn: coordinates, ParameterNet output a, targets, sample weight -> LDS rows
  auto prefetch_inputs = [&](long tgn) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      long t16n = (tgn * CW + wid) * PT + t;
      if (t16n >= nt16) t16n = nt16 - 1;
      const long tile32n = t16n >> 1;
      const int poffn = 16 * (int)(t16n & 1) + p;
      long ptn = t16n * 16 + p;
      if (ptn >= A.B) ptn = A.B - 1;
      float* dst = inp + t * NI;
      for (int i0 = 0; i0 < CX; i0 += 4) {
        const int c = i0 + g < si ? i0 + g : si - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.xin + ptn * A.ncol + A.col0 + c),
                                         (__attribute__((address_space(3))) void*)(dst + i0 * 16), 4, 0, 0);
      }
      for (int i0 = 0; i0 < CZ; i0 += 4) {
        const int c = i0 + g < rl ? i0 + g : rl - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.Z + (tile32n * rl + c) * 32 + poffn),
                                         (__attribute__((address_space(3))) void*)(dst + (CX + i0) * 16), 4, 0, 0);
      }
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
    if (wid < CW) prefetch_inputs(blockIdx.x);
    for (int e = tid; e < sm_tot; e += NT) {
      float v = 0.f;
      if (e < o_b1) { const int dd = e / NP, f = e - dd * NP; if (f < n) v = A.omega * A.theta[(long)dd * n + f]; }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = A.theta[s_b1 + f]; }
      else if (e < o_bl) { const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP; if (f < n) v = A.theta[s_bh + (long)j * n + f]; }
      else if (e < o_bl + so) v = A.theta[s_bl + (e - o_bl)];
      else if (e >= o_llb && e < o_llb + sou) v = A.theta[s_bl + so + (e - o_llb)];
      else if (e >= o_lw && e < o_lw + rl * rl) v = A.theta[s_bl + so + sou + (e - o_lw)];
      sm[e] = v;
    }
  }
  if (wid == CW) {
    // ---- the loader wave: k_snet4's chunk stream (a running source pointer and a phase counter; r = 0), DIST chunk steps ahead --------
    const bf16x8* cs_src = reinterpret_cast<const bf16x8*>(A.WF4);
    int cs_units = CF, cs_left = nh * NPC, cs_phase = 0;
    long cs_groups = my_groups - 1;
    auto cs_next = [&](int buf) -> int {        // DMA of the stream's next chunk into buffer `buf`; returns the instructions issued
      if (cs_left < 0) return 0;
      bf16x8* dst = chunks + buf * CF;
      int nis = 0;
      for (int u = 0; u < cs_units; u += 64) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cs_src + u + lane),
                                         (__attribute__((address_space(3))) void*)(dst + u), 16, 0, 0);
        ++nis;
      }
      asm volatile("" ::: "memory");
      cs_src += cs_units;
      if (--cs_left == 0) {
        ++cs_phase;
        if (cs_phase == 1) { cs_src = reinterpret_cast<const bf16x8*>(A.WPF); cs_units = PHF; cs_left = NCH; }
        else if (cs_phase == 2) { cs_src = reinterpret_cast<const bf16x8*>(A.WPB); cs_units = CB; cs_left = SPL; }
        else if (cs_phase < 3 + nh) { cs_src = reinterpret_cast<const bf16x8*>(A.WB4) + (long)(nh - 1 - (cs_phase - 3)) * NPC * CB; cs_units = CB; cs_left = NPC; }
        else if (cs_groups <= 0) cs_left = -1;
        else { --cs_groups; cs_phase = 0; cs_src = reinterpret_cast<const bf16x8*>(A.WF4); cs_units = CF; cs_left = nh * NPC; }
      }
      return nis;
    };
    int y0 = 0, y1 = 0;                    // instructions of the newest / second newest DMA
    int nb = 0;
    for (int d = 0; d < DIST; ++d) { y1 = y0; y0 = cs_next(nb); nb = nb == NBUF - 1 ? 0 : nb + 1; }
    LLG_VMW(0);
    __syncthreads();                       // (1) the first DIST chunks and the small vectors are in LDS
    const long total = my_groups * steps_per_group;
    for (long s_ = 0; s_ < total; ++s_) {
      // the compute waves passed barrier s_ - 1: the buffer of chunk s_ - 1 is free; chunk s_ + 1 must have landed at barrier s_
      y1 = y0; y0 = cs_next(nb); nb = nb == NBUF - 1 ? 0 : nb + 1;
      wait_vm(DIST == 1 ? 0 : (DIST == 2 ? y0 : y0 + y1));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    __syncthreads();                       // (2) the loss partials
    return;
  }
  __syncthreads();                         // (1)
  int cbuf = 0;
  int vm_since = 0;                        // vector-memory instructions this wave issued after its newest input prefetch
  auto vm_note = [&](int k) { vm_since += k; };
  float loss_lane = 0.f;
  const long sstride = A.slot_stride, tstride = (long)FP * 32;
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * sstride;

// one chunk step of a compute wave: multiply chunk c (it landed before barrier c - 1), then meet the others and the loader
#define LLG_CHUNK(...)                                                        \
  {                                                                           \
    const bf16x8* cur = chunks + cbuf * CF;                                   \
    __VA_ARGS__                                                               \
    asm volatile("" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                             \
    asm volatile("" ::: "memory");                                            \
    cbuf = cbuf == NBUF - 1 ? 0 : cbuf + 1;                                   \
  }
#define LLG_FWD_STEP(KS_, T_)                                                                                  \
  _Pragma("unroll") for (int sp_ = 0; sp_ < SPL; ++sp_) {                                                      \
    if (sp_ == 0) LLG_CHUNK({ llg_fwd<NBS, NBL, 0, PR, PT>(cur, b0, b1, b2, KS_, T_, lane); })                 \
    else LLG_CHUNK({ llg_fwd<NBS, NBL, (SPL > 1 ? NBS : 0), PR, PT>(cur, b0, b1, b2, KS_, T_, lane); })        \
  }
#define LLG_BWD_STEP(KS_, U_, ZI_, PR_)                                                                        \
  _Pragma("unroll") for (int sp_ = 0; sp_ < SPL; ++sp_) {                                                      \
    if (sp_ == 0) LLG_CHUNK({ llg_bwd<NBS, NBL, 0, PR_, ZI_, PT>(cur, b0, b1, KS_, U_, lane); })               \
    else LLG_CHUNK({ llg_bwd<NBS, NBL, (SPL > 1 ? NBS : 0), PR_, ZI_, PT>(cur, b0, b1, KS_, U_, lane); })      \
  }
  static_assert(SPL <= 2, "chunk pieces per K-step");

  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
    if (tg != blockIdx.x) wait_vm(vm_since);      // this group's input rows (prefetched behind the previous group's phi layer) are in LDS
    // the wave's PT tiles are consecutive and start at an even 16-point tile: tile t = half (t & 1) of tile32 base32 + (t >> 1)
    const long base16 = (tg * CW + wid) * PT;
    const long base32 = base16 >> 1;
    const long rowb = base32 * tstride + p;
#define LLG_ACTIVE(t) (base16 + (t) < nt16)
#define LLG_T32(t) (base32 + ((t) >> 1))
#define LLG_POFF(t) (16 * ((t) & 1) + p)
#define LLG_ROW0(t) (rowb + ((t) >> 1) * tstride + 16 * ((t) & 1))
#define LLG_VALID(t) (LLG_ACTIVE(t) && (base16 + (t)) * 16 + p < A.B)
    f32x4 H[PF ? PT : 1][NBL];    // PF: the tagged sine of the adjoint's next layer, prefetched from the stash
    f32x4 X[PT][NBL];             // the activations of a layer / its accumulators / dL/dh in the adjoint
    // ---- first layer: a = x . (w0 W1) + b1, h = sin(a) tagged with the sign of cos(a) ---------------------------------------------
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const float* xs = inp + t * NI + p;
      const float* s0 = sm + 4 * g;
      f32x4 acc[NBL];
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
        for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        acc[b] = s;
      }
      sine16_tag<NBL>(acc, X[t]);
    }
    // ---- hidden matrices ------------------------------------------------------------------------------------------------------
    for (int j = 0; j < nh; ++j) {
      bf16x8 b0[PT][NCH], b1[PT][NCH], b2[PT][NCH];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        if (LLG_ACTIVE(t)) { st_store16<NBL>(IN0 + (long)j * sstride, LLG_ROW0(t), X[t], g); vm_note(4 * NBL); }
        if (PR) {
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) b0[t][ks][e] = (__bf16)X[t][2 * ks + (e >> 2)][e & 3];
        } else split3<NBL>(X[t], b0[t], b1[t], b2[t]);
        const float* sb = sm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) X[t][b] = *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) LLG_FWD_STEP(ks, X)
#pragma unroll
      for (int t = 0; t < PT; ++t) sine16_tag<NBL>(X[t], X[t]);
    }
    // ---- phi layer (exact products under either policy), u = Dot(phi, a) + bias, loss, start of the adjoint -------------------
    f32x4 T2[PT][2];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      if (LLG_ACTIVE(t)) { st_store16<NBL>(IN0 + (long)nh * sstride, LLG_ROW0(t), X[t], g); vm_note(4 * NBL); }
      LLG_Z4(T2[t][0]) LLG_Z4(T2[t][1])
    }
#pragma unroll
    for (int ks = 0; ks < NCH; ++ks)
      LLG_CHUNK({
        _Pragma("unroll") for (int t = 0; t < PT; ++t) {
          bf16x8 q0, q1, q2;
          _Pragma("unroll") for (int e = 0; e < 8; ++e) {
            const float x = X[t][2 * ks + (e >> 2)][e & 3];
            const __bf16 x0 = (__bf16)x;
            const float r1 = x - (float)x0;
            const __bf16 x1 = (__bf16)r1;
            q0[e] = x0; q1[e] = x1; q2[e] = (__bf16)(r1 - (float)x1);
          }
          mfma_x6<2>(cur, q0, q1, q2, T2[t], lane);
        }
      })
    {
      bf16x8 b0[PT][NCH], b1[PT][NCH];      // only K-step 0 is used: dphi (32 padded outputs) as the adjoint's B operand
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const float* zl = inp + t * NI + CX * 16;
        const float* ys = zl + CZ * 16 + p;
        const float* wsp = zl + (CZ + CY) * 16 + p;
        const float wsamp = (LLG_VALID(t) ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
        const long pt = (LLG_T32(t) * 32) + LLG_POFF(t);
        float se = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int o = 16 * b + 4 * g + v;
            if (o < so) phis[o * 16 + p] = T2[t][b][v] + sm[o_bl + o];
          }
        for (int s_ = 0; s_ < sou; ++s_) {
          float uo = sm[o_llb + s_];
          for (int jj = 0; jj < rl; ++jj) uo = fmaf(phis[(s_ * rl + jj) * 16 + p], zl[jj * 16 + p], uo);
          if (LLG_VALID(t) && g == 0 && A.u_out) A.u_out[pt * sou + s_] = uo;
          const float e = uo - ys[s_ * 16];
          NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
          const float du = dfac * wsamp * A.inv_bg / (float)sou;
          if (g == 0) {
            dul[s_ * 16 + p] = du;
            if (LLG_ACTIVE(t)) A.DU[(LLG_T32(t) * sou + s_) * 32 + LLG_POFF(t)] = du;
          }
        }
        if (g == 0) loss_lane += wsamp * se / (float)sou * A.inv_bg;
        // dL/da[j] = sum_s du[s] phi[s][j], then dL/dlatent through the rl x rl map
        for (int jj = g; jj < rl; jj += 4) {
          float da = 0.f;
          for (int s_ = 0; s_ < sou; ++s_) da = fmaf(dul[s_ * 16 + p], phis[(s_ * rl + jj) * 16 + p], da);
          das[jj * 16 + p] = da;
          if (LLG_ACTIVE(t)) A.DA_ll[(LLG_T32(t) * rl + jj) * 32 + LLG_POFF(t)] = da;
        }
        for (int k = g; k < rl; k += 4) {
          float dz = 0.f;
          for (int c = 0; c < rl; ++c) dz = fmaf(das[c * 16 + p], sm[o_lw + k * rl + c], dz);
          if (LLG_ACTIVE(t)) A.DZL[(LLG_T32(t) * rl + k) * 32 + LLG_POFF(t)] = dz;
        }
        // dphi[o] = du[s] a[j] replaces phi in LDS; it is also the "dL/dout" stash of the phi layer's weight gradient
        for (int s_ = 0; s_ < sou; ++s_) {
          const float du = dul[s_ * 16 + p];
          for (int jj = g; jj < rl; jj += 4) {
            const int o = s_ * rl + jj;
            const float dq = du * zl[jj * 16 + p];
            phis[o * 16 + p] = dq;
            if (LLG_ACTIVE(t)) A.DPHI[(LLG_T32(t) * so + o) * 32 + LLG_POFF(t)] = dq;
          }
        }
        f32x4 dq2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int o = 16 * b + 4 * g + v;
            dq2[b][v] = o < so ? phis[o * 16 + p] : 0.f;
          }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = dq2[e >> 2][e & 3];
          const __bf16 x0 = (__bf16)x;
          b0[t][0][e] = x0; b1[t][0][e] = (__bf16)(x - (float)x0);
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): every LDS read of this group's input rows has returned before the DMA may overwrite them
      vm_since = 0;
      prefetch_inputs(tg + gridDim.x);      // this group's inputs are consumed; the next group's land behind the adjoint's barriers
      if (PF) {      // the top layer's tagged sine is still in registers: the first adjoint step takes its cosine from there
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int b = 0; b < NBL; ++b) H[PF ? t : 0][b] = X[t][b];
      }
      // dL/dh = Wl dphi: the phi layer's adjoint pieces (3-product form), the chains of X start here
      LLG_BWD_STEP(0, X, true, false)
    }
    // ---- adjoint through the hidden matrices: dL/da_{j+1} = cos(a_{j+1}) dL/dh_{j+1}, dL/dh_j = (w0 W_j) dL/da_{j+1} -------------
    for (int j = nh - 1; j >= 0; --j) {
      bf16x8 b0[PT][NCH], b1[PT][NCH];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        if (PF) {
          if (PR && A.da_bf16) llg_adj_regs<NBL, true>(H[PF ? t : 0], DA0 + (long)(j + 1) * sstride, LLG_ROW0(t), X[t], LLG_ACTIVE(t), g);
          else llg_adj_regs<NBL, false>(H[PF ? t : 0], DA0 + (long)(j + 1) * sstride, LLG_ROW0(t), X[t], LLG_ACTIVE(t), g);
        } else if (PR && A.da_bf16) llg_adj_tile<NBL, true>(IN0 + (long)(j + 1) * sstride, DA0 + (long)(j + 1) * sstride, LLG_ROW0(t), X[t], LLG_ACTIVE(t), g);
        else llg_adj_tile<NBL, false>(IN0 + (long)(j + 1) * sstride, DA0 + (long)(j + 1) * sstride, LLG_ROW0(t), X[t], LLG_ACTIVE(t), g);
        if (LLG_ACTIVE(t)) vm_note(4 * NBL);
        if (PR) {
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) b0[t][ks][e] = (__bf16)X[t][2 * ks + (e >> 2)][e & 3];
        } else split2<NBL>(X[t], b0[t], b1[t]);
      }
      if (PF) {      // the next step's tagged sine (slot j; the first layer's when j == 0): in flight behind this layer's matrix work
#pragma unroll
        for (int t = 0; t < PT; ++t)
          if (LLG_ACTIVE(t)) { st_load16<NBL>(IN0 + (long)j * sstride, LLG_ROW0(t), H[PF ? t : 0], g); vm_note(4 * NBL); }
      }
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) {
        if (ks == 0) { LLG_BWD_STEP(0, X, true, PR) }
        else { LLG_BWD_STEP(ks, X, false, PR) }
      }
    }
    // ---- first layer: dL/da_0 (its weight gradient is k_gw_first's) ------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      if (PF) llg_adj_regs<NBL, false>(H[PF ? t : 0], DA0, LLG_ROW0(t), X[t], LLG_ACTIVE(t), g);
      else llg_adj_tile<NBL, false>(IN0, DA0, LLG_ROW0(t), X[t], LLG_ACTIVE(t), g);
      if (LLG_ACTIVE(t)) vm_note(4 * NBL);
    }
  }
#undef LLG_ACTIVE
#undef LLG_T32
#undef LLG_POFF
#undef LLG_ROW0
#undef LLG_VALID
#undef LLG_BWD_STEP
#undef LLG_FWD_STEP
#undef LLG_CHUNK
#undef LLG_VMW
  for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
  if (lane == 0) lsum[wid] = loss_lane;
  __syncthreads();                         // (2)
  if (tid == 0) A.loss_partial[blockIdx.x] = ((lsum[0] + lsum[1]) + (lsum[2] + lsum[3])) + ((lsum[4] + lsum[5]) + lsum[6]);
}

// ---- host side ---------------------------------------------------------------------------------
static int llg_pt(const SNetArgs&) { return 2; }      // 16-point tiles per compute wave
static size_t llg_shmem(const SNetArgs& a, int NBL, int PT) {
  const int SPL = NBL == 8 ? NIF_LLG_SPLIT8 : 1, NP = 16 * NBL;
  const size_t cf = (size_t)(NBL / SPL) * 3 * 64 * 16;
  const size_t sm_tot = (size_t)(a.si + 1 + a.nh) * NP + ((a.so + 3) & ~3) + ((a.so_u + 3) & ~3) + ((a.rl * a.rl + 3) & ~3) + 4;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + ((a.rl + 3) & ~3) + ((a.so_u + 3) & ~3) + 4) * 16;
  const size_t pw = (size_t)(a.so + a.rl + a.so_u) * 16 + PT * ni;
  return NIF_LLG_NBUF * cf + (sm_tot + 7 * pw + 8) * sizeof(float);
}
// training step of a plain (no resblocks) shared-weight SIREN of the last-layer class, 49-64 or 97-128 units
bool llg_supported(const SNetArgs& a, bool train) {
  static const int on = [] { const char* e = getenv("NIF_LLG"); return e ? atoi(e) : 1; }();
  const int NBL = snet3_nbl(a.n);
  if (!on || !train || !a.ll || a.res || a.nif_skip || a.act != ACT_SINE) return false;
  if ((NBL != 4 && NBL != 8) || a.nh < 1 || a.so > 32 || !a.WF4 || !a.WB4 || !a.WPF || !a.WPB) return false;
  return llg_shmem(a, NBL, llg_pt(a)) <= 160u * 1024u;
}
int launch_llg(const SNetArgs& a, bool query_only, hipStream_t st) {
  const int NBL = snet3_nbl(a.n), PT = llg_pt(a);
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 7 * PT - 1) / (7 * PT);
  const size_t shm = llg_shmem(a, NBL, PT);
  long cap = 256;                                   // one 8-wave workgroup per CU
  if (a.wg_cap > 0 && a.wg_cap < cap) cap = a.wg_cap;
  const int nblk = (int)(ngroups < cap ? ngroups : cap);
  if (query_only) return nblk;
#define LLG_L(NBL_, PR_, PT_)                                                                                              \
  {                                                                                                                        \
    (void)hipFuncSetAttribute((const void*)k_llg<NBL_, PR_, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);   \
    hipLaunchKernelGGL((k_llg<NBL_, PR_, PT_>), dim3(nblk), dim3(512), shm, st, a);                                        \
  }
  if (NBL == 8) { if (a.prec == 1) LLG_L(8, true, 2) else LLG_L(8, false, 2) }
  else { if (a.prec == 1) LLG_L(4, true, 2) else LLG_L(4, false, 2) }
#undef LLG_L
  return nblk;
}
