// k_snet5.hip -- the plain-SIREN training instantiation of k_snet4 with the hidden layers' WEIGHT GRADIENTS FUSED IN.
//
// Why: with the gradient reductions in separate kernels (k_gw_lds), every layer input h_j and every dL/da_j of every point
// makes a round trip through HBM: 2.5 KB/point written by the fused forward/adjoint kernel and 2.5 KB/point read back by the
// reductions -- ~6 GB per step at 2^20 points against 12.6 MB of algorithmic input (305x).  Here the reduction
//     dL/dM_j^(k)[in][out] = w0 sum_p zt_k(p) h_j[p][in] dL/da_{j+1}[p][out]
// is accumulated where both operands are live, in persistent MFMA accumulators:
//   * one workgroup = 8 waves = 8 sixteen-point tiles per step, ONE workgroup per CU (2 waves per SIMD, 256 registers each:
//     measured, the kernel loses ~8 % going from 3 to 2 waves per SIMD);
//   * the accumulators of ALL hidden matrices and planes (nh (r+1) n^2 floats = 128 KB at 4x64, r = 1) are spread over the 8
//     waves' registers: per layer, wave w owns the 32x32 block (plane k, input block bi, output block bo) number w: 16
//     registers per layer, 64 for four layers (+4 for a 16-column slice of the bias rows);
//   * in the adjoint sweep of layer j every wave deposits its tile's h_j and zt_k dL/da (k = 0..r) in LDS, transposed to
//     [feature][16 points] and already split into bf16 (hi, lo) pairs packed in one word; after the next chunk barrier every
//     wave runs its block over the 8 tiles: K = 16 points x (hi, lo) = 32 per tile, two v_mfma_f32_32x32x16_bf16 for
//     hi.hi + lo.lo and two (B halves swapped) for hi.lo + lo.hi -- all four terms, fp32 accumulation; the bias rows are
//     column sums of the same deposits (one v_mfma_f32_16x16x32_bf16 per tile with an indicator A operand);
//   * the block product runs right after the layer's own adjoint products (few live registers there), one extra barrier per layer;
//   * at the end every workgroup writes ITS accumulators into its partial-gradient row, exactly where k_gw_lds would have.
// What is left of the stash: the inputs h_j of the hidden layers go through a PRIVATE per-wave ring (forward -> adjoint of the
// same tile: 1 KB/point that stays in L2), and only the first layer's dL/da and the last layer's input still travel to the
// skinny gradient kernels (k_gw_first / k_gw_out): 0.5 KB/point instead of 2.5.
//
// Built for: NIFMultiScale without resblocks (sign-bit cosine), n <= 64 (NBL = 2, 4), nh <= 4 hidden matrices,
// (r+1) (NBL/2)^2 <= 8 blocks per layer.  Everything else keeps the k_snet4 + k_gw_lds path.
#include "k_snet3_dev.h"

typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define ZERO_T5(x) _Pragma("unroll") for (int b_ = 0; b_ < NBL; ++b_) { (x)[b_][0] = 0.f; (x)[b_][1] = 0.f; (x)[b_][2] = 0.f; (x)[b_][3] = 0.f; }

// fp32 -> one word (bf16 hi | bf16 lo << 16), x ~= hi + lo to 2^-17
__device__ __forceinline__ unsigned pack_hilo(float x) {
  const __bf16 hi = (__bf16)x;
  const __bf16 lo = (__bf16)(x - (float)hi);
  return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}
// exchange-buffer word of (feature f, point p): rows of 16 words, the 16-byte columns XOR-swizzled with the row so that
// the consumers' ds_read_b128 (lane = row) and the producers' ds_write_b32 (lane = point) are both conflict-free enough
// (swizzle key (f >> 2) & 3: for the producer lane (p, g) with f = 16 b + 4 g + v it is just g, so all its 16 stores share one
// address register and differ by immediate offsets; the consumers' 16-byte reads see a 2-way bank conflict)
__device__ __forceinline__ int ex_word(int f, int p) { return f * 16 + ((((p >> 2) ^ (f >> 2)) & 3) << 2) + (p & 3); }

struct S5Args {
  SNetArgs s;
  float* partial; long pstride;     // partial-gradient rows [gridDim.x][pstride]
};

template <int NBL>
__global__ __launch_bounds__(512, 1) void k_snet5(S5Args F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SNetArgs& A = F.s;
  constexpr int NT = 512, WAVES = 8;
  constexpr int NCH = NBL / 2;
  constexpr int NBH = NBL / 2;                           // 32-feature blocks per dimension
  constexpr int CF = NBL * 3 * 64, CB = NBL * 2 * 64;
  constexpr int QF = (CF + NT - 1) / NT;
  constexpr int FPAD = 16 * NBL;
  constexpr int NP = 16 * NBL;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  bf16x8* chunks = reinterpret_cast<bf16x8*>(smem);
  float* sm = smem + 2 * CF * 4;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  const int CX = (si + 3) & ~3, CZ = (r + 3) & ~3, CY = (so + 3) & ~3;
  const int NI = (CX + CZ + CY + 4) * 16;
  const int pw = 2 * r * 64 + 2 * NI;
  float* dzs = sm + sm_tot + (long)wid * pw;
  float* sks = dzs + r * 64;
  float* inp = sks + r * 64;
  float* lsum = sm + sm_tot + (long)WAVES * pw;
  unsigned* EX = reinterpret_cast<unsigned*>(lsum + 8);          // [tile 8][plane r+2][FPAD][16] words
  const int EXT = (r + 2) * FPAD * 16;                           // words per tile
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  const int NPL = nh * (r + 1);
  const int nfwd = NPL * NCH;
  const int nchunks = 2 * nfwd;
  const bf16x8* WF = reinterpret_cast<const bf16x8*>(A.WF4);
  const bf16x8* WB = reinterpret_cast<const bf16x8*>(A.WB4);
  auto chunk_units = [&](int i) -> int { return i < nfwd ? CF : CB; };
  auto chunk_src = [&](int i) -> const bf16x8* {
    if (i < nfwd) return WF + (long)i * CF;
    const int ii = i - nfwd;
    const int pp = ii / NCH, ks = ii - pp * NCH;
    const int j = nh - 1 - pp / (r + 1), k = pp % (r + 1);
    return WB + (((long)j * (r + 1) + k) * NCH + ks) * CB;
  };
  auto dma = [&](int i, int buf) {
    const bf16x8* src = chunk_src(i);
    bf16x8* dst = chunks + buf * CF;
    const int nun = chunk_units(i);
#pragma unroll
    for (int q = 0; q < QF; ++q)
      if (wid * 64 + NT * q < nun)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid + NT * q),
                                         (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);
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
  };
  {
    const long s_wl = (long)si * n + (long)nh * n * n;
    const long s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
    prefetch_inputs(blockIdx.x, 0);
    for (int idx = tid; idx < (r + 1) * nsm; idx += NT) {
      const int k = idx / nsm, e = idx - k * nsm;
      float v = 0.f;
      if (e < o_wl) { const int dd = e / NP, f = e - dd * NP; if (f < n) v = hyp3(A, k, (long)dd * n + f); }
      else if (e < o_b1) { const int o = (e - o_wl) / NP, f = (e - o_wl) - o * NP; if (f < n) v = hyp3(A, k, s_wl + (long)f * so + o); }
      else if (e < o_bh) { const int f = e - o_b1; if (f < n) v = hyp3(A, k, s_b1 + f); }
      else if (e < o_bl) { const int j = (e - o_bh) / NP, f = (e - o_bh) - j * NP; if (f < n) v = hyp3(A, k, s_bh + (long)j * n + f); }
      else if (e < o_bl + so) v = hyp3(A, k, s_bl + (e - o_bl));
      sm[idx] = v;
    }
    if (nchunks > 0) dma(0, 0);
  }
  __syncthreads();
  int gpar = 0;
  float loss_lane = 0.f;
  constexpr int SW = 4 * NBL;
  // hidden-layer inputs: private ring of this wave, [layer j][feature][16 points] -- written in the forward sweep, read back
  // in the adjoint sweep of the SAME tile, never by another kernel
  float* ring = A.dring + ((long)blockIdx.x * WAVES + wid) * (long)nh * (FPAD * 16);
  float* INL = A.stash + (long)nh * A.slot_stride;            // last layer's input: for k_gw_out
  float* DA0 = A.stash + (long)(nh + 1) * A.slot_stride;      // first layer's dL/da: for k_gw_first
  const int FP = ((n + 31) / 32) * 32;

  // ---- this wave's gradient block: (plane uk, input block ubi, output block ubo) of every layer ----------------------
  const int ULY = (r + 1) * NBH * NBH;                          // blocks per layer (<= 8)
  const bool has_unit = wid < ULY;
  const int uk = has_unit ? wid / (NBH * NBH) : 0;
  const int ubi = (wid % (NBH * NBH)) / NBH, ubo = wid % NBH;
  const int i32 = lane & 31, kg = lane >> 5;
  // layer j = 0..3: this wave's matrix block
  f32x16v G0, G1, G2, G3;
#pragma unroll
  for (int e = 0; e < 16; ++e) { G0[e] = 0.f; G1[e] = 0.f; G2[e] = 0.f; G3[e] = 0.f; }
  // bias rows: wave w also owns the 16 columns (plane bkb, column block bcb) of every layer's bias gradient = column sums of
  // zt_k dL/da over the points: one v_mfma_f32_16x16x32_bf16 per tile whose A operand is the indicator of row j, so that
  // row j of the 16x16 accumulator (4 registers) collects layer j
  const bool has_bunit = wid < (r + 1) * NBL;
  const int bkb = has_bunit ? wid / NBL : 0, bcb = wid % NBL;
  const int c16 = lane & 15, kq = lane >> 4;
  const int sbb5 = ((16 * bcb + c16) >> 2) & 3;
  f32x4 GBs = {0.f, 0.f, 0.f, 0.f};

  // consume tiles [T0, T1) of the deposited layer into the block accumulator GM (a macro: the accumulators must stay
  // named registers -- handed to a lambda by reference they end up in scratch memory)
#define NIF5_CONSUME(GM, JROW, T0, T1)                                                                                  \
  if (has_bunit) {                                                                                                      \
    bf16x8 rowj;                                                                                                        \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) rowj[e_] = (__bf16)(c16 == (JROW) ? 1.0f : 0.0f);                  \
    _Pragma("unroll 1") for (int t_ = (T0); t_ < (T1) && t_ < WAVES; ++t_) {                                            \
      const unsigned* eb_ = EX + t_ * EXT + (1 + bkb) * FPAD * 16 + (16 * bcb + c16) * 16;                              \
      const u32x4 bw_ = *reinterpret_cast<const u32x4*>(eb_ + ((kq ^ sbb5) << 2));                                      \
      GBs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowj, __builtin_bit_cast(bf16x8, bw_), GBs, 0, 0, 0);              \
    }                                                                                                                   \
  }                                                                                                                     \
  if (has_unit) {                                                                                                       \
    _Pragma("unroll 1") for (int t_ = (T0); t_ < (T1) && t_ < WAVES; ++t_) {                                            \
      const unsigned* ea_ = EX + t_ * EXT + (32 * ubi + i32) * 16;                        /* plane 0: h */             \
      const unsigned* eb_ = EX + t_ * EXT + (1 + uk) * FPAD * 16 + (32 * ubo + i32) * 16; /* plane 1+k: zt_k dL/da */  \
      _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                                \
        const u32x4 aw_ = *reinterpret_cast<const u32x4*>(ea_ + (((2 * q_ + kg) ^ sa5) << 2));                          \
        const u32x4 bw_ = *reinterpret_cast<const u32x4*>(eb_ + (((2 * q_ + kg) ^ sb5) << 2));                          \
        u32x4 bs_;                                                                                                      \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) bs_[c_] = __builtin_amdgcn_alignbit(bw_[c_], bw_[c_], 16);     \
        const bf16x8 av_ = __builtin_bit_cast(bf16x8, aw_), bv_ = __builtin_bit_cast(bf16x8, bw_);                      \
        const bf16x8 bx_ = __builtin_bit_cast(bf16x8, bs_);                                                             \
        GM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av_, bx_, GM, 0, 0, 0);   /* hi.lo + lo.hi */                      \
        GM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av_, bv_, GM, 0, 0, 0);   /* hi.hi + lo.lo */                      \
      }                                                                                                                 \
    }                                                                                                                   \
  }
  const int sa5 = ((32 * ubi + i32) >> 2) & 3, sb5 = ((32 * ubo + i32) >> 2) & 3;
#define NIF5_CONSUME_J(T0, T1)                                                              \
  switch (j) {                                                                              \
    case 0: NIF5_CONSUME(G0, 0, T0, T1) break;                                              \
    case 1: NIF5_CONSUME(G1, 1, T0, T1) break;                                              \
    case 2: NIF5_CONSUME(G2, 2, T0, T1) break;                                              \
    default: NIF5_CONSUME(G3, 3, T0, T1) break;                                             \
  }

#define NIF5_CHUNK(...)                                                        \
  {                                                                            \
    if ((cc + 1 < nchunks) || !last_group) dma(cc + 1 < nchunks ? cc + 1 : 0, (gpar + 1) & 1); \
    const bf16x8* cur = chunks + (gpar & 1) * CF;                              \
    __VA_ARGS__                                                                \
    __syncthreads();                                                           \
    ++gpar; ++cc;                                                              \
  }

  int iset = 0;
  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x, ++iset) {
    const bool last_group = tg + gridDim.x >= ngroups;
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
    const long row0 = tile32 * (long)FP * 32 + poff;
    for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    f32x4 h[NBL], acc[NBL];
    unsigned long long sg_lo = 0ull, sg_hi = 0ull;
    // ---- first layer ---------------------------------------------------------------------------
    ZERO_T5(acc)
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 16] : 1.0f;
      const float* s0 = sm + k * nsm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dd = 0; dd < si; ++dd) s += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        acc[b] += zt * (A.omega * s + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b));
      }
    }
    {
      f32x4 d[NBL];
      sine16_sign<NBL>(acc, h, d);
      sgn_push(sg_lo, sg_hi, sgn_pack<NBL>(d), SW);
    }
    prefetch_inputs(tg + gridDim.x, (iset + 1) & 1);
    // ---- hidden hyper-matrices, forward ----------------------------------------------------------
    int cc = 0;
    for (int j = 0; j < nh; ++j) {
      {   // h_j -> private ring [j][feature][16 points]
        float* rj = ring + (long)j * (FPAD * 16) + p;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) rj[(16 * b + 4 * g + v) * 16] = h[b][v];
      }
      bf16x8 b0[NCH], b1[NCH], b2[NCH];
      split3<NBL>(h, b0, b1, b2);
      ZERO_T5(acc)
      for (int k = 0; k <= r; ++k) {
        if (k < r) {
          f32x4 T[NBL];
          ZERO_T5(T)
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) NIF5_CHUNK({ mfma_x6<NBL>(cur, b0[ks], b1[ks], b2[ks], T, lane); })
          const float zt = zt_base[k * 16];
#pragma unroll
          for (int b = 0; b < NBL; ++b) acc[b] += zt * T[b];
        } else {
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) NIF5_CHUNK({ mfma_x6<NBL>(cur, b0[ks], b1[ks], b2[ks], acc, lane); })
        }
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) acc[b] *= A.omega;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
      {
        f32x4 d[NBL];
        sine16_sign<NBL>(acc, acc, d);
        sgn_push(sg_lo, sg_hi, sgn_pack<NBL>(d), SW);
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) h[b] = acc[b];
    }
    // ---- last layer (n -> so, linear), MSE, start of the adjoint ---------------------------------
    if (active) st_store16<NBL>(INL, row0, h, g);
    f32x4 gh[NBL];
    ZERO_T5(gh)
    const float wsamp = (valid ? (A.sw ? wsp[0] : 1.0f) : 0.0f);
    float se = 0.f;
    for (int o = 0; o < so; ++o) {
      f32x4 wg[NBL];
      ZERO_T5(wg)
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
      const float uo = part + bias;
      const float e = uo - ys[o * 16];
      se = fmaf(e, e, se);
      const float du = 2.0f * wsamp * e * A.inv_bg / (float)so;
      if (active && g == 0) A.DU[(tile32 * so + o) * 32 + poff] = du;
#pragma unroll
      for (int b = 0; b < NBL; ++b) gh[b] += du * wg[b];
      for (int k = 0; k < r; ++k) {
        float t = du * sks[k * 64 + lane];
        if (g == 0) t = fmaf(du, sm[k * nsm + o_bl + o], t);
        dzs[k * 64 + lane] += t;
      }
    }
    if (g == 0) loss_lane += wsamp * se / (float)so * A.inv_bg;
    // ---- adjoint through the hidden hyper-matrices, weight gradients on the way -------------------
    f32x4 dnext[NBL], hin[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) hin[b] = h[b];
    for (int j = nh - 1; j >= 0; --j) {
      f32x4 ga[NBL];
      sgn_cos<NBL>(hin, sgn_pop(sg_lo, sg_hi, SW), dnext);
      {   // h_j back from the private ring
        const float* rj = ring + (long)j * (FPAD * 16) + p;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) hin[b][v] = rj[(16 * b + 4 * g + v) * 16];
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
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
#ifndef NIF5_ABL_NODEPOSIT
      {   // deposit: plane 0 = h_j, plane 1 + k = zt_k dL/da (k = r: dL/da itself); an idle wave deposits zeros
        unsigned* et = EX + wid * EXT + ex_word(4 * g, p);     // + (16 b + v) * 16 words per value: immediate offsets
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int w_ = (16 * b + v) * 16;
            et[w_] = active ? pack_hilo(hin[b][v]) : 0u;
            for (int k = 0; k <= r; ++k) {
              const float zt = k < r ? zt_base[k * 16] : 1.0f;
              et[(1 + k) * FPAD * 16 + w_] = active ? pack_hilo(zt * ga[b][v]) : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);     // one value at a time: packing all 16 first costs ~40 live registers
          }
      }
#endif
      bf16x8 b0[NCH], b1[NCH];
      split2<NBL>(ga, b0, b1);
      ZERO_T5(gh)
      for (int k = 0; k <= r; ++k) {
        if (k < r) {
          f32x4 U[NBL];
          ZERO_T5(U)
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) {
            NIF5_CHUNK({ mfma_x3<NBL>(cur, b0[ks], b1[ks], U, lane); })
          }
          const float zt = zt_base[k * 16];
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            gh[b] += zt * U[b];
#pragma unroll
            for (int v = 0; v < 4; ++v) s = fmaf(hin[b][v], U[b][v], s);
          }
          dzs[k * 64 + lane] = fmaf(A.omega, s, dzs[k * 64 + lane]);
        } else {
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks) {
            NIF5_CHUNK({ mfma_x3<NBL>(cur, b0[ks], b1[ks], gh, lane); })
          }
        }
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) gh[b] *= A.omega;
      // this layer's weight-gradient block over the 8 deposited tiles (the deposits are visible: the chunk barriers lie in
      // between); one more barrier before the next layer's deposit overwrites the exchange buffer
#ifndef NIF5_ABL_NOCONSUME
      NIF5_CONSUME_J(0, WAVES)
#endif
      __syncthreads();
    }
    // ---- first layer ---------------------------------------------------------------------------
    {
      f32x4 ga[NBL];
      sgn_cos<NBL>(hin, sgn_pop(sg_lo, sg_hi, SW), dnext);
#pragma unroll
      for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
      if (active) st_store16<NBL>(DA0, row0, ga, g);
      for (int k = 0; k < r; ++k) {
        const float* s0 = sm + k * nsm + 4 * g;
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 xw = {0.f, 0.f, 0.f, 0.f};
          for (int dd = 0; dd < si; ++dd) xw += xs[dd * 16] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
          const f32x4 t = A.omega * xw + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
          s += (ga[b][0] * t[0] + ga[b][1] * t[1]) + (ga[b][2] * t[2] + ga[b][3] * t[3]);
        }
        float tot = dzs[k * 64 + lane] + s;
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        if (active && g == 0) A.DZ[(tile32 * r + k) * 32 + poff] = tot;
      }
    }
  }
#undef NIF5_CHUNK
#undef NIF5_CONSUME_J
#undef NIF5_CONSUME
  // ---- epilogue: loss partial, this workgroup's accumulators -> its partial-gradient row -----------------------------
  for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
  if (lane == 0) lsum[wid] = loss_lane;
  __syncthreads();
  if (tid == 0)
    A.loss_partial[blockIdx.x] = ((lsum[0] + lsum[1]) + (lsum[2] + lsum[3])) + ((lsum[4] + lsum[5]) + (lsum[6] + lsum[7]));
  if (has_unit) {
    float* prow = F.partial + (long)blockIdx.x * F.pstride;
    const long kbase = uk < r ? A.off_Wh + (long)uk * A.po : A.off_bh;
    const long s_bh = (long)si * n + (long)nh * n * n + (long)n * so + n;
    const int out = 32 * ubo + i32;
#define NIF5_PUT(GM, J)                                                                                          \
    if ((J) < nh) {                                                                                              \
      const long wslot_ = kbase + (long)si * n + (long)(J) * n * n;                                             \
      _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                        \
        const int in_ = 32 * ubi + fmap(e_, kg);                                                                 \
        if (in_ < n && out < n) prow[wslot_ + (long)in_ * n + out] = A.omega * GM[e_];                           \
      }                                                                                                          \
    }
    NIF5_PUT(G0, 0) NIF5_PUT(G1, 1) NIF5_PUT(G2, 2) NIF5_PUT(G3, 3)
#undef NIF5_PUT
  }
  if (has_bunit && lane < 16) {       // rows 0..3 of the 16x16 bias accumulator live on lanes 0..15, element j
    float* prow = F.partial + (long)blockIdx.x * F.pstride;
    const long kb_ = bkb < r ? A.off_Wh + (long)bkb * A.po : A.off_bh;
    const long s_bh = (long)si * n + (long)nh * n * n + (long)n * so + n;
    const int out = 16 * bcb + c16;
    for (int j = 0; j < nh; ++j)
      if (out < n) prow[kb_ + s_bh + (long)j * n + out] = j == 0 ? GBs[0] : (j == 1 ? GBs[1] : (j == 2 ? GBs[2] : GBs[3]));
  }
}

// ---- host side ---------------------------------------------------------------------------------
static size_t snet5_shmem(const SNetArgs& a, int NBL) {
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const size_t ni = (size_t)(((a.si + 3) & ~3) + ((a.r + 3) & ~3) + ((a.so + 3) & ~3) + 4) * 16;
  const size_t pw = 2 * a.r * 64 + 2 * ni;
  const size_t ex = (size_t)8 * (a.r + 2) * (16 * NBL) * 16;
  return (size_t)2 * NBL * 3 * 64 * 16 + (sm_tot + 8 * pw + 8 + ex) * sizeof(float);
}
bool snet5_supported(const SNetArgs& a) {
  const int NBL = snet3_nbl(a.n);
  if (a.ll || a.nif_skip || a.res || a.prec != 0 || a.r < 1) return false;
  if (NBL != 2 && NBL != 4) return false;
  if (a.nh < 1 || a.nh > 4) return false;
  if ((long)(a.nh + 1) * 4 * NBL > 128) return false;                 // sign-bit shift register
  if ((a.r + 1) * (NBL / 2) * (NBL / 2) > 8) return false;             // one gradient block per wave and layer
  if ((a.r + 1) * NBL > 8) return false;                                // one 16-column bias block per wave and layer
  return snet5_shmem(a, NBL) <= 160u * 1024u;
}
long snet5_ring_floats_per_wave(int n, int nh) { return (long)nh * 16 * snet3_nbl(n) * 16; }
int launch_snet5(const SNetArgs& a, float* partial, long pstride, bool query_only, hipStream_t st) {
  const int NBL = snet3_nbl(a.n);
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 7) / 8;
  const int nblk = (int)(ngroups < 256 ? ngroups : 256);
  if (query_only) return nblk;
  S5Args F; F.s = a; F.partial = partial; F.pstride = pstride;
  const size_t shm = snet5_shmem(a, NBL);
  dim3 grid(nblk), block(512);
  if (NBL == 4) {
    (void)hipFuncSetAttribute((const void*)k_snet5<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_snet5<4>), grid, block, shm, st, F);
  } else {
    (void)hipFuncSetAttribute((const void*)k_snet5<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((k_snet5<2>), grid, block, shm, st, F);
  }
  return nblk;
}
