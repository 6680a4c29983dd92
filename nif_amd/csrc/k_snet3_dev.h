// k_snet3_dev.h -- device helpers shared by the 16-point-tile kernels (k_snet3.hip, k_jac.hip)
#pragma once
#include "nif_internal.h"

#ifndef NIF_S3_SETPRIO
#define NIF_S3_SETPRIO 1
#endif

__device__ __forceinline__ float hyp3(const SNetArgs& A, int k, long slot) {
  return k < A.r ? A.theta[A.off_Wh + (long)k * A.po + slot] : A.theta[A.off_bh + slot];
}

// ---- device helpers ---------------------------------------------------------------------------
// T[ob] (+)= sum_ib,v A(plane) x B(hin): 16x16x4 fp32 MFMAs, output blocks rotated innermost so that
// consecutive MFMAs hit independent accumulators
template <int NBL, bool ACCUM>
__device__ __forceinline__ void mfma16(const f32x4* plane, const f32x4 (&hin)[NBL], f32x4 (&T)[NBL], int lane) {
#if NIF_S3_SETPRIO
  __builtin_amdgcn_s_setprio(1);   // MFMA cluster wins VALU-issue arbitration against co-resident waves (+8 %)
#endif
  if (!ACCUM) {
#pragma unroll
    for (int ob = 0; ob < NBL; ++ob) { T[ob][0] = 0.f; T[ob][1] = 0.f; T[ob][2] = 0.f; T[ob][3] = 0.f; }
  }
#pragma unroll
  for (int ib = 0; ib < NBL; ++ib) {
    f32x4 a[NBL];
#pragma unroll
    for (int ob = 0; ob < NBL; ++ob) a[ob] = plane[(ob * NBL + ib) * 64 + lane];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int ob = 0; ob < NBL; ++ob)
#ifdef NIF_ABL_NOMFMA
        T[ob][v] += a[ob][v] * hin[ib][v];
#else
        T[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][v], hin[ib][v], T[ob], 0, 0, 0);
#endif
  }
#if NIF_S3_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// stash [tile32][feature][32]: this wave's 16-point tile is half `hx` of tile32
template <int NBL>
__device__ __forceinline__ void st_store16(float* __restrict__ slot, long row0, const f32x4 (&h)[NBL], int g) {
#ifdef NIF_ABL_NOSTORE
  if (h[0][0] != 12345.678f) return;
#endif
  // row0 = (tile32 * FP) * 32 + 16*half + p   (floats); feature f lives at row0 + f*32.  One base address per PAIR of
  // blocks: the 8 accesses of a pair then sit within the 12-bit immediate offset (r2: hipcc kept 8 extra 64-bit per-feature
  // offsets in 16 VGPRs and spent a v_lshl_add_u64 per access on the second half of the tile)
#pragma unroll
  for (int b = 0; b < NBL; b += 2) {
    float* q = slot + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
    for (int bb = 0; bb < 2 && b + bb < NBL; ++bb)
#pragma unroll
      for (int v = 0; v < 4; ++v) q[(16 * bb + v) * 32] = h[b + bb][v];
  }
}
// the same tile as bf16 rows [tile32][feature][32 points] of 64 B (dL/da under mixed_bfloat16): same element index, half the bytes
template <int NBL>
__device__ __forceinline__ void st_store16_bf(float* __restrict__ slot, long row0, const f32x4 (&h)[NBL], int g) {
#ifdef NIF_ABL_NOSTORE
  if (h[0][0] != 12345.678f) return;
#endif
#pragma unroll
  for (int b = 0; b < NBL; b += 2) {
    __bf16* q = reinterpret_cast<__bf16*>(slot) + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
    for (int bb = 0; bb < 2 && b + bb < NBL; ++bb)
#pragma unroll
      for (int v = 0; v < 4; ++v) q[(16 * bb + v) * 32] = (__bf16)h[b + bb][v];
  }
}
template <int NBL>
__device__ __forceinline__ void st_load16(const float* __restrict__ slot, long row0, f32x4 (&h)[NBL], int g) {
#ifdef NIF_ABL_NOLOAD      // measurement builds: how much of the kernel is the stash round trip (results are wrong)
  if (row0 != -12345) {
#pragma unroll
    for (int b = 0; b < NBL; ++b) { h[b][0] = 0.5f; h[b][1] = 0.25f; h[b][2] = 0.125f; h[b][3] = 0.75f; }
    return;
  }
#endif
#pragma unroll
  for (int b = 0; b < NBL; b += 2) {
    const float* q = slot + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
    for (int bb = 0; bb < 2 && b + bb < NBL; ++bb)
#pragma unroll
      for (int v = 0; v < 4; ++v) h[b + bb][v] = q[(16 * bb + v) * 32];
  }
}

// activation of a tile.  Padded features (>= n) need no masking: their weight rows/columns in the packed
// planes and their entries in the LDS small vectors are zero, so whatever act(0) is never propagates.
template <int NBL, int ACT>
__device__ __forceinline__ void act16_t(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL], int n, int g) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float hv, dv;
      act_eval<ACT>(a[b][v], &hv, &dv);
      h[b][v] = hv; d[b][v] = dv;
    }
}
// |a| >= 2^20 somewhere in the tile (never on a working SIREN; the wave-uniform test costs 8 v_max3 per tile): a ROLLED loop
// over the elements of two blocks at a time (dynamic register indexing with a uniform index), so that the hot kernels carry
// NBL / 2 copies of the fp64 argument reduction instead of 16 x NBL inlined copies per call site (r2: 900 fp64 instructions
// and 32 spilled registers in k_snet4's benchmark instantiation)
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NBL>
__device__ __forceinline__ void sine16_slow(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL]) {
#pragma unroll
  for (int q = 0; q < NBL; q += 2) {
    f32x8 va, vs, vc;
#pragma unroll
    for (int t = 0; t < 8; ++t) { va[t] = q + (t >> 2) < NBL ? a[(q + (t >> 2)) % NBL][t & 3] : 0.f; vs[t] = 0.f; vc[t] = 0.f; }
#pragma nounroll
    for (int i = 0; i < 8; ++i) {
      float sv, cv;
      nif_sincosf(va[i], &sv, &cv);
      vs[i] = sv; vc[i] = cv;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (q + (t >> 2) < NBL) { h[(q + (t >> 2)) % NBL][t & 3] = vs[t]; d[(q + (t >> 2)) % NBL][t & 3] = vc[t]; }
  }
}
template <int NBL>
__device__ __forceinline__ bool sine16_big(const f32x4 (&a)[NBL]) {
  float mx = 0.f;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) mx = fmaxf(mx, fabsf(a[b][v]));
  return __builtin_expect(__any(!(mx < NIF_SINCOS_FAST_LIMIT)), 0);   // wave-uniform, once per tile
}
template <int NBL>
__device__ __forceinline__ void sine16(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL]) {
  if (sine16_big<NBL>(a)) { sine16_slow<NBL>(a, h, d); return; }
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float hv, dv;
      nif_sincosf_core(a[b][v], &hv, &dv);
      h[b][v] = hv; d[b][v] = dv;
    }
}
// SIREN tile, training with the sign-bit cosine: h = sin(a) and d = a float carrying only the SIGN of cos(a)
template <int NBL>
__device__ __forceinline__ void sine16_sign(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL]) {
  if (sine16_big<NBL>(a)) { sine16_slow<NBL>(a, h, d); return; }
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float hv, dv;
      nif_sin_cossign_core(a[b][v], &hv, &dv);
      h[b][v] = hv; d[b][v] = dv;
    }
}
// SIREN tile, training, the cosine's sign TAGGED into the sine (k_snet4): h = sin(a) with the least significant mantissa bit
// replaced by [cos(a) < 0].  cos(a) < 0  <=>  rint(a / pi) is odd, and the parity of rint(a / pi) is the mantissa LSB of
// fma(a, 1/pi, 1.5 * 2^23) (|a| < 2^20 here): one packed fma per two elements and one v_bfi_b32 per element -- instead of
// v_sub + 3 pack instructions + a 128-bit shift register (r2).  The activation moves by at most one ulp (6e-8): the tagged
// value is what the next layer, the stash and the weight-gradient kernels see; near cos(a) = 0, where the parity can
// disagree with the true sign, the cosine rebuilt from it is ~0 anyway.
// |a| >= 2^20 somewhere in the tile (never on a working SIREN): the argument reduction in fp64 -- t = x / 2 pi, f = t - rint(t) -- and
// the same v_sin_f32; cos(2 pi f) < 0 <=> |f| > 1/4 gives the tag.  r5: this replaces the fp64 reduction + cephes kernels of
// sine16_slow in the TAGGED forms (8 instructions per element instead of ~40: the training kernels carry 3 .. 6 copies of it, and
// the ping-pong form of k_snet6 has to fit the instruction cache); same accuracy as the fast path (2.6e-7 abs)
template <int NBL>
__device__ __forceinline__ void sine16_tag_big(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], float inv) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double t = (double)(a[b][v] * inv) * 0.15915494309189535;
      t -= __builtin_rint(t);
      const float f = (float)t;
      const float sv = __builtin_amdgcn_sinf(f);
      h[b][v] = __uint_as_float((__float_as_uint(sv) & ~1u) | (fabsf(f) > 0.25f ? 1u : 0u));
    }
}
template <int NBL>
__device__ __forceinline__ void sine16_tag(const f32x4 (&a)[NBL], f32x4 (&h)[NBL]) {
  if (sine16_big<NBL>(a)) { sine16_tag_big<NBL>(a, h, 1.0f); return; }
  // two elements per instruction (v_pk_fma_f32 / v_pk_add_f32); rint(x / 2pi) by the 1.5 * 2^23 trick (|x / 2pi| < 2^22 here)
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
        unsigned o;
        // (t & 1) | (sv & ~1).  The s_nop is the wait state a non-transcendental VALU read of a v_sin_f32 result needs on gfx950:
        // hipcc inserts it for its own instructions but cannot see into the asm (without it: stale lanes, r3 first attempt)
        asm("s_nop 0\n\tv_bfi_b32 %0, 1, %1, %2" : "=v"(o) : "v"(__float_as_uint(t[e])), "v"(__float_as_uint(sv)));
        h[b][v + e] = __uint_as_float(o);
      }
    }
}
// the same for a = inv x with inv a power of two (k_snet6's half products carry the plane's scale): the scale rides in the
// constants, so that not one instruction is added and the result is bit for bit that of sine16_tag(inv x)
template <int NBL>
__device__ __forceinline__ void sine16_tag_sc(const f32x4 (&x_)[NBL], f32x4 (&h)[NBL], float inv) {
  {
    float mx = 0.f;
#pragma unroll
    for (int b = 0; b < NBL; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) mx = fmaxf(mx, fabsf(x_[b][v]));
    if (__builtin_expect(__any(!(mx * inv < NIF_SINCOS_FAST_LIMIT)), 0)) { sine16_tag_big<NBL>(x_, h, inv); return; }
  }
  const float c0 = 0.15915493667125702f * inv, c1 = 6.420638326565253e-09f * inv, c2 = 0.318309886183790672f * inv;
  const f32x2 C = {c0, c0}, CL = {c1, c1}, M = {12582912.0f, 12582912.0f}, IP = {c2, c2};
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; v += 2) {
      const f32x2 x = {x_[b][v], x_[b][v + 1]};
      const f32x2 k = __builtin_elementwise_fma(x, C, M) - M;
      f32x2 f = __builtin_elementwise_fma(x, C, -k);
      f = __builtin_elementwise_fma(x, CL, f);
      const f32x2 t = __builtin_elementwise_fma(x, IP, M);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float sv = __builtin_amdgcn_sinf(f[e]);
        unsigned o;
        asm("s_nop 0\n\tv_bfi_b32 %0, 1, %1, %2" : "=v"(o) : "v"(__float_as_uint(t[e])), "v"(__float_as_uint(sv)));
        h[b][v + e] = __uint_as_float(o);
      }
    }
}
// ---- 16-bit PHASE stash (r5; mixed_bfloat16, 128-wide nets: k_snet4<8, .., PR = 1> -> k_gw8<R, true, true>) ---------------------------
// Under the policy the stashed layer input h = sin(a) has two readers: the adjoint sweep (cos(a), and h itself for the dL/dz dot
// products) and the weight-gradient kernel, whose operand is bf16(h).  Both can be rebuilt from the reduced argument: the stash row
// holds q = rint(65536 f) as int16, f = a / 2 pi - rint(a / 2 pi) in [-1/2, 1/2] (q = +32768 wraps to -32768: the same angle), and a
// reader takes sin / cos of q / 65536 revolutions on v_sin_f32 / v_cos_f32.  |delta a| <= 2 pi 2^-17 = 4.8e-5, i.e. the rebuilt
// values are within 4.8e-5 of the exact ones where the policy's own operand rounding is 2^-9 relative -- for HALF the bytes of the
// fp32 row on the store and on both reads (the stash traffic IS the 128-wide kernels' time: DESIGN 5.4).
// sine16_tag with the phases on the side: ph[2 b + (v >> 1)] = q of element (b, v) in its low (v even) / high (v odd) half
template <int NBL>
__device__ __forceinline__ void sine16_tag_ph(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], unsigned (&ph)[2 * NBL]) {
  const f32x2 M = {12582912.0f, 12582912.0f}, Q = {65536.0f, 65536.0f};
  if (sine16_big<NBL>(a)) {
#pragma unroll
    for (int b = 0; b < NBL; ++b)
#pragma unroll
      for (int v = 0; v < 4; v += 2) {
        f32x2 f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          double t = (double)a[b][v + e] * 0.15915494309189535;
          t -= __builtin_rint(t);
          f[e] = (float)t;
          const float sv = __builtin_amdgcn_sinf(f[e]);
          h[b][v + e] = __uint_as_float((__float_as_uint(sv) & ~1u) | (fabsf(f[e]) > 0.25f ? 1u : 0u));
        }
        const f32x2 q = __builtin_elementwise_fma(f, Q, M);      // the mantissa's low 16 bits = rint(65536 f) mod 2^16
        ph[2 * b + (v >> 1)] = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), 0x05040100u);
      }
    return;
  }
  const f32x2 C = {0.15915493667125702f, 0.15915493667125702f}, CL = {6.420638326565253e-09f, 6.420638326565253e-09f};
  const f32x2 IP = {0.318309886183790672f, 0.318309886183790672f};
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; v += 2) {
      const f32x2 x = {a[b][v], a[b][v + 1]};
      const f32x2 k = __builtin_elementwise_fma(x, C, M) - M;
      f32x2 f = __builtin_elementwise_fma(x, C, -k);
      f = __builtin_elementwise_fma(x, CL, f);
      const f32x2 t = __builtin_elementwise_fma(x, IP, M);
      const f32x2 q = __builtin_elementwise_fma(f, Q, M);
      ph[2 * b + (v >> 1)] = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), 0x05040100u);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float sv = __builtin_amdgcn_sinf(f[e]);
        unsigned o;
        asm("s_nop 0\n\tv_bfi_b32 %0, 1, %1, %2" : "=v"(o) : "v"(__float_as_uint(t[e])), "v"(__float_as_uint(sv)));
        h[b][v + e] = __uint_as_float(o);
      }
    }
}
// the tile's phases as int16 rows [tile32][feature][32 points] of 64 B (element index as st_store16)
template <int NBL>
__device__ __forceinline__ void st_store16_ph(float* __restrict__ slot, long row0, const unsigned (&ph)[2 * NBL], int g) {
#ifdef NIF_ABL_NOSTORE
  if (ph[0] != 0x12345678u) return;
#endif
#pragma unroll
  for (int b = 0; b < NBL; b += 2) {
    unsigned short* q = reinterpret_cast<unsigned short*>(slot) + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
    for (int bb = 0; bb < 2 && b + bb < NBL; ++bb)
#pragma unroll
      for (int v = 0; v < 4; ++v) q[(16 * bb + v) * 32] = (unsigned short)(ph[2 * (b + bb) + (v >> 1)] >> (16 * (v & 1)));
  }
}
// ... read back as phases in REVOLUTIONS (f = q / 65536): ph_sin / ph_cos rebuild sin(a) / cos(a)
template <int NBL>
__device__ __forceinline__ void st_load16_ph(const float* __restrict__ slot, long row0, f32x4 (&f)[NBL], int g) {
#ifdef NIF_ABL_NOLOAD
  if (row0 != -12345) {
#pragma unroll
    for (int b = 0; b < NBL; ++b) { f[b][0] = 0.05f; f[b][1] = 0.25f; f[b][2] = 0.125f; f[b][3] = -0.075f; }
    return;
  }
#endif
#pragma unroll
  for (int b = 0; b < NBL; b += 2) {
    const short* q = reinterpret_cast<const short*>(slot) + (row0 + (long)(16 * b + 4 * g) * 32);
#pragma unroll
    for (int bb = 0; bb < 2 && b + bb < NBL; ++bb)
#pragma unroll
      for (int v = 0; v < 4; ++v) f[b + bb][v] = (float)(int)q[(16 * bb + v) * 32] * (1.0f / 65536.0f);
  }
}
template <int NBL>
__device__ __forceinline__ void ph_cos(const f32x4 (&f)[NBL], f32x4 (&d)[NBL]) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) d[b][v] = __builtin_amdgcn_cosf(f[b][v]);
}
// cos(a) from the tagged sine: sqrt(1 - s^2) with the sign from the tag bit
template <int NBL>
__device__ __forceinline__ void tag_cos(const f32x4 (&sn)[NBL], f32x4 (&d)[NBL]) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float s = sn[b][v];
      const float c = __builtin_amdgcn_sqrtf(__builtin_amdgcn_fmed3f(fmaf(-s, s, 1.0f), 0.0f, 1.0f));   // v_sqrt_f32 (1 ulp)
      d[b][v] = __uint_as_float(__float_as_uint(c) | (__float_as_uint(s) << 31));
    }
}
template <int NBL, int ACT>
__device__ __forceinline__ void act16(int act, const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL], int n, int g) {
#ifdef NIF_ABL_NOACT
  _Pragma("unroll") for (int b = 0; b < NBL; ++b) { f32x4 t = a[b]; h[b] = t * 0.5f; d[b] = t + 1.0f; }
  return;
#endif
  if (ACT == ACT_SINE) { sine16<NBL>(a, h, d); return; }
  switch (act) {
    case ACT_SINE: sine16<NBL>(a, h, d); break;
    case ACT_SWISH: act16_t<NBL, ACT_SWISH>(a, h, d, n, g); break;
    case ACT_TANH: act16_t<NBL, ACT_TANH>(a, h, d, n, g); break;
    case ACT_RELU: act16_t<NBL, ACT_RELU>(a, h, d, n, g); break;
    case ACT_SIGMOID: act16_t<NBL, ACT_SIGMOID>(a, h, d, n, g); break;
    case ACT_ELU: act16_t<NBL, ACT_ELU>(a, h, d, n, g); break;
    case ACT_SOFTPLUS: act16_t<NBL, ACT_SOFTPLUS>(a, h, d, n, g); break;
    case ACT_GELU: act16_t<NBL, ACT_GELU>(a, h, d, n, g); break;
    case ACT_SELU: act16_t<NBL, ACT_SELU>(a, h, d, n, g); break;
    case ACT_SOFTSIGN: act16_t<NBL, ACT_SOFTSIGN>(a, h, d, n, g); break;
    case ACT_EXPONENTIAL: act16_t<NBL, ACT_EXPONENTIAL>(a, h, d, n, g); break;
    case ACT_HARD_SIGMOID: act16_t<NBL, ACT_HARD_SIGMOID>(a, h, d, n, g); break;
    default: act16_t<NBL, ACT_LINEAR>(a, h, d, n, g); break;
  }
}

// second derivative of the activation (HessianLayer): f''(a)
template <int ACT>
__device__ __forceinline__ float act_d2(int act, float a) {
  const int id = ACT >= 0 ? ACT : act;
  switch (id) {
    case ACT_SINE: { float s, c; nif_sincosf(a, &s, &c); return -s; }
    case ACT_SWISH: { const float s = 1.0f / (1.0f + expf(-a)); return s * (1.0f - s) * (2.0f + a * (1.0f - 2.0f * s)); }
    case ACT_TANH: { const float t = tanhf(a); return -2.0f * t * (1.0f - t * t); }
    case ACT_SIGMOID: { const float s = 1.0f / (1.0f + expf(-a)); return s * (1.0f - s) * (1.0f - 2.0f * s); }
    case ACT_ELU: return a > 0.f ? 0.f : expf(a);
    case ACT_SOFTPLUS: { const float s = 1.0f / (1.0f + expf(-a)); return s * (1.0f - s); }
    case ACT_GELU: return 0.3989422804014327f * expf(-0.5f * a * a) * (2.0f - a * a);
    case ACT_SELU: return a > 0.f ? 0.f : NIF_SELU_SCALE * NIF_SELU_ALPHA * expf(a);
    case ACT_SOFTSIGN: { const float q = 1.0f / (1.0f + fabsf(a)); return (a > 0.f ? -2.0f : (a < 0.f ? 2.0f : 0.f)) * q * q * q; }
    case ACT_EXPONENTIAL: return expf(a);
    default: return 0.f;   // linear, relu, hard_sigmoid
  }
}


// s_setprio around the MFMA clusters of mfma_x6 / mfma_x3 (a translation unit that manages the priority itself defines the two away)
#ifndef NIF_MFMA_PRIO_ON
#define NIF_MFMA_PRIO_ON __builtin_amdgcn_s_setprio(1);
#define NIF_MFMA_PRIO_OFF __builtin_amdgcn_s_setprio(0);
#endif
// ---- fp32 products as exact bf16 splits (k_snet4.hip has the derivation and the measured accuracy) ---------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NBL>
__device__ __forceinline__ void split3(const f32x4 (&h)[NBL], bf16x8 (&s0)[NBL / 2], bf16x8 (&s1)[NBL / 2], bf16x8 (&s2)[NBL / 2]) {
#pragma unroll
  for (int ks = 0; ks < NBL / 2; ++ks)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x = h[2 * ks + (t >> 2)][t & 3];
      const __bf16 x0 = (__bf16)x;
      const float r1 = x - (float)x0;
      const __bf16 x1 = (__bf16)r1;
      s0[ks][t] = x0; s1[ks][t] = x1; s2[ks][t] = (__bf16)(r1 - (float)x1);
    }
}
template <int NBL>
__device__ __forceinline__ void split2(const f32x4 (&h)[NBL], bf16x8 (&s0)[NBL / 2], bf16x8 (&s1)[NBL / 2]) {
#pragma unroll
  for (int ks = 0; ks < NBL / 2; ++ks)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x = h[2 * ks + (t >> 2)][t & 3];
      const __bf16 x0 = (__bf16)x;
      s0[ks][t] = x0; s1[ks][t] = (__bf16)(x - (float)x0);
    }
}
// mixed_float16 (PR == 2, r4): ONE half-precision operand per value (RNE, saturated at the largest finite half: no infinity can
// enter a product), carried in the bf16x8 register type of the split planes; `scale` = the loss scale of the data adjoint (a
// power of two: exact), 1 in the forward sweep
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NBL>
__device__ __forceinline__ void cast_f16(const f32x4 (&h)[NBL], bf16x8 (&s0)[NBL / 2], float scale) {
#pragma unroll
  for (int ks = 0; ks < NBL / 2; ++ks) {
    f16x8 q;
#pragma unroll
    for (int t = 0; t < 8; ++t) q[t] = (_Float16)__builtin_amdgcn_fmed3f(scale * h[2 * ks + (t >> 2)][t & 3], -65504.0f, 65504.0f);
    s0[ks] = __builtin_bit_cast(bf16x8, q);
  }
}
// the exact-product HALF form (r5, k_snet6): x = hi + lo, hi = half(scale x), lo = half(scale x - hi); `scale` a power of two that
// brings the tile into half's range (4096 for sines, the per-point loss scale for dL/da) -- 11 + 11 significand bits, so that
// hi.hi + hi.lo + lo.hi of two such pairs is an fp32 product (tools/exp/f16_split_mfma.hip)
template <int NBL>
__device__ __forceinline__ void split2h(const f32x4 (&h)[NBL], float scale, bf16x8 (&s0)[NBL / 2], bf16x8 (&s1)[NBL / 2]) {
#pragma unroll
  for (int ks = 0; ks < NBL / 2; ++ks) {
    f16x8 q0, q1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x = scale * h[2 * ks + (t >> 2)][t & 3];
      const _Float16 x0 = (_Float16)x;
      q0[t] = x0; q1[t] = (_Float16)(x - (float)x0);
    }
    s0[ks] = __builtin_bit_cast(bf16x8, q0); s1[ks] = __builtin_bit_cast(bf16x8, q1);
  }
}
// PR-aware operand forms of a layer's activation / dL/da tile (the exact splits unless a policy asks for one rounded operand)
template <int NBL, int PR>
__device__ __forceinline__ void split3p(const f32x4 (&h)[NBL], bf16x8 (&s0)[NBL / 2], bf16x8 (&s1)[NBL / 2], bf16x8 (&s2)[NBL / 2]) {
  if (PR == 2) cast_f16<NBL>(h, s0, 1.0f);
  else split3<NBL>(h, s0, s1, s2);
}
template <int NBL, int PR>
__device__ __forceinline__ void split2p(const f32x4 (&h)[NBL], bf16x8 (&s0)[NBL / 2], bf16x8 (&s1)[NBL / 2], float scale) {
  if (PR == 2) cast_f16<NBL>(h, s0, scale);
  else split2<NBL>(h, s0, s1);
}
__device__ __forceinline__ f32x4 mfma_f16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// one K-step chunk of a forward plane: T[ob] (+)= sum over the 32 features of the chunk, 6-product fp32-exact form.
// Two output blocks at a time: their 6-MFMA chains interleave (a dependent v_mfma_f32_16x16x32_bf16 cannot issue
// back to back) and one LDS round trip feeds 12 MFMAs
// PR = 1 (the mixed_bfloat16 policy of the build): operands rounded to bf16, ONE product a0*b0 instead of the exact split
// PR = 2 (mixed_float16, r4): the same with half-precision operands (slot 0 of the chunk holds the f16 plane: k_pack16b)
// ZI: the chains start from zero (the first MFMA of every chain takes the inline constant 0 as C: no v_mov zeroing)
// NT / OB0: the chunk holds NBL of the NT output blocks of T, starting at block OB0 (128-wide nets stream half chunks)
// CP (late r4): the chunk holds ONE 16-bit plane per block (the policies' compact plane set: unit ob * 64 + lane) instead of the split groups
template <int NBL, int PR = 0, bool ZI = false, int NT = NBL, int OB0 = 0, bool CP = false>
__device__ __forceinline__ void mfma_x6(const bf16x8* cur, const bf16x8 b0, const bf16x8 b1, const bf16x8 b2, f32x4 (&T_)[NT], int lane) {
  NIF_MFMA_PRIO_ON
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4* T = T_ + OB0;
#pragma unroll
  for (int ob = 0; ob < NBL; ob += 2) {
    if (PR == 2) {
      const bf16x8 a0 = cur[(CP ? ob : ob * 3) * 64 + lane], c0 = cur[(CP ? ob + 1 : ob * 3 + 3) * 64 + lane];
      T[ob] = mfma_f16(a0, b0, ZI ? z4 : T[ob]);
      T[ob + 1] = mfma_f16(c0, b0, ZI ? z4 : T[ob + 1]);
      continue;
    }
    if (PR) {
      const bf16x8 a0 = cur[(CP ? ob : ob * 3) * 64 + lane], c0 = cur[(CP ? ob + 1 : ob * 3 + 3) * 64 + lane];
      T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, ZI ? z4 : T[ob], 0, 0, 0);
      T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0, ZI ? z4 : T[ob + 1], 0, 0, 0);
      continue;
    }
    const bf16x8 a0 = cur[(ob * 3 + 0) * 64 + lane], a1 = cur[(ob * 3 + 1) * 64 + lane], a2 = cur[(ob * 3 + 2) * 64 + lane];
    const bf16x8 c0 = cur[(ob * 3 + 3) * 64 + lane], c1 = cur[(ob * 3 + 4) * 64 + lane], c2 = cur[(ob * 3 + 5) * 64 + lane];
    T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, ZI ? z4 : T[ob], 0, 0, 0);
    T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, b1, ZI ? z4 : T[ob + 1], 0, 0, 0);
    T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, T[ob], 0, 0, 0);
    T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b2, T[ob + 1], 0, 0, 0);
    T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, T[ob], 0, 0, 0);
    T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c2, b0, T[ob + 1], 0, 0, 0);
    T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, T[ob], 0, 0, 0);
    T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b1, T[ob + 1], 0, 0, 0);
    T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, T[ob], 0, 0, 0);
    T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, b0, T[ob + 1], 0, 0, 0);
    T[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, T[ob], 0, 0, 0);
    T[ob + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0, T[ob + 1], 0, 0, 0);
  }
  NIF_MFMA_PRIO_OFF
}
// one K-step chunk of an adjoint plane, 3-product form, two blocks' chains interleaved
#ifndef NIF_X16_LOLO
#define NIF_X16_LOLO 0
#endif
template <int NBL, int PR = 0, bool ZI = false, int NT = NBL, int OB0 = 0, bool CP = false>
__device__ __forceinline__ void mfma_x3(const bf16x8* cur, const bf16x8 b0, const bf16x8 b1, f32x4 (&T_)[NT], int lane) {
  NIF_MFMA_PRIO_ON
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4* T = T_ + OB0;
#pragma unroll
  for (int ib = 0; ib < NBL; ib += 2) {
    if (PR == 2) {
      const bf16x8 a0 = cur[(CP ? ib : ib * 2) * 64 + lane], c0 = cur[(CP ? ib + 1 : ib * 2 + 2) * 64 + lane];
      T[ib] = mfma_f16(a0, b0, ZI ? z4 : T[ib]);
      T[ib + 1] = mfma_f16(c0, b0, ZI ? z4 : T[ib + 1]);
      continue;
    }
    if (PR == 1) {
      const bf16x8 a0 = cur[(CP ? ib : ib * 2) * 64 + lane], c0 = cur[(CP ? ib + 1 : ib * 2 + 2) * 64 + lane];
      T[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, ZI ? z4 : T[ib], 0, 0, 0);
      T[ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0, ZI ? z4 : T[ib + 1], 0, 0, 0);
      continue;
    }
    const bf16x8 a0 = cur[(ib * 2 + 0) * 64 + lane], a1 = cur[(ib * 2 + 1) * 64 + lane];
    const bf16x8 c0 = cur[(ib * 2 + 2) * 64 + lane], c1 = cur[(ib * 2 + 3) * 64 + lane];
    if (PR == 3) {      // exact-product half planes (hi, lo) x half operand pair (b0 = hi, b1 = lo): three products, small terms first
#if NIF_X16_LOLO      // measurement builds (r6): the lo.lo term as a fourth product (2^-24 of the product; DESIGN 7: what it buys the gradient)
      T[ib] = mfma_f16(a1, b1, ZI ? z4 : T[ib]);
      T[ib + 1] = mfma_f16(c1, b1, ZI ? z4 : T[ib + 1]);
      T[ib] = mfma_f16(a0, b1, T[ib]);
      T[ib + 1] = mfma_f16(c0, b1, T[ib + 1]);
#else
      T[ib] = mfma_f16(a0, b1, ZI ? z4 : T[ib]);
      T[ib + 1] = mfma_f16(c0, b1, ZI ? z4 : T[ib + 1]);
#endif
      T[ib] = mfma_f16(a1, b0, T[ib]);
      T[ib + 1] = mfma_f16(c1, b0, T[ib + 1]);
      T[ib] = mfma_f16(a0, b0, T[ib]);
      T[ib + 1] = mfma_f16(c0, b0, T[ib + 1]);
      continue;
    }
    T[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, ZI ? z4 : T[ib], 0, 0, 0);
    T[ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b1, ZI ? z4 : T[ib + 1], 0, 0, 0);
    T[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, T[ib], 0, 0, 0);
    T[ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, b0, T[ib + 1], 0, 0, 0);
    T[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, T[ib], 0, 0, 0);
    T[ib + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, b0, T[ib + 1], 0, 0, 0);
  }
  NIF_MFMA_PRIO_OFF
}

// the PR = 3 form with the first block pair's A operands already in registers (k_snet6, r6: read in front of the step's DMA issue)
template <int NBL, bool ZI>
__device__ __forceinline__ void mfma_x3_pre(const bf16x8* cur, const bf16x8 (&pa)[4], const bf16x8 b0, const bf16x8 b1, f32x4 (&T)[NBL], int lane) {
  NIF_MFMA_PRIO_ON
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ib = 0; ib < NBL; ib += 2) {
    const bf16x8 a0 = ib == 0 ? pa[0] : cur[(ib * 2 + 0) * 64 + lane], a1 = ib == 0 ? pa[1] : cur[(ib * 2 + 1) * 64 + lane];
    const bf16x8 c0 = ib == 0 ? pa[2] : cur[(ib * 2 + 2) * 64 + lane], c1 = ib == 0 ? pa[3] : cur[(ib * 2 + 3) * 64 + lane];
    T[ib] = mfma_f16(a0, b1, ZI ? z4 : T[ib]);
    T[ib + 1] = mfma_f16(c0, b1, ZI ? z4 : T[ib + 1]);
    T[ib] = mfma_f16(a1, b0, T[ib]);
    T[ib + 1] = mfma_f16(c1, b0, T[ib + 1]);
    T[ib] = mfma_f16(a0, b0, T[ib]);
    T[ib + 1] = mfma_f16(c0, b0, T[ib + 1]);
  }
  NIF_MFMA_PRIO_OFF
}

// ... and the software-pipelined form (k_snet6, NIF_S6_PF): the first block pair's operands of THIS chunk come from registers when the
// previous step prefetched them (USEPF), and the first pair of the NEXT chunk (already landed: three chunk buffers, DMA two steps
// ahead) is read behind this chunk's first six products (MAKEPF) -- the LDS latency of a step's first reads leaves the critical path
template <int NBL, bool ZI, bool USEPF, bool MAKEPF>
__device__ __forceinline__ void mfma_x3_pf(const bf16x8* cur, const bf16x8* nxt, bf16x8 (&pf)[4], const bf16x8 b0, const bf16x8 b1,
                                           f32x4 (&T)[NBL], int lane) {
  static_assert(NBL == 4, "two block pairs per chunk");
  NIF_MFMA_PRIO_ON
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const bf16x8 a0 = USEPF ? pf[0] : cur[0 * 64 + lane], a1 = USEPF ? pf[1] : cur[1 * 64 + lane];
  const bf16x8 c0 = USEPF ? pf[2] : cur[2 * 64 + lane], c1 = USEPF ? pf[3] : cur[3 * 64 + lane];
  const bf16x8 d0 = cur[4 * 64 + lane], d1 = cur[5 * 64 + lane], e0 = cur[6 * 64 + lane], e1 = cur[7 * 64 + lane];
  T[0] = mfma_f16(a0, b1, ZI ? z4 : T[0]);
  T[1] = mfma_f16(c0, b1, ZI ? z4 : T[1]);
  T[0] = mfma_f16(a1, b0, T[0]);
  T[1] = mfma_f16(c1, b0, T[1]);
  T[0] = mfma_f16(a0, b0, T[0]);
  T[1] = mfma_f16(c0, b0, T[1]);
  if (MAKEPF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = nxt[i * 64 + lane];
  }
  T[2] = mfma_f16(d0, b1, ZI ? z4 : T[2]);
  T[3] = mfma_f16(e0, b1, ZI ? z4 : T[3]);
  T[2] = mfma_f16(d1, b0, T[2]);
  T[3] = mfma_f16(e1, b0, T[3]);
  T[2] = mfma_f16(d0, b0, T[2]);
  T[3] = mfma_f16(e0, b0, T[3]);
  NIF_MFMA_PRIO_OFF
}

// ... and a whole PLANE of the exact-product form in one step (k_snet6, NIF_S6_BIGCHUNK: the 16 KB chunk = K-step halves at cur and cur + CFU
// units): four operand groups (K step, block pair); the reads of group g + 2 are issued BEHIND the products of group g, so that the
// step exposes ONE LDS round trip instead of one per group (r6 timeline: ~1 200 ticks of `mfma` phase for 384 matrix cycles)
template <int NBL, bool ZI, int CFU>
__device__ __forceinline__ void mfma_x3_plane(const bf16x8* cur, const bf16x8 (&b0)[2], const bf16x8 (&b1)[2], f32x4 (&T)[NBL], int lane) {
  static_assert(NBL == 4, "two block pairs per K step");
  NIF_MFMA_PRIO_ON
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const bf16x8* c0_ = cur + lane;
  bf16x8 p0 = c0_[0 * 64], p1 = c0_[1 * 64], p2 = c0_[2 * 64], p3 = c0_[3 * 64];                   // (K step 0, blocks 0 1)
  bf16x8 q0 = c0_[4 * 64], q1 = c0_[5 * 64], q2 = c0_[6 * 64], q3 = c0_[7 * 64];                   // (K step 0, blocks 2 3)
  __builtin_amdgcn_sched_barrier(0);
  T[0] = mfma_f16(p0, b1[0], ZI ? z4 : T[0]);
  T[1] = mfma_f16(p2, b1[0], ZI ? z4 : T[1]);
  T[0] = mfma_f16(p1, b0[0], T[0]);
  T[1] = mfma_f16(p3, b0[0], T[1]);
  T[0] = mfma_f16(p0, b0[0], T[0]);
  T[1] = mfma_f16(p2, b0[0], T[1]);
  __builtin_amdgcn_sched_barrier(0);
  const bf16x8* c1_ = cur + CFU + lane;
  p0 = c1_[0 * 64]; p1 = c1_[1 * 64]; p2 = c1_[2 * 64]; p3 = c1_[3 * 64];                          // (K step 1, blocks 0 1)
  __builtin_amdgcn_sched_barrier(0);
  T[2] = mfma_f16(q0, b1[0], ZI ? z4 : T[2]);
  T[3] = mfma_f16(q2, b1[0], ZI ? z4 : T[3]);
  T[2] = mfma_f16(q1, b0[0], T[2]);
  T[3] = mfma_f16(q3, b0[0], T[3]);
  T[2] = mfma_f16(q0, b0[0], T[2]);
  T[3] = mfma_f16(q2, b0[0], T[3]);
  __builtin_amdgcn_sched_barrier(0);
  q0 = c1_[4 * 64]; q1 = c1_[5 * 64]; q2 = c1_[6 * 64]; q3 = c1_[7 * 64];                          // (K step 1, blocks 2 3)
  __builtin_amdgcn_sched_barrier(0);
  T[0] = mfma_f16(p0, b1[1], T[0]);
  T[1] = mfma_f16(p2, b1[1], T[1]);
  T[0] = mfma_f16(p1, b0[1], T[0]);
  T[1] = mfma_f16(p3, b0[1], T[1]);
  T[0] = mfma_f16(p0, b0[1], T[0]);
  T[1] = mfma_f16(p2, b0[1], T[1]);
  T[2] = mfma_f16(q0, b1[1], T[2]);
  T[3] = mfma_f16(q2, b1[1], T[3]);
  T[2] = mfma_f16(q1, b0[1], T[2]);
  T[3] = mfma_f16(q3, b0[1], T[3]);
  T[2] = mfma_f16(q0, b0[1], T[2]);
  T[3] = mfma_f16(q2, b0[1], T[3]);
  NIF_MFMA_PRIO_OFF
}

// ---- sign-of-cosine shift register (plain SIREN: cos(a) = +-sqrt(1 - sin^2(a)), sin(a) is the next layer's stashed
// input; see k_snet4.hip) ------------------------------------------------------------------------------------
__device__ __forceinline__ void sgn_push(unsigned long long& lo, unsigned long long& hi, unsigned bits, int w) {
  hi = (hi << w) | (lo >> (64 - w));
  lo = (lo << w) | bits;
}
__device__ __forceinline__ unsigned sgn_pop(unsigned long long& lo, unsigned long long& hi, int w) {
  const unsigned bits = (unsigned)(lo & ((1ull << w) - 1ull));
  lo = (lo >> w) | (hi << (64 - w));
  hi >>= w;
  return bits;
}
template <int NBL>
__device__ __forceinline__ unsigned sgn_pack(const f32x4 (&d)[NBL]) {
  unsigned bits = 0;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) bits |= (__float_as_uint(d[b][v]) >> 31) << (4 * b + v);
  return bits;
}
// cos(a) from sin(a) and the sign bit
template <int NBL>
__device__ __forceinline__ void sgn_cos(const f32x4 (&sn)[NBL], unsigned bits, f32x4 (&d)[NBL]) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float c = __builtin_amdgcn_sqrtf(fmaxf(fmaf(-sn[b][v], sn[b][v], 1.0f), 0.0f));   // v_sqrt_f32 (1 ulp), not the IEEE fix-up sequence
      d[b][v] = __uint_as_float(__float_as_uint(c) | (((bits >> (4 * b + v)) & 1u) << 31));
    }
}
