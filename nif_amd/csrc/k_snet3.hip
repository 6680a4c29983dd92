// k_snet3.hip -- the dominant kernel of the training step on gfx950: hypernetwork ShapeNet forward,
// fused MSE and the data adjoint, persistent workgroups, 16-point tiles on v_mfma_f32_16x16x4_f32.
//
// Why 16-point tiles: with 32-point tiles (32x32x2 MFMA) one activation tile of a 64-wide ShapeNet is
// 32 VGPRs and the fused forward+adjoint needs ~6 of them live -> 1 wave/SIMD or heavy spilling.  With
// 16x16x4 MFMAs the same tile is 16 VGPRs (4 per 16-feature block), the kernel fits 2-3 waves/SIMD and
// the matrix pipe rate is unchanged (64 FLOP/clk/SIMD; the 40-cycle dependent latency is covered by
// rotating over the >= 2 independent output blocks).
//
// Register layout of an activation tile of width 16*NBL (lane = (p, g): p = lane & 15 the point,
// g = lane >> 4): element v of block b is feature 16*b + 4*g + v of point p -- the 16x16 MFMA C/D
// layout with features on rows and points on columns, and (with the K order folded into the packed
// weights) directly the B operand of the next layer.
//
//   * weight planes (one per hidden hyper-matrix j and k in 0..r, NBL^2 KiB each) stream L2 -> LDS,
//     double buffered, one barrier per plane, shared by all waves of the workgroup;
//   * small hyper-vectors ((r+1) x nsm floats) are copied to LDS once per workgroup;
//   * act'(a) goes to a per-wave ring in register-dump order (one 1-KiB store per block, stays in
//     L2 / Infinity Cache); layer inputs h and dL/da go to the [tile32][feature][32] stashes that the
//     weight-gradient GEMMs read.
//
// Semantics: NIF._call_shape_net nif/model.py:233-324, NIFMultiScale._call_shape_net_mres :738-954,
// Keras 'mse' (README.md:33), adjoint per SURVEY a-10.
#include "nif_internal.h"

#ifndef NIF_S3_OCC4
#define NIF_S3_OCC4 2   // waves/SIMD requested for the 64-wide (NBL = 4) instantiation
#endif

__device__ __forceinline__ float hyp3(const SNetArgs& A, int k, long slot) {
  return k < A.r ? A.theta[A.off_Wh + (long)k * A.po + slot] : A.theta[A.off_bh + slot];
}

// ---- packing for the 16x16x4 path -------------------------------------------------------------
//   fwd plane: ((ob*NBL + ib)*64 + lane)*4 + v : M[in = 16ib + 4(lane>>4) + v][out = 16ob + (lane&15)]
//   bwd plane: ((ib*NBL + ob)*64 + lane)*4 + v : M[in = 16ib + (lane&15)][out = 16ob + 4(lane>>4) + v]
__global__ void k_pack16(const float* __restrict__ theta, MatRef m, int NBL, float* __restrict__ WF,
                         float* __restrict__ WB) {
  const long per_plane = (long)NBL * NBL * 256;
  const long total = per_plane * (m.r + 1);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = (int)(idx / per_plane);
    const long rem = idx - (long)k * per_plane;
    const int v = rem & 3, lane = (rem >> 2) & 63, blk = (int)(rem >> 8);
    {
      const int ob = blk / NBL, ib = blk % NBL;
      const int in = 16 * ib + 4 * (lane >> 4) + v, out = 16 * ob + (lane & 15);
      WF[idx] = (in < m.nin && out < m.nout) ? theta[matref_index(m, k, in, out)] : 0.f;
    }
    {
      const int ib = blk / NBL, ob = blk % NBL;
      const int in = 16 * ib + (lane & 15), out = 16 * ob + 4 * (lane >> 4) + v;
      WB[idx] = (in < m.nin && out < m.nout) ? theta[matref_index(m, k, in, out)] : 0.f;
    }
  }
}
void launch_pack16(const float* theta, const MatRef& m, int NBL, f32x4* WF, f32x4* WB, hipStream_t st) {
  const long total = (long)NBL * NBL * 256 * (m.r + 1);
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_pack16, dim3(grid), dim3(256), 0, st, theta, m, NBL, (float*)WF, (float*)WB);
}

// ---- device helpers ---------------------------------------------------------------------------
// T[ob] (+)= sum_ib,v A(plane) x B(hin): 16x16x4 fp32 MFMAs, output blocks rotated innermost so that
// consecutive MFMAs hit independent accumulators
template <int NBL, bool ACCUM>
__device__ __forceinline__ void mfma16(const f32x4* plane, const f32x4 (&hin)[NBL], f32x4 (&T)[NBL], int lane) {
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
        T[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][v], hin[ib][v], T[ob], 0, 0, 0);
  }
}

// stash [tile32][feature][32]: this wave's 16-point tile is half `hx` of tile32
template <int NBL>
__device__ __forceinline__ void st_store16(float* __restrict__ slot, long row0, const f32x4 (&h)[NBL], int g) {
  // row0 = (tile32 * FP) * 32 + 16*half + p   (floats); feature f lives at row0 + f*32
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) slot[row0 + (long)(16 * b + 4 * g + v) * 32] = h[b][v];
}
template <int NBL>
__device__ __forceinline__ void st_load16(const float* __restrict__ slot, long row0, f32x4 (&h)[NBL], int g) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) h[b][v] = slot[row0 + (long)(16 * b + 4 * g + v) * 32];
}

// activation of a tile (features >= n forced to h = 0, d = 0)
template <int NBL, int ACT>
__device__ __forceinline__ void act16_t(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL], int n, int g) {
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float hv, dv;
      act_eval<ACT>(a[b][v], &hv, &dv);
      const bool ok = (16 * b + 4 * g + v) < n;
      h[b][v] = ok ? hv : 0.f;
      d[b][v] = ok ? dv : 0.f;
    }
}
template <int NBL>
__device__ __forceinline__ void sine16(const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL], int n, int g) {
  float mx = 0.f;
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) mx = fmaxf(mx, fabsf(a[b][v]));
  const bool big = __any(!(mx < NIF_SINCOS_FAST_LIMIT));
#pragma unroll
  for (int b = 0; b < NBL; ++b)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float hv, dv;
      if (__builtin_expect(big, 0)) nif_sincosf_big(a[b][v], &hv, &dv);
      else nif_sincosf_core(a[b][v], &hv, &dv);
      const bool ok = (16 * b + 4 * g + v) < n;
      h[b][v] = ok ? hv : 0.f;
      d[b][v] = ok ? dv : 0.f;
    }
}
template <int NBL, int ACT>
__device__ __forceinline__ void act16(int act, const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&d)[NBL], int n, int g) {
  if (ACT == ACT_SINE) { sine16<NBL>(a, h, d, n, g); return; }
  switch (act) {
    case ACT_SINE: sine16<NBL>(a, h, d, n, g); break;
    case ACT_SWISH: act16_t<NBL, ACT_SWISH>(a, h, d, n, g); break;
    case ACT_TANH: act16_t<NBL, ACT_TANH>(a, h, d, n, g); break;
    case ACT_RELU: act16_t<NBL, ACT_RELU>(a, h, d, n, g); break;
    case ACT_SIGMOID: act16_t<NBL, ACT_SIGMOID>(a, h, d, n, g); break;
    case ACT_ELU: act16_t<NBL, ACT_ELU>(a, h, d, n, g); break;
    case ACT_SOFTPLUS: act16_t<NBL, ACT_SOFTPLUS>(a, h, d, n, g); break;
    case ACT_GELU: act16_t<NBL, ACT_GELU>(a, h, d, n, g); break;
    default: act16_t<NBL, ACT_LINEAR>(a, h, d, n, g); break;
  }
}

// MODE: 0 = plain (NIFMultiScale without resblock), 1 = SIREN resblock, 2 = NIF skip connection
template <int NBL, int WAVES, bool TRAIN, int ACT, int MODE>
__global__ __launch_bounds__(WAVES * 64, (NBL <= 2 ? 4 : (NBL == 3 ? 3 : (NBL == 4 ? NIF_S3_OCC4 : 1)))) void k_snet3(SNetArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = WAVES * 64;
  constexpr int PLANE = NBL * NBL * 256;                  // floats per weight plane
  constexpr int PF4 = (PLANE / 4 + NT - 1) / NT;          // f32x4 per thread per plane
  constexpr bool PEXACT = (PLANE / 4) % NT == 0;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm;
  const int FP = ((n + 31) / 32) * 32;                    // feature rows per stash tile
  const long nt16 = 2 * ((A.B + 31) / 32);                // 16-point tiles: always both halves of a stash tile
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  f32x4* planes = reinterpret_cast<f32x4*>(smem);
  float* sm = smem + 2 * PLANE;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  float* dzs = sm + sm_tot + (long)wid * (2 * r * 64);   // per wave [r][64]
  float* sks = dzs + r * 64;
  float* lsum = sm + sm_tot + (long)WAVES * (2 * r * 64);
  const int o_w1 = 0, o_wl = si * n, o_b1 = o_wl + n * so, o_bh = o_b1 + n, o_bl = o_bh + nh * n;

  const int NPL = nh * (r + 1);
  const int nplanes = TRAIN ? 2 * NPL : NPL;
  auto plane_src = [&](int i) -> const f32x4* {
    if (i < NPL) return A.WF + (long)i * (PLANE / 4);
    const int ii = i - NPL;
    const int j = nh - 1 - ii / (r + 1), k = ii % (r + 1);
    return A.WB + ((long)j * (r + 1) + k) * (PLANE / 4);
  };

  {  // prologue: small hyper-vectors and plane 0 into LDS
    const long s_wl = (long)si * n + (long)nh * n * n;
    for (int idx = tid; idx < (r + 1) * nsm; idx += NT) {
      const int k = idx / nsm, e = idx - k * nsm;
      const long slot = e < si * n ? e : s_wl + (e - si * n);
      sm[idx] = hyp3(A, k, slot);
    }
    if (nplanes > 0) {
      const f32x4* src = plane_src(0);
#pragma unroll
      for (int q = 0; q < PF4; ++q)
        if (PEXACT || tid + NT * q < PLANE / 4) planes[tid + NT * q] = src[tid + NT * q];
    }
  }
  __syncthreads();
  int gpar = 0;
  float loss_lane = 0.f;
  float* dring = TRAIN ? A.dring + ((long)blockIdx.x * WAVES + wid) * (long)(nh + 1) * (NBL * 256) : nullptr;
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * A.slot_stride;

#define NIF_PLANE(...)                                                                        \
  {                                                                                           \
    const bool has_next = (pl + 1 < nplanes) || !last_group;                                  \
    f32x4 pre[PF4];                                                                           \
    if (has_next) {                                                                           \
      const f32x4* src = plane_src(pl + 1 < nplanes ? pl + 1 : 0);                            \
      _Pragma("unroll") for (int q = 0; q < PF4; ++q)                                         \
        if (PEXACT || tid + NT * q < PLANE / 4) pre[q] = src[tid + NT * q];                   \
    }                                                                                         \
    const f32x4* cur = planes + (gpar & 1) * (PLANE / 4);                                     \
    __VA_ARGS__                                                                               \
    if (has_next) {                                                                           \
      f32x4* dst = planes + ((gpar + 1) & 1) * (PLANE / 4);                                   \
      _Pragma("unroll") for (int q = 0; q < PF4; ++q)                                         \
        if (PEXACT || tid + NT * q < PLANE / 4) dst[tid + NT * q] = pre[q];                   \
    }                                                                                         \
    __syncthreads();                                                                          \
    ++gpar; ++pl;                                                                             \
  }

  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
    const bool last_group = tg + gridDim.x >= ngroups;
    const long t16_raw = tg * WAVES + wid;
    const bool active = t16_raw < nt16;
    const long t16 = active ? t16_raw : nt16 - 1;
    const long tile32 = t16 >> 1;
    const int poff = 16 * (int)(t16 & 1) + p;               // point column inside the 32-wide stash rows
    const long pt = t16 * 16 + p;
    const bool valid = active && pt < A.B;
    const long ptc = pt < A.B ? pt : A.B - 1;
    const float* xrow = A.xin + ptc * A.ncol + A.col0;
    const float* zt_base = A.Z + tile32 * r * 32 + poff;    // zt_k = zt_base[k*32]
    const long row0 = tile32 * (long)FP * 32 + poff;
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    f32x4 h[NBL], acc[NBL];
    // ---- first layer: a0 = sum_k zt_k (w0 * x . W1^(k) + b1^(k)) --------------------------------
#pragma unroll
    for (int b = 0; b < NBL; ++b) { acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; acc[b][3] = 0.f; }
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 32] : 1.0f;
      const float* s0 = sm + k * nsm;
#pragma unroll
      for (int b = 0; b < NBL; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int f = 16 * b + 4 * g + v;
          if (f < n) {
            float s = 0.f;
            for (int dd = 0; dd < si; ++dd) s = fmaf(xrow[dd], s0[o_w1 + dd * n + f], s);
            acc[b][v] = fmaf(zt, fmaf(A.omega, s, s0[o_b1 + f]), acc[b][v]);
          }
        }
    }
    {
      f32x4 d[NBL];
      act16<NBL, ACT>(A.act, acc, h, d, n, g);
      if (TRAIN) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) reinterpret_cast<f32x4*>(dring)[b * 64 + lane] = d[b];
      }
    }

    // ---- hidden hyper-matrices -------------------------------------------------------------------
    int pl = 0;
    f32x4 ublk[MODE == 1 ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
      if (TRAIN && active) st_store16<NBL>(IN0 + (long)j * A.slot_stride, row0, h, g);
#pragma unroll
      for (int b = 0; b < NBL; ++b) { acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; acc[b][3] = 0.f; }
      for (int k = 0; k <= r; ++k) {
        NIF_PLANE({
          if (k < r) {
            const float zt = zt_base[k * 32];
            f32x4 hz[NBL];
            _Pragma("unroll") for (int b = 0; b < NBL; ++b) hz[b] = zt * h[b];
            mfma16<NBL, true>(cur, hz, acc, lane);
          } else {
            mfma16<NBL, true>(cur, h, acc, lane);
          }
        })
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) acc[b] *= A.omega;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 32] : 1.0f;
        const float* sb = sm + k * nsm + o_bh + j * n;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int f = 16 * b + 4 * g + v;
            if (f < n) acc[b][v] = fmaf(zt, sb[f], acc[b][v]);
          }
      }
      {
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
        }
      }
    }

    // ---- last layer (n -> so, linear), MSE, start of the adjoint ---------------------------------
    if (TRAIN && active) st_store16<NBL>(IN0 + (long)nh * A.slot_stride, row0, h, g);
    f32x4 gh[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) { gh[b][0] = 0.f; gh[b][1] = 0.f; gh[b][2] = 0.f; gh[b][3] = 0.f; }
    const float wsamp = (valid ? (A.sw ? A.sw[ptc] : 1.0f) : 0.0f);
    float se = 0.f;
    for (int o = 0; o < so; ++o) {
      f32x4 wg[NBL];
#pragma unroll
      for (int b = 0; b < NBL; ++b) { wg[b][0] = 0.f; wg[b][1] = 0.f; wg[b][2] = 0.f; wg[b][3] = 0.f; }
      float part = 0.f, bias = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 32] : 1.0f;
        const float* s0 = sm + k * nsm;
        float sk = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int f = 16 * b + 4 * g + v;
            const float w = f < n ? s0[o_wl + f * so + o] : 0.f;
            sk = fmaf(h[b][v], w, sk);
            if (TRAIN) wg[b][v] = fmaf(zt, w, wg[b][v]);
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
        const float e = uo - A.y[ptc * so + o];
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
    }
    if (TRAIN) {
      if (g == 0) loss_lane += wsamp * se / (float)so * A.inv_bg;

      // ---- adjoint through the hidden hyper-matrices --------------------------------------------
      f32x4 skip[MODE == 0 ? 1 : NBL];
      for (int j = nh - 1; j >= 0; --j) {
        f32x4 ga[NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = reinterpret_cast<const f32x4*>(dring)[((j + 1) * NBL + b) * 64 + lane];
        if (MODE == 1 && (j & 1)) {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { skip[b] = 0.5f * gh[b]; ga[b] *= skip[b]; }
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) ga[b] *= gh[b];
          if (MODE == 2) {
#pragma unroll
            for (int b = 0; b < NBL; ++b) skip[b] = gh[b];
          }
        }
        if (active) st_store16<NBL>(DA0 + (long)(j + 1) * A.slot_stride, row0, ga, g);
#pragma unroll
        for (int b = 0; b < NBL; ++b) { gh[b][0] = 0.f; gh[b][1] = 0.f; gh[b][2] = 0.f; gh[b][3] = 0.f; }
        for (int k = 0; k <= r; ++k) {
          NIF_PLANE({
            if (k < r) {
              const float zt = zt_base[k * 32];
              f32x4 U[NBL];
              mfma16<NBL, false>(cur, ga, U, lane);
              _Pragma("unroll") for (int b = 0; b < NBL; ++b) gh[b] += zt * U[b];
              f32x4 hin[NBL];
              st_load16<NBL>(IN0 + (long)j * A.slot_stride, row0, hin, g);
              const float* sb = sm + k * nsm + o_bh + j * n;
              float s = 0.f, sbv = 0.f;
              _Pragma("unroll") for (int b = 0; b < NBL; ++b)
                _Pragma("unroll") for (int v = 0; v < 4; ++v) {
                  const int f = 16 * b + 4 * g + v;
                  s = fmaf(hin[b][v], U[b][v], s);
                  if (f < n) sbv = fmaf(ga[b][v], sb[f], sbv);
                }
              dzs[k * 64 + lane] += fmaf(A.omega, s, sbv);
            } else {
              mfma16<NBL, true>(cur, ga, gh, lane);
            }
          })
        }
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          gh[b] *= A.omega;
          if (MODE == 2 || (MODE == 1 && !(j & 1))) gh[b] += skip[b];
        }
      }
      // ---- first layer ---------------------------------------------------------------------------
      {
        f32x4 ga[NBL];
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = reinterpret_cast<const f32x4*>(dring)[b * 64 + lane] * gh[b];
        if (active) st_store16<NBL>(DA0, row0, ga, g);
        for (int k = 0; k < r; ++k) {
          const float* s0 = sm + k * nsm;
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int f = 16 * b + 4 * g + v;
              if (f < n) {
                float xw = 0.f;
                for (int dd = 0; dd < si; ++dd) xw = fmaf(xrow[dd], s0[o_w1 + dd * n + f], xw);
                s = fmaf(ga[b][v], fmaf(A.omega, xw, s0[o_b1 + f]), s);
              }
            }
          float tot = dzs[k * 64 + lane] + s;
          tot += __shfl_xor(tot, 16);
          tot += __shfl_xor(tot, 32);
          if (active && g == 0) A.DZ[(tile32 * r + k) * 32 + poff] = tot;
        }
      }
    }
  }
#undef NIF_PLANE
  if (TRAIN) {
    for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
    if (lane == 0) lsum[wid] = loss_lane;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int w = 0; w < WAVES; ++w) s += lsum[w];
      A.loss_partial[blockIdx.x] = s;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
static int nbl_of(int n) {
  const int c = (n + 15) / 16;   // pad to a supported block count
  return c <= 4 ? c : (c <= 6 ? 6 : 8);
}
int snet3_nbl(int n) { return nbl_of(n); }
static int waves_of(int NBL) { (void)NBL; return 4; }

static size_t snet3_shmem(const SNetArgs& a, int NBL, int waves) {
  const size_t plane = (size_t)NBL * NBL * 256;
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  return (2 * plane + sm_tot + (size_t)waves * 2 * a.r * 64 + 8) * sizeof(float);
}
bool snet3_supported(const SNetArgs& a) {
  const int NBL = nbl_of(a.n);
  if (a.n > 128) return false;
  return snet3_shmem(a, NBL, waves_of(NBL)) <= (NBL <= 4 ? 52u : 160u) * 1024u;
}
long snet3_plane_floats(int n) { const int NBL = nbl_of(n); return (long)NBL * NBL * 256; }
long snet3_ring_floats_per_wave(int n, int nh) { return (long)(nh + 1) * nbl_of(n) * 256; }

// returns #workgroups (x WAVES = ring owners); query_only: no launch
int launch_snet3(const SNetArgs& a, bool train, bool query_only, int* waves_out, hipStream_t st) {
  const int NBL = nbl_of(a.n);
  const int waves = waves_of(NBL);
  if (waves_out) *waves_out = waves;
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + waves - 1) / waves;
  const long cap = NBL <= 2 ? 256 * 4 : (NBL == 3 ? 256 * 3 : (NBL == 4 ? 256 * NIF_S3_OCC4 : 256));
  const int nblk = (int)(ngroups < cap ? ngroups : cap);
  if (query_only) return nblk;
  dim3 grid(nblk), block(waves * 64);
  const size_t shm = snet3_shmem(a, NBL, waves);
#define S3L(NBL_, W_, TR_, ACT_, MODE_)                                                                               \
  {                                                                                                                   \
    if (shm > 48 * 1024)                                                                                              \
      (void)hipFuncSetAttribute((const void*)k_snet3<NBL_, W_, TR_, ACT_, MODE_>,                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                \
    hipLaunchKernelGGL((k_snet3<NBL_, W_, TR_, ACT_, MODE_>), grid, block, shm, st, a);                               \
  }
#define S3(NBL_, W_)                                                             \
  if (a.nif_skip) {                                                              \
    if (train) S3L(NBL_, W_, true, -1, 2) else S3L(NBL_, W_, false, -1, 2)       \
  } else if (a.res) {                                                            \
    if (train) S3L(NBL_, W_, true, ACT_SINE, 1) else S3L(NBL_, W_, false, ACT_SINE, 1) \
  } else {                                                                       \
    if (train) S3L(NBL_, W_, true, ACT_SINE, 0) else S3L(NBL_, W_, false, ACT_SINE, 0) \
  }
  switch (NBL) {
    case 1: S3(1, 4) break;
    case 2: S3(2, 4) break;
    case 3: S3(3, 4) break;
    case 4: S3(4, 4) break;
    case 6: S3(6, 4) break;
    default: S3(8, 4) break;
  }
#undef S3
#undef S3L
  return nblk;
}
