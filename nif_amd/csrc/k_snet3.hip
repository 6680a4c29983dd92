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
#include "k_snet3_dev.h"

// Ablation switches for timing experiments (NEVER set in a product build: results become wrong).
//   NIF_ABL_NOSTORE  skip the stash / ring stores      NIF_ABL_NOACT   cheap stand-in for the activation
//   NIF_ABL_NOBAR    skip the per-plane barrier         NIF_ABL_NOMFMA  skip the MFMAs
#ifndef NIF_S3_PREFETCH_ADJ
#define NIF_S3_PREFETCH_ADJ 0   // fetch act'(a) and h_in one layer ahead in the adjoint
#endif
#ifndef NIF_S3_LDSDMA
#define NIF_S3_LDSDMA 1          // stage weight planes with global_load_lds instead of through registers
#endif
#ifndef NIF_S3_OCC4
#define NIF_S3_OCC4 3   // waves/SIMD requested for the 64-wide (NBL = 4) instantiation
#endif


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
  const int FP = stash_fp(n);                             // feature rows per stash tile (nif_internal.h)
  const long nt16 = 2 * ((A.B + 31) / 32);                // 16-point tiles: always both halves of a stash tile
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  f32x4* planes = reinterpret_cast<f32x4*>(smem);
  float* sm = smem + 2 * PLANE;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  float* dzs = sm + sm_tot + (long)wid * (2 * r * 64 + r * 16);   // per wave [r][64]
  float* sks = dzs + r * 64;
  float* zs = sks + r * 64;                                        // per wave [r][16]: this tile's latent
  float* lsum = sm + sm_tot + (long)WAVES * (2 * r * 64 + r * 16);
  // LDS copy of the small hyper-vectors, per k: feature-contiguous, zero-padded to NP = 16*NBL so that a
  // lane fetches its 4 features of a block with one ds_read_b128 and needs no bounds checks
  constexpr int NP = 16 * NBL;
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;

  const int NPL = nh * (r + 1);
  const int nplanes = TRAIN ? 2 * NPL : NPL;
  auto plane_src = [&](int i) -> const f32x4* {
    if (i < NPL) return A.WF + (long)i * (PLANE / 4);
    const int ii = i - NPL;
    const int j = nh - 1 - ii / (r + 1), k = ii % (r + 1);
    return A.WB + ((long)j * (r + 1) + k) * (PLANE / 4);
  };

  {  // prologue: small hyper-vectors and plane 0 into LDS
    const long s_wl = (long)si * n + (long)nh * n * n;        // pnet_output slots (model.py:253-300)
    const long s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
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
    if (nplanes > 0) {
      const f32x4* src = plane_src(0);
#pragma unroll
      for (int q = 0; q < PF4; ++q)
        if (PEXACT || tid + NT * q < PLANE / 4) planes[tid + NT * q] = src[tid + NT * q];
    }
  }
  __syncthreads();
#ifdef NIF_S3_STAGGER
  // de-phase the workgroups that share a CU (the second wave of residents starts NIF_S3_STAGGER*64 cycles
  // late), so that one wave's VALU epilogue overlaps its SIMD partner's MFMA plane instead of colliding
  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(NIF_S3_STAGGER);
#endif
  int gpar = 0;
  int tlc = 0; (void)tlc;
  float loss_lane = 0.f;
  float* dring = TRAIN ? A.dring + ((long)blockIdx.x * WAVES + wid) * (long)(nh + 1) * (NBL * 256) : nullptr;
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * A.slot_stride;

#ifdef NIF_TIMELINE
#define NIF_TL(id) do { if (A.tl && blockIdx.x == 0 && tid == 0 && tlc < 250) { A.tl[2 * tlc] = (id); A.tl[2 * tlc + 1] = (long long)__builtin_amdgcn_s_memtime(); ++tlc; } } while (0)
#else
#define NIF_TL(id) do { } while (0)
#endif
#ifdef NIF_ABL_NOPREFETCH
#define NIF_HAS_NEXT false
#else
#define NIF_HAS_NEXT ((pl + 1 < nplanes) || !last_group)
#endif
#ifdef NIF_ABL_NOBAR
#define NIF_BAR() __builtin_amdgcn_wave_barrier()
#else
#define NIF_BAR() __syncthreads()
#endif
#if NIF_S3_LDSDMA
// the next plane goes L2 -> LDS by DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, our plane
// image is lane-linear), no staging registers; hipcc drains it (vmcnt(0)) in front of the barrier
#define NIF_PLANE(...)                                                                        \
  {                                                                                           \
    if (NIF_HAS_NEXT) {                                                                       \
      const f32x4* src = plane_src(pl + 1 < nplanes ? pl + 1 : 0);                            \
      f32x4* dst = planes + ((gpar + 1) & 1) * (PLANE / 4);                                   \
      _Pragma("unroll") for (int q = 0; q < PF4; ++q)                                         \
        if (PEXACT || wid * 64 + NT * q < PLANE / 4)                                          \
          __builtin_amdgcn_global_load_lds(                                                   \
              (const __attribute__((address_space(1))) void*)(src + tid + NT * q),            \
              (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);  \
    }                                                                                         \
    const f32x4* cur = planes + (gpar & 1) * (PLANE / 4);                                     \
    __VA_ARGS__                                                                               \
    NIF_BAR();                                                                                \
    ++gpar; ++pl;                                                                             \
  }
#else
#define NIF_PLANE(...)                                                                        \
  {                                                                                           \
    const bool has_next = NIF_HAS_NEXT;                                                       \
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
    NIF_BAR();                                                                                \
    ++gpar; ++pl;                                                                             \
  }

#endif

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
    // the latent of this tile goes through LDS: a global load inside a plane would make its s_waitcnt
    // vmcnt drain the in-flight prefetch of the next weight plane (vmcnt retires in order)
    if (g == 0)
      for (int k = 0; k < r; ++k) zs[k * 16 + p] = A.Z[(tile32 * r + k) * 32 + poff];
    const float* zt_base = zs + p;                           // zt_k = zt_base[k*16]
    const long row0 = tile32 * (long)FP * 32 + poff;
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    NIF_TL(1);
    f32x4 h[NBL], acc[NBL];
    // ---- first layer: a0 = sum_k zt_k (w0 * x . W1^(k) + b1^(k)) --------------------------------
#pragma unroll
    for (int b = 0; b < NBL; ++b) { acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; acc[b][3] = 0.f; }
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 16] : 1.0f;
      const float* s0 = sm + k * nsm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dd = 0; dd < si; ++dd) s += xrow[dd] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        acc[b] += zt * (A.omega * s + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b));
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

    NIF_TL(2);
    // ---- hidden hyper-matrices -------------------------------------------------------------------
    int pl = 0;
    f32x4 ublk[MODE == 1 ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
      if (TRAIN && active) st_store16<NBL>(IN0 + (long)j * A.slot_stride, row0, h, g);
      NIF_TL(10 + j);
#pragma unroll
      for (int b = 0; b < NBL; ++b) { acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; acc[b][3] = 0.f; }
      for (int k = 0; k <= r; ++k) {
        NIF_PLANE({
          if (k < r) {
            const float zt = zt_base[k * 16];
            f32x4 hz[NBL];
            _Pragma("unroll") for (int b = 0; b < NBL; ++b) hz[b] = zt * h[b];
            mfma16<NBL, true>(cur, hz, acc, lane);
          } else {
            mfma16<NBL, true>(cur, h, acc, lane);
          }
        })
      }
      NIF_TL(30 + j);
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

    NIF_TL(3);
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
        const float e = uo - A.y[ptc * so + o];
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
      if (g == 0) loss_lane += wsamp * se / (float)so * A.inv_bg;
      NIF_TL(4);

      // ---- adjoint through the hidden hyper-matrices --------------------------------------------
      // act'(a_j) and h_{j-1} are fetched one layer ahead so that their L2/HBM latency hides behind the
      // previous layer's MFMA planes
      f32x4 skip[MODE == 0 ? 1 : NBL];
      f32x4 dnext[NBL], hin[NBL];
#if NIF_S3_PREFETCH_ADJ
#pragma unroll
      for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[(nh * NBL + b) * 64 + lane];
      if (nh > 0 && r > 0) st_load16<NBL>(IN0 + (long)(nh - 1) * A.slot_stride, row0, hin, g);
#endif
      for (int j = nh - 1; j >= 0; --j) {
        f32x4 ga[NBL];
#if !NIF_S3_PREFETCH_ADJ
#pragma unroll
        for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[((j + 1) * NBL + b) * 64 + lane];
#endif
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
#if NIF_S3_PREFETCH_ADJ
        // next layer's act'(a) (layer j-1, or the first layer's when j == 0)
#pragma unroll
        for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[(j * NBL + b) * 64 + lane];
#endif
        if (active) st_store16<NBL>(DA0 + (long)(j + 1) * A.slot_stride, row0, ga, g);
        NIF_TL(50 + j);
#pragma unroll
        for (int b = 0; b < NBL; ++b) { gh[b][0] = 0.f; gh[b][1] = 0.f; gh[b][2] = 0.f; gh[b][3] = 0.f; }
        for (int k = 0; k <= r; ++k) {
          NIF_PLANE({
            if (k < r) {
              const float zt = zt_base[k * 16];
              f32x4 U[NBL];
              mfma16<NBL, false>(cur, ga, U, lane);
              _Pragma("unroll") for (int b = 0; b < NBL; ++b) gh[b] += zt * U[b];
#if !NIF_S3_PREFETCH_ADJ
              st_load16<NBL>(IN0 + (long)j * A.slot_stride, row0, hin, g);
#endif
              const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
              float s = 0.f, sbv = 0.f;
              _Pragma("unroll") for (int b = 0; b < NBL; ++b) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
                _Pragma("unroll") for (int v = 0; v < 4; ++v) {
                  s = fmaf(hin[b][v], U[b][v], s);
                  sbv = fmaf(ga[b][v], bb[v], sbv);
                }
              }
              dzs[k * 64 + lane] += fmaf(A.omega, s, sbv);
#if NIF_S3_PREFETCH_ADJ
              if (k == r - 1 && j > 0) st_load16<NBL>(IN0 + (long)(j - 1) * A.slot_stride, row0, hin, g);
#endif
            } else {
              mfma16<NBL, true>(cur, ga, gh, lane);
            }
          })
        }
        NIF_TL(70 + j);
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          gh[b] *= A.omega;
          if (MODE == 2 || (MODE == 1 && !(j & 1))) gh[b] += skip[b];
        }
      }
      NIF_TL(5);
      // ---- first layer ---------------------------------------------------------------------------
      {
        f32x4 ga[NBL];
#if !NIF_S3_PREFETCH_ADJ
#pragma unroll
        for (int b = 0; b < NBL; ++b) dnext[b] = reinterpret_cast<const f32x4*>(dring)[b * 64 + lane];
#endif
#pragma unroll
        for (int b = 0; b < NBL; ++b) ga[b] = dnext[b] * gh[b];
        if (active) st_store16<NBL>(DA0, row0, ga, g);
        for (int k = 0; k < r; ++k) {
          const float* s0 = sm + k * nsm + 4 * g;
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            f32x4 xw = {0.f, 0.f, 0.f, 0.f};
            for (int dd = 0; dd < si; ++dd) xw += xrow[dd] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
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
int snet3_nsm(int si, int so, int nh, int n) { return (si + so + 1 + nh) * 16 * nbl_of(n) + ((so + 3) & ~3); }
static int waves_of(int NBL) { (void)NBL; return 4; }

static size_t snet3_shmem(const SNetArgs& a, int NBL, int waves) {
  const size_t plane = (size_t)NBL * NBL * 256;
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  return (2 * plane + sm_tot + (size_t)waves * (2 * a.r * 64 + a.r * 16) + 8) * sizeof(float);
}
bool snet3_supported(const SNetArgs& a) {
  const int NBL = nbl_of(a.n);
  if (a.n > 128) return false;
  // (narrow nets: up to 52 KB keeps three workgroups per CU; beyond that the kernel still runs, at lower occupancy -- far better
  // than the 32-point fallback kernel, and JacobianLayer / HessianLayer / Sobolev exist only on this path)
  return snet3_shmem(a, NBL, waves_of(NBL)) <= 160u * 1024u;
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
