// k_sob.hip -- Sobolev training step of the SIREN hypernetwork ShapeNet (BASELINE config 5): the model
// outputs (u, du/dx_d) through JacobianLayer (reference nif/layers/gradient.py:36-49) and the loss is
//     mse(u, y) + w_J * mse(du/dx, g)                      (Keras: two outputs, loss='mse', loss_weights)
// Forward = primal + forward-mode tangents (as k_jac); backward = the hand-derived adjoint of that pair:
// with lambda = dL/dh, mu^d = dL/dh'^d, c = cos(a), s = sin(a), a'^d the tangent pre-activation,
//     nu^d  = mu^d * c                        (dL/da'^d)
//     da    = lambda * c - sum_d mu^d * s * a'^d
//     lambda_in = w0 W(a) da ,  mu_in^d = w0 W(a) nu^d          (same MFMA planes, 1+ns right-hand sides)
//     dL/dM^(k) = w0 sum_p zt_k (h_in da^T + sum_d h'_in^d nu^dT)   -> the weight-gradient GEMMs simply see
//                 (1+ns) x more "points": the tangent pairs are stashed as pseudo-tiles
//     dL/dz_k  += w0 <h_in, M^(k) da> + <da, b^(k)> + w0 sum_d <h'_in^d, M^(k) nu^d>
// Coordinate seeds only (spatial derivatives); NIFMultiScale with or without resblocks and class NIF
// (MODE 2: any Keras activation f with skip connections; the ring then holds c = f'(a) and -f''(a) in place of cos / sin).
#include "k_snet3_dev.h"

#define NIF_SOB_MAXSEED 3
#ifndef NIF_SOB_OCC
#define NIF_SOB_OCC 1   // workgroups per CU for n <= 64 (2 = 256 registers each: 149 spills, 5.3 -> 8.3 ms)
#endif

struct SobArgs {
  SNetArgs s;
  int ns;                       // number of seeds (<= NIF_SOB_MAXSEED)
  int seed[NIF_SOB_MAXSEED];    // coordinate index d of each seed
  const float* gt;              // target derivatives [B][so][ns]
  float wj;                     // loss weight of the derivative term
  float* ring;                  // per wave [(nh+1)][2+NS][NBL][64][4]
  float* JU;                    // optional outputs du/dx [B][so][ns] (predict) or null
};

// BF: n x n products as exact bf16 splits on v_mfma_f32_16x16x32_bf16 (forward 6-product, adjoint 3-product form, see
// k_snet4.hip), whole bf16 planes per LDS step; otherwise the f32-input MFMA planes (odd block counts, n = 128)
// SGN (plain SIREN, training): the ring keeps only the tangent pre-activations a'^d of the ACTIVE seeds; cos(a) is
// rebuilt from the stashed sin(a) (the next layer's primal input) and its sign bit (k_snet4's shift register) --
// the ring was 5 blocks written + 5 read per layer, now ns written + ns read and one stash read
// (h, c, sn) of a pre-activation tile: SIREN: (sin, cos, sin); class NIF (MODE 2): (f, f', -f'') of the runtime activation, so that
// the adjoint formulas  nu = mu c ,  da = lambda c - sum mu sn a'  hold for both
template <int NBL, int MODE>
__device__ __forceinline__ void sob_act(int act, const f32x4 (&a)[NBL], f32x4 (&h)[NBL], f32x4 (&c)[NBL], f32x4 (&sn)[NBL], int n, int g) {
  if (MODE != 2) {
    sine16<NBL>(a, h, c);
#pragma unroll
    for (int b = 0; b < NBL; ++b) sn[b] = h[b];
  } else {
#pragma unroll
    for (int b = 0; b < NBL; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) sn[b][v] = -act_d2<-1>(act, a[b][v]);
    act16<NBL, -1>(act, a, h, c, n, g);
  }
}

// BF: 0 = f32-input MFMA planes, 1 = exact bf16 splits, 2 = one bf16 product (mixed_bfloat16 policy)
// NSD: seed streams the instantiation carries (register arrays and loops are sized by it): 1 or 2 seeds at n <= 64 leave room
// for TWO workgroups per CU (256 registers), the 3-seed form needs all 512
template <int NBL, int MODE, bool TRAIN, int BF, bool SGN, int NSD = NIF_SOB_MAXSEED>
__global__ __launch_bounds__(256, ((NBL <= 2 && NSD <= 2) || (NBL <= 4 && NSD == 1)) ? 2 : (NBL <= 4 ? NIF_SOB_OCC : 1)) void k_sob(SobArgs J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SNetArgs& A = J.s;
  constexpr int NT = 256, WAVES = 4, NS = NSD, NQ = 1 + NS;
  constexpr int NCH = NBL / 2, CF = NBL * 3 * 64, CB = NBL * 2 * 64;       // bf16 planes: K-step chunks, 16-B units
  constexpr int PLANE = BF ? NCH * CF * 4 : NBL * NBL * 256;              // floats per LDS plane buffer
  constexpr int UF = BF ? NCH * CF : PLANE / 4, UB = BF ? NCH * CB : PLANE / 4;   // 16-B units of a forward / adjoint plane
  constexpr int PF4 = (UF + NT - 1) / NT;
  constexpr int NP = 16 * NBL;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm, ns = J.ns;
  const int FP = ((n + 31) / 32) * 32;
  const long nt32 = (A.B + 31) / 32;
  const long nt16 = 2 * nt32;
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  f32x4* planes = reinterpret_cast<f32x4*>(smem);
  float* sm = smem + 2 * PLANE;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  float* dzs = sm + sm_tot + (long)wid * (r * 64 + r * 16);   // per wave dz partials [r][64], latent [r][16]
  float* zs = dzs + r * 64;
  float* lsum = sm + sm_tot + (long)WAVES * (r * 64 + r * 16);
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;
  const int NPL = nh * (r + 1);
  const int nplanes = TRAIN ? 2 * NPL : NPL;
  auto plane_src = [&](int i) -> const f32x4* {
    if (BF) {
      if (i < NPL) return reinterpret_cast<const f32x4*>(A.WF4) + (long)i * UF;
      const int ii = i - NPL;
      const int j = nh - 1 - ii / (r + 1), k = ii % (r + 1);
      return reinterpret_cast<const f32x4*>(A.WB4) + ((long)j * (r + 1) + k) * UB;
    }
    if (i < NPL) return A.WF + (long)i * (PLANE / 4);
    const int ii = i - NPL;
    const int j = nh - 1 - ii / (r + 1), k = ii % (r + 1);
    return A.WB + ((long)j * (r + 1) + k) * (PLANE / 4);
  };
  auto plane_units = [&](int i) -> int { return i < NPL ? UF : UB; };
  {
    const long s_wl = (long)si * n + (long)nh * n * n;
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
        if (tid + NT * q < plane_units(0)) planes[tid + NT * q] = src[tid + NT * q];
    }
  }
  __syncthreads();
  int gpar = 0;
  float loss_lane = 0.f;
  // ring: per layer l: block index (l*(2+NS) + which)*NBL + b, which = 0: cos(a), 1: sin(a), 2+d: a'^d
  f32x4* ring = TRAIN ? reinterpret_cast<f32x4*>(J.ring + ((long)blockIdx.x * WAVES + wid) * (long)(nh + 1) * (2 + NS) * (NBL * 256))
                      : nullptr;
  float* IN0 = A.stash;
  float* DA0 = A.stash + (long)(nh + 1) * A.slot_stride;

#define SOB_PLANE(...)                                                                        \
  {                                                                                           \
    if ((pl + 1 < nplanes) || !last_group) {                                                  \
      const int nxt_ = pl + 1 < nplanes ? pl + 1 : 0;                                         \
      const f32x4* src = plane_src(nxt_);                                                     \
      const int nu_ = plane_units(nxt_);                                                      \
      f32x4* dst = planes + ((gpar + 1) & 1) * (PLANE / 4);                                   \
      _Pragma("unroll") for (int q = 0; q < PF4; ++q)                                         \
        if (wid * 64 + NT * q < nu_)                                                          \
          __builtin_amdgcn_global_load_lds(                                                   \
              (const __attribute__((address_space(1))) void*)(src + tid + NT * q),            \
              (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);  \
    }                                                                                         \
    const f32x4* cur = planes + (gpar & 1) * (PLANE / 4);                                     \
    __VA_ARGS__                                                                               \
    __syncthreads();                                                                          \
    ++gpar; ++pl;                                                                             \
  }
#define ZERO4(x) { (x)[0] = 0.f; (x)[1] = 0.f; (x)[2] = 0.f; (x)[3] = 0.f; }

  for (long tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
    const bool last_group = tg + gridDim.x >= ngroups;
    const long t16_raw = tg * WAVES + wid;
    const bool active = t16_raw < nt16;
    const long t16 = active ? t16_raw : nt16 - 1;
    const long tile32 = t16 >> 1;
    const int poff = 16 * (int)(t16 & 1) + p;
    const long pt = t16 * 16 + p;
    const bool valid = active && pt < A.B;
    const long ptc = pt < A.B ? pt : A.B - 1;
    const float* xrow = A.xin + ptc * A.ncol + A.col0;
    if (g == 0)
      for (int k = 0; k < r; ++k) zs[k * 16 + p] = A.Z[(tile32 * r + k) * 32 + poff];
    const float* zt_base = zs + p;
    // stash rows of stream q (0 = primal, 1+d = tangent d): pseudo-tile q*nt32 + tile32
    auto row0 = [&](int q) -> long { return ((long)q * nt32 + tile32) * (long)FP * 32 + poff; };
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    // hq[0] = h, hq[1+d] = h'^d ; aq likewise for the pre-activation accumulators
    f32x4 hq[NQ][NBL], aq[NQ][NBL];
    unsigned long long sg_lo = 0ull, sg_hi = 0ull;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int b = 0; b < NBL; ++b) ZERO4(aq[q][b]);
    // ---- first layer ---------------------------------------------------------------------------
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 16] : 1.0f;
      const float* s0 = sm + k * nsm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dd = 0; dd < si; ++dd) s += xrow[dd] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        aq[0][b] += zt * (A.omega * s + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b));
#pragma unroll
        for (int d = 0; d < NS; ++d)
          if (d < ns) aq[1 + d][b] += (zt * A.omega) * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[d] * NP + 16 * b);
      }
    }
    {
      f32x4 c[NBL], sn0[NBL];
      sob_act<NBL, MODE>(A.act, aq[0], hq[0], c, sn0, n, g);
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        if (TRAIN && !SGN) { ring[0 * NBL * 64 + b * 64 + lane] = c[b]; ring[1 * NBL * 64 + b * 64 + lane] = sn0[b]; }
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          if (TRAIN && d < ns) ring[(2 + d) * NBL * 64 + b * 64 + lane] = aq[1 + d][b];
          hq[1 + d][b] = c[b] * aq[1 + d][b];
        }
      }
      if (TRAIN && SGN) sgn_push(sg_lo, sg_hi, sgn_pack<NBL>(c), 4 * NBL);
    }
    // ---- hidden hyper-matrices -------------------------------------------------------------------
    int pl = 0;
    f32x4 ub[MODE == 1 ? NQ : 1][MODE == 1 ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
      if (TRAIN && active) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q <= ns) st_store16<NBL>(IN0 + (long)j * A.slot_stride, row0(q), hq[q], g);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int b = 0; b < NBL; ++b) ZERO4(aq[q][b]);
      for (int k = 0; k <= r; ++k) {
        SOB_PLANE({
          const float zt = k < r ? zt_base[k * 16] : 1.0f;
          _Pragma("unroll") for (int q = 0; q < NQ; ++q)
            if (q <= ns) {
              f32x4 hz[NBL];
              _Pragma("unroll") for (int b = 0; b < NBL; ++b) hz[b] = zt * hq[q][b];
              if constexpr (BF) {
                bf16x8 b0[NCH], b1[NCH], b2[NCH];
                split3<NBL>(hz, b0, b1, b2);
                _Pragma("unroll") for (int ks = 0; ks < NCH; ++ks)
                  mfma_x6<NBL, BF == 2>(reinterpret_cast<const bf16x8*>(cur) + ks * CF, b0[ks], b1[ks], b2[ks], aq[q], lane);
              } else {
                mfma16<NBL, true>(cur, hz, aq[q], lane);
              }
            }
        })
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int b = 0; b < NBL; ++b) aq[q][b] *= A.omega;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) aq[0][b] += zt * *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
      f32x4 sn[NBL], c[NBL], snr[NBL];      // sn = f(a); snr = what the adjoint needs in the ring (SIREN: sin again; NIF: -f'')
      sob_act<NBL, MODE>(A.act, aq[0], sn, c, snr, n, g);
      f32x4* rl = ring + (long)(j + 1) * (2 + NS) * NBL * 64;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        if (TRAIN && !SGN) { rl[0 * NBL * 64 + b * 64 + lane] = c[b]; rl[1 * NBL * 64 + b * 64 + lane] = snr[b]; }
#pragma unroll
        for (int d = 0; d < NS; ++d)
          if (TRAIN && d < ns) rl[(2 + d) * NBL * 64 + b * 64 + lane] = aq[1 + d][b];
      }
      if (TRAIN && SGN) sgn_push(sg_lo, sg_hi, sgn_pack<NBL>(c), 4 * NBL);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 t = q == 0 ? sn[b] : c[b] * aq[q][b];   // f(a) | f'(a) a'
          if (MODE == 0) hq[q][b] = t;
          else if (MODE == 2) hq[q][b] += t;                   // class NIF: h = f(a) + h_in
          else if (!(j & 1)) { ub[q][b] = hq[q][b]; hq[q][b] = t; }
          else hq[q][b] = 0.5f * (ub[q][b] + t);
        }
    }
    // ---- last layer, loss, start of the adjoint ------------------------------------------------
    if (TRAIN && active) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q <= ns) st_store16<NBL>(IN0 + (long)nh * A.slot_stride, row0(q), hq[q], g);
    }
    f32x4 lam[NQ][NBL];   // lam[0] = dL/dh, lam[1+d] = dL/dh'^d
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int b = 0; b < NBL; ++b) ZERO4(lam[q][b]);
    const float wsamp = (valid ? (A.sw ? A.sw[ptc] : 1.0f) : 0.0f);
    float se = 0.f, sej = 0.f;
    for (int o = 0; o < so; ++o) {
      f32x4 wg[NBL];
#pragma unroll
      for (int b = 0; b < NBL; ++b) ZERO4(wg[b]);
      float part[NQ], bias = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) part[q] = 0.f;
      // sk[q][k] partial dots are needed again for dz: recompute in a second k loop below
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* s0 = sm + k * nsm;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            part[q] = fmaf(zt, (hq[q][b][0] * w[0] + hq[q][b][1] * w[1]) + (hq[q][b][2] * w[2] + hq[q][b][3] * w[3]), part[q]);
          if (TRAIN) wg[b] += zt * w;
        }
        bias = fmaf(zt, s0[o_bl + o], bias);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) { part[q] += __shfl_xor(part[q], 16); part[q] += __shfl_xor(part[q], 32); }
      const float uo = part[0] + bias;
      if (valid && g == 0) {
        if (A.u_out) A.u_out[pt * so + o] = uo;
        if (J.JU)
#pragma unroll
          for (int d = 0; d < NS; ++d)
            if (d < ns) J.JU[(pt * so + o) * ns + d] = part[1 + d];
      }
      if (TRAIN) {
        float dq[NQ];
        const float e = uo - A.y[ptc * so + o];
        se = fmaf(e, e, se);
        dq[0] = 2.0f * wsamp * e * A.inv_bg / (float)so;
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          dq[1 + d] = 0.f;
          if (d < ns) {
            const float ej = part[1 + d] - J.gt[(ptc * so + o) * ns + d];
            sej = fmaf(ej, ej, sej);
            dq[1 + d] = 2.0f * J.wj * wsamp * ej * A.inv_bg / (float)(so * ns);
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q <= ns && active && g == 0) A.DU[(((long)q * nt32 + tile32) * so + o) * 32 + poff] = dq[q];
#pragma unroll
          for (int b = 0; b < NBL; ++b) lam[q][b] += dq[q] * wg[b];
        }
        // dL/dz_k += sum_q dq[q] * <hq[q], Wl^(k)[:,o]>  + dq[0] * bl^(k)[o]
        for (int k = 0; k < r; ++k) {
          const float* s0 = sm + k * nsm;
          float t = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              t = fmaf(dq[q], (hq[q][b][0] * w[0] + hq[q][b][1] * w[1]) + (hq[q][b][2] * w[2] + hq[q][b][3] * w[3]), t);
          }
          if (g == 0) t = fmaf(dq[0], s0[o_bl + o], t);
          dzs[k * 64 + lane] += t;
        }
      }
    }
    if (TRAIN) {
      if (g == 0) loss_lane += wsamp * A.inv_bg * (se / (float)so + J.wj * sej / (float)(so * ns));

      // ---- adjoint through the hidden hyper-matrices --------------------------------------------
      f32x4 skip[MODE != 0 ? NQ : 1][MODE != 0 ? NBL : 1];
      for (int j = nh - 1; j >= 0; --j) {
        const f32x4* rl = ring + (long)(j + 1) * (2 + NS) * NBL * 64;
        f32x4 vq[NQ][NBL];   // vq[0] = da, vq[1+d] = nu^d
        if (MODE == 1 && (j & 1)) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int b = 0; b < NBL; ++b) { lam[q][b] *= 0.5f; skip[q][b] = lam[q][b]; }
        }
        if (MODE == 2) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int b = 0; b < NBL; ++b) skip[q][b] = lam[q][b];
        }
        f32x4 cv[NBL], snv[NBL];
        if (SGN) {
          if (j == nh - 1) {
#pragma unroll
            for (int b = 0; b < NBL; ++b) snv[b] = hq[0][b];                  // sin(a) of the top layer: still in registers
          } else {
            st_load16<NBL>(IN0 + (long)(j + 1) * A.slot_stride, row0(0), snv, g);   // = the next layer's primal input
          }
          sgn_cos<NBL>(snv, sgn_pop(sg_lo, sg_hi, 4 * NBL), cv);
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { cv[b] = rl[0 * NBL * 64 + b * 64 + lane]; snv[b] = rl[1 * NBL * 64 + b * 64 + lane]; }
        }
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 c = cv[b], sn = snv[b];
          f32x4 da = lam[0][b] * c;
#pragma unroll
          for (int d = 0; d < NS; ++d) {
            vq[1 + d][b] = lam[1 + d][b] * c;
            if (d < ns) da -= lam[1 + d][b] * sn * rl[(2 + d) * NBL * 64 + b * 64 + lane];
          }
          vq[0][b] = da;
        }
        if (active) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q <= ns) st_store16<NBL>(DA0 + (long)(j + 1) * A.slot_stride, row0(q), vq[q], g);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int b = 0; b < NBL; ++b) ZERO4(lam[q][b]);
        for (int k = 0; k <= r; ++k) {
          SOB_PLANE({
            const float zt = k < r ? zt_base[k * 16] : 1.0f;
            float dzk = 0.f;
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)
              if (q <= ns) {
                if (k < r) {
                  f32x4 U[NBL], hin[NBL];
                  if constexpr (BF) {
                    bf16x8 b0[NCH], b1[NCH];
                    split2<NBL>(vq[q], b0, b1);
                    _Pragma("unroll") for (int b = 0; b < NBL; ++b) ZERO4(U[b]);
                    _Pragma("unroll") for (int ks = 0; ks < NCH; ++ks)
                      mfma_x3<NBL, BF == 2>(reinterpret_cast<const bf16x8*>(cur) + ks * CB, b0[ks], b1[ks], U, lane);
                  } else {
                    mfma16<NBL, false>(cur, vq[q], U, lane);
                  }
                  st_load16<NBL>(IN0 + (long)j * A.slot_stride, row0(q), hin, g);
                  _Pragma("unroll") for (int b = 0; b < NBL; ++b) {
                    lam[q][b] += zt * U[b];
                    dzk += (hin[b][0] * U[b][0] + hin[b][1] * U[b][1]) + (hin[b][2] * U[b][2] + hin[b][3] * U[b][3]);
                  }
                } else {
                  if constexpr (BF) {
                    bf16x8 b0[NCH], b1[NCH];
                    split2<NBL>(vq[q], b0, b1);
                    _Pragma("unroll") for (int ks = 0; ks < NCH; ++ks)
                      mfma_x3<NBL, BF == 2>(reinterpret_cast<const bf16x8*>(cur) + ks * CB, b0[ks], b1[ks], lam[q], lane);
                  } else {
                    mfma16<NBL, true>(cur, vq[q], lam[q], lane);
                  }
                }
              }
            if (k < r) {
              const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
              float sbv = 0.f;
              _Pragma("unroll") for (int b = 0; b < NBL; ++b) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
                sbv += (vq[0][b][0] * bb[0] + vq[0][b][1] * bb[1]) + (vq[0][b][2] * bb[2] + vq[0][b][3] * bb[3]);
              }
              dzs[k * 64 + lane] += fmaf(A.omega, dzk, sbv);
            }
          })
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            lam[q][b] *= A.omega;
            if (MODE == 2 || (MODE == 1 && !(j & 1))) lam[q][b] += skip[q][b];
          }
      }
      // ---- first layer ---------------------------------------------------------------------------
      {
        f32x4 vq[NQ][NBL];
        f32x4 cv[NBL], snv[NBL];
        if (SGN) {
          st_load16<NBL>(IN0, row0(0), snv, g);                               // h_0 = sin(a_0)
          sgn_cos<NBL>(snv, sgn_pop(sg_lo, sg_hi, 4 * NBL), cv);
        } else {
#pragma unroll
          for (int b = 0; b < NBL; ++b) { cv[b] = ring[0 * NBL * 64 + b * 64 + lane]; snv[b] = ring[1 * NBL * 64 + b * 64 + lane]; }
        }
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 c = cv[b], sn = snv[b];
          f32x4 da = lam[0][b] * c;
#pragma unroll
          for (int d = 0; d < NS; ++d) {
            vq[1 + d][b] = lam[1 + d][b] * c;
            if (d < ns) da -= lam[1 + d][b] * sn * ring[(2 + d) * NBL * 64 + b * 64 + lane];
          }
          vq[0][b] = da;
        }
        if (active) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q <= ns) st_store16<NBL>(DA0, row0(q), vq[q], g);
        }
        for (int k = 0; k < r; ++k) {
          const float* s0 = sm + k * nsm + 4 * g;
          float s = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            f32x4 xw = {0.f, 0.f, 0.f, 0.f};
            for (int dd = 0; dd < si; ++dd) xw += xrow[dd] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
            const f32x4 t = A.omega * xw + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
            s += (vq[0][b][0] * t[0] + vq[0][b][1] * t[1]) + (vq[0][b][2] * t[2] + vq[0][b][3] * t[3]);
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (d < ns) {
                const f32x4 wd = A.omega * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[d] * NP + 16 * b);
                s += (vq[1 + d][b][0] * wd[0] + vq[1 + d][b][1] * wd[1]) + (vq[1 + d][b][2] * wd[2] + vq[1 + d][b][3] * wd[3]);
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
#undef SOB_PLANE
#undef ZERO4
  if (TRAIN) {
    for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
    if (lane == 0) lsum[wid] = loss_lane;
    __syncthreads();
    if (tid == 0) A.loss_partial[blockIdx.x] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  }
}

long sob_ring_floats_per_wave(int n, int nh) { return (long)(nh + 1) * (2 + NIF_SOB_MAXSEED) * snet3_nbl(n) * 256; }

int launch_sob(const SNetArgs& a, bool train, int ns, const int* seeds, const float* gt, float wj, float* ring, float* ju,
               bool query_only, hipStream_t st) {
  const int NBL = snet3_nbl(a.n);
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 3) / 4;
  const bool bf_ = a.WF4 && a.WB4 && !(NBL & 1) && NBL <= 6;
  const bool slim = bf_ && NBL <= 4 && ns <= 2 && !a.nif_skip;     // the 1- / 2-seed instantiations: two workgroups per CU
  // measured on cfg-5 (n = 64): one seed 5.21 -> 3.98 ms at two workgroups per CU; two seeds spill 83 registers there (6.97 -> 7.87 ms)
  const bool two = slim && (NBL <= 2 || ns == 1);
  const long cap = two ? 512 : (NBL <= 4 ? 256 * NIF_SOB_OCC : 256);
  const int nblk = (int)(ngroups < cap ? ngroups : cap);
  if (query_only) return nblk;
  SobArgs J;
  J.s = a; J.ns = ns; J.gt = gt; J.wj = wj; J.ring = ring; J.JU = ju;
  for (int d = 0; d < NIF_SOB_MAXSEED; ++d) J.seed[d] = d < ns ? seeds[d] : 0;
  dim3 grid(nblk), block(256);
  const bool bf = a.WF4 && a.WB4 && !(NBL & 1) && NBL <= 6;      // whole bf16 planes in LDS: up to n = 96
  const bool sgn = !a.res && !a.nif_skip && (long)(a.nh + 1) * 4 * NBL <= 128;  // sign bits fit the 128-bit shift register (SIREN only)
  const size_t plane = bf ? (size_t)(NBL / 2) * NBL * 3 * 64 * 4 : (size_t)NBL * NBL * 256;
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const size_t shm = (2 * plane + sm_tot + 4 * (size_t)(a.r * 64 + a.r * 16) + 8) * sizeof(float);
#define SBL(NBL_, MODE_, TR_, BF_, SGN_)                                                                            \
  {                                                                                                             \
    if (shm > 48 * 1024)                                                                                        \
      (void)hipFuncSetAttribute((const void*)k_sob<NBL_, MODE_, TR_, BF_, SGN_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)shm);                                                                      \
    hipLaunchKernelGGL((k_sob<NBL_, MODE_, TR_, BF_, SGN_>), grid, block, shm, st, J);                                \
  }
#define SBN(NBL_, MODE_, TR_, BF_, SGN_, NSD_)                                                                      \
  {                                                                                                             \
    if (shm > 48 * 1024)                                                                                        \
      (void)hipFuncSetAttribute((const void*)k_sob<NBL_, MODE_, TR_, BF_, SGN_, NSD_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)shm);                                                                      \
    hipLaunchKernelGGL((k_sob<NBL_, MODE_, TR_, BF_, SGN_, NSD_>), grid, block, shm, st, J);                          \
  }
  // SIREN nets on the bf16 planes with 1 or 2 seeds (BASELINE configs[4]: d/dx, d/dy)
#define SBS(NBL_, BF_, NSD_)                                                         \
  if (a.res) { if (train) SBN(NBL_, 1, true, BF_, false, NSD_) else SBN(NBL_, 1, false, BF_, false, NSD_) }   \
  else if (train) { if (sgn) SBN(NBL_, 0, true, BF_, true, NSD_) else SBN(NBL_, 0, true, BF_, false, NSD_) } \
  else SBN(NBL_, 0, false, BF_, false, NSD_)
#define SBK(NBL_, BF_)                                                               \
  if (a.nif_skip) { if (train) SBL(NBL_, 2, true, BF_, false) else SBL(NBL_, 2, false, BF_, false) }   \
  else if (a.res) { if (train) SBL(NBL_, 1, true, BF_, false) else SBL(NBL_, 1, false, BF_, false) }   \
  else if (train) { if (sgn) SBL(NBL_, 0, true, BF_, true) else SBL(NBL_, 0, true, BF_, false) } \
  else SBL(NBL_, 0, false, BF_, false)
  if (slim) {
    const int bfv = a.prec == 1 ? 2 : 1;
    if (NBL == 4) {
      if (bfv == 1) { if (ns == 1) { SBS(4, 1, 1) } else { SBS(4, 1, 2) } }
      else { if (ns == 1) { SBS(4, 2, 1) } else { SBS(4, 2, 2) } }
    } else {
      if (bfv == 1) { if (ns == 1) { SBS(2, 1, 1) } else { SBS(2, 1, 2) } }
      else { if (ns == 1) { SBS(2, 2, 1) } else { SBS(2, 2, 2) } }
    }
    return nblk;
  }
  switch (NBL) {
    case 1: SBK(1, 0) break;
    case 2: if (bf && a.prec == 1) { SBK(2, 2) } else if (bf) { SBK(2, 1) } else { SBK(2, 0) } break;
    case 3: SBK(3, 0) break;
    case 4: if (bf && a.prec == 1) { SBK(4, 2) } else if (bf) { SBK(4, 1) } else { SBK(4, 0) } break;
    case 6: if (bf && a.prec == 1) { SBK(6, 2) } else if (bf) { SBK(6, 1) } else { SBK(6, 0) } break;
    default: SBK(8, 0) break;
  }
#undef SBK
#undef SBS
#undef SBN
#undef SBL
  return nblk;
}
