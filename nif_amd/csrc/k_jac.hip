// k_jac.hip -- JacobianLayer for the hypernetwork classes (reference nif/layers/gradient.py:36-49,
// :207-231): (y, dy/dx) w.r.t. the coordinate columns of the model input, by forward-mode tangents
// carried next to the primal through the same register-resident MFMA chain as k_snet3:
//
//   first layer   a0' = w0 * sum_k zt_k W1^(k)[d][:]            h0' = act'(a0) * a0'
//   hidden        a'  = w0 * sum_k zt_k (h' . M^(k))            plain: h' = act'(a) a'
//                                                              NIF:   h' = act'(a) a' + h'_in
//                                                              res:   t' = act'(a1) a1';  u' = 0.5 (u'_in + act'(a2) a2')
//   last          u'  = h' . Wl(a)
//
// Parameter columns (p = t, mu): the per-sample weights move too, W'(a) = sum_k z'_k M^(k) with z' = dz/dp from
// k_mlp_jac, so every layer adds  sum_{k<r} z'_k (h . M^(k))  (+ z'_k b^(k)) -- the unscaled partial products
// T_k = h . M^(k) the forward pass forms anyway.
//
// The per-sample weights do not depend on x, so one extra B operand per seed reuses every A operand
// (weight plane) already in LDS.  The reference needs len(y_index) extra reverse sweeps instead.
#include "k_snet3_dev.h"

#define NIF_JAC_MAXSEED 3

struct JacArgs {
  SNetArgs s;                 // primal arguments (u_out = y)
  int ns;                     // seeds in this launch (<= NIF_JAC_MAXSEED)
  int seed[NIF_JAC_MAXSEED];  // coordinate index d (0..si-1) of a coordinate seed, or -1 for a parameter seed
  const float* ZD[NIF_JAC_MAXSEED];  // parameter seed: dz/dp_col of the latent, [tiles][r][32] (from k_mlp_jac)
  int nx_total, x0;           // dydx row stride (number of requested x columns) and first column of this launch
  float* dydx;                // [B][so][nx_total]
  // HessianLayer launches (k_jac<..., HESS>): seeds 0, 1 = the coordinate pair at x positions (hj, hk); stream 2 -> d2[B][so][nx][nx]
  int hj, hk; float* d2ydx2;
  // 1: ONE plane buffer in LDS instead of two (shapes whose small hyper-vectors leave no room for the second 64-KB plane of a
  // 128-wide net: latent_dim >= 4 with many matrices).  The next plane still travels through registers while the current one
  // is multiplied; it lands after an extra barrier.  Slower, but the derivative layers never refuse a shape the step trains.
  int one_buf;
};

// HESS (HessianLayer, gradient.py:130-180, :234-261): streams 0 and 1 are the first-order tangents of two coordinate seeds
// (j, k), stream 2 is the SECOND-order tangent of the pair:  a'' = w0 W(a) h'' ,  h'' = f'(a) a'' + f''(a) a'_j a'_k  (the
// first layer is linear in x: a'' = 0); it leaves through the linear last layer like a first-order tangent.
// Parameter columns in the pair (seed < 0, ZD = dz/dp, and for two of them ZD[2] = d2z/dp_j dp_k from k_pjac2): the weights carry
// tangents of their own and every layer gets the second-order product rule,
//   a'' = w0 sum_k (zt_k h'' + z'_j,k h'_k + z'_k,k h'_j + z''_k h) M^(k) + sum_k z''_k b^(k)
// formed from the UNSCALED partial products T = h . M^(k), T_j = h'_j . M^(k), T_k = h'_k . M^(k) of the plane in LDS.
template <int NBL, int ACT, int MODE, bool HESS = false>
__global__ __launch_bounds__(256, (NBL <= 4 ? 2 : 1)) void k_jac(JacArgs J) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SNetArgs& A = J.s;
  constexpr int NT = 256, WAVES = 4, NS = NIF_JAC_MAXSEED;
  constexpr int PLANE = NBL * NBL * 256;
  constexpr int PF4 = (PLANE / 4 + NT - 1) / NT;
  constexpr bool PEXACT = (PLANE / 4) % NT == 0;
  constexpr int NP = 16 * NBL;
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int n = A.n, r = A.r, nh = A.nh, si = A.si, so = A.so, nsm = A.nsm, ns = J.ns;
  const long nt16 = 2 * ((A.B + 31) / 32);
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  f32x4* planes = reinterpret_cast<f32x4*>(smem);
  const int one_buf = J.one_buf;
  float* sm = smem + (one_buf ? 1 : 2) * PLANE;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  float* zs = sm + sm_tot + (long)wid * ((1 + NIF_JAC_MAXSEED) * r * 16);
  float* zds = zs + r * 16;   // [seed][r][16]: dz_k/dp for parameter seeds (0 for coordinate seeds)
  bool anyp = false;
#pragma unroll
  for (int d = 0; d < NIF_JAC_MAXSEED; ++d) anyp = anyp || (d < J.ns && J.seed[d] < 0);
  const int o_w1 = 0, o_wl = si * NP, o_b1 = o_wl + so * NP, o_bh = o_b1 + NP, o_bl = o_bh + nh * NP;
  const int nplanes = nh * (r + 1);

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
#pragma unroll
      for (int q = 0; q < PF4; ++q)
        if (PEXACT || tid + NT * q < PLANE / 4) planes[tid + NT * q] = A.WF[tid + NT * q];
    }
  }
  __syncthreads();
  int gpar = 0;

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
      for (int k = 0; k < r; ++k) {
        zs[k * 16 + p] = A.Z[(tile32 * r + k) * 32 + poff];
#pragma unroll
        for (int d = 0; d < NS; ++d)
          zds[(d * r + k) * 16 + p] = (d < ns && J.seed[d] < 0 && J.ZD[d]) ? J.ZD[d][(tile32 * r + k) * 32 + poff] : 0.f;
      }
    const float* zt_base = zs + p;
    const float* zd_base = zds + p;   // zd(d,k) = zd_base[(d*r+k)*16]

    f32x4 h[NBL], acc[NBL], hd[NS][NBL], accd[NS][NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
      acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; acc[b][3] = 0.f;
#pragma unroll
      for (int d = 0; d < NS; ++d) { accd[d][b][0] = 0.f; accd[d][b][1] = 0.f; accd[d][b][2] = 0.f; accd[d][b][3] = 0.f; }
    }
    // ---- first layer ---------------------------------------------------------------------------
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 16] : 1.0f;
      const float* s0 = sm + k * nsm + 4 * g;
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dd = 0; dd < si; ++dd) s += xrow[dd] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
        const f32x4 tk = A.omega * s + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
        acc[b] += zt * tk;
#pragma unroll
        for (int d = 0; d < NS; ++d)
          if (d < ns && !(HESS && d == 2)) {
            if (J.seed[d] >= 0) accd[d][b] += (zt * A.omega) * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[d] * NP + 16 * b);
            else if (k < r) accd[d][b] += zd_base[(d * r + k) * 16] * tk;
          }
        if (HESS && anyp && k < r) {   // a''_0 = sum_k (z'_j,k dt_k/dx_k-seed + z'_k,k dt_k/dx_j-seed + z''_k t_k), t_k linear in x
          f32x4 t2 = zd_base[(2 * r + k) * 16] * tk;
          if (J.seed[1] >= 0) t2 += (zd_base[(0 * r + k) * 16] * A.omega) * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[1] * NP + 16 * b);
          if (J.seed[0] >= 0) t2 += (zd_base[(1 * r + k) * 16] * A.omega) * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[0] * NP + 16 * b);
          accd[2][b] += t2;
        }
      }
    }
    {
      f32x4 dv[NBL], d2[NBL];
      if (HESS) {
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) d2[b][v] = act_d2<ACT>(A.act, acc[b][v]);
      }
      act16<NBL, ACT>(A.act, acc, h, dv, n, g);
#pragma unroll
      for (int d = 0; d < NS; ++d)
#pragma unroll
        for (int b = 0; b < NBL; ++b) hd[d][b] = dv[b] * accd[d][b];
      if (HESS) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) hd[2][b] = dv[b] * accd[2][b] + d2[b] * accd[0][b] * accd[1][b];
      }
    }
    // ---- hidden hyper-matrices -------------------------------------------------------------------
    int pl = 0;
    f32x4 ublk[MODE == 1 ? NBL : 1], ublkd[MODE == 1 ? NS : 1][MODE == 1 ? NBL : 1];
    for (int j = 0; j < nh; ++j) {
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        acc[b][0] = 0.f; acc[b][1] = 0.f; acc[b][2] = 0.f; acc[b][3] = 0.f;
#pragma unroll
        for (int d = 0; d < NS; ++d) { accd[d][b][0] = 0.f; accd[d][b][1] = 0.f; accd[d][b][2] = 0.f; accd[d][b][3] = 0.f; }
      }
      for (int k = 0; k <= r; ++k) {
        const bool has_next = (pl + 1 < nplanes) || !last_group;
        f32x4 pre[PF4];
        if (has_next) {
          const f32x4* src = A.WF + (long)(pl + 1 < nplanes ? pl + 1 : 0) * (PLANE / 4);
#pragma unroll
          for (int q = 0; q < PF4; ++q)
            if (PEXACT || tid + NT * q < PLANE / 4) pre[q] = src[tid + NT * q];
        }
        const f32x4* cur = planes + (one_buf ? 0 : (gpar & 1)) * (PLANE / 4);
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        if (anyp && k < r) {
          // parameter seeds need the unscaled partial product T_k = h . M^(k)
          f32x4 Tk[NBL];
          mfma16<NBL, false>(cur, h, Tk, lane);
#pragma unroll
          for (int b = 0; b < NBL; ++b) acc[b] += zt * Tk[b];
#pragma unroll
          for (int d = 0; d < NS; ++d)
            if (d < ns && J.seed[d] < 0) {
              const float zd = zd_base[(d * r + k) * 16];
#pragma unroll
              for (int b = 0; b < NBL; ++b) accd[d][b] += zd * Tk[b];
            }
        } else {
          f32x4 hz[NBL];
#pragma unroll
          for (int b = 0; b < NBL; ++b) hz[b] = zt * h[b];
          mfma16<NBL, true>(cur, hz, acc, lane);
        }
#pragma unroll
        for (int d = 0; d < NS; ++d)
          if (d < ns) {
            if (HESS && anyp && k < r && d < 2) {
              // unscaled T_d = h'_d . M^(k): a'_d += zt T_d, and the pair's second-order stream takes z'_(other) T_d
              f32x4 Td[NBL];
              mfma16<NBL, false>(cur, hd[d], Td, lane);
              const float zo = zd_base[((1 - d) * r + k) * 16];
#pragma unroll
              for (int b = 0; b < NBL; ++b) { accd[d][b] += zt * Td[b]; accd[2][b] += zo * Td[b]; }
            } else {
              f32x4 hz[NBL];
#pragma unroll
              for (int b = 0; b < NBL; ++b) hz[b] = zt * hd[d][b];
              mfma16<NBL, true>(cur, hz, accd[d], lane);
            }
          }
        if (has_next) {
          if (one_buf) __syncthreads();     // every wave is done with the plane that is about to be overwritten
          f32x4* dst = planes + (one_buf ? 0 : ((gpar + 1) & 1)) * (PLANE / 4);
#pragma unroll
          for (int q = 0; q < PF4; ++q)
            if (PEXACT || tid + NT * q < PLANE / 4) dst[tid + NT * q] = pre[q];
        }
        __syncthreads();
        ++gpar; ++pl;
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) acc[b] *= A.omega;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) acc[b] += zt * *reinterpret_cast<const f32x4*>(sb + 16 * b);
      }
      // tangents: a' = w0 * accd (+ sum_k z'_k b^(k) for parameter seeds); then the layer's combination rule
#pragma unroll
      for (int d = 0; d < NS; ++d) {
#pragma unroll
        for (int b = 0; b < NBL; ++b) accd[d][b] *= A.omega;
        if (d < ns && J.seed[d] < 0)
          for (int k = 0; k < r; ++k) {
            const float zd = zd_base[(d * r + k) * 16];
            const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
            for (int b = 0; b < NBL; ++b) accd[d][b] += zd * *reinterpret_cast<const f32x4*>(sb + 16 * b);
          }
      }
      f32x4 dv[NBL], d2[NBL];
      if (HESS) {
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
          for (int v = 0; v < 4; ++v) d2[b][v] = act_d2<ACT>(A.act, acc[b][v]);
      }
      act16<NBL, ACT>(A.act, acc, acc, dv, n, g);
#pragma unroll
      for (int d = 0; d < NS; ++d)
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          f32x4 t = dv[b] * accd[d][b];
          if (HESS && d == 2) t += d2[b] * accd[0][b] * accd[1][b];
          if (MODE == 0) hd[d][b] = t;
          else if (MODE == 2) hd[d][b] += t;
          else {
            if (!(j & 1)) { ublkd[d][b] = hd[d][b]; hd[d][b] = t; }
            else hd[d][b] = 0.5f * (ublkd[d][b] + t);
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
    // ---- last layer: u and u' ---------------------------------------------------------------------
    for (int o = 0; o < so; ++o) {
      float part = 0.f, bias = 0.f, pd[NS];
#pragma unroll
      for (int d = 0; d < NS; ++d) pd[d] = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* s0 = sm + k * nsm;
        float sk = 0.f, skd[NS];
#pragma unroll
        for (int d = 0; d < NS; ++d) skd[d] = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
          sk += (h[b][0] * w[0] + h[b][1] * w[1]) + (h[b][2] * w[2] + h[b][3] * w[3]);
#pragma unroll
          for (int d = 0; d < NS; ++d)
            skd[d] += (hd[d][b][0] * w[0] + hd[d][b][1] * w[1]) + (hd[d][b][2] * w[2] + hd[d][b][3] * w[3]);
        }
        part = fmaf(zt, sk, part);
        bias = fmaf(zt, s0[o_bl + o], bias);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          pd[d] = fmaf(zt, skd[d], pd[d]);
          if (k < r && d < ns && J.seed[d] < 0) {
            const float zd = zd_base[(d * r + k) * 16];
            pd[d] = fmaf(zd, sk, pd[d]);
            if (g == 0) pd[d] = fmaf(zd, s0[o_bl + o], pd[d]);   // the bias term once per point (pd is summed over g)
          }
        }
        if (HESS && anyp && k < r)      // u'' += sum_k (z'_j,k <h'_k, Wl^(k)> + z'_k,k <h'_j, Wl^(k)>)
          pd[2] = fmaf(zd_base[(0 * r + k) * 16], skd[1], fmaf(zd_base[(1 * r + k) * 16], skd[0], pd[2]));
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
#pragma unroll
      for (int d = 0; d < NS; ++d) { pd[d] += __shfl_xor(pd[d], 16); pd[d] += __shfl_xor(pd[d], 32); }
      if (valid && g == 0) {
        if (A.u_out) A.u_out[pt * so + o] = part + bias;
        if (HESS) {
          J.dydx[(pt * so + o) * J.nx_total + J.hj] = pd[0];
          J.dydx[(pt * so + o) * J.nx_total + J.hk] = pd[1];
          J.d2ydx2[((pt * so + o) * J.nx_total + J.hj) * J.nx_total + J.hk] = pd[2];
          J.d2ydx2[((pt * so + o) * J.nx_total + J.hk) * J.nx_total + J.hj] = pd[2];
        } else {
#pragma unroll
          for (int d = 0; d < NS; ++d)
            if (d < ns) J.dydx[(pt * so + o) * J.nx_total + J.x0 + d] = pd[d];
        }
      }
    }
  }
}

static void launch_jac_impl(JacArgs& J, bool hess, hipStream_t st);
void launch_jac(const SNetArgs& a, int ns, const int* seeds, const float* const* zd, int nx_total, int x0, float* dydx,
                hipStream_t st) {
  JacArgs J;
  J.s = a; J.ns = ns; J.nx_total = nx_total; J.x0 = x0; J.dydx = dydx; J.hj = J.hk = 0; J.d2ydx2 = nullptr;
  for (int d = 0; d < NIF_JAC_MAXSEED; ++d) { J.seed[d] = d < ns ? seeds[d] : 0; J.ZD[d] = d < ns ? zd[d] : nullptr; }
  launch_jac_impl(J, false, st);
}
// one coordinate pair (seed_j at x position hj, seed_k at hk) of the Hessian: fills columns hj, hk of dydx and the entries
// (hj, hk), (hk, hj) of d2ydx2 [B][so][nx][nx]
void launch_hess(const SNetArgs& a, int seed_j, int seed_k, int hj, int hk, int nx_total, float* dydx, float* d2ydx2, hipStream_t st,
                 const float* zd_j, const float* zd_k, const float* zdd) {
  JacArgs J;
  J.s = a; J.ns = 3; J.nx_total = nx_total; J.x0 = 0; J.dydx = dydx; J.hj = hj; J.hk = hk; J.d2ydx2 = d2ydx2;
  J.seed[0] = seed_j; J.seed[1] = seed_k;
  J.seed[2] = zdd ? -1 : 0;          // stream 2 carries z'' like a parameter seed carries z' (bias / partial-product terms)
  J.ZD[0] = seed_j < 0 ? zd_j : nullptr; J.ZD[1] = seed_k < 0 ? zd_k : nullptr; J.ZD[2] = zdd;
  launch_jac_impl(J, true, st);
}
// shapes the Jacobian / Hessian kernels take: one 16-point-tile plane (and the small hyper-vectors) must fit the LDS
bool jac_supported(const SNetArgs& a) {
  if (a.n > 128) return false;
  const int NBL = snet3_nbl(a.n);
  const size_t plane = (size_t)NBL * NBL * 256;
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  return (plane + sm_tot + 4 * (size_t)(1 + NIF_JAC_MAXSEED) * a.r * 16 + 8) * sizeof(float) <= 160u * 1024u;
}
static void launch_jac_impl(JacArgs& J, bool hess, hipStream_t st) {
  const SNetArgs& a = J.s;
  const int NBL = snet3_nbl(a.n);
  const long nt16 = 2 * ((a.B + 31) / 32);
  const long ngroups = (nt16 + 3) / 4;
  const long cap = NBL <= 4 ? 512 : 256;
  dim3 grid((unsigned)(ngroups < cap ? ngroups : cap)), block(256);
  const size_t plane = (size_t)NBL * NBL * 256;
  const size_t sm_tot = (((size_t)(a.r + 1) * a.nsm) + 3) & ~(size_t)3;
  const size_t rest = sm_tot + 4 * (size_t)(1 + NIF_JAC_MAXSEED) * a.r * 16 + 8;
  J.one_buf = (2 * plane + rest) * sizeof(float) > 160u * 1024u ? 1 : 0;
  const size_t shm = ((J.one_buf ? 1 : 2) * plane + rest) * sizeof(float);
#define JL(NBL_, ACT_, MODE_)                                                                                        \
  {                                                                                                                  \
    if (hess) {                                                                                                      \
      if (shm > 48 * 1024)                                                                                           \
        (void)hipFuncSetAttribute((const void*)k_jac<NBL_, ACT_, MODE_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)shm);                                                                         \
      hipLaunchKernelGGL((k_jac<NBL_, ACT_, MODE_, true>), grid, block, shm, st, J);                                 \
    } else {                                                                                                         \
      if (shm > 48 * 1024)                                                                                           \
        (void)hipFuncSetAttribute((const void*)k_jac<NBL_, ACT_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)shm);                                                                         \
      hipLaunchKernelGGL((k_jac<NBL_, ACT_, MODE_>), grid, block, shm, st, J);                                       \
    }                                                                                                                \
  }
#define JK(NBL_)                                                      \
  if (a.nif_skip) JL(NBL_, -1, 2) else if (a.res) JL(NBL_, ACT_SINE, 1) else JL(NBL_, ACT_SINE, 0)
  switch (NBL) {
    case 1: JK(1) break;
    case 2: JK(2) break;
    case 3: JK(3) break;
    case 4: JK(4) break;
    case 6: JK(6) break;
    default: JK(8) break;
  }
#undef JK
#undef JL
}
