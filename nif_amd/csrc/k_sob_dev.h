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
// NIFMultiScale with or without resblocks and class NIF (MODE 2: any Keras activation f with skip
// connections; the ring then holds c = f'(a) and -f''(a) in place of cos / sin).
// PAR: x_index may also address ParameterNet inputs (JacobianLayer takes any column, gradient.py:207-231).  Such a stream
// carries zt'_k = dz_k/dp_c (k < r; the constant plane has none) next to the latent, and every product gets its product rule:
//     a'  = w0 sum_k (zt_k h' + zt'_k h) M^(k) + sum_k zt'_k b^(k)           (one MFMA operand per plane, as before)
//     lambda_in += w0 sum_k zt'_k M^(k) nu ,   dL/dzt'_k = w0 <h_in, M^(k) nu> + <nu, b^(k)>
//     dL/dM^(k) += w0 sum_p zt'_k h_in nu^T ,  dL/db^(k) += sum_p zt'_k nu   -> a second reduction over (h_in, nu) with zt'
//                                                                               in the latent's place (nif_api.hip)
#pragma once
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
  int gcol[NIF_SOB_MAXSEED];    // column of gt / JU that stream d fills (streams are ordered coordinates first, x_index is not)
  // parameter seeds (PAR instantiations): stream d differentiates w.r.t. ParameterNet input par[d] (< 0: a coordinate seed)
  int par[NIF_SOB_MAXSEED];
  const float* ZT;              // z' = dz/dp_c of every parameter column [pi][tiles][r][32] (k_pjac, forward mode)
  float* DZT;                   // dL/dz' of stream d [ns][tiles][r][32] (rows of the parameter streams are written)
  int ll_plane;                 // LL: the epilogue's per-wave scratch lives in the idle plane buffer (wide nets: no LDS left for it)
  // LL with parameter columns: the ShapeNet does not see p, so such a column is no stream but one more contraction of the SAME
  // phi: du_i/dp_c = sum_k phi[i,k] a'_k with a' = (dz/dp_c) last_w ("head" e).  dL/da' -> DAT, dL/dz' -> DZT (block e), and a
  // zero-padded copy of z' in the latent-row layout of the r x r layer's gradient kernel -> ZTL
  int npar;                     // heads (0..3); the kernel then carries ns = coordinate streams only
  int nx_tot;                   // ns + npar = columns of gt / JU (LL epilogue)
  int parc[NIF_SOB_MAXSEED];    // parameter column of head e
  int pcol[NIF_SOB_MAXSEED];    // column of gt / JU that head e fills
  float* DAT;                   // [npar][tiles][rl][32]
  float* ZTL; int zl_rows;      // [npar][tiles][zl_rows][32]
  // r4 (gradient.py:207-231 takes any y_index / any number of x_index columns): the derivative term covers the outputs of ymask
  // only (ny of them) and a launch may carry one GROUP of the x_index columns -- gt / JU rows hold gstride columns, the term's
  // weight is wjn = w_jac / (ny * all columns) per squared error, and wu = 0 switches the primal mse off for the groups after the
  // first (host: nif_api.hip sobolev passes)
  float wu, wjn; unsigned ymask; int gstride;
  int one_buf;                  // 1: ONE plane buffer in LDS (shapes where two do not fit next to the small hyper-vectors): the next
                                // plane's DMA starts after the barrier behind the current plane's products, and is waited for
};

// parameter-seed instantiations (k_sob_par.hip)
void launch_sob_par(const SobArgs& J, bool train, bool bf, int nblk, size_t shm, hipStream_t st);
// two coordinate seeds of a plain SIREN net, the streams on separate waves (k_sobw.hip)
bool sobw_supported(const SNetArgs& a, int ns, bool any_par);
void launch_sobw(const SobArgs& J, int nblk, hipStream_t st, bool train = true);
int sobw_tiles_per_group(int n, int ns);
int sobw_grid_cap();
// last-layer class (k_sob_ll.hip)
void launch_sob_ll(const SobArgs& J, bool train, bool bf, int nblk, size_t shm, hipStream_t st);

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

#ifndef NIF_SOB_TWO_BF2
#define NIF_SOB_TWO_BF2 0   // two seeds under mixed_bfloat16 at two workgroups per CU (256 registers, 17 spilled): measured 4.09 -> 4.40 ms, off
#endif
// BF: 0 = f32-input MFMA planes, 1 = exact bf16 splits, 2 = one bf16 product (mixed_bfloat16 policy)
// NSD: seed streams the instantiation carries (register arrays and loops are sized by it): 1 or 2 seeds at n <= 64 leave room
// for TWO workgroups per CU (256 registers), the 3-seed form needs all 512
// LL: the last-layer-parameterised class (model.py:1044-1068, :1219-1269): SNetArgs as k_snet4 takes them for that class
// (fill_snet_ll: r = 0, one shared plane per layer, so = so_u * rl outputs phi, Z = the ParameterNet output a [tiles][rl][32]).
// u_i = sum_c phi[i*rl+c] a_c + bias_i and du_i/dx_d = sum_c phi'_d[i*rl+c] a_c; the adjoint starts from dphi = du (x) a,
// dphi'_d = du'_d (x) a and also yields dL/da (and dL/dlatent through the rl x rl map of the ParameterNet's last layer).
template <int NBL, int MODE, bool TRAIN, int BF, bool SGN, int NSD = NIF_SOB_MAXSEED, bool PAR = false, bool LL = false>
__global__ __launch_bounds__(256, ((NBL <= 2 && NSD <= 2) || (NBL <= 4 && NSD == 1) || (NBL <= 4 && NSD == 2 && BF == 2 && NIF_SOB_TWO_BF2)) ? 2 : (NBL <= 4 ? NIF_SOB_OCC : 1)) void k_sob(SobArgs J) {
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
  const int FP = stash_fp(n);                             // feature rows per stash tile (nif_internal.h)
  const float om_post = BF ? 1.0f : A.omega;              // the bf16-split planes hold omega_0 M (k_pack16b); the fp32 planes do not
  const long nt32 = (A.B + 31) / 32;
  const long nt16 = 2 * nt32;
  const long ngroups = (nt16 + WAVES - 1) / WAVES;

  f32x4* planes = reinterpret_cast<f32x4*>(smem);
  const int one_buf = J.one_buf;
  float* sm = smem + (one_buf ? 1 : 2) * PLANE;
  const int sm_tot = ((r + 1) * nsm + 3) & ~3;
  // PAR: the per-wave (dL/dzt, zt) block once more per PARAMETER stream (they are the last npar of the ns streams)
  int nsc = ns;
#pragma unroll
  for (int d = NS - 1; d >= 0; --d)
    if (PAR && d < ns && J.par[d] >= 0) nsc = d;
  const int npar_s = PAR ? ns - nsc : 0;
  const int NPW = 1 + npar_s;
  const int nrl = LL ? A.rl : 0, sou = LL ? A.so_u : so;
  // LL per wave: a, dL/da [rl][16], phi / phi' [NQ][so][16], du / du' [NQ][so_u][16]; heads: a', dL/da' [3][rl][16], du_e [3][so_u][16]
  const int llw = LL ? (2 * nrl + NQ * so + NQ * sou + NIF_SOB_MAXSEED * (2 * nrl + sou)) * 16 : 0;
  const int pwf = NPW * (r * 64 + r * 16) + ((LL && J.ll_plane) ? 0 : llw);   // per-wave floats
  float* dzs = sm + sm_tot + (long)wid * pwf;                 // per wave dz partials [r][64], latent [r][16]
  float* zs = dzs + r * 64;
  float* dzts = zs + r * 16 - nsc * r * 64;                   // [npar][r][64], indexed by the STREAM number d >= nsc
  float* zts = zs + r * 16 + npar_s * r * 64 - nsc * r * 16;  // [npar][r][16], likewise
  float* lsum = sm + sm_tot + (long)WAVES * pwf;
  const int o_llb = LL ? ((nsm - ((sou + 3) & ~3) - ((nrl * nrl + 3) & ~3))) : 0;   // LL extras sit at the end of sm (snet4_nsm_ll)
  const int o_lw = o_llb + ((sou + 3) & ~3);
  bool ispar[NS];
#pragma unroll
  for (int d = 0; d < NS; ++d) ispar[d] = PAR && d < ns && J.par[d] >= 0;
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
      else if (LL && e >= o_llb && e < o_llb + sou) v = hyp3(A, k, s_bl + so + (e - o_llb));
      else if (LL && e >= o_lw && e < o_lw + nrl * nrl) v = hyp3(A, k, s_bl + so + sou + (e - o_lw));
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

#define SOB_DMA_(DST_)                                                                         \
    {                                                                                         \
      const int nxt_ = pl + 1 < nplanes ? pl + 1 : 0;                                         \
      const f32x4* src = plane_src(nxt_);                                                     \
      const int nu_ = plane_units(nxt_);                                                      \
      f32x4* dst = (DST_);                                                                    \
      _Pragma("unroll") for (int q = 0; q < PF4; ++q)                                         \
        if (wid * 64 + NT * q < nu_)                                                          \
          __builtin_amdgcn_global_load_lds(                                                   \
              (const __attribute__((address_space(1))) void*)(src + tid + NT * q),            \
              (__attribute__((address_space(3))) void*)(dst + wid * 64 + NT * q), 16, 0, 0);  \
    }
#define SOB_PLANE(...)                                                                        \
  {                                                                                           \
    const bool has_next_ = (pl + 1 < nplanes) || !last_group;                                 \
    if (has_next_ && !one_buf) SOB_DMA_(planes + ((gpar + 1) & 1) * (PLANE / 4))              \
    const f32x4* cur = planes + (one_buf ? 0 : (gpar & 1)) * (PLANE / 4);                     \
    __VA_ARGS__                                                                               \
    __syncthreads();                                                                          \
    if (has_next_ && one_buf) { SOB_DMA_(planes) __syncthreads(); }                           \
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
    if (PAR && g == 0)
      for (int d = 0; d < NS; ++d)
        for (int k = 0; k < r; ++k)
          if (ispar[d]) zts[(d * r + k) * 16 + p] = J.ZT[(((long)J.par[d] * nt32 + tile32) * r + k) * 32 + poff];
    const float* zt_base = zs + p;
    const float* ztd_base = zts + p;                          // zt'_k of stream d at ztd_base[(d * r + k) * 16]
    // stash rows of stream q (0 = primal, 1+d = tangent d): pseudo-tile q*nt32 + tile32
    auto row0 = [&](int q) -> long { return ((long)q * nt32 + tile32) * (long)FP * 32 + poff; };
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;
    if (TRAIN && PAR)
      for (int e = nsc * r; e < ns * r; ++e) dzts[e * 64 + lane] = 0.f;

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
        const f32x4 t0 = A.omega * s + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
        aq[0][b] += zt * t0;
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          if (PAR && ispar[d]) { if (k < r) aq[1 + d][b] += ztd_base[(d * r + k) * 16] * t0; }
          else if (d < ns) aq[1 + d][b] += (zt * A.omega) * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[d] * NP + 16 * b);
        }
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
      // mixed_bfloat16 (BF = 2): every stream's tile is rounded ONCE per layer and the latent factor scales the product -- the cast
      // points of the plain step (k_snet4<PR>) and of k_sobw<PR>; streams with parameter seeds keep the combined operand below
      bf16x8 hb[(BF == 2 && !PAR) ? NQ : 1][(BF == 2 && !PAR) ? NCH : 1];
      if constexpr (BF == 2 && !PAR) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int ks = 0; ks < NCH; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) hb[q][ks][t] = (__bf16)hq[q][2 * ks + (t >> 2)][t & 3];
      }
      for (int k = 0; k <= r; ++k) {
        SOB_PLANE({
          const float zt = k < r ? zt_base[k * 16] : 1.0f;
          _Pragma("unroll") for (int q = 0; q < NQ; ++q)
            if (q <= ns) {
              if constexpr (BF == 2 && !PAR) {
                f32x4 T[NBL];
                _Pragma("unroll") for (int ks = 0; ks < NCH; ++ks) {
                  if (ks == 0) mfma_x6<NBL, true, true>(reinterpret_cast<const bf16x8*>(cur), hb[q][0], hb[q][0], hb[q][0], T, lane);
                  else mfma_x6<NBL, true>(reinterpret_cast<const bf16x8*>(cur) + ks * CF, hb[q][ks], hb[q][ks], hb[q][ks], T, lane);
                }
                _Pragma("unroll") for (int b = 0; b < NBL; ++b) aq[q][b] += zt * T[b];
                continue;
              }
              f32x4 hz[NBL];
              _Pragma("unroll") for (int b = 0; b < NBL; ++b) hz[b] = zt * hq[q][b];
              if (PAR && q > 0 && k < r && ispar[q > 0 ? q - 1 : 0]) {
                const float ztd = ztd_base[((q - 1) * r + k) * 16];
                _Pragma("unroll") for (int b = 0; b < NBL; ++b) hz[b] += ztd * hq[0][b];
              }
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
        for (int b = 0; b < NBL; ++b) aq[q][b] *= om_post;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 16] : 1.0f;
        const float* sb = sm + k * nsm + o_bh + j * NP + 4 * g;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
          aq[0][b] += zt * bb;
          if (PAR && k < r) {
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (ispar[d]) aq[1 + d][b] += ztd_base[(d * r + k) * 16] * bb;
          }
        }
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
    if constexpr (LL) {
      // per-wave scratch: its own LDS region, or (wide nets) a quarter of the plane buffer that is idle between the sweeps --
      // the forward sweep's last plane has been consumed behind a barrier, the next DMA into it is issued after this block
      float* lla = J.ll_plane ? reinterpret_cast<float*>(planes + ((gpar + 1) & 1) * (PLANE / 4)) + wid * llw
                              : dzs + NPW * (r * 64 + r * 16);
      float* llph = lla + nrl * 16;
      float* lldq = llph + NQ * so * 16;
      float* llda = lldq + NQ * sou * 16;
      float* llap = llda + nrl * 16;                         // a'_e [3][rl][16]
      float* lldap = llap + NIF_SOB_MAXSEED * nrl * 16;      // dL/da'_e
      float* lldqp = lldap + NIF_SOB_MAXSEED * nrl * 16;     // du_e [3][so_u][16]
      const int npar = J.npar, nxt = J.gstride;
      if (g == 0) {
        for (int cc = 0; cc < nrl; ++cc) lla[cc * 16 + p] = A.Z[(tile32 * nrl + cc) * 32 + poff];
        for (int e = 0; e < npar; ++e)
          for (int cc = 0; cc < nrl; ++cc) {
            float t = 0.f;
            for (int c2 = 0; c2 < nrl; ++c2)
              t = fmaf(J.ZT[(((long)J.parc[e] * nt32 + tile32) * nrl + c2) * 32 + poff], sm[o_lw + c2 * nrl + cc], t);
            llap[(e * nrl + cc) * 16 + p] = t;
          }
      }
      // phi[q][o] of the tile into LDS (one value per point: lanes g = 0 write, every lane of the point reads)
      for (int o = 0; o < so; ++o) {
        float part[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) part[q] = 0.f;
#pragma unroll
        for (int b = 0; b < NBL; ++b) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(sm + o_wl + o * NP + 16 * b + 4 * g);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            part[q] += (hq[q][b][0] * w[0] + hq[q][b][1] * w[1]) + (hq[q][b][2] * w[2] + hq[q][b][3] * w[3]);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) { part[q] += __shfl_xor(part[q], 16); part[q] += __shfl_xor(part[q], 32); }
        part[0] += sm[o_bl + o];
        if (g == 0) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) llph[(q * so + o) * 16 + p] = part[q];
        }
      }
      __builtin_amdgcn_wave_barrier();
      // u_i = <phi[i,:], a> + bias_i , u'_i = <phi'[i,:], a> ; loss ; du, du' (every lane of the point computes the same)
      for (int i = 0; i < sou; ++i) {
        float uq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float t = 0.f;
          for (int cc = 0; cc < nrl; ++cc) t = fmaf(llph[(q * so + i * nrl + cc) * 16 + p], lla[cc * 16 + p], t);
          uq[q] = t;
        }
        const float uo = uq[0] + sm[o_llb + i];
        if (valid && g == 0) {
          if (A.u_out) A.u_out[pt * sou + i] = uo;
          if (J.JU)
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (d < ns) J.JU[(pt * sou + i) * nxt + J.gcol[d]] = uq[1 + d];
        }
        float up[NIF_SOB_MAXSEED];
        for (int e = 0; e < npar; ++e) {
          float t = 0.f;
          for (int cc = 0; cc < nrl; ++cc) t = fmaf(llph[(i * nrl + cc) * 16 + p], llap[(e * nrl + cc) * 16 + p], t);
          up[e] = t;
          if (valid && g == 0 && J.JU) J.JU[(pt * sou + i) * nxt + J.pcol[e]] = t;
        }
        if (TRAIN) {
          float dq[NQ];
          const float e = uo - A.y[ptc * sou + i];
          NIF_LOSS_ACC(A.loss_kind, e, se, dfac0)
          dq[0] = J.wu * dfac0 * wsamp * A.inv_bg / (float)sou;
          const float ysel = ((J.ymask >> i) & 1u) ? 1.0f : 0.0f;
#pragma unroll
          for (int d = 0; d < NS; ++d) {
            dq[1 + d] = 0.f;
            if (d < ns) {
              const float ej = uq[1 + d] - J.gt[(ptc * sou + i) * nxt + J.gcol[d]];
              float vj = 0.f;
              NIF_LOSS_ACC(A.loss_kind, ej, vj, dfj)
              sej = fmaf(ysel, vj, sej);
              dq[1 + d] = ysel * dfj * J.wjn * wsamp * A.inv_bg;
            }
          }
          for (int e2 = 0; e2 < npar; ++e2) {
            const float ej = up[e2] - J.gt[(ptc * sou + i) * nxt + J.pcol[e2]];
            float vj = 0.f;
            NIF_LOSS_ACC(A.loss_kind, ej, vj, dfj)
            sej = fmaf(ysel, vj, sej);
            if (g == 0) lldqp[(e2 * sou + i) * 16 + p] = ysel * dfj * J.wjn * wsamp * A.inv_bg;
          }
          if (g == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) lldq[(q * sou + i) * 16 + p] = dq[q];
            if (active) A.DU[(tile32 * sou + i) * 32 + poff] = dq[0];        // -> last_layer_bias gradient
          }
        }
      }
      if (TRAIN) {
        __builtin_amdgcn_wave_barrier();
        // dL/da_c = sum_i sum_q du^q_i phi^q[i*rl+c] ; dL/dlatent_c' = sum_c dL/da_c last_w[c'][c] (a = latent last_w + last_b)
        if (g == 0) {
          for (int cc = 0; cc < nrl; ++cc) {
            float t = 0.f;
            for (int i = 0; i < sou; ++i)
#pragma unroll
              for (int q = 0; q < NQ; ++q)
                if (q <= ns) t = fmaf(lldq[(q * sou + i) * 16 + p], llph[(q * so + i * nrl + cc) * 16 + p], t);
            if (active) A.DA_ll[(tile32 * nrl + cc) * 32 + poff] = t;
            llda[cc * 16 + p] = t;
          }
          for (int c2 = 0; c2 < nrl; ++c2) {
            float t = 0.f;
            for (int cc = 0; cc < nrl; ++cc) t = fmaf(llda[cc * 16 + p], sm[o_lw + c2 * nrl + cc], t);
            if (active) A.DZL[(tile32 * nrl + c2) * 32 + poff] = t;
          }
          // heads: dL/da'_e = sum_i du_e,i phi[i,:] ; dL/dz'_e through the rl x rl map ; z'_e in latent-row layout
          for (int e = 0; e < npar; ++e) {
            for (int cc = 0; cc < nrl; ++cc) {
              float t = 0.f;
              for (int i = 0; i < sou; ++i) t = fmaf(lldqp[(e * sou + i) * 16 + p], llph[(i * nrl + cc) * 16 + p], t);
              if (active) J.DAT[(((long)e * nt32 + tile32) * nrl + cc) * 32 + poff] = t;
              lldap[(e * nrl + cc) * 16 + p] = t;
            }
            for (int c2 = 0; c2 < nrl; ++c2) {
              float t = 0.f;
              for (int cc = 0; cc < nrl; ++cc) t = fmaf(lldap[(e * nrl + cc) * 16 + p], sm[o_lw + c2 * nrl + cc], t);
              if (active) {
                J.DZT[(((long)e * nt32 + tile32) * nrl + c2) * 32 + poff] = t;
                J.ZTL[(((long)e * nt32 + tile32) * J.zl_rows + c2) * 32 + poff] =
                    valid ? J.ZT[(((long)J.parc[e] * nt32 + tile32) * nrl + c2) * 32 + poff] : 0.f;
              }
            }
          }
        }
        // dphi^q[o] = du^q_{o / rl} a_{o % rl} -> the phi layer's weight gradient (k_gw_out over real + pseudo tiles) and lambda
        for (int o = 0; o < so; ++o) {
          const int i = o / nrl, cc = o - i * nrl;
          const float av = lla[cc * 16 + p];
          float dph[NQ];
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            dph[q] = lldq[(q * sou + i) * 16 + p] * av;
            if (q == 0)
              for (int e = 0; e < npar; ++e) dph[0] = fmaf(lldqp[(e * sou + i) * 16 + p], llap[(e * nrl + cc) * 16 + p], dph[0]);
            if (q <= ns && active && g == 0) A.DPHI[(((long)q * nt32 + tile32) * so + o) * 32 + poff] = dph[q];
          }
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(sm + o_wl + o * NP + 16 * b + 4 * g);
#pragma unroll
            for (int q = 0; q < NQ; ++q) lam[q][b] += dph[q] * w;
          }
        }
      }
      if (J.ll_plane) __syncthreads();       // every wave is done with its scratch before the next plane's DMA lands there
    } else
    for (int o = 0; o < so; ++o) {
      f32x4 wg[NBL];
#pragma unroll
      for (int b = 0; b < NBL; ++b) ZERO4(wg[b]);
      float part[NQ], bias = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) part[q] = 0.f;
      f32x4 wgd[PAR ? NS : 1][PAR ? NBL : 1];       // PAR: sum_k zt'_k Wl^(k)[:,o] of each parameter stream
      float biasd[NS];
#pragma unroll
      for (int d = 0; d < NS; ++d) biasd[d] = 0.f;
      if (PAR) {
#pragma unroll
        for (int d = 0; d < NS; ++d)
#pragma unroll
          for (int b = 0; b < NBL; ++b) ZERO4(wgd[d][b]);
      }
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
          if (PAR && k < r) {
            const float hw = (hq[0][b][0] * w[0] + hq[0][b][1] * w[1]) + (hq[0][b][2] * w[2] + hq[0][b][3] * w[3]);
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (ispar[d]) {
                const float ztd = ztd_base[(d * r + k) * 16];
                part[1 + d] = fmaf(ztd, hw, part[1 + d]);
                if (TRAIN) wgd[d][b] += ztd * w;
              }
          }
        }
        bias = fmaf(zt, s0[o_bl + o], bias);
        if (PAR && k < r) {
#pragma unroll
          for (int d = 0; d < NS; ++d)
            if (ispar[d]) biasd[d] = fmaf(ztd_base[(d * r + k) * 16], s0[o_bl + o], biasd[d]);
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) { part[q] += __shfl_xor(part[q], 16); part[q] += __shfl_xor(part[q], 32); }
      const float uo = part[0] + bias;
      if (PAR) {
#pragma unroll
        for (int d = 0; d < NS; ++d) part[1 + d] += biasd[d];
      }
      if (valid && g == 0) {
        if (A.u_out) A.u_out[pt * so + o] = uo;
        if (J.JU)
#pragma unroll
          for (int d = 0; d < NS; ++d)
            if (d < ns) J.JU[(pt * so + o) * J.gstride + J.gcol[d]] = part[1 + d];
      }
      if (TRAIN) {
        float dq[NQ];
        const float e = uo - A.y[ptc * so + o];
        NIF_LOSS_ACC(A.loss_kind, e, se, dfac0)
        dq[0] = J.wu * dfac0 * wsamp * A.inv_bg / (float)so;
        const float ysel = ((J.ymask >> o) & 1u) ? 1.0f : 0.0f;
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          dq[1 + d] = 0.f;
          if (d < ns) {
            const float ej = part[1 + d] - J.gt[(ptc * so + o) * J.gstride + J.gcol[d]];
            float vj = 0.f;
            NIF_LOSS_ACC(A.loss_kind, ej, vj, dfj)
            sej = fmaf(ysel, vj, sej);
            dq[1 + d] = ysel * dfj * J.wjn * wsamp * A.inv_bg;
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q <= ns && active && g == 0) A.DU[(((long)q * nt32 + tile32) * so + o) * 32 + poff] = dq[q];
#pragma unroll
          for (int b = 0; b < NBL; ++b) lam[q][b] += dq[q] * wg[b];
        }
        if (PAR) {
#pragma unroll
          for (int d = 0; d < NS; ++d)
            if (ispar[d]) {
#pragma unroll
              for (int b = 0; b < NBL; ++b) lam[0][b] += dq[1 + d] * wgd[d][b];
            }
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
          if (PAR) {      // dL/dzt'_k += dq[1+d] * (<h, Wl^(k)[:,o]> + bl^(k)[o])
            float hw = 0.f;
#pragma unroll
            for (int b = 0; b < NBL; ++b) {
              const f32x4 w = *reinterpret_cast<const f32x4*>(s0 + o_wl + o * NP + 16 * b + 4 * g);
              hw += (hq[0][b][0] * w[0] + hq[0][b][1] * w[1]) + (hq[0][b][2] * w[2] + hq[0][b][3] * w[3]);
            }
            if (g == 0) hw += s0[o_bl + o];
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (ispar[d]) dzts[(d * r + k) * 64 + lane] += dq[1 + d] * hw;
          }
        }
      }
    }
    if (TRAIN) {
      if (g == 0) loss_lane += wsamp * A.inv_bg * (J.wu * se / (float)sou + J.wjn * sej);

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
            float dztk[NS];
            f32x4 hin0[PAR ? NBL : 1];
            _Pragma("unroll") for (int d = 0; d < NS; ++d) dztk[d] = 0.f;
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
                  if constexpr (PAR) {
                    if (q == 0) { _Pragma("unroll") for (int b = 0; b < NBL; ++b) hin0[b] = hin[b]; }
                    else if (ispar[q > 0 ? q - 1 : 0]) {
                      const float ztd = ztd_base[((q - 1) * r + k) * 16];
                      float t = 0.f;
                      _Pragma("unroll") for (int b = 0; b < NBL; ++b) {
                        lam[0][b] += ztd * U[b];
                        t += (hin0[b][0] * U[b][0] + hin0[b][1] * U[b][1]) + (hin0[b][2] * U[b][2] + hin0[b][3] * U[b][3]);
                      }
                      dztk[q > 0 ? q - 1 : 0] = t;
                    }
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
              dzs[k * 64 + lane] += fmaf(om_post, dzk, sbv);
              if (PAR) {
                _Pragma("unroll") for (int d = 0; d < NS; ++d)
                  if (ispar[d]) {
                    float sv = 0.f;
                    _Pragma("unroll") for (int b = 0; b < NBL; ++b) {
                      const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + 16 * b);
                      sv += (vq[1 + d][b][0] * bb[0] + vq[1 + d][b][1] * bb[1]) + (vq[1 + d][b][2] * bb[2] + vq[1 + d][b][3] * bb[3]);
                    }
                    dzts[(d * r + k) * 64 + lane] += fmaf(om_post, dztk[d], sv);
                  }
              }
            }
          })
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            lam[q][b] *= om_post;
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
          float sd[NS];
#pragma unroll
          for (int d = 0; d < NS; ++d) sd[d] = 0.f;
#pragma unroll
          for (int b = 0; b < NBL; ++b) {
            f32x4 xw = {0.f, 0.f, 0.f, 0.f};
            for (int dd = 0; dd < si; ++dd) xw += xrow[dd] * *reinterpret_cast<const f32x4*>(s0 + o_w1 + dd * NP + 16 * b);
            const f32x4 t = A.omega * xw + *reinterpret_cast<const f32x4*>(s0 + o_b1 + 16 * b);
            s += (vq[0][b][0] * t[0] + vq[0][b][1] * t[1]) + (vq[0][b][2] * t[2] + vq[0][b][3] * t[3]);
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (PAR && ispar[d]) {
                sd[d] += (vq[1 + d][b][0] * t[0] + vq[1 + d][b][1] * t[1]) + (vq[1 + d][b][2] * t[2] + vq[1 + d][b][3] * t[3]);
              } else if (d < ns) {
                const f32x4 wd = A.omega * *reinterpret_cast<const f32x4*>(s0 + o_w1 + J.seed[d] * NP + 16 * b);
                s += (vq[1 + d][b][0] * wd[0] + vq[1 + d][b][1] * wd[1]) + (vq[1 + d][b][2] * wd[2] + vq[1 + d][b][3] * wd[3]);
              }
          }
          float tot = dzs[k * 64 + lane] + s;
          tot += __shfl_xor(tot, 16);
          tot += __shfl_xor(tot, 32);
          if (active && g == 0) A.DZ[(tile32 * r + k) * 32 + poff] = tot;
          if (PAR) {
#pragma unroll
            for (int d = 0; d < NS; ++d)
              if (ispar[d]) {
                float td = dzts[(d * r + k) * 64 + lane] + sd[d];
                td += __shfl_xor(td, 16);
                td += __shfl_xor(td, 32);
                if (active && g == 0) J.DZT[(((long)d * nt32 + tile32) * r + k) * 32 + poff] = td;
              }
          }
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

