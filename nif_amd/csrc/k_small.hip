// k_small.hip (r6) -- loss and EVERY gradient of a small batch in ONE launch.
//
// configs[0] of BASELINE.json (tutorial/1: NIF 2 x 32 + ParameterNet 2 x 32, 10 k points, Model.fit with batch 512) ran its 512-point
// steps on the kernels built for 10^6 points: eleven launches whose EXECUTION times added up to 82 us (LDS images of
// the small vectors, weight planes split per workgroup, block reductions behind a dozen barriers -- fixed costs paid for 16 tiles).
// This kernel is the small-batch form of the same step (reference: NIF.call model.py:130-154, _call_shape_net :233-324,
// NIFMultiScale._call_shape_net_mres :738-954 plain form, ParameterNet layers mlp.py:148-160 / siren.py:256-281, Keras 'mse' with
// sample weights README.md:33, and the reverse sweep GradientTape builds over them -- SURVEY 8 a-1 .. a-10):
//   * plain fp32 FMAs, no plane packing, no MFMA operand forms: the hyper layer's r + 1 planes and the ParameterNet's matrices are
//     copied from theta into LDS as they lie (matrix rows padded to an odd stride: forward and transposed reads conflict free);
//   * a wave = two points, a lane = one feature (units <= 32): the ParameterNet, the plane formulation of the ShapeNet
//     (h W(p) = sum_k zt_k h M^(k)), the loss, the data adjoint and the ParameterNet adjoint of a point run inside its
//     half wave, layer inputs and dL/da rows of every layer kept in an LDS tape;
//   * behind ONE barrier the workgroup's 512 threads form every entry of the flat gradient over its 16 points from the tapes
//     (K = 16 sums in registers) and write ONE partial row per workgroup -- the existing k_reduce sums the rows in fixed order
//     (deterministic), k_adam / the RCCL all-reduce see the same [grad | loss] buffer as after the large-batch kernels.
// Shapes: class NIF (any Keras activation, skip connections) and plain-SIREN NIFMultiScale; ParameterNet MLP_SimpleShortCut / SIREN
// hidden layers; units <= 32 in both nets, <= 4 hidden matrices each, latent_dim <= 4, <= 4 inputs / outputs; float32 policy;
// batches <= NIF_SMALL_MAX_B points.  Everything else (and NIF_SMALL_STEP=0 / nif_set_option("small_step", 0)) keeps the tile kernels.
#include "nif_internal.h"

#define SMALL_T 16           // points per workgroup (8 waves x 2)
#define SMALL_NT 512
#define SMALL_MAXH 4
#define SMALL_MAXR 4

struct SmallArgs {
  PNetArgs p;
  SNetArgs s;
  float* partial; long pstride; long P;
};

struct SmallLay {            // LDS layout (floats), the same function on host and device
  int RS, RP;                // padded row strides of the n x n / nst x nst matrices
  int o_w1, o_wh, o_wl, o_b1, o_bh, o_bl, PS;                       // inside one ShapeNet plane
  int q_fw, q_fb, q_hw, q_hb, q_bw, q_bb, PW;                       // ParameterNet image
  int t_x, t_hp, t_dp, t_z, t_dz, t_hs, t_ds, t_du, TP;             // per-point tape
  int planes, pnet, tapes, total;
};
__host__ __device__ inline SmallLay small_layout(int pi, int nst, int lst, int r, int si, int so, int n, int nh) {
  SmallLay L;
  L.RS = n | 1; L.RP = nst | 1;
  L.o_w1 = 0; L.o_wh = si * n; L.o_wl = L.o_wh + nh * n * L.RS; L.o_b1 = L.o_wl + n * so; L.o_bh = L.o_b1 + n;
  L.o_bl = L.o_bh + nh * n; L.PS = L.o_bl + so;
  L.q_fw = 0; L.q_fb = pi * nst; L.q_hw = L.q_fb + nst; L.q_hb = L.q_hw + lst * nst * L.RP; L.q_bw = L.q_hb + lst * nst;
  L.q_bb = L.q_bw + nst * r; L.PW = L.q_bb + r;
  L.t_x = 0; L.t_hp = pi + si; L.t_dp = L.t_hp + (lst + 1) * nst; L.t_z = L.t_dp + (lst + 1) * nst; L.t_dz = L.t_z + r;
  L.t_hs = L.t_dz + r; L.t_ds = L.t_hs + (nh + 1) * n; L.t_du = L.t_ds + (nh + 1) * n; L.TP = (L.t_du + so + 1) | 1;
  L.planes = 0; L.pnet = (r + 1) * L.PS; L.tapes = L.pnet + L.PW; L.total = L.tapes + SMALL_T * L.TP + 16;
  return L;
}

__device__ __forceinline__ void small_act(int act, float a, float* h, float* d) {
  switch (act) {
    case ACT_SINE: nif_sincosf_core(a, h, d); break;
    case ACT_SWISH: act_eval<ACT_SWISH>(a, h, d); break;
    case ACT_TANH: act_eval<ACT_TANH>(a, h, d); break;
    case ACT_RELU: act_eval<ACT_RELU>(a, h, d); break;
    case ACT_SIGMOID: act_eval<ACT_SIGMOID>(a, h, d); break;
    case ACT_ELU: act_eval<ACT_ELU>(a, h, d); break;
    case ACT_SOFTPLUS: act_eval<ACT_SOFTPLUS>(a, h, d); break;
    case ACT_GELU: act_eval<ACT_GELU>(a, h, d); break;
    case ACT_SELU: act_eval<ACT_SELU>(a, h, d); break;
    case ACT_SOFTSIGN: act_eval<ACT_SOFTSIGN>(a, h, d); break;
    case ACT_EXPONENTIAL: act_eval<ACT_EXPONENTIAL>(a, h, d); break;
    case ACT_HARD_SIGMOID: act_eval<ACT_HARD_SIGMOID>(a, h, d); break;
    default: *h = a; *d = 1.0f; break;
  }
}
// sum over the 32 lanes of a half wave (every lane of the half gets the total)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(SMALL_NT) void k_small(SmallArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sml[];
  const PNetArgs& P = A.p;
  const SNetArgs& S = A.s;
  const int pi = P.pi, nst = P.nst, lst = P.lst, r = P.r, si = S.si, so = S.so, n = S.n, nh = S.nh;
  const SmallLay L = small_layout(pi, nst, lst, r, si, so, n, nh);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hf = lane >> 5, f = lane & 31;
  float* planes = sml;
  float* pw = sml + L.pnet;
  float* tapes = sml + L.tapes;
  float* lred = tapes + SMALL_T * L.TP;
  const float* th = P.theta;

  // ---- LDS images: plane k of the hyper layer (k < r: row k of the kernel, k = r: the bias), matrix rows at stride RS ------------
  for (int k = 0; k <= r; ++k) {
    const float* src = k < r ? th + S.off_Wh + (long)k * S.po : th + S.off_bh;
    float* dst = planes + k * L.PS;
    const int s_wh = si * n, s_wl = s_wh + nh * n * n;
    for (int e = tid; e < (int)S.po; e += SMALL_NT) {
      int d;
      if (e < s_wh) d = L.o_w1 + e;
      else if (e < s_wl) { const int q = e - s_wh, j = q / (n * n), ij = q - j * n * n, i = ij / n, o = ij - i * n; d = L.o_wh + j * n * L.RS + i * L.RS + o; }
      else d = L.o_wl + (e - s_wl);        // wl, b1, bh, bl follow contiguously in both layouts
      dst[d] = src[e];
    }
  }
  for (int e = tid; e < pi * nst; e += SMALL_NT) pw[L.q_fw + e] = th[P.first_w + e];
  for (int e = tid; e < nst; e += SMALL_NT) pw[L.q_fb + e] = th[P.first_b + e];
  for (int m = 0; m < lst; ++m) {
    for (int e = tid; e < nst * nst; e += SMALL_NT) { const int i = e / nst, o = e - i * nst; pw[L.q_hw + m * nst * L.RP + i * L.RP + o] = th[P.hid_w[m] + e]; }
    for (int e = tid; e < nst; e += SMALL_NT) pw[L.q_hb + m * nst + e] = th[P.hid_b[m] + e];
  }
  for (int e = tid; e < nst * r; e += SMALL_NT) pw[L.q_bw + e] = th[P.bott_w + e];
  for (int e = tid; e < r; e += SMALL_NT) pw[L.q_bb + e] = th[P.bott_b + e];
  __syncthreads();

  // ---- one point per half wave ---------------------------------------------------------------------------------------------
  const int pl = 2 * wid + hf;                          // point of this half wave inside the workgroup
  const long pt = (long)blockIdx.x * SMALL_T + pl;
  const bool valid = pt < S.B;
  const long ptc = valid ? pt : S.B - 1;
  float* tp = tapes + pl * L.TP;
  const int ncol = P.ncol;
  if (f < ncol) tp[L.t_x + f] = P.xin[ptc * ncol + f];
  const float* xp = tp + L.t_x;                         // ParameterNet inputs: columns 0 .. pi-1, coordinates behind them
  const float* xs = tp + L.t_x + S.col0;
  const float om_p = P.omega, om_s = S.omega;
  const bool fp = f < nst, fs = f < n;
  float loss_lane = 0.f;
  {
    // ParameterNet forward
    float h = 0.f, dd = 0.f;
    {
      float a = 0.f;
      for (int d = 0; d < pi; ++d) a = fmaf(xp[d], pw[L.q_fw + d * nst + (fp ? f : 0)], a);
      a = fmaf(om_p, a, pw[L.q_fb + (fp ? f : 0)]);
      small_act(P.act, a, &h, &dd);
      if (!fp) { h = 0.f; dd = 0.f; }
    }
    if (fp) { tp[L.t_hp + f] = h; tp[L.t_dp + f] = dd; }
    for (int m = 0; m < lst; ++m) {
      const float* w = pw + L.q_hw + m * nst * L.RP + (fp ? f : 0);
      const float* hin = tp + L.t_hp + m * nst;
      float a = 0.f;
      for (int i = 0; i < nst; ++i) a = fmaf(hin[i], w[i * L.RP], a);
      a = fmaf(om_p, a, pw[L.q_hb + m * nst + (fp ? f : 0)]);
      float t, d;
      small_act(P.act, a, &t, &d);
      h = P.siren ? t : h + t;
      if (fp) { tp[L.t_hp + (m + 1) * nst + f] = h; tp[L.t_dp + (m + 1) * nst + f] = d; }
    }
    for (int c = 0; c < r; ++c) {          // the latent: every lane of the half gets it; kept in the tape (zt_k = tp[t_z + k], zt_r = 1)
      const float v = half_sum(fp ? h * pw[L.q_bw + f * r + c] : 0.f) + pw[L.q_bb + c];
      if (f == 0) tp[L.t_z + c] = v;
    }
    const float* ztp = tp + L.t_z;
#define ZT(k_) ((k_) < r ? ztp[k_] : 1.0f)
    // ShapeNet forward (plane formulation: the per-point matrix is never formed)
    float hs = 0.f;
    {
      float a = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float* pk = planes + k * L.PS;
        float s = 0.f;
        for (int d = 0; d < si; ++d) s = fmaf(xs[d], pk[L.o_w1 + d * n + (fs ? f : 0)], s);
        s = fmaf(om_s, s, pk[L.o_b1 + (fs ? f : 0)]);
        a = fmaf(ZT(k), s, a);
      }
      float d;
      small_act(S.act, a, &hs, &d);
      if (fs) { tp[L.t_hs + f] = hs; tp[L.t_ds + f] = d; }
    }
    for (int j = 0; j < nh; ++j) {
      const float* hin = tp + L.t_hs + j * n;
      float a = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float* w = planes + k * L.PS + L.o_wh + j * n * L.RS + (fs ? f : 0);
        float s = 0.f;
        for (int i = 0; i < n; ++i) s = fmaf(hin[i], w[i * L.RS], s);
        s = fmaf(om_s, s, planes[k * L.PS + L.o_bh + j * n + (fs ? f : 0)]);
        a = fmaf(ZT(k), s, a);
      }
      float t, d;
      small_act(S.act, a, &t, &d);
      hs = S.nif_skip ? t + hs : t;
      if (fs) { tp[L.t_hs + (j + 1) * n + f] = hs; tp[L.t_ds + (j + 1) * n + f] = d; }
    }
    // last layer, loss, dL/du
    const float wsamp = valid ? (S.sw ? S.sw[ptc] : 1.0f) : 0.0f;
    float gh = 0.f;
    float dz[SMALL_MAXR] = {0.f, 0.f, 0.f, 0.f};
    float se = 0.f;
    for (int o = 0; o < so; ++o) {
      float part = 0.f, wg = 0.f, sk[SMALL_MAXR] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k <= SMALL_MAXR; ++k) {
        if (k <= r) {
          const float* pk = planes + k * L.PS;
          const float w = fs ? pk[L.o_wl + f * so + o] : 0.f;
          float s = hs * w;
          if (f == 0) s += pk[L.o_bl + o];
          const float z = ZT(k);
          part = fmaf(z, s, part);
          wg = fmaf(z, w, wg);
          if (k < SMALL_MAXR && k < r) sk[k < SMALL_MAXR ? k : 0] = s;
        }
      }
      const float uo = half_sum(part);
      const float e = uo - S.y[ptc * so + o];
      NIF_LOSS_ACC(S.loss_kind, e, se, dfac)
      const float du = dfac * wsamp * S.inv_bg / (float)so;
      if (f == 0) tp[L.t_du + o] = du;
      gh = fmaf(du, wg, gh);
#pragma unroll
      for (int k = 0; k < SMALL_MAXR; ++k) dz[k] = fmaf(du, sk[k], dz[k]);
    }
    if (f == 0) loss_lane = wsamp * se / (float)so * S.inv_bg;
    // ShapeNet adjoint
    for (int j = nh - 1; j >= 0; --j) {
      const float da = fs ? gh * tp[L.t_ds + (j + 1) * n + f] : 0.f;
      if (fs) tp[L.t_ds + (j + 1) * n + f] = da;
      const float* dar = tp + L.t_ds + (j + 1) * n;
      const float hin = fs ? tp[L.t_hs + j * n + f] : 0.f;
      float gin = S.nif_skip ? gh : 0.f;
#pragma unroll
      for (int k = 0; k <= SMALL_MAXR; ++k) {
        if (k <= r) {
          const float* w = planes + k * L.PS + L.o_wh + j * n * L.RS + (fs ? f : 0) * L.RS;
          float v = 0.f;
          for (int o = 0; o < n; ++o) v = fmaf(w[o], dar[o], v);
          v *= om_s;
          gin = fmaf(ZT(k), v, gin);
          if (k < SMALL_MAXR && k < r) dz[k < SMALL_MAXR ? k : 0] += hin * v + da * planes[k * L.PS + L.o_bh + j * n + (fs ? f : 0)];
        }
      }
      gh = fs ? gin : 0.f;
    }
    {
      const float da = fs ? gh * tp[L.t_ds + f] : 0.f;
      if (fs) tp[L.t_ds + f] = da;
#pragma unroll
      for (int k = 0; k < SMALL_MAXR; ++k) {
        if (k < r) {
          const float* pk = planes + k * L.PS;
          float s = 0.f;
          for (int d = 0; d < si; ++d) s = fmaf(xs[d], pk[L.o_w1 + d * n + (fs ? f : 0)], s);
          s = fmaf(om_s, s, pk[L.o_b1 + (fs ? f : 0)]);
          dz[k] += da * s;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < SMALL_MAXR; ++c) {
      if (c < r) {
        dz[c] = half_sum(dz[c]);
        if (f == 0) tp[L.t_dz + c] = dz[c];
      }
    }
    // ParameterNet adjoint
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < SMALL_MAXR; ++c)
      if (c < r) g = fmaf(dz[c], fp ? pw[L.q_bw + f * r + c] : 0.f, g);
    for (int m = lst - 1; m >= 0; --m) {
      const float da = fp ? g * tp[L.t_dp + (m + 1) * nst + f] : 0.f;
      if (fp) tp[L.t_dp + (m + 1) * nst + f] = da;
      const float* dar = tp + L.t_dp + (m + 1) * nst;
      const float* w = pw + L.q_hw + m * nst * L.RP + (fp ? f : 0) * L.RP;
      float v = 0.f;
      for (int o = 0; o < nst; ++o) v = fmaf(w[o], dar[o], v);
      g = P.siren ? om_p * v : g + om_p * v;
      if (!fp) g = 0.f;
    }
    if (fp) tp[L.t_dp + f] = g * tp[L.t_dp + f];
  }
#undef ZT
  {   // loss partial of the workgroup (fixed order)
    float v = loss_lane + __shfl_xor(loss_lane, 32);
    if (lane == 0) lred[wid] = v;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < SMALL_NT / 64; ++w) s += lred[w];
    S.loss_partial[blockIdx.x] = s;
  }

  // ---- every entry of the flat gradient over the workgroup's points: sum_p zt_k(p) A(p) B(p) ---------------------------------------
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
  const int TP = L.TP;
  for (long e = tid; e < A.P; e += SMALL_NT) {
    int oa = -1, ob = -1, kk = -1;       // tape offsets of the two operands (-1: the constant 1), latent index (-1: none)
    float scale = 1.0f;
    long q;
    if ((q = e - P.first_w) >= 0 && q < (long)pi * nst) { oa = L.t_x + (int)(q / nst); ob = L.t_dp + (int)(q % nst); scale = om_p; }
    else if ((q = e - P.first_b) >= 0 && q < nst) { ob = L.t_dp + (int)q; }
    else if ((q = e - P.bott_w) >= 0 && q < (long)nst * r) { oa = L.t_hp + lst * nst + (int)(q / r); ob = L.t_dz + (int)(q % r); }
    else if ((q = e - P.bott_b) >= 0 && q < r) { ob = L.t_dz + (int)q; }
    else {
      bool hit = false;
      for (int m = 0; m < lst && !hit; ++m) {
        if ((q = e - P.hid_w[m]) >= 0 && q < (long)nst * nst) { oa = L.t_hp + m * nst + (int)(q / nst); ob = L.t_dp + (m + 1) * nst + (int)(q % nst); scale = om_p; hit = true; }
        else if ((q = e - P.hid_b[m]) >= 0 && q < nst) { ob = L.t_dp + (m + 1) * nst + (int)q; hit = true; }
      }
      if (!hit) {
        long s;
        if ((q = e - S.off_Wh) >= 0 && q < (long)r * S.po) { kk = (int)(q / S.po); s = q - (long)kk * S.po; }
        else if ((q = e - S.off_bh) >= 0 && q < S.po) { s = q; }
        else { prow[e] = 0.f; continue; }     // (not a tensor of this model: cannot happen)
        const long s_wh = (long)si * n, s_wl = s_wh + (long)nh * n * n, s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
        if (s < s_wh) { oa = L.t_x + S.col0 + (int)(s / n); ob = L.t_ds + (int)(s % n); scale = om_s; }
        else if (s < s_wl) { const long u = s - s_wh; const int j = (int)(u / (n * n)), ij = (int)(u - (long)j * n * n); oa = L.t_hs + j * n + ij / n; ob = L.t_ds + (j + 1) * n + ij % n; scale = om_s; }
        else if (s < s_b1) { const int u = (int)(s - s_wl); oa = L.t_hs + nh * n + u / so; ob = L.t_du + u % so; }
        else if (s < s_bh) { ob = L.t_ds + (int)(s - s_b1); }
        else if (s < s_bl) { ob = L.t_ds + n + (int)(s - s_bh); }
        else { ob = L.t_du + (int)(s - s_bl); }
      }
    }
    float acc = 0.f;
#pragma unroll 4
    for (int p_ = 0; p_ < SMALL_T; ++p_) {
      const float* t = tapes + p_ * TP;
      float v = t[ob];
      if (oa >= 0) v *= t[oa];
      if (kk >= 0) v *= t[L.t_z + kk];
      acc += v;
    }
    prow[e] = scale * acc;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
bool small_supported(const PNetArgs& p, const SNetArgs& s) {
  if (p.ll_kind || p.res || s.res || s.ll) return false;
  if (s.prec != 0) return false;
  if (s.n > 32 || p.nst > 32 || s.nh > SMALL_MAXH || p.lst > SMALL_MAXH || p.r > SMALL_MAXR || p.r < 1) return false;
  if (s.si > 4 || s.so > 4 || p.pi > 4 || s.si < 1 || p.pi < 1 || s.nh < 0 || p.lst < 0) return false;
  if (p.ncol > 32) return false;
  const SmallLay L = small_layout(p.pi, p.nst, p.lst, p.r, s.si, s.so, s.n, s.nh);
  return (size_t)L.total * sizeof(float) <= 128u * 1024u;      // (one workgroup per CU is plenty for <= 128 workgroups)
}
int small_rows(long B) { return (int)((B + SMALL_T - 1) / SMALL_T); }
// one workgroup per 16 points; partial rows [small_rows(B)][pstride], loss partials [small_rows(B)]
void launch_small(const PNetArgs& p, const SNetArgs& s, float* partial, long pstride, long P, hipStream_t st) {
  SmallArgs A; A.p = p; A.s = s; A.partial = partial; A.pstride = pstride; A.P = P;
  const SmallLay L = small_layout(p.pi, p.nst, p.lst, p.r, s.si, s.so, s.n, s.nh);
  const size_t shm = (size_t)L.total * sizeof(float);
  if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL(k_small, dim3((unsigned)small_rows(s.B)), dim3(SMALL_NT), shm, st, A);
}
