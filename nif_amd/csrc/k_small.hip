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
#include <vector>

#define SMALL_T 16           // points per workgroup (8 waves x 2)
#define SMALL_NT 512
#define SMALL_MAXH 4
#define SMALL_MAXR 4

struct SmallArgs {
  PNetArgs p;
  SNetArgs s;
  float* partial; long pstride; long P;
  const int* idx;          // [L.tapes]: theta index of every word of the padded LDS images (-1: padding = 0); built once per context
  const int* desc;         // [ntasks][SMALL_DW]: the tensor descriptors of the gradient phase
  double* metric; float metric_w; const float* g_loss;      // a pending nif_metric_accumulate of the PREVIOUS step (g_loss = &grad[P]) or null
};

// LDS layout (floats), the same function on host and device.  Every feature dimension is padded to SMALL_W = 32 (zeros), matrix rows to
// 33 floats: all matvec loops have the compile-time trip count 32 (fully unrolled: the LDS reads of a product are in flight together --
// the first form, with run-time trip counts, spent 54 us on 512 points, one LDS round trip per FMA), forward (column) and transposed
// (row) reads of a matrix are both bank-conflict free, and lanes / rows beyond the net's width contribute exact zeros
#define SMALL_W 32
#define SMALL_RS 33
struct SmallLay {
  int o_w1, o_wh, o_wl, o_b1, o_bh, o_bl, PS;                       // inside one ShapeNet plane
  int q_fw, q_fb, q_hw, q_hb, q_bw, q_bb, PW;                       // ParameterNet image
  int t_x, t_hp, t_dp, t_z, t_dz, t_hs, t_ds, t_du, TP;             // per-point tape
  int pnet, tapes, total;
};
__host__ __device__ inline SmallLay small_layout(int pi, int lst, int r, int si, int nh) {
  SmallLay L;
  L.o_w1 = 0; L.o_wh = si * SMALL_W; L.o_wl = L.o_wh + nh * SMALL_W * SMALL_RS; L.o_b1 = L.o_wl + SMALL_W * 4; L.o_bh = L.o_b1 + SMALL_W;
  L.o_bl = L.o_bh + nh * SMALL_W; L.PS = L.o_bl + 4;
  L.q_fw = 0; L.q_fb = pi * SMALL_W; L.q_hw = L.q_fb + SMALL_W; L.q_hb = L.q_hw + lst * SMALL_W * SMALL_RS; L.q_bw = L.q_hb + lst * SMALL_W;
  L.q_bb = L.q_bw + SMALL_W * 4; L.PW = L.q_bb + 4;
  L.t_x = 0; L.t_hp = 8; L.t_dp = L.t_hp + (lst + 1) * SMALL_W; L.t_z = L.t_dp + (lst + 1) * SMALL_W; L.t_dz = L.t_z + 4;
  L.t_hs = L.t_dz + 4; L.t_ds = L.t_hs + (nh + 1) * SMALL_W; L.t_du = L.t_ds + (nh + 1) * SMALL_W; L.TP = L.t_du + 4;
  if ((L.TP & 63) == 0) L.TP += 4;     // (two points of a wave read their tapes in one instruction: not the same banks)
  L.pnet = (r + 1) * L.PS; L.tapes = L.pnet + L.PW; L.total = L.tapes + SMALL_T * L.TP + 16;
  return L;
}

// ONE copy of the activation switch in the code object (five call sites): the kernel runs every instruction about once, so its time is
// largely instruction FETCH -- the first forms (everything inlined and unrolled: 53 KB of code) spent 4-5 k cycles per phase whatever
// the phase computed
struct SmallHD { float h, d; };
__device__ __attribute__((noinline)) SmallHD small_act(int act, float a) {
  float h, d;
  switch (act) {
    case ACT_SINE: nif_sincosf_core(a, &h, &d); break;
    case ACT_SWISH: act_eval<ACT_SWISH>(a, &h, &d); break;
    case ACT_TANH: act_eval<ACT_TANH>(a, &h, &d); break;
    case ACT_RELU: act_eval<ACT_RELU>(a, &h, &d); break;
    case ACT_SIGMOID: act_eval<ACT_SIGMOID>(a, &h, &d); break;
    case ACT_ELU: act_eval<ACT_ELU>(a, &h, &d); break;
    case ACT_SOFTPLUS: act_eval<ACT_SOFTPLUS>(a, &h, &d); break;
    case ACT_GELU: act_eval<ACT_GELU>(a, &h, &d); break;
    case ACT_SELU: act_eval<ACT_SELU>(a, &h, &d); break;
    case ACT_SOFTSIGN: act_eval<ACT_SOFTSIGN>(a, &h, &d); break;
    case ACT_EXPONENTIAL: act_eval<ACT_EXPONENTIAL>(a, &h, &d); break;
    case ACT_HARD_SIGMOID: act_eval<ACT_HARD_SIGMOID>(a, &h, &d); break;
    default: h = a; d = 1.0f; break;
  }
  SmallHD o; o.h = h; o.d = d;
  return o;
}
// sum over the 32 lanes of a half wave (every lane of the half gets the total)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// sum_i vec[i] * col[i * stride], 32 terms, two chains; vec is a 16-byte aligned tape row (broadcast reads of four values)
__device__ __forceinline__ float small_dot32(const float* __restrict__ vec, const float* __restrict__ col, int stride) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < SMALL_W; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(vec + i);
    s0 = fmaf(v[0], col[(i + 0) * stride], s0); s1 = fmaf(v[1], col[(i + 1) * stride], s1);
    s0 = fmaf(v[2], col[(i + 2) * stride], s0); s1 = fmaf(v[3], col[(i + 3) * stride], s1);
  }
  return s0 + s1;
}

// the model's tensors, one descriptor each (Keras order of the ParameterNet's, then the hyper layer plane by plane): where the tensor
// lies in theta, where its padded image goes in LDS, and which tape rows make its gradient (A_i, B_o, latent index)
#define SMALL_DW 9
#define SMALL_MAXT 80
__host__ __device__ inline int small_ntasks(int lst, int r, int nh) { return 4 + 2 * lst + (r + 1) * (4 + 2 * nh); }
__host__ __device__ inline void small_desc(int g, const PNetArgs& P, const SNetArgs& S, const SmallLay& L, int* D) {
  const int pi = P.pi, nst = P.nst, lst = P.lst, r = P.r, si = S.si, so = S.so, n = S.n, nh = S.nh;
  long off = 0; int rows = 1, cols = 1, img = 0, ds = SMALL_W, oa = -1, ob = 0, kk = -1; float sc = 1.0f;
  const int pb = L.pnet;
  if (g == 0) { off = P.first_w; rows = pi; cols = nst; img = pb + L.q_fw; oa = L.t_x; ob = L.t_dp; sc = P.omega; }
  else if (g == 1) { off = P.first_b; cols = nst; img = pb + L.q_fb; ob = L.t_dp; }
  else if (g < 2 + 2 * lst) {
    const int m = (g - 2) >> 1;
    if (((g - 2) & 1) == 0) { off = P.hid_w[m]; rows = nst; cols = nst; img = pb + L.q_hw + m * SMALL_W * SMALL_RS; ds = SMALL_RS; oa = L.t_hp + m * SMALL_W; ob = L.t_dp + (m + 1) * SMALL_W; sc = P.omega; }
    else { off = P.hid_b[m]; cols = nst; img = pb + L.q_hb + m * SMALL_W; ob = L.t_dp + (m + 1) * SMALL_W; }
  }
  else if (g == 2 + 2 * lst) { off = P.bott_w; rows = nst; cols = r; img = pb + L.q_bw; ds = 4; oa = L.t_hp + lst * SMALL_W; ob = L.t_dz; }
  else if (g == 3 + 2 * lst) { off = P.bott_b; cols = r; img = pb + L.q_bb; ds = 4; ob = L.t_dz; }
  else {
    const int per = 4 + 2 * nh, q = g - (4 + 2 * lst), k = q / per, u = q - k * per;
    const long tb = k < r ? S.off_Wh + (long)k * S.po : S.off_bh;
    const int ib = k * L.PS;
    kk = k < r ? k : -1;
    const long s_wl = (long)si * n + (long)nh * n * n, s_b1 = s_wl + (long)n * so;
    if (u == 0) { off = tb; rows = si; cols = n; img = ib + L.o_w1; oa = L.t_x + S.col0; ob = L.t_ds; sc = S.omega; }
    else if (u <= nh) { const int j = u - 1; off = tb + (long)si * n + (long)j * n * n; rows = n; cols = n; img = ib + L.o_wh + j * SMALL_W * SMALL_RS; ds = SMALL_RS; oa = L.t_hs + j * SMALL_W; ob = L.t_ds + (j + 1) * SMALL_W; sc = S.omega; }
    else if (u == nh + 1) { off = tb + s_wl; rows = n; cols = so; img = ib + L.o_wl; ds = 4; oa = L.t_hs + nh * SMALL_W; ob = L.t_du; }
    else if (u == nh + 2) { off = tb + s_b1; cols = n; img = ib + L.o_b1; ob = L.t_ds; }
    else if (u < 2 * nh + 3) { const int j = u - nh - 3; off = tb + s_b1 + n + (long)j * n; cols = n; img = ib + L.o_bh + j * SMALL_W; ob = L.t_ds + (j + 1) * SMALL_W; }
    else { off = tb + s_b1 + n + (long)nh * n; cols = so; img = ib + L.o_bl; ds = 4; ob = L.t_du; }
  }
  D[0] = (int)off; D[1] = rows; D[2] = cols; D[3] = img; D[4] = ds; D[5] = oa; D[6] = ob; D[7] = kk; { union { float f; int i; } cv; cv.f = sc; D[8] = cv.i; }
}

#ifdef NIF_TIMELINE
#define SM_TL(i_) do { if (A.s.tl && blockIdx.x == 0 && threadIdx.x == 0) A.s.tl[i_] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SM_TL(i_) do { } while (0)
#endif
__global__ __launch_bounds__(SMALL_NT) void k_small(SmallArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sml[];
  SM_TL(0);
  const PNetArgs& P = A.p;
  const SNetArgs& S = A.s;
  const int pi = P.pi, nst = P.nst, lst = P.lst, r = P.r, si = S.si, so = S.so, n = S.n, nh = S.nh;
  const SmallLay L = small_layout(pi, lst, r, si, nh);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hf = lane >> 5, f = lane & 31;
  float* planes = sml;
  float* pw = sml + L.pnet;
  float* tapes = sml + L.tapes;
  float* lred = tapes + SMALL_T * L.TP;
  const float* th = P.theta;
  const int ntasks = small_ntasks(lst, r, nh);
  const int* desc = A.desc;

  // ---- the padded LDS images in ONE gather: word e of the images is theta[idx[e]] (idx < 0: padding).  The index map is the context's
  // (built on the host once); two dependent global round trips with every load of a thread in flight -- the earlier forms copied theta
  // flat and cut the images tensor by tensor from the copy (11 k cycles for the two phases, most of them instruction fetch and LDS
  // round trips of code that runs once)
  {
    constexpr int NQ = 8;
    for (int e0 = 0; e0 < L.tapes; e0 += NQ * SMALL_NT) {
      int ix[NQ]; float v[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const int e = e0 + tid + q * SMALL_NT; ix[q] = e < L.tapes ? A.idx[e] : -1; }
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[q] = ix[q] >= 0 ? th[ix[q]] : 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const int e = e0 + tid + q * SMALL_NT; if (e < L.tapes) sml[e] = v[q]; }
    }
  }
  if (A.metric && blockIdx.x == 0 && tid == 0) {      // the previous step's Keras loss metric (nif_metric_accumulate, deferred to this launch)
    A.metric[0] += (double)A.metric_w * (double)A.g_loss[0];
    A.metric[1] += (double)A.metric_w;
  }
  const int sc = lane & 31, sg = lane >> 5;
  __syncthreads();
  SM_TL(2);

  // ---- one point per half wave ---------------------------------------------------------------------------------------------
  const int pl = 2 * wid + hf;                          // point of this half wave inside the workgroup
  const long pt = (long)blockIdx.x * SMALL_T + pl;
  const bool valid = pt < S.B;
  const long ptc = valid ? pt : S.B - 1;
  float* tp = tapes + pl * L.TP;
  const int ncol = P.ncol;
  if (f < 8) tp[L.t_x + f] = f < ncol ? P.xin[ptc * ncol + f] : 0.f;
  if (f < 4) { tp[L.t_z + f] = 0.f; tp[L.t_dz + f] = 0.f; tp[L.t_du + f] = 0.f; }
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
      for (int d = 0; d < pi; ++d) a = fmaf(xp[d], pw[L.q_fw + d * SMALL_W + f], a);
      a = fmaf(om_p, a, pw[L.q_fb + f]);
      { const SmallHD q_ = small_act(P.act, a); h = q_.h; dd = q_.d; }
      if (!fp) { h = 0.f; dd = 0.f; }
    }
    tp[L.t_hp + f] = h; tp[L.t_dp + f] = dd;
    for (int m = 0; m < lst; ++m) {
      float a = small_dot32(tp + L.t_hp + m * SMALL_W, pw + L.q_hw + m * SMALL_W * SMALL_RS + f, SMALL_RS);
      a = fmaf(om_p, a, pw[L.q_hb + m * SMALL_W + f]);
      float t, d;
      { const SmallHD q_ = small_act(P.act, a); t = q_.h; d = q_.d; }
      h = P.siren ? t : h + t;
      if (!fp) { h = 0.f; d = 0.f; }
      tp[L.t_hp + (m + 1) * SMALL_W + f] = h; tp[L.t_dp + (m + 1) * SMALL_W + f] = d;
    }
    for (int c = 0; c < r; ++c) {          // the latent: every lane of the half gets it; kept in the tape (zt_k = tp[t_z + k], zt_r = 1)
      const float v = half_sum(h * pw[L.q_bw + f * 4 + c]) + pw[L.q_bb + c];
      if (f == 0) tp[L.t_z + c] = v;
    }
    const float* ztp = tp + L.t_z;
#define ZT(k_) ((k_) < r ? ztp[k_] : 1.0f)
    SM_TL(3);
    // ShapeNet forward (plane formulation: the per-point matrix is never formed)
    float hs = 0.f;
    {
      float a = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float* pk = planes + k * L.PS;
        float s = 0.f;
        for (int d = 0; d < si; ++d) s = fmaf(xs[d], pk[L.o_w1 + d * SMALL_W + f], s);
        s = fmaf(om_s, s, pk[L.o_b1 + f]);
        a = fmaf(ZT(k), s, a);
      }
      float d;
      { const SmallHD q_ = small_act(S.act, a); hs = q_.h; d = q_.d; }
      if (!fs) { hs = 0.f; d = 0.f; }
      tp[L.t_hs + f] = hs; tp[L.t_ds + f] = d;
    }
    for (int j = 0; j < nh; ++j) {
      const float* hin = tp + L.t_hs + j * SMALL_W;
      float a = 0.f;
      for (int k = 0; k <= r; ++k) {
        float s = small_dot32(hin, planes + k * L.PS + L.o_wh + j * SMALL_W * SMALL_RS + f, SMALL_RS);
        s = fmaf(om_s, s, planes[k * L.PS + L.o_bh + j * SMALL_W + f]);
        a = fmaf(ZT(k), s, a);
      }
      float t, d;
      { const SmallHD q_ = small_act(S.act, a); t = q_.h; d = q_.d; }
      hs = S.nif_skip ? t + hs : t;
      if (!fs) { hs = 0.f; d = 0.f; }
      tp[L.t_hs + (j + 1) * SMALL_W + f] = hs; tp[L.t_ds + (j + 1) * SMALL_W + f] = d;
    }
    SM_TL(4);
    // last layer, loss, dL/du.  The lane's dL/dz partial sums: four scalars and a select per latent row (a run-time indexed array
    // would live in scratch, an unrolled plane loop multiplies the code)
    const float wsamp = valid ? (S.sw ? S.sw[ptc] : 1.0f) : 0.0f;
    float gh = 0.f, dz0 = 0.f, dz1 = 0.f, dz2 = 0.f, dz3 = 0.f;
#define DZADD(k_, v_) { const float v__ = (v_); dz0 += (k_) == 0 ? v__ : 0.f; dz1 += (k_) == 1 ? v__ : 0.f; dz2 += (k_) == 2 ? v__ : 0.f; dz3 += (k_) == 3 ? v__ : 0.f; }
    float se = 0.f;
    for (int o = 0; o < so; ++o) {
      float part = 0.f, wg = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float* pk = planes + k * L.PS;
        const float w = pk[L.o_wl + f * 4 + o];
        float s = hs * w;
        if (f == 0) s += pk[L.o_bl + o];
        const float z = ZT(k);
        part = fmaf(z, s, part);
        wg = fmaf(z, w, wg);
      }
      const float uo = half_sum(part);
      const float e = uo - S.y[ptc * so + o];
      NIF_LOSS_ACC(S.loss_kind, e, se, dfac)
      const float du = dfac * wsamp * S.inv_bg / (float)so;
      if (f == 0) tp[L.t_du + o] = du;
      gh = fmaf(du, wg, gh);
      for (int k = 0; k < r; ++k) {
        const float* pk = planes + k * L.PS;
        float s = hs * pk[L.o_wl + f * 4 + o];
        if (f == 0) s += pk[L.o_bl + o];
        DZADD(k, du * s)
      }
    }
    if (f == 0) loss_lane = wsamp * se / (float)so * S.inv_bg;
    SM_TL(5);
    // ShapeNet adjoint
    for (int j = nh - 1; j >= 0; --j) {
      const float da = gh * tp[L.t_ds + (j + 1) * SMALL_W + f];
      tp[L.t_ds + (j + 1) * SMALL_W + f] = da;
      const float* dar = tp + L.t_ds + (j + 1) * SMALL_W;
      const float hin = tp[L.t_hs + j * SMALL_W + f];
      float gin = S.nif_skip ? gh : 0.f;
      for (int k = 0; k <= r; ++k) {
        const float v = om_s * small_dot32(dar, planes + k * L.PS + L.o_wh + j * SMALL_W * SMALL_RS + f * SMALL_RS, 1);
        gin = fmaf(ZT(k), v, gin);
        if (k < r) DZADD(k, hin * v + da * planes[k * L.PS + L.o_bh + j * SMALL_W + f])
      }
      gh = fs ? gin : 0.f;
    }
    {
      const float da = gh * tp[L.t_ds + f];
      tp[L.t_ds + f] = da;
      for (int k = 0; k < r; ++k) {
        const float* pk = planes + k * L.PS;
        float s = 0.f;
        for (int d = 0; d < si; ++d) s = fmaf(xs[d], pk[L.o_w1 + d * SMALL_W + f], s);
        s = fmaf(om_s, s, pk[L.o_b1 + f]);
        DZADD(k, da * s)
      }
    }
#undef DZADD
    dz0 = half_sum(dz0);
    if (r > 1) dz1 = half_sum(dz1);
    if (r > 2) dz2 = half_sum(dz2);
    if (r > 3) dz3 = half_sum(dz3);
    if (f == 0) { tp[L.t_dz] = dz0; tp[L.t_dz + 1] = dz1; tp[L.t_dz + 2] = dz2; tp[L.t_dz + 3] = dz3; }
    SM_TL(6);
    // ParameterNet adjoint (rows of the bottleneck image beyond latent_dim are zero)
    const f32x4 bw4 = *reinterpret_cast<const f32x4*>(pw + L.q_bw + f * 4);
    float g = fmaf(dz0, bw4[0], fmaf(dz1, bw4[1], fmaf(dz2, bw4[2], dz3 * bw4[3])));
    for (int m = lst - 1; m >= 0; --m) {
      const float da = g * tp[L.t_dp + (m + 1) * SMALL_W + f];
      tp[L.t_dp + (m + 1) * SMALL_W + f] = da;
      const float v = small_dot32(tp + L.t_dp + (m + 1) * SMALL_W, pw + L.q_hw + m * SMALL_W * SMALL_RS + f * SMALL_RS, 1);
      g = P.siren ? om_p * v : g + om_p * v;
      if (!fp) g = 0.f;
    }
    tp[L.t_dp + f] = g * tp[L.t_dp + f];
  }
#undef ZT
  SM_TL(7);
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

  SM_TL(8);
  // ---- every entry of the flat gradient over the workgroup's 16 points, tensor by tensor: entry (i, o) = scale * sum_p A_i(p) B_o(p) zt_kk(p),
  // A = tape value oa + i (oa < 0: the constant 1, one row), B = tape value ob + o (kk < 0: no latent factor).  One tensor per WAVE and
  // pass as eight v_mfma_f32_32x32x2_f32 (K = 2 points each; fp32 products and sums): lane (i | o, kh) holds A_i / B_o of point 2 q + kh,
  // 24 LDS words per lane and tensor.  (Per-thread dot products read 32-48 LDS words PER ENTRY: LDS-instruction bound.)  Rows / columns
  // beyond the tensor come out as entries of the 32 x 32 product that are never stored.
  float* prow = A.partial + (long)blockIdx.x * A.pstride;
  const int TP = L.TP;
  for (int t = wid; t < ntasks; t += SMALL_NT / 64) {
    const int* D = desc + __builtin_amdgcn_readfirstlane(t) * SMALL_DW;
    const int off = D[0], rows = D[1], cols = D[2], oa = D[5], ob = D[6], kk = D[7];
    const float scale = __int_as_float(D[8]);
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int oae = (oa >= 0 ? oa : 0) + sc, kke = L.t_z + (kk >= 0 ? kk : 0);
    const float one0 = sc == 0 ? 1.0f : 0.0f;
    float a[SMALL_T / 2], b[SMALL_T / 2], z[SMALL_T / 2];
#pragma unroll
    for (int q = 0; q < SMALL_T / 2; ++q) {
      const float* tq = tapes + (2 * q + sg) * TP;
      a[q] = tq[oae]; b[q] = tq[ob + sc]; z[q] = tq[kke];
    }
#pragma unroll
    for (int q = 0; q < SMALL_T / 2; ++q) {
      const float av = oa >= 0 ? a[q] : one0, bv = kk >= 0 ? b[q] * z[q] : b[q];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (sc < cols) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = fmap(v, sg);
        if (row < rows) prow[off + row * cols + sc] = scale * acc[v];
      }
    }
  }
  SM_TL(9);
}

// ---- host side -------------------------------------------------------------------------------------------------------------
bool small_supported(const PNetArgs& p, const SNetArgs& s) {
  if (p.ll_kind || p.res || s.res || s.ll) return false;
  if (s.prec != 0) return false;
  if (s.n > 32 || p.nst > 32 || s.nh > SMALL_MAXH || p.lst > SMALL_MAXH || p.r > SMALL_MAXR || p.r < 1) return false;
  if (s.si > 4 || s.so > 4 || p.pi > 4 || s.si < 1 || p.pi < 1 || s.nh < 0 || p.lst < 0) return false;
  if (p.ncol > 8) return false;
  if (small_ntasks(p.lst, p.r, s.nh) > SMALL_MAXT) return false;
  const SmallLay L = small_layout(p.pi, p.lst, p.r, s.si, s.nh);
  return (size_t)L.total * sizeof(float) <= 128u * 1024u;      // (one workgroup per CU is plenty for <= 128 workgroups)
}
int small_rows(long B) { return (int)((B + SMALL_T - 1) / SMALL_T); }
// the context's tables (offsets only: they do not change with the weights): idx[L.tapes] = theta index of every word of the padded LDS
// images (-1: zero padding), desc[ntasks][SMALL_DW] = the tensors of the gradient phase
void small_tables(const PNetArgs& p, const SNetArgs& s, std::vector<int>& idx, std::vector<int>& desc) {
  const SmallLay L = small_layout(p.pi, p.lst, p.r, s.si, s.nh);
  const int nt = small_ntasks(p.lst, p.r, s.nh);
  idx.assign((size_t)L.tapes, -1);
  desc.assign((size_t)nt * SMALL_DW, 0);
  for (int g = 0; g < nt; ++g) {
    int* D = desc.data() + (size_t)g * SMALL_DW;
    small_desc(g, p, s, L, D);
    for (int i = 0; i < D[1]; ++i)
      for (int o = 0; o < D[2]; ++o) idx[(size_t)D[3] + (size_t)i * D[4] + o] = D[0] + i * D[2] + o;
  }
}
// one workgroup per 16 points; partial rows [small_rows(B)][pstride], loss partials [small_rows(B)]
void launch_small(const PNetArgs& p, const SNetArgs& s, float* partial, long pstride, long P, const int* idx_dev, const int* desc_dev,
                  double* metric, float metric_w, const float* g_loss, hipStream_t st) {
  SmallArgs A; A.p = p; A.s = s; A.partial = partial; A.pstride = pstride; A.P = P; A.idx = idx_dev; A.desc = desc_dev;
  A.metric = metric; A.metric_w = metric_w; A.g_loss = g_loss;
  const SmallLay L = small_layout(p.pi, p.lst, p.r, s.si, s.nh);
  const size_t shm = (size_t)L.total * sizeof(float);
  if (shm > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL(k_small, dim3((unsigned)small_rows(s.B)), dim3(SMALL_NT), shm, st, A);
}
