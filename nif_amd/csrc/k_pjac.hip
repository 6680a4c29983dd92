// k_pjac.hip -- latent Jacobian regulariser of the ParameterNet (reference cfg_parameter_net["jac_reg"]:
// nif/model.py:353-375 wraps the model in JacRegLatentLayer, nif/layers/gradient.py:52-127, :182-205):
//     loss += l1 * mean_{a, c, d} (d z_c / d p_d)^2        z = latent (bottleneck output), p = the parameter inputs
// Keras differentiates THROUGH the inner batch_jacobian; here, as for the Sobolev step (k_sob.hip): forward-mode tangents
// z'_d of the ParameterNet MLP next to the primal, then the hand-derived adjoint of that (primal, tangent) program.  With
// lambda = dL/dh, mu_d = dL/dh'_d, and a'_d the tangent pre-activation of a layer h_out = f(a):
//     nu_d = mu_d f'(a)                     (dL/da'_d)
//     da   = lambda f'(a) + sum_d mu_d f''(a) a'_d
//     dL/dW = s (h_in (x) da + sum_d h'_in,d (x) nu_d) ,  dL/db = da ,   lambda_in = s W da ,  mu_in,d = s W nu_d
// (s = omega_0 for SIREN layers).  The weight gradients are K = batch GEMMs: like k_sob, the kernel only writes the operand
// pairs into the ParameterNet's stash -- (h_in, da) as the real tiles, (h'_in,d, nu_d) as one block of pseudo-tiles per
// parameter column d behind them -- and the unchanged gradient kernels (k_gw_first / k_gw_mfma / k_gw_out, GwArgs::zt_mod /
// bias_ntiles / seed) reduce over all of them.
//
// One thread per point, loops over the features (the ParameterNet is small: units <= 64, <= 4 hidden matrices, <= 3
// parameter columns; weights arrive by wave-uniform loads); pre-activations are parked in the dL/da stash slots during the
// forward sweep and replaced by dL/da in the adjoint sweep.  An optional regulariser, not the benchmark path.
#include "nif_internal.h"

#define NIF_PJ_MAXPI 3

struct PJacArgs {
  PNetArgs p;           // theta offsets, activation, inputs; stash = the ParameterNet stash with room for (1 + pi) x tiles
  float coef;           // l1 / (B_global * r * pi)
  float* MU;            // [(1 + pi) * tiles][r][32]: dL/dz'_d of the pseudo-tiles (zeros for the real tiles) -> k_gw_out
  float* loss_partial;  // [gridDim.x]
  // Sobolev training with parameter columns in x_index (k_sob.hip, PAR) reuses this kernel on either side of the ShapeNet:
  int mode;             // 0: the regulariser; 1: forward only, z'_d -> ZT; 2: the adjoint for GIVEN dL/dz'_d (MU_in), no loss
  float* ZT;            // mode 1: [pi][tiles][r][32]
  const float* MU_in;   // mode 2: blocks of [tiles][r][32]; column d reads block mu_blk[d] (< 0: zero)
  int mu_blk[NIF_PJ_MAXPI];
  // r4: ParameterNets with more than NIF_PJ_MAXPI inputs run in passes over groups of parameter columns -- tangent d of a pass is
  // column c0 + d, nd <= NIF_PJ_MAXPI of them; every term is linear in the tangent seeds, so the passes' gradients simply add
  int c0, nd;
};

__device__ __forceinline__ void pj_act(int act, float a, float* f0, float* f1, float* f2) {
  switch (act) {
    case ACT_SINE: { float s, c; nif_sincosf(a, &s, &c); *f0 = s; *f1 = c; *f2 = -s; } break;
    case ACT_SWISH: { const float s = 1.0f / (1.0f + expf(-a)); *f0 = a * s; *f1 = s * (1.0f + a * (1.0f - s));
                      *f2 = s * (1.0f - s) * (2.0f + a * (1.0f - 2.0f * s)); } break;
    case ACT_TANH: { const float t = tanhf(a); *f0 = t; *f1 = 1.0f - t * t; *f2 = -2.0f * t * (1.0f - t * t); } break;
    case ACT_RELU: *f0 = a > 0.f ? a : 0.f; *f1 = a > 0.f ? 1.f : 0.f; *f2 = 0.f; break;
    case ACT_SIGMOID: { const float s = 1.0f / (1.0f + expf(-a)); *f0 = s; *f1 = s * (1.0f - s); *f2 = s * (1.0f - s) * (1.0f - 2.0f * s); } break;
    case ACT_ELU: { const float e = expf(fminf(a, 0.f)); *f0 = a > 0.f ? a : e - 1.0f; *f1 = a > 0.f ? 1.0f : e; *f2 = a > 0.f ? 0.f : e; } break;
    case ACT_SOFTPLUS: { const float s = 1.0f / (1.0f + expf(-a)); *f0 = fmaxf(a, 0.f) + log1pf(expf(-fabsf(a))); *f1 = s; *f2 = s * (1.0f - s); } break;
    case ACT_GELU: { const float cdf = 0.5f * (1.0f + nif_erff(a * 0.70710678118654752440f)), pdf = 0.3989422804014327f * expf(-0.5f * a * a);
                     *f0 = a * cdf; *f1 = cdf + a * pdf; *f2 = pdf * (2.0f - a * a); } break;
    case ACT_SELU: { const float e = NIF_SELU_ALPHA * expf(fminf(a, 0.f)); *f0 = NIF_SELU_SCALE * (a > 0.f ? a : e - NIF_SELU_ALPHA);
                     *f1 = NIF_SELU_SCALE * (a > 0.f ? 1.0f : e); *f2 = a > 0.f ? 0.f : NIF_SELU_SCALE * e; } break;
    case ACT_SOFTSIGN: { const float q = 1.0f / (1.0f + fabsf(a)); *f0 = a * q; *f1 = q * q;
                         *f2 = (a > 0.f ? -2.0f : (a < 0.f ? 2.0f : 0.f)) * q * q * q; } break;
    case ACT_EXPONENTIAL: { const float e = expf(a); *f0 = e; *f1 = e; *f2 = e; } break;
    case ACT_HARD_SIGMOID: { const float t = fmaf(0.2f, a, 0.5f); *f0 = fminf(fmaxf(t, 0.f), 1.f); *f1 = (t > 0.f && t < 1.f) ? 0.2f : 0.f; *f2 = 0.f; } break;
    default: *f0 = a; *f1 = 1.0f; *f2 = 0.f; break;
  }
}

template <int NST>
__global__ __launch_bounds__(128) void k_pjac(PJacArgs J) {
  const PNetArgs& A = J.p;
  __shared__ float red[128];
  const long pt = (long)blockIdx.x * 128 + threadIdx.x;
  const long ntiles = (A.B + 31) / 32;
  const bool ok = pt < A.B;                 // a real point
  const bool wr = pt < ntiles * 32;         // a row of the stash tiles (the padding rows of the last tile are written as zeros)
  const long ptc = ok ? pt : A.B - 1;       // inputs of a padding thread: any valid point, its results are discarded
  const long pta = wr ? pt : ntiles * 32 - 1;
  const long tile = pta >> 5; const int pp = (int)(pta & 31);
  const int pia = A.pi, pi = J.nd, c0 = J.c0, nst = A.nst, lst = A.lst, r = A.r, res = A.res;      // pi: the tangents of THIS pass
  const int nm = lst * (res ? 2 : 1);
  const int FP = stash_fp(nst);           // feature rows of a stash tile as the gradient kernels read them (32, 64 or 128)
  const float s = A.siren ? A.omega : 1.0f;
  const int act = A.siren ? ACT_SINE : A.act;
  const float* th = A.theta;
  // stash element (slot, tile block blk = 0 real / 1 + d pseudo, feature f)
  auto at = [&](int slot, int blk, int f) -> float* {
    return A.stash + (long)slot * A.slot_stride + (((long)blk * ntiles + tile) * FP + f) * 32 + pp;
  };
  const bool fwd_only = J.mode == 1;
  auto ST = [&](float* q, float v) { if (wr && !fwd_only) *q = v; };
  auto LD = [&](const float* q) -> float { return wr ? *q : 0.f; };
  const int S_IN = 0, S_DA0 = nm + 1, S_DA = nm + 2;
  float h[NST], hd[NIF_PJ_MAXPI][NST];

  // ---------------- forward: primal + tangents; IN_m <- layer inputs, DA_m <- pre-activations (for now) ----------------
  for (int j = 0; j < nst; ++j) {
    float a = 0.f;
    for (int d = 0; d < pia; ++d) a = fmaf(A.xin[ptc * A.ncol + A.col0 + d], th[A.first_w + (long)d * nst + j], a);
    a = s * a + th[A.first_b + j];
    float f0, f1, f2; pj_act(act, a, &f0, &f1, &f2);
    h[j] = f0;
    ST(at(S_DA0, 0, j), a);
    for (int d = 0; d < pi; ++d) {
      const float ad = s * th[A.first_w + (long)(c0 + d) * nst + j];
      hd[d][j] = f1 * ad;
      ST(at(S_DA0, 1 + d, j), ad);
    }
  }
  // y = s * (v W) (+ b for the primal) for the primal vector v and its tangents; results into out / outd
  float out[NST], outd[NIF_PJ_MAXPI][NST];
  auto matvec = [&](long w_off, long b_off, const float* v, const float (*vd)[NST]) {
    for (int j = 0; j < nst; ++j) {
      float a = 0.f, ad[NIF_PJ_MAXPI] = {0.f, 0.f, 0.f};
      for (int i = 0; i < nst; ++i) {
        const float w = th[w_off + (long)i * nst + j];
        a = fmaf(v[i], w, a);
        for (int d = 0; d < pi; ++d) ad[d] = fmaf(vd[d][i], w, ad[d]);
      }
      out[j] = s * a + th[b_off + j];
      for (int d = 0; d < pi; ++d) outd[d][j] = s * ad[d];
    }
  };
  auto put_in = [&](int m, const float* v, const float (*vd)[NST]) {
    for (int j = 0; j < FP; ++j) {
      ST(at(S_IN + m, 0, j), (ok && j < nst) ? v[j] : 0.f);
      for (int d = 0; d < pi; ++d) ST(at(S_IN + m, 1 + d, j), (ok && j < nst) ? vd[d][j] : 0.f);
    }
  };
  auto put_a = [&](int m) {   // pre-activations of matrix m (out / outd) into its dL/da slot
    for (int j = 0; j < nst; ++j) {
      ST(at(S_DA + m, 0, j), out[j]);
      for (int d = 0; d < pi; ++d) ST(at(S_DA + m, 1 + d, j), outd[d][j]);
    }
  };
  for (int i = 0; i < lst; ++i) {
    if (!res) {
      put_in(i, h, hd);
      matvec(A.hid_w[i], A.hid_b[i], h, hd);
      put_a(i);
      for (int j = 0; j < nst; ++j) {
        float f0, f1, f2; pj_act(act, out[j], &f0, &f1, &f2);
        if (A.siren) { h[j] = f0; for (int d = 0; d < pi; ++d) hd[d][j] = f1 * outd[d][j]; }
        else { h[j] += f0; for (int d = 0; d < pi; ++d) hd[d][j] += f1 * outd[d][j]; }          // MLP_SimpleShortCut
      }
    } else {
      put_in(2 * i, h, hd);
      matvec(A.hid_w[i], A.hid_b[i], h, hd);
      put_a(2 * i);
      float t[NST], td[NIF_PJ_MAXPI][NST];
      for (int j = 0; j < nst; ++j) {
        float f0, f1, f2; pj_act(act, out[j], &f0, &f1, &f2);
        t[j] = f0; for (int d = 0; d < pi; ++d) td[d][j] = f1 * outd[d][j];
      }
      put_in(2 * i + 1, t, td);
      matvec(A.hid_w2[i], A.hid_b2[i], t, td);
      if (!A.siren) for (int j = 0; j < nst; ++j) { out[j] += h[j]; for (int d = 0; d < pi; ++d) outd[d][j] += hd[d][j]; }   // MLP_ResNet: a2 = x + L2(..)
      put_a(2 * i + 1);
      for (int j = 0; j < nst; ++j) {
        float f0, f1, f2; pj_act(act, out[j], &f0, &f1, &f2);
        if (A.siren) { h[j] = 0.5f * (h[j] + f0); for (int d = 0; d < pi; ++d) hd[d][j] = 0.5f * (hd[d][j] + f1 * outd[d][j]); }
        else { h[j] = f0; for (int d = 0; d < pi; ++d) hd[d][j] = f1 * outd[d][j]; }
      }
    }
  }
  put_in(nm, h, hd);          // bottleneck input
  // ---------------- bottleneck (linear): z'_d, the loss, the adjoint seeds ----------------
  float lam[NST], mu[NIF_PJ_MAXPI][NST];
  for (int j = 0; j < nst; ++j) { lam[j] = 0.f; for (int d = 0; d < NIF_PJ_MAXPI; ++d) mu[d][j] = 0.f; }
  float lsum = 0.f;
  for (int c = 0; c < r; ++c) {
    ST(J.MU + (tile * r + c) * 32 + pp, 0.f);
    for (int d = 0; d < pi; ++d) {
      float zd = 0.f;
      for (int i = 0; i < nst; ++i) zd = fmaf(hd[d][i], th[A.bott_w + (long)i * r + c], zd);
      if (fwd_only) { if (wr) J.ZT[(((long)(c0 + d) * ntiles + tile) * r + c) * 32 + pp] = ok ? zd : 0.f; continue; }
      float m_;
      if (J.mode == 2) m_ = (ok && J.mu_blk[d] >= 0) ? J.MU_in[(((long)J.mu_blk[d] * ntiles + tile) * r + c) * 32 + pp] : 0.f;
      else { lsum = fmaf(zd, zd, lsum); m_ = ok ? 2.0f * J.coef * zd : 0.f; }
      ST(J.MU + (((long)(1 + d) * ntiles + tile) * r + c) * 32 + pp, m_);
      for (int i = 0; i < nst; ++i) mu[d][i] = fmaf(m_, th[A.bott_w + (long)i * r + c], mu[d][i]);
    }
  }
  if (fwd_only) return;
  // ---------------- adjoint through the hidden layers: DA_m <- (da | nu_d) ----------------
  auto back = [&](long w_off, const float* da, const float (*nu)[NST], float* lo, float (*mo)[NST], float keep) {
    // lo = keep * lo + s W da ;  mo_d = keep * mo_d + s W nu_d
    for (int i = 0; i < nst; ++i) {
      float a = 0.f, ad[NIF_PJ_MAXPI] = {0.f, 0.f, 0.f};
      for (int j = 0; j < nst; ++j) {
        const float w = th[w_off + (long)i * nst + j];
        a = fmaf(da[j], w, a);
        for (int d = 0; d < pi; ++d) ad[d] = fmaf(nu[d][j], w, ad[d]);
      }
      lo[i] = keep * lo[i] + s * a;
      for (int d = 0; d < pi; ++d) mo[d][i] = keep * mo[d][i] + s * ad[d];
    }
  };
  // (da, nu) of one matrix from the parked pre-activations and the incoming (l, m); overwrites the slot
  float da[NST], nu[NIF_PJ_MAXPI][NST];
  auto adj = [&](int slot, const float* l, const float (*m)[NST], float scale) {
    for (int j = 0; j < FP; ++j) {
      if (j >= nst) { ST(at(slot, 0, j), 0.f); for (int d = 0; d < pi; ++d) ST(at(slot, 1 + d, j), 0.f); continue; }
      const float a = LD(at(slot, 0, j));
      float f0, f1, f2; pj_act(act, a, &f0, &f1, &f2);
      float v = scale * l[j] * f1;
      for (int d = 0; d < pi; ++d) {
        const float ad = LD(at(slot, 1 + d, j));
        v = fmaf(scale * m[d][j] * f2, ad, v);
        nu[d][j] = scale * m[d][j] * f1;
        ST(at(slot, 1 + d, j), ok ? nu[d][j] : 0.f);
      }
      da[j] = v;
      ST(at(slot, 0, j), ok ? v : 0.f);
    }
  };
  for (int i = lst - 1; i >= 0; --i) {
    if (!res) {
      adj(S_DA + i, lam, mu, 1.0f);
      back(A.hid_w[i], da, nu, lam, mu, A.siren ? 0.0f : 1.0f);       // shortcut: lambda_in = lambda + W da
    } else if (A.siren) {   // h_out = 0.5 (h + sin(a2)), a2 = s t W2 + b2, t = sin(a1), a1 = s h W1 + b1
      adj(S_DA + 2 * i + 1, lam, mu, 0.5f);
      float lt[NST], mt[NIF_PJ_MAXPI][NST];
      for (int j = 0; j < nst; ++j) { lt[j] = 0.f; for (int d = 0; d < NIF_PJ_MAXPI; ++d) mt[d][j] = 0.f; }
      back(A.hid_w2[i], da, nu, lt, mt, 0.0f);
      adj(S_DA + 2 * i, lt, mt, 1.0f);
      back(A.hid_w[i], da, nu, lam, mu, 0.5f);
    } else {                // h_out = f(a2), a2 = h + t W2 + b2, t = f(a1), a1 = h W1 + b1
      adj(S_DA + 2 * i + 1, lam, mu, 1.0f);
      float lt[NST], mt[NIF_PJ_MAXPI][NST];
      for (int j = 0; j < nst; ++j) { lam[j] = da[j]; lt[j] = 0.f; for (int d = 0; d < NIF_PJ_MAXPI; ++d) { mu[d][j] = d < pi ? nu[d][j] : 0.f; mt[d][j] = 0.f; } }
      back(A.hid_w2[i], da, nu, lt, mt, 0.0f);
      adj(S_DA + 2 * i, lt, mt, 1.0f);
      back(A.hid_w[i], da, nu, lam, mu, 1.0f);
    }
  }
  adj(S_DA0, lam, mu, 1.0f);     // first layer: k_gw_first pairs da with the inputs p and nu_d with the one-hot e_d
  red[threadIdx.x] = ok ? J.coef * lsum : 0.f;
  __syncthreads();
  for (int off = 64; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) J.loss_partial[blockIdx.x] = red[0];
}

// Second-order forward mode for ONE pair of parameter columns (cj, ck): d2z/dp_cj dp_ck -> ZDD [tiles][r][32] (HessianLayer with
// parameter columns, nif/layers/gradient.py:130-180).  h'' = f'(a) a'' + f''(a) a'_j a'_k through the same layer kinds; one
// thread per point, no stash traffic.
template <int NST>
__global__ __launch_bounds__(128) void k_pjac2(PNetArgs A, int cj, int ck, float* __restrict__ ZDD) {
  const long pt = (long)blockIdx.x * 128 + threadIdx.x;
  const long ntiles = (A.B + 31) / 32;
  if (pt >= ntiles * 32) return;
  const bool ok = pt < A.B;
  const long ptc = ok ? pt : A.B - 1;
  const long tile = pt >> 5; const int pp = (int)(pt & 31);
  const int pi = A.pi, nst = A.nst, lst = A.lst, r = A.r, res = A.res;
  const float s = A.siren ? A.omega : 1.0f;
  const int act = A.siren ? ACT_SINE : A.act;
  const float* th = A.theta;
  float h[NST], hj[NST], hk[NST], hjk[NST];
  for (int j = 0; j < nst; ++j) {
    float a = 0.f;
    for (int d = 0; d < pi; ++d) a = fmaf(A.xin[ptc * A.ncol + A.col0 + d], th[A.first_w + (long)d * nst + j], a);
    a = s * a + th[A.first_b + j];
    const float aj = s * th[A.first_w + (long)cj * nst + j], ak = s * th[A.first_w + (long)ck * nst + j];
    float f0, f1, f2; pj_act(act, a, &f0, &f1, &f2);
    h[j] = f0; hj[j] = f1 * aj; hk[j] = f1 * ak; hjk[j] = f2 * aj * ak;
  }
  float o0[NST], oj[NST], ok_[NST], ojk[NST];
  // (o0 | oj | ok_ | ojk) = s (v W) (+ b), v in (v0 | vj | vk | vjk)
  auto matvec = [&](long w_off, long b_off, const float* v0, const float* vj, const float* vk, const float* vjk) {
    for (int j = 0; j < nst; ++j) {
      float a = 0.f, aj = 0.f, ak = 0.f, ajk = 0.f;
      for (int i = 0; i < nst; ++i) {
        const float w = th[w_off + (long)i * nst + j];
        a = fmaf(v0[i], w, a); aj = fmaf(vj[i], w, aj); ak = fmaf(vk[i], w, ak); ajk = fmaf(vjk[i], w, ajk);
      }
      o0[j] = s * a + th[b_off + j]; oj[j] = s * aj; ok_[j] = s * ak; ojk[j] = s * ajk;
    }
  };
  // in place: (o0 | ..) <- (f(a), f' a'_j, f' a'_k, f' a'' + f'' a'_j a'_k)
  auto act4 = [&]() {
    for (int j = 0; j < nst; ++j) {
      float f0, f1, f2; pj_act(act, o0[j], &f0, &f1, &f2);
      const float aj = oj[j], ak = ok_[j];
      o0[j] = f0; oj[j] = f1 * aj; ok_[j] = f1 * ak; ojk[j] = fmaf(f1, ojk[j], f2 * aj * ak);
    }
  };
  for (int i = 0; i < lst; ++i) {
    if (!res) {
      matvec(A.hid_w[i], A.hid_b[i], h, hj, hk, hjk);
      act4();
      for (int j = 0; j < nst; ++j) {
        if (A.siren) { h[j] = o0[j]; hj[j] = oj[j]; hk[j] = ok_[j]; hjk[j] = ojk[j]; }
        else { h[j] += o0[j]; hj[j] += oj[j]; hk[j] += ok_[j]; hjk[j] += ojk[j]; }          // MLP_SimpleShortCut
      }
    } else {
      matvec(A.hid_w[i], A.hid_b[i], h, hj, hk, hjk);
      act4();
      float t0[NST], tj[NST], tk[NST], tjk[NST];
      for (int j = 0; j < nst; ++j) { t0[j] = o0[j]; tj[j] = oj[j]; tk[j] = ok_[j]; tjk[j] = ojk[j]; }
      matvec(A.hid_w2[i], A.hid_b2[i], t0, tj, tk, tjk);
      if (!A.siren) for (int j = 0; j < nst; ++j) { o0[j] += h[j]; oj[j] += hj[j]; ok_[j] += hk[j]; ojk[j] += hjk[j]; }   // MLP_ResNet
      act4();
      for (int j = 0; j < nst; ++j) {
        if (A.siren) { h[j] = 0.5f * (h[j] + o0[j]); hj[j] = 0.5f * (hj[j] + oj[j]); hk[j] = 0.5f * (hk[j] + ok_[j]); hjk[j] = 0.5f * (hjk[j] + ojk[j]); }
        else { h[j] = o0[j]; hj[j] = oj[j]; hk[j] = ok_[j]; hjk[j] = ojk[j]; }
      }
    }
  }
  for (int c = 0; c < r; ++c) {
    float zdd = 0.f;
    for (int i = 0; i < nst; ++i) zdd = fmaf(hjk[i], th[A.bott_w + (long)i * r + c], zdd);
    ZDD[(tile * r + c) * 32 + pp] = ok ? zdd : 0.f;
  }
}
void launch_pjac2(const PNetArgs& a, int cj, int ck, float* ZDD, hipStream_t st) {
  const long ntiles = (a.B + 31) / 32;
  const int nblk = (int)((ntiles * 32 + 127) / 128);
  if (a.nst <= 32) hipLaunchKernelGGL((k_pjac2<32>), dim3(nblk), dim3(128), 0, st, a, cj, ck, ZDD);
  else if (a.nst <= 64) hipLaunchKernelGGL((k_pjac2<64>), dim3(nblk), dim3(128), 0, st, a, cj, ck, ZDD);
  else hipLaunchKernelGGL((k_pjac2<128>), dim3(nblk), dim3(128), 0, st, a, cj, ck, ZDD);
}

bool pjac_supported(const PNetArgs& a) {
  const int nm = a.lst * (a.res ? 2 : 1);
  // (first / hidden / bottleneck only: the same for every class).  r3: every ParameterNet width / depth the engine accepts --
  // the 128-unit instantiation keeps its per-thread vectors in scratch (a scalar kernel for optional terms: slow, never refused);
  // r4: any number of parameter inputs (passes over groups of NIF_PJ_MAXPI columns)
  return a.nst <= 128 && nm <= NIF_MAX_HID;
}
int pjac_group() { return NIF_PJ_MAXPI; }
static int launch_pjac_any(const PJacArgs& J, hipStream_t st);
// the regulariser's pass over parameter columns [c0, c0 + nd): MU / stash hold (1 + nd) blocks of tiles
int launch_pjac(const PNetArgs& a, float coef, float* MU, float* loss_partial, int c0, int nd, hipStream_t st) {
  PJacArgs J; J.p = a; J.coef = coef; J.MU = MU; J.loss_partial = loss_partial;
  J.mode = 0; J.ZT = nullptr; J.MU_in = nullptr; J.c0 = c0; J.nd = nd;
  for (int d = 0; d < NIF_PJ_MAXPI; ++d) J.mu_blk[d] = -1;
  return launch_pjac_any(J, st);
}
// z'_d = dz/dp_d of every parameter column -> ZT [pi][tiles][r][32] (no stash traffic)
int launch_pjac_fwd(const PNetArgs& a, float* ZT, hipStream_t st) {
  int nblk = 0;
  for (int c0 = 0; c0 < a.pi; c0 += NIF_PJ_MAXPI) {
    PJacArgs J; J.p = a; J.coef = 0.f; J.MU = ZT; J.loss_partial = nullptr;
    J.mode = 1; J.ZT = ZT; J.MU_in = nullptr; J.c0 = c0; J.nd = a.pi - c0 < NIF_PJ_MAXPI ? a.pi - c0 : NIF_PJ_MAXPI;
    for (int d = 0; d < NIF_PJ_MAXPI; ++d) J.mu_blk[d] = -1;
    nblk = launch_pjac_any(J, st);
  }
  return nblk;
}
// adjoint of the (primal, tangent) ParameterNet for given dL/dz'_d of the columns [c0, c0 + nd) (block mu_blk[c0 + d] of MU_in,
// mu_blk indexed by COLUMN; the primal dL/dz part is the ordinary ParameterNet adjoint's business): operand pairs into the stash,
// MU for k_gw_out, like the regulariser
int launch_pjac_adj(const PNetArgs& a, const float* MU_in, const int* mu_blk, float* MU, float* loss_partial, int c0, int nd, hipStream_t st) {
  PJacArgs J; J.p = a; J.coef = 0.f; J.MU = MU; J.loss_partial = loss_partial;
  J.mode = 2; J.ZT = nullptr; J.MU_in = MU_in; J.c0 = c0; J.nd = nd;
  for (int d = 0; d < NIF_PJ_MAXPI; ++d) J.mu_blk[d] = d < nd ? mu_blk[c0 + d] : -1;
  return launch_pjac_any(J, st);
}
static int launch_pjac_any(const PJacArgs& J, hipStream_t st) {
  const PNetArgs& a = J.p;
  const int nblk = (int)((a.B + 127) / 128);
  if (a.nst <= 32) hipLaunchKernelGGL((k_pjac<32>), dim3(nblk), dim3(128), 0, st, J);
  else if (a.nst <= 64) hipLaunchKernelGGL((k_pjac<64>), dim3(nblk), dim3(128), 0, st, J);
  else hipLaunchKernelGGL((k_pjac<128>), dim3(nblk), dim3(128), 0, st, J);
  return nblk;
}
