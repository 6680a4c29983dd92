/*
 * nif_ref_cpu.c -- CPU restatement (C, OpenMP, fp32) of the REFERENCE FORMULATION of the NIF training
 * step, used ONLY as (a) a second oracle checked against oracle/nif_oracle.py and (b) the `cpu_baseline`
 * leg of bench.py.  TEST / MEASUREMENT INFRASTRUCTURE: the product (nif_amd/) never links or calls it.
 *
 * It deliberately keeps the data movement TensorFlow performs for nif/model.py (file:line under the
 * reference tree): the ParameterNet Dense/SIREN chain (model.py:326-343), the MATERIALISED
 * pnet_output[b, po] = z @ Wh + bh (siren.py:514-522 / model.py:220-230), the slicing of
 * model.py:253-300 / :883-933, the per-sample einsum('ai,aij->aj') chain (mlp.py:219; model.py:304-322,
 * :936-951), Keras 'mse', and a reverse sweep that materialises the per-sample gradient g_pnet_out[b, po]
 * (what GradientTape's StridedSliceGrad/AddN build) before contracting it with z.  PARITY UNPINNED against
 * TensorFlow itself (absent here); pinned against the NumPy oracle in tests/test_ref_cpu.py.
 *
 * Supported: class NIF (activation swish|tanh, shortcut ParameterNet) and NIFMultiScale without
 * resblocks (SIREN ShapeNet; ParameterNet sine or swish|tanh shortcut) -- what BASELINE.json configs 1-2 use.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int kind;      /* 0 NIF, 1 NIFMultiScale */
  int pi, si, so, n, L, nst, lst, r;
  int s_act;     /* 1 sine, 2 swish, 3 tanh */
  int p_act;
  float omega_s, omega_p;
} ref_cfg;

static inline float act_f(int act, float a) {
  switch (act) {
    case 1: return sinf(a);
    case 2: return a / (1.0f + expf(-a));
    case 3: return tanhf(a);
    default: return a;
  }
}
static inline float act_d(int act, float a) {
  switch (act) {
    case 1: return cosf(a);
    case 2: { float s = 1.0f / (1.0f + expf(-a)); return s * (1.0f + a * (1.0f - s)); }
    case 3: { float t = tanhf(a); return 1.0f - t * t; }
    default: return 1.0f;
  }
}

long nifref_po(const ref_cfg* c) { return (long)c->L * c->n * c->n + (long)(c->si + c->so + 1 + c->L) * c->n + c->so; }
long nifref_nparams(const ref_cfg* c) {
  return (long)c->pi * c->nst + c->nst + (long)c->lst * ((long)c->nst * c->nst + c->nst) + (long)c->nst * c->r + c->r +
         (long)c->r * nifref_po(c) + nifref_po(c);
}
int nifref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* loss (scaled by 1/Bg) and gradient += of one micro-batch.  grad must be zeroed by the caller before the
 * first micro-batch.  Returns 0, or -1 on allocation failure. */
int nifref_loss_grad(const ref_cfg* c, const float* theta, const float* xin, const float* y, const float* sw, long B,
                     long Bg, float* grad, double* loss_out, int nthreads) {
  const int pi = c->pi, si = c->si, so = c->so, n = c->n, L = c->L, nst = c->nst, lst = c->lst, r = c->r;
  const long po = nifref_po(c);
  const int siren_p = (c->kind == 1 && c->p_act == 1);
  const float om_p = siren_p ? c->omega_p : 1.0f, om_s = c->kind == 1 ? c->omega_s : 1.0f;
  const int sact = c->kind == 1 ? 1 : c->s_act;
  /* parameter offsets, Keras variable order */
  long off = 0;
  const long o_w0 = off; off += (long)pi * nst;
  const long o_b0 = off; off += nst;
  long o_wh[64], o_bh[64];
  for (int i = 0; i < lst; ++i) { o_wh[i] = off; off += (long)nst * nst; o_bh[i] = off; off += nst; }
  const long o_wb = off; off += (long)nst * r;
  const long o_bb = off; off += r;
  const long o_Wh = off; off += (long)r * po;
  const long o_Bh = off; off += po;
  /* pnet_output slots */
  const long s_w1 = 0, s_wh = (long)si * n, s_wl = s_wh + (long)L * n * n, s_b1 = s_wl + (long)n * so, s_bh = s_b1 + n,
             s_bl = s_bh + (long)L * n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  /* workspace is kept between calls (a framework would reuse its arena too; re-faulting 0.5 GB per micro-batch
   * on a many-core host otherwise dominates) */
  static float* ws = NULL;
  static size_t ws_cap = 0;
  const size_t n_pout = (size_t)B * po, n_pn = (size_t)B * (lst + 1) * nst, n_sn = (size_t)B * (L + 1) * n, n_z = (size_t)B * r;
  const size_t need = 2 * n_pout + 2 * n_pn + 2 * n_sn + 2 * n_z;
  if (need > ws_cap) {
    free(ws);
    ws = (float*)malloc(sizeof(float) * need);
    ws_cap = ws ? need : 0;
    if (!ws) return -1;
  }
  float* pout = ws;                 /* [B, po]  the tensor TF materialises */
  float* gp = pout + n_pout;        /* its gradient */
  float* pa = gp + n_pout;          /* pnet pre-activations */
  float* ph = pa + n_pn;            /* pnet layer outputs */
  float* sa_ = ph + n_pn;           /* snet pre-activations */
  float* sh = sa_ + n_sn;           /* snet layer outputs */
  float* z = sh + n_sn;
  float* gz = z + n_z;
  double loss = 0.0;

  /* ---- forward ------------------------------------------------------------------------------------- */
#pragma omp parallel for schedule(static) reduction(+ : loss)
  for (long a = 0; a < B; ++a) {
    const float* p = xin + a * (pi + si);
    const float* x = p + pi;
    float* A0 = pa + a * (lst + 1) * nst;
    float* H0 = ph + a * (lst + 1) * nst;
    for (int j = 0; j < nst; ++j) {
      float s = 0.f;
      for (int d = 0; d < pi; ++d) s += p[d] * theta[o_w0 + (long)d * nst + j];
      A0[j] = om_p * s + theta[o_b0 + j];
      H0[j] = act_f(c->p_act, A0[j]);
    }
    for (int i = 0; i < lst; ++i) {
      const float* hin = H0 + i * nst;
      float* ai = A0 + (i + 1) * nst;
      float* ho = H0 + (i + 1) * nst;
      for (int j = 0; j < nst; ++j) ai[j] = 0.f;
      for (int q = 0; q < nst; ++q) {
        const float hv = hin[q];
        const float* wrow = theta + o_wh[i] + (long)q * nst;
        for (int j = 0; j < nst; ++j) ai[j] += hv * wrow[j];
      }
      for (int j = 0; j < nst; ++j) {
        ai[j] = om_p * ai[j] + theta[o_bh[i] + j];
        ho[j] = siren_p ? sinf(ai[j]) : hin[j] + act_f(c->p_act, ai[j]);   /* SIREN | MLP_SimpleShortCut */
      }
    }
    const float* hl = H0 + lst * nst;
    for (int k = 0; k < r; ++k) {
      float s = theta[o_bb + k];
      for (int q = 0; q < nst; ++q) s += hl[q] * theta[o_wb + (long)q * r + k];
      z[a * r + k] = s;
    }
    /* pnet_output = z @ Wh + bh : the [B, po] tensor */
    float* w = pout + a * po;
    for (long s = 0; s < po; ++s) w[s] = theta[o_Bh + s];
    for (int k = 0; k < r; ++k) {
      const float zk = z[a * r + k];
      const float* row = theta + o_Wh + (long)k * po;
      for (long s = 0; s < po; ++s) w[s] += zk * row[s];
    }
    /* ShapeNet: einsum('ai,aij->aj') chain on this sample's slices */
    float* SA = sa_ + a * (L + 1) * n;
    float* SH = sh + a * (L + 1) * n;
    for (int j = 0; j < n; ++j) {
      float s = 0.f;
      for (int d = 0; d < si; ++d) s += x[d] * w[s_w1 + (long)d * n + j];
      SA[j] = om_s * s + w[s_b1 + j];
      SH[j] = act_f(sact, SA[j]);
    }
    for (int l = 0; l < L; ++l) {
      const float* hin = SH + l * n;
      float* al = SA + (l + 1) * n;
      float* ho = SH + (l + 1) * n;
      const float* W = w + s_wh + (long)l * n * n;
      for (int j = 0; j < n; ++j) al[j] = 0.f;
      for (int q = 0; q < n; ++q) {
        const float hv = hin[q];
        const float* wrow = W + (long)q * n;
        for (int j = 0; j < n; ++j) al[j] += hv * wrow[j];
      }
      for (int j = 0; j < n; ++j) {
        al[j] = om_s * al[j] + w[s_bh + (long)l * n + j];
        ho[j] = c->kind == 1 ? sinf(al[j]) : act_f(sact, al[j]) + hin[j];
      }
    }
    const float* hL = SH + L * n;
    const float wa = sw ? sw[a] : 1.0f;
    float se = 0.f;
    float gu[16];
    for (int o = 0; o < so; ++o) {
      float s = w[s_bl + o];
      for (int q = 0; q < n; ++q) s += hL[q] * w[s_wl + (long)q * so + o];
      const float e = s - y[a * so + o];
      se += e * e;
      gu[o] = 2.0f * wa * e / ((float)Bg * so);
    }
    loss += (double)wa * se / so / (double)Bg;
    /* ---- reverse sweep of the ShapeNet for this sample: fills g_pnet_out[a, :] ---------------------- */
    float* g = gp + a * po;
    float gh[256], gh2[256];
    for (int q = 0; q < n; ++q) {
      float s = 0.f;
      for (int o = 0; o < so; ++o) { g[s_wl + (long)q * so + o] = hL[q] * gu[o]; s += w[s_wl + (long)q * so + o] * gu[o]; }
      gh[q] = s;
    }
    for (int o = 0; o < so; ++o) g[s_bl + o] = gu[o];
    for (int l = L - 1; l >= 0; --l) {
      const float* hin = SH + l * n;
      const float* al = SA + (l + 1) * n;
      const float* W = w + s_wh + (long)l * n * n;
      float* gW = g + s_wh + (long)l * n * n;
      float ga[256];
      for (int j = 0; j < n; ++j) { ga[j] = gh[j] * act_d(sact, al[j]); g[s_bh + (long)l * n + j] = ga[j]; }
      for (int q = 0; q < n; ++q) {
        const float hv = om_s * hin[q];
        const float* wrow = W + (long)q * n;
        float* grow = gW + (long)q * n;
        float s = 0.f;
        for (int j = 0; j < n; ++j) { grow[j] = hv * ga[j]; s += wrow[j] * ga[j]; }
        gh2[q] = om_s * s + (c->kind == 1 ? 0.f : gh[q]);
      }
      for (int q = 0; q < n; ++q) gh[q] = gh2[q];
    }
    for (int j = 0; j < n; ++j) {
      const float ga = gh[j] * act_d(sact, SA[j]);
      g[s_b1 + j] = ga;
      for (int d = 0; d < si; ++d) g[s_w1 + (long)d * n + j] = om_s * x[d] * ga;
    }
    /* g_z = g_pnet_out @ Wh^T */
    for (int k = 0; k < r; ++k) {
      const float* row = theta + o_Wh + (long)k * po;
      float s = 0.f;
      for (long q = 0; q < po; ++q) s += g[q] * row[q];
      gz[a * r + k] = s;
    }
  }

  /* ---- gradients of the hyper layer: gWh = z^T g_pnet_out, gbh = sum_a g_pnet_out -------------------- */
#pragma omp parallel for schedule(static)
  for (long s0 = 0; s0 < po; s0 += 512) {   /* column blocks: rows of g_pnet_out are read contiguously */
    const int w = (int)(po - s0 < 512 ? po - s0 : 512);
    float gb[512], gw[512];
    for (int q = 0; q < w; ++q) gb[q] = 0.f;
    for (long a = 0; a < B; ++a) {
      const float* row = gp + a * po + s0;
      for (int q = 0; q < w; ++q) gb[q] += row[q];
    }
    for (int q = 0; q < w; ++q) grad[o_Bh + s0 + q] += gb[q];
    for (int k = 0; k < r; ++k) {
      for (int q = 0; q < w; ++q) gw[q] = 0.f;
      for (long a = 0; a < B; ++a) {
        const float zk = z[a * r + k];
        const float* row = gp + a * po + s0;
        for (int q = 0; q < w; ++q) gw[q] += zk * row[q];
      }
      for (int q = 0; q < w; ++q) grad[o_Wh + (long)k * po + s0 + q] += gw[q];
    }
  }
  /* ---- ParameterNet reverse sweep (small): per-thread accumulators ------------------------------------ */
  const long Pp = o_Wh;  /* number of pnet parameters before the hyper layer */
  int nth = nifref_max_threads();
  float* acc = (float*)calloc((size_t)nth * Pp, sizeof(float));
  if (!acc) return -1;
#pragma omp parallel
  {
#ifdef _OPENMP
    float* ga_ = acc + (size_t)omp_get_thread_num() * Pp;
#else
    float* ga_ = acc;
#endif
#pragma omp for schedule(static)
    for (long a = 0; a < B; ++a) {
      const float* p = xin + a * (pi + si);
      const float* A0 = pa + a * (lst + 1) * nst;
      const float* H0 = ph + a * (lst + 1) * nst;
      const float* hl = H0 + lst * nst;
      float gh[256], gn[256];
      for (int q = 0; q < nst; ++q) {
        float s = 0.f;
        for (int k = 0; k < r; ++k) { ga_[o_wb + (long)q * r + k] += hl[q] * gz[a * r + k]; s += theta[o_wb + (long)q * r + k] * gz[a * r + k]; }
        gh[q] = s;
      }
      for (int k = 0; k < r; ++k) ga_[o_bb + k] += gz[a * r + k];
      for (int i = lst - 1; i >= 0; --i) {
        const float* hin = H0 + i * nst;
        const float* ai = A0 + (i + 1) * nst;
        float gai[256];
        for (int j = 0; j < nst; ++j) { gai[j] = gh[j] * (siren_p ? cosf(ai[j]) : act_d(c->p_act, ai[j])); ga_[o_bh[i] + j] += gai[j]; }
        for (int q = 0; q < nst; ++q) {
          const float hv = om_p * hin[q];
          float s = 0.f;
          for (int j = 0; j < nst; ++j) { ga_[o_wh[i] + (long)q * nst + j] += hv * gai[j]; s += theta[o_wh[i] + (long)q * nst + j] * gai[j]; }
          gn[q] = om_p * s + (siren_p ? 0.f : gh[q]);
        }
        for (int q = 0; q < nst; ++q) gh[q] = gn[q];
      }
      for (int j = 0; j < nst; ++j) {
        const float g0 = gh[j] * act_d(c->p_act, A0[j]);
        ga_[o_b0 + j] += g0;
        for (int d = 0; d < pi; ++d) ga_[o_w0 + (long)d * nst + j] += om_p * p[d] * g0;
      }
    }
  }
  for (int t = 0; t < nth; ++t)
    for (long q = 0; q < Pp; ++q) grad[q] += acc[(size_t)t * Pp + q];
  free(acc);
  *loss_out += loss;
  return 0;
}

/* Keras-2.11 Adam, t = 1-based step */
void nifref_adam(float* theta, const float* g, float* m, float* v, long P, int t, float lr, float b1, float b2, float eps) {
  const double alpha = (double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t));
#pragma omp parallel for schedule(static)
  for (long i = 0; i < P; ++i) {
    m[i] += (g[i] - m[i]) * (1.0f - b1);
    v[i] += (g[i] * g[i] - v[i]) * (1.0f - b2);
    theta[i] -= (float)alpha * m[i] / (sqrtf(v[i]) + eps);
  }
}
