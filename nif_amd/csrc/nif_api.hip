// nif_api.hip -- C-ABI of libnif_hip.so (include/nif_hip.h): context, parameter layout, kernel
// orchestration.  gfx950 only; there is no CPU fallback.
#include "../../include/nif_hip.h"
#include "nif_internal.h"
#include <cstdlib>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "nif_ctx.h"

#define NIF_ACT_SLABS 32   // point slabs of the activity regulariser's plane pass
static inline bool act_on(const nif_ctx* c) { return c->act_l1 != 0.f || c->act_l2 != 0.f; }
static int jac_reg_pass(nif_ctx* c, const float* xin, long B, long Bg, const int* mu_blk = nullptr);
// Sobolev streams of one x_index: coordinate seeds first, then the parameter seeds (their pseudo-tiles trail the stashes, so
// that the first-layer reduction simply stops in front of them); gcol = the x_index position (column of dydx) of each stream
struct SobPlan { int ns, nsc; int seeds[3]; int par[3]; int gcol[3]; bool any_par;
                 int gstride, nx_all, ny, no_primal; unsigned ymask; };      // r4: one group of the x_index columns / a y_index subset (SobPar)
static int sob_par_pass(nif_ctx* c, const float* xin, long B, long Bg, const SobPlan& sp, const SNetArgs& sa);
static int fill_snet_ll_sob(nif_ctx* c, SNetArgs& sa, const float* xin, long B, bool f32_planes = false);

#ifndef NIF_PIPE_CHUNK_DEFAULT
#define NIF_PIPE_CHUNK_DEFAULT 131072L
#endif

static thread_local std::string g_err;
int nif_fail(int code, const std::string& msg) { g_err = msg; return code; }

extern "C" const char* nif_last_error(void) { return g_err.c_str(); }
extern "C" int nif_abi_version(void) { return NIF_ABI_VERSION; }
extern "C" int nif_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static void add_desc(nif_ctx* c, const char* name, long& off, int rows, int cols) {
  nif_tensor_desc d;
  memset(&d, 0, sizeof(d));
  snprintf(d.name, sizeof(d.name), "%s", name);
  d.offset = off; d.rows = rows; d.cols = cols;
  c->layout.push_back(d);
  off += (long)rows * (cols ? cols : 1);
}

// Keras variable order (nif/model.py:178-231, :591-734, :1162-1215)
static int build_layout(nif_ctx* c) {
  long off = 0;
  char nm[48];
  c->first_w = off; add_desc(c, "pnet_first_w", off, c->pi, c->nst);
  c->first_b = off; add_desc(c, "pnet_first_b", off, c->nst, 0);
  for (int i = 0; i < c->lst; ++i) {
    snprintf(nm, sizeof(nm), "pnet_h%d_w", i); c->hid_w[i] = off; add_desc(c, nm, off, c->nst, c->nst);
    snprintf(nm, sizeof(nm), "pnet_h%d_b", i); c->hid_b[i] = off; add_desc(c, nm, off, c->nst, 0);
    if (c->cfg.p_resblock) {
      snprintf(nm, sizeof(nm), "pnet_h%d_w2", i); c->hid_w2[i] = off; add_desc(c, nm, off, c->nst, c->nst);
      snprintf(nm, sizeof(nm), "pnet_h%d_b2", i); c->hid_b2[i] = off; add_desc(c, nm, off, c->nst, 0);
    }
  }
  c->bott_w = off; add_desc(c, "pnet_bottleneck_w", off, c->nst, c->r);
  c->bott_b = off; add_desc(c, "pnet_bottleneck_b", off, c->r, 0);
  c->last_w = off; add_desc(c, "pnet_last_w", off, c->r, (int)c->po);
  c->last_b = off; add_desc(c, "pnet_last_b", off, (int)c->po, 0);
  if (c->kind == NIF_KIND_LASTLAYER) {
    const int nout = (int)c->po * c->so;
    c->s_first_w = off; add_desc(c, "snet_first_w", off, c->si, c->n);
    c->s_first_b = off; add_desc(c, "snet_first_b", off, c->n, 0);
    for (int i = 0; i < c->L; ++i) {
      snprintf(nm, sizeof(nm), "snet_h%d_w", i); c->s_hid_w[i] = off; add_desc(c, nm, off, c->n, c->n);
      snprintf(nm, sizeof(nm), "snet_h%d_b", i); c->s_hid_b[i] = off; add_desc(c, nm, off, c->n, 0);
      if (c->cfg.s_resblock) {
        snprintf(nm, sizeof(nm), "snet_h%d_w2", i); c->s_hid_w2[i] = off; add_desc(c, nm, off, c->n, c->n);
        snprintf(nm, sizeof(nm), "snet_h%d_b2", i); c->s_hid_b2[i] = off; add_desc(c, nm, off, c->n, 0);
      }
    }
    c->s_bott_w = off; add_desc(c, "snet_bottleneck_w", off, c->n, nout);
    c->s_bott_b = off; add_desc(c, "snet_bottleneck_b", off, nout, 0);
    c->ll_bias = off; add_desc(c, "last_layer_bias", off, c->so, 0);
  }
  c->P = off;
  return 0;
}

extern "C" int nif_create(const nif_cfg* cfg, int device_id, nif_ctx** out) {
  if (!cfg || !out) return fail(NIF_ERR_INVALID, "null argument");
  if (cfg->abi_version != NIF_ABI_VERSION) return fail(NIF_ERR_INVALID, "abi_version mismatch");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NIF_ERR_NODEVICE, "no HIP device visible: libnif_hip has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail(NIF_ERR_INVALID, "device_id out of range");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NIF_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", libnif_hip is built for gfx950 only");
  if (cfg->kind != NIF_KIND_NIF && cfg->kind != NIF_KIND_MULTISCALE && cfg->kind != NIF_KIND_LASTLAYER)
    return fail(NIF_ERR_INVALID, "unknown model kind");
  if (cfg->mixed_policy != NIF_POLICY_FLOAT32 && cfg->mixed_policy != NIF_POLICY_MIXED_BF16 && cfg->mixed_policy != NIF_POLICY_MIXED_F16)
    return fail(NIF_ERR_INVALID, "unknown mixed_policy");
  for (int i = 0; i < 7; ++i) if (cfg->reserved[i] != 0) return fail(NIF_ERR_INVALID, "reserved fields must be zero");
  if (cfg->p_act < 0 || cfg->p_act > NIF_ACT_HARD_SIGMOID || cfg->s_act < 0 || cfg->s_act > NIF_ACT_HARD_SIGMOID)
    return fail(NIF_ERR_INVALID, "unknown activation id (nif_act)");
  if (cfg->kind == NIF_KIND_LASTLAYER && cfg->latent_dim * cfg->so_dim > 64)
    return fail(NIF_ERR_INVALID, "last-layer class: latent_dim * output_dim must be <= 64");
  if (cfg->pi_dim < 1 || cfg->si_dim < 1 || cfg->so_dim < 1 || cfg->latent_dim < 1)
    return fail(NIF_ERR_INVALID, "dims must be >= 1");
  if (cfg->n_sx < 1 || cfg->n_sx > 128 || cfg->n_st < 1 || cfg->n_st > 128)
    return fail(NIF_ERR_INVALID, "units must be in [1,128]");
  if (cfg->latent_dim > 64) return fail(NIF_ERR_INVALID, "latent_dim must be <= 64");
  if (cfg->pi_dim > 16 || cfg->si_dim > 16 || cfg->so_dim > 16)
    return fail(NIF_ERR_INVALID, "input/output dims must be <= 16");
  const int nh = cfg->l_sx * (cfg->s_resblock ? 2 : 1);
  const int nm = cfg->l_st * (cfg->p_resblock ? 2 : 1);
  if (cfg->l_sx < 0 || nh > NIF_MAX_HID || cfg->l_st < 0 || cfg->l_st > NIF_MAX_HID)
    return fail(NIF_ERR_INVALID, "too many layers");
  if (cfg->kind == NIF_KIND_NIF && (cfg->s_resblock || cfg->p_resblock || cfg->p_act == NIF_ACT_SINE))
    return fail(NIF_ERR_INVALID, "class NIF has no resblock / sine ParameterNet");
  if (cfg->p_resblock && cfg->kind == NIF_KIND_NIF) return fail(NIF_ERR_INVALID, "bad cfg");

  nif_ctx* c = new nif_ctx();
  c->cfg = *cfg;
  { const char* e = getenv("NIF_FP32_MFMA"); c->opt_fp32_mfma = e && e[0] == '1'; }
  { const char* e = getenv("NIF_FUSE_GW"); c->opt_fuse_gw = !(e && e[0] == '0'); }
  { const char* e = getenv("NIF_SMALL_STEP"); c->opt_small_step = !(e && e[0] == '0'); }
  { const char* e = getenv("NIF_FUSE_TAIL"); c->opt_fuse_tail = !(e && e[0] == '0'); }
  { const char* e = getenv("NIF_PIPE_CHUNK"); if (e && e[0]) c->opt_pipe_chunk = atol(e); }
  { const char* e = getenv("NIF_SIDE_PNET"); if (e && e[0]) c->opt_side_pnet = e[0] != '0'; }
  { const char* e = getenv("NIF_PIPE_WGS"); if (e && e[0]) c->opt_pipe_wgs = atoi(e); }
  c->dev = device_id;
  c->kind = cfg->kind; c->pi = cfg->pi_dim; c->si = cfg->si_dim; c->so = cfg->so_dim;
  c->n = cfg->n_sx; c->L = cfg->l_sx; c->nst = cfg->n_st; c->lst = cfg->l_st; c->r = cfg->latent_dim;
  c->nh = nh; c->nm = nm;
  c->NB = c->n <= 32 ? 1 : (c->n <= 64 ? 2 : 4);
  c->NSTB = c->nst <= 32 ? 1 : (c->nst <= 64 ? 2 : 4);
  c->po = (long)nh * c->n * c->n + (long)(c->si + c->so + 1 + nh) * c->n + c->so;
  if (c->kind == NIF_KIND_LASTLAYER) c->po = c->r;   // model.py:583-585
  c->RB = (c->r + 31) / 32;
  build_layout(c);
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
  const size_t pb = (size_t)(c->P + 2) * sizeof(float);   // [grad | loss | one scratch word of the two-level row reduction]
  if (e == hipSuccess) e = hipMalloc(&c->theta, pb);
  if (e == hipSuccess) e = hipMalloc(&c->grad, pb);
  if (e == hipSuccess) e = hipMalloc(&c->m, pb);
  if (e == hipSuccess) e = hipMalloc(&c->v, pb);
  if (e == hipSuccess) e = hipMemset(c->m, 0, pb);
  if (e == hipSuccess) e = hipMemset(c->v, 0, pb);
  if (e == hipSuccess) e = hipMemset(c->grad, 0, pb);
  const size_t pk_p = (size_t)(nm > 0 ? nm : 1) * c->NSTB * c->NSTB * 256 * sizeof(f32x4);
  size_t pk_s = (size_t)(nh > 0 ? nh : 1) * (c->r + 1) * c->NB * c->NB * 256 * sizeof(f32x4);
  {
    const size_t pk16 = (size_t)(nh > 0 ? nh : 1) * (c->r + 1) * snet3_plane_floats(c->n) * sizeof(float);
    if (pk16 > pk_s) pk_s = pk16;
  }
  if (e == hipSuccess) e = hipMalloc(&c->pWF, pk_p);
  if (e == hipSuccess) e = hipMalloc(&c->pWB, pk_p);
  if (e == hipSuccess) e = hipMalloc(&c->sWF, pk_s);
  if (e == hipSuccess) e = hipMalloc(&c->sWB, pk_s);
  if (nh > 0 && !(snet3_nbl(c->n) & 1) && c->n <= 128) {
    const int rr = c->kind == NIF_KIND_LASTLAYER ? 0 : c->r;   // last-layer class: shared dense weights, one plane
    if (e == hipSuccess) e = hipMalloc(&c->sWF4, (size_t)nh * snet4_fwd_elems(c->n, rr) * 2);
    if (e == hipSuccess) e = hipMalloc(&c->sWB4, (size_t)nh * snet4_bwd_elems(c->n, rr) * 2);
    if (c->kind != NIF_KIND_NIF) {   // SIREN nets (r5): the exact-product half planes of k_snet4<.., PR = 3> / k_snet6 (class NIF keeps the bf16 splits)
      if (e == hipSuccess) e = hipMalloc(&c->sWF4x, (size_t)nh * snet4_bwd_elems(c->n, rr) * 2);
      if (e == hipSuccess) e = hipMalloc(&c->sWB4x, (size_t)nh * snet4_bwd_elems(c->n, rr) * 2);
      if (e == hipSuccess) e = hipMalloc(&c->sWscale, sizeof(float) * (size_t)nh * (rr + 1) * 2);
    }
    if (c->cfg.mixed_policy != NIF_POLICY_FLOAT32) {     // the policy's compact plane set (k_snet4 / k_snet6<.., PR>), next to the exact splits
      if (e == hipSuccess) e = hipMalloc(&c->sWF4h, (size_t)nh * (snet4_fwd_elems(c->n, rr) / 3) * 2);
      if (e == hipSuccess) e = hipMalloc(&c->sWB4h, (size_t)nh * (snet4_bwd_elems(c->n, rr) / 2) * 2);
    }
    if (c->kind == NIF_KIND_LASTLAYER && e == hipSuccess) e = hipMalloc(&c->ll_wpf, (size_t)snet4_phi_fwd_elems(c->n) * 2);
    if (c->kind == NIF_KIND_LASTLAYER && e == hipSuccess) e = hipMalloc(&c->ll_wpb, (size_t)snet4_phi_bwd_elems(c->n) * 2);
  }
  // slot-ordered copy of the dense ShapeNet parameters of the last-layer class (k_snet4<LL>, k_sob<LL>, k_jac on the r = 0 arguments)
  if (c->kind == NIF_KIND_LASTLAYER && nh > 0 && c->n <= 128 && e == hipSuccess)
    e = hipMalloc(&c->ll_slots, sizeof(float) * (size_t)((long)c->si * c->n + (long)nh * c->n * c->n + (long)c->n * c->so * c->r +
                                                           c->n + (long)nh * c->n + c->so * c->r + c->so + c->r * c->r + 64));
  const size_t pk_l = (size_t)(nh > 0 ? nh : 1) * c->NB * c->NB * 256 * sizeof(f32x4);
  if (e == hipSuccess && c->kind == NIF_KIND_LASTLAYER) e = hipMalloc(&c->lWF, pk_l);
  if (e == hipSuccess && c->kind == NIF_KIND_LASTLAYER) e = hipMalloc(&c->lWB, pk_l);
  if (e != hipSuccess) {
    std::string msg = std::string("nif_create: ") + hipGetErrorString(e);
    delete c;
    return fail(NIF_ERR_HIP, msg);
  }
  *out = c;
  return NIF_OK;
}

extern "C" int nif_destroy(nif_ctx* c) {
  if (!c) return NIF_OK;
  hipSetDevice(c->dev);
  if (c->st) hipStreamSynchronize(c->st);
  if (c->small_idx) hipFree(c->small_idx);
  if (c->small_desc) hipFree(c->small_desc);
  if (c->comm) (void)nif_comm_destroy(c);
  if (c->st2) { hipStreamSynchronize(c->st2); hipStreamDestroy(c->st2); }
  if (c->st_copy) { hipStreamSynchronize(c->st_copy); hipStreamDestroy(c->st_copy); }
  for (int s_ = 0; s_ < 2; ++s_) { if (c->ev_copied[s_]) hipEventDestroy(c->ev_copied[s_]); if (c->ev_consumed[s_]) hipEventDestroy(c->ev_consumed[s_]); }
  if (c->ev_start) hipEventDestroy(c->ev_start);
  if (c->ev_done) hipEventDestroy(c->ev_done);
  for (hipEvent_t e : c->ev_chunk) hipEventDestroy(e);
  for (hipGraphExec_t ex : c->graphs) if (ex) (void)hipGraphExecDestroy(ex);
  if (c->adam_host) (void)hipHostFree(c->adam_host);
  void* ptrs[] = {c->adam_dev, c->sob_acc, c->comm_scratch, c->chunk_grad, c->act_part, c->act_loss, c->jac_mu, c->jac_tmp, c->zt_par, c->dzt_par, c->dat_par, c->ztl_par, c->theta, c->grad, c->m, c->v, c->pWF, c->pWB, c->sWF, c->sWB, c->stash_s, c->stash_p, c->Z, c->DZ,
                  c->DU, c->ZL, c->partial, c->loss_partial, c->dring, c->metric, c->tl, c->lWF, c->lWB, c->sWF4, c->sWB4, c->sWF4x, c->sWB4x, c->sWscale, c->sWF4h, c->sWB4h, c->ll_slots, c->ll_wpf, c->ll_wpb, c->stash_l, c->PHI, c->DPHI, c->DA, c->DZL, c->d_a, c->d_b, c->d_c, c->d_d};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->st) hipStreamDestroy(c->st);
  delete c;
  return NIF_OK;
}

extern "C" int nif_param_count(nif_ctx* c, int64_t* n) { if (!c || !n) return fail(NIF_ERR_INVALID, "null"); *n = c->P; return NIF_OK; }
extern "C" int nif_po_dim(nif_ctx* c, int64_t* po) { if (!c || !po) return fail(NIF_ERR_INVALID, "null"); *po = c->po; return NIF_OK; }
extern "C" int nif_param_layout(nif_ctx* c, nif_tensor_desc* descs, int32_t* n_inout) {
  if (!c || !n_inout) return fail(NIF_ERR_INVALID, "null");
  const int need = (int)c->layout.size();
  if (!descs || *n_inout < need) { *n_inout = need; return descs ? fail(NIF_ERR_INVALID, "descs too small") : NIF_OK; }
  memcpy(descs, c->layout.data(), sizeof(nif_tensor_desc) * need);
  *n_inout = need;
  return NIF_OK;
}

extern "C" int nif_set_params(nif_ctx* c, const float* host, int64_t n) {
  if (!c || !host) return fail(NIF_ERR_INVALID, "null");
  if (n != c->P) return fail(NIF_ERR_INVALID, "parameter count mismatch");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(c->theta, host, sizeof(float) * n, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  c->have_params = true; c->packed = false; c->packed32 = false; c->packed_p32 = false;
  return NIF_OK;
}
extern "C" int nif_get_params(nif_ctx* c, float* host, int64_t n) {
  if (!c || !host) return fail(NIF_ERR_INVALID, "null");
  if (n != c->P) return fail(NIF_ERR_INVALID, "parameter count mismatch");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(host, c->theta, sizeof(float) * n, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}
extern "C" int nif_get_opt_state(nif_ctx* c, float* mh, float* vh, int64_t n, int64_t* step) {
  if (!c || !mh || !vh || !step) return fail(NIF_ERR_INVALID, "null");
  if (n != c->P) return fail(NIF_ERR_INVALID, "parameter count mismatch");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(mh, c->m, sizeof(float) * n, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemcpyAsync(vh, c->v, sizeof(float) * n, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  *step = c->step;
  return NIF_OK;
}
extern "C" int nif_set_opt_state(nif_ctx* c, const float* mh, const float* vh, int64_t n, int64_t step) {
  if (!c || !mh || !vh) return fail(NIF_ERR_INVALID, "null");
  if (n != c->P) return fail(NIF_ERR_INVALID, "parameter count mismatch");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(c->m, mh, sizeof(float) * n, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipMemcpyAsync(c->v, vh, sizeof(float) * n, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  c->step = step;
  return NIF_OK;
}

extern "C" int nif_dev_alloc(nif_ctx* c, int64_t bytes, void** dptr) {
  if (!c || !dptr || bytes < 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMalloc(dptr, bytes > 0 ? (size_t)bytes : 4));
  return NIF_OK;
}
extern "C" int nif_dev_free(nif_ctx* c, void* dptr) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  if (dptr) HIPCHK(hipFree(dptr));
  return NIF_OK;
}
extern "C" int nif_h2d(nif_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c || !dst || !src) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}
extern "C" int nif_d2h(nif_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c || !dst || !src) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}
extern "C" int nif_sync(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}
// ---- shard streaming: pinned host staging + asynchronous H2D on a copy stream, double buffered -----------------------
extern "C" int nif_host_alloc(nif_ctx* c, int64_t bytes, void** hptr) {
  if (!c || !hptr || bytes < 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipHostMalloc(hptr, bytes > 0 ? (size_t)bytes : 4, hipHostMallocDefault));
  return NIF_OK;
}
extern "C" int nif_host_free(nif_ctx* c, void* hptr) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  if (c->st_copy) HIPCHK(hipStreamSynchronize(c->st_copy));
  if (hptr) HIPCHK(hipHostFree(hptr));
  return NIF_OK;
}
static int ensure_copy_stream(nif_ctx* c) {
  if (c->st_copy) return NIF_OK;
  HIPCHK(hipStreamCreateWithFlags(&c->st_copy, hipStreamNonBlocking));
  for (int s = 0; s < 2; ++s) {
    HIPCHK(hipEventCreateWithFlags(&c->ev_copied[s], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_consumed[s], hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_copied[s], c->st_copy));       // both start out signalled
    HIPCHK(hipEventRecord(c->ev_consumed[s], c->st));
  }
  return NIF_OK;
}
extern "C" int nif_h2d_async(nif_ctx* c, void* dst_dev, const void* src_pinned, int64_t bytes, int32_t slot) {
  if (!c || !dst_dev || !src_pinned || bytes < 0 || slot < 0 || slot > 1) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = ensure_copy_stream(c); if (rc) return rc;
  HIPCHK(hipStreamWaitEvent(c->st_copy, c->ev_consumed[slot], 0));   // the steps that read this slot's device buffer are done
  HIPCHK(hipMemcpyAsync(dst_dev, src_pinned, (size_t)bytes, hipMemcpyHostToDevice, c->st_copy));
  HIPCHK(hipEventRecord(c->ev_copied[slot], c->st_copy));
  return NIF_OK;
}
extern "C" int nif_copy_acquire(nif_ctx* c, int32_t slot) {       // compute stream: wait until the slot's copies have landed
  if (!c || slot < 0 || slot > 1) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = ensure_copy_stream(c); if (rc) return rc;
  HIPCHK(hipStreamWaitEvent(c->st, c->ev_copied[slot], 0));
  return NIF_OK;
}
extern "C" int nif_copy_release(nif_ctx* c, int32_t slot) {       // compute stream: everything enqueued so far has consumed the slot
  if (!c || slot < 0 || slot > 1) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = ensure_copy_stream(c); if (rc) return rc;
  HIPCHK(hipEventRecord(c->ev_consumed[slot], c->st));
  return NIF_OK;
}
extern "C" int nif_copy_wait_host(nif_ctx* c, int32_t slot) {     // host: the slot's pinned staging buffer may be overwritten
  if (!c || slot < 0 || slot > 1) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = ensure_copy_stream(c); if (rc) return rc;
  HIPCHK(hipEventSynchronize(c->ev_copied[slot]));
  return NIF_OK;
}
extern "C" int nif_gather_rows_dev(nif_ctx* c, const float* src_dev, const int32_t* perm_dev, int64_t n, int32_t ncol, float* dst_dev) {
  if (!c || !src_dev || !perm_dev || !dst_dev || n < 0 || ncol < 1) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  if (n > 0) launch_gather_rows(src_dev, perm_dev, n, ncol, dst_dev, c->st);
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

extern "C" void* nif_stream(nif_ctx* c) { return c ? (void*)c->st : nullptr; }
extern "C" void* nif_grad_dev(nif_ctx* c) { if (c) (void)nif_tail_flush(c); return c ? (void*)c->grad : nullptr; }
extern "C" void* nif_params_dev(nif_ctx* c) { return c ? (void*)c->theta : nullptr; }

// ------------------------------------------------------------------------------------------
// internal orchestration
// ------------------------------------------------------------------------------------------
static int grow(float** p, long* cap, long need) {
  if (need <= *cap) return NIF_OK;
  if (*p) HIPCHK(hipFree(*p));
  *p = nullptr; *cap = 0;
  HIPCHK(hipMalloc(p, sizeof(float) * (size_t)need));
  *cap = need;
  return NIF_OK;
}

static int ensure_capacity(nif_ctx* c, long B, bool train) {
  const long ntiles = (B + 31) / 32;
  const long pts = ntiles * 32;
  if (pts > c->cap || (train && !c->stash_s)) {
    HIPCHK(hipStreamSynchronize(c->st));
    const long newcap = pts > c->cap ? pts : c->cap;
    const bool ll = c->kind == NIF_KIND_LASTLAYER;
    float** ws[] = {&c->Z, &c->DZ, &c->DU, &c->ZL, &c->PHI, &c->DPHI, &c->DA, &c->DZL};
    const long wsz[] = {(long)c->r * newcap, (long)c->r * newcap, (long)c->so * newcap, (long)32 * c->RB * newcap,
                        (long)c->r * c->so * newcap, (long)c->r * c->so * newcap, (long)c->r * newcap, (long)c->r * newcap};
    if (newcap > c->cap)
      for (int i = 0; i < (ll ? 8 : 4); ++i) {
        if (*ws[i]) HIPCHK(hipFree(*ws[i]));
        *ws[i] = nullptr;
        HIPCHK(hipMalloc(ws[i], sizeof(float) * (size_t)wsz[i]));
      }
    if (train) {
      if (c->stash_s) HIPCHK(hipFree(c->stash_s));
      if (c->stash_p) HIPCHK(hipFree(c->stash_p));
      c->stash_s = c->stash_p = nullptr;
      c->slot_s = (long)c->NB * 32 * newcap;
      c->slot_p = (long)c->NSTB * 32 * newcap;
      // r6 (fuzz case 606/16): k_snet6 keeps its private h ring in this buffer -- [workgroups][8 waves][nh][64 features][16 points],
      // EVERY wave of a workgroup writes its slice, active or not -- and for batches of <= 32 points of a net with nh >= 2 that is more
      // than the 2 (nh + 1) stash slots of one 32-point tile: the ring ran 32 KB past the allocation into the ParameterNet's stash
      // (wrong ParameterNet gradients on the stash path: ParameterNets with > 2 hidden matrices or > 32 units)
      long s_floats = c->slot_s * 2 * (c->nh + 1);
      {
        long nblk6 = (2 * (newcap / 32) + 7) / 8; if (nblk6 > 256) nblk6 = 256; if (nblk6 < 1) nblk6 = 1;
        const long ring6 = nblk6 * 8 * (long)c->nh * 64 * 16;
        if (ring6 > s_floats) s_floats = ring6;
      }
      HIPCHK(hipMalloc(&c->stash_s, sizeof(float) * (size_t)s_floats));
      HIPCHK(hipMalloc(&c->stash_p, sizeof(float) * (size_t)(c->slot_p * (2 * c->nm + 2))));
      if (c->NB * 32 > ((c->n + 15) / 16) * 16)    // 65..96 (and 33..48) units: the rows the fused kernels never write must read as zero
        HIPCHK(hipMemsetAsync(c->stash_s, 0, sizeof(float) * (size_t)s_floats, c->st));
    } else if (newcap > c->cap && c->stash_s) {
      HIPCHK(hipFree(c->stash_s)); HIPCHK(hipFree(c->stash_p));
      c->stash_s = c->stash_p = nullptr;
    }
    c->cap = newcap;
  }
  if (train) {
    // one loss partial per workgroup of the fused kernel: the 16-point-tile kernels launch up to ceil(2*ntiles/4)
    // workgroups (k_snet3/4, k_sob), k_ll_out ntiles/8, k_snet ntiles/4
    // (r6: k_small one per 16 points = 2 ntiles)
    const long nblk = 2 * ntiles + 1;
    if (nblk > c->nloss_cap) {
      HIPCHK(hipStreamSynchronize(c->st));
      if (c->loss_partial) HIPCHK(hipFree(c->loss_partial));
      c->loss_partial = nullptr;
      HIPCHK(hipMalloc(&c->loss_partial, sizeof(float) * (size_t)nblk));
      c->nloss_cap = nblk;
    }
    if (!c->partial) {
      c->rows_cap = 256;
      c->pstride = (c->P + 1 + 63) / 64 * 64;
      HIPCHK(hipMalloc(&c->partial, sizeof(float) * (size_t)c->rows_cap * c->pstride));
    }
  }
  return NIF_OK;
}

static MatRef dense_ref(long w_off, int nin, int nout) { MatRef m; m.r = 0; m.base_k = 0; m.kstride = 0; m.base_last = w_off; m.ld = nout; m.nin = nin; m.nout = nout; return m; }
static MatRef vec_ref(long b_off, int nout) { MatRef m; m.r = 0; m.base_k = 0; m.kstride = 0; m.base_last = b_off; m.ld = 0; m.nin = 1; m.nout = nout; return m; }
static MatRef hyper_ref(const nif_ctx* c, long slot, int ld, int nin, int nout) {
  MatRef m; m.r = c->r; m.base_k = c->last_w + slot; m.kstride = c->po; m.base_last = c->last_b + slot; m.ld = ld; m.nin = nin; m.nout = nout; return m;
}

static void fill_pnet(const nif_ctx* c, PNetArgs& a, const float* xin, long B) {
  memset(&a, 0, sizeof(a));
  a.theta = c->theta; a.xin = xin; a.ncol = c->pi + c->si; a.col0 = 0; a.B = B;
  a.pi = c->pi; a.nst = c->nst; a.lst = c->lst; a.r = c->r;
  a.act = c->cfg.p_act; a.res = c->cfg.p_resblock; a.siren = (c->cfg.p_act == NIF_ACT_SINE);
  a.omega = a.siren ? c->cfg.p_omega0 : 1.0f;
  a.first_w = c->first_w; a.first_b = c->first_b;
  for (int i = 0; i < c->lst; ++i) { a.hid_w[i] = c->hid_w[i]; a.hid_b[i] = c->hid_b[i]; a.hid_w2[i] = c->hid_w2[i]; a.hid_b2[i] = c->hid_b2[i]; }
  a.bott_w = c->bott_w; a.bott_b = c->bott_b; a.last_w = c->last_w; a.last_b = c->last_b;
  a.ll_kind = (c->kind == NIF_KIND_LASTLAYER); a.zl_rows = 32 * c->RB;
  a.WF = c->pWF; a.WB = c->pWB; a.stash = c->stash_p; a.slot_stride = c->slot_p;
  a.Z = c->Z; a.DZ = a.ll_kind ? c->DZL : c->DZ; a.ZL = c->ZL;
}
// last-layer class: the shared-weight SIREN ShapeNet x -> phi is the same MLP machinery (model.py:1219-1238)
// f32-input MFMA planes of the last-layer class's shared ShapeNet for the 32-point MLP kernels (k_pnet as the dense SIREN: the NIF_LL_MLP
// path, nets k_snet4<LL> does not take, model_x_to_phi / x_to_u_given_w).  r4: on demand -- the default training step (k_snet4<LL>) never
// reads them, and packing them after every Adam step cost L launches per step
static void ensure_ll_mlp_planes(nif_ctx* c) {
  if (c->ll_mlp_packed) return;
  const long plane_l = (long)c->NB * c->NB * 256;
  for (int i = 0; i < c->L; ++i) {
    if (!c->cfg.s_resblock) {
      launch_pack(c->theta, dense_ref(c->s_hid_w[i], c->n, c->n), c->NB, c->NB, c->lWF + i * plane_l, c->lWB + i * plane_l, c->st);
    } else {
      launch_pack(c->theta, dense_ref(c->s_hid_w[i], c->n, c->n), c->NB, c->NB, c->lWF + (2 * i) * plane_l, c->lWB + (2 * i) * plane_l, c->st);
      launch_pack(c->theta, dense_ref(c->s_hid_w2[i], c->n, c->n), c->NB, c->NB, c->lWF + (2 * i + 1) * plane_l, c->lWB + (2 * i + 1) * plane_l, c->st);
    }
  }
  c->ll_mlp_packed = true;
}
static void fill_snet_mlp(const nif_ctx* c, PNetArgs& a, const float* xin, int ncol, int col0, long B) {
  memset(&a, 0, sizeof(a));
  a.theta = c->theta; a.xin = xin; a.ncol = ncol; a.col0 = col0; a.B = B;
  a.pi = c->si; a.nst = c->n; a.lst = c->L; a.r = c->r * c->so;
  a.act = NIF_ACT_SINE; a.res = c->cfg.s_resblock; a.siren = 1; a.omega = c->cfg.s_omega0;
  a.first_w = c->s_first_w; a.first_b = c->s_first_b;
  for (int i = 0; i < c->L; ++i) { a.hid_w[i] = c->s_hid_w[i]; a.hid_b[i] = c->s_hid_b[i]; a.hid_w2[i] = c->s_hid_w2[i]; a.hid_b2[i] = c->s_hid_b2[i]; }
  a.bott_w = c->s_bott_w; a.bott_b = c->s_bott_b; a.ll_kind = 0;
  a.WF = c->lWF; a.WB = c->lWB; a.stash = c->stash_s; a.slot_stride = c->slot_s;
  a.Z = c->PHI; a.DZ = c->DPHI; a.ZL = nullptr;
}
static void fill_ll(const nif_ctx* c, LLArgs& a, long B) {
  memset(&a, 0, sizeof(a));
  a.theta = c->theta; a.bias_off = c->ll_bias; a.last_w = c->last_w; a.loss_kind = c->loss_kind;
  a.PHI = c->PHI; a.Z = c->Z; a.B = B; a.r = c->r; a.so = c->so;
  a.DU = c->DU; a.DPHI = c->DPHI; a.DA = c->DA; a.DZL = c->DZL; a.loss_partial = c->loss_partial;
}
// last-layer class on k_snet4: slot offsets as in k_snet4's prologue (r = 0)
static void fill_snet_ll(const nif_ctx* c, SNetArgs& a, const float* xin, int ncol, int col0, long B) {
  memset(&a, 0, sizeof(a));
  const int sop = c->so * c->r;
  a.theta = c->ll_slots; a.xin = xin; a.ncol = ncol; a.col0 = col0; a.B = B; a.loss_kind = c->loss_kind;
  a.si = c->si; a.so = sop; a.n = c->n; a.nh = c->nh; a.r = 0; a.po = 0;
  a.act = NIF_ACT_SINE; a.res = c->cfg.s_resblock; a.nif_skip = 0; a.omega = c->cfg.s_omega0;
  a.off_Wh = 0; a.off_bh = 0;
  a.Z = c->Z; a.WF4 = c->sWF4; a.WB4 = c->sWB4; a.stash = c->stash_s; a.slot_stride = c->slot_s;
  a.DU = c->DU; a.DZ = nullptr; a.dring = c->dring;
  a.ll = 1; a.rl = c->r; a.so_u = c->so; a.DPHI = c->DPHI; a.DA_ll = c->DA; a.DZL = c->DZL;
  a.WPF = c->ll_wpf; a.WPB = c->ll_wpb;
  a.prec = c->opt_fp32_mfma ? 0 : (c->cfg.mixed_policy == NIF_POLICY_MIXED_BF16 ? 1 : (c->cfg.mixed_policy == NIF_POLICY_MIXED_F16 ? 2 : 0));     // (k_snet4<LL> only; k_sob / k_jac stay exact)
  a.WF4h = c->sWF4h; a.WB4h = c->sWB4h;
  a.WF4x = c->sWF4x; a.WB4x = c->sWB4x; a.wscale = c->sWscale;
  a.nsm = snet4_nsm_ll(c->si, sop, c->nh, c->n, c->so, c->r);
  a.tl = c->tl;
}
static void fill_snet(const nif_ctx* c, SNetArgs& a, const float* xin, int ncol, int col0, long B) {
  memset(&a, 0, sizeof(a));
  a.theta = c->theta; a.xin = xin; a.ncol = ncol; a.col0 = col0; a.B = B; a.loss_kind = c->loss_kind;
  a.si = c->si; a.so = c->so; a.n = c->n; a.nh = c->nh; a.r = c->r; a.po = c->po;
  a.act = c->kind == NIF_KIND_NIF ? c->cfg.s_act : NIF_ACT_SINE;
  a.res = c->cfg.s_resblock; a.nif_skip = (c->kind == NIF_KIND_NIF);
  a.omega = c->kind == NIF_KIND_NIF ? 1.0f : c->cfg.s_omega0;
  a.off_Wh = c->last_w; a.off_bh = c->last_b;
  a.Z = c->Z; a.WF = c->sWF; a.WB = c->sWB; a.stash = c->stash_s; a.slot_stride = c->slot_s;
  a.DU = c->DU; a.DZ = c->DZ;
  a.nsm = snet3_nsm(c->si, c->so, c->nh, c->n);
  a.WF4 = c->use_snet4 ? c->sWF4 : nullptr; a.WB4 = c->use_snet4 ? c->sWB4 : nullptr;   // packed only then
  a.WF4x = c->use_snet4 ? c->sWF4x : nullptr; a.WB4x = c->use_snet4 ? c->sWB4x : nullptr; a.wscale = c->sWscale;
  a.prec = c->opt_fp32_mfma ? 0 : (c->cfg.mixed_policy == NIF_POLICY_MIXED_BF16 ? 1 : (c->cfg.mixed_policy == NIF_POLICY_MIXED_F16 ? 2 : 0));
  if (!c->use_snet4) a.prec = a.prec == 2 ? 0 : a.prec;      // (no k_snet4 for this shape: the policy runs on the exact kernels)
  a.WF4h = c->use_snet4 ? c->sWF4h : nullptr; a.WB4h = c->use_snet4 ? c->sWB4h : nullptr;
  a.dring = c->dring;
  a.tl = c->tl;
}

// fp32 MFMA planes of the hidden hyper-matrices: needed by k_snet3 / k_snet (when the bf16-split kernel is not in
// use), k_jac and k_sob -- packed on demand, the training step on k_snet4 never pays for them
static int ensure_packed32(nif_ctx* c) {
  if (c->packed32 || c->kind == NIF_KIND_LASTLAYER) return NIF_OK;
  ProfScope ps_(c, NIF_PROF_PACK);
  const bool fmt16 = c->use_snet3 || c->jac_ok;     // 16-point-tile plane format (k_snet3, k_jac, k_sob) / 32-point (k_snet)
  const long plane_s = fmt16 ? snet3_plane_floats(c->n) / 4 : (long)c->NB * c->NB * 256;
  for (int j = 0; j < c->nh; ++j) {
    const long slot = (long)c->si * c->n + (long)j * c->n * c->n;
    f32x4* wf = c->sWF + (long)j * (c->r + 1) * plane_s;
    f32x4* wb = c->sWB + (long)j * (c->r + 1) * plane_s;
    if (fmt16) launch_pack16(c->theta, hyper_ref(c, slot, c->n, c->n, c->n), snet3_nbl(c->n), wf, wb, c->st);
    else launch_pack(c->theta, hyper_ref(c, slot, c->n, c->n, c->n), c->NB, c->NB, wf, wb, c->st);
  }
  HIPCHK(hipGetLastError());
  c->packed32 = true;
  return NIF_OK;
}
// fp32 MFMA planes of the ParameterNet's hidden matrices.  A 32-wide ParameterNet (k_pnet<1>, k_pnet_bwg) splits
// its weights into bf16 operands itself, straight from theta -- the planes are then only needed by the stash-path
// adjoint (k_pnet_bwd) and k_mlpjac, and packed on demand
static int ensure_packed_p32(nif_ctx* c) {
  if (c->packed_p32) return NIF_OK;
  ProfScope ps_(c, NIF_PROF_PACK);
  const long plane_p = (long)c->NSTB * c->NSTB * 256;
  for (int i = 0; i < c->lst; ++i) {
    if (!c->cfg.p_resblock) {
      launch_pack(c->theta, dense_ref(c->hid_w[i], c->nst, c->nst), c->NSTB, c->NSTB, c->pWF + i * plane_p, c->pWB + i * plane_p, c->st);
    } else {
      launch_pack(c->theta, dense_ref(c->hid_w[i], c->nst, c->nst), c->NSTB, c->NSTB, c->pWF + (2 * i) * plane_p, c->pWB + (2 * i) * plane_p, c->st);
      launch_pack(c->theta, dense_ref(c->hid_w2[i], c->nst, c->nst), c->NSTB, c->NSTB, c->pWF + (2 * i + 1) * plane_p, c->pWB + (2 * i + 1) * plane_p, c->st);
    }
  }
  HIPCHK(hipGetLastError());
  c->packed_p32 = true;
  return NIF_OK;
}
static int ensure_packed(nif_ctx* c) {
  if (!c->have_params) return fail(NIF_ERR_STATE, "parameters not set (call nif_set_params first)");
  if (c->packed) return NIF_OK;
  c->packed_p32 = false;
  if (c->NSTB > 1) { const int rcp = ensure_packed_p32(c); if (rcp) return rcp; }
  ProfScope ps_(c, NIF_PROF_PACK);
  if (c->kind == NIF_KIND_LASTLAYER) {
    c->ll_packed32 = false;
    c->ll_mlp_packed = false;       // the f32 planes of the 32-point MLP kernels: packed when one of them runs (ensure_ll_mlp_planes)
    {
      SNetArgs probe; fill_snet_ll(c, probe, nullptr, 0, 0, 32);
      static const bool ll_old = [] { const char* e = getenv("NIF_LL_MLP"); return e && e[0] == '1'; }();
      c->use_ll4 = c->sWF4 && c->ll_slots && c->ll_wpf && c->ll_wpb && !ll_old && snet4_supported(probe);
    }
    if (c->ll_slots) {
      const int n = c->n, nh = c->nh, sop = c->so * c->r;
      LLSlotMap m; m.nseg = 0;
      auto seg = [&](long src, long dst, long len) { m.seg[m.nseg].src = src; m.seg[m.nseg].dst = dst; m.seg[m.nseg].len = len; ++m.nseg; };
      const long s_wl = (long)c->si * n + (long)nh * n * n, s_b1 = s_wl + (long)n * sop, s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
      seg(c->s_first_w, 0, (long)c->si * n);
      seg(c->s_bott_w, s_wl, (long)n * sop);
      seg(c->s_first_b, s_b1, n);
      for (int j = 0; j < nh; ++j) {
        long w_off, b_off;
        if (!c->cfg.s_resblock) { w_off = c->s_hid_w[j]; b_off = c->s_hid_b[j]; }
        else { const int i = j / 2; w_off = (j & 1) ? c->s_hid_w2[i] : c->s_hid_w[i]; b_off = (j & 1) ? c->s_hid_b2[i] : c->s_hid_b[i]; }
        seg(b_off, s_bh + (long)j * n, n);
        if (c->use_ll4 && c->sWF4x)       // (r5: split groups + half planes + their scales in one launch)
          launch_pack16b_dual(c->theta, dense_ref(w_off, n, n), 0, 1, snet3_nbl(n),
                              (char*)c->sWF4 + (size_t)j * snet4_fwd_elems(n, 0) * 2, (char*)c->sWB4 + (size_t)j * snet4_bwd_elems(n, 0) * 2, 0, 0,
                              (char*)c->sWF4x + (size_t)j * snet4_bwd_elems(n, 0) * 2, (char*)c->sWB4x + (size_t)j * snet4_bwd_elems(n, 0) * 2,
                              0, 0, c->cfg.s_omega0, c->sWscale + (size_t)j * 2, c->st);
        else if (c->use_ll4)
          launch_pack16b(c->theta, dense_ref(w_off, n, n), snet3_nbl(n),
                         (char*)c->sWF4 + (size_t)j * snet4_fwd_elems(n, 0) * 2, (char*)c->sWB4 + (size_t)j * snet4_bwd_elems(n, 0) * 2,
                         c->cfg.s_omega0, c->st);
        if (c->use_ll4 && c->sWF4h)
          launch_pack16b(c->theta, dense_ref(w_off, n, n), snet3_nbl(n),
                         (char*)c->sWF4h + (size_t)j * (snet4_fwd_elems(n, 0) / 3) * 2, (char*)c->sWB4h + (size_t)j * (snet4_bwd_elems(n, 0) / 2) * 2,
                         c->cfg.s_omega0, c->st, c->cfg.mixed_policy == NIF_POLICY_MIXED_F16 ? 2 : 1);
      }
      seg(c->s_bott_b, s_bl, sop);
      seg(c->ll_bias, s_bl + sop, c->so);
      seg(c->last_w, s_bl + sop + c->so, (long)c->r * c->r);
      launch_ll_slots(c->theta, m, c->ll_slots, c->st);
      if (c->use_ll4) launch_pack_phi(c->theta, c->s_bott_w, n, sop, c->ll_wpf, c->ll_wpb, c->st);
    }
    HIPCHK(hipGetLastError());
    c->packed = true;
    return NIF_OK;
  }
  SNetArgs probe; fill_snet(c, probe, nullptr, 0, 0, 32);
  c->use_snet3 = snet3_supported(probe);
  c->jac_ok = jac_supported(probe);       // (a superset: one plane buffer when two do not fit)
  // NIF_FP32_MFMA=1 in the environment keeps every product on the f32-input MFMAs (k_snet3) for A/B runs
  const bool fp32_only = c->opt_fp32_mfma;
  // (the bf16-split kernel streams its planes in chunks: it also takes the shapes whose whole fp32 planes do not fit the LDS --
  // 128 units with latent_dim >= 3 and many matrices -- which k_snet3 / k_jac / k_sob at that width cannot)
  c->use_snet4 = c->sWF4 && !fp32_only && snet4_supported(probe);
  c->packed32 = false;
  if (c->use_snet4 && c->nh > 0 && c->sWF4x)   // all hidden hyper-matrices (n^2 slots apart): split groups, half planes and their scales in ONE launch (r5)
    launch_pack16b_dual(c->theta, hyper_ref(c, (long)c->si * c->n, c->n, c->n, c->n), (long)c->n * c->n, c->nh, snet3_nbl(c->n),
                        c->sWF4, c->sWB4, snet4_fwd_elems(c->n, c->r), snet4_bwd_elems(c->n, c->r),
                        c->sWF4x, c->sWB4x, snet4_bwd_elems(c->n, c->r), snet4_bwd_elems(c->n, c->r), probe.omega, c->sWscale, c->st);
  else if (c->use_snet4 && c->nh > 0)
    launch_pack16b_batch(c->theta, hyper_ref(c, (long)c->si * c->n, c->n, c->n, c->n), (long)c->n * c->n, c->nh, snet3_nbl(c->n),
                         c->sWF4, c->sWB4, snet4_fwd_elems(c->n, c->r), snet4_bwd_elems(c->n, c->r), probe.omega, c->st);
  if (c->use_snet4 && c->nh > 0 && c->sWF4h)
    launch_pack16b_batch(c->theta, hyper_ref(c, (long)c->si * c->n, c->n, c->n, c->n), (long)c->n * c->n, c->nh, snet3_nbl(c->n),
                         c->sWF4h, c->sWB4h, snet4_fwd_elems(c->n, c->r) / 3, snet4_bwd_elems(c->n, c->r) / 2, probe.omega, c->st,
                         c->cfg.mixed_policy == NIF_POLICY_MIXED_F16 ? 2 : 1);
  HIPCHK(hipGetLastError());
  c->packed = true;
  if (!c->use_snet4) return ensure_packed32(c);
  return NIF_OK;
}

extern "C" int nif_forward_dev(nif_ctx* c, const float* xin, int64_t B, float* u) {
  if (!c || !xin || !u || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = ensure_packed(c); if (rc) return rc;
  rc = ensure_capacity(c, B, false); if (rc) return rc;
  PNetArgs pa; fill_pnet(c, pa, xin, B);
  { ProfScope p_(c, NIF_PROF_PNET_FWD); launch_pnet(pa, c->NSTB, false, c->st); }
  if (c->kind == NIF_KIND_LASTLAYER) {
    ProfScope p_(c, NIF_PROF_SNET_FWD);
    if (c->use_ll4) {
      SNetArgs sa; fill_snet_ll(c, sa, xin, c->pi + c->si, c->pi, B);
      sa.u_out = u;
      if (launch_snet4(sa, false, false, c->st) < 0) return fail(NIF_ERR_STATE, "internal: no k_snet4 form for this net (SIREN planes not packed as half pairs)");
      HIPCHK(hipGetLastError());
      return NIF_OK;
    }
    ensure_ll_mlp_planes(c); PNetArgs ma; fill_snet_mlp(c, ma, xin, c->pi + c->si, c->pi, B);
    launch_pnet(ma, c->NB, false, c->st);
    LLArgs la; fill_ll(c, la, B); la.u_out = u;
    launch_ll_out(la, false, c->st);
    HIPCHK(hipGetLastError());
    return NIF_OK;
  }
  SNetArgs sa; fill_snet(c, sa, xin, c->pi + c->si, c->pi, B);
  sa.u_out = u;
  {
    ProfScope p_(c, NIF_PROF_SNET_FWD);
    if (c->use_snet4) { if (launch_snet4(sa, false, false, c->st) < 0) return fail(NIF_ERR_STATE, "internal: no k_snet4 form for this net (SIREN planes not packed as half pairs)"); }
    else if (c->use_snet3) launch_snet3(sa, false, false, nullptr, c->st);
    else launch_snet(sa, c->NB, false, c->st);
  }
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

static int stage(nif_ctx* c, float** buf, long* cap, const float* host, long n) {
  int rc = grow(buf, cap, n); if (rc) return rc;
  if (host) HIPCHK(hipMemcpyAsync(*buf, host, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->st));
  return NIF_OK;
}

extern "C" int nif_forward(nif_ctx* c, const float* xin, int64_t B, float* u) {
  if (!c || !xin || !u || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = stage(c, &c->d_a, &c->cap_a, xin, B * (c->pi + c->si)); if (rc) return rc;
  rc = stage(c, &c->d_d, &c->cap_d, nullptr, B * c->so); if (rc) return rc;
  rc = nif_forward_dev(c, c->d_a, B, c->d_d); if (rc) return rc;
  HIPCHK(hipMemcpyAsync(u, c->d_d, sizeof(float) * (size_t)(B * c->so), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

extern "C" int nif_jacobian(nif_ctx* c, const float* xin, int64_t B, const int32_t* y_idx, int32_t ny,
                            const int32_t* x_idx, int32_t nx, float* y_out, float* dydx_out) {
  if (!c || !xin || !y_idx || !x_idx || !y_out || !dydx_out || B <= 0 || ny <= 0 || nx <= 0)
    return fail(NIF_ERR_INVALID, "bad argument");
  for (int i = 0; i < ny; ++i)
    if (y_idx[i] < 0 || y_idx[i] >= c->so) return fail(NIF_ERR_INVALID, "y_index out of range");
  for (int j = 0; j < nx; ++j)
    if (x_idx[j] < 0 || x_idx[j] >= c->pi + c->si) return fail(NIF_ERR_INVALID, "x_index out of range");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = ensure_packed(c); if (rc) return rc;
  rc = ensure_packed32(c); if (rc) return rc;
  rc = ensure_packed_p32(c); if (rc) return rc;
  const bool ll = c->kind == NIF_KIND_LASTLAYER;
  if (!ll && !c->jac_ok) return fail(NIF_ERR_INVALID, "JacobianLayer: one weight plane and the small hyper-vectors of this shape exceed the 160 KB LDS of a CU");
  rc = ensure_capacity(c, B, false); if (rc) return rc;
  const long ntiles = (B + 31) / 32;
  const int ncol = c->pi + c->si;
  rc = stage(c, &c->d_a, &c->cap_a, xin, B * ncol); if (rc) return rc;
  rc = stage(c, &c->d_d, &c->cap_d, nullptr, B * c->so); if (rc) return rc;
  rc = stage(c, &c->d_b, &c->cap_b, nullptr, B * c->so * nx); if (rc) return rc;
  // tangent buffers: dz/dp for up to 3 parameter seeds (and d phi / dx for the last-layer class)
  const long zd_sz = ntiles * 32 * (long)c->r * (ll ? c->so : 1);
  rc = grow(&c->d_c, &c->cap_c, 3 * zd_sz); if (rc) return rc;
  PNetArgs pa; fill_pnet(c, pa, c->d_a, B);
  if (ll) {
    // u = Dot(phi(x), a(p)) + bias: coordinate columns move phi, parameter columns move a
    rc = nif_forward_dev(c, c->d_a, B, c->d_d); if (rc) return rc;     // leaves Z (= a) on the device
    ensure_ll_mlp_planes(c); PNetArgs ma; fill_snet_mlp(c, ma, c->d_a, ncol, c->pi, B);
    if (c->use_ll4) launch_pnet(ma, c->NB, false, c->st);               // the fused forward keeps phi on chip: compute it here
    for (int j = 0; j < nx; ++j) {
      if (x_idx[j] >= c->pi) {
        PNetArgs mj = ma; mj.Z = c->DPHI;   // primal again into a scratch, tangent into d_c
        launch_mlp_jac(mj, c->NB, x_idx[j] - c->pi, c->d_c, c->st);
        launch_ll_jac_out(c->PHI, c->Z, c->d_c, nullptr, B, c->r, c->so, nx, j, c->d_b, c->st);
      } else {
        PNetArgs pj = pa; pj.Z = c->DA;
        launch_mlp_jac(pj, c->NSTB, x_idx[j], c->d_c, c->st);
        launch_ll_jac_out(c->PHI, c->Z, nullptr, c->d_c, B, c->r, c->so, nx, j, c->d_b, c->st);
      }
    }
  } else {
    launch_pnet(pa, c->NSTB, false, c->st);
    SNetArgs sa; fill_snet(c, sa, c->d_a, ncol, c->pi, B);
    for (int x0 = 0; x0 < nx; x0 += 3) {
      int seeds[3] = {0, 0, 0};
      const float* zd[3] = {nullptr, nullptr, nullptr};
      const int ns = nx - x0 < 3 ? nx - x0 : 3;
      for (int d = 0; d < ns; ++d) {
        const int col = x_idx[x0 + d];
        if (col >= c->pi) { seeds[d] = col - c->pi; }
        else {
          seeds[d] = -1;
          PNetArgs pj = pa; pj.Z = c->DZ;     // primal latent again into a scratch
          launch_mlp_jac(pj, c->NSTB, col, c->d_c + d * zd_sz, c->st);
          zd[d] = c->d_c + d * zd_sz;
        }
      }
      sa.u_out = x0 == 0 ? c->d_d : nullptr;
      launch_jac(sa, ns, seeds, zd, nx, x0, c->d_b, c->st);
    }
  }
  HIPCHK(hipGetLastError());
  std::vector<float> full((size_t)B * c->so * nx);
  HIPCHK(hipMemcpyAsync(y_out, c->d_d, sizeof(float) * (size_t)(B * c->so), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemcpyAsync(full.data(), c->d_b, sizeof(float) * full.size(), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  for (int64_t a_ = 0; a_ < B; ++a_)
    for (int i = 0; i < ny; ++i)
      for (int j = 0; j < nx; ++j) dydx_out[(a_ * ny + i) * nx + j] = full[((size_t)a_ * c->so + y_idx[i]) * nx + j];
  return NIF_OK;
}

// HessianLayer (gradient.py:130-180, :234-261): y [B, so], dy/dx [B, ny, nx] and d2y/dx2 [B, ny, nx, nx] for ANY input columns
// x_idx: one launch of the second-order tangent kernel per column pair (j <= k), then the gather of the rows y_idx -- and, for the
// last-layer class, the contraction with the ParameterNet output -- ON THE DEVICE (r3; r2 did both in host loops).  Everything is
// enqueued on the context's stream; the outputs are device pointers.
static int hessian_core(nif_ctx* c, const float* xin_dev, int64_t B, const int32_t* y_idx, int32_t ny, const int32_t* x_idx,
                        int32_t nx, float* y_dev, float* dydx_dev, float* d2_dev) {
  bool anyp = false;
  for (int j = 0; j < nx; ++j) anyp = anyp || x_idx[j] < c->pi;
  int rc = ensure_packed(c); if (rc) return rc;
  rc = ensure_capacity(c, B, false); if (rc) return rc;
  const int ncol = c->pi + c->si;
  const long ntl = (B + 31) / 32;
  HessIdx I; I.ny = ny;
  for (int i = 0; i < 16; ++i) I.y_idx[i] = i < ny ? y_idx[i] : 0;
  PNetArgs pa; fill_pnet(c, pa, xin_dev, B);
  launch_pnet(pa, c->NSTB, false, c->st);
  // parameter columns: z' = dz/dp of every parameter column once (k_pjac, forward mode), z'' per pair of them (k_pjac2)
  std::vector<int> xc, xp;                         // positions (in x_idx) of the coordinate / parameter columns
  for (int j = 0; j < nx; ++j) (x_idx[j] >= c->pi ? xc : xp).push_back(j);
  const int np_ = (int)xp.size();
  if (xc.size() > 16 || np_ > 16) return fail(NIF_ERR_INVALID, "HessianLayer: at most 16 coordinate and 16 parameter columns in x_index");
  const long blk = ntl * 32 * c->r;                // floats of one latent-layout vector
  if (anyp) {
    if (!pjac_supported(pa))
      return fail(NIF_ERR_INVALID, "HessianLayer on parameter columns: ParameterNets of up to 128 units");
    const long need_zt = (long)c->pi * blk, need_dd = (long)np_ * np_ * blk;
    if (need_zt > c->zt_par_cap || need_dd > c->dzt_par_cap) HIPCHK(hipStreamSynchronize(c->st));
    if (need_zt > c->zt_par_cap) { rc = grow(&c->zt_par, &c->zt_par_cap, need_zt); if (rc) return rc; }
    if (need_dd > c->dzt_par_cap) { rc = grow(&c->dzt_par, &c->dzt_par_cap, need_dd); if (rc) return rc; }
    launch_pjac_fwd(pa, c->zt_par, c->st);
  }
  auto zt_of = [&](int col) -> const float* { return c->zt_par + (long)col * blk; };
  auto zdd_of = [&](int j, int k) -> float* { return c->dzt_par + (long)(j * np_ + k) * blk; };   // pair of xp positions, j <= k
  if (c->kind == NIF_KIND_LASTLAYER) {
    // u_i = sum_c phi[i,c](x) a_c(p) + bias_i: the coordinates only move phi (second-order tangents of the shared SIREN ShapeNet,
    // the r = 0 case of the same kernel with so * latent_dim outputs), the parameters only move a = latent last_w + last_b
    const int rl = c->r, sop = c->so * c->r;
    const int nxc = xc.empty() ? 1 : (int)xc.size();
    rc = stage(c, &c->d_d, &c->cap_d, nullptr, B * sop); if (rc) return rc;
    rc = stage(c, &c->d_b, &c->cap_b, nullptr, B * sop * nxc); if (rc) return rc;
    rc = stage(c, &c->d_c, &c->cap_c, nullptr, B * sop * nxc * nxc); if (rc) return rc;
    SNetArgs sa; rc = fill_snet_ll_sob(c, sa, xin_dev, B, true); if (rc) return rc;
    if (xc.empty()) { sa.u_out = c->d_d; launch_hess(sa, 0, 0, 0, 0, 1, c->d_b, c->d_c, c->st); }     // phi alone
    for (size_t j = 0; j < xc.size(); ++j)
      for (size_t k = j; k < xc.size(); ++k) {
        sa.u_out = (j == 0 && k == 0) ? c->d_d : nullptr;
        launch_hess(sa, x_idx[xc[j]] - c->pi, x_idx[xc[k]] - c->pi, (int)j, (int)k, nxc, c->d_b, c->d_c, c->st);
      }
    // a'_j = z'_j last_w and a''_jk = z''_jk last_w: one k_pjac2 launch per parameter pair, ONE k_through_lw over all of them
    const int nvec = np_ + np_ * np_;
    float* ap = nullptr;
    if (nvec > 0) {
      const long need = (long)nvec * blk + 64;        // + the source offsets (as raw bytes behind the vectors)
      if (need > c->jac_mu_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->jac_mu, &c->jac_mu_cap, need); if (rc) return rc; }
      ap = c->jac_mu;
      for (int j = 0; j < np_; ++j)
        for (int k = j; k < np_; ++k) launch_pjac2(pa, x_idx[xp[j]], x_idx[xp[k]], zdd_of(j, k), c->st);
      // source of vector v: z'_j inside zt_par, z''_jk inside dzt_par -- addressed relative to zt_par (both are hipMalloc'ed floats)
      std::vector<long> off(nvec, 0);
      for (int j = 0; j < np_; ++j) off[j] = (long)(zt_of(x_idx[xp[j]]) - c->zt_par);
      for (int j = 0; j < np_; ++j)
        for (int k = 0; k < np_; ++k) off[np_ + j * np_ + k] = (long)(zdd_of(j < k ? j : k, j < k ? k : j) - c->zt_par);
      long* off_dev = reinterpret_cast<long*>(ap + (long)nvec * blk);
      HIPCHK(hipMemcpyAsync(off_dev, off.data(), sizeof(long) * nvec, hipMemcpyHostToDevice, c->st));
      HIPCHK(hipStreamSynchronize(c->st));            // (off is a host temporary)
      launch_through_lw(c->zt_par, off_dev, nvec, c->theta + c->last_w, rl, ntl * 32, ap, c->st);
    }
    HessLLArgs H;
    memset(&H, 0, sizeof(H));
    H.f0 = c->d_d; H.fj = c->d_b; H.fh = c->d_c; H.Za = c->Z; H.AP = ap; H.bias = c->theta + c->ll_bias;
    H.B = B; H.npts = ntl * 32; H.so = c->so; H.rl = rl; H.nxc = (int)xc.size(); H.np = np_; H.nx = nx; H.I = I;
    for (size_t j = 0; j < xc.size(); ++j) H.xc[j] = xc[j];
    for (int j = 0; j < np_; ++j) H.xp[j] = xp[j];
    H.y = y_dev; H.dydx = dydx_dev; H.d2 = d2_dev;
    launch_ll_hess(H, c->st);
    HIPCHK(hipGetLastError());
    return NIF_OK;
  }
  rc = ensure_packed32(c); if (rc) return rc;
  if (!c->jac_ok) return fail(NIF_ERR_INVALID, "HessianLayer: one weight plane and the small hyper-vectors of this shape exceed the 160 KB LDS of a CU");
  rc = stage(c, &c->d_b, &c->cap_b, nullptr, B * c->so * nx); if (rc) return rc;
  rc = stage(c, &c->d_c, &c->cap_c, nullptr, B * c->so * nx * nx); if (rc) return rc;
  SNetArgs sa; fill_snet(c, sa, xin_dev, ncol, c->pi, B);
  std::vector<int> ppos(nx, -1);                    // position among the parameter columns of x_idx
  for (int j = 0; j < np_; ++j) ppos[xp[j]] = j;
  for (int j = 0; j < nx; ++j)
    for (int k = j; k < nx; ++k) {
      sa.u_out = (j == 0 && k == 0) ? y_dev : nullptr;
      const bool pj = x_idx[j] < c->pi, pk = x_idx[k] < c->pi;
      float* zdd = (pj && pk) ? zdd_of(ppos[j], ppos[k]) : nullptr;
      if (pj && pk) launch_pjac2(pa, x_idx[j], x_idx[k], zdd, c->st);
      launch_hess(sa, pj ? -1 : x_idx[j] - c->pi, pk ? -1 : x_idx[k] - c->pi, j, k, nx, c->d_b, c->d_c, c->st,
                  pj ? zt_of(x_idx[j]) : nullptr, pk ? zt_of(x_idx[k]) : nullptr, zdd);
    }
  launch_hess_gather(c->d_b, c->d_c, B, c->so, nx, I, dydx_dev, d2_dev, c->st);
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

static int hessian_check(nif_ctx* c, const void* xin, int64_t B, const int32_t* y_idx, int32_t ny, const int32_t* x_idx, int32_t nx,
                         const void* y_out, const void* dydx_out, const void* d2_out) {
  if (!c || !xin || !y_idx || !x_idx || !y_out || !dydx_out || !d2_out || B <= 0 || ny <= 0 || nx <= 0)
    return fail(NIF_ERR_INVALID, "bad argument");
  if (ny > 16) return fail(NIF_ERR_INVALID, "y_index: at most 16 entries");
  for (int i = 0; i < ny; ++i)
    if (y_idx[i] < 0 || y_idx[i] >= c->so) return fail(NIF_ERR_INVALID, "y_index out of range");
  for (int j = 0; j < nx; ++j)
    if (x_idx[j] < 0 || x_idx[j] >= c->pi + c->si) return fail(NIF_ERR_INVALID, "x_index out of range (0 <= i < pi_dim + si_dim)");
  return NIF_OK;
}
extern "C" int nif_hessian_dev(nif_ctx* c, const float* xin_dev, int64_t B, const int32_t* y_idx, int32_t ny, const int32_t* x_idx,
                               int32_t nx, float* y_dev, float* dydx_dev, float* d2_dev) {
  int rc = hessian_check(c, xin_dev, B, y_idx, ny, x_idx, nx, y_dev, dydx_dev, d2_dev); if (rc) return rc;
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  return hessian_core(c, xin_dev, B, y_idx, ny, x_idx, nx, y_dev, dydx_dev, d2_dev);
}
extern "C" int nif_hessian(nif_ctx* c, const float* xin, int64_t B, const int32_t* y_idx, int32_t ny, const int32_t* x_idx,
                           int32_t nx, float* y_out, float* dydx_out, float* d2_out) {
  int rc = hessian_check(c, xin, B, y_idx, ny, x_idx, nx, y_out, dydx_out, d2_out); if (rc) return rc;
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  rc = stage(c, &c->d_a, &c->cap_a, xin, B * (c->pi + c->si)); if (rc) return rc;
  const size_t n_y = (size_t)B * c->so, n_d = (size_t)B * ny * nx, n_h = n_d * nx;
  float* out = nullptr;
  HIPCHK(hipMalloc(&out, sizeof(float) * (n_y + n_d + n_h)));
  rc = hessian_core(c, c->d_a, B, y_idx, ny, x_idx, nx, out, out + n_y, out + n_y + n_d);
  if (rc == NIF_OK) {
    hipError_t e = hipMemcpyAsync(y_out, out, sizeof(float) * n_y, hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipMemcpyAsync(dydx_out, out + n_y, sizeof(float) * n_d, hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipMemcpyAsync(d2_out, out + n_y + n_d, sizeof(float) * n_h, hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipStreamSynchronize(c->st);
    if (e != hipSuccess) rc = fail(NIF_ERR_HIP, hipGetErrorString(e));
  } else (void)hipStreamSynchronize(c->st);
  (void)hipFree(out);
  return rc;
}

extern "C" int nif_pnet_latent(nif_ctx* c, const float* p, int64_t B, float* lr) {
  if (!c || !p || !lr || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = ensure_packed(c); if (rc) return rc;
  rc = ensure_capacity(c, B, false); if (rc) return rc;
  rc = stage(c, &c->d_a, &c->cap_a, p, B * c->pi); if (rc) return rc;
  rc = stage(c, &c->d_d, &c->cap_d, nullptr, B * c->r); if (rc) return rc;
  PNetArgs pa; fill_pnet(c, pa, c->d_a, B);
  pa.ncol = c->pi;
  launch_pnet(pa, c->NSTB, false, c->st);
  launch_tiles_to_rows(c->Z, B, c->r, c->d_d, c->st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(lr, c->d_d, sizeof(float) * (size_t)(B * c->r), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

extern "C" int nif_x_to_phi(nif_ctx* c, const float* x, int64_t B, float* phi) {
  if (!c || !x || !phi || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  if (c->kind != NIF_KIND_LASTLAYER) return fail(NIF_ERR_INVALID, "model_x_to_phi exists only for the last-layer class");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = ensure_packed(c); if (rc) return rc;
  rc = ensure_capacity(c, B, false); if (rc) return rc;
  rc = stage(c, &c->d_a, &c->cap_a, x, B * c->si); if (rc) return rc;
  rc = stage(c, &c->d_d, &c->cap_d, nullptr, B * c->so * c->r); if (rc) return rc;
  ensure_ll_mlp_planes(c); PNetArgs ma; fill_snet_mlp(c, ma, c->d_a, c->si, 0, B);
  launch_pnet(ma, c->NB, false, c->st);
  launch_tiles_to_rows(c->PHI, B, c->so * c->r, c->d_d, c->st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(phi, c->d_d, sizeof(float) * (size_t)(B * c->so * c->r), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

extern "C" int nif_latent_to_w_dev(nif_ctx* c, const float* lr, int64_t B, float* w) {
  if (!c || !lr || !w || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  if (!c->have_params) return fail(NIF_ERR_STATE, "parameters not set");
  if (c->kind == NIF_KIND_LASTLAYER)
    return fail(NIF_ERR_INVALID, "In this class: NIFMultiScaleLastLayerParameterization, `w` is the same as `lr`");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  { ProfScope p_(c, NIF_PROF_LATENT_TO_W); launch_latent_to_w(c->theta, c->last_w, c->last_b, c->r, c->po, lr, B, w, c->st); }
  HIPCHK(hipGetLastError());
  return NIF_OK;
}
extern "C" int nif_latent_to_w(nif_ctx* c, const float* lr, int64_t B, float* w) {
  if (!c || !lr || !w || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = stage(c, &c->d_a, &c->cap_a, lr, B * c->r); if (rc) return rc;
  rc = stage(c, &c->d_b, &c->cap_b, nullptr, B * c->po); if (rc) return rc;
  rc = nif_latent_to_w_dev(c, c->d_a, B, c->d_b); if (rc) return rc;
  HIPCHK(hipMemcpyAsync(w, c->d_b, sizeof(float) * (size_t)(B * c->po), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

extern "C" int nif_shapenet_given_w_dev(nif_ctx* c, const float* x, const float* w, int64_t B, float* u) {
  if (!c || !x || !w || !u || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  if (c->kind == NIF_KIND_LASTLAYER) {   // u = Dot(phi(x), w) + bias with caller-supplied w [B, r]
    int rc = ensure_packed(c); if (rc) return rc;
    rc = ensure_capacity(c, B, false); if (rc) return rc;
    ensure_ll_mlp_planes(c); PNetArgs ma; fill_snet_mlp(c, ma, x, c->si, 0, B);
    launch_pnet(ma, c->NB, false, c->st);
    launch_rows_to_tiles(w, B, c->r, c->Z, c->st);
    LLArgs la; fill_ll(c, la, B); la.u_out = u;
    launch_ll_out(la, false, c->st);
    HIPCHK(hipGetLastError());
    return NIF_OK;
  }
  const int act = c->kind == NIF_KIND_NIF ? c->cfg.s_act : NIF_ACT_SINE;
  const float om = c->kind == NIF_KIND_NIF ? 1.0f : c->cfg.s_omega0;
  { ProfScope p_(c, NIF_PROF_GIVEN_W);
    launch_given_w(x, w, u, B, c->si, c->so, c->n, c->nh, c->po, act, c->cfg.s_resblock, c->kind == NIF_KIND_NIF, om, c->st); }
  HIPCHK(hipGetLastError());
  return NIF_OK;
}
extern "C" int nif_shapenet_given_w(nif_ctx* c, const float* x, const float* w, int64_t B, float* u) {
  if (!c || !x || !w || !u || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = stage(c, &c->d_a, &c->cap_a, x, B * c->si); if (rc) return rc;
  rc = stage(c, &c->d_b, &c->cap_b, w, B * c->po); if (rc) return rc;   // po == r for the last-layer class
  rc = stage(c, &c->d_d, &c->cap_d, nullptr, B * c->so); if (rc) return rc;
  rc = nif_shapenet_given_w_dev(c, c->d_a, c->d_b, B, c->d_d); if (rc) return rc;
  HIPCHK(hipMemcpyAsync(u, c->d_d, sizeof(float) * (size_t)(B * c->so), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

// ---- last-layer-parameterised class: loss and gradient ---------------------------------------------
// SNetArgs of the last-layer class for k_sob (Sobolev step / its predict): k_snet4's arguments for that class plus, beyond the
// widths whose bf16 planes fit the LDS (n > 96), the f32-input MFMA planes of the shared hidden matrices, packed on demand
static int fill_snet_ll_sob(nif_ctx* c, SNetArgs& sa, const float* xin, long B, bool f32_planes) {
  if (!c->ll_slots) return fail(NIF_ERR_INVALID, "JacobianLayer-as-output / HessianLayer on the last-layer class: ShapeNet with at least one hidden layer and units <= 128");
  fill_snet_ll(c, sa, xin, c->pi + c->si, c->pi, B);
  if (!sob_ll_supported(sa)) return fail(NIF_ERR_INVALID, "JacobianLayer-as-output / HessianLayer on the last-layer class: so * latent_dim <= 64");
  const int NBL = snet3_nbl(c->n);
  if (!c->use_ll4) { sa.WF4 = nullptr; sa.WB4 = nullptr; f32_planes = true; }   // (bf16-split planes are packed for k_snet4<LL> only)
  if (NBL > 6 || f32_planes) {     // (k_jac / the Hessian take the f32-input planes at every width)
    const long plane_s = snet3_plane_floats(c->n) / 4;
    if (!c->ll_packed32) {
      for (int j = 0; j < c->nh; ++j) {
        long w_off;
        if (!c->cfg.s_resblock) w_off = c->s_hid_w[j];
        else { const int i = j / 2; w_off = (j & 1) ? c->s_hid_w2[i] : c->s_hid_w[i]; }
        launch_pack16(c->theta, dense_ref(w_off, c->n, c->n), NBL, c->sWF + (long)j * plane_s, c->sWB + (long)j * plane_s, c->st);
      }
      c->ll_packed32 = true;
    }
    sa.WF = c->sWF; sa.WB = c->sWB;
    if (NBL > 6) { sa.WF4 = nullptr; sa.WB4 = nullptr; }
  }
  return NIF_OK;
}

static int loss_grad_ll(nif_ctx* c, const float* xin, const float* y, const float* sw, long B, long Bg, int ns = 0,
                        const SobPlan* sp = nullptr, const float* gt = nullptr, float wj = 0.f) {
  const long ntiles = (B + 31) / 32;
  const int ncol = c->pi + c->si;
  const int nsc = ns > 0 ? sp->nsc : 0, nhead = ns > 0 ? sp->ns - sp->nsc : 0;   // coordinate streams / parameter-column heads
  PNetArgs pa; fill_pnet(c, pa, xin, B);
  ensure_ll_mlp_planes(c); PNetArgs ma; fill_snet_mlp(c, ma, xin, ncol, c->pi, B);
  LLArgs la; fill_ll(c, la, B);
  la.y = y; la.sw = sw; la.inv_bg = 1.0f / (float)Bg;
  static const bool force_stash_ll = [] { const char* e = getenv("NIF_PNET_STASH"); return e && e[0] == '1'; }();
  const bool fused_p = !force_stash_ll && pnet_bwg_supported(pa);
  if (!fused_p) { const int rcp = ensure_packed_p32(c); if (rcp) return rcp; }
  { ProfScope p_(c, NIF_PROF_PNET_FWD); launch_pnet(pa, c->NSTB, !fused_p, c->st); }
  int nloss = (int)((ntiles * 32 + 255) / 256);
  bool ll_dab = false, ll_ph = false;
  if (ns > 0) {   // Sobolev: primal + tangents + their adjoint on k_sob<.., LL>; stashes and DPHI hold (1 + ns) blocks of tiles
    SNetArgs sa; int rc = fill_snet_ll_sob(c, sa, xin, B); if (rc) return rc;
    sa.y = y; sa.sw = sw; sa.loss_partial = c->loss_partial; sa.inv_bg = 1.0f / (float)Bg;
    nloss = launch_sob(sa, true, nsc, sp->seeds, nullptr, 0.f, nullptr, nullptr, true, c->st);
    if (nloss < 0) return fail(NIF_ERR_INVALID, "Sobolev step: the kernel's working set of this shape does not fit the 160 KB LDS of a CU");
    const long need = (long)nloss * 4 * sob_ring_floats_per_wave(c->n, c->nh);
    if (need > c->dring_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->dring, &c->dring_cap, need); if (rc) return rc; }
    SobPar spar{};
    for (int d = 0; d < 3; ++d) { spar.par[d] = -1; spar.gcol[d] = sp->gcol[d]; }
    spar.gstride = sp->gstride; spar.nx_all = sp->nx_all; spar.ny = sp->ny; spar.no_primal = sp->no_primal; spar.ymask = sp->ymask;
    if (nhead > 0) {     // parameter columns: heads of the epilogue (z' = dz/dp sits in c->zt_par, loss_grad_core)
      const long need_a = 3 * ntiles * 32 * c->r, need_l = 3 * ntiles * 32 * 32 * c->RB;
      if (need_a > c->dat_par_cap || need_l > c->ztl_par_cap) HIPCHK(hipStreamSynchronize(c->st));
      if (need_a > c->dat_par_cap) { rc = grow(&c->dat_par, &c->dat_par_cap, need_a); if (rc) return rc; }
      if (need_l > c->ztl_par_cap) {
        rc = grow(&c->ztl_par, &c->ztl_par_cap, need_l); if (rc) return rc;
        HIPCHK(hipMemsetAsync(c->ztl_par, 0, sizeof(float) * (size_t)need_l, c->st));     // the padding rows stay zero
      }
      spar.npar = nhead; spar.ZT = c->zt_par; spar.DZT = c->dzt_par; spar.DAT = c->dat_par; spar.ZTL = c->ztl_par;
      spar.zl_rows = 32 * c->RB;
      for (int e = 0; e < nhead; ++e) { spar.parc[e] = sp->par[nsc + e]; spar.pcol[e] = sp->gcol[nsc + e]; }
    }
    ProfScope p_(c, NIF_PROF_SNET);
    launch_sob(sa, true, nsc, sp->seeds, gt, wj, c->dring, nullptr, false, c->st, &spar);
  } else if (c->use_ll4) {
    SNetArgs sa; fill_snet_ll(c, sa, xin, ncol, c->pi, B);
    sa.y = y; sa.sw = sw; sa.loss_partial = c->loss_partial; sa.inv_bg = 1.0f / (float)Bg;
    {   // mixed_bfloat16: bf16 dL/da stash rows of the shared hidden layers (step_chunk has the hypernetwork classes' rule)
      const int nbl = snet3_nbl(c->n);
      ll_dab = sa.prec == 1 && (nbl == 2 || nbl == 4 || nbl == 8) && gw_da_bf16_ok(c->NB, c->NB, 0);
      sa.da_bf16 = ll_dab ? 1 : 0;
      if (snet4_writes_da_bf16(sa) != ll_dab) return fail(NIF_ERR_STATE, "internal: dL/da stash format of producer and plan disagree");
      // ... and, on the 128-wide kernel, the hidden matrices' input rows as 16-bit phases (k_gw8<0, true, true> reads them)
      sa.h_ph16 = (ll_dab && nbl == 8 && !c->cfg.s_resblock && gw_in_ph16_ok(c->NB, c->NB, 0)) ? 1 : 0;
      ll_ph = snet4_writes_h_ph16(sa);
      if (ll_ph != (sa.h_ph16 != 0)) return fail(NIF_ERR_STATE, "internal: layer-input stash format of producer and plan disagree");
    }
    nloss = launch_snet4(sa, true, true, c->st);
    const long need = (long)nloss * 4 * snet3_ring_floats_per_wave(c->n, c->nh);
    if (need > c->dring_cap) {
      HIPCHK(hipStreamSynchronize(c->st));
      int rc = grow(&c->dring, &c->dring_cap, need); if (rc) return rc;
    }
    sa.dring = c->dring;
    ProfScope p_(c, NIF_PROF_SNET);
    if (launch_snet4(sa, true, false, c->st) < 0) return fail(NIF_ERR_STATE, "internal: no k_snet4 form for this net (SIREN planes not packed as half pairs)");
  } else {
    ProfScope p_(c, NIF_PROF_SNET);
    launch_pnet(ma, c->NB, true, c->st);
    launch_ll_out(la, true, c->st);
    launch_pnet_bwd(ma, c->NB, c->st);
  }
  int n_act = 0;
  if (act_on(c)) {    // + c/Bg phi'(a) into dL/da and dL/dlatent, before their consumers; the loss term joins after the row reduction
    const long nlp = (B + 255) / 256;
    if (nlp > c->act_loss_cap) { HIPCHK(hipStreamSynchronize(c->st)); int rc = grow(&c->act_loss, &c->act_loss_cap, nlp); if (rc) return rc; }
    const bool l1 = c->act_l2 == 0.f;
    n_act = launch_ll_actreg(c->Z, c->theta + c->last_w, c->r, B, (l1 ? c->act_l1 : c->act_l2) / (float)Bg, l1, c->DA, c->DZL, c->act_loss, c->st);
  }
  int rows = (int)((ntiles + 3) / 4);
  if (rows > c->rows_cap) rows = c->rows_cap;
  if (rows < 1) rows = 1;
  { ProfScope p_(c, NIF_PROF_PNET_BWD);
    if (fused_p) launch_pnet_bwg(pa, c->partial, c->pstride, rows, c->st);
    else launch_pnet_bwd(pa, c->NSTB, c->st); }
  {
    ProfScope p_(c, NIF_PROF_GW);
    GwArgs g;
    auto base = [&](GwArgs& q) {
      memset(&q, 0, sizeof(q));
      q.ntiles = ntiles; q.B = B; q.partial = c->partial; q.pstride = c->pstride; q.has_bias = 1; q.scale = 1.0f; q.r = 0;
    };
    const int nms = c->nh;  // hidden matrices of the ShapeNet (2L with resblocks)
    float* sST = c->stash_s;
    auto sbase = [&](GwArgs& q) {   // Sobolev: the ShapeNet reductions also run over the tangent pseudo-tiles
      base(q);
      if (nsc > 0) {
        q.ntiles = ntiles * (1 + nsc); q.zt_mod = ntiles; q.bias_ntiles = ntiles;
        for (int d = 0; d < 3; ++d) q.seed[d] = d < nsc ? sp->seeds[d] : 0;
      }
    };
    // ShapeNet (dense SIREN): first, hidden matrices, bottleneck (n -> r*so), last_layer_bias
    sbase(g); g.DA = sST + (long)(nms + 1) * c->slot_s; g.xin = xin; g.ncol = ncol; g.col0 = c->pi; g.nd = c->si; g.scale = ma.omega;
    g.W = dense_ref(c->s_first_w, c->si, c->n); g.Bv = vec_ref(c->s_first_b, c->n);
    launch_gw_first(g, c->NB, rows, c->st);
    for (int mi = 0; mi < nms; ++mi) {
      sbase(g); g.IN = sST + (long)mi * c->slot_s; g.DA = sST + (long)(nms + 2 + mi) * c->slot_s; g.scale = ma.omega;
      long w_off, b_off;
      if (!c->cfg.s_resblock) { w_off = c->s_hid_w[mi]; b_off = c->s_hid_b[mi]; }
      else { const int i = mi / 2; w_off = (mi & 1) ? c->s_hid_w2[i] : c->s_hid_w[i]; b_off = (mi & 1) ? c->s_hid_b2[i] : c->s_hid_b[i]; }
      g.W = dense_ref(w_off, c->n, c->n); g.Bv = vec_ref(b_off, c->n);
      g.da_bf16 = ll_dab ? 1 : 0;
      g.in_ph16 = ll_ph ? 1 : 0;
      if (launch_gw_mfma(g, c->NB, c->NB, rows, c->st) < 0) return fail(NIF_ERR_STATE, "internal: bf16 dL/da stash rows without a reader of that form");
    }
    sbase(g); g.IN = sST + (long)nms * c->slot_s; g.SM = c->DPHI; g.nc = c->r * c->so;
    g.W = dense_ref(c->s_bott_w, c->n, c->r * c->so); g.Bv = vec_ref(c->s_bott_b, c->r * c->so);
    launch_gw_out(g, c->NB, rows, c->st);
    base(g); g.IN = sST + (long)nms * c->slot_s; g.SM = c->DU; g.nc = c->so;   // only the bias part: W.nin = 0
    g.W = dense_ref(0, 0, c->so); g.Bv = vec_ref(c->ll_bias, c->so);
    launch_gw_out(g, c->NB, rows, c->st);
    // ParameterNet: first, hidden matrices, bottleneck, last (r x r)
    float* pST = c->stash_p;
    if (!fused_p) {
    base(g); g.DA = pST + (long)(c->nm + 1) * c->slot_p; g.xin = xin; g.ncol = ncol; g.col0 = 0; g.nd = c->pi; g.scale = pa.omega;
    g.W = dense_ref(c->first_w, c->pi, c->nst); g.Bv = vec_ref(c->first_b, c->nst);
    launch_gw_first(g, c->NSTB, rows, c->st);
    for (int mi = 0; mi < c->nm; ++mi) {
      base(g); g.IN = pST + (long)mi * c->slot_p; g.DA = pST + (long)(c->nm + 2 + mi) * c->slot_p; g.scale = pa.omega;
      long w_off, b_off;
      if (!c->cfg.p_resblock) { w_off = c->hid_w[mi]; b_off = c->hid_b[mi]; }
      else { const int i = mi / 2; w_off = (mi & 1) ? c->hid_w2[i] : c->hid_w[i]; b_off = (mi & 1) ? c->hid_b2[i] : c->hid_b[i]; }
      g.W = dense_ref(w_off, c->nst, c->nst); g.Bv = vec_ref(b_off, c->nst);
      launch_gw_mfma(g, c->NSTB, c->NSTB, rows, c->st);
    }
    base(g); g.IN = pST + (long)c->nm * c->slot_p; g.SM = c->DZL; g.nc = c->r;
    g.W = dense_ref(c->bott_w, c->nst, c->r); g.Bv = vec_ref(c->bott_b, c->r);
    launch_gw_out(g, c->NSTB, rows, c->st);
    }
    base(g); g.IN = c->ZL; g.SM = c->DA; g.nc = c->r;   // latent (padded rows) x dL/da
    g.W = dense_ref(c->last_w, c->r, c->r); g.Bv = vec_ref(c->last_b, c->r);
    launch_gw_out(g, c->RB, rows, c->st);
  }
  {
    ProfScope pr_(c, NIF_PROF_REDUCE);
    launch_reduce(c->partial, c->pstride, rows, c->loss_partial, nloss, c->grad, c->P, c->st);
  }
  if (n_act > 0) launch_add_sum(c->act_loss, n_act, c->grad + c->P, c->st);
  if (c->jac_l1 != 0.f) { int rc = jac_reg_pass(c, xin, B, Bg); if (rc) return rc; }
  if (nhead > 0) {
    // the heads' share of the r x r layer: dL/dlast_w += z'^T dL/da' (a' = z' last_w has no bias), the same reduction as the
    // main one with (z', dL/da') as the operand pair; then the (primal, tangent) ParameterNet for dL/dz' (jac_reg_pass, given mu)
    if (!c->jac_tmp) HIPCHK(hipMalloc(&c->jac_tmp, sizeof(float) * (size_t)(c->P + 2)));
    int mu_blk[16];
  for (int q = 0; q < 16; ++q) mu_blk[q] = -1;
    const long rr = (long)c->r * c->r;
    for (int e = 0; e < nhead; ++e) {
      mu_blk[sp->par[nsc + e]] = e;
      GwArgs g; memset(&g, 0, sizeof(g));
      g.ntiles = ntiles; g.B = B; g.partial = c->partial; g.pstride = c->pstride; g.has_bias = 1; g.scale = 1.0f; g.r = 0;
      g.IN = c->ztl_par + (long)e * ntiles * 32 * 32 * c->RB; g.SM = c->dat_par + (long)e * ntiles * 32 * c->r; g.nc = c->r;
      g.W = dense_ref(c->last_w, c->r, c->r); g.Bv = vec_ref(c->last_b, c->r);     // (the bias columns are not taken over)
      launch_gw_out(g, c->RB, rows, c->st);
      launch_reduce(c->partial + c->last_w, c->pstride, rows, nullptr, 0, c->jac_tmp, rr, c->st);
      launch_axpy_cols(c->grad + c->last_w, c->jac_tmp, rr, c->P - c->last_w, c->st);
    }
    HIPCHK(hipGetLastError());
    return jac_reg_pass(c, xin, B, Bg, mu_blk);
  }
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

// Workspace sizing of the fused ShapeNet kernel the step will launch (no launch): number of workgroups (= loss partials),
// the act'(a) ring, the optional edge-gradient partials.  Grows buffers when needed (stream sync + hipMalloc): call
// nif_reserve() once up front to keep that out of the timed steps.
static int snet_plan(nif_ctx* c, SNetArgs& sa, int ns, const int* seeds, int* nloss, const int* par_of = nullptr) {
  int rc;
  if (ns > 0) {
    SobPar spq{};     // (the parameter streams' extra per-wave LDS counts)
    for (int d = 0; d < 3; ++d) spq.par[d] = par_of ? par_of[d] : -1;
    const int nblk = launch_sob(sa, true, ns, seeds, nullptr, 0.f, nullptr, nullptr, true, c->st, &spq);
    if (nblk < 0)
      return fail(NIF_ERR_INVALID, "Sobolev step: the kernel's working set of this shape (units, latent_dim, parameter columns) does not fit the 160 KB LDS of a CU");
    const long need = (long)nblk * 4 * sob_ring_floats_per_wave(c->n, c->nh);
    if (need > c->dring_cap) {
      HIPCHK(hipStreamSynchronize(c->st));
      rc = grow(&c->dring, &c->dring_cap, need); if (rc) return rc;
    }
    *nloss = nblk;
  } else if (c->use_snet3 || c->use_snet4) {
    int waves = 4;
    const int nblk = c->use_snet4 ? launch_snet4(sa, true, true, c->st) : launch_snet3(sa, true, true, &waves, c->st);
    const long need = (long)nblk * waves * snet3_ring_floats_per_wave(c->n, c->nh);
    if (need > c->dring_cap) {
      HIPCHK(hipStreamSynchronize(c->st));
      rc = grow(&c->dring, &c->dring_cap, need); if (rc) return rc;
    }
    *nloss = nblk;
  }
  sa.dring = c->dring;
  return NIF_OK;
}

static int rows_for(const nif_ctx* c, long ntiles) {
  int rows = (int)((ntiles + 3) / 4);
  if (rows > c->rows_cap) rows = c->rows_cap;
  return rows < 1 ? 1 : rows;
}

// chunk size (points) of the two-stream pipeline, 0 = off.  nif_set_option("pipe_chunk", points) / NIF_PIPE_CHUNK;
// default: on for batches of at least four chunks of 2^17 points
static long pipe_chunk_points(const nif_ctx* c, long B) {
  long ch = c->opt_pipe_chunk;
  if (ch < 0) ch = 0;     // measured slower than the single-stream step: off unless asked for
  if (ch <= 0) return 0;
  ch = (ch + 2047) / 2048 * 2048;          // whole groups of 4 x 16-point tiles x 32 (and 32-point stash tiles)
  return ch;
}

static int ensure_pipe(nif_ctx* c, int nchunk) {
  if (!c->st2) {
    HIPCHK(hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  }
  while ((int)c->ev_chunk.size() < nchunk) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_chunk.push_back(e);
  }
  if (nchunk > c->chunk_cap) {
    HIPCHK(hipStreamSynchronize(c->st)); HIPCHK(hipStreamSynchronize(c->st2));
    if (c->chunk_grad) HIPCHK(hipFree(c->chunk_grad));
    c->chunk_grad = nullptr; c->chunk_cap = 0;
    HIPCHK(hipMalloc(&c->chunk_grad, sizeof(float) * (size_t)nchunk * c->pstride));
    c->chunk_cap = nchunk;
  }
  return NIF_OK;
}

// Forward + adjoint + weight-gradient partial rows of the points [off, off + Bc) of a batch (NIF / NIFMultiScale).
// The ParameterNet forward and the fused ShapeNet kernel go to stream sa_st, everything that consumes their output (the
// ParameterNet adjoint, the weight-gradient reductions) to sb_st, ordered behind them by an event when the streams differ.
// ns = 0: loss = mse(u, y).  ns > 0 (Sobolev, whole batch only): + wj * mse(du/dx_seed, gt), k_sob instead of k_snet3/4;
// the ShapeNet stashes then hold (1+ns) blocks of tiles (real, then one block of tangent pseudo-tiles per seed).
static int step_chunk(nif_ctx* c, const float* xin0, const float* y0, const float* sw0, long off, long B, long Bg, int ns,
                      const int* seeds, const float* gt, float wj, hipStream_t sa_st, hipStream_t sb_st, hipStream_t pb_st,
                      float* partial, float* loss_partial, int* nloss_out, int chunk_idx, bool whole, SNetArgs* sa_out = nullptr,
                      const SobPlan* sp = nullptr) {
  int rc;
  const int ncol = c->pi + c->si;
  const long ntiles = (B + 31) / 32, t0 = off / 32;       // off is a multiple of 32 points
  const float* xin = xin0 + off * ncol;
  const float* y = y0 + off * c->so;
  const float* sw = sw0 ? sw0 + off : nullptr;
  // forward + adjoint
  PNetArgs pa; fill_pnet(c, pa, xin, B);
  pa.Z = c->Z + t0 * 32 * c->r; pa.DZ = c->DZ + t0 * 32 * c->r;
  if (pa.stash) pa.stash += t0 * 32 * 32 * c->NSTB;
  // small ParameterNets: no stash -- the adjoint kernel recomputes the forward pass and reduces the weight
  // gradients itself (k_pnetbw.hip).  NIF_PNET_STASH=1 forces the stash path (A/B runs, tests)
  static const bool force_stash = [] { const char* e = getenv("NIF_PNET_STASH"); return e && e[0] == '1'; }();
  const bool fused_p = !force_stash && pnet_bwg_supported(pa);
  if (!fused_p) { const int rcp = ensure_packed_p32(c); if (rcp) return rcp; }
  { ProfScope p_(c, NIF_PROF_PNET_FWD, sa_st); launch_pnet(pa, c->NSTB, !fused_p, sa_st); }
  SNetArgs sa; fill_snet(c, sa, xin, ncol, c->pi, B);
  sa.Z = pa.Z; sa.DZ = pa.DZ; sa.DU = c->DU + t0 * 32 * c->so;
  sa.stash = c->stash_s + t0 * 32 * 32 * c->NB;
  sa.y = y; sa.sw = sw; sa.u_out = nullptr; sa.loss_partial = loss_partial; sa.inv_bg = 1.0f / (float)Bg;
  if (!whole) sa.wg_cap = c->opt_pipe_wgs;        // leave room on every CU for the reductions of the previous chunk
  int nloss = (int)((ntiles + 3) / 4);
  rc = snet_plan(c, sa, ns, seeds, &nloss, sp ? sp->par : nullptr); if (rc) return rc;
  // plain SIREN step on the bf16-split kernel: every ShapeNet weight gradient inside the training kernel (k_snet6) -- its workgroups
  // are the partial-gradient rows, so the row count of this step has to be the kernel's grid
  const bool fused_gw = ns == 0 && whole && c->use_snet4 && c->opt_fuse_gw && snet6_supported(sa) &&
                        snet6_rows(sa) == rows_for(c, ntiles);
  if (fused_gw) nloss = snet6_rows(sa);
  *nloss_out = nloss;
  {   // mixed_bfloat16: the hidden layers' dL/da stash rows in bf16 when both the producer of this step (k_snet4<PR> / k_sobw<PR>)
      // and the consumer (k_gw_lds) have the form -- half the bytes of that operand on either side
    const int nbl = snet3_nbl(c->n);
    bool dab = sa.prec == 1 && (nbl == 2 || nbl == 4 || nbl == 8) && gw_da_bf16_ok(c->NB, c->NB, c->r);
    if (ns > 0) dab = dab && sobw_supported(sa, ns, sp && sp->any_par);
    else dab = dab && c->use_snet4 && !fused_gw;
    sa.da_bf16 = dab ? 1 : 0;
    // ... and, on the 128-wide kernel, the hidden matrices' input rows as 16-bit phases (k_gw8<R, true, true> reads them)
    sa.h_ph16 = (dab && ns == 0 && nbl == 8 && !c->cfg.s_resblock && !sa.nif_skip && gw_in_ph16_ok(c->NB, c->NB, c->r)) ? 1 : 0;
  }
  // what the producer of this step really writes (its own predicate, next to its kernels); the readers below follow THAT
  const bool wrote_da_bf16 = ns > 0 ? sob_writes_da_bf16(sa, ns, sp && sp->any_par)
                                    : (!fused_gw && c->use_snet4 && snet4_writes_da_bf16(sa));
  if (wrote_da_bf16 != (sa.da_bf16 != 0)) return fail(NIF_ERR_STATE, "internal: dL/da stash format of producer and plan disagree");
  const bool wrote_h_ph16 = ns == 0 && !fused_gw && c->use_snet4 && snet4_writes_h_ph16(sa);
  if (wrote_h_ph16 != (sa.h_ph16 != 0)) return fail(NIF_ERR_STATE, "internal: layer-input stash format of producer and plan disagree");
  {
    ProfScope p_(c, NIF_PROF_SNET, sa_st);
    if (ns > 0) {
      SobPar spar{}; const SobPar* sparp = nullptr;
      if (sp) { spar.gstride = sp->gstride; spar.nx_all = sp->nx_all; spar.ny = sp->ny; spar.no_primal = sp->no_primal; spar.ymask = sp->ymask; }
      if (sp && sp->any_par) {
        for (int d = 0; d < 3; ++d) { spar.par[d] = sp->par[d]; spar.gcol[d] = sp->gcol[d]; }
        spar.ZT = c->zt_par; spar.DZT = c->dzt_par; sparp = &spar;
      } else if (sp) {
        for (int d = 0; d < 3; ++d) { spar.par[d] = -1; spar.gcol[d] = sp->gcol[d]; }
        spar.ZT = nullptr; spar.DZT = nullptr; sparp = &spar;
      }
      launch_sob(sa, true, ns, seeds, gt, wj, c->dring, nullptr, false, sa_st, sparp);
    }
    else if (fused_gw) launch_snet6(sa, partial, c->pstride, sa_st);
    else if (c->use_snet4) { if (launch_snet4(sa, true, false, sa_st) < 0) return fail(NIF_ERR_STATE, "internal: no k_snet4 form for this net (SIREN planes not packed as half pairs)"); }
    else if (c->use_snet3) launch_snet3(sa, true, false, nullptr, sa_st);
    else launch_snet(sa, c->NB, true, sa_st);
  }
  if (act_on(c)) {   // + c/Bg sum phi'(out) M^(k) into dL/dz, before the ParameterNet adjoint consumes it; its loss partials
    const long nlp = (B + 255) / 256;
    if (nlp > c->act_loss_cap) { HIPCHK(hipStreamSynchronize(sa_st)); rc = grow(&c->act_loss, &c->act_loss_cap, nlp); if (rc) return rc; }
    const bool l1 = c->act_l2 == 0.f;
    launch_actreg_points(l1, c->theta, c->last_w, c->last_b, c->r, c->po, sa.Z, B, (l1 ? c->act_l1 : c->act_l2) / (float)Bg, sa.DZ,
                         c->act_loss, sa_st);
  }
  if (sb_st != sa_st || pb_st != sa_st) {
    HIPCHK(hipEventRecord(c->ev_chunk[chunk_idx], sa_st));
    if (sb_st != sa_st) HIPCHK(hipStreamWaitEvent(sb_st, c->ev_chunk[chunk_idx], 0));
    if (pb_st != sa_st && pb_st != sb_st) HIPCHK(hipStreamWaitEvent(pb_st, c->ev_chunk[chunk_idx], 0));
  }
  // weight gradients -> partial rows
  const int rows = rows_for(c, ntiles);
  { ProfScope p_(c, NIF_PROF_PNET_BWD, pb_st);
    // the compute-bound adjoint also pulls the first gradient kernel's stash slot (dL/da of the first layer) through
    // the cache hierarchy -- see PbwArgs::touch.  NIF_PBW_TOUCH=0 turns it off (A/B)
    static const bool touch_on = [] { const char* e = getenv("NIF_PBW_TOUCH"); return !(e && e[0] == '0'); }();
    const float* touch = (touch_on && whole && pb_st == sb_st && !fused_gw) ? sa.stash + (long)(c->nh + 1) * c->slot_s : nullptr;
    if (fused_p) launch_pnet_bwg(pa, partial, c->pstride, rows, pb_st, touch, (long)c->NB * 1024);
    else launch_pnet_bwd(pa, c->NSTB, pb_st); }
  ProfScope pgw(c, NIF_PROF_GW, sb_st);
  GwArgs g;
  auto base = [&](GwArgs& q) {
    memset(&q, 0, sizeof(q));
    q.ntiles = ntiles; q.B = B; q.partial = partial; q.pstride = c->pstride; q.has_bias = 1; q.scale = 1.0f;
  };
  auto sbase = [&](GwArgs& q) {   // ShapeNet reductions also run over the tangent pseudo-tiles
    base(q);
    q.ntiles = ntiles * (1 + ns); q.zt_mod = ntiles; q.bias_ntiles = ntiles;
    for (int d = 0; d < 3; ++d) q.seed[d] = (seeds && d < ns) ? seeds[d] : 0;
  };
  const float om_s = sa.omega, om_p = pa.omega;
  float* sIN = sa.stash; float* sDA = sa.stash + (long)(c->nh + 1) * c->slot_s;
  if (!fused_gw) {
  // ShapeNet first layer
  sbase(g); g.DA = sDA; g.xin = xin; g.ncol = ncol; g.col0 = c->pi; g.nd = c->si; g.Z = sa.Z; g.r = c->r; g.scale = om_s;
  g.W = hyper_ref(c, 0, c->n, c->si, c->n);
  g.Bv = hyper_ref(c, (long)c->si * c->n + (long)c->nh * c->n * c->n + (long)c->n * c->so, 0, 1, c->n);
  if (sp) g.ntiles = ntiles * (1 + sp->nsc);      // a parameter stream has no tangent input here (x' = 0): its pairs go to sob_par_pass
  launch_gw_first(g, c->NB, rows, sb_st);
  // ShapeNet hidden matrices
  for (int j = 0; j < c->nh; ++j) {
    sbase(g); g.IN = sIN + (long)j * c->slot_s; g.DA = sDA + (long)(j + 1) * c->slot_s; g.Z = sa.Z; g.r = c->r; g.scale = om_s;
    g.da_bf16 = wrote_da_bf16 ? 1 : 0;
    g.in_ph16 = wrote_h_ph16 ? 1 : 0;
    const long wslot = (long)c->si * c->n + (long)j * c->n * c->n;
    const long bslot = (long)c->si * c->n + (long)c->nh * c->n * c->n + (long)c->n * c->so + c->n + (long)j * c->n;
    g.W = hyper_ref(c, wslot, c->n, c->n, c->n);
    g.Bv = hyper_ref(c, bslot, 0, 1, c->n);
    if (launch_gw_mfma(g, c->NB, c->NB, rows, sb_st) < 0) return fail(NIF_ERR_STATE, "internal: bf16 dL/da stash rows without a reader of that form");
  }
  // ShapeNet last layer
  {
    sbase(g); g.IN = sIN + (long)c->nh * c->slot_s; g.SM = sa.DU; g.nc = c->so; g.Z = sa.Z; g.r = c->r; g.scale = 1.0f;
    const long wslot = (long)c->si * c->n + (long)c->nh * c->n * c->n;
    const long bslot = wslot + (long)c->n * c->so + c->n + (long)c->nh * c->n;
    g.W = hyper_ref(c, wslot, c->so, c->n, c->so);
    g.Bv = hyper_ref(c, bslot, 0, 1, c->so);
    launch_gw_out(g, c->NB, rows, sb_st);
  }
  }
  // ParameterNet: first, hidden matrices, bottleneck
  float* pST = pa.stash;
  if (!fused_p) {
  base(g); g.DA = pST + (long)(c->nm + 1) * c->slot_p; g.xin = xin; g.ncol = ncol; g.col0 = 0; g.nd = c->pi; g.r = 0; g.scale = om_p;
  g.W = dense_ref(c->first_w, c->pi, c->nst); g.Bv = vec_ref(c->first_b, c->nst);
  launch_gw_first(g, c->NSTB, rows, pb_st);
  for (int mi = 0; mi < c->nm; ++mi) {
    base(g); g.IN = pST + (long)mi * c->slot_p; g.DA = pST + (long)(c->nm + 2 + mi) * c->slot_p; g.r = 0; g.scale = om_p;
    long w_off, b_off;
    if (!c->cfg.p_resblock) { w_off = c->hid_w[mi]; b_off = c->hid_b[mi]; }
    else { const int i = mi / 2; w_off = (mi & 1) ? c->hid_w2[i] : c->hid_w[i]; b_off = (mi & 1) ? c->hid_b2[i] : c->hid_b[i]; }
    g.W = dense_ref(w_off, c->nst, c->nst); g.Bv = vec_ref(b_off, c->nst);
    launch_gw_mfma(g, c->NSTB, c->NSTB, rows, pb_st);
  }
  base(g); g.IN = pST + (long)c->nm * c->slot_p; g.SM = pa.DZ; g.nc = c->r; g.r = 0; g.scale = 1.0f;
  g.W = dense_ref(c->bott_w, c->nst, c->r); g.Bv = vec_ref(c->bott_b, c->r);
  launch_gw_out(g, c->NSTB, rows, pb_st);
  }
  if (sa_out) *sa_out = sa;
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

// k_small's tables of this context (the index map of its LDS images and the tensor descriptors: offsets only), built once
static int ensure_small_tables(nif_ctx* c, const PNetArgs& pa, const SNetArgs& sa) {
  if (c->small_idx) return NIF_OK;
  if (c->capturing) return fail(NIF_ERR_STATE, "small-batch step tables inside a graph capture (nif_graph_begin builds them)");
  std::vector<int> idx, desc;
  small_tables(pa, sa, idx, desc);
  HIPCHK(hipMalloc(&c->small_idx, idx.size() * sizeof(int)));
  HIPCHK(hipMalloc(&c->small_desc, desc.size() * sizeof(int)));
  HIPCHK(hipMemcpy(c->small_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->small_desc, desc.data(), desc.size() * sizeof(int), hipMemcpyHostToDevice));
  return NIF_OK;
}
// r6: the row reduction of a plain step waits for its consumer when nothing else needs [grad | loss] first: nif_adam_step_dev runs it fused
// with the update (k_reduce_adam: one launch less per step -- 5 of the 31 us of a 512-point step, 0.5 % of the 2^20-point one).  Every
// other entry point of the library starts with nif_tail_flush; the reduction is NOT deferred with a communicator attached, inside a graph
// capture, under the profiler (its per-group events would lose the REDUCE group) or when a regulariser adds to the gradient behind it.
int nif_tail_flush(nif_ctx* c) {
  if (!c || !c->tail_pending) return NIF_OK;
  c->tail_pending = false;
  HIPCHK(hipSetDevice(c->dev));
  launch_reduce(c->partial, c->pstride, c->tail_rows, c->loss_partial, c->tail_nloss, c->grad, c->P, c->st);
  HIPCHK(hipGetLastError());
  return NIF_OK;
}
static bool tail_can_defer(const nif_ctx* c) {
  if (!c->opt_fuse_tail || c->comm || c->capturing || c->prof_on) return false;
  if ((c->reg_l1 != 0.f || c->reg_l2 != 0.f) && c->reg_hi > c->reg_lo) return false;
  if (c->sreg_l1 != 0.f || c->sreg_l2 != 0.f) return false;
  return true;
}
// a Keras loss-metric accumulation that nif_metric_accumulate left for the next k_small launch (r6: one launch less per small step):
// everything that would change grad[P] or read the metric without such a launch runs it now
static int metric_flush(nif_ctx* c) {
  if (!c->metric_pending) return NIF_OK;
  TAIL_FLUSH(c)
  c->metric_pending = false;
  launch_metric(c->grad, c->P, c->metric_pending_w, c->metric, c->st);
  HIPCHK(hipGetLastError());
  return NIF_OK;
}
static int loss_grad_core(nif_ctx* c, const float* xin, const float* y, const float* sw, int64_t B, int64_t Bg, int ns,
                          const int* seeds, const float* gt, float wj, const SobPlan* sp = nullptr) {
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)      // (a step whose gradient was never consumed: its rows are about to be overwritten)
  c->last_step_small = false;
  // r6: a small batch of a small net -- ONE launch for the loss and every gradient (k_small: fp32 FMAs straight from theta, no plane
  // packing), then the usual row reduction.  configs[0]'s 512-point steps spent 82 us in eleven tile-kernel launches
  if (ns == 0 && c->opt_small_step && B <= NIF_SMALL_MAX_B && c->kind != NIF_KIND_LASTLAYER && !act_on(c) && !c->opt_fp32_mfma) {
    if (!c->have_params) return fail(NIF_ERR_STATE, "parameters not set (call nif_set_params first)");
    PNetArgs pa; fill_pnet(c, pa, xin, B);
    SNetArgs sa; fill_snet(c, sa, xin, c->pi + c->si, c->pi, B);
    if (small_supported(pa, sa) && small_rows(B) <= 256) {
      int rc = ensure_capacity(c, ((B + 31) / 32) * 32, true); if (rc) return rc;      // (first call: allocates the partial rows, rows_cap = 256)
      const int rows = small_rows(B);
      if (rows > c->rows_cap) return fail(NIF_ERR_STATE, "internal: partial-row buffer too small for the small-batch step");
      if (rows > c->nloss_cap) return fail(NIF_ERR_STATE, "internal: loss partial buffer too small for the small-batch step");
      c->reg_applied = false;
      sa.y = y; sa.sw = sw; sa.loss_partial = c->loss_partial; sa.inv_bg = 1.0f / (float)Bg;
      rc = ensure_small_tables(c, pa, sa); if (rc) return rc;
      const bool mp = c->metric_pending && c->metric != nullptr;
      { ProfScope p_(c, NIF_PROF_SNET);
        launch_small(pa, sa, c->partial, c->pstride, c->P, c->small_idx, c->small_desc, mp ? c->metric : nullptr, c->metric_pending_w,
                     c->grad + c->P, c->st); }
      if (mp) c->metric_pending = false;
      c->last_step_small = true;
      if (c->jac_l1 == 0.f && tail_can_defer(c)) { c->tail_pending = true; c->tail_rows = rows; c->tail_nloss = rows; }
      else { ProfScope pr_(c, NIF_PROF_REDUCE); launch_reduce(c->partial, c->pstride, rows, c->loss_partial, rows, c->grad, c->P, c->st); }
      if (c->jac_l1 != 0.f) { rc = jac_reg_pass(c, xin, B, Bg); if (rc) return rc; }
      HIPCHK(hipGetLastError());
      return NIF_OK;
    }
  }
  int rc = metric_flush(c); if (rc) return rc;
  rc = ensure_packed(c); if (rc) return rc;
  const long ntiles = (B + 31) / 32;
  if (ns > 0) { rc = ensure_packed32(c); if (rc) return rc; }
  if (ns > 0) {
    if (c->kind != NIF_KIND_LASTLAYER && !c->jac_ok)
      return fail(NIF_ERR_INVALID, "Sobolev training: one weight plane and the small hyper-vectors of this shape exceed the 160 KB LDS of a CU");
    if (c->cfg.s_resblock && (c->nh & 1)) return fail(NIF_ERR_INVALID, "resblock ShapeNet with an odd matrix count");
  }
  rc = ensure_capacity(c, ntiles * 32 * (1 + ns), true); if (rc) return rc;
  if (sp && sp->any_par) {   // z' = dz/dp of the parameter columns, in front of the ShapeNet
    PNetArgs pa; fill_pnet(c, pa, xin, B);
    if (!pjac_supported(pa))
      return fail(NIF_ERR_INVALID, "Sobolev x_index on parameter columns: ParameterNets of up to 128 units");
    const long need_zt = (long)c->pi * ntiles * 32 * c->r, need_dzt = 3 * ntiles * 32 * c->r;
    if (need_zt > c->zt_par_cap || need_dzt > c->dzt_par_cap) HIPCHK(hipStreamSynchronize(c->st));
    if (need_zt > c->zt_par_cap) { rc = grow(&c->zt_par, &c->zt_par_cap, need_zt); if (rc) return rc; }
    if (need_dzt > c->dzt_par_cap) { rc = grow(&c->dzt_par, &c->dzt_par_cap, need_dzt); if (rc) return rc; }
    launch_pjac_fwd(pa, c->zt_par, c->st);
  }
  c->reg_applied = false;
  if (c->kind == NIF_KIND_LASTLAYER) return loss_grad_ll(c, xin, y, sw, B, Bg, ns, sp, gt, wj);
  // Two-stream pipeline over chunks of the batch (plain step on the 16-point-tile kernels): the fused ShapeNet kernel of
  // chunk i+1 (VALU / latency bound, 2 workgroups per CU) overlaps the HBM-bound weight-gradient reductions of chunk i
  const long chunk = (ns == 0 && c->use_snet3 && !act_on(c) && c->jac_l1 == 0.f) ? pipe_chunk_points(c, B) : 0;
  if (chunk <= 0 || chunk >= B) {
    int nloss = 0;
    SNetArgs sae;
    // the compute-bound ParameterNet adjoint runs on a second stream NEXT TO the HBM-bound weight-gradient reductions of
    // the ShapeNet (both only need the fused kernel's outputs); joined in front of the row reduction
    const bool side = c->opt_side_pnet && B >= 65536;
    if (side) { rc = ensure_pipe(c, 1); if (rc) return rc; }
    rc = step_chunk(c, xin, y, sw, 0, B, Bg, ns, seeds, gt, wj, c->st, c->st, side ? c->st2 : c->st, c->partial, c->loss_partial,
                    &nloss, 0, true, &sae, sp);
    if (rc) return rc;
    if (side) {
      HIPCHK(hipEventRecord(c->ev_done, c->st2));
      HIPCHK(hipStreamWaitEvent(c->st, c->ev_done, 0));
    }
    const int rows = rows_for(c, ntiles);
    if (ns == 0 && !act_on(c) && c->jac_l1 == 0.f && !(sp && sp->any_par) && tail_can_defer(c)) {
      c->tail_pending = true; c->tail_rows = rows; c->tail_nloss = nloss;
      HIPCHK(hipGetLastError());
      return NIF_OK;
    }
    ProfScope pr_(c, NIF_PROF_REDUCE);
    launch_reduce(c->partial, c->pstride, rows, c->loss_partial, nloss, c->grad, c->P, c->st);
    if (act_on(c)) {   // the plane side of the activity regulariser and its loss, on top of the reduced gradient
      const long need = (long)NIF_ACT_SLABS * (c->r + 1) * c->po;
      if (need > c->act_part_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->act_part, &c->act_part_cap, need); if (rc) return rc; }
      const bool l1 = c->act_l2 == 0.f;
      launch_actreg_planes(l1, c->theta, c->last_w, c->last_b, c->r, c->po, c->Z, B, NIF_ACT_SLABS, c->act_part, c->st);
      launch_actreg_apply(c->act_part, NIF_ACT_SLABS, c->r, c->po, (l1 ? c->act_l1 : c->act_l2) / (float)Bg, c->last_w, c->last_b,
                          c->act_loss, (int)((B + 255) / 256), c->grad, c->P, c->st);
    }
    if (c->jac_l1 != 0.f) { rc = jac_reg_pass(c, xin, B, Bg); if (rc) return rc; }
    if (sp && sp->any_par) { rc = sob_par_pass(c, xin, B, Bg, *sp, sae); if (rc) return rc; }
    HIPCHK(hipGetLastError());
    return NIF_OK;
  }
  const int nchunk = (int)((B + chunk - 1) / chunk);
  rc = ensure_pipe(c, nchunk); if (rc) return rc;
  HIPCHK(hipEventRecord(c->ev_start, c->st));             // stream B starts behind whatever st has queued (packing, Adam)
  HIPCHK(hipStreamWaitEvent(c->st2, c->ev_start, 0));
  const long lp_stride = c->nloss_cap / nchunk;            // loss partials of chunk i at i * lp_stride
  for (int i = 0; i < nchunk; ++i) {
    const long off = (long)i * chunk;
    const long Bc = B - off < chunk ? B - off : chunk;
    int nloss = 0;
    rc = step_chunk(c, xin, y, sw, off, Bc, Bg, 0, nullptr, nullptr, 0.f, c->st, c->st2, c->st2, c->partial,
                    c->loss_partial + i * lp_stride, &nloss, i, false);
    if (rc) return rc;
    if (nloss > lp_stride) return fail(NIF_ERR_STATE, "loss partial buffer too small for the chunk pipeline");
    // this chunk's partial rows -> row i of the second-level buffer (loss in column P); stream B is in order, so the
    // next chunk's reductions may overwrite the partial rows afterwards
    const int rows = rows_for(c, (Bc + 31) / 32);
    launch_reduce(c->partial, c->pstride, rows, c->loss_partial + i * lp_stride, nloss, c->chunk_grad + (long)i * c->pstride, c->P, c->st2);
  }
  HIPCHK(hipEventRecord(c->ev_done, c->st2));
  HIPCHK(hipStreamWaitEvent(c->st, c->ev_done, 0));
  {
    ProfScope pr_(c, NIF_PROF_REDUCE);   // rows of the chunks -> flat gradient | loss (column P rides along as a regular column)
    launch_reduce(c->chunk_grad, c->pstride, nchunk, nullptr, 0, c->grad, c->P + 1, c->st);
  }
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

extern "C" int nif_loss_grad_dev(nif_ctx* c, const float* xin, const float* y, const float* sw, int64_t B, int64_t Bg) {
  if (!c || !xin || !y || B <= 0 || Bg < B) return fail(NIF_ERR_INVALID, "bad argument");
  return loss_grad_core(c, xin, y, sw, B, Bg, 0, nullptr, nullptr, 0.f);
}

// Size every workspace of a training step over up to B_max points (n_tangents Sobolev seeds, 0 = plain step) now, so
// that no hipMalloc / stream synchronisation happens inside a later (timed) step.
extern "C" int nif_reserve(nif_ctx* c, int64_t B_max, int32_t n_tangents) {
  if (!c || B_max <= 0 || n_tangents < 0 || n_tangents > 16) return fail(NIF_ERR_INVALID, "bad argument");
  if (n_tangents > 3) n_tangents = 3;      // (more x_index columns run as passes over groups of three)
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = ensure_packed(c); if (rc) return rc;
  const long ntiles = (B_max + 31) / 32;
  rc = ensure_capacity(c, ntiles * 32 * (1 + n_tangents), true); if (rc) return rc;
  if (c->kind == NIF_KIND_LASTLAYER) {
    if (c->use_ll4) {
      SNetArgs sa; fill_snet_ll(c, sa, nullptr, c->pi + c->si, c->pi, B_max);
      const int nblk = launch_snet4(sa, true, true, c->st);
      const long need = (long)nblk * 4 * snet3_ring_floats_per_wave(c->n, c->nh);
      if (need > c->dring_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->dring, &c->dring_cap, need); if (rc) return rc; }
    }
    return NIF_OK;
  }
  SNetArgs sa; fill_snet(c, sa, nullptr, c->pi + c->si, c->pi, B_max);
  int nloss = 0;
  const int seeds[3] = {0, 0, 0};
  rc = snet_plan(c, sa, n_tangents, seeds, &nloss); if (rc) return rc;
  const long chunk = (n_tangents == 0 && c->use_snet3) ? pipe_chunk_points(c, B_max) : 0;
  if (chunk > 0 && chunk < B_max) { rc = ensure_pipe(c, (int)((B_max + chunk - 1) / chunk)); if (rc) return rc; }
  return NIF_OK;
}

static int sobolev_plan(nif_ctx* c, const int32_t* x_idx, int32_t nx, SobPlan* sp) {
  if (!x_idx || nx < 1 || nx > 3) return fail(NIF_ERR_INVALID, "internal: a Sobolev pass carries 1..3 columns");
  memset(sp, 0, sizeof(*sp));
  sp->ns = nx;
  for (int d = 0; d < nx; ++d) {
    if (x_idx[d] < 0 || x_idx[d] >= c->pi + c->si) return fail(NIF_ERR_INVALID, "Sobolev x_index out of range (0 <= i < pi_dim + si_dim)");
    for (int e = 0; e < d; ++e)
      if (x_idx[e] == x_idx[d]) return fail(NIF_ERR_INVALID, "Sobolev x_index lists a column twice");
  }
  int q = 0;
  for (int d = 0; d < nx; ++d)
    if (x_idx[d] >= c->pi) { sp->seeds[q] = x_idx[d] - c->pi; sp->par[q] = -1; sp->gcol[q] = d; ++q; }
  sp->nsc = q;
  for (int d = 0; d < nx; ++d)
    if (x_idx[d] < c->pi) { sp->seeds[q] = 0; sp->par[q] = x_idx[d]; sp->gcol[q] = d; ++q; sp->any_par = true; }
  for (; q < 3; ++q) { sp->seeds[q] = 0; sp->par[q] = -1; sp->gcol[q] = q; }
  return NIF_OK;
}
// The columns of x_index in groups of <= 3 (the kernels carry up to three tangent streams): the derivative term is a sum over the
// columns, so group k > 0 is one more pass with the primal mse (and every regularisation term) switched off, its [grad | loss]
// added to the first pass's.  y_idx = NULL: every output; otherwise the derivative term covers the listed outputs only
// (gradient.py:207-231).  dydx rows are [so][nx] either way (rows of unlisted outputs are not read into the loss).
static int sobolev_check(nif_ctx* c, const int32_t* x_idx, int32_t nx, const int32_t* y_idx, int32_t ny, unsigned* ymask, int* ny_out) {
  if (!x_idx || nx < 1 || nx > 16) return fail(NIF_ERR_INVALID, "Sobolev training takes 1..16 input columns in x_index");
  for (int d = 0; d < nx; ++d) {
    if (x_idx[d] < 0 || x_idx[d] >= c->pi + c->si) return fail(NIF_ERR_INVALID, "Sobolev x_index out of range (0 <= i < pi_dim + si_dim)");
    for (int e = 0; e < d; ++e)
      if (x_idx[e] == x_idx[d]) return fail(NIF_ERR_INVALID, "Sobolev x_index lists a column twice");
  }
  *ymask = 0; *ny_out = 0;
  if (y_idx) {
    if (ny < 1 || ny > c->so) return fail(NIF_ERR_INVALID, "Sobolev y_index: 1..output_dim distinct outputs");
    for (int i = 0; i < ny; ++i) {
      if (y_idx[i] < 0 || y_idx[i] >= c->so) return fail(NIF_ERR_INVALID, "Sobolev y_index out of range");
      if (*ymask & (1u << y_idx[i])) return fail(NIF_ERR_INVALID, "Sobolev y_index lists an output twice");
      *ymask |= 1u << y_idx[i];
    }
    *ny_out = ny;
  }
  return NIF_OK;
}
// Columns per pass of a Sobolev step / forward: three tangent streams where the kernel's working set allows it.  A parameter-column
// stream of the hypernetwork classes keeps 4 x (r x 80) floats of per-wave LDS next to the planes (k_sob.hip launch_sob): at 128
// units and latent_dim 8 three of them exceed the 160 KB by a few KB (fuzz sweep r04, seed 7, case 5) -- the passes then take two
// columns, or one, instead of refusing the shape.  Probed with the query the step itself makes (snet_plan), no launch.
static int sob_cols_per_pass(nif_ctx* c, const int32_t* x_idx, int nx) {
  if (c->kind == NIF_KIND_LASTLAYER) return 3;        // (parameter columns are heads of the epilogue there, not streams)
  if (ensure_packed(c) != NIF_OK) return 3;           // (surfaces again, with its message, in the step)
  for (int gs = 3; gs > 1; --gs) {
    bool ok = true;
    for (int g0 = 0; g0 < nx && ok; g0 += gs) {
      const int ng = nx - g0 < gs ? nx - g0 : gs;
      SobPlan sp;
      if (sobolev_plan(c, x_idx + g0, ng, &sp) != NIF_OK) return 3;
      SNetArgs sa; fill_snet(c, sa, nullptr, c->pi + c->si, c->pi, 64);
      SobPar spq{};
      for (int d = 0; d < 3; ++d) spq.par[d] = sp.par[d];
      ok = launch_sob(sa, true, ng, sp.seeds, nullptr, 0.f, nullptr, nullptr, true, c->st, &spq) >= 0;
    }
    if (ok) return gs;
  }
  return 1;
}

extern "C" int nif_sobolev_loss_grad_dev_y(nif_ctx* c, const float* xin, const float* y, const float* dydx, const float* sw,
                                           int64_t B, int64_t Bg, const int32_t* x_idx, int32_t nx, const int32_t* y_idx, int32_t ny,
                                           float w_jac) {
  if (!c || !xin || !y || !dydx || B <= 0 || Bg < B) return fail(NIF_ERR_INVALID, "bad argument");
  unsigned ymask; int nys;
  int rc = sobolev_check(c, x_idx, nx, y_idx, ny, &ymask, &nys); if (rc) return rc;
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  const int gs = sob_cols_per_pass(c, x_idx, nx);
  const int ngroups = (nx + gs - 1) / gs;
  if (ngroups > 1 && !c->sob_acc) HIPCHK(hipMalloc(&c->sob_acc, sizeof(float) * (size_t)(c->P + 1)));
  const float jac_l1 = c->jac_l1, act_l1 = c->act_l1, act_l2 = c->act_l2;
  for (int k = 0; k < ngroups; ++k) {
    const int g0 = gs * k, ng = nx - g0 < gs ? nx - g0 : gs;
    SobPlan sp;
    rc = sobolev_plan(c, x_idx + g0, ng, &sp); if (rc) break;
    for (int q = 0; q < 3; ++q) sp.gcol[q] += g0;
    sp.gstride = nx; sp.nx_all = nx; sp.ny = nys; sp.ymask = ymask; sp.no_primal = k > 0;
    if (k > 0) { c->jac_l1 = 0.f; c->act_l1 = 0.f; c->act_l2 = 0.f; }      // the regularisation losses belong to the first pass
    rc = loss_grad_core(c, xin, y, sw, B, Bg, ng, sp.seeds, dydx, w_jac, &sp);
    if (rc) break;
    if (ngroups > 1) {
      if (k == 0) { if (hipMemcpyAsync(c->sob_acc, c->grad, sizeof(float) * (size_t)(c->P + 1), hipMemcpyDeviceToDevice, c->st) != hipSuccess) { rc = fail(NIF_ERR_HIP, "hipMemcpyAsync"); break; } }
      else launch_axpy_cols(c->sob_acc, c->grad, c->P, c->P, c->st);
    }
  }
  c->jac_l1 = jac_l1; c->act_l1 = act_l1; c->act_l2 = act_l2;
  if (rc) return rc;
  if (ngroups > 1) HIPCHK(hipMemcpyAsync(c->grad, c->sob_acc, sizeof(float) * (size_t)(c->P + 1), hipMemcpyDeviceToDevice, c->st));
  c->reg_applied = false;
  HIPCHK(hipGetLastError());
  return NIF_OK;
}
extern "C" int nif_sobolev_loss_grad_dev(nif_ctx* c, const float* xin, const float* y, const float* dydx, const float* sw,
                                         int64_t B, int64_t Bg, const int32_t* x_idx, int32_t nx, float w_jac) {
  return nif_sobolev_loss_grad_dev_y(c, xin, y, dydx, sw, B, Bg, x_idx, nx, nullptr, 0, w_jac);
}
static int sobolev_forward_group(nif_ctx* c, const float* xin, int64_t B, const SobPlan& sp, int nx, float* u, float* dudx);
extern "C" int nif_sobolev_forward_dev(nif_ctx* c, const float* xin, int64_t B, const int32_t* x_idx, int32_t nx, float* u,
                                       float* dudx) {
  if (!c || !xin || !u || !dudx || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  unsigned ymask; int nys;
  int rc = sobolev_check(c, x_idx, nx, nullptr, 0, &ymask, &nys); if (rc) return rc;
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  const int gs = sob_cols_per_pass(c, x_idx, nx);
  for (int g0 = 0; g0 < nx; g0 += gs) {      // groups of <= 3 columns, each filling its columns of the [B][so][nx] rows
    const int ng = nx - g0 < gs ? nx - g0 : gs;
    SobPlan sp;
    rc = sobolev_plan(c, x_idx + g0, ng, &sp); if (rc) return rc;
    for (int q = 0; q < 3; ++q) sp.gcol[q] += g0;
    sp.gstride = nx; sp.nx_all = nx;
    rc = sobolev_forward_group(c, xin, B, sp, ng, u, dudx); if (rc) return rc;
  }
  return NIF_OK;
}
static int sobolev_forward_group(nif_ctx* c, const float* xin, int64_t B, const SobPlan& sp, int nx, float* u, float* dudx) {
  int rc;
  HIPCHK(hipSetDevice(c->dev));
  rc = ensure_packed(c); if (rc) return rc;
  rc = ensure_packed32(c); if (rc) return rc;
  if (c->kind != NIF_KIND_LASTLAYER && !c->jac_ok)
    return fail(NIF_ERR_INVALID, "Sobolev path: one weight plane and the small hyper-vectors of this shape exceed the 160 KB LDS of a CU");
  rc = ensure_capacity(c, B, false); if (rc) return rc;
  PNetArgs pa; fill_pnet(c, pa, xin, B);
  launch_pnet(pa, c->NSTB, false, c->st);
  SobPar spar{};
  for (int d = 0; d < 3; ++d) { spar.par[d] = sp.par[d]; spar.gcol[d] = sp.gcol[d]; }
  spar.gstride = sp.gstride; spar.nx_all = sp.nx_all;
  if (c->kind == NIF_KIND_LASTLAYER) {
    if (sp.any_par) {       // parameter columns: heads of the epilogue on z' = dz/dp
      if (!pjac_supported(pa))
        return fail(NIF_ERR_INVALID, "Sobolev x_index on parameter columns: ParameterNets of up to 128 units");
      const long need_zt = (long)c->pi * ((B + 31) / 32) * 32 * c->r;
      if (need_zt > c->zt_par_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->zt_par, &c->zt_par_cap, need_zt); if (rc) return rc; }
      launch_pjac_fwd(pa, c->zt_par, c->st);
      spar.npar = sp.ns - sp.nsc; spar.ZT = c->zt_par;
      for (int e = 0; e < spar.npar; ++e) { spar.parc[e] = sp.par[sp.nsc + e]; spar.pcol[e] = sp.gcol[sp.nsc + e]; }
      for (int d = 0; d < 3; ++d) spar.par[d] = -1;
    }
    SNetArgs sl; rc = fill_snet_ll_sob(c, sl, xin, B); if (rc) return rc;
    sl.u_out = u;
    if (launch_sob(sl, false, sp.nsc, sp.seeds, nullptr, 0.f, nullptr, dudx, false, c->st, &spar) < 0)
      return fail(NIF_ERR_INVALID, "Sobolev path: the kernel's working set of this shape does not fit the 160 KB LDS of a CU");
    HIPCHK(hipGetLastError());
    return NIF_OK;
  }
  if (sp.any_par) {
    if (!pjac_supported(pa))
      return fail(NIF_ERR_INVALID, "Sobolev x_index on parameter columns: ParameterNets of up to 128 units");
    const long need_zt = (long)c->pi * ((B + 31) / 32) * 32 * c->r;
    if (need_zt > c->zt_par_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->zt_par, &c->zt_par_cap, need_zt); if (rc) return rc; }
    launch_pjac_fwd(pa, c->zt_par, c->st);
    spar.ZT = c->zt_par;
  }
  SNetArgs sa; fill_snet(c, sa, xin, c->pi + c->si, c->pi, B);
  sa.u_out = u;
  if (launch_sob(sa, false, nx, sp.seeds, nullptr, 0.f, nullptr, dudx, false, c->st, &spar) < 0)
    return fail(NIF_ERR_INVALID, "Sobolev path: the kernel's working set of this shape (units, latent_dim, parameter columns) does not fit the 160 KB LDS of a CU");
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

static void apply_reg(nif_ctx* c) {
  if (c->reg_applied) return;
  if ((c->reg_l1 != 0.f || c->reg_l2 != 0.f) && c->reg_hi > c->reg_lo)
    launch_reg(c->theta, c->grad, c->reg_lo, c->reg_hi, c->P, c->reg_l1, c->reg_l2, c->st);
  if ((c->sreg_l1 != 0.f || c->sreg_l2 != 0.f) && c->kind == NIF_KIND_LASTLAYER)
    launch_reg(c->theta, c->grad, c->s_first_w, c->ll_bias, c->P, c->sreg_l1, c->sreg_l2, c->st);
  c->reg_applied = true;
}
// cfg_shape_net["l1_reg" / "l2_reg"] of the last-layer class (nif/model.py:1028-1039, handed to every SIREN / SIREN_ResNet of the
// shared ShapeNet at :1168-1211; siren.py:266-269, :393-398 add them for kernels AND biases): theta[s_first_w, ll_bias) -- first,
// hidden and bottleneck layers, not last_layer_bias (BiasAddLayer has no regulariser, mlp.py:245-262)
extern "C" int nif_set_shapenet_regularizer(nif_ctx* c, float l1, float l2) {
  if (!c || l1 < 0.f || l2 < 0.f) return fail(NIF_ERR_INVALID, "bad argument");
  if ((l1 != 0.f || l2 != 0.f) && c->kind != NIF_KIND_LASTLAYER)
    return fail(NIF_ERR_INVALID, "ShapeNet weight regularisers exist for the last-layer-parameterised class only (the other classes' ShapeNet weights are ParameterNet outputs)");
  if (c->kind == NIF_KIND_LASTLAYER && !(c->s_first_w < c->ll_bias && c->ll_bias <= c->P)) return fail(NIF_ERR_STATE, "ShapeNet parameter range");
  c->sreg_l1 = l1; c->sreg_l2 = l2;
  return NIF_OK;
}
extern "C" int nif_set_regularizer(nif_ctx* c, float l1, float l2, int64_t lo, int64_t hi) {
  if (!c || lo < 0 || hi > c->P || lo > hi || l1 < 0.f || l2 < 0.f) return fail(NIF_ERR_INVALID, "bad argument");
  c->reg_l1 = l1; c->reg_l2 = l2; c->reg_lo = lo; c->reg_hi = hi;
  return NIF_OK;
}
// cfg_parameter_net["jac_reg"] (nif/model.py:353-375 -> JacRegLatentLayer, nif/layers/gradient.py:52-127)
extern "C" int nif_set_jac_regularizer(nif_ctx* c, float l1) {
  if (!c || l1 < 0.f) return fail(NIF_ERR_INVALID, "bad argument");
  if (l1 != 0.f) {
    PNetArgs pa; fill_pnet(c, pa, nullptr, 32);
    if (!pjac_supported(pa)) return fail(NIF_ERR_INVALID, "jac_reg: at most 3 parameter inputs (pi_dim <= 3)");
  }
  c->jac_l1 = l1;
  return NIF_OK;
}
// The regulariser's own pass, on top of the reduced main gradient: loss += l1 mean (dz/dp)^2 and its gradient w.r.t. the
// ParameterNet's first / hidden / bottleneck variables (the hyper layer is downstream of z and does not see it).
// mu_blk != null: the same pass as the adjoint of the Sobolev step's parameter streams -- dL/dz'_d is GIVEN (block mu_blk[d]
// of c->dzt_par, written by k_sob), no loss term
static int jac_reg_pass(nif_ctx* c, const float* xin, long B, long Bg, const int* mu_blk) {
  const int pi = c->pi, grp = pjac_group(), ndmax = pi < grp ? pi : grp;
  const long ntiles = (B + 31) / 32;
  int rc = ensure_capacity(c, ntiles * 32 * (1 + ndmax), true); if (rc) return rc;
  const long need_mu = (long)(1 + ndmax) * ntiles * 32 * c->r;
  if (need_mu > c->jac_mu_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->jac_mu, &c->jac_mu_cap, need_mu); if (rc) return rc; }
  if (!c->jac_tmp) HIPCHK(hipMalloc(&c->jac_tmp, sizeof(float) * (size_t)(c->P + 2)));
  const long nlp = (B + 127) / 128;
  if (nlp > c->act_loss_cap) { HIPCHK(hipStreamSynchronize(c->st)); rc = grow(&c->act_loss, &c->act_loss_cap, nlp); if (rc) return rc; }
  PNetArgs pa; fill_pnet(c, pa, xin, B);
  const float coef = c->jac_l1 / ((float)Bg * (float)c->r * (float)pi);
  // r4: passes over groups of parameter columns (k_pjac carries pjac_group() tangents): loss and gradient are sums over the
  // columns -- every pass reduces its own operand pairs (the primal pair carries the part of lambda that ITS tangents feed)
  for (int c0 = 0; c0 < pi; c0 += grp) {
    const int nd = pi - c0 < grp ? pi - c0 : grp;
    if (mu_blk) { bool any = false; for (int d = 0; d < nd; ++d) any = any || mu_blk[c0 + d] >= 0; if (!any) continue; }
    const int nloss = mu_blk ? launch_pjac_adj(pa, c->dzt_par, mu_blk, c->jac_mu, c->act_loss, c0, nd, c->st)
                             : launch_pjac(pa, coef, c->jac_mu, c->act_loss, c0, nd, c->st);
    const long nt_all = ntiles * (1 + nd);
    const int rows = rows_for(c, nt_all);
    GwArgs g;
    auto base = [&](GwArgs& q) {
      memset(&q, 0, sizeof(q));
      q.ntiles = nt_all; q.zt_mod = ntiles; q.bias_ntiles = ntiles; q.B = B; q.partial = c->partial; q.pstride = c->pstride;
      q.has_bias = 1; q.scale = 1.0f; q.r = 0;
      for (int d = 0; d < 3; ++d) q.seed[d] = d < nd ? c0 + d : 0;
    };
    float* pST = c->stash_p;
    const int ncol = c->pi + c->si;
    base(g); g.DA = pST + (long)(c->nm + 1) * c->slot_p; g.xin = xin; g.ncol = ncol; g.col0 = 0; g.nd = pi; g.scale = pa.omega;
    g.W = dense_ref(c->first_w, pi, c->nst); g.Bv = vec_ref(c->first_b, c->nst);
    launch_gw_first(g, c->NSTB, rows, c->st);
    for (int mi = 0; mi < c->nm; ++mi) {
      base(g); g.IN = pST + (long)mi * c->slot_p; g.DA = pST + (long)(c->nm + 2 + mi) * c->slot_p; g.scale = pa.omega;
      long w_off, b_off;
      if (!c->cfg.p_resblock) { w_off = c->hid_w[mi]; b_off = c->hid_b[mi]; }
      else { const int i = mi / 2; w_off = (mi & 1) ? c->hid_w2[i] : c->hid_w[i]; b_off = (mi & 1) ? c->hid_b2[i] : c->hid_b[i]; }
      g.W = dense_ref(w_off, c->nst, c->nst); g.Bv = vec_ref(b_off, c->nst);
      launch_gw_mfma(g, c->NSTB, c->NSTB, rows, c->st);
    }
    base(g); g.IN = pST + (long)c->nm * c->slot_p; g.SM = c->jac_mu; g.nc = c->r; g.scale = 1.0f;
    g.W = dense_ref(c->bott_w, c->nst, c->r); g.Bv = vec_ref(c->bott_b, c->r);
    launch_gw_out(g, c->NSTB, rows, c->st);
    // the ParameterNet core variables are the first last_w columns of a partial row; column last_w of the result = the loss term
    launch_reduce(c->partial, c->pstride, rows, c->act_loss, nloss, c->jac_tmp, c->last_w, c->st);
    launch_axpy_cols(c->grad, c->jac_tmp, c->last_w, c->P, c->st);
  }
  HIPCHK(hipGetLastError());
  return NIF_OK;
}

// Sobolev step with parameter seeds, the part the main reductions do not see (k_sob_dev.h, PAR).  For a parameter stream d:
//   dL/dM^(k) += w0 sum_p zt'_k h_in nu_d^T , dL/db^(k) += sum_p zt'_k nu_d  (k < r): the SAME reductions as the main step with
//   (h_in, nu_d) as the operand pair and zt' = dz/dp in the latent's place.  They run after the main row reduction, into the
//   same partial rows; only the hyper KERNEL columns are kept (the kernels also fill the constant plane = hyper bias columns
//   with sum h_in nu^T, which zt'_r = 0 excludes).  Then the ParameterNet side: the adjoint of (z, z') for the dL/dz' that
//   k_sob left in c->dzt_par (jac_reg_pass with given mu).
static int sob_par_pass(nif_ctx* c, const float* xin, long B, long Bg, const SobPlan& sp, const SNetArgs& sa) {
  const long ntiles = (B + 31) / 32;
  const int ncol = c->pi + c->si;
  const int rows = rows_for(c, ntiles);
  if (!c->jac_tmp) HIPCHK(hipMalloc(&c->jac_tmp, sizeof(float) * (size_t)(c->P + 2)));
  const long blk_s = ntiles * 1024 * c->NB;                 // floats of one block of tiles in a ShapeNet stash slot
  float* sIN = sa.stash; float* sDA = sa.stash + (long)(c->nh + 1) * c->slot_s;
  const long kcols = (long)c->r * c->po;                     // the hyper kernel [r][po] = columns last_w .. last_w + r*po
  int mu_blk[16];
  for (int q = 0; q < 16; ++q) mu_blk[q] = -1;
  for (int d = sp.nsc; d < sp.ns; ++d) {
    const int col = sp.par[d];
    mu_blk[col] = d;
    const float* ZTd = c->zt_par + (long)col * ntiles * 32 * c->r;
    GwArgs g;
    auto base = [&](GwArgs& q) {
      memset(&q, 0, sizeof(q));
      q.ntiles = ntiles; q.zt_mod = ntiles; q.bias_ntiles = ntiles; q.B = B; q.partial = c->partial; q.pstride = c->pstride;
      q.has_bias = 1; q.scale = 1.0f; q.Z = ZTd; q.r = c->r;
    };
    base(g); g.DA = sDA + (1 + d) * blk_s; g.xin = xin; g.ncol = ncol; g.col0 = c->pi; g.nd = c->si; g.scale = sa.omega;
    g.W = hyper_ref(c, 0, c->n, c->si, c->n);
    g.Bv = hyper_ref(c, (long)c->si * c->n + (long)c->nh * c->n * c->n + (long)c->n * c->so, 0, 1, c->n);
    launch_gw_first(g, c->NB, rows, c->st);
    for (int j = 0; j < c->nh; ++j) {
      base(g); g.IN = sIN + (long)j * c->slot_s; g.DA = sDA + (long)(j + 1) * c->slot_s + (1 + d) * blk_s; g.scale = sa.omega;
      const long wslot = (long)c->si * c->n + (long)j * c->n * c->n;
      const long bslot = (long)c->si * c->n + (long)c->nh * c->n * c->n + (long)c->n * c->so + c->n + (long)j * c->n;
      g.W = hyper_ref(c, wslot, c->n, c->n, c->n);
      g.Bv = hyper_ref(c, bslot, 0, 1, c->n);
      launch_gw_mfma(g, c->NB, c->NB, rows, c->st);
    }
    {
      base(g); g.IN = sIN + (long)c->nh * c->slot_s; g.SM = sa.DU + (long)(1 + d) * ntiles * c->so * 32; g.nc = c->so;
      const long wslot = (long)c->si * c->n + (long)c->nh * c->n * c->n;
      const long bslot = wslot + (long)c->n * c->so + c->n + (long)c->nh * c->n;
      g.W = hyper_ref(c, wslot, c->so, c->n, c->so);
      g.Bv = hyper_ref(c, bslot, 0, 1, c->so);
      launch_gw_out(g, c->NB, rows, c->st);
    }
    launch_reduce(c->partial + c->last_w, c->pstride, rows, nullptr, 0, c->jac_tmp, kcols, c->st);
    launch_axpy_cols(c->grad + c->last_w, c->jac_tmp, kcols, c->P - c->last_w, c->st);
  }
  HIPCHK(hipGetLastError());
  return jac_reg_pass(c, xin, B, Bg, mu_blk);
}

// compile(loss=...) of the Keras surface (README.md:33 'mse'): the per-element loss of every training / evaluation entry point of this
// context -- 0 'mse', 1 'mae', 2 'huber' (delta 1), 3 'log_cosh' (include/nif_hip.h nif_loss); the Sobolev step applies it to both outputs
extern "C" int nif_set_loss(nif_ctx* c, int32_t kind) {
  if (!c || kind < 0 || kind > 3) return fail(NIF_ERR_INVALID, "nif_set_loss: 0 mse, 1 mae, 2 huber, 3 log_cosh");
  c->loss_kind = kind;
  return NIF_OK;
}
// Keras activity_regularizer of the last ParameterNet layer (nif/model.py:118-125, :226, :659, :731)
extern "C" int nif_set_activity_regularizer(nif_ctx* c, float l1, float l2) {
  if (!c || l1 < 0.f || l2 < 0.f) return fail(NIF_ERR_INVALID, "bad argument");
  if ((l1 != 0.f || l2 != 0.f) && c->kind != NIF_KIND_LASTLAYER && c->r > actreg_max_r()) return fail(NIF_ERR_INVALID, "activity regularisers: latent_dim <= 64");
  c->act_l1 = l2 != 0.f ? 0.f : l1; c->act_l2 = l2;
  return NIF_OK;
}
extern "C" int nif_metric_accumulate(nif_ctx* c, float weight) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  if (!c->metric) { HIPCHK(hipMalloc(&c->metric, 2 * sizeof(double))); HIPCHK(hipMemsetAsync(c->metric, 0, 2 * sizeof(double), c->st)); }
  int rc = metric_flush(c); if (rc) return rc;
  if (c->last_step_small && c->opt_small_step && !c->capturing && !c->comm) {     // behind a small step: rides in the next k_small launch (grad[P] is not
    c->metric_pending = true; c->metric_pending_w = weight;          // touched before that launch's row reduction; every other path flushes)
    return NIF_OK;
  }
  launch_metric(c->grad, c->P, weight, c->metric, c->st);
  HIPCHK(hipGetLastError());
  return NIF_OK;
}
extern "C" int nif_metric_read(nif_ctx* c, double* sum, double* cnt, int reset) {
  if (!c || !sum || !cnt) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  double h[2] = {0.0, 0.0};
  { const int rcf = metric_flush(c); if (rcf) return rcf; }
  if (c->metric) {
    HIPCHK(hipMemcpyAsync(h, c->metric, 2 * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (reset) HIPCHK(hipMemsetAsync(c->metric, 0, 2 * sizeof(double), c->st));
  }
  *sum = h[0]; *cnt = h[1];
  return NIF_OK;
}
extern "C" int nif_adam_step_dev(nif_ctx* c, const nif_adam* opt) {
  if (!c || !opt) return fail(NIF_ERR_INVALID, "null");
  if (!c->have_params) return fail(NIF_ERR_STATE, "parameters not set");
  HIPCHK(hipSetDevice(c->dev));
  if (c->tail_pending && !c->capturing && tail_can_defer(c)) {      // the step's row reduction and the update in ONE launch
    c->tail_pending = false;
    c->step += 1;
    const double t = (double)c->step;
    const double lr_t = (double)opt->lr * std::sqrt(1.0 - std::pow((double)opt->beta2, t)) / (1.0 - std::pow((double)opt->beta1, t));
    launch_reduce_adam(c->partial, c->pstride, c->tail_rows, c->loss_partial, c->tail_nloss, c->grad, c->P, c->theta, c->m, c->v,
                       (float)lr_t, opt->beta1, opt->beta2, opt->eps, c->st);
    HIPCHK(hipGetLastError());
    c->packed = false; c->packed32 = false; c->packed_p32 = false;
    c->reg_applied = false;
    return NIF_OK;
  }
  TAIL_FLUSH(c)
  apply_reg(c);
  if (c->capturing) {      // inside nif_graph_begin / _end: hyper-parameters and iteration count come from device memory at replay time
    launch_adam_dev(c->theta, c->grad, c->m, c->v, c->P, c->adam_dev, c->st);
    HIPCHK(hipGetLastError());
    c->step += 1; c->cap_steps += 1;
    c->packed = false; c->packed32 = false; c->packed_p32 = false;
    c->reg_applied = false;
    return NIF_OK;
  }
  c->step += 1;
  const double t = (double)c->step;
  const double lr_t = (double)opt->lr * std::sqrt(1.0 - std::pow((double)opt->beta2, t)) / (1.0 - std::pow((double)opt->beta1, t));
  { ProfScope p_(c, NIF_PROF_ADAM);
    launch_adam(c->theta, c->grad, c->m, c->v, c->P, (float)lr_t, opt->beta1, opt->beta2, opt->eps, c->st); }
  HIPCHK(hipGetLastError());
  c->packed = false; c->packed32 = false; c->packed_p32 = false;
  // the regulariser term belongs to ONE gradient: a following step that skips nif_loss_grad_dev (nif_zero_grad on a rank
  // whose shard ran out of rows) must add it again, like the ranks that did compute a gradient
  c->reg_applied = false;
  return NIF_OK;
}

// ---- captured training steps ----------------------------------------------------------------------------------------------------
// Small batches are launch bound (configs[0]: 13 kernels of 3-5 us behind 4-8 us of launch overhead each): between nif_graph_begin
// and nif_graph_end every device-side call of this context (nif_loss_grad_dev, nif_sobolev_loss_grad_dev[_y], nif_adam_step_dev,
// nif_metric_accumulate, nif_gather_rows_dev ...) is RECORDED into a hipGraph instead of executed; nif_graph_launch replays the
// whole sequence -- an epoch of Model.fit -- with one submission.  The batch pointers are baked in (the resident table does not
// move between epochs); Adam's hyper-parameters and iteration count live in device memory (AdamDev) and are refreshed per launch.
// Everything a step needs must exist before the capture (nif_reserve): a workspace that would have to grow fails the capture.
extern "C" int nif_graph_begin(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  if (c->capturing) return fail(NIF_ERR_STATE, "nif_graph_begin: already capturing");
  if (c->comm) return fail(NIF_ERR_STATE, "nif_graph_begin: not with a communicator attached (the all-reduce is not captured)");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = metric_flush(c); if (rc) return rc;
  c->last_step_small = false;
  rc = ensure_packed(c); if (rc) return rc;
  if (c->kind != NIF_KIND_LASTLAYER) {      // (a captured small step must find its tables)
    PNetArgs pa; fill_pnet(c, pa, nullptr, 32);
    SNetArgs sa; fill_snet(c, sa, nullptr, c->pi + c->si, c->pi, 32);
    if (small_supported(pa, sa)) { rc = ensure_small_tables(c, pa, sa); if (rc) return rc; }
  }
  if (!c->metric) { HIPCHK(hipMalloc(&c->metric, 2 * sizeof(double))); HIPCHK(hipMemsetAsync(c->metric, 0, 2 * sizeof(double), c->st)); }
  if (!c->adam_dev) HIPCHK(hipMalloc(&c->adam_dev, sizeof(AdamDev)));
  if (!c->adam_host) HIPCHK(hipHostMalloc(&c->adam_host, sizeof(AdamDev)));
  HIPCHK(hipStreamSynchronize(c->st));
  // (ensure_packed above did every first-use initialisation eagerly; the RECORDED sequence must start with the packing of whatever
  // weights the previous replay left behind)
  c->packed = false; c->packed32 = false; c->packed_p32 = false;
  HIPCHK(hipStreamBeginCapture(c->st, hipStreamCaptureModeRelaxed));
  c->capturing = true; c->cap_steps = 0; c->cap_step0 = c->step;
  return NIF_OK;
}
extern "C" int nif_graph_end(nif_ctx* c, int32_t* graph_id) {
  if (!c || !graph_id) return fail(NIF_ERR_INVALID, "null");
  if (!c->capturing) return fail(NIF_ERR_STATE, "nif_graph_end without nif_graph_begin");
  c->capturing = false;
  c->step = c->cap_step0;                  // nothing has run yet: the recorded steps count when the graph is launched
  c->packed = false; c->packed32 = false; c->packed_p32 = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(c->st, &g);
  if (e != hipSuccess || !g) { (void)hipGetLastError(); return fail(NIF_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(NIF_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
  c->graphs.push_back(ex); c->graph_steps.push_back(c->cap_steps);
  *graph_id = (int32_t)c->graphs.size() - 1;
  return NIF_OK;
}
extern "C" int nif_graph_launch(nif_ctx* c, int32_t graph_id, const nif_adam* opt) {
  if (!c || !opt || graph_id < 0 || graph_id >= (int32_t)c->graphs.size() || !c->graphs[graph_id]) return fail(NIF_ERR_INVALID, "bad argument");
  if (c->capturing) return fail(NIF_ERR_STATE, "nif_graph_launch while capturing");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));      // (the pinned staging struct is reused: the previous launch's copy must be through)
  c->adam_host->lr = opt->lr; c->adam_host->beta1 = opt->beta1; c->adam_host->beta2 = opt->beta2; c->adam_host->eps = opt->eps;
  c->adam_host->step = c->step;
  HIPCHK(hipMemcpyAsync(c->adam_dev, c->adam_host, sizeof(AdamDev), hipMemcpyHostToDevice, c->st));
  HIPCHK(hipGraphLaunch(c->graphs[graph_id], c->st));
  c->step += c->graph_steps[graph_id];
  c->packed = false; c->packed32 = false; c->packed_p32 = false; c->reg_applied = false;
  return NIF_OK;
}
extern "C" int nif_graph_destroy(nif_ctx* c, int32_t graph_id) {
  if (!c || graph_id < 0 || graph_id >= (int32_t)c->graphs.size()) return fail(NIF_ERR_INVALID, "bad argument");
  if (c->graphs[graph_id]) { HIPCHK(hipStreamSynchronize(c->st)); (void)hipGraphExecDestroy(c->graphs[graph_id]); c->graphs[graph_id] = nullptr; }
  return NIF_OK;
}

// A rank whose shard has no rows left for a step still joins the collective: with a zero [grad | loss] buffer.
extern "C" int nif_zero_grad(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  { const int rcf = metric_flush(c); if (rcf) return rcf; }      // (a deferred loss-metric accumulation reads grad[P]: before it is cleared)
  c->last_step_small = false;
  HIPCHK(hipMemsetAsync(c->grad, 0, sizeof(float) * (size_t)(c->P + 1), c->st));
  c->reg_applied = false;
  return NIF_OK;
}

// A/B switches (measurement and tests; the defaults are the product path)
// PCI bus id ("0000:c1:00.0") of HIP device `dev`: the host side looks up the device's NUMA node under /sys/bus/pci/devices/ and
// pins the rank's process to that node's cores (the reference leaves placement to TensorFlow's runtime)
extern "C" int nif_device_pci_bus_id(int32_t dev, char* out, int32_t cap) {
  if (!out || cap < 16) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipDeviceGetPCIBusId(out, cap, dev));
  return NIF_OK;
}

extern "C" int nif_set_option(nif_ctx* c, const char* key, int32_t value) {
  if (!c || !key) return fail(NIF_ERR_INVALID, "null");
  if (strcmp(key, "fuse_gw") == 0) { c->opt_fuse_gw = value != 0; return NIF_OK; }   // 0: k_snet4 + k_gw_* instead of the fused-gradient kernel
  if (strcmp(key, "fuse_tail") == 0) { c->opt_fuse_tail = value != 0; return NIF_OK; }     // 0: the row reduction runs inside nif_loss_grad_dev (A/B, tests)
  if (strcmp(key, "small_step") == 0) { c->opt_small_step = value != 0; return NIF_OK; }   // 0: small batches on the tile kernels too (A/B, tests)
  if (strcmp(key, "fp32_mfma") == 0) {      // 1: every product on the f32-input MFMAs (k_snet3) instead of the bf16 splits
    c->opt_fp32_mfma = value != 0;
    c->packed = false; c->packed32 = false; c->packed_p32 = false;
    return NIF_OK;
  }
  if (strcmp(key, "side_pnet") == 0) { c->opt_side_pnet = value != 0; return NIF_OK; }   // ParameterNet adjoint on the second stream
  if (strcmp(key, "pipe_chunk") == 0) { c->opt_pipe_chunk = value; return NIF_OK; }   // points per chunk, 0 = off, -1 = default
  if (strcmp(key, "pipe_wgs") == 0) { c->opt_pipe_wgs = value; return NIF_OK; }       // workgroups of the fused kernel per chunk
  return fail(NIF_ERR_INVALID, std::string("unknown option ") + key);
}

// The weight-regulariser term is added (once) and [grad | loss] copied to the host: the read-out half of
// nif_loss_and_grad for callers that keep the dataset resident (L-BFGS closure).  Either pointer may be NULL.
extern "C" int nif_grad_read(nif_ctx* c, float* loss, float* grad) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  apply_reg(c);
  if (grad) HIPCHK(hipMemcpyAsync(grad, c->grad, sizeof(float) * (size_t)c->P, hipMemcpyDeviceToHost, c->st));
  if (loss) HIPCHK(hipMemcpyAsync(loss, c->grad + c->P, sizeof(float), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

extern "C" int nif_last_loss(nif_ctx* c, float* loss) {
  if (!c || !loss) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipMemcpyAsync(loss, c->grad + c->P, sizeof(float), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

static int stage_batch(nif_ctx* c, const float* xin, const float* y, const float* sw, long B) {
  HIPCHK(hipStreamSynchronize(c->st));
  int rc = stage(c, &c->d_a, &c->cap_a, xin, B * (c->pi + c->si)); if (rc) return rc;
  rc = stage(c, &c->d_b, &c->cap_b, y, B * c->so); if (rc) return rc;
  if (sw) { rc = stage(c, &c->d_c, &c->cap_c, sw, B); if (rc) return rc; }
  return NIF_OK;
}
int nif_stage_batch(nif_ctx* c, const float* xin, const float* y, const float* sw, int64_t B, float** dx, float** dy, float** dsw) {
  HIPCHK(hipSetDevice(c->dev));
  int rc = stage_batch(c, xin, y, sw, B); if (rc) return rc;
  *dx = c->d_a; *dy = c->d_b; *dsw = sw ? c->d_c : nullptr;
  return NIF_OK;
}

extern "C" int nif_loss_and_grad(nif_ctx* c, const float* xin, const float* y, const float* sw, int64_t B, float* loss,
                                 float* grad) {
  if (!c || !xin || !y || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = stage_batch(c, xin, y, sw, B); if (rc) return rc;
  rc = nif_loss_grad_dev(c, c->d_a, c->d_b, sw ? c->d_c : nullptr, B, B); if (rc) return rc;
  TAIL_FLUSH(c)
  apply_reg(c);
  if (grad) HIPCHK(hipMemcpyAsync(grad, c->grad, sizeof(float) * (size_t)c->P, hipMemcpyDeviceToHost, c->st));
  if (loss) HIPCHK(hipMemcpyAsync(loss, c->grad + c->P, sizeof(float), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

extern "C" int nif_train_step(nif_ctx* c, const float* xin, const float* y, const float* sw, int64_t B,
                              const nif_adam* opt, float* loss) {
  if (!c || !xin || !y || !opt || B <= 0) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = stage_batch(c, xin, y, sw, B); if (rc) return rc;
  rc = nif_loss_grad_dev(c, c->d_a, c->d_b, sw ? c->d_c : nullptr, B, B); if (rc) return rc;
  rc = nif_adam_step_dev(c, opt); if (rc) return rc;
  if (loss) HIPCHK(hipMemcpyAsync(loss, c->grad + c->P, sizeof(float), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

// ------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------
static int drain_profile(nif_ctx* c) {
  HIPCHK(hipStreamSynchronize(c->st));
  for (auto& r : c->recs) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    c->prof_ms[r.id] += ms; c->prof_cnt[r.id] += 1;
    c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b);
  }
  c->recs.clear();
  return NIF_OK;
}
extern "C" int nif_profile_enable(nif_ctx* c, int on) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = drain_profile(c); if (rc) return rc;
  c->prof_on = on != 0;
  return NIF_OK;
}
extern "C" int nif_profile_read(nif_ctx* c, float* ms, int64_t* cnt, int n, int reset) {
  if (!c || !ms || !cnt || n < NIF_PROF_N) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  int rc = drain_profile(c); if (rc) return rc;
  for (int i = 0; i < NIF_PROF_N; ++i) { ms[i] = (float)c->prof_ms[i]; cnt[i] = c->prof_cnt[i]; }
  if (reset) for (int i = 0; i < NIF_PROF_N; ++i) { c->prof_ms[i] = 0; c->prof_cnt[i] = 0; }
  return NIF_OK;
}
extern "C" int nif_debug_timeline(nif_ctx* c, int64_t* out, int32_t n_pairs) {
  if (!c || n_pairs < 0 || n_pairs > 2048) return fail(NIF_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipStreamSynchronize(c->st));
  if (!c->tl) {
    HIPCHK(hipMalloc(&c->tl, 4096 * sizeof(long long)));
    HIPCHK(hipMemset(c->tl, 0, 4096 * sizeof(long long)));
    return NIF_OK;   // first call arms the buffer
  }
  if (out && n_pairs > 0) HIPCHK(hipMemcpy(out, c->tl, sizeof(long long) * 2 * n_pairs, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(c->tl, 0, 4096 * sizeof(long long)));
  return NIF_OK;
}
extern "C" int nif_timer_start(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  if (!c->t0) { HIPCHK(hipEventCreate(&c->t0)); HIPCHK(hipEventCreate(&c->t1)); }
  HIPCHK(hipEventRecord(c->t0, c->st));
  return NIF_OK;
}
extern "C" int nif_timer_stop(nif_ctx* c, float* ms) {
  if (!c || !ms || !c->t0) return fail(NIF_ERR_INVALID, "timer not started");
  HIPCHK(hipSetDevice(c->dev));
  TAIL_FLUSH(c)
  HIPCHK(hipEventRecord(c->t1, c->st));
  HIPCHK(hipEventSynchronize(c->t1));
  HIPCHK(hipEventElapsedTime(ms, c->t0, c->t1));
  return NIF_OK;
}
