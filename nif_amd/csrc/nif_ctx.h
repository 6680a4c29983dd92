// nif_ctx.h -- the context object behind the opaque nif_ctx* of include/nif_hip.h, shared by the translation units
// that implement the C-ABI (nif_api.hip: orchestration; nif_comm.hip: RCCL).
#pragma once
#include "../../include/nif_hip.h"
#include "nif_internal.h"
#include <string>
#include <vector>

int nif_fail(int code, const std::string& msg);   // sets the thread-local message behind nif_last_error()
static inline int fail(int code, const std::string& msg) { return nif_fail(code, msg); }
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (void)hipGetLastError(); /* HIP >= 7 keeps the last failure until it is read: drain it, or the next launch check of a   \
                                  caller that recovered (fit()'s host-shuffle fallback after a failed hipMalloc) reports it */ \
      return fail(NIF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    }                                                                                             \
  } while (0)

struct nif_ctx {
  nif_cfg cfg;
  int dev = 0;
  hipStream_t st = nullptr;
  // derived sizes
  int kind, pi, si, so, n, L, nst, lst, r, nh, nm, NB, NSTB;
  long po, P;
  std::vector<nif_tensor_desc> layout;
  // theta offsets
  long first_w, first_b, hid_w[NIF_MAX_HID], hid_b[NIF_MAX_HID], hid_w2[NIF_MAX_HID], hid_b2[NIF_MAX_HID];
  long bott_w, bott_b, last_w, last_b;
  // last-layer class: shared-weight SIREN ShapeNet (model.py:1147-1217) + last_layer_bias
  long s_first_w = 0, s_first_b = 0, s_hid_w[NIF_MAX_HID], s_hid_b[NIF_MAX_HID], s_hid_w2[NIF_MAX_HID], s_hid_b2[NIF_MAX_HID];
  long s_bott_w = 0, s_bott_b = 0, ll_bias = 0;
  int RB = 1;   // ZL rows per tile = 32*RB
  // device state
  float *theta = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr;
  long step = 0;
  bool have_params = false, packed = false, packed32 = false, packed_p32 = false, use_snet3 = false, use_snet4 = false;
  bool jac_ok = false;        // JacobianLayer / HessianLayer kernels take this shape (jac_supported)
  void *sWF4 = nullptr, *sWB4 = nullptr;   // bf16-split planes of the hidden hyper-matrices (k_snet4)
  void *sWF4x = nullptr, *sWB4x = nullptr; // k_snet6 (r5): exact-product HALF (hi, lo) planes of the hidden hyper-matrices (k_pack16b mode 3)
  float* sWscale = nullptr;                //   and their powers of two [matrix][plane]
  void *sWF4h = nullptr, *sWB4h = nullptr; // the policies' compact plane set (one bf16 / half plane per block: k_snet4 / k_snet6<.., PR>)
  bool use_ll4 = false;                    // last-layer class: dense ShapeNet on k_snet4
  float* ll_slots = nullptr;               // its parameters in k_snet4's slot order (launch_ll_slots)
  void *ll_wpf = nullptr, *ll_wpb = nullptr;   // phi layer as bf16-split MFMA operands (launch_pack_phi)
  f32x4 *pWF = nullptr, *pWB = nullptr, *sWF = nullptr, *sWB = nullptr, *lWF = nullptr, *lWB = nullptr;
  // workspaces (capacity in points)
  long cap = 0;
  float *stash_s = nullptr, *stash_p = nullptr, *Z = nullptr, *DZ = nullptr, *DU = nullptr, *ZL = nullptr;
  long slot_s = 0, slot_p = 0;
  float* partial = nullptr; int rows_cap = 0; long pstride = 0;
  float* loss_partial = nullptr; long nloss_cap = 0;
  float* dring = nullptr; long dring_cap = 0;
  long long* tl = nullptr;   // timeline stamps (measurement builds)
  float reg_l1 = 0.f, reg_l2 = 0.f; long reg_lo = 0, reg_hi = 0; bool reg_applied = false;
  float sreg_l1 = 0.f, sreg_l2 = 0.f;   // last-layer class: cfg_shape_net l1_reg / l2_reg over the shared ShapeNet's kernels and biases [s_first_w, ll_bias)
  double* metric = nullptr;  // device {sum, count}
  // activity regulariser of the ParameterNet output (nif_set_activity_regularizer): L2 wins over L1 like in the reference
  bool ll_packed32 = false;     // last-layer class under k_sob at n > 96: f32-input MFMA planes of the shared hidden matrices in sWF / sWB
  float* zt_par = nullptr; long zt_par_cap = 0; float* dzt_par = nullptr; long dzt_par_cap = 0;
  float* dat_par = nullptr; long dat_par_cap = 0; float* ztl_par = nullptr; long ztl_par_cap = 0;   // last-layer class: dL/da', z' in latent-row layout   // Sobolev with parameter seeds: dz/dp, dL/d(dz/dp)
  float jac_l1 = 0.f; float* jac_mu = nullptr; long jac_mu_cap = 0; float* jac_tmp = nullptr;   // latent Jacobian regulariser (k_pjac)
  // captured training steps (nif_graph_*): hipGraph executables, the steps each one carries, the device-side Adam state
  std::vector<hipGraphExec_t> graphs; std::vector<int> graph_steps; bool capturing = false; int cap_steps = 0; long cap_step0 = 0;
  AdamDev* adam_dev = nullptr; AdamDev* adam_host = nullptr;
  bool ll_mlp_packed = false;        // last-layer class: the f32 planes of the 32-point MLP kernels are current
  int loss_kind = 0;                 // NIF_LOSS_* (nif_set_loss)
  float* sob_acc = nullptr;          // [grad | loss] summed over the column groups of a Sobolev step with more than three x_index columns
  float act_l1 = 0.f, act_l2 = 0.f; float* act_part = nullptr; long act_part_cap = 0; float* act_loss = nullptr; long act_loss_cap = 0;
  float *stash_l = nullptr, *PHI = nullptr, *DPHI = nullptr, *DA = nullptr, *DZL = nullptr; long slot_l = 0;
  // profiling: (group id, start, stop) event triples recorded on st
  bool prof_on = false;
  std::vector<hipEvent_t> ev_pool;
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<Rec> recs;
  double prof_ms[NIF_PROF_N] = {0};
  long prof_cnt[NIF_PROF_N] = {0};
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // staging for the host-pointer API
  float *d_a = nullptr, *d_b = nullptr, *d_c = nullptr, *d_d = nullptr;
  long cap_a = 0, cap_b = 0, cap_c = 0, cap_d = 0;
  // RCCL communicator of this context (nif_comm.hip): one rank = one ctx = one GPU
  void* comm = nullptr; int comm_rank = 0, comm_world = 1;
  float* comm_scratch = nullptr;   // 64 B device scratch for barrier()
  // two-stream chunk pipeline of the training step (nif_api.hip: loss_grad_core)
  hipStream_t st2 = nullptr; hipEvent_t ev_start = nullptr, ev_done = nullptr; std::vector<hipEvent_t> ev_chunk;
  float* chunk_grad = nullptr; int chunk_cap = 0;      // [chunks][pstride]: per-chunk gradient | loss rows
  bool opt_side_pnet = false;                          // whole-batch step: ParameterNet adjoint on st2 next to the gradient reductions
  long opt_pipe_chunk = -1; int opt_pipe_wgs = 512;    // points per chunk (-1 default, 0 off); fused-kernel workgroups per chunk
  // shard streaming (nif_h2d_async): a copy stream and, per staging slot, 'copy landed' / 'slot consumed' events
  hipStream_t st_copy = nullptr; hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  bool opt_fp32_mfma = false;      // nif_set_option("fp32_mfma"): A/B switch, default from NIF_FP32_MFMA
  int* small_idx = nullptr; int* small_desc = nullptr;     // k_small's tables (offsets only), built at the first small step
  bool metric_pending = false; float metric_pending_w = 0.f;   // a nif_metric_accumulate deferred into the next k_small launch
  bool last_step_small = false;
  // r6: the row reduction of a plain step may wait for its consumer -- nif_adam_step_dev then runs it fused with the update (one launch
  // less per step); every other entry point of the library runs it first (tail_flush).  nif_set_option("fuse_tail") / NIF_FUSE_TAIL
  bool tail_pending = false; int tail_rows = 0, tail_nloss = 0; bool opt_fuse_tail = true;
  bool opt_small_step = true;      // nif_set_option("small_step"): batches <= NIF_SMALL_MAX_B points of a net k_small takes run on it (one launch for loss + gradient); default from NIF_SMALL_STEP
  bool opt_fuse_gw = true;         // nif_set_option("fuse_gw"): ShapeNet weight gradients inside the training kernel (k_snet6) where it has the shape; default from NIF_FUSE_GW
};

// RAII-ish helper: records an event pair around a kernel group when profiling is on
struct ProfScope {
  nif_ctx* c; int id; hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  ProfScope(nif_ctx* c_, int id_, hipStream_t s_ = nullptr) : c(c_), id(id_), s(s_ ? s_ : c_->st) {
    if (!c->prof_on) return;
    auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    a = get(); b = get();
    (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, s);
    c->recs.push_back({id, a, b});
  }
};

// host batch -> this context's staging buffers (asynchronous H2D on c->st); used by the host-pointer entry points
int nif_stage_batch(nif_ctx* c, const float* xin, const float* y, const float* sw, int64_t B, float** dx, float** dy, float** dsw);
// the deferred row reduction of the last plain step (nif_ctx::tail_pending), run before anything but nif_adam_step_dev touches
// [grad | loss], the partial rows or the weights (nif_api.hip)
int nif_tail_flush(nif_ctx* c);
#define TAIL_FLUSH(c_) { const int rct_ = nif_tail_flush(c_); if (rct_) return rct_; }
