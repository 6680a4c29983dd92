/*
 * nif_hip.h -- C-ABI of libnif_hip.so: the MI355X (gfx950) hot path of pswpswpsw/nif.
 *
 * The reference (pure Python on TensorFlow 2.11) exposes NO plugin / FFI interface; its boundary
 * is the Python object surface of nif/model.py.  This header defines the C-ABI underneath the
 * drop-in Python surface `nif_amd.NIF / NIFMultiScale / NIFMultiScaleLastLayerParameterized`.
 * Every entry point names the reference interface it replaces (file:line under the reference
 * tree).  Plain C types only: no torch / numpy / HIP types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a negative nif_status on failure; the message is
 *     available (thread-local) from nif_last_error().
 *   - "host" pointers are caller-owned, C-contiguous float32; "dev" pointers are device memory
 *     obtained from nif_dev_alloc() (or any hipMalloc'ed pointer on ctx's device).
 *   - inputs are rows [t, mu..., x...]: parameter columns first, then coordinates
 *     (nif/model.py:142-143); targets are [B, so]; sample_weight is [B] or NULL.
 *   - all device work is enqueued on the context's HIP stream; *_dev calls are asynchronous,
 *     host-pointer calls synchronise before returning.
 *   - there is NO CPU fallback: without a gfx950 device nif_create fails.
 */
#ifndef NIF_HIP_H
#define NIF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIF_ABI_VERSION 2

typedef enum {
  NIF_OK = 0,
  NIF_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  NIF_ERR_HIP = -2,       /* HIP runtime error (message has hipGetErrorString) */
  NIF_ERR_NODEVICE = -3,  /* no gfx950 device */
  NIF_ERR_STATE = -4,     /* call order (e.g. train step before params are set) */
  NIF_ERR_COMM = -5       /* RCCL error (message has ncclGetErrorString) */
} nif_status;

/* model classes of nif/model.py */
typedef enum {
  NIF_KIND_NIF = 0,         /* class NIF                                   model.py:48   */
  NIF_KIND_MULTISCALE = 1,  /* class NIFMultiScale                         model.py:483  */
  NIF_KIND_LASTLAYER = 2    /* class NIFMultiScaleLastLayerParameterized   model.py:989  */
} nif_kind;

/* activations (`keras.activations.get(name)` model.py:303, mlp.py:40; 'sine' = SIREN) */
typedef enum {
  NIF_ACT_LINEAR = 0, NIF_ACT_SINE = 1, NIF_ACT_SWISH = 2, NIF_ACT_TANH = 3, NIF_ACT_RELU = 4,
  NIF_ACT_SIGMOID = 5, NIF_ACT_ELU = 6, NIF_ACT_SOFTPLUS = 7, NIF_ACT_GELU = 8,
  NIF_ACT_SELU = 9, NIF_ACT_SOFTSIGN = 10, NIF_ACT_EXPONENTIAL = 11, NIF_ACT_HARD_SIGMOID = 12   /* (r4; nif_create rejects ids beyond) */
} nif_act;

/* `mixed_policy` of the model constructors (tf.keras.mixed_precision.Policy, model.py:101-105).  Variables are fp32 under
 * both.  NIF_POLICY_MIXED_BF16: the operands of the ShapeNet's hidden n x n products (activations, hyper-planes, dL/da in
 * the data adjoint) are rounded to bfloat16, accumulation in fp32; biases, activations, first/last layer, loss, the
 * ParameterNet and the weight-gradient sums stay fp32 (strictly more accurate than Keras' policy, which also stores
 * every layer output in bf16).  Nets of 17..32 / 49..64 units (and 113..128 with at most two planes per layer) also keep the hidden layers' dL/da stash rows in bf16 and form
 * the weight-gradient sums of those layers as one bf16 product.  Kernels without a bf16 path (odd 16-feature block
 * counts, 128-wide Sobolev) keep fp32.
 * NIF_POLICY_MIXED_F16 (r4, Keras' 'mixed_float16'): the same rounding points with IEEE half-precision operands on
 * v_mfma_f32_16x16x32_f16 (RNE, saturated at 65504), in the plain training step and the forward pass of every shape k_snet4
 * takes (even 16-feature block counts up to 128 units, all three classes); dL/da enters the adjoint products as
 * half(s dL/da) and the chain is scaled back, with s the power of two that brings the POINT's largest |dL/da| into
 * [2^14, 2^15) -- in place of the single dynamic loss scale Keras' compile() puts around the optimizer under this policy
 * (LossScaleOptimizer): no overflow, no skipped step.  The dL/da stash rows and
 * every weight-gradient sum stay fp32; Sobolev / Jacobian / Hessian kernels and shapes outside k_snet4 run the exact products. */
typedef enum { NIF_POLICY_FLOAT32 = 0, NIF_POLICY_MIXED_BF16 = 1, NIF_POLICY_MIXED_F16 = 2 } nif_policy;

/* What NIF.__init__ (model.py:73-128) / NIFMultiScale._initialize_pnet (model.py:541-736)
 * derive from cfg_shape_net / cfg_parameter_net. */
typedef struct {
  int32_t abi_version;   /* = NIF_ABI_VERSION */
  int32_t kind;          /* nif_kind */
  int32_t pi_dim;        /* cfg_parameter_net["input_dim"]   model.py:88 */
  int32_t si_dim;        /* cfg_shape_net["input_dim"]       model.py:84 */
  int32_t so_dim;        /* cfg_shape_net["output_dim"]      model.py:85 */
  int32_t n_sx;          /* cfg_shape_net["units"]           model.py:86 */
  int32_t l_sx;          /* cfg_shape_net["nlayers"]         model.py:87 */
  int32_t n_st;          /* cfg_parameter_net["units"]       model.py:90 */
  int32_t l_st;          /* cfg_parameter_net["nlayers"]     model.py:91 */
  int32_t latent_dim;    /* cfg_parameter_net["latent_dim"]  model.py:89 */
  int32_t s_act;         /* NIF: cfg_shape_net["activation"]; MultiScale: NIF_ACT_SINE */
  int32_t s_resblock;    /* cfg_shape_net["use_resblock"]    model.py:532 */
  float   s_omega0;      /* cfg_shape_net["omega_0"]         model.py:533 (1.0 for NIF) */
  int32_t p_act;         /* cfg_parameter_net["activation"]; NIF_ACT_SINE => SIREN pnet (model.py:591) */
  int32_t p_resblock;    /* cfg_parameter_net["use_resblock"] model.py:609,681 */
  float   p_omega0;      /* cfg_parameter_net["omega_0"]     model.py:599 */
  int32_t mixed_policy;  /* nif_policy: the `mixed_policy` argument of NIF(...)           model.py:73,101-105 */
  int32_t reserved[7];   /* must be zero */
} nif_cfg;

/* One trainable tensor in Keras variable order. */
typedef struct {
  char    name[48];
  int64_t offset;        /* into the flat parameter vector */
  int32_t rows, cols;    /* cols == 0 for vectors */
} nif_tensor_desc;

/* Keras-2.11 Adam hyper-parameters (README.md:33 `compile(optimizer, loss='mse')`). */
typedef struct {
  float lr, beta1, beta2, eps;
} nif_adam;

typedef struct nif_ctx nif_ctx;

const char* nif_last_error(void);
int nif_abi_version(void);
/* number of visible HIP devices (0 if none / no driver) */
int nif_device_count(void);

/* ---- lifetime ------------------------------------------------------------------------- */
/* replaces: NIF(cfg_shape_net, cfg_parameter_net, mixed_policy) + .build()  model.py:73,345 */
int nif_create(const nif_cfg* cfg, int device_id, nif_ctx** out);
int nif_destroy(nif_ctx* ctx);

/* ---- parameters (model.get_weights / set_weights / trainable_variables, README.md:179-195) */
int nif_param_count(nif_ctx* ctx, int64_t* n_params);
int nif_po_dim(nif_ctx* ctx, int64_t* po_dim);                 /* model.py:169-173,:569-587 */
int nif_param_layout(nif_ctx* ctx, nif_tensor_desc* descs, int32_t* n_inout);
int nif_set_params(nif_ctx* ctx, const float* host, int64_t n);
int nif_get_params(nif_ctx* ctx, float* host, int64_t n);
/* optimizer slots (Adam m, v, and step count) for checkpoint/resume */
int nif_get_opt_state(nif_ctx* ctx, float* m_host, float* v_host, int64_t n, int64_t* step);
int nif_set_opt_state(nif_ctx* ctx, const float* m_host, const float* v_host, int64_t n, int64_t step);

/* ---- device memory / stream plumbing (no reference counterpart; TF owned these) -------- */
int nif_dev_alloc(nif_ctx* ctx, int64_t bytes, void** dptr);
int nif_dev_free(nif_ctx* ctx, void* dptr);
int nif_h2d(nif_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int nif_d2h(nif_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
int nif_sync(nif_ctx* ctx);
/* shard streaming for tables larger than HBM (replaces the TFRecord meta-dataset of nif/data/tfr_dataset.py:85-163, whose
 * tf.data pipeline prefetches the next file while the current one trains): pinned host staging buffers, asynchronous
 * H2D on a copy stream, two staging slots.  nif_h2d_async(slot) first waits (on the device) until the steps that read
 * the slot's device buffers have finished (nif_copy_release), then copies; nif_copy_acquire makes the compute stream
 * wait for the slot's copies; nif_copy_wait_host blocks the host until the slot's pinned buffers may be refilled. */
int nif_host_alloc(nif_ctx* ctx, int64_t bytes, void** hptr);
int nif_host_free(nif_ctx* ctx, void* hptr);
int nif_h2d_async(nif_ctx* ctx, void* dst_dev, const void* src_pinned, int64_t bytes, int32_t slot);
int nif_copy_acquire(nif_ctx* ctx, int32_t slot);
int nif_copy_release(nif_ctx* ctx, int32_t slot);
int nif_copy_wait_host(nif_ctx* ctx, int32_t slot);
/* device-side shuffle of a resident table (Keras fit(shuffle=True) / tf.data .shuffle, tfr_dataset.py:104-108):
 * dst[i][:] = src[perm[i]][:] for n rows of ncol floats; perm is a device int32 array */
int nif_gather_rows_dev(nif_ctx* ctx, const float* src_dev, const int32_t* perm_dev, int64_t n, int32_t ncol, float* dst_dev);
void* nif_stream(nif_ctx* ctx);          /* hipStream_t of the context */
void* nif_grad_dev(nif_ctx* ctx);        /* device float[P+1]: flat gradient || loss  (the RCCL all-reduce buffer) */
void* nif_params_dev(nif_ctx* ctx);      /* device float[P] */

/* ---- inference ------------------------------------------------------------------------ */
/* model.predict / model(x): NIF.call model.py:130-154, NIFMultiScale.call :510-539,
 * NIFMultiScaleLastLayerParameterized.call :1044-1068.   xin [B, pi+si] -> u [B, so] */
int nif_forward(nif_ctx* ctx, const float* xin_host, int64_t B, float* u_host);
int nif_forward_dev(nif_ctx* ctx, const float* xin_dev, int64_t B, float* u_dev);
/* model_p_to_lr().predict(p): model.py:406-420 (last-layer class: :1070-1083).  p [B,pi] -> [B,r] */
int nif_pnet_latent(nif_ctx* ctx, const float* p_host, int64_t B, float* lr_host);
/* model_x_to_phi().predict(x) of the last-layer class: model.py:1085-1104.  x [B,si] -> phi [B,so,r] */
int nif_x_to_phi(nif_ctx* ctx, const float* x_host, int64_t B, float* phi_host);
/* model_lr_to_w().predict(lr): model.py:422-433 == last pnet layer (siren.py:514-522).
 * lr [B,r] -> w [B,po].  NIF_ERR_INVALID for the last-layer class (model.py:1106-1115). */
int nif_latent_to_w(nif_ctx* ctx, const float* lr_host, int64_t B, float* w_host);
int nif_latent_to_w_dev(nif_ctx* ctx, const float* lr_dev, int64_t B, float* w_dev);
/* model_x_to_u_given_w().predict([x, w]): model.py:435-464 / :956-986 -- the per-sample
 * batched matvec  einsum('ai,aij->aj')  (mlp.py:209-219).  x [B,si], w [B,po] -> u [B,so] */
int nif_shapenet_given_w(nif_ctx* ctx, const float* x_host, const float* w_host, int64_t B, float* u_host);
int nif_shapenet_given_w_dev(nif_ctx* ctx, const float* x_dev, const float* w_dev, int64_t B, float* u_dev);

/* JacobianLayer(model, y_index, x_index)(x): nif/layers/gradient.py:36-49, :207-231.
 * y_out [B, so] (all outputs, like the reference) and dydx_out [B, ny, nx] with
 * dydx[a,i,j] = d y[a, y_idx[i]] / d input[a, x_idx[j]], for any input column (parameters and coordinates)
 * of all three classes; forward-mode tangents (k_jac, k_mlp_jac). */
int nif_jacobian(nif_ctx* ctx, const float* xin_host, int64_t B, const int32_t* y_idx, int32_t ny,
                 const int32_t* x_idx, int32_t nx, float* y_out, float* dydx_out);

/* HessianLayer(model, y_index, x_index)(x): nif/layers/gradient.py:130-180, :234-261.  y_out [B, so], dydx_out [B, ny, nx],
 * d2ydx2_out [B, ny, nx, nx] with d2[a,i,j,k] = d^2 y[a, y_idx[i]] / d input[a, x_idx[j]] d input[a, x_idx[k]]: second-order
 * forward-mode tangents (k_jac<.., HESS>), one launch per column pair.  Any input columns of all three classes: a parameter
 * column brings dz/dp (k_pjac) and, for a pair of them, d2z/dp dp' (k_pjac2) and the second-order product rule of every layer;
 * the last-layer class: second-order tangents of the shared ShapeNet x -> phi contracted with the ParameterNet output and its
 * parameter derivatives. */
int nif_hessian(nif_ctx* ctx, const float* xin_host, int64_t B, const int32_t* y_idx, int32_t ny, const int32_t* x_idx,
                int32_t nx, float* y_out, float* dydx_out, float* d2ydx2_out);
/* The same with device pointers (inputs [B, pi+si] and the three outputs resident in HBM; y_idx / x_idx are host arrays): the
 * row gather and, for the last-layer class, the contraction with the ParameterNet output run in kernels on the context's stream;
 * nothing is copied to the host and no host loop runs over the points (a PDE residual over 10^6 collocation points,
 * gradient.py:234-261).  Asynchronous: nif_sync() before the results are read by anyone else.  ny <= 16. */
int nif_hessian_dev(nif_ctx* ctx, const float* xin_dev, int64_t B, const int32_t* y_idx, int32_t ny, const int32_t* x_idx,
                    int32_t nx, float* y_dev, float* dydx_dev, float* d2ydx2_dev);

/* ---- training ------------------------------------------------------------------------- */
/* Keras train_step body without the update: loss = mse(y, model(x), sample_weight) and
 * d loss / d theta (GradientTape).  Result stays on the device in nif_grad_dev():
 * grad[0..P) and grad[P] = loss, both already scaled by 1/B_global so that a SUM all-reduce
 * over shards gives the global-batch mean (tf.distribute.MirroredStrategy, README.md:39-49). */
int nif_loss_grad_dev(nif_ctx* ctx, const float* xin_dev, const float* y_dev, const float* sw_dev_or_null,
                      int64_t B_local, int64_t B_global);
/* Sobolev training: the Keras model  Model(x, JacobianLayer(nif, y_index=all, x_index)(x))  compiled with
 * loss='mse', loss_weights=[1, w_jac]  (nif/layers/gradient.py:36-49 used as a trainable output; Keras then
 * differentiates THROUGH the Jacobian).  loss = mse(u, y) + w_jac * mse(du/dx, dydx) with
 * dydx [B, so, nx]; x_idx are 1..3 distinct input-vector columns, coordinates AND/OR ParameterNet inputs (gradient.py:207-231
 * takes any column; a parameter column runs its tangent through the ParameterNet, the hyper layer and the product rule of every
 * h W(p): k_pjac + k_sob<PAR> + a second weight-gradient reduction).  Built as forward tangents + their hand-derived adjoint in
 * one kernel (k_sob_dev.h) for NIFMultiScale (with or without resblocks), class NIF (any activation, skip connections) and
 * the last-layer class (a parameter column there is one more contraction of phi with a' = (dz/dp) last_w in the kernel's
 * epilogue).  Shapes whose working set exceeds one CU's LDS return NIF_ERR_INVALID.
 * Same conventions as nif_loss_grad_dev (result in nif_grad_dev(), scaled by 1/B_global). */
int nif_sobolev_loss_grad_dev(nif_ctx* ctx, const float* xin_dev, const float* y_dev, const float* dydx_dev,
                              const float* sw_dev_or_null, int64_t B_local, int64_t B_global, const int32_t* x_idx,
                              int32_t nx, float w_jac);
/* The same step for ANY y_index / x_index of JacobianLayer (nif/layers/gradient.py:207-231): y_idx = NULL means every output,
 * otherwise the derivative term is the mean over the ny listed outputs (distinct) x nx columns; 1 <= nx <= 16 distinct columns --
 * more than three run as passes over groups of three tangent streams whose [grad | loss] add up.  dydx_dev rows stay [so][nx]
 * (entries of unlisted outputs are not part of the loss). */
int nif_sobolev_loss_grad_dev_y(nif_ctx* ctx, const float* xin_dev, const float* y_dev, const float* dydx_dev,
                                const float* sw_dev_or_null, int64_t B_local, int64_t B_global, const int32_t* x_idx,
                                int32_t nx, const int32_t* y_idx_or_null, int32_t ny, float w_jac);
/* predict() of that two-output model: u [B,so] and du/dx [B,so,nx], device pointers */
int nif_sobolev_forward_dev(nif_ctx* ctx, const float* xin_dev, int64_t B, const int32_t* x_idx, int32_t nx, float* u_dev,
                            float* dudx_dev);
/* Captured training steps (hipGraph) for the launch-bound small-batch regime (BASELINE configs[0]: batch 512 = 13 kernels of a few
 * microseconds each per step; Keras' fit() runs such steps from one tf.function graph, README.md:23-37).  Between nif_graph_begin
 * and nif_graph_end the device-side calls on this context are recorded instead of executed -- e.g. all batches of one epoch:
 * nif_loss_grad_dev / nif_sobolev_loss_grad_dev[_y], nif_adam_step_dev, nif_metric_accumulate; nif_graph_launch replays them in one
 * submission (pointers as recorded; Adam's hyper-parameters from `opt`, its iteration count continues from the context's).
 * Workspaces must have been sized by nif_reserve; contexts with a communicator attached are refused. */
int nif_graph_begin(nif_ctx* ctx);
int nif_graph_end(nif_ctx* ctx, int32_t* graph_id_out);
int nif_graph_launch(nif_ctx* ctx, int32_t graph_id, const nif_adam* opt);
int nif_graph_destroy(nif_ctx* ctx, int32_t graph_id);
/* model.compile(loss=...) (README.md:33 passes 'mse'; keras.losses.get resolves any name): the per-element loss of every training /
 * evaluation entry point of the context.  Reduction as Keras': mean over the outputs, sample-weighted sum over the batch / B_global. */
typedef enum { NIF_LOSS_MSE_ = 0, NIF_LOSS_MAE_ = 1, NIF_LOSS_HUBER_ = 2, NIF_LOSS_LOG_COSH_ = 3 } nif_loss;
int nif_set_loss(nif_ctx* ctx, int32_t kind /* nif_loss */);
/* optimizer.apply_gradients with Adam on the (already all-reduced) nif_grad_dev() buffer */
int nif_adam_step_dev(nif_ctx* ctx, const nif_adam* opt);
/* zero [grad | loss]: what a rank contributes to the step's all-reduce when its shard has no rows left (uneven shards of
 * Model.fit under data parallelism); the weight-regulariser term is still added by the following nif_adam_step_dev */
int nif_zero_grad(nif_ctx* ctx);
/* size every workspace of a training step over up to B_max points now (n_tangents = Sobolev seeds, 0 for the plain
 * step), so that later steps never hipMalloc / synchronise (TensorFlow's allocator owned this in the reference) */
int nif_reserve(nif_ctx* ctx, int64_t B_max, int32_t n_tangents);
/* host-pointer conveniences */
int nif_loss_and_grad(nif_ctx* ctx, const float* xin_host, const float* y_host, const float* sw_host_or_null,
                      int64_t B, float* loss_out, float* grad_host);        /* lbfgs.py:66-74 */
int nif_train_step(nif_ctx* ctx, const float* xin_host, const float* y_host, const float* sw_host_or_null,
                   int64_t B, const nif_adam* opt, float* loss_out);        /* Model.fit's train_step */
/* Weight regularisers of cfg_parameter_net["l1_reg"/"l2_reg"] (nif/model.py:109-117: L2(l2) or else L1(l1) on
 * every ParameterNet kernel and bias): loss += l2*sum(w^2) + l1*sum(|w|) over theta[lo, hi); the gradient
 * term is added once, after the cross-rank all-reduce, inside nif_adam_step_dev / nif_loss_and_grad. */
int nif_set_regularizer(nif_ctx* ctx, float l1, float l2, int64_t lo, int64_t hi);
/* cfg_shape_net["l1_reg"/"l2_reg"] of NIFMultiScaleLastLayerParameterized (nif/model.py:1028-1039; added by every SIREN /
 * SIREN_ResNet layer of the shared ShapeNet for kernels and biases, nif/layers/siren.py:266-269, :393-398): the same term over
 * the ShapeNet's first / hidden / bottleneck variables (not last_layer_bias).  The caller passes the coefficient the reference
 * ends up with (it reads cfg_parameter_net's number there, model.py:1031-1036).  Other classes: NIF_ERR_INVALID unless 0, 0. */
int nif_set_shapenet_regularizer(nif_ctx* ctx, float l1, float l2);
/* Latent Jacobian regulariser cfg_parameter_net["jac_reg"] (nif/model.py:353-375 wraps the model in JacRegLatentLayer,
 * nif/layers/gradient.py:52-127): loss += l1 * mean_{a,c,d} (d latent_c / d parameter_d)^2, differentiated through the
 * Jacobian: forward tangents of the ParameterNet + their adjoint (k_pjac), weight gradients by the batch GEMM kernels over
 * tangent pseudo-tiles.  All three classes; at most 3 parameter inputs (pi_dim <= 3). */
int nif_set_jac_regularizer(nif_ctx* ctx, float l1);
/* Activity regulariser of cfg_parameter_net["act_l1_reg"/"act_l2_reg"] (nif/model.py:118-125: Keras activity_regularizer
 * L2(l2) or else L1(l1) on the last ParameterNet layer, :226, :659, :731): loss += c / B * sum_a sum_i phi(pnet_out[a, i]),
 * phi = (.)^2 or |.|, Keras dividing the activity loss by the batch size.  pnet_out [B, po] is never materialised: two
 * passes recompute it on the fly (k_actreg_*; latent_dim <= 64).  On the last-layer class pnet_out is the small [B, latent_dim]
 * tensor itself: one pass (k_ll_actreg). */
int nif_set_activity_regularizer(nif_ctx* ctx, float l1, float l2);
/* Keras' epoch loss metric without a host sync per batch: sum += weight * grad[P], count += weight (device side) */
int nif_metric_accumulate(nif_ctx* ctx, float weight);
int nif_metric_read(nif_ctx* ctx, double* sum_out, double* count_out, int reset);
/* reads grad[P] (the loss of the last nif_loss_grad_dev) */
int nif_last_loss(nif_ctx* ctx, float* loss_out);
/* read-out half of nif_loss_and_grad for a resident dataset (lbfgs.py:66-74 closure): adds the weight-regulariser term
 * once, copies loss and/or the flat gradient to the host, synchronises.  Either pointer may be NULL. */
int nif_grad_read(nif_ctx* ctx, float* loss_out_or_null, float* grad_host_or_null);
/* A/B switches for measurement and tests (no reference counterpart).  "fp32_mfma" = 1: every product of the
 * ShapeNet on the f32-input MFMAs instead of the exact bf16 splits (default 0, or NIF_FP32_MFMA=1 in the environment);
 * "fuse_gw", "small_step", "fuse_tail" (default 1; NIF_FUSE_GW / NIF_SMALL_STEP / NIF_FUSE_TAIL = 0): the fused-gradient kernel, the
 * one-launch small-batch step, the row reduction deferred to nif_adam_step_dev (which then runs it fused with the update: the
 * [grad | loss] buffer is complete after ANY other call of this library on the context -- nif_grad_dev included -- and after the
 * update; a caller that reads the buffer through a pointer it cached earlier, without such a call, sets "fuse_tail" to 0) */
int nif_set_option(nif_ctx* ctx, const char* key, int32_t value);

/* ---- multi-GPU: RCCL over xGMI, called directly (replaces `tf.distribute.MirroredStrategy().scope()`, reference
 * README.md:39-49: data parallelism over the GPUs of one node).  One nif_ctx per GPU.  The point batch is sharded
 * over ranks; nif_loss_grad_dev pre-scales by 1/B_global; ONE sum all-reduce of the flat float buffer
 * nif_grad_dev() = [grad(P) | loss] per step, enqueued on the context's stream (no host sync, no copy); every rank
 * then applies the identical Adam update (replicated optimizer state, as MirroredStrategy does). */
#define NIF_COMM_ID_BYTES 128
typedef enum { NIF_DT_F32 = 0, NIF_DT_F64 = 1, NIF_DT_I64 = 2 } nif_dtype;
typedef enum { NIF_OP_SUM = 0, NIF_OP_MAX = 1, NIF_OP_MIN = 2 } nif_redop;
/* one process per GPU: rank 0 creates the id (ncclGetUniqueId), the host side carries the 128 bytes to the other
 * processes, every rank joins (ncclCommInitRank on ctx's device; collective: returns when all `world` ranks called) */
int nif_device_pci_bus_id(int32_t device_id, char* out, int32_t capacity);   /* "0000:c1:00.0": NUMA placement of the rank's process */
int nif_comm_unique_id(void* id_out_128_bytes);
int nif_comm_init_rank(nif_ctx* ctx, const void* id_128_bytes, int32_t rank, int32_t world);
/* one process driving n GPUs: ncclCommInitAll over the contexts' devices (rank i = ctxs[i]) */
int nif_comm_init_all(nif_ctx** ctxs, int32_t n);
int nif_comm_destroy(nif_ctx* ctx);                       /* also done by nif_destroy */
int nif_comm_info(nif_ctx* ctx, int32_t* rank_out, int32_t* world_out);   /* (0, 1) without a communicator */
/* THE collective of the training step: ncclAllReduce(sum, f32, P+1) in place on nif_grad_dev(), on ctx's stream.
 * No-op for a context without communicator (world size 1). */
int nif_allreduce_grad(nif_ctx* ctx);
int nif_comm_selftest(nif_ctx* ctx, int32_t* ranks_seen_out);   /* rank + 1 through nif_allreduce_grad's own buffer / stream: sum must be N (N + 1) / 2 */
int nif_allreduce_grad_multi(nif_ctx** ctxs, int32_t n);  /* the n contexts of nif_comm_init_all, one RCCL group call */
/* plumbing: in-place all-reduce of a caller-owned device buffer on ctx's stream (Model.fit: every step's global
 * batch size, agreed once per call; bench.py: max-over-ranks wall time) */
int nif_comm_allreduce(nif_ctx* ctx, void* dev_buf, int64_t count, int32_t dtype /* nif_dtype */, int32_t op /* nif_redop */);
/* every rank reached this call and ctx's stream has drained (one-word all-reduce + stream synchronise) */
int nif_comm_barrier(nif_ctx* ctx);
/* Model.fit's train_step on n GPUs from ONE process: rows split contiguously and evenly over the
 * contexts, per-shard loss/gradient, one grouped all-reduce, identical Adam update everywhere.  Host pointers. */
int nif_train_step_multi(nif_ctx** ctxs, int32_t n, const float* xin_host, const float* y_host,
                         const float* sw_host_or_null, int64_t B, const nif_adam* opt, float* loss_out);

/* ---- measurement (HIP events on the context's stream; no reference counterpart) ------- */
/* kernel groups timed when profiling is on */
typedef enum {
  NIF_PROF_PACK = 0,      /* theta -> MFMA operand order */
  NIF_PROF_PNET_FWD = 1,  /* ParameterNet forward */
  NIF_PROF_SNET = 2,      /* ShapeNet forward + MSE + adjoint (the dominant kernel) */
  NIF_PROF_PNET_BWD = 3,  /* ParameterNet adjoint */
  NIF_PROF_GW = 4,        /* all weight-gradient reductions */
  NIF_PROF_REDUCE = 5,    /* partial rows -> flat gradient */
  NIF_PROF_ADAM = 6,
  NIF_PROF_GIVEN_W = 7,   /* model_x_to_u_given_w kernel */
  NIF_PROF_LATENT_TO_W = 8,
  NIF_PROF_SNET_FWD = 9,  /* ShapeNet forward only (predict) */
  NIF_PROF_N = 10
} nif_prof_id;
int nif_profile_enable(nif_ctx* ctx, int on);
/* synchronises, then adds the elapsed milliseconds / launch counts per group since the last reset */
int nif_profile_read(nif_ctx* ctx, float* ms_out, int64_t* count_out, int n, int reset);
/* measurement builds (-DNIF_TIMELINE): (id, s_memtime) stamps of one wavefront of the dominant kernel.
 * The first call arms the buffer; later calls copy out up to n_pairs pairs and clear it. */
int nif_debug_timeline(nif_ctx* ctx, int64_t* out_pairs, int32_t n_pairs);
/* a plain stopwatch on the stream */
int nif_timer_start(nif_ctx* ctx);
int nif_timer_stop(nif_ctx* ctx, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* NIF_HIP_H */
