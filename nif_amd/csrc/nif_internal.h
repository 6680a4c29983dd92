// nif_internal.h -- shared between the HIP translation units of libnif_hip.so (gfx950 only).
//
// Data layout conventions:
//   * a "tile" is 32 consecutive points; one wavefront owns one tile.  Lane l serves point
//     p = l & 31; the two lane halves hf = l >> 5 hold different features of that point.
//   * an activation tile of width 32*NB lives in registers as f32x16 h[NB]: element v of block b
//     on lane (p, hf) is feature 32*b + fmap(v, hf) of point p.  This is exactly the C/D layout of
//     v_mfma_f32_32x32x2_f32 with features on rows and points on columns, AND a legal B-operand
//     sequence for the next layer's MFMAs (K order permuted consistently in the packed weights),
//     so an MLP chain never leaves registers.
//   * "stash" arrays (activations kept for the adjoint and for the weight-gradient GEMMs) are
//     [tile][feature][32 points] fp32: every wave store/load is two full 128-B lines.
//   * small per-point vectors (latent z, dL/dz, dL/du) use the same [tile][c][32] layout.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NIF_MAX_HID 16      // max pnet hidden layers / snet hidden layers supported
#define NIF_TP 32           // points per tile

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

enum { ACT_LINEAR = 0, ACT_SINE = 1, ACT_SWISH = 2, ACT_TANH = 3, ACT_RELU = 4, ACT_SIGMOID = 5,
       ACT_ELU = 6, ACT_SOFTPLUS = 7, ACT_GELU = 8,
       ACT_SELU = 9, ACT_SOFTSIGN = 10, ACT_EXPONENTIAL = 11, ACT_HARD_SIGMOID = 12 };    // r4: the rest of keras.activations (Keras 2.11) that is elementwise
#define NIF_SELU_ALPHA 1.6732632423543772f
#define NIF_SELU_SCALE 1.0507009873554805f

// Reference to a (possibly hypernetwork-generated) matrix inside the flat parameter / gradient
// vector.  element(k, in, out) lives at (k < r ? base_k + k*kstride : base_last) + in*ld + out.
// A plain dense matrix has r == 0 (only the "last" plane exists).
struct MatRef {
  int r;
  long base_k, kstride, base_last;
  int ld, nin, nout;
};
__host__ __device__ inline long matref_index(const MatRef& m, int k, int in, int out) {
  return (k < m.r ? m.base_k + (long)k * m.kstride : m.base_last) + (long)in * m.ld + out;
}

// ------------------------------------------------------------------------------------------
// ParameterNet (shared-weight MLP) kernels
// ------------------------------------------------------------------------------------------
struct PNetArgs {
  const float* theta;
  const float* xin; int ncol; int col0;   // input rows [B][ncol], pnet reads columns col0..col0+pi
  long B;
  int pi, nst, lst, r;
  int act, res, siren; float omega;
  long first_w, first_b;
  long hid_w[NIF_MAX_HID], hid_b[NIF_MAX_HID], hid_w2[NIF_MAX_HID], hid_b2[NIF_MAX_HID];
  long bott_w, bott_b;
  int ll_kind; long last_w, last_b;       // last-layer class: pnet_out = z @ W[r,r] + b
  int pbf2;                               // k_pnet<2>: hidden products as bf16 splits from LDS planes (set by launch_pnet)
  const f32x4* WF; const f32x4* WB;       // packed MFMA operands, mat m at m*NSTB*NSTB*256 f32x4
  float* stash; long slot_stride;         // slot s at stash + s*slot_stride  (floats)
  float* Z;                               // out: [tiles][r][32]
  float* DZ;                              // in (bwd): [tiles][r][32]
  float* ZL;                              // LL kind: latent before the r x r map [tiles][zl_rows][32]
  int zl_rows;                            //          rows per tile of ZL (r padded to a multiple of 32)
};
// stash slots of the pnet: IN_m (input of hidden matrix m) = m, IN_bott = nm, DA_first = nm+1,
// DA_m = nm+2+m.   nm = lst * (res ? 2 : 1)

// ------------------------------------------------------------------------------------------
// hypernetwork ShapeNet kernels (NIF / NIFMultiScale)
// ------------------------------------------------------------------------------------------
// Keras loss of compile(loss=...) (README.md:33 uses 'mse'; r4: the other regression losses keras.losses.get knows for this surface).
// Per element e = prediction - target: value v(e) and derivative v'(e); the reduction is Keras' for all of them (mean over the last
// axis, sample-weighted sum over the batch / batch size).  kind 0 keeps the mse arithmetic of r1-r3 bit for bit.
enum { NIF_LOSS_MSE = 0, NIF_LOSS_MAE = 1, NIF_LOSS_HUBER = 2, NIF_LOSS_LOGCOSH = 3 };
__device__ __forceinline__ void nif_loss_vd(int kind, float e, float* v, float* d) {
  if (kind == NIF_LOSS_MAE) { *v = fabsf(e); *d = e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.0f); }
  else if (kind == NIF_LOSS_HUBER) {            // keras.losses.huber, delta = 1
    const float a = fabsf(e);
    *v = a <= 1.0f ? 0.5f * e * e : a - 0.5f;
    *d = a <= 1.0f ? e : (e > 0.f ? 1.0f : -1.0f);
  } else {                                      // log_cosh: e + softplus(-2 e) - log 2 (keras.losses.log_cosh), derivative tanh(e)
    const float t = -2.0f * e;
    *v = e + (fmaxf(t, 0.f) + log1pf(expf(-fabsf(t)))) - 0.69314718055994531f;
    *d = tanhf(e);
  }
}
// se += v(e), dfac = v'(e): the squared error inline (fmaf, factor 2 e as before), everything else through nif_loss_vd
#define NIF_LOSS_ACC(KIND_, E_, SE_, DFAC_)                                   \
  float DFAC_;                                                                \
  if ((KIND_) == NIF_LOSS_MSE) { SE_ = fmaf(E_, E_, SE_); DFAC_ = 2.0f * (E_); } \
  else { float v_; nif_loss_vd(KIND_, E_, &v_, &DFAC_); SE_ += v_; }

struct SNetArgs {
  int loss_kind;                          // NIF_LOSS_*
  const float* theta;
  const float* xin; int ncol; int col0;   // coordinates are columns col0..col0+si
  long B;
  int si, so, n, nh, r; long po;
  int act, res, nif_skip; float omega;
  long off_Wh, off_bh;                    // theta offsets of the hyper kernel [r,po] and bias [po]
  const float* Z;                         // [tiles][r][32]
  const f32x4* WF; const f32x4* WB;       // packed, plane (j*(r+1)+k) at *NB*NB*256 f32x4
  float* stash; long slot_stride;         // slots: IN_l (l=1..nh+1) = l-1 ; DA_l (l=0..nh) = nh+1+l
  float* DU;                              // [tiles][so][32]
  float* DZ;                              // [tiles][r][32]
  const float* y; const float* sw;        // [B][so], [B] or null
  float* u_out;                           // [B][so] or null
  float* loss_partial;                    // [gridDim.x]
  float inv_bg;                           // 1 / B_global
  float* dring;                           // k_snet3: per-wave ring for act'(a) (register-dump order)
  int nsm;                                // k_snet3: floats per k of the LDS copy of the small hyper-vectors
  long long* tl;                          // -DNIF_TIMELINE builds: s_memtime stamps of wave 0 of block 0
  const void* WF4; const void* WB4;       // k_snet4: bf16-split planes (k_pack16b), per plane NCH chunks
  const void* WF4h; const void* WB4h;     // the policies' compact plane set (prec != 0): one bf16 / half plane per block (k_pack16b mode 1 / 2)
  const void* WF4x; const void* WB4x;     // k_snet6 (r5): the exact-product HALF planes (hi, lo), k_pack16b mode 3, both in the adjoint geometry
  const float* wscale;                    //   their powers of two [matrix][plane][s | 1 / s] (k_plane_scales)
  // last-layer-parameterised class on k_snet4 (r = 0: shared dense SIREN; theta = the slot-ordered copy built by
  // launch_ll_slots; so = so_u * rl outputs phi): u = Dot(phi, a) + bias, a = Z [tiles][rl][32]
  int ll, rl, so_u;
  const void* WPF; const void* WPB;       // bf16-split planes of the phi layer [n][so <= 32] (launch_pack_phi)
  float* DPHI;                            // [tiles][so][32]   dL/dphi (weight gradient of the phi layer)
  float* DA_ll;                           // [tiles][rl][32]   dL/da
  float* DZL;                             // [tiles][rl][32]   dL/d latent (through the rl x rl map of the ParameterNet)
  int prec;                               // 0: fp32-exact products; 1: mixed_bfloat16 policy (operands of the n x n products rounded to bf16);
                                          // 2: mixed_float16 (k_snet4<.., PR = 2> alone; every other kernel family treats it as 0)
  int da_bf16;                            // prec == 1: the hidden layers' dL/da stash rows in bf16 (the consumer is k_gw_lds<.., DAB>; nif_api decides)
  int h_ph16;                             // prec == 1, 128-wide plain SIREN training: the hidden matrices' INPUT stash rows as 16-bit phases (k_snet3_dev.h;
                                          // the readers are k_snet4's own adjoint sweep and k_gw8<R, true, true>: snet4_writes_h_ph16)
  int wg_cap;                             // k_snet4: at most this many workgroups (0 = fill the device); the chunk pipeline leaves room for stream B
};
// slot-ordered copy of the dense ShapeNet parameters of the last-layer class: [W1 | (hidden: unused) | Wl | b1 | bh_j | bl |
// last_layer_bias | pnet last W (rl x rl)] -- the order k_snet4's prologue indexes (hyp3 with r = 0)
struct LLSlotSeg { long src, dst, len; };
struct LLSlotMap { int nseg; LLSlotSeg seg[48]; };
void launch_ll_slots(const float* theta, const LLSlotMap& m, float* slots, hipStream_t st);
// phi layer (dense [n][sop], sop <= 32) as bf16-split MFMA operands: NBL/2 forward K-step chunks of 2*3*64 16-byte units
// and one adjoint chunk of NBL*2*64 units
long snet4_phi_fwd_elems(int n);
long snet4_phi_bwd_elems(int n);
void launch_pack_phi(const float* theta, long w_off, int n, int sop, void* WPF, void* WPB, hipStream_t st);
// slot offsets inside pnet_output (nif/model.py:253-300): computed on the fly
__host__ __device__ inline long slot_w1(const SNetArgs& a) { return 0; }
__host__ __device__ inline long slot_wh(const SNetArgs& a, int j) { return (long)a.si * a.n + (long)j * a.n * a.n; }
__host__ __device__ inline long slot_wl(const SNetArgs& a) { return (long)a.si * a.n + (long)a.nh * a.n * a.n; }
__host__ __device__ inline long slot_b1(const SNetArgs& a) { return slot_wl(a) + (long)a.n * a.so; }
__host__ __device__ inline long slot_bh(const SNetArgs& a, int j) { return slot_b1(a) + a.n + (long)j * a.n; }
__host__ __device__ inline long slot_bl(const SNetArgs& a) { return slot_b1(a) + a.n + (long)a.nh * a.n; }

// feature rows of a ShapeNet stash tile: the gradient kernels come in 32-, 64- and 128-row forms (nif_ctx::NB = 1, 2, 4), so a
// 65..96-unit layer is stored with 128 rows like a 128-unit one (rows >= 16 * ceil(n / 16) are never written: zero from the
// allocation's memset).  Producers (k_snet3 / k_snet4 / k_sob) and consumers (k_gw*) must agree on this.
__host__ __device__ inline int stash_fp(int n) { return n <= 32 ? 32 : (n <= 64 ? 64 : 128); }

// ------------------------------------------------------------------------------------------
// weight-gradient kernels: C[k][in][out] = scale * sum_p zt_k[p] * IN[p][in] * DA[p][out]
// ------------------------------------------------------------------------------------------
struct GwArgs {
  const float* IN;      // stash [tiles][32*NBI][32]   (mfma / out kernels)
  const float* DA;      // stash [tiles][32*NBO][32]   (mfma / first kernels)
  const float* SM;      // small per-point vectors [tiles][nc][32] (out kernel: dL/dout)
  const float* xin; int ncol; int col0; int nd;   // first kernel: input columns
  int nc;               // out kernel: number of output columns
  const float* Z; int r;    // latent [tiles][r][32] or null (r = 0)
  long ntiles; long B;
  float scale;
  MatRef W;             // where the matrix gradient goes
  MatRef Bv;            // bias gradient (nin = 1, ld = 0): element(k, 0, out)
  float* partial; long pstride;   // partial[row*pstride + index], row = blockIdx.x
  int has_bias;
  // Sobolev training (k_sob.hip): the stashes hold (1+ns) * zt_mod tiles -- the real tiles followed by one
  // block of tangent pseudo-tiles per seed.  Z is indexed by t % zt_mod; tiles >= bias_ntiles carry no bias
  // gradient and (first layer) use the one-hot input e_seed[t / zt_mod - 1].  0 = plain batch (launchers fix up).
  long zt_mod, bias_ntiles;
  int seed[3];
  int in_ph16;          // IN holds 16-bit phase rows (k_snet4<8, .., PR = 1> wrote them: snet4_writes_h_ph16); k_gw8<R, true, true> alone reads that form
  int da_bf16;          // DA holds bf16 rows (mixed_bfloat16: k_snet4<PR> / k_sobw<PR> wrote them); k_gw_lds / k_gw8 (gw_da_bf16_ok)
};
bool gw_da_bf16_ok(int NBI, int NBO, int r);
bool sobw_supported(const SNetArgs& a, int ns, bool any_par);      // k_sobw.hip takes this Sobolev training step

// launchers (implemented in the .hip files); all enqueue on `st`
void launch_pack(const float* theta, const MatRef& m, int NBI, int NBO, f32x4* WF, f32x4* WB, hipStream_t st);
void launch_pnet(const PNetArgs& a, int NSTB, bool train, hipStream_t st);
void launch_pnet_bwd(const PNetArgs& a, int NSTB, hipStream_t st);
void launch_snet(const SNetArgs& a, int NB, bool train, hipStream_t st);
// persistent, LDS-staged, 16-point-tile variant (k_snet3.hip).  launch_snet3 returns the number of
// workgroups (query_only: without launching) and the waves per workgroup, for sizing dring / loss_partial.
int launch_snet3(const SNetArgs& a, bool train, bool query_only, int* waves_out, hipStream_t st);
// ParameterNet adjoint + weight gradients without a stash (k_pnetbw.hip); writes the ParameterNet columns of
// `rows` partial-gradient rows
bool pnet_bwg_supported(const PNetArgs& a);
void launch_pnet_bwg(const PNetArgs& a, float* partial, long pstride, int rows, hipStream_t st, const float* touch = nullptr,
                     long touch_floats = 0);
// bf16-split variant of k_snet3 (k_snet4.hip): fp32-exact 6-product forward, 3-product adjoint on the bf16 MFMA
bool snet4_supported(const SNetArgs& a);
long snet4_fwd_elems(int n, int r);
long snet4_bwd_elems(int n, int r);
// (scale = omega_0 of the layer: the bf16-split planes hold omega_0 M, see k_snet4.hip)
void launch_pack16b(const float* theta, const MatRef& m, int NBL, void* WF, void* WB, float scale, hipStream_t st, int mode = 0);
void launch_pack16b_dual(const float* theta, const MatRef& m0, long mstride, int nmat, int NBL, void* WF, void* WB, long fstride_elems,
                         long bstride_elems, void* WFx, void* WBx, long fxstride_elems, long bxstride_elems, float scale, float* pscale,
                         hipStream_t st);     // mode 0 + mode 3 (scales found in the kernel) in one launch (k_snet4.hip)
void launch_snet4_f16(const SNetArgs& a, bool train, int nblk, size_t shm, hipStream_t st);     // k_snet4_f16.hip
void launch_snet4_x16(const SNetArgs& a, bool train, int nblk, size_t shm, hipStream_t st);     // k_snet4_x16.hip (r5: half-pair exact products)
void launch_pack16b_batch(const float* theta, const MatRef& m0, long mstride, int nmat, int NBL, void* WF, void* WB,
                          long fstride_elems, long bstride_elems, float scale, hipStream_t st, int mode = 0, float* pscale = nullptr);
int launch_snet4(const SNetArgs& a, bool train, bool query_only, hipStream_t st);
// k_small.hip (r6): loss + every gradient of a small batch in one launch (one partial row per 16 points)
#define NIF_SMALL_MAX_B 2048
bool small_supported(const PNetArgs& p, const SNetArgs& s);
int small_rows(long B);
void small_tables(const PNetArgs& p, const SNetArgs& s, std::vector<int>& idx, std::vector<int>& desc);
void launch_small(const PNetArgs& p, const SNetArgs& s, float* partial, long pstride, long P, const int* idx_dev, const int* desc_dev,
                  double* metric, float metric_w, const float* g_loss, hipStream_t st);
bool snet4_writes_h_ph16(const SNetArgs& a);    // ... and its hidden-matrix input rows as 16-bit phases
bool gw_in_ph16_ok(int NBI, int NBO, int r);    // a reader of that form exists for this shape (k_gw8<R, true, true>; NIF_H_PH16=0 switches it off)
bool snet4_writes_da_bf16(const SNetArgs& a);   // stash format the training launch of `a` produces (k_snet4.hip)
bool sob_writes_da_bf16(const SNetArgs& a, int ns, bool any_par);   // same for launch_sob (k_sob.hip)
// k_snet4's plain-SIREN training step with every ShapeNet weight gradient fused in (k_snet6.hip): one partial-gradient row and one
// loss partial per workgroup; no dL/da stash, no k_gw_* launches
bool snet6_supported(const SNetArgs& a);
int snet6_rows(const SNetArgs& a);
int launch_snet6(const SNetArgs& a, float* partial, long pstride, hipStream_t st);
int snet4_nsm_ll(int si, int sop, int nh, int n, int sou, int rl);
// Sobolev step (k_sob.hip): primal + tangents w.r.t. `ns` coordinate seeds, loss mse(u,y) + wj*mse(du/dx,gt), adjoint
long sob_ring_floats_per_wave(int n, int nh);
bool sob_ll_supported(const SNetArgs& a);   // last-layer class under k_sob (k_sob_ll.hip)
// parameter seeds (x_index < pi_dim): stream d is a parameter stream iff par[d] >= 0 (then seeds[d] is unused); gcol[d] = the
// column of gt / ju the stream fills; ZT = dz/dp [pi][tiles][r][32] (launch_pjac_fwd), DZT = dL/d(that) per stream
struct SobPar {
  int par[3]; int gcol[3]; const float* ZT; float* DZT;
  // last-layer class: parameter columns as heads of the epilogue (k_sob_dev.h): column, dydx position, dL/da', padded z'
  int npar; int parc[3]; int pcol[3]; float* DAT; float* ZTL; int zl_rows;
  // r4, all "0 = as before": the launch carries one GROUP of the x_index columns (gt / JU rows have gstride columns, the derivative
  // term averages over nx_all columns), only the outputs of ymask (ny of them) enter the derivative term, no_primal drops mse(u)
  int gstride, nx_all, ny, no_primal; unsigned ymask;
};
int launch_sob(const SNetArgs& a, bool train, int ns, const int* seeds, const float* gt, float wj, float* ring, float* ju,
               bool query_only, hipStream_t st, const SobPar* par = nullptr);
bool snet3_supported(const SNetArgs& a);
int snet3_nbl(int n);
int snet3_nsm(int si, int so, int nh, int n);
long snet3_plane_floats(int n);
long snet3_ring_floats_per_wave(int n, int nh);
bool jac_supported(const SNetArgs& a);      // JacobianLayer / HessianLayer: wider than snet3_supported (single plane buffer when needed)
void launch_jac(const SNetArgs& a, int ns, const int* seeds, const float* const* zd, int nx_total, int x0, float* dydx,
                hipStream_t st);
// seed_j / seed_k: coordinate index, or -1 for a parameter column (then zd_j / zd_k = dz/dp of that column [tiles][r][32], and
// zdd = d2z/dp_j dp_k when both are parameter columns)
void launch_hess(const SNetArgs& a, int seed_j, int seed_k, int hj, int hk, int nx_total, float* dydx, float* d2ydx2, hipStream_t st,
                 const float* zd_j = nullptr, const float* zd_k = nullptr, const float* zdd = nullptr);
void launch_mlp_jac(const PNetArgs& a, int NB, int seed, float* ZD, hipStream_t st);
void launch_ll_jac_out(const float* PHI, const float* Z, const float* PHID, const float* ZD, long B, int r, int so,
                       int nx_total, int xcol, float* dydx, hipStream_t st);
void launch_pack16(const float* theta, const MatRef& m, int NBL, f32x4* WF, f32x4* WB, hipStream_t st);
int launch_gw_mfma(const GwArgs& a, int NBI, int NBO, int rows, hipStream_t st);   // -1: bf16 dL/da rows handed to a shape without a DAB form
void launch_gw_first(const GwArgs& a, int NBO, int rows, hipStream_t st);
void launch_gw_out(const GwArgs& a, int NBI, int rows, hipStream_t st);
void launch_reduce(const float* partial, long pstride, int rows, const float* loss_partial, int nloss,
                   float* g, long P, hipStream_t st);
void launch_reg(const float* theta, float* g, long lo, long hi, long P, float l1, float l2, hipStream_t st);
void launch_metric(const float* g, long P, float weight, double* acc, hipStream_t st);
void launch_reduce_adam(const float* partial, long pstride, int rows, const float* loss_partial, int nloss, float* g, long P,
                        float* theta, float* m, float* v, float lr_t, float b1, float b2, float eps, hipStream_t st);
struct AdamDev { float lr, beta1, beta2, eps; long step; };      // device-resident Adam state of a captured graph of steps (k_adam_dev)
void launch_adam_dev(float* theta, const float* g, float* m, float* v, long P, AdamDev* ad, hipStream_t st);
void launch_adam(float* theta, const float* g, float* m, float* v, long P, float lr_t, float b1, float b2,
                 float eps, hipStream_t st);
void launch_latent_to_w(const float* theta, long off_Wh, long off_bh, int r, long po, const float* lr, long B,
                        float* w, hipStream_t st);
void launch_given_w(const float* x, const float* w, float* u, long B, int si, int so, int n, int nh, long po,
                    int act, int res, int nif_skip, float omega, hipStream_t st);
// last-layer-parameterised class: u = Dot(phi(x), a) + bias and its adjoint (nif/model.py:1240-1269)
struct LLArgs {
  int loss_kind;
  const float* theta; long bias_off; long last_w;   // last_layer_bias [so]; pnet last layer W[r][r]
  const float* PHI;     // [tiles][so*r][32]  (row s*r + j)
  const float* Z;       // [tiles][r][32]     pnet output a
  long B; int r, so;
  const float* y; const float* sw; float inv_bg;
  float* u_out;         // [B][so] or null
  float* DU;            // [tiles][so][32]
  float* DPHI;          // [tiles][so*r][32]
  float* DA;            // [tiles][r][32]   dL/da
  float* DZL;           // [tiles][r][32]   dL/d latent (through the r x r map)
  float* loss_partial;  // [gridDim.x]
};
void launch_ll_out(const LLArgs& a, bool train, hipStream_t st);
// latent Jacobian regulariser (k_pjac.hip): tangents of the ParameterNet + their adjoint; operand pairs into the stash
bool pjac_supported(const PNetArgs& a);
int launch_pjac(const PNetArgs& a, float coef, float* MU, float* loss_partial, int c0, int nd, hipStream_t st);   // columns [c0, c0 + nd), nd <= pjac_group()
int pjac_group();
void launch_pjac2(const PNetArgs& a, int cj, int ck, float* ZDD, hipStream_t st);   // d2z/dp_cj dp_ck -> ZDD [tiles][r][32]
int launch_pjac_fwd(const PNetArgs& a, float* ZT, hipStream_t st);      // dz/dp_d of every parameter column -> ZT [pi][tiles][r][32]
int launch_pjac_adj(const PNetArgs& a, const float* MU_in, const int* mu_blk, float* MU, float* loss_partial, int c0, int nd, hipStream_t st);   // mu_blk[column]
void launch_axpy_cols(float* g, const float* tmp, long ncols, long P, hipStream_t st);
// activity regulariser of the last-layer class (the ParameterNet output is the materialised [B, r] tensor there)
int launch_ll_actreg(const float* Za, const float* lw, int r, long B, float coef, bool l1, float* DA, float* DZL, float* loss_partial,
                     hipStream_t st);
void launch_add_sum(const float* parts, int n, float* dst, hipStream_t st);
// activity regulariser of the (virtual) pnet_output (k_misc.hip)
int actreg_max_r();
void launch_actreg_points(bool l1, const float* theta, long off_W, long off_b, int r, long po, const float* Z, long B, float coef,
                          float* DZ, float* loss_partial, hipStream_t st);
void launch_actreg_planes(bool l1, const float* theta, long off_W, long off_b, int r, long po, const float* Z, long B, int nslab,
                          float* part, hipStream_t st);
void launch_actreg_apply(const float* part, int nslab, int r, long po, float coef, long off_W, long off_b, const float* loss_partial,
                         int nloss, float* g, long P, hipStream_t st);
// HessianLayer epilogues on the device (k_misc.hip)
struct HessIdx { int ny; int y_idx[16]; };
struct HessLLArgs {
  const float *f0, *fj, *fh;      // phi [B][so*rl], phi' [B][so*rl][nxc], phi'' [B][so*rl][nxc][nxc]
  const float *Za, *AP, *bias;    // a [tiles][rl][32]; AP: [np + np*np][npts][rl] blocks in latent layout: a'_j, then a''_{jk} at np + j*np + k (j <= k)
  long B, npts; int so, rl, nxc, np, nx;
  HessIdx I; int xc[16], xp[16];   // positions (in x_index) of the coordinate / parameter columns (r4: up to 16 parameter columns)
  float *y, *dydx, *d2;
};
void launch_hess_gather(const float* fj, const float* fh, long B, int so, int nx, const HessIdx& I, float* dydx, float* d2, hipStream_t st);
void launch_through_lw(const float* SRC, const long* src_off_dev, int nvec, const float* lw, int rl, long npts, float* DST, hipStream_t st);
void launch_ll_hess(const HessLLArgs& H, hipStream_t st);
void launch_gather_rows(const float* src, const int* perm, long n, int ncol, float* dst, hipStream_t st);
void launch_rows_to_tiles(const float* rows, long B, int c, float* tiles, hipStream_t st);
void launch_tiles_to_rows(const float* tiles, long B, int c, float* rows, hipStream_t st);

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int fmap(int v, int hf) { return 8 * (v >> 2) + 4 * hf + (v & 3); }

// sin and cos of x in one go: 3-term Cody-Waite reduction by pi/2 (exact products via fma) and the
// cephes single-precision minimax kernels on [-pi/4, pi/4].  Max abs error 1.1e-7 for |x| < 2^20
// (checked against fp64 on 2e6 samples per decade); larger arguments reduce in double precision.
__device__ __forceinline__ void nif_sincos_poly(float r, int q, float* sp, float* cp) {
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  const float s = fmaf(ps * r2, r, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  const float c = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const float ss = (q & 1) ? c : s;
  const float cc = (q & 1) ? s : c;
  *sp = (q & 2) ? -ss : ss;
  *cp = ((q + 1) & 2) ? -cc : cc;
}
#ifndef NIF_HW_SINCOS
#define NIF_HW_SINCOS 1
#endif
#if NIF_HW_SINCOS
// Fast path: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS; the reduction f = x/2pi - rint(x/2pi)
// is done with two fmas against a hi/lo split of 1/2pi (the first fma is exact up to its single rounding, so
// |f| <= 0.5 carries <= 2^-25 rev = 1.9e-7 rad).  Measured on MI355X against fp64 (tools/exp/hw_sincos.hip,
// 4M samples per range, |x| <= 0.5 ... 1e6): max abs error 2.6e-7, rms 5.3e-8 -- the same as fp32 rounding of the
// result, and ~2.3x fewer VALU issue cycles than the polynomial kernels below (2 quarter-rate + 4 full-rate ops).
__device__ __forceinline__ void nif_sincosf_core(float x, float* sp, float* cp) {
  const float k = rintf(x * 0.15915493667125702f);
  float f = fmaf(x, 0.15915493667125702f, -k);
  f = fmaf(x, 6.420638326565253e-09f, f);
  *sp = __builtin_amdgcn_sinf(f);
  *cp = __builtin_amdgcn_cosf(f);
}
// sine only, plus a float whose SIGN is the sign of the cosine: cos(2 pi f) < 0  <=>  |f| > 1/4 on the reduced revolution
// fraction f in [-1/2, 1/2] -- saves the quarter-rate v_cos_f32 where only the sign of cos(a) is kept (k_snet4 SGN)
__device__ __forceinline__ void nif_sin_cossign_core(float x, float* sp, float* csign) {
  const float k = rintf(x * 0.15915493667125702f);
  float f = fmaf(x, 0.15915493667125702f, -k);
  f = fmaf(x, 6.420638326565253e-09f, f);
  *sp = __builtin_amdgcn_sinf(f);
  *csign = 0.25f - fabsf(f);
}
#else
__device__ __forceinline__ void nif_sincosf_core(float x, float* sp, float* cp) {
  const float k = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-k, 1.57079637050628662109375f, x);
  r = fmaf(-k, -4.371138828673793e-08f, r);
  r = fmaf(-k, -1.7151245100058819e-15f, r);
  nif_sincos_poly(r, (int)k, sp, cp);
}
#endif
// |x| >= 2^20: same kernels, argument reduction in fp64 (2-term Cody-Waite, k < 2^52)
__device__ __forceinline__ void nif_sincosf_big(float x, float* sp, float* cp) {
  const double xd = (double)x;
  const double k = rint(xd * 0.63661977236758134308);
  double r = fma(-k, 1.5707963267948966, xd);
  r = fma(-k, 6.123233995736766e-17, r);
  const long long kq = (long long)fmod(k, 4.0);
  nif_sincos_poly((float)r, (int)(kq & 3), sp, cp);
}
#define NIF_SINCOS_FAST_LIMIT 1048576.0f
__device__ __forceinline__ void nif_sincosf(float x, float* sp, float* cp) {
  if (__builtin_expect(!(fabsf(x) < NIF_SINCOS_FAST_LIMIT), 0)) nif_sincosf_big(x, sp, cp);
  else nif_sincosf_core(x, sp, cp);
}

// erf, branch free (both ranges evaluated, one select): 1-ulp minimax forms for |x| <= 0.9277 (x P(x^2)) and beyond
// (1 - exp(Q(|x|))), max abs error 5.8e-8 against fp64 over [-6, 6] (checked on 2e6 points).  r3: ocml's erff is a branchy
// routine; inlined 64 times into the 128-unit k_mlp_jac instantiation (1100 spilled registers) it produced WRONG tangents for
// gelu ParameterNets of 65-128 units (tools/fuzz_parity.py over the wider shapes; every other activation was right)
__device__ __forceinline__ float nif_erff(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}

// h = f(a), d = f'(a) for the Keras activation ACT (compile time)
template <int ACT>
__device__ __forceinline__ void act_eval(float a, float* h, float* d) {
  if (ACT == ACT_SINE) {
    nif_sincosf(a, h, d);
  } else if (ACT == ACT_SWISH) {
    // sigmoid on v_exp_f32 + v_rcp_f32 (1 ulp each; the IEEE division sequence is ~10 instructions)
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
    *h = a * s; *d = s * (1.0f + a * (1.0f - s));
  } else if (ACT == ACT_TANH) {
    const float t = tanhf(a); *h = t; *d = 1.0f - t * t;
  } else if (ACT == ACT_RELU) {
    *h = a > 0.f ? a : 0.f; *d = a > 0.f ? 1.f : 0.f;
  } else if (ACT == ACT_SIGMOID) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-a)); *h = s; *d = s * (1.0f - s);
  } else if (ACT == ACT_ELU) {
    const float e = expf(fminf(a, 0.f)); *h = a > 0.f ? a : e - 1.0f; *d = a > 0.f ? 1.0f : e;
  } else if (ACT == ACT_SOFTPLUS) {
    *h = fmaxf(a, 0.f) + log1pf(expf(-fabsf(a)));
    *d = 1.0f / (1.0f + expf(-a));
  } else if (ACT == ACT_GELU) {
    const float cdf = 0.5f * (1.0f + nif_erff(a * 0.70710678118654752440f));
    *h = a * cdf; *d = cdf + a * 0.3989422804014327f * expf(-0.5f * a * a);
  } else if (ACT == ACT_SELU) {          // scale * elu(a, alpha) (keras/activations.py selu)
    const float e = NIF_SELU_ALPHA * expf(fminf(a, 0.f));
    *h = NIF_SELU_SCALE * (a > 0.f ? a : e - NIF_SELU_ALPHA); *d = NIF_SELU_SCALE * (a > 0.f ? 1.0f : e);
  } else if (ACT == ACT_SOFTSIGN) {      // a / (1 + |a|)
    const float q = 1.0f / (1.0f + fabsf(a));
    *h = a * q; *d = q * q;
  } else if (ACT == ACT_EXPONENTIAL) {
    const float e = expf(a); *h = e; *d = e;
  } else if (ACT == ACT_HARD_SIGMOID) {  // Keras 2.11: clip(0.2 a + 0.5, 0, 1)
    const float t = fmaf(0.2f, a, 0.5f);
    *h = fminf(fmaxf(t, 0.f), 1.f); *d = (t > 0.f && t < 1.f) ? 0.2f : 0.f;
  } else {
    *h = a; *d = 1.0f;
  }
}

// activation of a whole register tile; features >= n (padding) are forced to h = 0, d = 0
template <int NB, int ACT>
__device__ __forceinline__ void act_tile_t(const f32x16 (&a)[NB], f32x16 (&h)[NB], f32x16 (&d)[NB], int n, int hf) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      float hv, dv;
      act_eval<ACT>(a[b][v], &hv, &dv);
      const bool ok = (32 * b + fmap(v, hf)) < n;
      h[b][v] = ok ? hv : 0.f;
      d[b][v] = ok ? dv : 0.f;
    }
}
// SIREN tile: one wave-uniform range check for the whole tile instead of a branch per element.
// sched_barriers keep the scheduler from interleaving all 16*NB sincos chains (register pressure).
template <int NB>
__device__ __forceinline__ void sine_tile(const f32x16 (&a)[NB], f32x16 (&h)[NB], f32x16 (&d)[NB], int n, int hf) {
  float mx = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) mx = fmaxf(mx, fabsf(a[b][v]));
  if (__builtin_expect(__any(!(mx < NIF_SINCOS_FAST_LIMIT)), 0)) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        float hv, dv;
        nif_sincosf_big(a[b][v], &hv, &dv);
        const bool ok = (32 * b + fmap(v, hf)) < n;
        h[b][v] = ok ? hv : 0.f;
        d[b][v] = ok ? dv : 0.f;
      }
    return;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      float hv, dv;
      nif_sincosf_core(a[b][v], &hv, &dv);
      const bool ok = (32 * b + fmap(v, hf)) < n;
      h[b][v] = ok ? hv : 0.f;
      d[b][v] = ok ? dv : 0.f;
    }
}

// wave-uniform dispatch on the runtime activation id (one switch per tile, not per element)
template <int NB>
__device__ __forceinline__ void act_tile(int act, const f32x16 (&a)[NB], f32x16 (&h)[NB], f32x16 (&d)[NB], int n, int hf) {
  switch (act) {
    case ACT_SINE: sine_tile<NB>(a, h, d, n, hf); break;
    case ACT_SWISH: act_tile_t<NB, ACT_SWISH>(a, h, d, n, hf); break;
    case ACT_TANH: act_tile_t<NB, ACT_TANH>(a, h, d, n, hf); break;
    case ACT_RELU: act_tile_t<NB, ACT_RELU>(a, h, d, n, hf); break;
    case ACT_SIGMOID: act_tile_t<NB, ACT_SIGMOID>(a, h, d, n, hf); break;
    case ACT_ELU: act_tile_t<NB, ACT_ELU>(a, h, d, n, hf); break;
    case ACT_SOFTPLUS: act_tile_t<NB, ACT_SOFTPLUS>(a, h, d, n, hf); break;
    case ACT_GELU: act_tile_t<NB, ACT_GELU>(a, h, d, n, hf); break;
    case ACT_SELU: act_tile_t<NB, ACT_SELU>(a, h, d, n, hf); break;
    case ACT_SOFTSIGN: act_tile_t<NB, ACT_SOFTSIGN>(a, h, d, n, hf); break;
    case ACT_EXPONENTIAL: act_tile_t<NB, ACT_EXPONENTIAL>(a, h, d, n, hf); break;
    case ACT_HARD_SIGMOID: act_tile_t<NB, ACT_HARD_SIGMOID>(a, h, d, n, hf); break;
    default: act_tile_t<NB, ACT_LINEAR>(a, h, d, n, hf); break;
  }
}

// T[ob] = sum over in-blocks/steps of A(packed weights) x B(hin): one 32x32x2 fp32 MFMA per K-pair.
// Wp: packed plane of NBO*NBI blocks, block (ob,ib) = 4 quads x 64 lanes of f32x4.
template <int NBI, int NBO>
__device__ __forceinline__ void dense_mfma(const f32x4* __restrict__ Wp, const f32x16 (&hin)[NBI], f32x16 (&T)[NBO],
                                           int lane) {
#pragma unroll
  for (int ob = 0; ob < NBO; ++ob) {
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib) {
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const f32x4 a = Wp[((ob * NBI + ib) * 4 + vq) * 64 + lane];
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], hin[ib][4 * vq + 0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], hin[ib][4 * vq + 1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], hin[ib][4 * vq + 2], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], hin[ib][4 * vq + 3], t, 0, 0, 0);
      }
    }
    T[ob] = t;
  }
}

// LDS copies of the small ParameterNet vectors (first-layer rows, biases, bottleneck columns) in REGISTER-TILE
// order: element ((b*2 + hf)*16 + v) = vec[(32b + fmap(v,hf)) * stride], zero beyond n -- a lane fetches its 16
// features of a block with four ds_read_b128 instead of 16 dependent global loads per tile
template <int NB>
__device__ __forceinline__ void psmall_fill(float* dst, const float* __restrict__ src, int n, int stride, int tid, int nthreads) {
  for (int e = tid; e < NB * 32; e += nthreads) {
    const int b = e >> 5, hf = (e >> 4) & 1, v = e & 15;
    const int f = 32 * b + fmap(v, hf);
    dst[e] = f < n ? src[(long)f * stride] : 0.f;
  }
}
__device__ __forceinline__ f32x16 psmall_get(const float* base, int b, int hf) {
  const f32x4* q = reinterpret_cast<const f32x4*>(base + (b * 2 + hf) * 16);
  const f32x4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
  f32x16 r;
  r[0] = a0[0]; r[1] = a0[1]; r[2] = a0[2]; r[3] = a0[3]; r[4] = a1[0]; r[5] = a1[1]; r[6] = a1[2]; r[7] = a1[3];
  r[8] = a2[0]; r[9] = a2[1]; r[10] = a2[2]; r[11] = a2[3]; r[12] = a3[0]; r[13] = a3[1]; r[14] = a3[2]; r[15] = a3[3];
  return r;
}
// all small vectors of a ParameterNet: [first_w: pi][first_b][hid_b: lst][hid_b2: lst (resblocks)][bott_w: r], NB*32 each
struct PSmall { const float *fw, *fb, *hb, *hb2, *bw; };
__host__ __device__ inline int psmall_floats(const PNetArgs& A, int NB) { return (A.pi + 1 + A.lst * (A.res ? 2 : 1) + A.r) * NB * 32; }
template <int NB>
__device__ __forceinline__ PSmall psmall_stage(const PNetArgs& A, float* lds, int tid, int nthreads) {
  constexpr int S = NB * 32;
  float* q = lds;
  PSmall P;
  P.fw = q; for (int dd = 0; dd < A.pi; ++dd) psmall_fill<NB>(q + dd * S, A.theta + A.first_w + (long)dd * A.nst, A.nst, 1, tid, nthreads);
  q += A.pi * S;
  P.fb = q; psmall_fill<NB>(q, A.theta + A.first_b, A.nst, 1, tid, nthreads); q += S;
  P.hb = q; for (int i = 0; i < A.lst; ++i) psmall_fill<NB>(q + i * S, A.theta + A.hid_b[i], A.nst, 1, tid, nthreads);
  q += A.lst * S;
  P.hb2 = q;
  if (A.res) { for (int i = 0; i < A.lst; ++i) psmall_fill<NB>(q + i * S, A.theta + A.hid_b2[i], A.nst, 1, tid, nthreads); q += A.lst * S; }
  P.bw = q; for (int c = 0; c < A.r; ++c) psmall_fill<NB>(q + c * S, A.theta + A.bott_w + c, A.nst, A.r, tid, nthreads);
  return P;
}

// ACT >= 0: activation fixed at compile time (the SIREN hot path); ACT < 0: runtime id
template <int NB, int ACT>
__device__ __forceinline__ void act_tile_sel(int act, const f32x16 (&a)[NB], f32x16 (&h)[NB], f32x16 (&d)[NB], int n, int hf) {
  if (ACT == ACT_SINE) sine_tile<NB>(a, h, d, n, hf);
  else if (ACT == ACT_SWISH) act_tile_t<NB, ACT_SWISH>(a, h, d, n, hf);
  else act_tile<NB>(act, a, h, d, n, hf);
}

// same, A operands from an LDS-resident plane; ACCUM keeps the incoming T
template <int NBI, int NBO, bool ACCUM>
__device__ __forceinline__ void dense_mfma_lds(const f32x4* plane, const f32x16 (&hin)[NBI], f32x16 (&T)[NBO], int lane) {
#pragma unroll
  for (int ob = 0; ob < NBO; ++ob) {
    f32x16 t;
    if (ACCUM) t = T[ob];
    else {
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = 0.f;
    }
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib) {
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const f32x4 a = plane[((ob * NBI + ib) * 4 + vq) * 64 + lane];
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], hin[ib][4 * vq + 0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], hin[ib][4 * vq + 1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], hin[ib][4 * vq + 2], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], hin[ib][4 * vq + 3], t, 0, 0, 0);
      }
    }
    T[ob] = t;
  }
}

template <int NB>
__device__ __forceinline__ void stash_store(float* __restrict__ slot, long tile, const f32x16 (&h)[NB], int p, int hf) {
  float* t = slot + tile * (long)(NB * 32 * 32);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) t[(32 * b + fmap(v, hf)) * 32 + p] = h[b][v];
}
template <int NB>
__device__ __forceinline__ void stash_load(const float* __restrict__ slot, long tile, f32x16 (&h)[NB], int p, int hf) {
  const float* t = slot + tile * (long)(NB * 32 * 32);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) h[b][v] = t[(32 * b + fmap(v, hf)) * 32 + p];
}
#endif  // __HIPCC__
