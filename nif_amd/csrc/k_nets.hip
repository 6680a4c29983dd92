// k_nets.hip -- ParameterNet and hypernetwork-ShapeNet forward / adjoint kernels for gfx950.
//
// One wavefront = one tile of 32 points.  All n x n layers run on v_mfma_f32_32x32x2_f32 with the
// activation tile resident in registers (see nif_internal.h for the layout); the hypernetwork's
// per-sample weights W(a) = sum_k z_k(a) Wh_k + Bh are never materialised:
//     h . W(a) = sum_k zt_k(a) * (h . M^(k)),   zt = (z_1..z_r, 1)
// i.e. (r+1) shared-weight GEMMs whose results are combined per point on the VALU.
//
// Reference semantics implemented here (file:line under the reference tree):
//   ParameterNet   nif/model.py:326-343 over the layers of :178-231 / :591-734
//                  (Dense, MLP_SimpleShortCut mlp.py:148-160, MLP_ResNet mlp.py:62-79,
//                   SIREN siren.py:256-281, SIREN_ResNet siren.py:381-410)
//   ShapeNet       NIF._call_shape_net model.py:233-324, NIFMultiScale._call_shape_net_mres :738-954
//   loss           Keras 'mse' (README.md:33); adjoint hand-derived (SURVEY a-10)
#include "nif_internal.h"
#include "k_pnet_bf16.h"

// (forcing a higher occupancy on the ParameterNet kernels with launch bounds was measured: 2-3x slower, spills)

// ============================================================================================
// weight packing: theta -> MFMA A-operand order
//   fwd plane: block (ob,ib), quad vq, lane, c : M[in = 32ib + fmap(4vq+c, lane>>5)][out = 32ob + (lane&31)]
//   bwd plane: block (ib,ob), quad vq, lane, c : M[in = 32ib + (lane&31)][out = 32ob + fmap(4vq+c, lane>>5)]
// ============================================================================================
__global__ void k_pack(const float* __restrict__ theta, MatRef m, int NBI, int NBO, float* __restrict__ WF,
                       float* __restrict__ WB) {
  const long per_plane = (long)NBI * NBO * 1024;
  const long total = per_plane * (m.r + 1);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = (int)(idx / per_plane);
    long rem = idx - (long)k * per_plane;
    const int c = rem & 3;
    const int lane = (rem >> 2) & 63;
    const int vq = (rem >> 8) & 3;
    const int blk = (int)(rem >> 10);
    const int v = 4 * vq + c;
    {  // forward plane: blk = ob*NBI + ib
      const int ob = blk / NBI, ib = blk % NBI;
      const int in = 32 * ib + fmap(v, lane >> 5), out = 32 * ob + (lane & 31);
      WF[idx] = (in < m.nin && out < m.nout) ? theta[matref_index(m, k, in, out)] : 0.f;
    }
    {  // backward plane: blk = ib*NBO + ob   (output blocks of U are the in-feature blocks)
      const int ib = blk / NBO, ob = blk % NBO;
      const int in = 32 * ib + (lane & 31), out = 32 * ob + fmap(v, lane >> 5);
      WB[idx] = (in < m.nin && out < m.nout) ? theta[matref_index(m, k, in, out)] : 0.f;
    }
  }
}

void launch_pack(const float* theta, const MatRef& m, int NBI, int NBO, f32x4* WF, f32x4* WB, hipStream_t st) {
  const long total = (long)NBI * NBO * 1024 * (m.r + 1);
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, st, theta, m, NBI, NBO, (float*)WF, (float*)WB);
}

// ============================================================================================
// ParameterNet
// ============================================================================================
// first layer (tiny K = pi): VALU.  a = omega * sum_d p_d W[d][f] + b[f]
template <int NB>
__device__ __forceinline__ void pnet_first(const PNetArgs& A, long ptc, int hf, f32x16 (&h)[NB], f32x16 (&d)[NB]) {
  const float* prow = A.xin + ptc * A.ncol + A.col0;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int f = 32 * b + fmap(v, hf);
      float a = 0.f;
      if (f < A.nst) {
        float acc = 0.f;
        for (int dd = 0; dd < A.pi; ++dd) acc = fmaf(prow[dd], A.theta[A.first_w + (long)dd * A.nst + f], acc);
        a = A.omega * acc + A.theta[A.first_b + f];
      }
      h[b][v] = a;
    }
  act_tile<NB>(A.act, h, h, d, A.nst, hf);
}

// floats of the bf16 plane region of k_pnet's LDS (the launcher decides A.pbf2 before the launch)
__host__ __device__ inline int pnet_plane_floats(const PNetArgs& A, int NB) {
  const int nm = A.lst * (A.res ? 2 : 1);
  return NB == 1 ? nm * PBF_FWD_U4 * 4 : ((NB == 2 && A.pbf2) ? nm * 4 * PBF_FWD_U4 * 4 : 0);
}
__host__ __device__ inline int pnet_tail_floats(const PNetArgs& A) { return A.r + (A.ll_kind ? A.r + A.r * A.r : 0); }

template <int NB, bool TRAIN, int ACT>
__global__ __launch_bounds__(256) void k_pnet(PNetArgs A) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = lane & 31, hf = lane >> 5;
  const long ntiles = (A.B + 31) / 32;
  const int nm = A.lst * (A.res ? 2 : 1);
  const long plane = (long)NB * NB * 256;  // f32x4 per packed matrix
  extern __shared__ __attribute__((aligned(16))) float pn_lds[];   // [LL kind: 4 waves x r x 32][small vectors]
  float* zl_lds = pn_lds;
  // r4: the bottleneck bias and the last-layer class's r x r map + bias live in LDS (tail of the small-vector region) -- the
  // r-trip loops at the end of a tile used to take them from global memory, one dependent scalar load per latent row (cfg-4, r = 10)
  float* tailv = pn_lds + (A.ll_kind ? 4 * A.r * 32 : 0) + ((psmall_floats(A, NB) + 3) & ~3) + pnet_plane_floats(A, NB);
  for (int e = threadIdx.x; e < A.r; e += 256) tailv[e] = A.theta[A.bott_b + e];
  if (A.ll_kind) {
    for (int e = threadIdx.x; e < A.r; e += 256) tailv[A.r + e] = A.theta[A.last_b + e];
    for (int e = threadIdx.x; e < A.r * A.r; e += 256) tailv[2 * A.r + e] = A.theta[A.last_w + e];
  }
  const PSmall S = psmall_stage<NB>(A, pn_lds + (A.ll_kind ? 4 * A.r * 32 : 0), threadIdx.x, 256);
  // one 32-feature block: the hidden products run as exact bf16 splits (k_pnet_bf16.h), forward planes built here
  pbf16x8* bpl = reinterpret_cast<pbf16x8*>(pn_lds + (A.ll_kind ? 4 * A.r * 32 : 0) + ((psmall_floats(A, NB) + 3) & ~3));
  // two blocks (33..64 units, r3): the same when the launcher found room for the planes (A.pbf2: 24 KB per matrix), else the
  // f32-input MFMAs on the packed planes in global memory
  const bool bf2 = NB == 2 && A.pbf2;
  if constexpr (NB == 1) {
    for (int m = 0; m < nm; ++m) {
      const long w_off = A.res ? ((m & 1) ? A.hid_w2[m >> 1] : A.hid_w[m >> 1]) : A.hid_w[m];
      pbf_build(bpl + m * PBF_FWD_U4, nullptr, A.theta, w_off, A.nst, threadIdx.x, 256);
    }
  } else if constexpr (NB == 2) {
    if (bf2)
      for (int m = 0; m < nm; ++m) {
        const long w_off = A.res ? ((m & 1) ? A.hid_w2[m >> 1] : A.hid_w[m >> 1]) : A.hid_w[m];
        pbfn_build_fwd<2>(bpl + m * 4 * PBF_FWD_U4, A.theta, w_off, A.nst, threadIdx.x, 256);
      }
  }
  __syncthreads();
  for (long tile = (long)blockIdx.x * 4 + wid; tile < ntiles; tile += (long)gridDim.x * 4) {
  long pt = tile * 32 + p;
  const long ptc = pt < A.B ? pt : A.B - 1;

  f32x16 h[NB], d[NB], T[NB];
  {
    const float* prow = A.xin + ptc * A.ncol + A.col0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
      for (int dd = 0; dd < A.pi; ++dd) acc += prow[dd] * psmall_get(S.fw + dd * NB * 32, b, hf);
      h[b] = A.omega * acc + psmall_get(S.fb, b, hf);
    }
    act_tile_sel<NB, ACT>(A.act, h, h, d, A.nst, hf);
  }
  if (TRAIN) stash_store<NB>(A.stash + (long)(nm + 1) * A.slot_stride, tile, d, p, hf);

  for (int i = 0; i < A.lst; ++i) {
    if (!A.res) {
      // MLP_SimpleShortCut: h + act(hK+b)   |   SIREN hidden: sin(w0 hW + b)
      if (TRAIN) stash_store<NB>(A.stash + (long)i * A.slot_stride, tile, h, p, hf);
      if constexpr (NB == 1) pbf_dense_fwd(bpl + i * PBF_FWD_U4, h[0], T[0], lane);
      else if (NB == 2 && bf2) pbfn_dense_fwd<NB>(bpl + i * (NB * NB * PBF_FWD_U4), h, T, lane);
      else dense_mfma<NB, NB>(A.WF + (long)i * plane, h, T, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b) T[b] = A.omega * T[b] + psmall_get(S.hb + i * NB * 32, b, hf);
      act_tile_sel<NB, ACT>(A.act, T, T, d, A.nst, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) h[b] = A.siren ? T[b] : h[b] + T[b];
      if (TRAIN) stash_store<NB>(A.stash + (long)(nm + 2 + i) * A.slot_stride, tile, d, p, hf);
    } else {
      // MLP_ResNet: act(h + L2(act(L1 h)))   |   SIREN_ResNet: 0.5 (h + sin(w0 sin(w0 hW+b) W2 + b2))
      f32x16 t[NB];
      if (TRAIN) stash_store<NB>(A.stash + (long)(2 * i) * A.slot_stride, tile, h, p, hf);
      if constexpr (NB == 1) pbf_dense_fwd(bpl + (2 * i) * PBF_FWD_U4, h[0], T[0], lane);
      else if (NB == 2 && bf2) pbfn_dense_fwd<NB>(bpl + (2 * i) * (NB * NB * PBF_FWD_U4), h, T, lane);
      else dense_mfma<NB, NB>(A.WF + (long)(2 * i) * plane, h, T, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b) T[b] = A.omega * T[b] + psmall_get(S.hb + i * NB * 32, b, hf);
      act_tile_sel<NB, ACT>(A.act, T, t, d, A.nst, hf);
      if (TRAIN) {
        stash_store<NB>(A.stash + (long)(nm + 2 + 2 * i) * A.slot_stride, tile, d, p, hf);
        stash_store<NB>(A.stash + (long)(2 * i + 1) * A.slot_stride, tile, t, p, hf);
      }
      if constexpr (NB == 1) pbf_dense_fwd(bpl + (2 * i + 1) * PBF_FWD_U4, t[0], T[0], lane);
      else if (NB == 2 && bf2) pbfn_dense_fwd<NB>(bpl + (2 * i + 1) * (NB * NB * PBF_FWD_U4), t, T, lane);
      else dense_mfma<NB, NB>(A.WF + (long)(2 * i + 1) * plane, t, T, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x16 lin = A.omega * T[b] + psmall_get(S.hb2 + i * NB * 32, b, hf);
        T[b] = A.siren ? lin : h[b] + lin;
      }
      act_tile_sel<NB, ACT>(A.act, T, T, d, A.nst, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) h[b] = A.siren ? 0.5f * (h[b] + T[b]) : T[b];
      if (TRAIN) stash_store<NB>(A.stash + (long)(nm + 2 + 2 * i + 1) * A.slot_stride, tile, d, p, hf);
    }
  }
  if (TRAIN) stash_store<NB>(A.stash + (long)nm * A.slot_stride, tile, h, p, hf);
  // bottleneck (linear, nst -> r): per-lane partial dot products + one cross-half exchange
  float* zl = zl_lds + (long)wid * A.r * 32;
  for (int c = 0; c < A.r; ++c) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f32x16 w = psmall_get(S.bw + c * NB * 32, b, hf);
#pragma unroll
      for (int v = 0; v < 16; ++v) s = fmaf(h[b][v], w[v], s);
    }
    s += __shfl_xor(s, 32);
    s += tailv[c];
    if (A.ll_kind) {
      if (hf == 0) { zl[c * 32 + p] = s; A.ZL[(tile * A.zl_rows + c) * 32 + p] = s; }
    } else if (hf == 0) {
      A.Z[(tile * A.r + c) * 32 + p] = s;
    }
  }
  if (A.ll_kind) {
    // last-layer class: pnet_out = latent @ W[r,r] + b   (HyperLinearForSIREN with po = r, model.py:583-585)
    const float* lw = tailv + 2 * A.r;
    for (int c = hf; c < A.r; c += 2) {
      float s = tailv[A.r + c];
#pragma unroll 4
      for (int kk = 0; kk < A.r; ++kk) s = fmaf(zl[kk * 32 + p], lw[kk * A.r + c], s);
      A.Z[(tile * A.r + c) * 32 + p] = s;
    }
  }
  }
}

// adjoint of k_pnet: consumes dL/dz (DZ), turns the stashed derivatives into dL/da in place
template <int NB>
__global__ __launch_bounds__(256) void k_pnet_bwd(PNetArgs A) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = lane & 31, hf = lane >> 5;
  const long ntiles = (A.B + 31) / 32;
  const long tile = (long)blockIdx.x * 4 + wid;
  if (tile >= ntiles) return;
  const int nm = A.lst * (A.res ? 2 : 1);
  const long plane = (long)NB * NB * 256;

  f32x16 gh[NB], d[NB], ga[NB], U[NB];
  // through the bottleneck: gh[f] = sum_c dz_c Wb[f][c]
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) gh[b][v] = 0.f;
  for (int c = 0; c < A.r; ++c) {
    const float dz = A.DZ[(tile * A.r + c) * 32 + p];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int f = 32 * b + fmap(v, hf);
        if (f < A.nst) gh[b][v] = fmaf(dz, A.theta[A.bott_w + (long)f * A.r + c], gh[b][v]);
      }
  }
  for (int i = A.lst - 1; i >= 0; --i) {
    if (!A.res) {
      float* da = A.stash + (long)(nm + 2 + i) * A.slot_stride;
      stash_load<NB>(da, tile, d, p, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) ga[b] = gh[b] * d[b];
      stash_store<NB>(da, tile, ga, p, hf);
      dense_mfma<NB, NB>(A.WB + (long)i * plane, ga, U, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b) gh[b] = A.siren ? A.omega * U[b] : gh[b] + A.omega * U[b];
    } else {
      float* da2 = A.stash + (long)(nm + 2 + 2 * i + 1) * A.slot_stride;
      float* da1 = A.stash + (long)(nm + 2 + 2 * i) * A.slot_stride;
      stash_load<NB>(da2, tile, d, p, hf);
      const float half = A.siren ? 0.5f : 1.0f;
#pragma unroll
      for (int b = 0; b < NB; ++b) ga[b] = half * gh[b] * d[b];
      stash_store<NB>(da2, tile, ga, p, hf);
      dense_mfma<NB, NB>(A.WB + (long)(2 * i + 1) * plane, ga, U, lane);
      // skip path: SIREN_ResNet 0.5*gh ; MLP_ResNet: ga (the act' already applied to x + L2(..))
      f32x16 skip[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) skip[b] = A.siren ? 0.5f * gh[b] : ga[b];
      stash_load<NB>(da1, tile, d, p, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) ga[b] = A.omega * U[b] * d[b];
      stash_store<NB>(da1, tile, ga, p, hf);
      dense_mfma<NB, NB>(A.WB + (long)(2 * i) * plane, ga, U, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b) gh[b] = skip[b] + A.omega * U[b];
    }
  }
  float* da0 = A.stash + (long)(nm + 1) * A.slot_stride;
  stash_load<NB>(da0, tile, d, p, hf);
#pragma unroll
  for (int b = 0; b < NB; ++b) ga[b] = gh[b] * d[b];
  stash_store<NB>(da0, tile, ga, p, hf);
}

void launch_pnet(const PNetArgs& a_, int NSTB, bool train, hipStream_t st) {
  // (LDS layout of k_pnet: [LL: 4 x r x 32 latent rows][small vectors][bf16 planes][tail vectors: r + (LL: r + r^2)])
  PNetArgs a = a_;
  const long ntiles = (a.B + 31) / 32;
  long nblk = (ntiles + 3) / 4;
  static const long cap = [] { const char* e = getenv("NIF_PNET_BLOCKS"); return e ? atol(e) : 2048L; }();
  if (nblk > cap) nblk = cap;            // persistent: the small vectors are staged in LDS once per workgroup
  const int nmat = a.lst * (a.res ? 2 : 1);
  const size_t shm0 = ((a.ll_kind ? (size_t)4 * a.r * 32 : 0) + (size_t)((psmall_floats(a, NSTB) + 3) & ~3)) * sizeof(float);
  static const bool bf2_on = [] { const char* e = getenv("NIF_PNET_BF2"); return !(e && e[0] == '0'); }();
  const size_t tailb = (size_t)pnet_tail_floats(a) * sizeof(float);
  a.pbf2 = (NSTB == 2 && bf2_on && shm0 + tailb + (size_t)nmat * 4 * PBF_FWD_U4 * 16 <= 64u * 1024u) ? 1 : 0;   // two workgroups per CU keep their planes
  if (a.pbf2 && nblk > 512) nblk = 512;  // the planes are built once per workgroup: fewer, longer-lived workgroups
  dim3 grid((unsigned)nblk), block(256);
  const size_t shm = shm0 + (size_t)pnet_plane_floats(a, NSTB) * sizeof(float) + tailb;
#define PNL(NB_, TR_, ACT_)                                                                                             \
  {                                                                                                                     \
    if (shm > 48 * 1024)                                                                                                \
      (void)hipFuncSetAttribute((const void*)k_pnet<NB_, TR_, ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((k_pnet<NB_, TR_, ACT_>), grid, block, shm, st, a);                                              \
  }
#define PNA(NB_, TR_)                              \
  if (a.act == ACT_SWISH) PNL(NB_, TR_, ACT_SWISH) \
  else if (a.act == ACT_SINE) PNL(NB_, TR_, ACT_SINE) \
  else PNL(NB_, TR_, -1)
#define PN(NB_) \
  if (train) { PNA(NB_, true) } else { PNA(NB_, false) }
  if (NSTB == 1) { PN(1) } else if (NSTB == 2) { PN(2) } else { PN(4) }
#undef PN
#undef PNA
#undef PNL
}
void launch_pnet_bwd(const PNetArgs& a, int NSTB, hipStream_t st) {
  const long ntiles = (a.B + 31) / 32;
  dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
  if (NSTB == 1) hipLaunchKernelGGL((k_pnet_bwd<1>), grid, block, 0, st, a);
  else if (NSTB == 2) hipLaunchKernelGGL((k_pnet_bwd<2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_pnet_bwd<4>), grid, block, 0, st, a);
}

// ============================================================================================
// hypernetwork ShapeNet: forward (+ fused MSE and adjoint when TRAIN)
// ============================================================================================
__device__ __forceinline__ float hyp(const SNetArgs& A, int k, long slot) {
  return k < A.r ? A.theta[A.off_Wh + (long)k * A.po + slot] : A.theta[A.off_bh + slot];
}

template <int NB, bool TRAIN, int ACT>
__global__ __launch_bounds__(256, (NB <= 2 ? 2 : 1)) void k_snet(SNetArgs A) {
  extern __shared__ float smem[];  // per wave: dzs[r][64], sks[r][64]; then lsum[4]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = lane & 31, hf = lane >> 5;
  const long ntiles = (A.B + 31) / 32;
  const long tile = (long)blockIdx.x * 4 + wid;
  const bool active = tile < ntiles;
  float loss_lane = 0.f;
  float* dzs = smem + (long)wid * (2 * A.r * 64);
  float* sks = dzs + A.r * 64;
  float* lsum = smem + 4L * (2 * A.r * 64);

  if (active) {
    const long pt = tile * 32 + p;
    const bool valid = pt < A.B;
    const long ptc = valid ? pt : A.B - 1;
    const long plane = (long)NB * NB * 256;
    const int n = A.n, r = A.r;
    const float* xrow = A.xin + ptc * A.ncol + A.col0;
    const float* zt_base = A.Z + tile * r * 32 + p;  // zt_k = k<r ? zt_base[k*32] : 1
    float* IN0 = A.stash;                                  // slot l-1 for IN_l
    float* DA0 = A.stash + (long)(A.nh + 1) * A.slot_stride;  // slot nh+1+l for DA_l
    if (TRAIN)
      for (int k = 0; k < r; ++k) dzs[k * 64 + lane] = 0.f;

    f32x16 h[NB], d[NB], acc[NB], T[NB], ublk[NB];
    // ---- first layer: a0 = w0 * x . W1(a) + b1(a) -------------------------------------------
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
    for (int k = 0; k <= r; ++k) {
      const float zt = k < r ? zt_base[k * 32] : 1.0f;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int f = 32 * b + fmap(v, hf);
          if (f < n) {
            float s = 0.f;
            for (int dd = 0; dd < A.si; ++dd) s = fmaf(xrow[dd], hyp(A, k, slot_w1(A) + (long)dd * n + f), s);
            acc[b][v] = fmaf(zt, fmaf(A.omega, s, hyp(A, k, slot_b1(A) + f)), acc[b][v]);
          }
        }
    }
    act_tile_sel<NB, ACT>(A.act, acc, h, d, n, hf);
    if (TRAIN) stash_store<NB>(DA0, tile, d, p, hf);

    // ---- hidden hyper-matrices ---------------------------------------------------------------
    for (int j = 0; j < A.nh; ++j) {
      if (TRAIN) stash_store<NB>(IN0 + (long)j * A.slot_stride, tile, h, p, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 32] : 1.0f;
        dense_mfma<NB, NB>(A.WF + ((long)j * (r + 1) + k) * plane, h, T, lane);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int f = 32 * b + fmap(v, hf);
            const float hb = f < n ? hyp(A, k, slot_bh(A, j) + f) : 0.f;
            acc[b][v] = fmaf(zt, fmaf(A.omega, T[b][v], hb), acc[b][v]);
          }
      }
      const bool res_first = A.res && !(j & 1);
      const bool res_second = A.res && (j & 1);
      if (res_first) {
#pragma unroll
        for (int b = 0; b < NB; ++b) ublk[b] = h[b];
      }
      act_tile_sel<NB, ACT>(A.act, acc, T, d, n, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 hn = T[b];
        if (A.nif_skip) hn += h[b];
        if (res_second) hn = 0.5f * (ublk[b] + hn);
        h[b] = hn;
      }
      if (TRAIN) stash_store<NB>(DA0 + (long)(j + 1) * A.slot_stride, tile, d, p, hf);
    }

    // ---- last layer (n -> so, linear) + MSE + start of the adjoint --------------------------------
    if (TRAIN) stash_store<NB>(IN0 + (long)A.nh * A.slot_stride, tile, h, p, hf);
    f32x16 gh[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) gh[b][v] = 0.f;
    const float wsamp = (valid ? (A.sw ? A.sw[ptc] : 1.0f) : 0.0f);
    float se = 0.f;
    for (int o = 0; o < A.so; ++o) {
      f32x16 wg[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) wg[b][v] = 0.f;
      float part = 0.f, bias = 0.f;
      for (int k = 0; k <= r; ++k) {
        const float zt = k < r ? zt_base[k * 32] : 1.0f;
        float sk = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int f = 32 * b + fmap(v, hf);
            const float w = f < n ? hyp(A, k, slot_wl(A) + (long)f * A.so + o) : 0.f;
            sk = fmaf(h[b][v], w, sk);
            wg[b][v] = fmaf(zt, w, wg[b][v]);
          }
        part = fmaf(zt, sk, part);
        bias = fmaf(zt, hyp(A, k, slot_bl(A) + o), bias);
        if (TRAIN && k < r) sks[k * 64 + lane] = sk;
      }
      part += __shfl_xor(part, 32);
      const float uo = part + bias;
      if (valid && hf == 0 && A.u_out) A.u_out[pt * A.so + o] = uo;
      if (TRAIN) {
        const float e = uo - A.y[ptc * A.so + o];
        NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
        const float du = dfac * wsamp * A.inv_bg / (float)A.so;
        if (hf == 0) A.DU[(tile * A.so + o) * 32 + p] = du;
#pragma unroll
        for (int b = 0; b < NB; ++b) gh[b] += du * wg[b];
        for (int k = 0; k < r; ++k) {
          float t = du * sks[k * 64 + lane];
          if (hf == 0) t = fmaf(du, A.theta[A.off_Wh + (long)k * A.po + slot_bl(A) + o], t);
          dzs[k * 64 + lane] += t;
        }
      }
    }
    if (TRAIN) {
      if (hf == 0) loss_lane = wsamp * se / (float)A.so * A.inv_bg;

      // ---- adjoint through the hidden hyper-matrices ------------------------------------------
      f32x16 ga[NB], U[NB], hin[NB], skip[NB];
      for (int j = A.nh - 1; j >= 0; --j) {
        float* da = DA0 + (long)(j + 1) * A.slot_stride;
        stash_load<NB>(da, tile, d, p, hf);
        const bool res_second = A.res && (j & 1);
        const bool res_first = A.res && !(j & 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (res_second) { ga[b] = 0.5f * gh[b] * d[b]; skip[b] = 0.5f * gh[b]; }
          else ga[b] = gh[b] * d[b];
          if (A.nif_skip) skip[b] = gh[b];
        }
        stash_store<NB>(da, tile, ga, p, hf);
        stash_load<NB>(IN0 + (long)j * A.slot_stride, tile, hin, p, hf);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
        for (int k = 0; k <= r; ++k) {
          const float zt = k < r ? zt_base[k * 32] : 1.0f;
          dense_mfma<NB, NB>(A.WB + ((long)j * (r + 1) + k) * plane, ga, U, lane);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] += zt * U[b];
          if (k < r) {
            float s = 0.f, sb = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                const int f = 32 * b + fmap(v, hf);
                s = fmaf(hin[b][v], U[b][v], s);
                if (f < n) sb = fmaf(ga[b][v], A.theta[A.off_Wh + (long)k * A.po + slot_bh(A, j) + f], sb);
              }
            dzs[k * 64 + lane] += fmaf(A.omega, s, sb);
          }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          gh[b] = A.omega * acc[b];
          if (A.nif_skip || res_first) gh[b] += skip[b];
        }
      }
      // ---- first layer ---------------------------------------------------------------------
      stash_load<NB>(DA0, tile, d, p, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) ga[b] = gh[b] * d[b];
      stash_store<NB>(DA0, tile, ga, p, hf);
      for (int k = 0; k < r; ++k) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int f = 32 * b + fmap(v, hf);
            if (f < n) {
              float xw = 0.f;
              for (int dd = 0; dd < A.si; ++dd)
                xw = fmaf(xrow[dd], A.theta[A.off_Wh + (long)k * A.po + slot_w1(A) + (long)dd * n + f], xw);
              s = fmaf(ga[b][v], fmaf(A.omega, xw, A.theta[A.off_Wh + (long)k * A.po + slot_b1(A) + f]), s);
            }
          }
        float tot = dzs[k * 64 + lane] + s;
        tot += __shfl_xor(tot, 32);
        if (hf == 0) A.DZ[(tile * r + k) * 32 + p] = tot;
      }
    }
  }
  if (TRAIN) {
    // deterministic block sum of the loss
    for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
    if (lane == 0) lsum[wid] = loss_lane;
    __syncthreads();
    if (threadIdx.x == 0) A.loss_partial[blockIdx.x] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  }
}

void launch_snet(const SNetArgs& a, int NB, bool train, hipStream_t st) {
  const long ntiles = (a.B + 31) / 32;
  dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
  const size_t shm = (size_t)(4 * 2 * a.r * 64 + 4) * sizeof(float);
#define SN(NB_) \
  if (a.act == ACT_SINE) { \
    if (train) hipLaunchKernelGGL((k_snet<NB_, true, ACT_SINE>), grid, block, shm, st, a); \
    else hipLaunchKernelGGL((k_snet<NB_, false, ACT_SINE>), grid, block, shm, st, a); \
  } else { \
    if (train) hipLaunchKernelGGL((k_snet<NB_, true, -1>), grid, block, shm, st, a); \
    else hipLaunchKernelGGL((k_snet<NB_, false, -1>), grid, block, shm, st, a); \
  }
  if (NB == 1) { SN(1) } else if (NB == 2) { SN(2) } else { SN(4) }
#undef SN
}
