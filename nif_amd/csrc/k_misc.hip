// k_misc.hip -- the HBM-bound kernels of the path: per-sample-weight ShapeNet ("given w"), the
// latent->weights map, gradient-row reduction, Adam, and row<->tile layout changes (gfx950).
#include "nif_internal.h"

// ============================================================================================
// model_x_to_u_given_w: u = ShapeNet(x; w) with an arbitrary per-sample weight vector w[po]
// (nif/model.py:435-464, :956-986; the einsum('ai,aij->aj') chain of mlp.py:209-219).
//
// HBM-bound by construction: 4*(si+po+so) bytes per point, 2 flop per weight.  One wavefront per
// point.  Each n x n matrix is streamed with 16-byte loads, 1 KiB per wave-instruction (all loads of
// a matrix are issued before the first use so a wave keeps up to 16 KiB in flight); the vec-mat
// product is done as partial dot products per lane + wavefront shuffle reductions:
//     lane = (g, cq):  g = row inside the 256-float chunk, cq = column quad; acc[c] += h[row]*w[row][4cq+c]
// ============================================================================================
template <int ACT>
__device__ __forceinline__ void act4(float (&a)[4], float (&h)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) { float d; act_eval<ACT>(a[c], &h[c], &d); }
}
__device__ __forceinline__ void act4_dyn(int act, float (&a)[4], float (&h)[4]) {
  switch (act) {
    case ACT_SINE: act4<ACT_SINE>(a, h); break;
    case ACT_SWISH: act4<ACT_SWISH>(a, h); break;
    case ACT_TANH: act4<ACT_TANH>(a, h); break;
    case ACT_RELU: act4<ACT_RELU>(a, h); break;
    case ACT_SIGMOID: act4<ACT_SIGMOID>(a, h); break;
    case ACT_ELU: act4<ACT_ELU>(a, h); break;
    case ACT_SOFTPLUS: act4<ACT_SOFTPLUS>(a, h); break;
    case ACT_GELU: act4<ACT_GELU>(a, h); break;
    case ACT_SELU: act4<ACT_SELU>(a, h); break;
    case ACT_SOFTSIGN: act4<ACT_SOFTSIGN>(a, h); break;
    case ACT_EXPONENTIAL: act4<ACT_EXPONENTIAL>(a, h); break;
    case ACT_HARD_SIGMOID: act4<ACT_HARD_SIGMOID>(a, h); break;
    default: act4<ACT_LINEAR>(a, h); break;
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_given_w(const float* __restrict__ x, const float* __restrict__ w,
                                                 float* __restrict__ u, long B, int si, int so, int nh, long po,
                                                 int act, int res, int nif_skip, float omega) {
  constexpr int LPR = N / 4;          // lanes per matrix row
  constexpr int RPC = 64 / LPR;       // rows per 256-float chunk
  constexpr int NCH = N * N / 256;    // chunks per matrix
  constexpr int CB = NCH < 16 ? NCH : 16;  // chunks in flight per batch
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane / LPR, cq = lane % LPR;
  const long nwaves = (long)gridDim.x * 4;
  const long s_w1 = 0, s_wh = (long)si * N, s_wl = s_wh + (long)nh * N * N, s_b1 = s_wl + (long)N * so;
  const long s_bh = s_b1 + N, s_bl = s_bh + (long)nh * N;

  for (long pt = (long)blockIdx.x * 4 + wid; pt < B; pt += nwaves) {
    const float* wp = w + pt * po;
    float h4[4], a4[4], ub[4];
    {  // first layer
      const f32x4u b = *reinterpret_cast<const f32x4u*>(wp + s_b1 + 4 * cq);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int d = 0; d < si; ++d) {
        const float xv = x[pt * si + d];
        const f32x4u wv = *reinterpret_cast<const f32x4u*>(wp + s_w1 + (long)d * N + 4 * cq);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, wv[c], acc[c]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) a4[c] = fmaf(omega, acc[c], b[c]);
      act4_dyn(act, a4, h4);
    }
    for (int j = 0; j < nh; ++j) {
      const float* wm = wp + s_wh + (long)j * N * N;
      const f32x4u b = *reinterpret_cast<const f32x4u*>(wp + s_bh + (long)j * N + 4 * cq);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < NCH; cb += CB) {
        f32x4u wv[CB];
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) wv[ch] = *reinterpret_cast<const f32x4u*>(wm + (long)(cb + ch) * 256 + lane * 4);
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) {
          // row = (cb+ch)*RPC + g ; it is component (row & 3) of the lane whose cq' = row >> 2
          const int rbase = (cb + ch) * RPC;          // compile-time
          const int comp0 = rbase & 3;                 // 0 (RPC>=4) or {0,2} (RPC=2)
          const int compi = (comp0 + g) & 3;
          const float hsel = compi == 0 ? h4[0] : compi == 1 ? h4[1] : compi == 2 ? h4[2] : h4[3];
          const int src = g * LPR + ((rbase + g) >> 2);
          const float hr = __shfl(hsel, src);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = fmaf(hr, wv[ch][c], acc[c]);
        }
      }
      // sum the partial dot products of the RPC row groups
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], off);
#pragma unroll
      for (int c = 0; c < 4; ++c) a4[c] = fmaf(omega, acc[c], b[c]);
      float hn[4];
      act4_dyn(act, a4, hn);
      const bool res_first = res && !(j & 1), res_second = res && (j & 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (res_first) ub[c] = h4[c];
        float v = hn[c];
        if (nif_skip) v += h4[c];
        if (res_second) v = 0.5f * (ub[c] + v);
        h4[c] = v;
      }
    }
    // last layer: u[o] = sum_f h[f] Wl[f][o] + bl[o]
    for (int o = 0; o < so; ++o) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) s = fmaf(h4[c], wp[s_wl + (long)(4 * cq + c) * so + o], s);
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off);
      if (lane == 0) u[pt * so + o] = s + wp[s_bl + o];
    }
  }
}

// generic width (n <= 128, any n): lane owns columns lane and lane+64, dword loads
__global__ __launch_bounds__(256) void k_given_w_generic(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ u, long B, int si, int so, int n, int nh,
                                                         long po, int act, int res, int nif_skip, float omega) {
  __shared__ float hs[4][128];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long nwaves = (long)gridDim.x * 4;
  const long s_w1 = 0, s_wh = (long)si * n, s_wl = s_wh + (long)nh * n * n, s_b1 = s_wl + (long)n * so;
  const long s_bh = s_b1 + n, s_bl = s_bh + (long)nh * n;
  float* hsw = hs[wid];
  for (long pt = (long)blockIdx.x * 4 + wid; pt < B; pt += nwaves) {
    const float* wp = w + pt * po;
    float hcur[2] = {0.f, 0.f}, ub[2] = {0.f, 0.f};
    for (int q = 0; q < 2; ++q) {
      const int f = lane + 64 * q;
      if (f < n) {
        float acc = 0.f;
        for (int d = 0; d < si; ++d) acc = fmaf(x[pt * si + d], wp[s_w1 + (long)d * n + f], acc);
        float a = fmaf(omega, acc, wp[s_b1 + f]), hv, dv;
        float a1[4] = {a, 0, 0, 0}, h1[4];
        act4_dyn(act, a1, h1);
        hv = h1[0]; (void)dv;
        hcur[q] = hv;
      }
    }
    for (int j = 0; j < nh; ++j) {
      hsw[lane] = hcur[0]; hsw[lane + 64] = hcur[1];
      __builtin_amdgcn_wave_barrier();
      const float* wm = wp + s_wh + (long)j * n * n;
      float acc[2] = {0.f, 0.f};
      for (int i = 0; i < n; ++i) {
        const float hi = hsw[i];
        if (lane < n) acc[0] = fmaf(hi, wm[(long)i * n + lane], acc[0]);
        if (lane + 64 < n) acc[1] = fmaf(hi, wm[(long)i * n + lane + 64], acc[1]);
      }
      __builtin_amdgcn_wave_barrier();
      const bool res_first = res && !(j & 1), res_second = res && (j & 1);
      for (int q = 0; q < 2; ++q) {
        const int f = lane + 64 * q;
        float v = 0.f;
        if (f < n) {
          float a1[4] = {fmaf(omega, acc[q], wp[s_bh + (long)j * n + f]), 0, 0, 0}, h1[4];
          act4_dyn(act, a1, h1);
          v = h1[0];
          if (res_first) ub[q] = hcur[q];
          if (nif_skip) v += hcur[q];
          if (res_second) v = 0.5f * (ub[q] + v);
        }
        hcur[q] = v;
      }
    }
    for (int o = 0; o < so; ++o) {
      float s = 0.f;
      if (lane < n) s = fmaf(hcur[0], wp[s_wl + (long)lane * so + o], s);
      if (lane + 64 < n) s = fmaf(hcur[1], wp[s_wl + (long)(lane + 64) * so + o], s);
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) u[pt * so + o] = s + wp[s_bl + o];
    }
  }
}

void launch_given_w(const float* x, const float* w, float* u, long B, int si, int so, int n, int nh, long po, int act,
                    int res, int nif_skip, float omega, hipStream_t st) {
  long blocks = (B + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  dim3 grid((unsigned)blocks), block(256);
  if (n == 32) hipLaunchKernelGGL((k_given_w<32>), grid, block, 0, st, x, w, u, B, si, so, nh, po, act, res, nif_skip, omega);
  else if (n == 64) hipLaunchKernelGGL((k_given_w<64>), grid, block, 0, st, x, w, u, B, si, so, nh, po, act, res, nif_skip, omega);
  else if (n == 128) hipLaunchKernelGGL((k_given_w<128>), grid, block, 0, st, x, w, u, B, si, so, nh, po, act, res, nif_skip, omega);
  else hipLaunchKernelGGL(k_given_w_generic, grid, block, 0, st, x, w, u, B, si, so, n, nh, po, act, res, nif_skip, omega);
}

// ============================================================================================
// model_lr_to_w: w[a][s] = sum_k lr[a][k] Wh[k][s] + bh[s]   (siren.py:514-522 / Dense model.py:220-230)
// A pure write stream (4*po bytes per point).  What decides its speed is the LENGTH of the contiguous bursts: 4 KB per
// workgroup per row (the r1 / early-r2 kernels, whatever the alignment) stops at 3.5 TB/s even with no arithmetic at all, one
// linear stream reaches 5.6 of the 6.4 TB/s hipMemset gets (tools/exp/l2w_probe.hip).  Two forms, both with Wh / bh staged in
// LDS by a 1024-thread workgroup: k_latent_to_w_flat (below) when the whole layer fits, else this one -- a column window as wide
// as the LDS takes, over a block of CONSECUTIVE rows.  Rows start at a * po floats with po odd: every store is still one 16-byte ALIGNED
// unit (unit U of row a = slots 4U - m .. 4U - m + 3, m = (a * po + buffer offset) mod 4, wave-uniform); the LDS reads are
// dword reads at the unaligned slot, the few units that hang over a row end are written slot by slot.
// ============================================================================================
#define NIF_L2W_T 1024
__global__ __launch_bounds__(NIF_L2W_T) void k_latent_to_w(const float* __restrict__ theta, long off_Wh, long off_bh, int r,
                                                          long po, const float* __restrict__ lr, long B,
                                                          float* __restrict__ w, int CU, int LW, long rows_per_block) {
  extern __shared__ float l2w_sm[];            // [(r+1)][LW]: planes k < r = Wh rows, plane r = bh; slot s = window element e = s - c0 + 3
  // r6: element e of a plane sits at (e & 3) * Q + (e >> 2), Q = LW / 4 (as in k_latent_to_w_flat): lane i reads element e0 + 4 i + c --
  // a stride of four dwords was a 4-way bank conflict on every read
  const int Q = LW >> 2;
  const long c0 = 4L * blockIdx.x * CU;        // first slot of this column window's units (row-relative, before the -m shift)
  for (int k = 0; k <= r; ++k) {
    const float* src = k < r ? theta + off_Wh + (long)k * po : theta + off_bh;
    float* dst = l2w_sm + (long)k * LW;
    for (int e = threadIdx.x; e < LW; e += NIF_L2W_T) {
      const long sc = c0 - 3 + e;
      dst[(int)__umul24(e & 3, Q) + (e >> 2)] = (sc >= 0 && sc < po) ? src[sc] : 0.f;
    }
  }
  __syncthreads();
  const int wmis = (int)((reinterpret_cast<size_t>(w) >> 2) & 3);       // misalignment of the buffer itself (floats)
  const long a0 = (long)blockIdx.y * rows_per_block;
  const long a1 = a0 + rows_per_block < B ? a0 + rows_per_block : B;
  const float* sB = l2w_sm + (long)r * LW;
  for (long a = a0; a < a1; ++a) {
    const int m = (int)((a * po + wmis) & 3);
    const float* lrow = lr + a * r;
    float* wrow = w + a * po;
    const long nun = (po + m + 3) / 4;                     // units of this row
    for (int ul = threadIdx.x; ul < CU; ul += NIF_L2W_T) {
      const long U = (long)blockIdx.x * CU + ul;
      if (U >= nun) break;
      const int e0 = 4 * ul - m + 3;                       // window element of the unit's first slot
      int ix[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ix[c] = (int)__umul24((e0 + c) & 3, Q) + ((e0 + c) >> 2);
      float acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = sB[ix[c]];
      for (int k = 0; k < r; ++k) {
        const float zk = lrow[k];
        const float* sW = l2w_sm + (long)k * LW;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(zk, sW[ix[c]], acc[c]);
      }
      const long s0 = 4 * U - m;
      float* dst = wrow + s0;                              // (a * po + s0) * 4 bytes is a multiple of 16 by construction
      if (s0 >= 0 && s0 + 4 <= po) {
        f32x4 v; v[0] = acc[0]; v[1] = acc[1]; v[2] = acc[2]; v[3] = acc[3];
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (s0 + c >= 0 && s0 + c < po) dst[c] = acc[c];
      }
    }
  }
}
// The whole hyper layer fits the LDS (cfg-2: 134 KB): the output is ONE array of B * po floats, cut into 16-byte aligned units;
// a workgroup streams a contiguous span of units, (row, column) of a unit by one division, the few units that straddle a row
// boundary pick up the next row's latent on the way.  5.6 TB/s where the row-walking form above stays at 3.5-4.1 (its bursts end
// at every row: 67 KB) -- tools/exp/l2w_probe.hip.
// R > 0: latent_dim known at compile time, the row's latent lives in registers (one load per unit, reloaded at a row
// boundary); R = 0: any latent_dim, loaded per element
// r6: (a) the LDS image is SWIZZLED -- column s of a plane sits at (s & 3) * Q + (s >> 2), Q = ceil(po / 4): lane i of a wave reads
// column sc0 + 4 i + c, a stride of four dwords = a 4-way bank conflict in the plain image (PMC r5: 4.1e8 SQ_LDS_BANK_CONFLICT cycles,
// 3.3x the kernel's busy cycles); swizzled, the 64 lanes of a read instruction touch 64 consecutive dwords.  (b) (row, column) of a
// thread's unit by ONE 64-bit division at its first unit, then incrementally (a 64-bit division per unit was ~100 VALU instructions
// of the ~128 a unit can afford at 5 TB/s).  (c) the latents of the block's rows (a contiguous span of the output = a few dozen rows)
// are staged in LDS too: gfx9 counts loads and stores in ONE in-order vmcnt, so a per-unit global load of lr made every iteration
// wait for the previous unit's STORE to be acknowledged -- 16 waves x 1 KB in flight per CU over ~2 k cycles of store latency is
// exactly the 8 B / clk / CU (4.9 TB/s) r5 measured; now nothing in the loop waits for vector memory.
#ifndef NIF_L2W_NT
#define NIF_L2W_NT 0      // 1: non-temporal stores of the output stream (measured r6: see DESIGN 5.6)
#endif
#ifndef NIF_L2W_NB
#define NIF_L2W_NB 4096   // workgroups = contiguous spans of the output
#endif
template <int R>
__global__ __launch_bounds__(NIF_L2W_T) void k_latent_to_w_flat(const float* __restrict__ theta, long off_Wh, long off_bh, int r_,
                                                               int po, const float* __restrict__ lr, long B,
                                                               float* __restrict__ w, long nunits, long span, int NR) {
  extern __shared__ float l2w_sm[];            // [(r+1)][4 Q]: planes k < r = Wh rows, plane r = bh; column s at (s & 3) Q + (s >> 2); then [NR][r] latents
  const int r = R > 0 ? R : r_;
  const int Q = (po + 3) >> 2, PS = 4 * Q;
  const int wmis = (int)((reinterpret_cast<size_t>(w) >> 2) & 3);       // misalignment of the buffer itself (floats)
  const long u0 = (long)blockIdx.x * span;
  const long u1 = u0 + span < nunits ? u0 + span : nunits;
  const long e0 = 4 * u0 - wmis;
  const long a_first = (e0 < 0 ? 0 : e0) / po;                          // first row this block touches (block-uniform)
  float* sL = l2w_sm + (r + 1) * PS;
  for (int k = 0; k <= r; ++k) {
    const float* src = k < r ? theta + off_Wh + (long)k * po : theta + off_bh;
    float* dst = l2w_sm + k * PS;
    for (int e = threadIdx.x; e < po; e += NIF_L2W_T) dst[(int)__umul24(e & 3, Q) + (e >> 2)] = src[e];
  }
  for (int i = threadIdx.x; i < NR * r; i += NIF_L2W_T) {
    const long row = a_first + i / r;
    sL[i] = row < B ? lr[a_first * r + i] : 0.f;
  }
  __syncthreads();
  const long n = B * (long)po;
  const float* sB = l2w_sm + r * PS;
  long u = u0 + threadIdx.x;
  if (u >= u1) return;
  long e = 4 * u - wmis;                       // flat index of the unit's first float (may be < 0 for u = 0)
  int a = (int)((e < 0 ? 0 : e) / po - a_first);       // row of the first float, relative to the block's first row
  int sc = (int)(e - (a_first + a) * po);              // its column (negative only for e < 0)
  const int da = (4 * NIF_L2W_T) / po, ds = (4 * NIF_L2W_T) - da * po;     // rows / columns a thread advances per iteration
  for (; u < u1; u += NIF_L2W_T) {
    if (sc >= 0 && sc + 4 <= po && e + 4 <= n) {
      // the common unit: four columns of ONE row, inside the buffer -- straight-line code (the general form below is a chain of
      // exec-masked blocks, one LDS round trip after the other: hipcc cannot hoist reads across its row-boundary tests)
      int idx[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) idx[c] = (int)__umul24((sc + c) & 3, Q) + ((sc + c) >> 2);
      f32x4 q;
#pragma unroll
      for (int c = 0; c < 4; ++c) q[c] = sB[idx[c]];
      if (R > 0) {
        float zr[R > 0 ? R : 1];
#pragma unroll
        for (int k = 0; k < R; ++k) zr[k] = sL[a * R + k];
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
          for (int c = 0; c < 4; ++c) q[c] = fmaf(zr[k], l2w_sm[k * PS + idx[c]], q[c]);
      } else {
        for (int k = 0; k < r; ++k) {
          const float zk = sL[a * r + k];
#pragma unroll
          for (int c = 0; c < 4; ++c) q[c] = fmaf(zk, l2w_sm[k * PS + idx[c]], q[c]);
        }
      }
#if NIF_L2W_NT
      __builtin_nontemporal_store(q, reinterpret_cast<f32x4*>(w + e));
#else
      *reinterpret_cast<f32x4*>(w + e) = q;
#endif
    } else {
      // a unit that straddles a row boundary (or an end of the buffer): element by element
      int ar = a, scc = sc;
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc = 0.f;
        if (scc >= 0 && a_first + ar < B) {
          const int idx = (int)__umul24(scc & 3, Q) + (scc >> 2);
          acc = sB[idx];
          for (int k = 0; k < r; ++k) acc = fmaf(sL[ar * r + k], l2w_sm[k * PS + idx], acc);
        }
        v[c] = acc;
        if (++scc == po) { scc = 0; ++ar; }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (e + c >= 0 && e + c < n) w[e + c] = v[c];
    }
    e += 4L * NIF_L2W_T;
    sc += ds; a += da;                         // (a unit that starts in front of the buffer has a = 0, sc < 0: the same step)
    if (sc >= po) { sc -= po; ++a; }
  }
}
void launch_latent_to_w(const float* theta, long off_Wh, long off_bh, int r, long po, const float* lr, long B, float* w,
                        hipStream_t st) {
  if ((size_t)(r + 1) * (4 * ((po + 3) / 4)) * sizeof(float) <= 144u * 1024u) {
    const long nunits = (B * po + 3 + 3) / 4;
    const long nb = NIF_L2W_NB;
    long span = ((nunits + nb - 1) / nb + NIF_L2W_T - 1) / NIF_L2W_T * NIF_L2W_T;
    // the block's rows of lr sit in LDS: at most 8 KB of them (small po: more, shorter spans)
    const long max_rows = (8 * 1024 / 4) / r;
    const long span_cap = ((max_rows - 2) * po / 4) / NIF_L2W_T * NIF_L2W_T;
    if (span > span_cap && span_cap >= NIF_L2W_T) span = span_cap;
    const int NR = (int)((4 * span + po - 1) / po + 2);
    const long nblk = (nunits + span - 1) / span;
    const size_t shm = sizeof(float) * ((size_t)(r + 1) * (4 * ((po + 3) / 4)) + (size_t)NR * r);
#define NIF_L2WF(R_)                                                                                                       \
    {                                                                                                                       \
      (void)hipFuncSetAttribute((const void*)k_latent_to_w_flat<R_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
      hipLaunchKernelGGL(k_latent_to_w_flat<R_>, dim3((unsigned)nblk), dim3(NIF_L2W_T), shm, st, theta, off_Wh, off_bh, r, (int)po, lr, \
                         B, w, nunits, span, NR);                                                                           \
    }
    switch (r) {
      case 1: NIF_L2WF(1) break;
      case 2: NIF_L2WF(2) break;
      case 3: NIF_L2WF(3) break;
      case 4: NIF_L2WF(4) break;
      default: NIF_L2WF(0) break;
    }
#undef NIF_L2WF
    return;
  }
  // units per column window: as many as fit 144 KB of LDS
  const long units_row = (po + 3 + 3) / 4;
  long CU = (144L * 1024 / 4 / (r + 1) - 8) / 4;
  if (CU > units_row) CU = units_row;
  const int LW = (int)(4 * CU + 8);
  const long ncw = (units_row + CU - 1) / CU;
  long nrb = 8192 / ncw; if (nrb < 1) nrb = 1; if (nrb > B) nrb = B;
  const long rpb = (B + nrb - 1) / nrb;
  nrb = (B + rpb - 1) / rpb;
  const size_t shm = sizeof(float) * (size_t)(r + 1) * LW;
  (void)hipFuncSetAttribute((const void*)k_latent_to_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL(k_latent_to_w, dim3((unsigned)ncw, (unsigned)nrb), dim3(NIF_L2W_T), shm, st, theta, off_Wh, off_bh, r, po, lr, B,
                     w, (int)CU, LW, rpb);
}

// ============================================================================================
// gradient rows -> flat gradient (fixed summation order), loss partials -> g[P]
// ============================================================================================
__global__ __launch_bounds__(512) void k_reduce(const float* __restrict__ partial, long pstride, int rows,
                                                const float* __restrict__ lossp, int nloss, float* __restrict__ g, long P) {
  // block = 64 columns x 8 row groups, four independent partial sums per thread (the kernel is a latency-bound
  // stream of <= 256 rows: more loads in flight, not more bandwidth, is what it needs); fixed order => deterministic
  __shared__ float red[8][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < P) {
    const float* p = partial + i;
    int rrow = rg;
    for (; rrow + 24 < rows; rrow += 32) {
      s0 += p[(long)rrow * pstride]; s1 += p[(long)(rrow + 8) * pstride];
      s2 += p[(long)(rrow + 16) * pstride]; s3 += p[(long)(rrow + 24) * pstride];
    }
    for (; rrow < rows; rrow += 8) s0 += p[(long)rrow * pstride];
  }
  red[rg][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && i < P)
    g[i] = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
  if (blockIdx.x == gridDim.x - 1) {
    // the loss: every thread sums a strided subset in a fixed order, then a fixed tree
    __syncthreads();
    float ls = 0.f;
    for (int b = threadIdx.x; b < nloss; b += 512) ls += lossp[b];
    red[rg][col] = ls;
    __syncthreads();
    if (threadIdx.x < 64) {
      float v = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (threadIdx.x == 0) g[P] = v;
    }
  }
}
void launch_reduce(const float* partial, long pstride, int rows, const float* loss_partial, int nloss, float* g, long P,
                   hipStream_t st) {
  dim3 grid((unsigned)((P + 63) / 64)), block(512);
  hipLaunchKernelGGL(k_reduce, grid, block, 0, st, partial, pstride, rows, loss_partial, nloss, g, P);
}

// k_reduce with the Adam update of column i behind its sum (r6: the deferred row reduction of a plain single-GPU step; the same
// summation order and the same update expressions as k_reduce + k_adam: bit-identical results, g is still written)
__global__ __launch_bounds__(512) void k_reduce_adam(const float* __restrict__ partial, long pstride, int rows,
                                                     const float* __restrict__ lossp, int nloss, float* __restrict__ g, long P,
                                                     float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                                                     float lr_t, float b1, float b2, float eps) {
  __shared__ float red[8][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < P) {
    const float* p = partial + i;
    int rrow = rg;
    for (; rrow + 24 < rows; rrow += 32) {
      s0 += p[(long)rrow * pstride]; s1 += p[(long)(rrow + 8) * pstride];
      s2 += p[(long)(rrow + 16) * pstride]; s3 += p[(long)(rrow + 24) * pstride];
    }
    for (; rrow < rows; rrow += 8) s0 += p[(long)rrow * pstride];
  }
  red[rg][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && i < P) {
    const float gi = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
    g[i] = gi;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
  if (blockIdx.x == gridDim.x - 1) {
    __syncthreads();
    float ls = 0.f;
    for (int b = threadIdx.x; b < nloss; b += 512) ls += lossp[b];
    red[rg][col] = ls;
    __syncthreads();
    if (threadIdx.x < 64) {
      float vv = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
      for (int off = 32; off > 0; off >>= 1) vv += __shfl_down(vv, off);
      if (threadIdx.x == 0) g[P] = vv;
    }
  }
}
void launch_reduce_adam(const float* partial, long pstride, int rows, const float* loss_partial, int nloss, float* g, long P,
                        float* theta, float* m, float* v, float lr_t, float b1, float b2, float eps, hipStream_t st) {
  dim3 grid((unsigned)((P + 63) / 64)), block(512);
  hipLaunchKernelGGL(k_reduce_adam, grid, block, 0, st, partial, pstride, rows, loss_partial, nloss, g, P, theta, m, v, lr_t, b1, b2, eps);
}

// Keras-2.11 Adam (SURVEY a-11): lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host
__global__ void k_adam(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long P, float lr_t, float b1, float b2, float eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float gi = g[i];
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
  const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
  m[i] = mi; v[i] = vi;
  theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
}
// The same update with the hyper-parameters and the iteration count in DEVICE memory (AdamDev: lr, beta1, beta2, eps, step): what a
// captured hipGraph of training steps needs -- a replayed launch cannot carry this step's lr_t as a kernel argument.  Every block
// forms lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), t = step + 1, in fp64 like the host does; k_adam_step_inc bumps the counter behind it.
__global__ void k_adam_dev(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long P,
                           const AdamDev* __restrict__ ad) {
  __shared__ float lr_s;
  if (threadIdx.x == 0) {
    const double t = (double)(ad->step + 1);
    lr_s = (float)((double)ad->lr * sqrt(1.0 - pow((double)ad->beta2, t)) / (1.0 - pow((double)ad->beta1, t)));
  }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float b1 = ad->beta1, b2 = ad->beta2, eps = ad->eps, lr_t = lr_s;
  const float gi = g[i];
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
  const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
  m[i] = mi; v[i] = vi;
  theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
}
__global__ void k_adam_step_inc(AdamDev* ad) { ad->step += 1; }
void launch_adam_dev(float* theta, const float* g, float* m, float* v, long P, AdamDev* ad, hipStream_t st) {
  dim3 grid((unsigned)((P + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_adam_dev, grid, block, 0, st, theta, g, m, v, P, ad);
  hipLaunchKernelGGL(k_adam_step_inc, dim3(1), dim3(1), 0, st, ad);
}
void launch_adam(float* theta, const float* g, float* m, float* v, long P, float lr_t, float b1, float b2, float eps,
                 hipStream_t st) {
  dim3 grid((unsigned)((P + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_adam, grid, block, 0, st, theta, g, m, v, P, lr_t, b1, b2, eps);
}

// rows [B][c]  <->  tiles [ceil(B/32)][c][32]
__global__ void k_rows_to_tiles(const float* __restrict__ rows, long B, int c, float* __restrict__ tiles) {
  const long ntot = ((B + 31) / 32) * 32 * c;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < ntot; idx += (long)gridDim.x * blockDim.x) {
    const long t = idx / (32L * c);
    const int rem = (int)(idx - t * 32L * c);
    const int cc = rem / 32, p = rem % 32;
    long pt = t * 32 + p;
    if (pt >= B) pt = B - 1;
    tiles[idx] = rows[pt * c + cc];
  }
}
__global__ void k_tiles_to_rows(const float* __restrict__ tiles, long B, int c, float* __restrict__ rows) {
  const long ntot = B * c;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < ntot; idx += (long)gridDim.x * blockDim.x) {
    const long pt = idx / c;
    const int cc = (int)(idx - pt * c);
    rows[idx] = tiles[((pt / 32) * c + cc) * 32 + (pt % 32)];
  }
}
void launch_rows_to_tiles(const float* rows, long B, int c, float* tiles, hipStream_t st) {
  const long ntot = ((B + 31) / 32) * 32 * c;
  long grid = (ntot + 255) / 256; if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_rows_to_tiles, dim3((unsigned)grid), dim3(256), 0, st, rows, B, c, tiles);
}
void launch_tiles_to_rows(const float* tiles, long B, int c, float* rows, hipStream_t st) {
  const long ntot = B * c;
  long grid = (ntot + 255) / 256; if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_tiles_to_rows, dim3((unsigned)grid), dim3(256), 0, st, tiles, B, c, rows);
}

// ============================================================================================
// last-layer-parameterised class: u = Dot(axes=(2,1))([phi, a]) + last_layer_bias
// (nif/model.py:1264-1269), Keras 'mse', and the adjoint w.r.t. phi, a and (through the r x r last
// ParameterNet layer) the latent.  One thread per point; everything is [tile][c][32] so accesses coalesce.
// ============================================================================================
template <bool TRAIN>
__global__ __launch_bounds__(256) void k_ll_out(LLArgs A) {
  __shared__ float lsum[4];
  const long pt = (long)blockIdx.x * 256 + threadIdx.x;
  const long ntiles = (A.B + 31) / 32;
  const long tile = pt >> 5;
  const int p = (int)(pt & 31);
  const int r = A.r, so = A.so;
  float loss_lane = 0.f;
  if (tile < ntiles) {
    const bool valid = pt < A.B;
    const long ptc = valid ? pt : A.B - 1;
    const float* phi = A.PHI + tile * (long)(so * r) * 32 + p;
    const float* z = A.Z + tile * (long)r * 32 + p;
    const float wsamp = TRAIN ? (valid ? (A.sw ? A.sw[ptc] : 1.0f) : 0.0f) : 0.f;
    float se = 0.f;
    for (int s = 0; s < so; ++s) {
      float u = A.theta[A.bias_off + s];
      for (int j = 0; j < r; ++j) u = fmaf(phi[(s * r + j) * 32], z[j * 32], u);
      if (valid && A.u_out) A.u_out[pt * so + s] = u;
      if (TRAIN) {
        const float e = u - A.y[ptc * so + s];
        NIF_LOSS_ACC(A.loss_kind, e, se, dfac)
        const float du = dfac * wsamp * A.inv_bg / (float)so;
        A.DU[(tile * so + s) * 32 + p] = du;
        for (int j = 0; j < r; ++j) A.DPHI[(tile * (long)(so * r) + s * r + j) * 32 + p] = du * z[j * 32];
      }
    }
    if (TRAIN) {
      loss_lane = wsamp * se / (float)so * A.inv_bg;
      for (int j = 0; j < r; ++j) {
        float da = 0.f;
        for (int s = 0; s < so; ++s) da = fmaf(phi[(s * r + j) * 32], A.DU[(tile * so + s) * 32 + p], da);
        A.DA[(tile * r + j) * 32 + p] = da;
      }
      for (int k = 0; k < r; ++k) {
        float dz = 0.f;
        for (int c = 0; c < r; ++c) dz = fmaf(A.DA[(tile * r + c) * 32 + p], A.theta[A.last_w + (long)k * r + c], dz);
        A.DZL[(tile * r + k) * 32 + p] = dz;
      }
    }
  }
  if (TRAIN) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) loss_lane += __shfl_down(loss_lane, off);
    if (lane == 0) lsum[wid] = loss_lane;
    __syncthreads();
    if (threadIdx.x == 0) A.loss_partial[blockIdx.x] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  }
}
void launch_ll_out(const LLArgs& a, bool train, hipStream_t st) {
  const long ntiles = (a.B + 31) / 32;
  dim3 grid((unsigned)((ntiles * 32 + 255) / 256)), block(256);
  if (train) hipLaunchKernelGGL((k_ll_out<true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_ll_out<false>), grid, block, 0, st, a);
}

// ============================================================================================
// weight regularisers (Keras L1 / L2 on the ParameterNet variables, nif/model.py:109-117): gradient term and
// loss term, deterministic single-block reduction of the penalty
// ============================================================================================
__global__ __launch_bounds__(1024) void k_reg(const float* __restrict__ theta, float* __restrict__ g, long lo, long hi,
                                              long P, float l1, float l2) {
  __shared__ float red[1024];
  float pen = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 1024) {
    const float w = theta[i];
    g[i] += 2.0f * l2 * w + (w > 0.f ? l1 : (w < 0.f ? -l1 : 0.f));
    pen += l2 * w * w + l1 * fabsf(w);
  }
  red[threadIdx.x] = pen;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) g[P] += red[0];
}
void launch_reg(const float* theta, float* g, long lo, long hi, long P, float l1, float l2, hipStream_t st) {
  hipLaunchKernelGGL(k_reg, dim3(1), dim3(1024), 0, st, theta, g, lo, hi, P, l1, l2);
}
__global__ void k_metric(const float* __restrict__ g, long P, float weight, double* __restrict__ acc) {
  acc[0] += (double)weight * (double)g[P];
  acc[1] += (double)weight;
}
void launch_metric(const float* g, long P, float weight, double* acc, hipStream_t st) {
  hipLaunchKernelGGL(k_metric, dim3(1), dim3(1), 0, st, g, P, weight, acc);
}

// ============================================================================================
// device-side shuffle of a resident point table: dst[i][:] = src[perm[i]][:]  (rows of ncol floats).  Model.fit keeps
// the table in HBM and uploads only the epoch's permutation (4 bytes per row) instead of re-gathering and re-uploading
// the table on the host every epoch.  One thread per output element: consecutive threads write consecutive floats.
// ============================================================================================
__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ perm, long n, int ncol,
                              float* __restrict__ dst) {
  const long total = n * ncol;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long i = idx / ncol;
    const int c = (int)(idx - i * ncol);
    dst[idx] = src[(long)perm[i] * ncol + c];
  }
}
void launch_gather_rows(const float* src, const int* perm, long n, int ncol, float* dst, hipStream_t st) {
  const long total = n * ncol;
  long grid = (total + 255) / 256; if (grid > 16384) grid = 16384; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)grid), dim3(256), 0, st, src, perm, n, ncol, dst);
}

// ============================================================================================
// Activity regulariser of the ParameterNet output (reference nif/model.py:118-125, :226, :659, :731: Keras
// `activity_regularizer=L1(l1)` or `L2(l2)` on the last ParameterNet layer): loss += c/B * sum_a sum_i phi(out_ai) with
// out_ai = sum_k zt_k(a) M^(k)_i the NEVER MATERIALISED pnet_output, phi = |.| (L1) or (.)^2 (L2), Keras dividing the
// activity loss by the batch size.  Two passes over the virtual [B, po] tensor, each recomputing out on the fly:
//   k_actreg_points   one thread per point, loop over the po outputs (the plane rows are wave-uniform loads):
//                     loss partial and dL/dz_k(a) += c/B sum_i phi'(out_ai) M^(k)_i        (before the ParameterNet adjoint)
//   k_actreg_planes   one thread per output i, loop over a slab of points (their zt staged in LDS):
//                     partial[slab][k][i] = sum_{a in slab} phi'(out_ai) zt_k(a)           (-> k_actreg_apply adds c/B * sum)
// Cost 4 (r+1) po flop per point (67 kflop at 4x64): an optional regulariser, not the benchmark path.
// ============================================================================================
// MAXR: register / LDS vectors sized for latent_dim <= 8 (the common case) or <= 64 (everything nif_create accepts; r3)
template <bool L1, int MAXR>
__global__ __launch_bounds__(256) void k_actreg_points(const float* __restrict__ theta, long off_W, long off_b, int r, long po,
                                                       const float* __restrict__ Z, long B, float coef, float* __restrict__ DZ,
                                                       float* __restrict__ loss_partial) {
  __shared__ float red[256];
  const long a = (long)blockIdx.x * 256 + threadIdx.x;
  float zt[MAXR], dz[MAXR];
  const bool ok = a < B;
  const long tile = a >> 5; const int pp = (int)(a & 31);
#pragma unroll
  for (int k = 0; k < MAXR; ++k) { zt[k] = (ok && k < r) ? Z[(tile * r + k) * 32 + pp] : 0.f; dz[k] = 0.f; }
  float acc = 0.f;
  for (long i = 0; i < po; ++i) {
    float out = theta[off_b + i];
#pragma unroll
    for (int k = 0; k < MAXR; ++k)
      if (k < r) out = fmaf(zt[k], theta[off_W + (long)k * po + i], out);
    const float d = L1 ? (out > 0.f ? 1.f : (out < 0.f ? -1.f : 0.f)) : 2.0f * out;
    acc += L1 ? fabsf(out) : out * out;
#pragma unroll
    for (int k = 0; k < MAXR; ++k)
      if (k < r) dz[k] = fmaf(d, theta[off_W + (long)k * po + i], dz[k]);
  }
  if (ok)
    for (int k = 0; k < r; ++k) DZ[(tile * r + k) * 32 + pp] += coef * dz[k];
  red[threadIdx.x] = ok ? acc : 0.f;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_partial[blockIdx.x] = coef * red[0];
}
template <bool L1, int MAXR>
__global__ __launch_bounds__(256) void k_actreg_planes(const float* __restrict__ theta, long off_W, long off_b, int r, long po,
                                                       const float* __restrict__ Z, long B, long slab, float* __restrict__ part) {
  __shared__ float zs[MAXR * 256];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long a0 = (long)blockIdx.y * slab, a1 = a0 + slab < B ? a0 + slab : B;
  float m[MAXR + 1], g[MAXR + 1];
#pragma unroll
  for (int k = 0; k <= MAXR; ++k) { m[k] = 0.f; g[k] = 0.f; }
  if (i < po) {
#pragma unroll
    for (int k = 0; k < MAXR; ++k) if (k < r) m[k] = theta[off_W + (long)k * po + i];
    m[MAXR] = theta[off_b + i];
  }
  for (long c0 = a0; c0 < a1; c0 += 256) {
    __syncthreads();
    const long a = c0 + threadIdx.x;
    for (int k = 0; k < r; ++k) zs[k * 256 + threadIdx.x] = a < a1 ? Z[((a >> 5) * r + k) * 32 + (a & 31)] : 0.f;
    __syncthreads();
    const int cnt = (int)(a1 - c0 < 256 ? a1 - c0 : 256);
    for (int q = 0; q < cnt; ++q) {
      float out = m[MAXR];
#pragma unroll
      for (int k = 0; k < MAXR; ++k) if (k < r) out = fmaf(zs[k * 256 + q], m[k], out);
      const float d = L1 ? (out > 0.f ? 1.f : (out < 0.f ? -1.f : 0.f)) : 2.0f * out;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) if (k < r) g[k] = fmaf(d, zs[k * 256 + q], g[k]);
      g[MAXR] += d;
    }
  }
  if (i < po) {
    float* row = part + (long)blockIdx.y * (r + 1) * po;
    for (int k = 0; k < r; ++k) row[(long)k * po + i] = g[k];
    row[(long)r * po + i] = g[MAXR];
  }
}
// g[hyper kernel rows | hyper bias] += coef * sum over slabs (fixed order); g[P] += sum of the loss partials
__global__ __launch_bounds__(256) void k_actreg_apply(const float* __restrict__ part, int nslab, int r, long po, float coef,
                                                      long off_W, long off_b, const float* __restrict__ loss_partial, int nloss,
                                                      float* __restrict__ g, long P) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;       // element of [(r+1)][po]
  if (e < (long)(r + 1) * po) {
    float s = 0.f;
    for (int sl = 0; sl < nslab; ++sl) s += part[(long)sl * (r + 1) * po + e];
    const long k = e / po, i = e - k * po;
    g[(k < r ? off_W + k * po : off_b) + i] += coef * s;
  }
  if (blockIdx.x == 0) {
    __shared__ float red[256];
    float ls = 0.f;
    for (int b = threadIdx.x; b < nloss; b += 256) ls += loss_partial[b];
    red[threadIdx.x] = ls;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) g[P] += red[0];
  }
}
int actreg_max_r() { return 64; }
void launch_actreg_points(bool l1, const float* theta, long off_W, long off_b, int r, long po, const float* Z, long B, float coef,
                          float* DZ, float* loss_partial, hipStream_t st) {
  dim3 grid((unsigned)((B + 255) / 256)), block(256);
  if (r <= 8) {
    if (l1) hipLaunchKernelGGL((k_actreg_points<true, 8>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, coef, DZ, loss_partial);
    else hipLaunchKernelGGL((k_actreg_points<false, 8>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, coef, DZ, loss_partial);
  } else {
    if (l1) hipLaunchKernelGGL((k_actreg_points<true, 64>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, coef, DZ, loss_partial);
    else hipLaunchKernelGGL((k_actreg_points<false, 64>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, coef, DZ, loss_partial);
  }
}
void launch_actreg_planes(bool l1, const float* theta, long off_W, long off_b, int r, long po, const float* Z, long B, int nslab,
                          float* part, hipStream_t st) {
  const long slab = ((B + nslab - 1) / nslab + 255) / 256 * 256;
  dim3 grid((unsigned)((po + 255) / 256), (unsigned)nslab), block(256);
  if (r <= 8) {
    if (l1) hipLaunchKernelGGL((k_actreg_planes<true, 8>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, slab, part);
    else hipLaunchKernelGGL((k_actreg_planes<false, 8>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, slab, part);
  } else {
    if (l1) hipLaunchKernelGGL((k_actreg_planes<true, 64>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, slab, part);
    else hipLaunchKernelGGL((k_actreg_planes<false, 64>), grid, block, 0, st, theta, off_W, off_b, r, po, Z, B, slab, part);
  }
}
void launch_actreg_apply(const float* part, int nslab, int r, long po, float coef, long off_W, long off_b, const float* loss_partial,
                         int nloss, float* g, long P, hipStream_t st) {
  const long ne = (long)(r + 1) * po;
  hipLaunchKernelGGL(k_actreg_apply, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, part, nslab, r, po, coef, off_W, off_b,
                     loss_partial, nloss, g, P);
}

// Activity regulariser on the last-layer class: the ParameterNet output IS the small tensor a [B, r] (model.py:583-585), so the
// term c/B sum phi(a) is one pass over the points: dL/da += c/B phi'(a) in front of the r x r layer's gradient, dL/dlatent +=
// (that) last_w^T in front of the ParameterNet adjoint, per-block loss partials.  phi = |.| (l1) or (.)^2.
__global__ __launch_bounds__(256) void k_ll_actreg(const float* __restrict__ Za, const float* __restrict__ lw, int r, long B, float coef,
                                                   int l1, float* __restrict__ DA, float* __restrict__ DZL,
                                                   float* __restrict__ loss_partial) {
  __shared__ float red[256];
  const long pt = (long)blockIdx.x * 256 + threadIdx.x;
  float ls = 0.f;
  if (pt < B) {
    const long base = (pt >> 5) * r * 32 + (pt & 31);
    for (int c2 = 0; c2 < r; ++c2) {
      float t = 0.f;
      for (int cc = 0; cc < r; ++cc) {
        const float a = Za[base + (long)cc * 32];
        const float g = coef * (l1 ? (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f)) : 2.0f * a);
        t = fmaf(g, lw[c2 * r + cc], t);
        if (c2 == 0) { DA[base + (long)cc * 32] += g; ls += coef * (l1 ? fabsf(a) : a * a); }
      }
      DZL[base + (long)c2 * 32] += t;
    }
  }
  red[threadIdx.x] = ls;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_partial[blockIdx.x] = red[0];
}
int launch_ll_actreg(const float* Za, const float* lw, int r, long B, float coef, bool l1, float* DA, float* DZL, float* loss_partial,
                     hipStream_t st) {
  const int nblk = (int)((B + 255) / 256);
  hipLaunchKernelGGL(k_ll_actreg, dim3(nblk), dim3(256), 0, st, Za, lw, r, B, coef, l1 ? 1 : 0, DA, DZL, loss_partial);
  return nblk;
}
// dst[0] += sum(parts[0..n)) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void k_add_sum(const float* __restrict__ parts, int n, float* __restrict__ dst) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += parts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[0] += red[0];
}
void launch_add_sum(const float* parts, int n, float* dst, hipStream_t st) { hipLaunchKernelGGL(k_add_sum, dim3(1), dim3(256), 0, st, parts, n, dst); }

// g[0..ncols) += tmp[0..ncols) ,  g[P] += tmp[ncols]   (a side pass's reduced columns and loss on top of the main gradient)
__global__ void k_axpy_cols(float* __restrict__ g, const float* __restrict__ tmp, long ncols, long P) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncols) g[i] += tmp[i];
  else if (i == ncols) g[P] += tmp[ncols];
}
void launch_axpy_cols(float* g, const float* tmp, long ncols, long P, hipStream_t st) {
  hipLaunchKernelGGL(k_axpy_cols, dim3((unsigned)((ncols + 1 + 255) / 256)), dim3(256), 0, st, g, tmp, ncols, P);
}

// ============================================================================================
// HessianLayer epilogues on the device (r3; r2 gathered / contracted on the host).
//   k_hess_gather   NIF / NIFMultiScale: rows y_idx of the kernels' [B][so][nx] / [B][so][nx][nx] results -> [B][ny][nx] / [B][ny][nx][nx]
//   k_through_lw    last-layer class: a' = z' last_w for a stack of latent-layout vectors (z' = dz/dp_j, z'' = d2z/dp_j dp_k)
//   k_ll_hess       last-layer class: u = phi.a + bias; the coordinates only move phi, the parameters only move a:
//                   du/dx = phi'_x.a, du/dp = phi.a'_p, d2u/dx dx' = phi''.a, d2u/dx dp = phi'_x.a'_p, d2u/dp dp' = phi.a''
// ============================================================================================
__global__ __launch_bounds__(256) void k_hess_gather(const float* __restrict__ fj, const float* __restrict__ fh, long B, int so, int nx,
                                                     HessIdx I, float* __restrict__ dydx, float* __restrict__ d2) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;      // (point, i)
  if (e >= B * I.ny) return;
  const long a = e / I.ny; const int i = (int)(e - a * I.ny);
  const long src = a * so + I.y_idx[i];
  for (int j = 0; j < nx; ++j) {
    dydx[e * nx + j] = fj[src * nx + j];
    for (int k = 0; k < nx; ++k) d2[(e * nx + j) * nx + k] = fh[(src * nx + j) * nx + k];
  }
}
void launch_hess_gather(const float* fj, const float* fh, long B, int so, int nx, const HessIdx& I, float* dydx, float* d2, hipStream_t st) {
  hipLaunchKernelGGL(k_hess_gather, dim3((unsigned)((B * I.ny + 255) / 256)), dim3(256), 0, st, fj, fh, B, so, nx, I, dydx, d2);
}
__global__ __launch_bounds__(256) void k_through_lw(const float* __restrict__ SRC, const long* __restrict__ src_off, int nvec, const float* __restrict__ lw,
                                                    int rl, long npts, float* __restrict__ DST) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;      // (vector, padded point)
  if (e >= (long)nvec * npts) return;
  const int v = (int)(e / npts); const long a = e - (long)v * npts;
  const long zoff = (a >> 5) * rl * 32 + (a & 31);
  const float* z = SRC + src_off[v];
  float* d = DST + (long)v * npts * rl;
  for (int cc = 0; cc < rl; ++cc) {
    float t = 0.f;
    for (int c2 = 0; c2 < rl; ++c2) t = fmaf(z[zoff + (long)c2 * 32], lw[c2 * rl + cc], t);
    d[zoff + (long)cc * 32] = t;
  }
}
void launch_through_lw(const float* SRC, const long* src_off_dev, int nvec, const float* lw, int rl, long npts, float* DST, hipStream_t st) {
  hipLaunchKernelGGL(k_through_lw, dim3((unsigned)(((long)nvec * npts + 255) / 256)), dim3(256), 0, st, SRC, src_off_dev, nvec, lw, rl, npts, DST);
}
__global__ __launch_bounds__(256) void k_ll_hess(HessLLArgs H) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;      // (point, i)
  const int ny = H.I.ny, nx = H.nx, rl = H.rl, so = H.so, nxc = H.nxc > 0 ? H.nxc : 1, np = H.np;
  if (e >= H.B * ny) return;
  const long a = e / ny; const int i = (int)(e - a * ny);
  const long zoff = (a >> 5) * rl * 32 + (a & 31);
  const long npts = H.npts;
  const long sop = (long)so * rl;
  auto av = [&](int c) { return H.Za[zoff + (long)c * 32]; };
  auto ap = [&](int j, int c) { return H.AP[((long)j * npts) * rl + zoff + (long)c * 32]; };           // a'_j
  auto app = [&](int j, int k, int c) {                                                                // a''_{jk} (stored for j <= k)
    const int lo = j < k ? j : k, hi = j < k ? k : j;
    return H.AP[((long)(np + lo * np + hi) * npts) * rl + zoff + (long)c * 32];
  };
  if (i == 0)
    for (int o = 0; o < so; ++o) {
      float t = H.bias[o];
      for (int c = 0; c < rl; ++c) t = fmaf(H.f0[a * sop + (long)o * rl + c], av(c), t);
      H.y[a * so + o] = t;
    }
  const long src = a * sop + (long)H.I.y_idx[i] * rl;        // phi[a, y_i, 0]
  float* dy = H.dydx + e * nx;
  float* h = H.d2 + e * nx * nx;
  for (int j = 0; j < H.nxc; ++j) {
    float t = 0.f;
    for (int c = 0; c < rl; ++c) t = fmaf(H.fj[(src + c) * nxc + j], av(c), t);
    dy[H.xc[j]] = t;
    for (int k = 0; k < H.nxc; ++k) {
      float t2 = 0.f;
      for (int c = 0; c < rl; ++c) t2 = fmaf(H.fh[((src + c) * nxc + j) * nxc + k], av(c), t2);
      h[H.xc[j] * nx + H.xc[k]] = t2;
    }
    for (int k = 0; k < np; ++k) {
      float t2 = 0.f;
      for (int c = 0; c < rl; ++c) t2 = fmaf(H.fj[(src + c) * nxc + j], ap(k, c), t2);
      h[H.xc[j] * nx + H.xp[k]] = t2; h[H.xp[k] * nx + H.xc[j]] = t2;
    }
  }
  for (int j = 0; j < np; ++j) {
    float t = 0.f;
    for (int c = 0; c < rl; ++c) t = fmaf(H.f0[src + c], ap(j, c), t);
    dy[H.xp[j]] = t;
    for (int k = 0; k < np; ++k) {
      float t2 = 0.f;
      for (int c = 0; c < rl; ++c) t2 = fmaf(H.f0[src + c], app(j, k, c), t2);
      h[H.xp[j] * nx + H.xp[k]] = t2;
    }
  }
}
void launch_ll_hess(const HessLLArgs& H, hipStream_t st) {
  hipLaunchKernelGGL(k_ll_hess, dim3((unsigned)((H.B * H.I.ny + 255) / 256)), dim3(256), 0, st, H);
}

