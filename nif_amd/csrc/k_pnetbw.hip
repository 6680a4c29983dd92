// k_pnetbw.hip -- ParameterNet adjoint AND its weight gradients in one pass, without any HBM stash.
//
// The ParameterNet is tiny (2 x 32 in the benchmark config) but, run as forward-with-stash + adjoint + four
// gradient-reduction launches, it moved 2.3 KB per point through HBM -- a quarter of the whole step's time
// for 2 % of its arithmetic.  Here every wave recomputes the forward pass of its 32-point tile from the 4-12
// input bytes, keeps each layer input in a PRIVATE LDS copy of the stash tile ([feature][32 points], the
// layout the K = batch gradient GEMM wants: a lane reads 16 consecutive points of one feature), runs the
// adjoint in registers and accumulates dL/dW of every layer in MFMA accumulators that live for the whole
// kernel.  HBM traffic: the input columns, dL/dz (4r bytes) -- and one partial-gradient row per workgroup.
//
// Built for nst <= 32 (one 32-feature block) and at most two hidden matrices (two Dense / SIREN layers, or
// one MLP_ResNet / SIREN_ResNet block); everything else takes the stash path (k_nets.hip + k_gw.hip).
// Same math as k_pnet / k_pnet_bwd (reference nif/layers/mlp.py:62-79, :148-160, siren.py:256-281, :381-410).
#include "nif_internal.h"
#include "k_pnet_bf16.h"

#ifndef NIF_PBW_DENSE_BF16
#define NIF_PBW_DENSE_BF16 1   // 0: f32-input MFMAs for the dense products (A/B builds)
#endif

struct PbwArgs {
  PNetArgs p;
  float* partial; long pstride;   // partial[row * pstride + theta index], row = blockIdx.x
  // Optional: a stash slot the NEXT kernel streams ([tiles][touch_floats]).  This kernel is compute-bound and leaves HBM
  // idle, while the ≈256 MB of stash writes that k_snet4 left dirty in the Infinity Cache would otherwise be written
  // back under the first streaming kernel after it (+35-50 us there).  Each wave touches one word per 64 bytes of its
  // tile's slice: the lines are pulled in (evicting the dirty ones now) and the next kernel finds them cached.
  const float* touch; long touch_floats;
};

__device__ __forceinline__ f32x4 lds4(const float* q) { return *reinterpret_cast<const f32x4*>(q); }

// r6: the wave-private LDS tiles [feature][32 points] are SWIZZLED -- the 16-byte column piece c of feature row f sits at piece
// c ^ (f & 7).  The gradient GEMM reads lane (i, hf) <- 16 consecutive points of feature i with ds_read_b128 at a row stride of
// 128 bytes: in the plain layout every second lane hits the same four banks (PMC r5: 1.45e7 SQ_LDS_BANK_CONFLICT cycles in this
// kernel); swizzled, eight consecutive lanes cover all 64 banks.  The dword stores of a register tile (32 consecutive points of one
// feature per half wave) stay conflict free: the swizzle permutes pieces inside a row.
__device__ __forceinline__ int pswz(int f, int p) { return f * 32 + ((((p >> 2) ^ (f & 7))) << 2) + (p & 3); }
__device__ __forceinline__ int pswz4(int f, int c) { return f * 32 + ((c ^ (f & 7)) << 2); }     // piece c (4 points) of row f
__device__ __forceinline__ void tile_store(float* __restrict__ t, const f32x16& h, int p, int hf) {
#pragma unroll
  for (int v = 0; v < 16; ++v) t[pswz(fmap(v, hf), p)] = h[v];
}

// C[in][out] += sum_p IN[p][in] * DA[p][out] over the 32 points of the tile; IN, DA are LDS tiles [feature][32].
// Gradient path: bf16 hi/lo splits, three v_mfma_f32_32x32x16_bf16 per 16 points (k_gw.hip) instead of 8 f32-input MFMAs
typedef __bf16 pbw_bf16x8 __attribute__((ext_vector_type(8)));
#ifndef NIF_PBW_BF16
#define NIF_PBW_BF16 1
#endif
// in_rows < 32: IN holds only that many feature rows (the X^T tile of the first layer): the others count as zero
__device__ __forceinline__ void grad_mfma(const float* IN, const float* DA, f32x16& C, int i, int hf, int in_rows = 32) {
  f32x4 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q] = lds4(IN + pswz4(i < in_rows ? i : 0, 4 * hf + q));
    if (i >= in_rows) { a[q][0] = 0.f; a[q][1] = 0.f; a[q][2] = 0.f; a[q][3] = 0.f; }
    b[q] = lds4(DA + pswz4(i, 4 * hf + q));
  }
#if NIF_PBW_BF16
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    pbw_bf16x8 ah, al, bh, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = a[2 * hh + (e >> 2)][e & 3], y = b[2 * hh + (e >> 2)][e & 3];
      const __bf16 x0 = (__bf16)x, y0 = (__bf16)y;
      ah[e] = x0; al[e] = (__bf16)(x - (float)x0);
      bh[e] = y0; bl[e] = (__bf16)(y - (float)y0);
    }
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, C, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, C, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, C, 0, 0, 0);
  }
#else
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) C = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][c], b[q][c], C, 0, 0, 0);
#endif
}
// column sums of a DA tile: lane (i, hf) -> sum over its 16 points of feature i (other half via shfl)
__device__ __forceinline__ float col_sum(const float* DA, int i, int hf) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const f32x4 b = lds4(DA + pswz4(i, 4 * hf + q)); s += (b[0] + b[1]) + (b[2] + b[3]); }
  return s;
}

#ifndef NIF_PBW_DZR
#define NIF_PBW_DZR 8     // registers of the dL/dz prefetch: 64 x 8 floats = latent rows 0..15 of the [r <= 32][32] tile (rows 16.. are
                          // fetched where they are parked: 16 registers across the forward pass cost the generic-activation forms 40 spills)
#endif
#ifndef NIF_PBW_WAVES
#define NIF_PBW_WAVES 8   // 2 waves per SIMD (256 registers each, some spills) beat 1 wave with 478 registers: 0.27 -> 0.24 ms
#endif
// ACT: ACT_SWISH / ACT_SINE fixed at compile time (the defaults of the reference's ParameterNets), -1 = runtime switch
// SMALL (pi == 1 and r == 1, every benchmark ParameterNet): the first-layer and bottleneck gradients are rank-one
// per point, so they are accumulated per lane with 48 FMAs a tile (reduced over lanes once, at the end) instead of
// 32 f32-input MFMAs (2 k matrix-pipe cycles) + an LDS round trip of the dL/da tile
template <int NM, bool RES, int ACT, bool SMALL>
__global__ __launch_bounds__(64 * NIF_PBW_WAVES) void k_pnet_bwg(PbwArgs G) {
  const PNetArgs& A = G.p;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WV = NIF_PBW_WAVES;
  __shared__ float red[(WV - 1) * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = lane & 31, hf = lane >> 5, i = p;
  const long ntiles = (A.B + 31) / 32;
  constexpr int WLDS = (NM + 2) * 1024 + 256;      // floats of LDS per wave
  float* hs = lds + (long)wid * WLDS;              // NM+1 layer-input tiles, then the dL/da tile and the X tile
  float* gaT = hs + (NM + 1) * 1024;
  float* xT = gaT + 1024;                          // [8 rows: the pi <= 6 inputs, then ones, rest zero][32]
  const long plane = 256;                           // f32x4 per packed 32x32 matrix

  f32x16 C[NM], C1, Cb, CB1;   // SMALL: C1 = per-lane sum x da0, CB1 = per-lane sum da0, Cb = per-lane sum dz h
  float gbh[NM], gb0 = 0.f, gbb = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { C1[e] = 0.f; Cb[e] = 0.f; CB1[e] = 0.f; }
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    gbh[m] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) C[m][e] = 0.f;
  }
  for (int q = lane; q < 256; q += 64) xT[q] = 0.f;
  const PSmall S = psmall_stage<1>(A, lds + (long)WV * WLDS, threadIdx.x, 64 * WV);
  // the 2 NM packed 32x32 weight planes (forward, then adjoint) live in LDS for the whole kernel: the per-tile
  // dense products read their A operands with ds_read_b128 instead of waiting on global loads four times a tile
  f32x4* wpl = reinterpret_cast<f32x4*>(lds + (long)WV * WLDS + ((psmall_floats(A, 1) + 3) & ~3));
#if NIF_PBW_DENSE_BF16
  // bf16 split planes built here from theta (k_pnet_bf16.h): per matrix 6 KB forward + 4 KB adjoint
  pbf16x8* bpl = reinterpret_cast<pbf16x8*>(wpl);
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const long w_off = RES ? (m == 0 ? A.hid_w[0] : A.hid_w2[0]) : A.hid_w[m];
    pbf_build(bpl + m * (PBF_FWD_U4 + PBF_BWD_U4), bpl + m * (PBF_FWD_U4 + PBF_BWD_U4) + PBF_FWD_U4, A.theta, w_off, A.nst,
              threadIdx.x, 64 * WV);
  }
  auto dense_f = [&](int m, const f32x16 (&x)[1], f32x16 (&y)[1]) { pbf_dense_fwd(bpl + m * (PBF_FWD_U4 + PBF_BWD_U4), x[0], y[0], lane); };
  auto dense_b = [&](int m, const f32x16 (&x)[1], f32x16 (&y)[1]) { pbf_dense_bwd(bpl + m * (PBF_FWD_U4 + PBF_BWD_U4) + PBF_FWD_U4, x[0], y[0], lane); };
#else
  for (int e = threadIdx.x; e < NM * 256; e += 64 * WV) { wpl[e] = A.WF[e]; wpl[NM * 256 + e] = A.WB[e]; }
  const f32x4* WFl = wpl;
  const f32x4* WBl = wpl + NM * 256;
  auto dense_f = [&](int m, const f32x16 (&x)[1], f32x16 (&y)[1]) { dense_mfma_lds<1, 1, false>(WFl + (long)m * plane, x, y, lane); };
  auto dense_b = [&](int m, const f32x16 (&x)[1], f32x16 (&y)[1]) { dense_mfma_lds<1, 1, false>(WBl + (long)m * plane, x, y, lane); };
#endif
  __syncthreads();

  for (long tile = (long)blockIdx.x * WV + wid; tile < ntiles; tile += (long)gridDim.x * WV) {
    const long pt = tile * 32 + p;
    const long ptc = pt < A.B ? pt : A.B - 1;
    const float* prow = A.xin + ptc * A.ncol + A.col0;
    // !SMALL (r > 1 or pi > 1: the last-layer class, multi-parameter nets): the tile's dL/dz rows ([r][32], contiguous) are
    // fetched HERE, all loads in flight together behind the forward recomputation, and parked in the dL/da tile of the LDS (free
    // until the adjoint) -- r4: the r-trip loop below used to issue one dependent global load per latent row (cfg-4, r = 10:
    // 10 serial memory latencies per tile)
    float dzr[NIF_PBW_DZR];
    if (!SMALL) {
      const float* dzg = A.DZ + tile * A.r * 32;
      const int nd = A.r * 32;
#pragma unroll
      for (int q = 0; q < NIF_PBW_DZR; ++q) { const int e = lane + 64 * q; dzr[q] = e < nd ? dzg[e] : 0.f; }
    }
    // ---- forward (recomputed), layer inputs into the private LDS stash ---------------------------
    f32x16 h[1], T[1], d[NM + 1][1];
    {
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
      for (int dd = 0; dd < A.pi; ++dd) acc += prow[dd] * psmall_get(S.fw + dd * 32, 0, hf);
      h[0] = A.omega * acc + psmall_get(S.fb, 0, hf);
    }
    act_tile_sel<1, ACT>(A.act, h, h, d[0], A.nst, hf);
    // SMALL: the last layer-input tile is not needed in LDS (the bottleneck gradient is taken from registers), so
    // the first layer's act'(a) -- live from here to the very end of the tile otherwise -- is parked there
    float* park = hs + NM * 1024;
    if (SMALL) {
#pragma unroll
      for (int v = 0; v < 16; ++v) park[v * 64 + lane] = d[0][0][v];
    }
    if (!SMALL && hf == 0) {
      for (int dd = 0; dd < A.pi; ++dd) xT[pswz(dd, p)] = prow[dd];
      xT[pswz(A.pi, p)] = 1.0f;
    }
    if (!RES) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        tile_store(hs + m * 1024, h[0], p, hf);
        dense_f(m, h, T);
        T[0] = A.omega * T[0] + psmall_get(S.hb + m * 32, 0, hf);
        act_tile_sel<1, ACT>(A.act, T, T, d[m + 1], A.nst, hf);
        h[0] = A.siren ? T[0] : h[0] + T[0];
      }
    } else {
      f32x16 t[1];
      tile_store(hs, h[0], p, hf);
      dense_f(0, h, T);
      T[0] = A.omega * T[0] + psmall_get(S.hb, 0, hf);
      act_tile_sel<1, ACT>(A.act, T, t, d[1], A.nst, hf);
      tile_store(hs + 1024, t[0], p, hf);
      dense_f(1, t, T);
      {
        const f32x16 lin = A.omega * T[0] + psmall_get(S.hb2, 0, hf);
        T[0] = A.siren ? lin : h[0] + lin;
      }
      act_tile_sel<1, ACT>(A.act, T, T, d[NM], A.nst, hf);
      h[0] = A.siren ? 0.5f * (h[0] + T[0]) : T[0];
    }
    if (!SMALL) tile_store(hs + NM * 1024, h[0], p, hf);
    // ---- bottleneck: dL/dW_b[f][c] = sum_p h[p][f] dz_c[p]; gh[f] = sum_c dz_c W_b[f][c] -------------
    f32x16 gh[1], ga[1], U[1];
#pragma unroll
    for (int v = 0; v < 16; ++v) gh[0][v] = 0.f;
    if (!SMALL) {
#pragma unroll
      for (int q = 0; q < NIF_PBW_DZR; ++q) gaT[pswz(2 * q + hf, p)] = dzr[q];       // element lane + 64 q = (row 2 q + hf, point p); rows >= r are zero
#pragma unroll
      for (int q = NIF_PBW_DZR; q < 16; ++q) {
        const int e = lane + 64 * q;
        gaT[pswz(2 * q + hf, p)] = e < A.r * 32 ? A.DZ[tile * A.r * 32 + e] : 0.f;
      }
      for (int c = 0; c < A.r; ++c) gh[0] += gaT[pswz(c, p)] * psmall_get(S.bw + c * 32, 0, hf);
    } else {
      gh[0] += A.DZ[tile * 32 + p] * psmall_get(S.bw, 0, hf);
    }
    if (SMALL) {
      const float dz = A.DZ[tile * 32 + p];
      Cb += dz * h[0];
      gbb += hf == 0 ? dz : 0.f;
    } else {
      // dL/dW_b = h^T dz over the tile's points: the parked dL/dz tile is the DA operand as it lies (zero rows beyond r).  r4: the
      // bf16 hi / lo form of the hidden matrices (6 x 32 matrix-pipe cycles) instead of 16 f32-input MFMAs (16 x 64)
      grad_mfma(hs + NM * 1024, gaT, Cb, i, hf);
      gbb += col_sum(gaT, i, hf);
    }
    // touch loads (see PbwArgs): issued once this tile's own global loads are consumed, waited for at the end of
    // the tile.  This needs a spill-free kernel: scratch reloads count in vmcnt and the queue completes in order, so
    // with the 5-8 spilled registers this kernel used to have, the first reload after this point waited for the touch
    // loads too (+22 us instead of +12)
    float tv[4] = {0.f, 0.f, 0.f, 0.f};
    if (G.touch) {
      const float* tb = G.touch + tile * G.touch_floats + lane * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) tv[q] = tb[q * 1024 < G.touch_floats ? q * 1024 : 0];
    }
    // ---- adjoint through the hidden matrices, gradient GEMMs on the way -----------------------------
    if (!RES) {
#pragma unroll
      for (int m = NM - 1; m >= 0; --m) {
        ga[0] = gh[0] * d[m + 1][0];
        tile_store(gaT, ga[0], p, hf);
        grad_mfma(hs + m * 1024, gaT, C[m], i, hf);
        gbh[m] += col_sum(gaT, i, hf);
        dense_b(m, ga, U);
        gh[0] = A.siren ? A.omega * U[0] : gh[0] + A.omega * U[0];
      }
    } else {
      const float half = A.siren ? 0.5f : 1.0f;
      ga[0] = half * gh[0] * d[NM][0];
      tile_store(gaT, ga[0], p, hf);
      grad_mfma(hs + 1024, gaT, C[NM - 1], i, hf);
      gbh[NM - 1] += col_sum(gaT, i, hf);
      dense_b(1, ga, U);
      f32x16 skip;
      skip = A.siren ? 0.5f * gh[0] : ga[0];
      ga[0] = A.omega * U[0] * d[1][0];
      tile_store(gaT, ga[0], p, hf);
      grad_mfma(hs, gaT, C[0], i, hf);
      gbh[0] += col_sum(gaT, i, hf);
      dense_b(0, ga, U);
      gh[0] = skip + A.omega * U[0];
    }
    // ---- first layer: dL/dW_1[d][f] = w0 sum_p x_d[p] da0[p][f] (rows of X^T), bias = column sums -----
    if (SMALL) {
#pragma unroll
      for (int v = 0; v < 16; ++v) ga[0][v] = gh[0][v] * park[v * 64 + lane];
    } else {
      ga[0] = gh[0] * d[0][0];
    }
    if (SMALL) {
      C1 += prow[0] * ga[0];
      CB1 += ga[0];
    } else {
      tile_store(gaT, ga[0], p, hf);
      grad_mfma(xT, gaT, C1, i, hf, 8);     // rows of X^T: the pi <= 6 inputs, then ones (the bias row)
    }
    asm volatile("" ::"v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3]));
  }

  // ---- workgroup reduction (fixed order) and this workgroup's partial row --------------------------------
  float* prow_out = G.partial + (long)blockIdx.x * G.pstride;
  auto sum8 = [&](float v) -> float {
    if (wid > 0) red[(wid - 1) * 64 + lane] = v;
    __syncthreads();
    if (wid == 0) {
#pragma unroll
      for (int w = 0; w < WV - 1; ++w) v += red[w * 64 + lane];
    }
    __syncthreads();
    return v;
  };
  // whole accumulator blocks per round; the per-wave layer tiles in LDS are dead by now and serve as scratch
  __syncthreads();
  auto sum16 = [&](f32x16 v) -> f32x16 {
    if (wid > 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) lds[((wid - 1) * 16 + e) * 64 + lane] = v[e];
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
      for (int w = 0; w < WV - 1; ++w)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += lds[(w * 16 + e) * 64 + lane];
    }
    __syncthreads();
    return v;
  };
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const long w_off = RES ? (m == 0 ? A.hid_w[0] : A.hid_w2[0]) : A.hid_w[m];
    const long b_off = RES ? (m == 0 ? A.hid_b[0] : A.hid_b2[0]) : A.hid_b[m];
    {
      const f32x16 vs = sum16(C[m]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int in = fmap(e, hf), out = i;
        if (wid == 0 && in < A.nst && out < A.nst) prow_out[w_off + (long)in * A.nst + out] = A.omega * vs[e];
      }
    }
    float vb = gbh[m];
    vb += __shfl_xor(vb, 32);
    vb = sum8(vb);
    if (wid == 0 && hf == 0 && i < A.nst) prow_out[b_off + i] = vb;
  }
  if (SMALL) {
    // sum over the 32 point lanes of each half (the xor butterfly stays inside the half), then over the waves
    auto lanes32 = [&](f32x16 v) -> f32x16 {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += __shfl_xor(v[e], off);
      return v;
    };
    const f32x16 s1 = sum16(lanes32(C1)), s0 = sum16(lanes32(CB1)), sb = sum16(lanes32(Cb));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = fmap(e, hf);
      if (wid == 0 && p == 0 && f < A.nst) {
        prow_out[A.first_w + f] = A.omega * s1[e];
        prow_out[A.first_b + f] = s0[e];
        prow_out[A.bott_w + f] = sb[e];
      }
    }
    float v = gbb;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    v = sum8(v);
    if (wid == 0 && lane == 0) prow_out[A.bott_b] = v;
    return;
  }
  const f32x16 s1 = sum16(C1), sb = sum16(Cb);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float v1 = s1[e];
    const float vb = sb[e];
    const int row = fmap(e, hf);
    if (wid == 0) {
      if (row < A.pi && i < A.nst) prow_out[A.first_w + (long)row * A.nst + i] = A.omega * v1;
      if (row == A.pi && i < A.nst) prow_out[A.first_b + i] = v1;
      if (row < A.nst && i < A.r) prow_out[A.bott_w + (long)row * A.r + i] = vb;
    }
  }
  (void)gb0;
  {
    float v = gbb;
    v += __shfl_xor(v, 32);
    v = sum8(v);
    if (wid == 0 && hf == 0 && i < A.r) prow_out[A.bott_b + i] = v;
  }
}

bool pnet_bwg_supported(const PNetArgs& a) {
  const int nm = a.lst * (a.res ? 2 : 1);
  // (last-layer class: DZ is then dL/dlatent in front of the rl x rl map, whose own gradient stays a k_gw_out launch)
  return a.nst <= 32 && a.pi <= 6 && a.r <= 32 && nm >= 1 && nm <= 2 && (!a.res || a.lst == 1);
}

void launch_pnet_bwg(const PNetArgs& a, float* partial, long pstride, int rows, hipStream_t st, const float* touch,
                     long touch_floats) {
  PbwArgs G; G.p = a; G.partial = partial; G.pstride = pstride; G.touch = touch; G.touch_floats = touch_floats;
  const int nm = a.lst * (a.res ? 2 : 1);
  dim3 grid(rows), block(64 * NIF_PBW_WAVES);
  const size_t shm = ((size_t)NIF_PBW_WAVES * ((nm + 2) * 1024 + 256) + (size_t)((psmall_floats(a, 1) + 3) & ~3) +
                      (size_t)nm * (NIF_PBW_DENSE_BF16 ? 2560 : 2048)) * sizeof(float);
#define PBW1(NM_, RES_, ACT_, SM_)                                                                                  \
  {                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)k_pnet_bwg<NM_, RES_, ACT_, SM_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((k_pnet_bwg<NM_, RES_, ACT_, SM_>), grid, block, shm, st, G);                                \
  }
#define PBW(NM_, RES_, ACT_)                                                                                        \
  {                                                                                                                 \
    if (a.pi == 1 && a.r == 1) PBW1(NM_, RES_, ACT_, true) else PBW1(NM_, RES_, ACT_, false)                        \
  }
#define PBWA(NM_, RES_)                                                       \
  if (a.act == ACT_SWISH) PBW(NM_, RES_, ACT_SWISH)                           \
  else if (a.act == ACT_SINE) PBW(NM_, RES_, ACT_SINE)                        \
  else PBW(NM_, RES_, -1)
  if (a.res) { PBWA(2, true) }
  else if (nm == 1) { PBWA(1, false) }
  else { PBWA(2, false) }
#undef PBWA
#undef PBW
#undef PBW1
}
