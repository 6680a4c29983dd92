// k_mlpjac.hip -- forward-mode tangent of the shared-weight MLPs (ParameterNet, and the dense SIREN
// ShapeNet of the last-layer class) w.r.t. ONE of their input columns, next to the primal.  Used by
// nif_jacobian for parameter columns of the hypernetwork classes (dz/dp feeds k_jac) and for both kinds
// of columns of the last-layer class.  Same tiles / MFMA scheme as k_pnet (nif_internal.h).
//
//   first     a = w0 p.W + b                 a' = w0 W[seed][:]              h' = act'(a) a'
//   SIREN     h+ = sin(w0 hW + b)            h+' = cos(.) w0 (h'W)
//   shortcut  h+ = h + act(hK + b)           h+' = h' + act'(.) (h'K)            (mlp.py:148-160)
//   MLP res   h+ = act(h + L2(act(L1 h)))    t' = act'(a1)(h'K1); h+' = act'(a2)(h' + t'K2)   (mlp.py:62-79)
//   SIREN res h+ = .5(h + sin(w0 tW2 + b2))  t' = cos(a1) w0 (h'W); h+' = .5(h' + cos(a2) w0 (t'W2))  (siren.py:381-410)
//   bottleneck z = hWb + bb                  z' = h'Wb ;  last-layer class: a' = z' Wl
#include "nif_internal.h"

struct MlpJacArgs {
  PNetArgs p;      // primal arguments (Z = primal output [tiles][r][32]; ZL as in k_pnet)
  int seed;        // input column (0..pi-1) the tangent is taken with respect to
  float* ZD;       // tangent of the output  [tiles][r][32]
};

template <int NB>
__global__ __launch_bounds__(256) void k_mlp_jac(MlpJacArgs J) {
  const PNetArgs& A = J.p;
  extern __shared__ float zl_lds[];  // LL kind: [4 waves][2][r][32]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = lane & 31, hf = lane >> 5;
  const long ntiles = (A.B + 31) / 32;
  const long tile = (long)blockIdx.x * 4 + wid;
  if (tile >= ntiles) return;
  const long pt = tile * 32 + p;
  const long ptc = pt < A.B ? pt : A.B - 1;
  const long plane = (long)NB * NB * 256;
  const float* prow = A.xin + ptc * A.ncol + A.col0;

  f32x16 h[NB], hd[NB], d[NB], T[NB];
  // first layer
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int f = 32 * b + fmap(v, hf);
      float a = 0.f, ad = 0.f;
      if (f < A.nst) {
        float acc = 0.f;
        for (int dd = 0; dd < A.pi; ++dd) acc = fmaf(prow[dd], A.theta[A.first_w + (long)dd * A.nst + f], acc);
        a = A.omega * acc + A.theta[A.first_b + f];
        ad = A.omega * A.theta[A.first_w + (long)J.seed * A.nst + f];
      }
      h[b][v] = a; hd[b][v] = ad;
    }
  act_tile<NB>(A.act, h, h, d, A.nst, hf);
#pragma unroll
  for (int b = 0; b < NB; ++b) hd[b] *= d[b];

  for (int i = 0; i < A.lst; ++i) {
    if (!A.res) {
      f32x16 Td[NB];
      dense_mfma<NB, NB>(A.WF + (long)i * plane, h, T, lane);
      dense_mfma<NB, NB>(A.WF + (long)i * plane, hd, Td, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int f = 32 * b + fmap(v, hf);
          T[b][v] = f < A.nst ? A.omega * T[b][v] + A.theta[A.hid_b[i] + f] : 0.f;
        }
      act_tile<NB>(A.act, T, T, d, A.nst, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x16 t = d[b] * (A.omega * Td[b]);
        hd[b] = A.siren ? t : hd[b] + t;
        h[b] = A.siren ? T[b] : h[b] + T[b];
      }
    } else {
      f32x16 t[NB], td[NB], Td[NB];
      dense_mfma<NB, NB>(A.WF + (long)(2 * i) * plane, h, T, lane);
      dense_mfma<NB, NB>(A.WF + (long)(2 * i) * plane, hd, Td, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int f = 32 * b + fmap(v, hf);
          T[b][v] = f < A.nst ? A.omega * T[b][v] + A.theta[A.hid_b[i] + f] : 0.f;
        }
      act_tile<NB>(A.act, T, t, d, A.nst, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) td[b] = d[b] * (A.omega * Td[b]);
      dense_mfma<NB, NB>(A.WF + (long)(2 * i + 1) * plane, t, T, lane);
      dense_mfma<NB, NB>(A.WF + (long)(2 * i + 1) * plane, td, Td, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int f = 32 * b + fmap(v, hf);
          const float lin = f < A.nst ? A.omega * T[b][v] + A.theta[A.hid_b2[i] + f] : 0.f;
          T[b][v] = A.siren ? lin : h[b][v] + lin;
        }
      act_tile<NB>(A.act, T, T, d, A.nst, hf);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (A.siren) {
          hd[b] = 0.5f * (hd[b] + d[b] * (A.omega * Td[b]));
          h[b] = 0.5f * (h[b] + T[b]);
        } else {
          hd[b] = d[b] * (hd[b] + Td[b]);
          h[b] = T[b];
        }
      }
    }
  }
  // bottleneck and (last-layer class) the r x r map, primal and tangent
  float* zl = zl_lds + (long)wid * 2 * A.r * 32;
  for (int c = 0; c < A.r; ++c) {
    float s = 0.f, sd = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int f = 32 * b + fmap(v, hf);
        if (f < A.nst) {
          const float w = A.theta[A.bott_w + (long)f * A.r + c];
          s = fmaf(h[b][v], w, s);
          sd = fmaf(hd[b][v], w, sd);
        }
      }
    s += __shfl_xor(s, 32);
    sd += __shfl_xor(sd, 32);
    s += A.theta[A.bott_b + c];
    if (A.ll_kind) {
      if (hf == 0) { zl[c * 32 + p] = s; zl[(A.r + c) * 32 + p] = sd; }
    } else if (hf == 0) {
      A.Z[(tile * A.r + c) * 32 + p] = s;
      J.ZD[(tile * A.r + c) * 32 + p] = sd;
    }
  }
  if (A.ll_kind) {
    for (int c = hf; c < A.r; c += 2) {
      float s = A.theta[A.last_b + c], sd = 0.f;
      for (int kk = 0; kk < A.r; ++kk) {
        const float w = A.theta[A.last_w + (long)kk * A.r + c];
        s = fmaf(zl[kk * 32 + p], w, s);
        sd = fmaf(zl[(A.r + kk) * 32 + p], w, sd);
      }
      A.Z[(tile * A.r + c) * 32 + p] = s;
      J.ZD[(tile * A.r + c) * 32 + p] = sd;
    }
  }
}

void launch_mlp_jac(const PNetArgs& a, int NB, int seed, float* ZD, hipStream_t st) {
  MlpJacArgs J;
  J.p = a; J.seed = seed; J.ZD = ZD;
  const long ntiles = (a.B + 31) / 32;
  dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
  const size_t shm = a.ll_kind ? (size_t)4 * 2 * a.r * 32 * sizeof(float) : 0;
  if (NB == 1) hipLaunchKernelGGL((k_mlp_jac<1>), grid, block, shm, st, J);
  else if (NB == 2) hipLaunchKernelGGL((k_mlp_jac<2>), grid, block, shm, st, J);
  else hipLaunchKernelGGL((k_mlp_jac<4>), grid, block, shm, st, J);
}

// last-layer class: dy/dcol of u = Dot(phi, a) + bias given (phi, a) and ONE tangent pair (phi', a')
// (either may be absent: coordinate columns move only phi, parameter columns only a)
__global__ void k_ll_jac_out(const float* __restrict__ PHI, const float* __restrict__ Z, const float* __restrict__ PHID,
                             const float* __restrict__ ZD, long B, int r, int so, int nx_total, int xcol,
                             float* __restrict__ dydx) {
  const long pt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= B) return;
  const long tile = pt >> 5;
  const int p = (int)(pt & 31);
  for (int s = 0; s < so; ++s) {
    float u = 0.f;
    for (int j = 0; j < r; ++j) {
      if (PHID) u = fmaf(PHID[(tile * (long)(so * r) + s * r + j) * 32 + p], Z[(tile * r + j) * 32 + p], u);
      if (ZD) u = fmaf(PHI[(tile * (long)(so * r) + s * r + j) * 32 + p], ZD[(tile * r + j) * 32 + p], u);
    }
    dydx[(pt * so + s) * nx_total + xcol] = u;
  }
}
void launch_ll_jac_out(const float* PHI, const float* Z, const float* PHID, const float* ZD, long B, int r, int so,
                       int nx_total, int xcol, float* dydx, hipStream_t st) {
  dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_ll_jac_out, grid, block, 0, st, PHI, Z, PHID, ZD, B, r, so, nx_total, xcol, dydx);
}
