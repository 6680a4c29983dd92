// tr_probe.hip -- the LDS exchange of the fused weight-gradient path (nif_amd/csrc/k_fuse_dev.h) on its own: one wave deposits a
// 64-feature x 16-point tile pair (h, d) held with points on lanes, then forms G[i][j] = sum_p h[i][p] d[j][p] (3 bf16 products of
// the hi/lo splits), the column sums sum_p d[j][p] and a weighted sum sum_p w[p] d[j][p] the way k_snet6 does; checked against fp64.
//   hipcc --offload-arch=gfx950 -O3 -I nif_amd/csrc tools/exp/tr_probe.hip -o tools/exp/tr_probe && tools/exp/tr_probe
#include "k_fuse_dev.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ __launch_bounds__(64) void k_probe(const float* H, const float* D, const float* W, float* G, float* S, float* X) {
  __shared__ __attribute__((aligned(256))) char ex[4 * FUSE_PLANE_BYTES + 64];
  const int lane = threadIdx.x, p = lane & 15, g = lane >> 4;
  f32x4 h[4], d[4];
  for (int b = 0; b < 4; ++b)
    for (int v = 0; v < 4; ++v) { h[b][v] = H[(16 * b + 4 * g + v) * 16 + p]; d[b][v] = D[(16 * b + 4 * g + v) * 16 + p]; }
  bf16x8 h0[2], h1[2], d0[2], d1[2];
  split2<4>(h, h0, h1);
  split2<4>(d, d0, d1);
  const FuseDep dep = fuse_dep_addr(p, g);
  fuse_deposit4(ex + 0 * FUSE_PLANE_BYTES, dep, h0);
  fuse_deposit4(ex + 1 * FUSE_PLANE_BYTES, dep, h1);
  fuse_deposit4(ex + 2 * FUSE_PLANE_BYTES, dep, d0);
  fuse_deposit4(ex + 3 * FUSE_PLANE_BYTES, dep, d1);
  // weights per point as bf16 (hi | lo) rows of 16
  __bf16* wv = reinterpret_cast<__bf16*>(ex + 4 * FUSE_PLANE_BYTES);
  if (lane < 16) {
    const float w = W[lane];
    const __bf16 w0 = (__bf16)w;
    wv[lane] = w0; wv[16 + lane] = (__bf16)(w - (float)w0);
  }
  __syncthreads();
  const FuseRd rd = fuse_rd_addr(lane);
  const int i = lane & 31, hf = lane >> 5;
  const bf16x8 whi = *reinterpret_cast<const bf16x8*>(wv + 8 * hf), wlo = *reinterpret_cast<const bf16x8*>(wv + 16 + 8 * hf);
  for (int I = 0; I < 2; ++I)
    for (int J = 0; J < 2; ++J) {
      const bf16x8 ah = fuse_read_op(ex + 0 * FUSE_PLANE_BYTES, rd, I), al = fuse_read_op(ex + 1 * FUSE_PLANE_BYTES, rd, I);
      const bf16x8 bh = fuse_read_op(ex + 2 * FUSE_PLANE_BYTES, rd, J), bl = fuse_read_op(ex + 3 * FUSE_PLANE_BYTES, rd, J);
      f32x16 acc;
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      for (int e = 0; e < 16; ++e) G[(32 * I + fmap(e, hf)) * 64 + 32 * J + i] = acc[e];
      if (I == 0) {
        float s = fuse_sum8(bh, bl, 0.f);
        s += __shfl_xor(s, 32);
        float x = fuse_dot8(bh, bl, whi, wlo, 0.f);
        x += __shfl_xor(x, 32);
        if (hf == 0) { S[32 * J + i] = s; X[32 * J + i] = x; }
      }
    }
}

int main() {
  const int N = 64 * 16;
  std::vector<float> H(N), D(N), W(16), G(64 * 64), S(64), X(64);
  srand(1);
  for (int i = 0; i < N; ++i) { H[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f; D[i] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f; }
  for (int i = 0; i < 16; ++i) W[i] = rand() / (float)RAND_MAX * 3.f - 1.f;
  float *dH, *dD, *dW, *dG, *dS, *dX;
  hipMalloc(&dH, N * 4); hipMalloc(&dD, N * 4); hipMalloc(&dW, 64); hipMalloc(&dG, 64 * 64 * 4); hipMalloc(&dS, 256); hipMalloc(&dX, 256);
  hipMemcpy(dH, H.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dD, D.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dH, dD, dW, dG, dS, dX);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  hipMemcpy(G.data(), dG, 64 * 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(S.data(), dS, 256, hipMemcpyDeviceToHost);
  hipMemcpy(X.data(), dX, 256, hipMemcpyDeviceToHost);
  double eg = 0, ng = 0, es = 0, ns = 0, ex = 0, nx = 0;
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < 64; ++j) {
      double r = 0;
      for (int p = 0; p < 16; ++p) r += (double)H[i * 16 + p] * D[j * 16 + p];
      eg += (G[i * 64 + j] - r) * (G[i * 64 + j] - r); ng += r * r;
    }
  for (int j = 0; j < 64; ++j) {
    double r = 0, q = 0;
    for (int p = 0; p < 16; ++p) { r += D[j * 16 + p]; q += (double)W[p] * D[j * 16 + p]; }
    es += (S[j] - r) * (S[j] - r); ns += r * r;
    ex += (X[j] - q) * (X[j] - q); nx += q * q;
  }
  const double rg = sqrt(eg / ng), rs = sqrt(es / ns), rx = sqrt(ex / nx);
  printf("tr_probe: G rel-L2 %.3e   column sums %.3e   weighted sums %.3e\n", rg, rs, rx);
  const bool ok = rg < 2e-5 && rs < 2e-5 && rx < 2e-5;
  printf(ok ? "tr_probe ok\n" : "tr_probe FAILED\n");
  return ok ? 0 : 1;
}
