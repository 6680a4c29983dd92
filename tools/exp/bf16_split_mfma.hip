// probe: fp32 GEMM tile D[16x16] = A[16x64] * B[64x16] on v_mfma_f32_16x16x32_bf16 with error-compensated bf16 splits
// (3 products: hi*hi + hi*lo + lo*hi ; 6 products: + hi*lo2 + lo2*hi + lo*lo) against fp64 and native fp32 MFMA.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)a; const float r1 = a - (float)h;
  m = (__bf16)r1; const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}
// A row-major [16][64], B [64][16] row-major (k, j).  One wave.
__global__ void k(const float* A, const float* B, float* D32, float* D3, float* D6) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // native fp32: 16 MFMAs 16x16x4: lane holds A[i][4kk+g], B[4kk+g][j]
  f32x4 c = {0, 0, 0, 0};
  for (int kk = 0; kk < 16; ++kk) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 64 + 4 * kk + g], B[(4 * kk + g) * 16 + i], c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D32[(4 * g + v) * 16 + i] = c[v];
  // bf16 16x16x32: lane holds K = 8g..8g+7 of row i (A) / col i (B); two K-steps for K=64
  f32x4 c3 = {0, 0, 0, 0}, c6 = {0, 0, 0, 0};
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int t = 0; t < 8; ++t) {
      __bf16 h, m, lo;
      split3(A[i * 64 + 32 * ks + 8 * g + t], h, m, lo); ah[t] = h; am[t] = m; al[t] = lo;
      split3(B[(32 * ks + 8 * g + t) * 16 + i], h, m, lo); bh[t] = h; bm[t] = m; bl[t] = lo;
    }
    // small terms first
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c6, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c3, 0, 0, 0);
  }
  for (int v = 0; v < 4; ++v) { D3[(4 * g + v) * 16 + i] = c3[v]; D6[(4 * g + v) * 16 + i] = c6[v]; }
}
int main() {
  const int NT = 2000;
  double e32 = 0, e3 = 0, e6 = 0, nrm = 0, m32 = 0, m3 = 0, m6 = 0;
  std::vector<float> A(16 * 64), B(64 * 16), d32(256), d3(256), d6(256);
  float *dA, *dB, *o32, *o3, *o6;
  (void)hipMalloc(&dA, 4096); (void)hipMalloc(&dB, 4096); (void)hipMalloc(&o32, 1024); (void)hipMalloc(&o3, 1024); (void)hipMalloc(&o6, 1024);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
  for (int t = 0; t < NT; ++t) {
    for (auto& x : A) x = (float)(rnd() * exp(3.0 * rnd()));
    for (auto& x : B) x = (float)(rnd() * exp(3.0 * rnd()));
    (void)hipMemcpy(dA, A.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, o32, o3, o6);
    (void)hipMemcpy(d32.data(), o32, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(d3.data(), o3, 1024, hipMemcpyDeviceToHost);
    (void)hipMemcpy(d6.data(), o6, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double r = 0, ab = 0;
        for (int kx = 0; kx < 64; ++kx) { r += (double)A[i * 64 + kx] * B[kx * 16 + j]; ab += fabs((double)A[i * 64 + kx] * B[kx * 16 + j]); }
        const double a32 = fabs(d32[i * 16 + j] - r) / ab, a3 = fabs(d3[i * 16 + j] - r) / ab, a6 = fabs(d6[i * 16 + j] - r) / ab;
        e32 += a32 * a32; e3 += a3 * a3; e6 += a6 * a6; nrm += 1;
        if (a32 > m32) m32 = a32; if (a3 > m3) m3 = a3; if (a6 > m6) m6 = a6;
      }
  }
  printf("error / sum|a_k b_k|  (K=64):  rms  fp32-mfma %.3e  bf16x3 %.3e  bf16x6 %.3e\n", sqrt(e32 / nrm), sqrt(e3 / nrm), sqrt(e6 / nrm));
  printf("                               max  fp32-mfma %.3e  bf16x3 %.3e  bf16x6 %.3e\n", m32, m3, m6);
  return 0;
}
