// Probe: how does v_mfma_f32_32x32x16_bf16 round?  (tools/probes; not part of the library.)
//   T1  C = 1, one product of 0.75 ulp(1):            RNE -> 1 + 2^-23, truncation -> 1
//   T2  C = 1, sixteen products of 1/8 ulp(1) each:   exact sum 2 ulp -> 1 + 2^-22; per-product truncation -> 1
//   T3  C = 1, products +1.25 ulp and -0.5 ulp ...:   sign handling
//   T4  chains of n accumulating MFMAs on random bf16 data vs fp64: error growth (biased ~ n, unbiased ~ sqrt n)
// build: hipcc --offload-arch=gfx950 -O2 mfma_bf16_numerics.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ bf16x8 ld8(const u16* p) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
// A (32 x 16n) and B (32 x 16n) row-major bf16 bits; D[i][j] = C0 + sum_k A[i][k] B[j][k], n chained MFMAs
__global__ void chain(const u16* A, const u16* B, float* D, int n, float c0) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = c0;
  for (int s = 0; s < n; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld8(A + (size_t)r * 16 * n + 16 * s + 8 * h), ld8(B + (size_t)r * 16 * n + 16 * s + 8 * h), acc, 0, 0, 0);
  for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}
// the fp32 MFMA for comparison: D[i][j] = c0 + sum_k A[i][k] B[j][k] as an fmaf chain
__global__ void chain32(const float* A, const float* B, float* D, int K, float c0) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = c0;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)r * K + k + h], B[(size_t)r * K + k + h], acc, 0, 0, 0);
  for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}
static u16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); unsigned r = u + 0x7fff + ((u >> 16) & 1); return (u16)(r >> 16); }
static float bf2f(u16 b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  u16 *dA, *dB; float *dD, *dA32, *dB32;
  const int NMAX = 4096;
  hipMalloc(&dA, 32 * 16 * NMAX * 2); hipMalloc(&dB, 32 * 16 * NMAX * 2); hipMalloc(&dD, 32 * 32 * 4);
  hipMalloc(&dA32, 32 * 16 * NMAX * 4); hipMalloc(&dB32, 32 * 16 * NMAX * 4);
  std::vector<u16> A(32 * 16 * NMAX), B(32 * 16 * NMAX);
  std::vector<float> D(1024);
  auto run = [&](int n, float c0) {
    hipMemcpy(dA, A.data(), 32 * 16 * n * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 32 * 16 * n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, n, c0);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  };
  auto clear = [&](int n) { std::fill(A.begin(), A.begin() + 32 * 16 * n, 0); std::fill(B.begin(), B.begin() + 32 * 16 * n, 0); };
  const float ulp = ldexpf(1.f, -23);
  // T1
  clear(1); A[0] = f2bf(ldexpf(1.f, -12)); B[0] = f2bf(ldexpf(1.5f, -12));
  run(1, 1.f); printf("T1 one product of 0.75 ulp:           D-1 = %.3f ulp  (RNE 1, truncation 0)\n", (D[0] - 1.f) / ulp);
  // T2
  clear(1); for (int k = 0; k < 16; ++k) { A[k] = f2bf(ldexpf(1.f, -13)); B[k] = f2bf(ldexpf(1.f, -13)); }
  run(1, 1.f); printf("T2 sixteen products of 1/8 ulp:      D-1 = %.3f ulp  (exact sum 2)\n", (D[0] - 1.f) / ulp);
  clear(1); for (int k = 0; k < 16; ++k) { A[k] = f2bf(ldexpf(1.f, -13)); B[k] = f2bf(ldexpf(1.f, -14)); }
  run(1, 1.f); printf("T2b sixteen products of 1/16 ulp:    D-1 = %.3f ulp  (exact sum 1)\n", (D[0] - 1.f) / ulp);
  clear(1); for (int k = 0; k < 3; ++k) { A[k] = f2bf(ldexpf(1.f, -13)); B[k] = f2bf(ldexpf(1.f, -13)); }
  run(1, 1.f); printf("T2c three products of 1/8 ulp:       D-1 = %.3f ulp  (exact sum 0.375 -> RNE 0)\n", (D[0] - 1.f) / ulp);
  clear(1); for (int k = 0; k < 5; ++k) { A[k] = f2bf(ldexpf(1.f, -13)); B[k] = f2bf(ldexpf(1.f, -13)); }
  run(1, 1.f); printf("T2d five products of 1/8 ulp:        D-1 = %.3f ulp  (exact sum 0.625 -> RNE 1)\n", (D[0] - 1.f) / ulp);
  // T3
  clear(1); A[0] = f2bf(ldexpf(1.f, -12)); B[0] = f2bf(ldexpf(1.25f, -11)); A[1] = f2bf(-ldexpf(1.f, -12)); B[1] = f2bf(ldexpf(1.f, -12));
  run(1, 1.f); printf("T3 +1.25 ulp - 0.5 ulp:              D-1 = %.3f ulp  (exact 0.75 -> RNE 1)\n", (D[0] - 1.f) / ulp);
  clear(1); A[0] = f2bf(ldexpf(1.f, -12)); B[0] = f2bf(-ldexpf(1.5f, -12));
  run(1, 1.f); printf("T3b one product of -0.75 ulp(1):     D-1 = %.3f ulp(1) (RNE -1 [= -2 half-ulps below 1], toward zero 0)\n", (D[0] - 1.f) / ulp);
  clear(1); A[0] = f2bf(ldexpf(1.f, -12)); B[0] = f2bf(ldexpf(1.5f, -12));
  run(1, -1.f); printf("T3c C = -1, product +0.75 ulp:       D+1 = %.3f ulp  (RNE +1 ... exact -1 + 0.75 ulp)\n", (D[0] + 1.f) / ulp);
  // T4: error growth of accumulation chains, random data; compare the bf16 MFMA chain, the fp32 MFMA chain (same bf16-valued
  // data, so products are exact in both) and fp64
  srand(1);
  for (int n : {4, 16, 64, 256, 1024, 4096}) {
    const int K = 16 * n;
    std::vector<float> A32(32 * K), B32(32 * K);
    for (int i = 0; i < 32 * K; ++i) {
      float a = (float)rand() / RAND_MAX * 2 - 1, b = (float)rand() / RAND_MAX * 2 - 1;
      A[i] = f2bf(a); B[i] = f2bf(b); A32[i] = bf2f(A[i]); B32[i] = bf2f(B[i]);
    }
    run(n, 0.f);
    std::vector<float> D16 = D;
    hipMemcpy(dA32, A32.data(), 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB32, B32.data(), 32 * K * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain32, dim3(1), dim3(64), 0, 0, dA32, dB32, dD, K, 0.f);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double e16 = 0, e32 = 0, b16 = 0, b32 = 0, sc = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0, sabs = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A32[i * K + k] * B32[j * K + k]; sabs += fabs((double)A32[i * K + k] * B32[j * K + k]); }
      const double d16 = D16[i * 32 + j] - ref, d32 = D[i * 32 + j] - ref;
      e16 += d16 * d16; e32 += d32 * d32; sc += sabs;
      b16 += d16 * (ref >= 0 ? 1 : -1); b32 += d32 * (ref >= 0 ? 1 : -1);          // bias toward / away from zero
    }
    sc /= 1024;
    printf("T4 K = %6d: rms err / mean sum|ab|:  bf16 MFMA chain %.3e   fp32 MFMA chain %.3e   ratio %.2f;  signed bias (toward zero < 0): bf16 %.2e fp32 %.2e\n",
           K, sqrt(e16 / 1024) / sc, sqrt(e32 / 1024) / sc, sqrt(e16 / e32), b16 / 1024 / sc, b32 / 1024 / sc);
  }
  return 0;
}
