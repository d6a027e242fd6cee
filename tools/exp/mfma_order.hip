// Experiment: does v_mfma_f32_32x32x2f32 accumulate like a sequential fp32 FMA chain in k order?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k32(const float *A, const float *B, float *D, int K) {  // A [32][K], B [K][32], D [32][32]
  const int l = threadIdx.x;
  float16v c = {0};
  for (int k = 0; k < K; k += 2) {
    const float a = A[(l % 32) * K + k + l / 32];
    const float b = B[(k + l / 32) * 32 + l % 32];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) D[(8 * (i / 4) + (l / 32) * 4 + i % 4) * 32 + l % 32] = c[i];
}
__global__ void k16(const float *A, const float *B, float *D, int K) {  // 16x16x4: A [16][K], B [K][16], D[16][16]
  const int l = threadIdx.x;
  float4v c = {0};
  for (int k = 0; k < K; k += 4) {
    const float a = A[(l % 16) * K + k + l / 16];
    const float b = B[(k + l / 16) * 16 + l % 16];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) D[((l / 16) * 4 + i) * 16 + l % 16] = c[i];
}
int main() {
  const int K = 128;
  std::mt19937 g(5); std::normal_distribution<float> nd;
  for (int variant = 0; variant < 2; ++variant) {
    const int N = variant == 0 ? 32 : 16, step = variant == 0 ? 2 : 4;
    long chain_ok = 0, block_ok = 0, total = 0;
    for (int trial = 0; trial < 50; ++trial) {
      std::vector<float> A(N * K), B(K * N), D(N * N);
      for (auto &v : A) v = std::floor(nd(g) * 60.f);      // integer-valued like the descriptors
      for (auto &v : B) v = nd(g);
      float *dA, *dB, *dD;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      if (variant == 0) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
        float c = 0.f; for (int k = 0; k < K; ++k) c = std::fmaf(A[i * K + k], B[k * N + j], c);
        float cb = 0.f;
        for (int k = 0; k < K; k += step) { long double s = cb; for (int t = 0; t < step; ++t) s += (long double)A[i * K + k + t] * (long double)B[(k + t) * N + j]; cb = (float)s; }
        chain_ok += c == D[i * N + j]; block_ok += cb == D[i * N + j]; ++total;
      }
      hipFree(dA); hipFree(dB); hipFree(dD);
    }
    printf("%s: sequential-fma-chain match %ld/%ld, exact-block-then-round match %ld/%ld\n", variant == 0 ? "32x32x2" : "16x16x4", chain_ok, total, block_ok, total);
  }
  return 0;
}
