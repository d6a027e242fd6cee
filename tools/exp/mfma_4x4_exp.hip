// Experiment: layout and rounding of v_mfma_f32_4x4x1_16B_f32 (16 blocks of D[4x4] += A[4x1] B[1x4]).
// Hypothesis: lane l supplies A[block l/4][i = l%4] and B[block l/4][j = l%4]; D[block][i][j] sits in
// register i of lane 4*block + j; one instruction is ONE fused multiply-add per element, so a chain
// over k equals std::fmaf applied in k order.  Also times it against v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k4(const float *A, const float *B, float *D, int K) {  // A [64 rows][K], B [K][4 cols] -> D [64][4]
  const int l = threadIdx.x;
  float4v c = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l * K + k], B[k * 4 + (l & 3)], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * (l >> 2) + i) * 4 + (l & 3)] = c[i];
}
__global__ void rate4(float *out, int iters) {
  float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void rate16(float *out, int iters) {
  float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  const int K = 128;
  std::mt19937 g(7); std::normal_distribution<float> nd;
  long ok = 0, total = 0;
  for (int trial = 0; trial < 200; ++trial) {
    std::vector<float> A(64 * K), B(K * 4), D(64 * 4);
    for (auto &v : A) v = trial % 2 ? std::floor(nd(g) * 60.f) : nd(g) * 60.f;
    for (auto &v : B) v = nd(g);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 64; ++r) for (int j = 0; j < 4; ++j) {
      float c = 0.f; for (int k = 0; k < K; ++k) c = std::fmaf(A[r * K + k], B[k * 4 + j], c);
      ok += c == D[r * 4 + j]; ++total;
    }
    hipFree(dA); hipFree(dB); hipFree(dD);
  }
  printf("4x4x1_16B: sequential-fma-chain match %ld/%ld (layout: A row = lane, B col = lane %% 4, D[i] = row 4*(lane/4)+i, col lane %% 4)\n", ok, total);
  float *out; hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(rate4, dim3(1024), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(rate16, dim3(1024), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 1024 blocks x 4 waves over 256 CUs x 4 SIMDs = 4 waves per SIMD, each 4*iters instructions
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * 4 * iters);
    printf("%s: %.3f ms, ~%.1f cycles per instruction per SIMD at 2.4 GHz\n", which == 0 ? "4x4x1_16B" : "16x16x4", ms, cyc);
  }
  return 0;
}
