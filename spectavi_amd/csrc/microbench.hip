// microbench.hip -- fixes the roofline constant of the L1 kernel on the real part:
// the sustained v_sad_hi_u8 issue rate with all operands in registers (no memory).
// Not on any product path; called by tools/ and bench diagnostics only.

#include "common.h"

namespace spv {
namespace {

template <int ACCS>
__global__ __launch_bounds__(256) void sad_rate_kernel(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[ACCS], q[8], x[8];
#pragma unroll
  for (int i = 0; i < ACCS; ++i) a[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    q[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;
    x[i] = seed ^ (0x9E3779B9u * (i + 1));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < ACCS; ++i) a[i] = __builtin_amdgcn_sad_hi_u8(q[(r + i) & 7], x[r], a[i]);
    }
    // keep the operands changing so nothing is hoisted
    x[it & 7] += 0x00010001u;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < ACCS; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace
}  // namespace spv

extern "C" int spv_microbench_sad(int blocks, int iters, double *lane_ops_per_s) {
  using namespace spv;
  clear_error();
  if (blocks <= 0 || iters <= 0 || !lane_ops_per_s) return set_error(SPV_ERR_INVALID, "bad args");
  int s = ensure_device();
  if (s != SPV_OK) return s;
  uint32_t *out = nullptr;
  SPV_HIP_CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint32_t)));
  hipEvent_t e0, e1;
  SPV_HIP_CHECK(hipEventCreate(&e0));
  SPV_HIP_CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((sad_rate_kernel<8>), dim3(blocks), dim3(256), 0, nullptr, out, iters / 8 + 1, 12345u);
  SPV_HIP_CHECK(hipEventRecord(e0, nullptr));
  hipLaunchKernelGGL((sad_rate_kernel<8>), dim3(blocks), dim3(256), 0, nullptr, out, iters, 12345u);
  SPV_HIP_CHECK(hipEventRecord(e1, nullptr));
  SPV_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SPV_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *lane_ops_per_s = (double)blocks * 256.0 * iters * 64.0 / (ms * 1e-3);  // 8 rounds x 8 accs per iter
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return SPV_OK;
}
