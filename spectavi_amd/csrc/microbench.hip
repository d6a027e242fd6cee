// microbench.hip -- fixes the roofline constants of the L1 kernel on the real part:
// sustained issue rate of v_sad_hi_u8 / v_sad_u8 / v_sad_u16 next to full-rate
// controls (v_add_u32, v_fma_f32, v_dot4), all operands in registers, plus the
// shader clock the chip holds while running them (s_memtime / s_memrealtime).
// Not on any product path; called by tools/microbench.py only.

#include "common.h"

namespace spv {
namespace {

enum Op { OP_SAD_HI = 0, OP_SAD = 1, OP_SAD_U16 = 2, OP_ADD = 3, OP_FMA = 4, OP_DOT4 = 5, OP_MED3 = 6 };

template <int OP>
__device__ __forceinline__ uint32_t apply(uint32_t q, uint32_t x, uint32_t a) {
  if (OP == OP_SAD_HI) return __builtin_amdgcn_sad_hi_u8(q, x, a);
  if (OP == OP_SAD) return __builtin_amdgcn_sad_u8(q, x, a);
  if (OP == OP_SAD_U16) return __builtin_amdgcn_sad_u16(q, x, a);
  if (OP == OP_ADD) return a + (q ^ x);  // v_xor + v_add (2 full-rate ops)
  if (OP == OP_FMA) return __float_as_uint(__builtin_fmaf(__uint_as_float(q), __uint_as_float(x), __uint_as_float(a)));
  if (OP == OP_DOT4) return __builtin_amdgcn_udot4(q, x, a, false);
  return max(min(q, a), min(max(q, a), x));  // v_med3_u32
}

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t *out, unsigned long long *clk, int iters,
                                                   uint32_t seed) {
  constexpr int ACCS = 8;
  uint32_t a[ACCS], q[8], x[8];
#pragma unroll
  for (int i = 0; i < ACCS; ++i) a[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    q[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;
    x[i] = seed ^ (0x9E3779B9u * (i + 1));
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < ACCS; ++i) a[i] = apply<OP>(q[(r + i) & 7], x[r], a[i]);
    }
    x[it & 7] += 0x00010001u;  // keep operands changing so nothing is hoisted
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < ACCS; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = c1 - c0;  // shader cycles
    clk[1] = r1 - r0;  // 100 MHz ticks
  }
}

template <int OP>
void launch(int blocks, uint32_t *out, unsigned long long *clk, int iters) {
  hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(256), 0, nullptr, out, clk, iters, 12345u);
}

}  // namespace
}  // namespace spv

// op: 0 v_sad_hi_u8, 1 v_sad_u8, 2 v_sad_u16, 3 v_xor+v_add (counted as 1), 4 v_fma_f32,
// 5 v_dot4_u32_u8, 6 v_med3_u32.  Each lane executes iters*64 ops.
extern "C" int spv_microbench_valu(int op, int blocks, int iters, double *lane_ops_per_s,
                                   double *clock_ghz) {
  using namespace spv;
  clear_error();
  if (blocks <= 0 || iters <= 0 || !lane_ops_per_s || !clock_ghz || op < 0 || op > 6)
    return set_error(SPV_ERR_INVALID, "bad args");
  int s = ensure_device();
  if (s != SPV_OK) return s;
  uint32_t *out = nullptr;
  unsigned long long *clk = nullptr;
  SPV_HIP_CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint32_t)));
  SPV_HIP_CHECK(hipMalloc(&clk, 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  SPV_HIP_CHECK(hipEventCreate(&e0));
  SPV_HIP_CHECK(hipEventCreate(&e1));
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up
    if (pass == 1) SPV_HIP_CHECK(hipEventRecord(e0, nullptr));
    switch (op) {
      case 0: launch<OP_SAD_HI>(blocks, out, clk, iters); break;
      case 1: launch<OP_SAD>(blocks, out, clk, iters); break;
      case 2: launch<OP_SAD_U16>(blocks, out, clk, iters); break;
      case 3: launch<OP_ADD>(blocks, out, clk, iters); break;
      case 4: launch<OP_FMA>(blocks, out, clk, iters); break;
      case 5: launch<OP_DOT4>(blocks, out, clk, iters); break;
      default: launch<OP_MED3>(blocks, out, clk, iters); break;
    }
  }
  SPV_HIP_CHECK(hipEventRecord(e1, nullptr));
  SPV_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SPV_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2] = {0, 1};
  SPV_HIP_CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  *lane_ops_per_s = (double)blocks * 256.0 * iters * 64.0 / (ms * 1e-3);
  *clock_ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  (void)hipFree(clk);
  return SPV_OK;
}

// ---------------------------------------------------------------------------------
// Memory ceilings used by DESIGN.md for the cascade kernels: (a) a plain 16-byte-per-lane
// streaming copy, (b) random 128-byte row gathers, 8 lanes per row (the access shape of
// probe_refine_group_kernel), from a table of `table_bytes`.
// ---------------------------------------------------------------------------------
namespace spv {
namespace {

__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4 *__restrict__ src,
                                                          uint4 *__restrict__ dst, size_t n) {
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
    dst[e] = src[e];
}

__global__ __launch_bounds__(256) void gather128_kernel(const uint4 *__restrict__ table, size_t rows,
                                                        uint32_t *__restrict__ out, int rounds,
                                                        uint32_t seed) {
  const int sub = threadIdx.x & 7;
  const size_t grp = (blockIdx.x * (size_t)256 + threadIdx.x) >> 3;
  uint32_t state = seed ^ (uint32_t)(grp * 2654435761u);
  uint32_t acc = 0;
  for (int r = 0; r < rounds; r += 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      state = state * 1664525u + 1013904223u;  // same row for the 8 lanes of a group
      const size_t row = ((size_t)state * rows) >> 32;
      v[u] = table[row * 8 + sub];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  out[blockIdx.x * (size_t)256 + threadIdx.x] = acc;
}

}  // namespace
}  // namespace spv

// mode 0: streaming copy of `table_bytes` (reports bytes read + written per second);
// mode 1: random 128-byte row gathers from a table of `table_bytes` (bytes gathered per second).
extern "C" int spv_microbench_memory(int mode, size_t table_bytes, double *bytes_per_s) {
  using namespace spv;
  clear_error();
  if (!bytes_per_s || table_bytes < 4096 || mode < 0 || mode > 1) return set_error(SPV_ERR_INVALID, "bad args");
  int s = ensure_device();
  if (s != SPV_OK) return s;
  table_bytes &= ~(size_t)127;
  void *a = nullptr, *b = nullptr;
  SPV_HIP_CHECK(hipMalloc(&a, table_bytes));
  SPV_HIP_CHECK(hipMalloc(&b, mode == 0 ? table_bytes : (size_t)2048 * 256 * 8 * 4));
  SPV_HIP_CHECK(hipMemset(a, 1, table_bytes));
  hipEvent_t e0, e1;
  SPV_HIP_CHECK(hipEventCreate(&e0));
  SPV_HIP_CHECK(hipEventCreate(&e1));
  const int rounds = 256, blocks = 2048 * 8;
  double moved = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) SPV_HIP_CHECK(hipEventRecord(e0, nullptr));
    if (mode == 0) {
      hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, nullptr, static_cast<const uint4 *>(a),
                         static_cast<uint4 *>(b), table_bytes / 16);
      moved = 2.0 * (double)table_bytes;
    } else {
      hipLaunchKernelGGL(gather128_kernel, dim3(blocks), dim3(256), 0, nullptr, static_cast<const uint4 *>(a),
                         table_bytes / 128, static_cast<uint32_t *>(b), rounds, 12345u + pass);
      moved = (double)blocks * 32.0 * rounds * 128.0;
    }
  }
  SPV_HIP_CHECK(hipEventRecord(e1, nullptr));
  SPV_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SPV_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *bytes_per_s = moved / (ms * 1e-3);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  return SPV_OK;
}
