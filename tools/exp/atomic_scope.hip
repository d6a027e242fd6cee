// Experiment (not part of the library): cost of 2M returning histogram atomics on 131 073 counters
// (the cascade's bucket histogram: one per row and table), device scope against workgroup scope on a
// per-XCD copy of the table.  hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

template <int SCOPE>  // 0 agent (default atomicAdd), 1 workgroup scope on the XCD's copy, 2 agent no-return
__global__ __launch_bounds__(256) void hist(const uint32_t *codes, int n, uint32_t *counts, int nb, uint32_t *ranks) {
  uint32_t xcc = 0;
  if (SCOPE == 1) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t c = codes[i];
    if (SCOPE == 0) ranks[i] = atomicAdd(&counts[c], 1u);
    else if (SCOPE == 1) ranks[i] = __hip_atomic_fetch_add(&counts[(size_t)xcc * nb + c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | (xcc << 28);
    else __hip_atomic_fetch_add(&counts[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int n = 2000000, nb = 131073;
  std::vector<uint32_t> h(n);
  std::mt19937 g(1);
  for (auto &v : h) v = g() % (nb - 1);
  uint32_t *codes, *counts, *ranks;
  hipMalloc(&codes, n * 4); hipMalloc(&counts, (size_t)8 * nb * 4); hipMalloc(&ranks, n * 4);
  hipMemcpy(codes, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto kern) {
    for (int i = 0; i < 50; ++i) { hipMemsetAsync(counts, 0, (size_t)8 * nb * 4, 0); hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, codes, n, counts, nb, ranks); }
    hipDeviceSynchronize();
    float tot = 0;
    for (int i = 0; i < 20; ++i) {
      hipMemsetAsync(counts, 0, (size_t)8 * nb * 4, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, codes, n, counts, nb, ranks);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    // check: the histogram summed over copies must count every element once
    std::vector<uint32_t> c((size_t)8 * nb);
    hipMemcpy(c.data(), counts, c.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0; for (auto v : c) sum += v;
    printf("%-44s %.4f ms per 2M atomics   (counted %llu of %d)\n", name, tot / 20, sum, n);
  };
  run("device scope, returning (atomicAdd)", hist<0>);
  run("device scope, no return", hist<2>);
  run("workgroup scope on the XCD's copy, returning", hist<1>);
  return 0;
}
