// Experiment harness (not part of the library): cycles per DEPENDENT v_add_f32 of one wave, alone on
// its SIMD, for 64 / 16 active lanes, with the operand coming from a register or from LDS reads issued
// ahead -- the floor of the row-ordered float32 column sum of the normalisation (adapter.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ACTIVE>
__global__ void chain_reg(float *out, int n, long long *cyc) {
  if ((int)threadIdx.x >= ACTIVE) return;
  float s = out[threadIdx.x], a = out[64 + threadIdx.x];
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(a));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *d; long long *c;
  hipMalloc(&d, 1024); hipMalloc(&c, 8); hipMemset(d, 0, 1024);
  const int n = 1 << 20;
  for (int rep = 0; rep < 2; ++rep) {
    long long h;
    hipLaunchKernelGGL(chain_reg<64>, dim3(1), dim3(64), 0, 0, d, n, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("64 lanes: %.2f s_memtime ticks per dependent add (100 MHz ticks: x24 = core cycles at 2.4 GHz)\n", (double)h / n);
    hipLaunchKernelGGL(chain_reg<16>, dim3(1), dim3(64), 0, 0, d, n, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("16 lanes: %.2f ticks per dependent add\n", (double)h / n);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain_reg<64>, dim3(1), dim3(64), 0, 0, d, n * 16, c);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("wall: %.3f ms for %d dependent adds = %.2f ns per add\n", ms, n * 16, ms * 1e6 / (n * 16.0));
  return 0;
}
