// Experiment (not part of the library): does the projection kernels' read pattern cost HBM bandwidth?
// A wave of project_mfma4_kernel reads, per 32-dim chunk, a 128-byte piece of each of its 64 rows
// (rows 512 bytes apart) and comes back to the same rows for the next piece a few microseconds later.
//   mode 0: every wave streams contiguous 8 KB blocks (64 lanes x 16 B x 8 loads)
//   mode 1: the projection's pattern: 8 loads cover 64 rows x 128 B, four passes over the rows
// hipcc --offload-arch=gfx950 -O3 -o strided_read strided_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void rd(const f4 *__restrict__ x, long long nrows, float *out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long row0 = ((long long)blockIdx.x * 4 + wv) * 64;
  if (row0 >= nrows) return;
  f4 acc = {0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = lane + 64 * j;
      long long idx;  // in units of 16 bytes; a row is 32 of them
      if (MODE == 0) idx = row0 * 32 + (long long)(c * 8 + j) * 64 + lane;         // contiguous 8 KB per (c)
      else idx = (row0 + (e >> 3)) * 32 + c * 8 + (e & 7);                          // 64 rows x 128 B
      const f4 v = __builtin_nontemporal_load(x + idx);
      acc += v;
    }
    // a chunk's worth of dependent work between the passes, like the MFMA phase (keeps the passes apart)
    for (int k = 0; k < 64; ++k) acc = acc * 1.0001f + 0.5f;
  }
  if (acc[0] == 123.f) out[0] = acc[1] + acc[2] + acc[3];
}

int main() {
  const long long nrows = 1000000;
  f4 *x; float *out;
  hipMalloc(&x, nrows * 512); hipMalloc(&out, 4);
  hipMemset(x, 0, nrows * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto kern) {
    const unsigned blocks = (unsigned)((nrows + 255) / 256);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, nrows, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, nrows, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 100;
    printf("%-52s %.4f ms  %.2f TB/s\n", name, ms, nrows * 512.0 / ms / 1e9);
  };
  run("contiguous 8 KB blocks per wave and pass", rd<0>);
  run("64 rows x 128-byte pieces, four passes (projection)", rd<1>);
  run("contiguous again", rd<0>);
  return 0;
}
