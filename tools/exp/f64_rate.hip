// Experiment (not part of the library): issue cost of the fp64 instructions the DLT solve is made
// of, one wave's independent stream per SIMD at full occupancy, and the accuracy of v_rcp_f64 /
// v_rsq_f64 with 0, 1, 2 Newton steps.  hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

enum { FMA, MUL, ADD, RCP, RSQ, SQRT, DIVSCALE, DIVFMAS, DIVFIXUP, LDEXP, FREXPM, FREXPE, MAX, CMP, CNDMASK, NOPS };
static const char *kNames[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
                               "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_ldexp_f64",
                               "v_frexp_mant_f64", "v_frexp_exp_i32_f64", "v_max_f64", "v_cmp_gt_f64", "v_cndmask_b32 x2"};

template <int OP>
__device__ __forceinline__ double apply(double a, double b, double c) {
  double r;
  if (OP == FMA) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  else if (OP == MUL) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else if (OP == ADD) asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else if (OP == RCP) asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(a));
  else if (OP == RSQ) asm volatile("v_rsq_f64 %0, %1" : "=v"(r) : "v"(a));
  else if (OP == SQRT) asm volatile("v_sqrt_f64 %0, %1" : "=v"(r) : "v"(a));
  else if (OP == DIVSCALE) asm volatile("v_div_scale_f64 %0, vcc, %1, %2, %1" : "=v"(r) : "v"(a), "v"(b) : "vcc");
  else if (OP == DIVFMAS) asm volatile("v_div_fmas_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
  else if (OP == DIVFIXUP) asm volatile("v_div_fixup_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  else if (OP == LDEXP) asm volatile("v_ldexp_f64 %0, %1, 3" : "=v"(r) : "v"(a));
  else if (OP == FREXPM) asm volatile("v_frexp_mant_f64 %0, %1" : "=v"(r) : "v"(a));
  else if (OP == FREXPE) { int e; asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(e) : "v"(a)); r = __hiloint2double(e, e); }
  else if (OP == MAX) asm volatile("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else if (OP == CMP) { asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc"); r = a; }
  else { int lo = __double2loint(a), hi = __double2hiint(a), lo2 = __double2loint(b), hi2 = __double2hiint(b);
         asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(lo) : "v"(lo), "v"(lo2) : "vcc");
         asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(hi) : "v"(hi), "v"(hi2) : "vcc");
         r = __hiloint2double(hi, lo); }
  return r;
}

template <int OP>
__global__ __launch_bounds__(256) void rate(double *out, unsigned long long *clk, int iters) {
  double a[8], b = 1.0000001 + threadIdx.x * 1e-9, c = 0.5;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = apply<OP>(a[i], b, c);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

__global__ void accuracy(const double *x, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  o[6 * i + 0] = r;
  double e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r);
  o[6 * i + 1] = r;
  e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r);
  o[6 * i + 2] = r;
  const double a = fabs(v);
  double q = __builtin_amdgcn_rsq(a);
  o[6 * i + 3] = q;
  // Newton for 1/sqrt(a): q += q * (1 - a q^2) / 2
  double h = 0.5 * q, t = __builtin_fma(-a * q, q, 1.0); q = __builtin_fma(h, t, q);
  o[6 * i + 4] = q;
  h = 0.5 * q; t = __builtin_fma(-a * q, q, 1.0); q = __builtin_fma(h, t, q);
  o[6 * i + 5] = q;
}

template <int OP>
void run(double *out, unsigned long long *clk, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / h[1] * 0.1;
  // blocks * 4 waves, each `iters * 64` instructions; SIMDs = 1024; waves per SIMD = blocks * 4 / 1024
  const double inst_per_simd = (double)blocks * 4 / 1024.0 * iters * 64.0;
  const double cycles = ms * 1e-3 * ghz * 1e9;
  printf("%-22s %6.2f cycles/wave-instruction/SIMD  (%.3f ms, %.2f GHz)\n", kNames[OP], cycles / inst_per_simd, ms, ghz);
}

int main() {
  const int blocks = 256 * 8, iters = 2000;  // 8 workgroups (32 waves) per CU = 8 waves per SIMD
  double *out; unsigned long long *clk;
  hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&clk, 16);
  run<FMA>(out, clk, blocks, iters); run<MUL>(out, clk, blocks, iters); run<ADD>(out, clk, blocks, iters);
  run<RCP>(out, clk, blocks, iters); run<RSQ>(out, clk, blocks, iters); run<SQRT>(out, clk, blocks, iters);
  run<DIVSCALE>(out, clk, blocks, iters); run<DIVFMAS>(out, clk, blocks, iters); run<DIVFIXUP>(out, clk, blocks, iters);
  run<LDEXP>(out, clk, blocks, iters); run<FREXPM>(out, clk, blocks, iters); run<FREXPE>(out, clk, blocks, iters);
  run<MAX>(out, clk, blocks, iters); run<CMP>(out, clk, blocks, iters); run<CNDMASK>(out, clk, blocks, iters);
  // accuracy
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * (size_t)n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> mant(1.0, 2.0);
  std::uniform_int_distribution<int> ex(-300, 300);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(mant(g), ex(g)) * ((i & 1) ? -1 : 1);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(accuracy, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double worst[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double r = 1.0L / (long double)x[i], q = 1.0L / sqrtl(fabsl((long double)x[i]));
    for (int k = 0; k < 6; ++k) {
      const long double ref = k < 3 ? r : q;
      const double rel = (double)fabsl(((long double)o[6 * (size_t)i + k] - ref) / ref);
      if (rel > worst[k]) worst[k] = rel;
    }
  }
  printf("max relative error over %d values (2^-53 = 1.11e-16):\n", n);
  printf("  v_rcp_f64 %.3e | +1 Newton %.3e | +2 Newton %.3e\n", worst[0], worst[1], worst[2]);
  printf("  v_rsq_f64 %.3e | +1 Newton %.3e | +2 Newton %.3e\n", worst[3], worst[4], worst[5]);
  return 0;
}
