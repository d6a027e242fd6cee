// Experiment harness (not part of the library): times variants of the DLT kernel.
#include "../../spectavi_amd/csrc/dlt.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cstdarg>
namespace spv {
int set_error(int s, const char *, ...) { return s; }
void clear_error() {}
int device_cu_count() { return 256; }
ProfScope::ProfScope(const char *n, hipStream_t s) : name_(n), stream_(s) {}
ProfScope::~ProfScope() {}
namespace {
// V1: memory pattern only
__global__ __launch_bounds__(kDltThreads) void v1_mem(Cameras cam, long long npt, const double *__restrict__ x,
                                                       const double *__restrict__ xp, double *__restrict__ dst) {
  __shared__ double sx[kDltThreads * 3];
  __shared__ double sxp[kDltThreads * 3];
  const long long base = (long long)blockIdx.x * kDltThreads;
  const long long nblk = min((long long)kDltThreads, npt - base);
  for (int e = threadIdx.x; e < nblk * 3; e += kDltThreads) { sx[e] = x[base * 3 + e]; sxp[e] = xp[base * 3 + e]; }
  __syncthreads();
  const int t = threadIdx.x;
  if (t >= nblk) return;
  double4 *o = reinterpret_cast<double4 *>(dst) + (base + t);
  *o = make_double4(sx[3 * t] + cam.p0[0], sx[3 * t + 1], sxp[3 * t + 2], sxp[3 * t]);
}
// V3: compute only (inputs synthesised, store never taken)
__global__ __launch_bounds__(kDltThreads) void v3_compute(Cameras cam, long long npt, const double *__restrict__ x,
                                                           const double *__restrict__ xp, double *__restrict__ dst) {
  const long long base = (long long)blockIdx.x * kDltThreads;
  const int t = threadIdx.x;
  if (base + t >= npt) return;
  const double f = (double)((base + t) & 1023) * 1e-3;
  double X[4], u, v, up, vp;
  const double Xw = f - 0.5, Yw = 0.3 - f, Zw = 5.0 + f;
  double a[3], b[3];
  for (int r = 0; r < 3; ++r) {
    a[r] = cam.p0[4 * r] * Xw + cam.p0[4 * r + 1] * Yw + cam.p0[4 * r + 2] * Zw + cam.p0[4 * r + 3];
    b[r] = cam.p1[4 * r] * Xw + cam.p1[4 * r + 1] * Yw + cam.p1[4 * r + 2] * Zw + cam.p1[4 * r + 3];
  }
  a[0] += 1e-3 * f * a[2]; b[1] -= 1e-3 * f * b[2];
  dlt_solve<true>(cam, a[0], a[1], a[2], b[0], b[1], b[2], X, u, v, up, vp);
  if (X[0] == 123.456) { double4 *o = reinterpret_cast<double4 *>(dst) + (base + t); *o = make_double4(X[0], X[1], X[2], X[3]); }
}
// V2: persistent blocks, grid-stride over tiles, register prefetch of the next tile
__global__ __launch_bounds__(kDltThreads) void v2_persist(Cameras cam, long long npt, const double *__restrict__ x,
                                                           const double *__restrict__ xp, double *__restrict__ dst) {
  __shared__ double sx[kDltThreads * 3];
  __shared__ double sxp[kDltThreads * 3];
  const long long ntiles = (npt + kDltThreads - 1) / kDltThreads;
  const int t = threadIdx.x;
  double px[3], pp[3];
  auto prefetch = [&](long long tile) {
    const long long base = tile * kDltThreads;
    const long long n3 = min((long long)kDltThreads, npt - base) * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = t + i * kDltThreads;
      px[i] = e < n3 ? x[base * 3 + e] : 1.0;
      pp[i] = e < n3 ? xp[base * 3 + e] : 1.0;
    }
  };
  long long tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kDltThreads;
    const long long nblk = min((long long)kDltThreads, npt - base);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) { sx[t + i * kDltThreads] = px[i]; sxp[t + i * kDltThreads] = pp[i]; }
    __syncthreads();
    if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);
    if (t < nblk) {
      double X[4], u, v, up, vp;
      dlt_solve<true>(cam, sx[3 * t], sx[3 * t + 1], sx[3 * t + 2], sxp[3 * t], sxp[3 * t + 1], sxp[3 * t + 2], X, u, v, up, vp);
      double4 *o = reinterpret_cast<double4 *>(dst) + (base + t);
      *o = make_double4(X[0], X[1], X[2], X[3]);
    }
  }
}

// V4: independent waves, persistent, each lane prefetches its own point (no LDS, no barriers)
__global__ __launch_bounds__(kDltThreads) void v4_wave(Cameras cam, long long npt, const double *__restrict__ x,
                                                        const double *__restrict__ xp, double *__restrict__ dst) {
  const long long stride = (long long)gridDim.x * kDltThreads;
  long long p = (long long)blockIdx.x * kDltThreads + threadIdx.x;
  double a0 = 1, a1 = 1, a2 = 1, b0 = 1, b1 = 1, b2 = 1;
  if (p < npt) { a0 = x[3 * p]; a1 = x[3 * p + 1]; a2 = x[3 * p + 2]; b0 = xp[3 * p]; b1 = xp[3 * p + 1]; b2 = xp[3 * p + 2]; }
  for (; p < npt; p += stride) {
    const double c0 = a0, c1 = a1, c2 = a2, d0 = b0, d1 = b1, d2 = b2;
    const long long q = p + stride;
    if (q < npt) { a0 = x[3 * q]; a1 = x[3 * q + 1]; a2 = x[3 * q + 2]; b0 = xp[3 * q]; b1 = xp[3 * q + 1]; b2 = xp[3 * q + 2]; }
    double X[4], u, v, up, vp;
    dlt_solve<true>(cam, c0, c1, c2, d0, d1, d2, X, u, v, up, vp);
    double4 *o = reinterpret_cast<double4 *>(dst) + p;
    *o = make_double4(X[0], X[1], X[2], X[3]);
  }
}
// V5: independent waves, persistent, coalesced prefetch + wave-private LDS transposition
__global__ __launch_bounds__(kDltThreads) void v5_wave_lds(Cameras cam, long long npt, const double *__restrict__ x,
                                                            const double *__restrict__ xp, double *__restrict__ dst) {
  __shared__ double sx[kDltThreads * 3];
  __shared__ double sxp[kDltThreads * 3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *wx = sx + wv * 192, *wp = sxp + wv * 192;
  const long long nw = (npt + 63) / 64;
  const long long wstride = (long long)gridDim.x * (kDltThreads / 64);
  long long w = (long long)blockIdx.x * (kDltThreads / 64) + wv;
  double px[3], pp[3];
  auto prefetch = [&](long long ww) {
    const long long base = ww * 64;
    const long long n3 = min(64LL, npt - base) * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = lane + 64 * i;
      px[i] = e < n3 ? x[base * 3 + e] : 1.0;
      pp[i] = e < n3 ? xp[base * 3 + e] : 1.0;
    }
  };
  if (w < nw) prefetch(w);
  for (; w < nw; w += wstride) {
    const long long base = w * 64;
#pragma unroll
    for (int i = 0; i < 3; ++i) { wx[lane + 64 * i] = px[i]; wp[lane + 64 * i] = pp[i]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double c0 = wx[3 * lane], c1 = wx[3 * lane + 1], c2 = wx[3 * lane + 2];
    const double d0 = wp[3 * lane], d1 = wp[3 * lane + 1], d2 = wp[3 * lane + 2];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (w + wstride < nw) prefetch(w + wstride);
    if (base + lane < npt) {
      double X[4], u, v, up, vp;
      dlt_solve<true>(cam, c0, c1, c2, d0, d1, d2, X, u, v, up, vp);
      double4 *o = reinterpret_cast<double4 *>(dst) + (base + lane);
      *o = make_double4(X[0], X[1], X[2], X[3]);
    }
  }
}

// V6: v4 with the prefetch two points deep (12 more VGPRs, twice the bytes in flight per wave)
__global__ __launch_bounds__(kDltThreads) void v6_wave2(Cameras cam, long long npt, const double *__restrict__ x,
                                                         const double *__restrict__ xp, double *__restrict__ dst) {
  const long long stride = (long long)gridDim.x * kDltThreads;
  long long p = (long long)blockIdx.x * kDltThreads + threadIdx.x;
  double a[2][3], b[2][3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long q = p + k * stride;
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[k][i] = 1.0; b[k][i] = 1.0; }
    if (q < npt) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { a[k][i] = x[3 * q + i]; b[k][i] = xp[3 * q + i]; }
    }
  }
  for (; p < npt; p += 2 * stride) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const long long cur = p + k * stride;
      const double c0 = a[k][0], c1 = a[k][1], c2 = a[k][2], d0 = b[k][0], d1 = b[k][1], d2 = b[k][2];
      const long long q = cur + 2 * stride;
      if (q < npt) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { a[k][i] = x[3 * q + i]; b[k][i] = xp[3 * q + i]; }
      }
      if (cur < npt) {
        double X[4], u, v, up, vp;
        dlt_solve<true>(cam, c0, c1, c2, d0, d1, d2, X, u, v, up, vp);
        double4 *o = reinterpret_cast<double4 *>(dst) + cur;
        *o = make_double4(X[0], X[1], X[2], X[3]);
      }
    }
  }
}

// V7: independent waves, persistent; the 1536-byte span of each view travels HBM -> LDS directly
// (global_load_lds_dwordx4, 16 bytes per lane, no staging registers), NS batches in flight per wave.
template <int NS>
__global__ __launch_bounds__(kDltThreads) void v7_ldsdma(Cameras cam, long long npt, const double *__restrict__ x,
                                                          const double *__restrict__ xp, double *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 waves][NS][2 views][1536 B]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  char *wbase = smem + (size_t)wv * NS * 3072;
  const long long nw = (npt + 63) / 64;
  const long long wstride = (long long)gridDim.x * (kDltThreads / 64);
  long long w = (long long)blockIdx.x * (kDltThreads / 64) + wv;
  typedef const __attribute__((address_space(1))) void *gptr;
  typedef __attribute__((address_space(3))) void *lptr;
  auto issue = [&](long long ww, int slot) {
    if (ww >= nw) return;  // wave-uniform
    const long long base = ww * 64;
    const long long nbytes = min(64LL, npt - base) * 24;  // valid bytes of the span
    const char *gx = reinterpret_cast<const char *>(x) + base * 24;
    const char *gp = reinterpret_cast<const char *>(xp) + base * 24;
    char *l = wbase + slot * 3072;
    if (lane * 16 < nbytes) {
      __builtin_amdgcn_global_load_lds((gptr)(gx + lane * 16), (lptr)(l), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(gp + lane * 16), (lptr)(l + 1536), 16, 0, 0);
    }
    if (lane < 32 && 1024 + lane * 16 < nbytes) {
      __builtin_amdgcn_global_load_lds((gptr)(gx + 1024 + lane * 16), (lptr)(l + 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(gp + 1024 + lane * 16), (lptr)(l + 1536 + 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(w + s * wstride, s);
  int slot = 0;
  for (; w < nw; w += wstride) {
    issue(w + (NS - 1) * wstride, (slot + NS - 1) % NS);
    // the oldest batch has landed when at most the (NS-1) younger batches' loads are outstanding;
    // every batch issues up to 4 loads, so wait for all but 4*(NS-1) (conservative at the tail)
    if (NS == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);        // vmcnt(4)
    else if (NS == 3) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);   // vmcnt(8)
    else __builtin_amdgcn_s_waitcnt(0x0F70 | 12);               // vmcnt(12)
    if (w + (NS - 1) * wstride >= nw) __builtin_amdgcn_s_waitcnt(0x0F70);  // tail: vmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const double *lx = reinterpret_cast<const double *>(wbase + slot * 3072);
    const double *lp = reinterpret_cast<const double *>(wbase + slot * 3072 + 1536);
    const long long base = w * 64;
    double c0 = 1, c1 = 1, c2 = 1, d0 = 1, d1 = 1, d2 = 1;
    if (base + lane < npt) {
      c0 = lx[3 * lane]; c1 = lx[3 * lane + 1]; c2 = lx[3 * lane + 2];
      d0 = lp[3 * lane]; d1 = lp[3 * lane + 1]; d2 = lp[3 * lane + 2];
    }
    __builtin_amdgcn_wave_barrier();
    if (base + lane < npt) {
      double X[4], u, v, up, vp;
      dlt_solve<true>(cam, c0, c1, c2, d0, d1, d2, X, u, v, up, vp);
      double4 *o = reinterpret_cast<double4 *>(dst) + (base + lane);
      *o = make_double4(X[0], X[1], X[2], X[3]);
    }
    slot = (slot + 1) % NS;
  }
}

// V8: v4 with non-temporal loads and stores (the stream is touched once: keep it out of the caches)
template <int WPE>
__global__ __launch_bounds__(kDltThreads, WPE) void v8_nt(Cameras cam, long long npt, const double *__restrict__ x,
                                                           const double *__restrict__ xp, double *__restrict__ dst) {
  const long long stride = (long long)gridDim.x * kDltThreads;
  long long p = (long long)blockIdx.x * kDltThreads + threadIdx.x;
  double a0 = 1, a1 = 1, a2 = 1, b0 = 1, b1 = 1, b2 = 1;
  auto ld = [&](long long q) {
    a0 = __builtin_nontemporal_load(x + 3 * q); a1 = __builtin_nontemporal_load(x + 3 * q + 1); a2 = __builtin_nontemporal_load(x + 3 * q + 2);
    b0 = __builtin_nontemporal_load(xp + 3 * q); b1 = __builtin_nontemporal_load(xp + 3 * q + 1); b2 = __builtin_nontemporal_load(xp + 3 * q + 2);
  };
  if (p < npt) ld(p);
  for (; p < npt; p += stride) {
    const double c0 = a0, c1 = a1, c2 = a2, d0 = b0, d1 = b1, d2 = b2;
    const long long q = p + stride;
    if (q < npt) ld(q);
    double X[4], u, v, up, vp;
    dlt_solve<true>(cam, c0, c1, c2, d0, d1, d2, X, u, v, up, vp);
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v *o = reinterpret_cast<d2v *>(dst + 4 * p);
    __builtin_nontemporal_store(d2v{X[0], X[1]}, o);
    __builtin_nontemporal_store(d2v{X[2], X[3]}, o + 1);
  }
}
}  // namespace
}  // namespace spv
using namespace spv;
int main(int argc, char **argv) {
  const long long npt = 10000000;
  std::mt19937_64 g(1); std::normal_distribution<double> nd;
  Cameras cam; double P0[12] = {1,0,0,0, 0,1,0,0, 0,0,1,0};
  double M[3][3]; for (auto &r : M) for (auto &v : r) v = nd(g);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < i; j++) { double d = 0; for (int k = 0; k < 3; k++) d += M[i][k]*M[j][k]; for (int k = 0; k < 3; k++) M[i][k] -= d*M[j][k]; } double n = 0; for (int k = 0; k < 3; k++) n += M[i][k]*M[i][k]; n = std::sqrt(n); for (int k = 0; k < 3; k++) M[i][k] /= n; }
  for (int i = 0; i < 12; i++) cam.p0[i] = P0[i];
  for (int i = 0; i < 3; i++) { for (int k = 0; k < 3; k++) cam.p1[4*i+k] = M[i][k]; cam.p1[4*i+3] = nd(g); }
  std::vector<double> x(3 * npt), xp(3 * npt);
  for (long long i = 0; i < npt; i++) { double X[4] = {nd(g), nd(g), nd(g) + 5, 1};
    for (int r = 0; r < 3; r++) { double a = 0, b = 0; for (int c = 0; c < 4; c++) { a += cam.p0[4*r+c]*X[c]; b += cam.p1[4*r+c]*X[c]; } x[3*i+r] = a; xp[3*i+r] = b; }
    for (int r = 0; r < 2; r++) { x[3*i+r] += 1e-3*nd(g)*x[3*i+2]; xp[3*i+r] += 1e-3*nd(g)*xp[3*i+2]; } }
  double *dx, *dxp, *dd, *dd2;
  hipMalloc(&dx, 24 * npt); hipMalloc(&dxp, 24 * npt); hipMalloc(&dd, 32 * npt); hipMalloc(&dd2, 32 * npt);
  hipMemcpy(dx, x.data(), 24 * npt, hipMemcpyHostToDevice); hipMemcpy(dxp, xp.data(), 24 * npt, hipMemcpyHostToDevice);
  const unsigned blocks = (unsigned)((npt + kDltThreads - 1) / kDltThreads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *name, auto launch) {
    // warm up for ~0.3 s: the first launches after an idle period (the H2D copies above, the D2H
    // copies of compare()) run at a lower clock state -- 20 launches (3 ms) were not enough and made
    // whatever variant came first after a pause look 8 % slower than the identical code later on
    for (int i = 0; i < 2000; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.4f ms\n", name, ms / 200); fflush(stdout);
  };
  timeit("v0 shipped (32 blk/CU)", [&] { hipLaunchKernelGGL((dlt_kernel<false>), dim3(256 * 32), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd); });
  timeit("v0 one point per lane", [&] { hipLaunchKernelGGL((dlt_kernel<false>), dim3(blocks), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  timeit("v1 memory only", [&] { hipLaunchKernelGGL(v1_mem, dim3(blocks), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  timeit("v3 compute only", [&] { hipLaunchKernelGGL(v3_compute, dim3(blocks), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  for (int per : {2, 4, 5, 6, 8, 16}) {
    char nm[64]; snprintf(nm, sizeof nm, "v2 persistent %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL(v2_persist, dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  }
  for (int per : {4, 5, 8, 16, 32}) {
    char nm[64]; snprintf(nm, sizeof nm, "v4 wave direct %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL(v4_wave, dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  }
  for (int per : {4, 5, 8, 16, 32}) {
    char nm[64]; snprintf(nm, sizeof nm, "v5 wave lds %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL(v5_wave_lds, dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  }
  auto compare = [&](const char *name) {
    std::vector<double> a(4 * npt), b(4 * npt);
    hipMemcpy(a.data(), dd, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dd2, b.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
    printf("%s vs v0 mismatches over all points: %zu\n", name, bad); fflush(stdout);
    hipMemset(dd2, 0, 32 * npt);
  };
  for (int per : {8, 16, 32}) {
    char nm[64]; snprintf(nm, sizeof nm, "v6 wave depth2 %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL(v6_wave2, dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  }
  compare("v6");
  for (int per : {4, 8}) {
    char nm[64]; snprintf(nm, sizeof nm, "v7 lds-dma NS=2 %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL((v7_ldsdma<2>), dim3(256 * per), dim3(kDltThreads), 4 * 2 * 3072, 0, cam, npt, dx, dxp, dd2); });
  }
  compare("v7<2>");
  for (int per : {4, 8}) {
    char nm[64]; snprintf(nm, sizeof nm, "v7 lds-dma NS=3 %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL((v7_ldsdma<3>), dim3(256 * per), dim3(kDltThreads), 4 * 3 * 3072, 0, cam, npt, dx, dxp, dd2); });
  }
  compare("v7<3>");
  for (int per : {4}) {
    char nm[64]; snprintf(nm, sizeof nm, "v7 lds-dma NS=4 %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL((v7_ldsdma<4>), dim3(256 * per), dim3(kDltThreads), 4 * 4 * 3072, 0, cam, npt, dx, dxp, dd2); });
  }
  compare("v7<4>");
  for (int per : {8, 16, 32}) {
    char nm[64]; snprintf(nm, sizeof nm, "v8 nt ld/st wpe4 %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL((v8_nt<4>), dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  }
  compare("v8<4>");
  for (int per : {8, 16, 32}) {
    char nm[64]; snprintf(nm, sizeof nm, "v8 nt ld/st wpe5 %d blk/CU", per);
    timeit(nm, [&] { hipLaunchKernelGGL((v8_nt<5>), dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  }
  compare("v8<5>");
  timeit("v0 shipped again", [&] { hipLaunchKernelGGL((dlt_kernel<false>), dim3(256 * 32), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd2); });
  std::vector<double> a(4 * 100000), b(4 * 100000);
  hipMemcpy(a.data(), dd, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dd2, b.size() * 8, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
  printf("v2 vs v0 mismatches in first 100k points: %zu\n", bad);
  return 0;
}
