// Experiment harness (not part of the library): does the 10M-point DLT kernel's time depend on where
// its three streams (x, xp: 24 B per point; dst: 32 B per point) lie relative to each other?
// One allocation, the three buffers placed at controlled offsets.
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
}  // namespace spv
using namespace spv;
int main() {
  const long long npt = 10000000;
  std::vector<double> x(3 * npt), xp(3 * npt);
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd(0, 1);
  Cameras cam;
  const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const double P1[12] = {0.98, -0.05, 0.19, 0.8, 0.06, 0.998, -0.03, 0.5, -0.19, 0.04, 0.98, 0.33};
  for (int i = 0; i < 12; i++) { cam.p0[i] = P0[i]; cam.p1[i] = P1[i]; }
  for (long long i = 0; i < npt; i++) {
    const double X[4] = {nd(g), nd(g), 5 + nd(g), 1};
    for (int r = 0; r < 3; r++) {
      double a = 0, b = 0;
      for (int c = 0; c < 4; c++) { a += P0[4 * r + c] * X[c]; b += P1[4 * r + c] * X[c]; }
      x[3 * i + r] = a; xp[3 * i + r] = b;
    }
    for (int r = 0; r < 2; r++) { x[3 * i + r] += 1e-3 * nd(g) * x[3 * i + 2]; xp[3 * i + r] += 1e-3 * nd(g) * xp[3 * i + 2]; }
  }
  char *base;
  const size_t MB = 1 << 20;
  if (hipMalloc(&base, 1400 * MB) != hipSuccess) return 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](size_t off_xp, size_t off_dst, int per) {
    double *dx = (double *)base, *dxp = (double *)(base + off_xp), *dd = (double *)(base + off_dst);
    hipMemcpy(dx, x.data(), 24 * npt, hipMemcpyHostToDevice);
    hipMemcpy(dxp, xp.data(), 24 * npt, hipMemcpyHostToDevice);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((dlt_kernel<false>), dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((dlt_kernel<false>), dim3(256 * per), dim3(kDltThreads), 0, 0, cam, npt, dx, dxp, dd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("xp at +%8.3f MB  dst at +%9.3f MB  %2d blk/CU  %.4f ms\n", off_xp / (double)MB, off_dst / (double)MB, per, ms / 200);
    fflush(stdout);
  };
  // what a packed allocation gives (x 228.9 MiB, xp right after, dst right after), then shifts
  const size_t xb = 24 * (size_t)npt, db = 32 * (size_t)npt;
  (void)db;
  const size_t shifts[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1 * MB, 2 * MB, 3 * MB, 5 * MB, 8 * MB, 16 * MB, 17 * MB, 32 * MB, 64 * MB};
  for (size_t s1 : {(size_t)0, (size_t)4096, (size_t)(1 * MB)})
    for (size_t s2 : shifts) run(256 * MB + s1, 512 * MB + s2, 32);
  run(xb, 2 * xb, 32);                // packed
  run(256 * MB, 512 * MB, 16);
  run(256 * MB, 1024 * MB, 32);
  return 0;
}
