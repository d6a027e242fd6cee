// ransac.hip -- the callers of the RANSAC candidate processing, on the device (gfx950).
//
// Reference counterpart (read as text, not copied): `seven_point_algorithm` and `ransac_fitter`
// (src/Spectavi.cpp:14-36, :70-87), i.e. FundamentalMatrixFitter::solve
// (src/FundamentalMatrixFitter.h:108-246) and RansacFitter::fit_essential
// (src/RansacFitter.h:152-272).  The reference runs one try at a time per OpenMP thread: seven
// correspondences, a 7 x 9 JacobiSVD, a cubic, then every root through process_fundamental_matrix,
// which triangulates ALL correspondences for each of four cameras (twice for the best one).
//
// Here a whole batch of tries is in flight:
//   seven_point_kernel   one lane per try: the 7 x 9 system, its two-dimensional null space, the
//                        cubic det(z F0 + (1 - z) F1) = 0, up to three candidate F per try (slots of
//                        missing roots are NaN, which the stages below treat as "no candidate")
//   ransac_process_run   (dlt.hip) gate, E, four cameras and the (camera, correspondence) scoring
//                        grid for all 3 x tries candidates at once -- this is where the time goes
//   ransac_reduce_kernel the reference's best-model rule (:196-214) over the candidates IN ORDER,
//                        carried from batch to batch in a small device-resident state
// The host loop only generates the 7-subsets, looks at 16 bytes of state after each batch (to stop
// at the first success, as the reference's `if (_success) continue` does) and fetches the winner.
//
// The null space: Eigen's JacobiSVD of a wide matrix starts with a column-pivoted Householder QR of
// the adjoint, and V.col(7), V.col(8) -- all the reference uses -- are the last two columns of that
// Q (the sweeps only touch the first seven).  The kernel runs the same published algorithm
// (Eigen 3.3/3.4 ColPivHouseholderQR::computeInPlace, makeHouseholder; see oracle/oracle_ransac.cpp
// for the restatement it is tested against) with every array in registers: loops fully unrolled, the
// pivot exchange done with selects.  Products and sums are rounded separately (the library is built
// with -ffp-contract=off), as in the reference's build.
#include "common.h"

#include <math.h>

namespace spv {
namespace {

constexpr int kFitThreads = 64;  // one wave per workgroup: a batch has a few thousand tries at most

// ---------------------------------------------------------------------------------
// null space of the 7 x 9 system
// ---------------------------------------------------------------------------------
struct Basis {
  double f0[9], f1[9];
};

// M: the 9 x 7 adjoint (already divided by the largest magnitude).  Returns the last two columns of
// the Householder Q of its column-pivoted QR.
__device__ __forceinline__ void null_space_basis(double (&M)[9][7], Basis &out) {
  const double kMin = 2.2250738585072014e-308, kEps = 2.220446049250313e-16;
  double norm_upd[7], norm_dir[7], hcoef[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += M[i][j] * M[i][j];
    norm_dir[j] = norm_upd[j] = sqrt(s);
  }
  const double downdate_threshold = sqrt(kEps);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    // pivot: the first column with the largest updated norm
    int big = k;
    double bigv = norm_upd[k];
#pragma unroll
    for (int j = k + 1; j < 7; ++j) {
      const bool gt = norm_upd[j] > bigv;
      big = gt ? j : big;
      bigv = gt ? norm_upd[j] : bigv;
    }
#pragma unroll
    for (int j = k + 1; j < 7; ++j) {
      const bool sw = (big == j);
      // rows above k belong to R, which nothing here reads again
#pragma unroll
      for (int i = k; i < 9; ++i) {
        const double a = M[i][k], b = M[i][j];
        M[i][k] = sw ? b : a;
        M[i][j] = sw ? a : b;
      }
      const double nu = norm_upd[k], nd = norm_dir[k];
      norm_upd[k] = sw ? norm_upd[j] : nu;
      norm_upd[j] = sw ? nu : norm_upd[j];
      norm_dir[k] = sw ? norm_dir[j] : nd;
      norm_dir[j] = sw ? nd : norm_dir[j];
    }
    // Householder vector of M[k..8][k]: essential part stored below the diagonal
    double tail = 0.0;
#pragma unroll
    for (int i = k + 1; i < 9; ++i) tail += M[i][k] * M[i][k];
    const double c0 = M[k][k];
    double tau;
    if (tail <= kMin) {
      tau = 0.0;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) M[i][k] = 0.0;
    } else {
      double beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0.0) beta = -beta;
      const double den = c0 - beta;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) M[i][k] = M[i][k] / den;
      tau = (beta - c0) / beta;
    }
    hcoef[k] = tau;
    if (tau != 0.0) {
#pragma unroll
      for (int j = k + 1; j < 7; ++j) {
        double tmp = 0.0;
#pragma unroll
        for (int i = k + 1; i < 9; ++i) tmp += M[i][k] * M[i][j];
        tmp += M[k][j];
        M[k][j] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 9; ++i) M[i][j] -= tau * M[i][k] * tmp;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 7; ++j) {  // norm downdate (LAPACK working note 176)
      if (norm_upd[j] != 0.0) {
        double temp = fabs(M[k][j]) / norm_upd[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double r = norm_upd[j] / norm_dir[j];
        const double temp2 = temp * (r * r);
        if (temp2 <= downdate_threshold) {
          double s = 0.0;
#pragma unroll
          for (int i = k + 1; i < 9; ++i) s += M[i][j] * M[i][j];
          norm_dir[j] = norm_upd[j] = sqrt(s);
        } else {
          norm_upd[j] *= sqrt(temp);
        }
      }
    }
  }
  // Q e7 and Q e8, Q = H0 H1 ... H6
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    out.f0[i] = (i == 7) ? 1.0 : 0.0;
    out.f1[i] = (i == 8) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 6; k >= 0; --k) {
    const double tau = hcoef[k];
    if (tau != 0.0) {
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) {
        t0 += M[i][k] * out.f0[i];
        t1 += M[i][k] * out.f1[i];
      }
      t0 += out.f0[k];
      t1 += out.f1[k];
      out.f0[k] -= tau * t0;
      out.f1[k] -= tau * t1;
#pragma unroll
      for (int i = k + 1; i < 9; ++i) {
        out.f0[i] -= tau * M[i][k] * t0;
        out.f1[i] -= tau * M[i][k] * t1;
      }
    }
  }
}

__device__ __forceinline__ double det3(const double *r0, const double *r1, const double *r2) {
  return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) +
         r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

// x^3 + a x^2 + b x + c = 0 as the reference solves it (src/FundamentalMatrixFitter.h:64-104): three
// real roots by the trigonometric form, otherwise the real root by Cardano (and the double root when
// the imaginary part of the pair is below 1e-14).  Returns the number of roots written.
__device__ __forceinline__ int solve_cubic(double (&x)[3], double a, double b, double c) {
  const double eps = 1e-14, two_pi = 6.28318530717958648;
  const double a2 = a * a;
  double q = (a2 - 3 * b) / 9;
  const double r = (a * (2 * a2 - 9 * b) + 27 * c) / 54;
  const double r2 = r * r;
  const double q3 = q * q * q;
  if (r2 < q3) {
    double t = r / sqrt(q3);
    if (t < -1) t = -1;
    if (t > 1) t = 1;
    t = acos(t);
    a /= 3;
    q = -2 * sqrt(q);
    x[0] = q * cos(t / 3) - a;
    x[1] = q * cos((t + two_pi) / 3) - a;
    x[2] = q * cos((t - two_pi) / 3) - a;
    return 3;
  }
  double A = -pow(fabs(r) + sqrt(r2 - q3), 1. / 3);
  if (r < 0) A = -A;
  const double B = A == 0 ? 0 : q / A;
  a /= 3;
  x[0] = (A + B) - a;
  x[1] = -0.5 * (A + B) - a;
  x[2] = 0.5 * sqrt(3.) * (A - B);
  if (fabs(x[2]) < eps) {
    x[2] = x[1];
    return 2;
  }
  return 1;
}

// One lane per try.  SAMPLED: the seven correspondences are rows samples[7 t ..] of the homogeneous
// x0 / x1 [npt,3], hnormalized here (src/RansacFitter.h:180-188); otherwise x0 / x1 are euclidean
// [n,7,2] (src/Spectavi.cpp:14-27).  Fs double[n,3,9]: root k of try t at Fs[(3 t + k) 9 ..], NaN
// where the try has fewer roots.  nroot int[n] (may be NULL), basis double[n,2,9] (may be NULL).
template <bool SAMPLED>
__global__ __launch_bounds__(kFitThreads) void seven_point_kernel(const double *__restrict__ x0,
                                                                  const double *__restrict__ x1,
                                                                  const int *__restrict__ samples, int n,
                                                                  double *__restrict__ Fs, int *__restrict__ nroot,
                                                                  double *__restrict__ basis) {
  const int t = blockIdx.x * kFitThreads + threadIdx.x;
  if (t >= n) return;
  double M[9][7];
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    double px, py, qx, qy;
    if (SAMPLED) {
      const size_t s = (size_t)samples[(size_t)t * 7 + i];
      const double w0 = x0[3 * s + 2], w1 = x1[3 * s + 2];
      px = x0[3 * s] / w0;
      py = x0[3 * s + 1] / w0;
      qx = x1[3 * s] / w1;
      qy = x1[3 * s + 1] / w1;
    } else {
      px = x0[(size_t)t * 14 + 2 * i];
      py = x0[(size_t)t * 14 + 2 * i + 1];
      qx = x1[(size_t)t * 14 + 2 * i];
      qy = x1[(size_t)t * 14 + 2 * i + 1];
    }
    M[0][i] = qx * px;
    M[1][i] = qx * py;
    M[2][i] = qx;
    M[3][i] = qy * px;
    M[4][i] = qy * py;
    M[5][i] = qy;
    M[6][i] = px;
    M[7][i] = py;
    M[8][i] = 1.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) scale = fmax(scale, fabs(M[j][i]));
  }
  if (scale == 0.0) scale = 1.0;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) M[j][i] = M[j][i] / scale;
  Basis b;
  null_space_basis(M, b);
  if (basis) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      basis[(size_t)t * 18 + i] = b.f0[i];
      basis[(size_t)t * 18 + 9 + i] = b.f1[i];
    }
  }
  // det(z F0 + w F1) = m0 z^3 + m1 z^2 w + m2 z w^2 + m3 w^3 with w = 1 - z
  const double *F0 = b.f0, *F1 = b.f1;
  const double m0 = det3(F0, F0 + 3, F0 + 6);
  const double m1 = det3(F1, F0 + 3, F0 + 6) + det3(F0, F1 + 3, F0 + 6) + det3(F0, F0 + 3, F1 + 6);
  const double m2 = det3(F0, F1 + 3, F1 + 6) + det3(F1, F0 + 3, F1 + 6) + det3(F1, F1 + 3, F0 + 6);
  const double m3 = det3(F1, F1 + 3, F1 + 6);
  const double ca = m0 - m1 + m2 - m3;
  const double cb = m1 - 2 * m2 + 3 * m3;
  const double cc = m2 - 3 * m3;
  const double cd = m3;
  int nr = 0;
  double alpha[3] = {0.0, 0.0, 0.0};
  if (!(fabs(ca) < 1e-14)) nr = solve_cubic(alpha, cb / ca, cc / ca, cd / ca);
  if (nroot) nroot[t] = nr;
  const double nan = __builtin_nan("");
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double z = alpha[k];
#pragma unroll
    for (int i = 0; i < 9; ++i) Fs[((size_t)t * 3 + k) * 9 + i] = k < nr ? z * F0[i] + (1 - z) * F1[i] : nan;
  }
}

// ---------------------------------------------------------------------------------
// the best-model rule over one batch of candidates, src/RansacFitter.h:196-214 with the tries in
// order: a candidate is taken when process_fundamental_matrix succeeded for it, its inlier share is
// above required (or find_best_even_in_failure) and above the best so far; the first one above
// required ends the search.  One workgroup; state persists between batches.
// ---------------------------------------------------------------------------------
struct FitState {
  int found;    // a model has been kept
  int success;  // ... and it is above required_percent_inliers: stop
  int count;    // its inliers
  int cand;     // 3 * try + root
  double F[9];
  double P[12];
};

constexpr int kReduceThreads = 256;

__global__ __launch_bounds__(kReduceThreads) void ransac_reduce_kernel(
    const int *__restrict__ ok, const int *__restrict__ count, const double *__restrict__ Fs,
    const double *__restrict__ best_P, int ncand, int cand_base, long long npt, double required_percent,
    int find_best, FitState *__restrict__ state) {
  __shared__ unsigned int s_first;
  __shared__ unsigned long long s_best;
  if (threadIdx.x == 0) {
    s_first = 0xFFFFFFFFu;
    s_best = 0ull;
  }
  __syncthreads();
  if (state->success) return;  // uniform: nothing after the first success is looked at
  unsigned int first = 0xFFFFFFFFu;
  unsigned long long best = 0ull;
  for (int i = threadIdx.x; i < ncand; i += kReduceThreads) {
    if (!ok[i]) continue;
    const int c = count[i];
    const double percent = (double)c / (double)npt;
    if (percent > required_percent) first = min(first, (unsigned int)i);
    // most inliers, the earliest among equals
    if (find_best) best = max(best, ((unsigned long long)(unsigned int)c << 32) | (0xFFFFFFFFu - (unsigned int)i));
  }
  if (first != 0xFFFFFFFFu) atomicMin(&s_first, first);
  if (best) atomicMax(&s_best, best);
  __syncthreads();
  if (threadIdx.x != 0) return;
  int win = -1, success = 0;
  if (s_first != 0xFFFFFFFFu) {
    win = (int)s_first;
    success = 1;
  } else if (s_best) {
    const int c = (int)(s_best >> 32);
    if (c > (state->found ? state->count : 0)) win = (int)(0xFFFFFFFFu - (unsigned int)(s_best & 0xFFFFFFFFu));
  }
  if (win < 0) return;
  state->found = 1;
  state->success = success;
  state->count = count[win];
  state->cand = cand_base + win;
  for (int i = 0; i < 9; ++i) state->F[i] = Fs[(size_t)win * 9 + i];
  for (int i = 0; i < 12; ++i) state->P[i] = best_P[(size_t)win * 12 + i];
}

}  // namespace

int seven_point_run(const double *d_x, const double *d_xp, int n, double *d_Fs, int *d_nroot, double *d_basis,
                    hipStream_t stream) {
  if (n < 0) return set_error(SPV_ERR_INVALID, "negative count");
  if (n == 0) return SPV_OK;
  if (!d_x || !d_xp || !d_Fs) return set_error(SPV_ERR_INVALID, "null device pointer");
  ProfScope prof("seven_point", stream);
  hipLaunchKernelGGL(seven_point_kernel<false>, dim3((n + kFitThreads - 1) / kFitThreads), dim3(kFitThreads), 0, stream,
                     d_x, d_xp, (const int *)nullptr, n, d_Fs, d_nroot, d_basis);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

size_t ransac_fit_state_bytes() { return round_up(sizeof(FitState), 256); }

// Tries per batch: three candidates each, at most 16383 candidates per ransac_process_run, and about
// 2e9 (camera, correspondence) solves so that one batch stays in the tens of milliseconds.
int ransac_fit_batch_limit(long long npt) {
  const long long by_work = 2000000000ll / (12 * std::max<long long>(npt, 1));
  return (int)std::max<long long>(1, std::min<long long>(16383 / 3, by_work));
}

size_t ransac_fit_workspace_bytes(int batch, long long npt) {
  const int nc = 3 * batch;
  size_t b = ransac_fit_state_bytes();
  b += round_up((size_t)batch * 7 * sizeof(int), 256);      // samples
  b += round_up((size_t)nc * 9 * sizeof(double), 256);      // candidate F
  b += 3 * round_up((size_t)nc * sizeof(int), 256);         // ok, inlier count, best camera
  b += round_up((size_t)nc * 12 * sizeof(double), 256);     // best camera matrices
  b += round_up((size_t)npt, 256);                          // the winner's inlier mask
  b += ransac_workspace_bytes(nc, npt, false);
  b += ransac_workspace_bytes(1, npt, true);
  return b;
}

namespace {
struct FitBuffers {
  FitState *state;
  int *samples;
  double *Fs;
  int *ok, *count, *best_cam;
  double *best_P;
  unsigned char *mask;
  void *ws;
  size_t ws_bytes;
  void *ws1;
  size_t ws1_bytes;
};

FitBuffers carve(void *d_ws, int batch, long long npt) {
  const int nc = 3 * batch;
  unsigned char *p = static_cast<unsigned char *>(d_ws);
  FitBuffers b;
  auto take = [&](size_t bytes) {
    unsigned char *q = p;
    p += round_up(bytes, 256);
    return q;
  };
  b.state = reinterpret_cast<FitState *>(take(sizeof(FitState)));
  b.samples = reinterpret_cast<int *>(take((size_t)batch * 7 * sizeof(int)));
  b.Fs = reinterpret_cast<double *>(take((size_t)nc * 9 * sizeof(double)));
  b.ok = reinterpret_cast<int *>(take((size_t)nc * sizeof(int)));
  b.count = reinterpret_cast<int *>(take((size_t)nc * sizeof(int)));
  b.best_cam = reinterpret_cast<int *>(take((size_t)nc * sizeof(int)));
  b.best_P = reinterpret_cast<double *>(take((size_t)nc * 12 * sizeof(double)));
  b.mask = take((size_t)npt);
  b.ws_bytes = ransac_workspace_bytes(nc, npt, false);
  b.ws = take(b.ws_bytes);
  b.ws1_bytes = ransac_workspace_bytes(1, npt, true);
  b.ws1 = take(b.ws1_bytes);
  return b;
}
}  // namespace

// d_x0, d_x1: double[npt,3] on the device.  next_samples(first_try, n, dst) fills the 7-subsets of
// tries [first_try, first_try + n) into host memory.  Host outputs: *success, essential double[9],
// camera double[12] (both untouched when no model was kept), *n_inliers, inlier_mask uint8[npt],
// *best_try / *best_root (-1 when none), *tries_run.
int ransac_fit_run(const double *d_x0, const double *d_x1, long long npt, double required_percent,
                   double max_error, int max_tries, int find_best, double ratio_allowed,
                   const std::function<void(int, int, int *)> &next_samples, int *success, double *essential,
                   double *camera, int *n_inliers, unsigned char *inlier_mask, int *best_try, int *best_root,
                   int *tries_run, void *d_ws, size_t ws_bytes, int batch, hipStream_t stream) {
  if (npt < 1 || max_tries < 0) return set_error(SPV_ERR_INVALID, "bad count");
  if (!d_x0 || !d_x1) return set_error(SPV_ERR_INVALID, "null device pointer");
  if (batch < 1 || batch > 16383 / 3) return set_error(SPV_ERR_INVALID, "batch %d", batch);
  if (!d_ws || ws_bytes < ransac_fit_workspace_bytes(batch, npt))
    return set_error(SPV_ERR_INVALID, "workspace too small: %zu < %zu", ws_bytes, ransac_fit_workspace_bytes(batch, npt));
  const FitBuffers b = carve(d_ws, batch, npt);
  SPV_HIP_CHECK(hipMemsetAsync(b.state, 0, sizeof(FitState), stream));
  // two host buffers: the subsets of the next batch are drawn while the device works on this one
  std::vector<int> host_a((size_t)batch * 7), host_b((size_t)batch * 7);
  int *cur = host_a.data(), *nxt = host_b.data();
  FitState head;  // only the four ints are read back per batch
  head.found = head.success = head.count = 0;
  head.cand = -1;
  int done = 0;
  // easy problems succeed within a few tries: start small -- 256 tries, fewer when a try is expensive
  // (about 1e8 (camera, correspondence) solves in the first batch, but at least 8 tries) -- and grow
  // fourfold to the full batch.  With a million correspondences a batch of 174 tries held some fifty
  // models that pass the gate, each scored in full, when the first of them already ended the search.
  int step = std::min(batch, (int)std::max<long long>(8, std::min<long long>(256, 100000000ll / (12 * std::max<long long>(npt, 1)))));
  auto draw = [&](int first, int n, int *dst) -> int {
    next_samples(first, n, dst);
    for (size_t i = 0; i < (size_t)n * 7; ++i)
      if (dst[i] < 0 || dst[i] >= npt) return set_error(SPV_ERR_INVALID, "sample index %d outside [0, %lld)", dst[i], npt);
    return SPV_OK;
  };
  if (max_tries > 0) SPV_TRY(draw(0, std::min(step, max_tries), cur));
  while (done < max_tries) {
    const int n = std::min(step, max_tries - done);
    // (pageable source: the copy has left `cur` when the call returns)
    SPV_HIP_CHECK(hipMemcpyAsync(b.samples, cur, (size_t)n * 7 * sizeof(int), hipMemcpyHostToDevice, stream));
    {
      ProfScope prof("seven_point", stream);
      hipLaunchKernelGGL(seven_point_kernel<true>, dim3((n + kFitThreads - 1) / kFitThreads), dim3(kFitThreads), 0, stream,
                         d_x0, d_x1, (const int *)b.samples, n, b.Fs, (int *)nullptr, (double *)nullptr);
      SPV_HIP_CHECK(hipGetLastError());
    }
    SPV_TRY(ransac_process_run(b.Fs, 3 * n, npt, d_x0, d_x1, ratio_allowed, required_percent, max_error, find_best, b.ok,
                               b.count, b.best_cam, b.best_P, nullptr, nullptr, nullptr, nullptr, b.ws, b.ws_bytes, stream,
                               2048));
    {
      ProfScope prof("ransac_reduce", stream);
      hipLaunchKernelGGL(ransac_reduce_kernel, dim3(1), dim3(kReduceThreads), 0, stream, (const int *)b.ok,
                         (const int *)b.count, (const double *)b.Fs, (const double *)b.best_P, 3 * n, 3 * done, npt,
                         required_percent, find_best, b.state);
      SPV_HIP_CHECK(hipGetLastError());
    }
    SPV_HIP_CHECK(hipMemcpyAsync(&head, b.state, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
    const int next_step = std::min(batch, step * 4);
    const int n_next = std::min(next_step, max_tries - (done + n));
    if (n_next > 0) SPV_TRY(draw(done + n, n_next, nxt));  // while the batch runs
    SPV_HIP_CHECK(hipStreamSynchronize(stream));
    done += n;
    if (head.success) break;
    step = next_step;
    std::swap(cur, nxt);
  }
  if (tries_run) *tries_run = done;
  *success = head.success;
  *n_inliers = head.found ? head.count : 0;
  if (best_try) *best_try = head.found ? head.cand / 3 : -1;
  if (best_root) *best_root = head.found ? head.cand % 3 : -1;
  if (!head.found) return SPV_OK;
  // the winner's inlier list: its candidate once more, this time with the mask (same kernels, same
  // arithmetic per (camera, correspondence), hence the same best camera and the same count)
  SPV_TRY(ransac_process_run(b.state->F, 1, npt, d_x0, d_x1, ratio_allowed, required_percent, max_error, find_best, b.ok,
                             b.count, b.best_cam, b.best_P, nullptr, nullptr, nullptr, b.mask, b.ws1, b.ws1_bytes, stream));
  FitState full;
  SPV_HIP_CHECK(hipMemcpyAsync(&full, b.state, sizeof(FitState), hipMemcpyDeviceToHost, stream));
  SPV_HIP_CHECK(hipMemcpyAsync(inlier_mask, b.mask, (size_t)npt, hipMemcpyDeviceToHost, stream));
  SPV_HIP_CHECK(hipStreamSynchronize(stream));
  for (int i = 0; i < 9; ++i) essential[i] = full.F[i];
  for (int i = 0; i < 12; ++i) camera[i] = full.P[i];
  return SPV_OK;
}

}  // namespace spv
