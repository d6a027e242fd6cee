// dlt.hip -- batched two-view DLT triangulation for gfx950 (MI355X).
//
// Replaces the reference's serial per-point loop (src/Spectavi.cpp:48-51, :64-67)
// around DltTriangulator::solve / reprojection_error
// (src/DltTriangulator.h:36-74).  The reference runs Eigen::JacobiSVD on a
// heap-allocated dynamic 4x4 per point; here one lane owns one point at a time and the
// whole 4x4 problem lives in registers (fp64):
//
//   A = [u P0[2]-P0[0]; v P0[2]-P0[1]; u' P1[2]-P1[0]; v' P1[2]-P1[1]]   (:51-54)
//   X = right singular vector of the smallest singular value = V.col(3) of the
//   reference (:56-58), by square-root-free Gram-Schmidt + inverse iteration (~400 fp64
//   instructions) with a one-sided (Hestenes) Jacobi fallback (~3100, always
//   converges; right rotations orthogonalise the columns of A, V accumulates them,
//   the column of smallest norm is the answer).
//
// Every fused multiply-add is written explicitly and implicit contraction is off
// (-ffp-contract=off).  Reciprocals and reciprocal square roots are v_rcp_f64 / v_rsq_f64 plus
// two Newton steps (< 1 ulp) instead of the IEEE division / square-root sequences (round 3: they
// were a third of the solve's issue time), so the host mirror of this operation sequence
// (oracle/oracle_dlt_mirror.cpp, exact 1/x and 1/sqrt(x) in their place) agrees to a few ulps
// times the conditioning of the point, not bit for bit.
//
// Sign: Eigen's sign of V.col(3) is arbitrary; the result is canonicalised to
// X[3] >= 0 (first nonzero component positive when X[3] == 0).

#include "common.h"

namespace spv {
namespace {

constexpr int kDltThreads = 256;
constexpr int kMaxSweeps = 30;

struct Cameras {
  double p0[12];
  double p1[12];
};

// ---- fp64 reciprocals without the IEEE sequences ------------------------------------------
// Measured on gfx950 (tools/exp/f64_rate.hip, profiles/r03_f64_rate.txt): v_fma / v_mul / v_add /
// v_div_scale / v_div_fmas / v_div_fixup / v_ldexp _f64 issue every 4 cycles per SIMD, v_rcp_f64 /
// v_rsq_f64 / v_sqrt_f64 every 16 and are good to 2^-24 (4.6e-8); one Newton step brings them to
// 2e-15, two to 1.1e-16 / 1.4e-16 (worst of 1M values, against 2^-53 = 1.1e-16).  An IEEE x / y
// compiles to 2 v_div_scale + v_rcp + 7 fma / mul + v_div_fmas + v_div_fixup = ~60 issue cycles, an
// IEEE sqrt to ~80; rcp_nr2 is 32, rsqrt_nr2 44.
__device__ __forceinline__ double rcp_nr2(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
// the same, keeping IEEE's answers at the ends: 1/0 = inf, 1/inf = 0 (Newton turns both into nan)
__device__ __forceinline__ double rcp_nr2_ieee(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r0, 1.0);
  double r = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return (r0 == 0.0 || __builtin_isinf(r0)) ? r0 : r;
}
__device__ __forceinline__ double rsqrt_nr2(double x) {
  double q = __builtin_amdgcn_rsq(x);
  double t = __builtin_fma(-(x * q), q, 1.0);
  q = __builtin_fma(0.5 * q, t, q);
  t = __builtin_fma(-(x * q), q, 1.0);
  return __builtin_fma(0.5 * q, t, q);
}
// sqrt(s) for s >= 0 as s * rsqrt(s); zero and the range whose reciprocal root would overflow or
// lose bits take the IEEE instruction sequence (exact zeros occur: noise-free points reproject exactly)
__device__ __forceinline__ double sqrt_fast(double s) {
  if (!(s >= 0x1p-900)) return sqrt(s);
  return s * rsqrt_nr2(s);
}

// ---- null vector, method 1: one-sided (Hestenes) Jacobi, always converges ----------------
__device__ __forceinline__ void null_jacobi(const double (&A0)[4][4], double (&xv)[4]) {
  double A[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) A[r][c] = A0[r][c];
  double V[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;

  const double eps2 = 1e-30;  // (1e-15)^2
  for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          alpha = __builtin_fma(A[i][p], A[i][p], alpha);
          beta = __builtin_fma(A[i][q], A[i][q], beta);
          gamma = __builtin_fma(A[i][p], A[i][q], gamma);
        }
        // skip test |gamma| > eps*sqrt(alpha*beta), squared to avoid the square root
        if (gamma * gamma > eps2 * (alpha * beta)) {
          rotated = true;
          // tan of the rotation angle, t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with
          // zeta = (beta - alpha) / (2 gamma), rearranged to one sqrt and one division
          const double dd = beta - alpha;
          const double g2 = 2.0 * gamma;
          const double hh = sqrt(__builtin_fma(dd, dd, g2 * g2));
          const double tn = g2 / (dd + (dd < 0.0 ? -hh : hh));
          const double cs = rsqrt_nr2(__builtin_fma(tn, tn, 1.0));  // argument >= 1
          const double sn = cs * tn;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double ap = A[i][p], aq = A[i][q];
            A[i][p] = __builtin_fma(cs, ap, -(sn * aq));
            A[i][q] = __builtin_fma(sn, ap, cs * aq);
            const double vp_ = V[i][p], vq_ = V[i][q];
            V[i][p] = __builtin_fma(cs, vp_, -(sn * vq_));
            V[i][q] = __builtin_fma(sn, vp_, cs * vq_);
          }
        }
      }
    }
    if (!rotated) break;
  }
  // smallest column norm -> null direction
  double best = 0.0;
  int kbest = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double nn = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) nn = __builtin_fma(A[i][c], A[i][c], nn);
    if (c == 0 || nn < best) {
      best = nn;
      kbest = c;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    xv[i] = kbest == 0 ? V[i][0] : (kbest == 1 ? V[i][1] : (kbest == 2 ? V[i][2] : V[i][3]));
}

// ---- null vector, method 2: square-root-free Gram-Schmidt + inverse iteration ------------
// A = Q U with orthogonal (not normalised) columns q_j, d_j = |q_j|^2 and U unit upper
// triangular (modified Gram-Schmidt, no pivoting: only U and d are used, and those are
// backward stable whatever the column order), so A^T A = U^T D U.  The smallest right singular
// vector of A is found by inverse iteration on U^T D U: two unit-triangular solves and one
// diagonal scaling per step, contraction (sigma4/sigma3)^2.  Division-free after the three
// reciprocals 1/d_0..1/d_2 the orthogonalisation needs: the diagonal solve is scaled by d_3
// (y_j = z_j d_3 / d_j, y_3 = z_3), which also keeps the un-normalised iterates from growing by
// 1/sigma4^2 per step (d_3 >= sigma4^2 is the squared distance of the last column from the span of
// the others: the growth per step is d_3 / sigma4^2, O(1) for a consistent point), and the
// convergence test compares DIRECTIONS of consecutive un-normalised iterates,
//   max_i |w_i max|v| - v_i max|w|| <= 1e-13 max|v| max|w|
// (no sign ambiguity: (A^T A)^-1 is positive definite).  Steps 0 (a bare back-substitution from
// e4), 1 and 2 run unconditionally, then a point stops at the first step that moved its direction
// by no more than 1e-13 (step 3 for pixel noise 1e-3; at most 8): what is left of the direction error
// is then 1e-13 rho / (1 - rho), rho = (sigma4/sigma3)^2 the contraction (1e-12 until fuzz seed
// 20261004 case 5249 -- an inconsistent pair, rho = 0.036 -- came out 3.4e-14 from the oracle).
// ~300 fp64 instructions.
// Returns false when the last step still moved it (ill-separated sigma3, sigma4), or the
// iterate left the range (inf / nan from a degenerate A, overflow of a badly conditioned one);
// the caller then falls back to method 1.
__device__ __forceinline__ bool null_gs_inverse_iteration(const double (&A0)[4][4], double (&xv)[4]) {
  double col[4][4];  // col[c][r]
  double U[4][4];    // strictly upper part used
  double g[3];       // d_3 / d_j
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) col[c][r] = A0[r][c];
  double tiny2 = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double d = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) d = __builtin_fma(col[j][r], col[j][r], d);
    if (j == 0) tiny2 = 4.930380657631324e-32 * d;  // eps^2 |a_0|^2
    if (!(d > tiny2)) d = tiny2;
    if (j == 3) {
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] *= d;
      break;
    }
    const double idj = rcp_nr2(d);
    g[j] = idj;
#pragma unroll
    for (int k = j + 1; k < 4; ++k) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) s = __builtin_fma(col[j][r], col[k][r], s);
      const double u = s * idj;
      U[j][k] = u;
#pragma unroll
      for (int r = 0; r < 4; ++r) col[k][r] = __builtin_fma(-u, col[j][r], col[k][r]);
    }
  }
  double v3 = 1.0;
  double v2 = -U[2][3];
  double v1 = __builtin_fma(-U[1][2], v2, -U[1][3]);
  double v0 = __builtin_fma(-U[0][1], v1, __builtin_fma(-U[0][2], v2, -U[0][3]));
  double w0, w1, w2, w3;
  auto step = [&]() {  // w = d_3 (U^T D U)^-1 v
    // U^T z = v
    const double z0 = v0;
    const double z1 = __builtin_fma(-U[0][1], z0, v1);
    const double z2 = __builtin_fma(-U[1][2], z1, __builtin_fma(-U[0][2], z0, v2));
    const double z3 = __builtin_fma(-U[2][3], z2, __builtin_fma(-U[1][3], z1, __builtin_fma(-U[0][3], z0, v3)));
    // D y = d_3 z
    const double y0 = z0 * g[0], y1 = z1 * g[1], y2 = z2 * g[2];
    // U w = y
    w3 = z3;
    w2 = __builtin_fma(-U[2][3], w3, y2);
    w1 = __builtin_fma(-U[1][3], w3, __builtin_fma(-U[1][2], w2, y1));
    w0 = __builtin_fma(-U[0][3], w3, __builtin_fma(-U[0][2], w2, __builtin_fma(-U[0][1], w1, y0)));
  };
#pragma unroll
  for (int it = 1; it <= 2; ++it) {
    step();
    v0 = w0;
    v1 = w1;
    v2 = w2;
    v3 = w3;
  }
  double bv = fmax(fmax(fabs(v0), fabs(v1)), fmax(fabs(v2), fabs(v3)));
  bool ok = false;
  for (int it = 3; it < 8; ++it) {
    step();
    const double bw = fmax(fmax(fabs(w0), fabs(w1)), fmax(fabs(w2), fabs(w3)));
    const double e0 = fabs(__builtin_fma(w0, bv, -(v0 * bw))), e1 = fabs(__builtin_fma(w1, bv, -(v1 * bw)));
    const double e2 = fabs(__builtin_fma(w2, bv, -(v2 * bw))), e3 = fabs(__builtin_fma(w3, bv, -(v3 * bw)));
    const double bound = 1e-13 * (bw * bv);
    // a bound of 0 or inf is an iterate out of range, never agreement
    ok = (fmax(fmax(e0, e1), fmax(e2, e3)) <= bound) && (bound >= 1e-290) && (bound <= 1e290);
    v0 = w0;
    v1 = w1;
    v2 = w2;
    v3 = w3;
    bv = bw;
    if (ok) break;
  }
  if (!ok) return false;
  xv[0] = v0;
  xv[1] = v1;
  xv[2] = v2;
  xv[3] = v3;
  return true;
}

// FAST: method 2 with method 1 as fallback (triangulate / reprojection error: consistent
// correspondences, sigma4 << sigma3); !FAST: method 1 only (RANSAC scoring, where most
// hypotheses are inconsistent and method 2 would rarely converge).
// A = [u P0[2]-P0[0]; v P0[2]-P0[1]; u' P1[2]-P1[0]; v' P1[2]-P1[1]] (reference src/DltTriangulator.h:38-54)
__device__ __forceinline__ void dlt_matrix(const Cameras &cam, double x0, double x1, double x2, double y0,
                                           double y1, double y2, double (&A)[4][4], double &u, double &v,
                                           double &up, double &vp) {
  const double ix = rcp_nr2_ieee(x2), iy = rcp_nr2_ieee(y2);  // one reciprocal per view; x / 0 stays +-inf or nan
  u = x0 * ix;
  v = x1 * ix;
  up = y0 * iy;
  vp = y1 * iy;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    A[0][c] = __builtin_fma(u, cam.p0[8 + c], -cam.p0[0 + c]);
    A[1][c] = __builtin_fma(v, cam.p0[8 + c], -cam.p0[4 + c]);
    A[2][c] = __builtin_fma(up, cam.p1[8 + c], -cam.p1[0 + c]);
    A[3][c] = __builtin_fma(vp, cam.p1[8 + c], -cam.p1[4 + c]);
  }
}

// unit 2-norm and the canonical sign.  xv is an un-normalised iterate of any magnitude (or a
// column of the Jacobi's V): it is first brought to max-norm in [0.5, 1) by an exact power of two,
// so the sum of squares neither overflows nor underflows.
__device__ __forceinline__ void dlt_finish(const double (&xv)[4], double (&X)[4]) {
  const double big = fmax(fmax(fabs(xv[0]), fabs(xv[1])), fmax(fabs(xv[2]), fabs(xv[3])));
  const int ex = -__builtin_amdgcn_frexp_exp(big);  // 0 for big = 0 / inf / nan
  double sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sv[i] = __builtin_amdgcn_ldexp(xv[i], ex);
  double nrm2 = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) nrm2 = __builtin_fma(sv[i], sv[i], nrm2);
  bool neg = false;
  if (xv[3] != 0.0)
    neg = xv[3] < 0.0;
  else if (xv[0] != 0.0)
    neg = xv[0] < 0.0;
  else if (xv[1] != 0.0)
    neg = xv[1] < 0.0;
  else
    neg = xv[2] < 0.0;
  const double q = rsqrt_nr2(nrm2);
  const double scale = neg ? -q : q;
#pragma unroll
  for (int i = 0; i < 4; ++i) X[i] = sv[i] * scale;
}

// reprojection_error() of reference src/DltTriangulator.h:61-62, 67-74 for a solved point: the sum
// of the two image-plane residual norms.  One reciprocal per camera; z0 / z1 = the depths P X [2].
__device__ __forceinline__ double reprojection_error(const Cameras &cam, const double (&X)[4], double u, double v,
                                                     double up, double vp, double &z0, double &z1) {
  double r0[3], r1[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a = __builtin_fma(cam.p0[4 * r + c], X[c], a);
      b = __builtin_fma(cam.p1[4 * r + c], X[c], b);
    }
    r0[r] = a;
    r1[r] = b;
  }
  z0 = r0[2];
  z1 = r1[2];
  const double i0 = rcp_nr2_ieee(r0[2]), i1 = rcp_nr2_ieee(r1[2]);
  const double e0x = __builtin_fma(r0[0], i0, -u), e0y = __builtin_fma(r0[1], i0, -v);
  const double e1x = __builtin_fma(r1[0], i1, -up), e1y = __builtin_fma(r1[1], i1, -vp);
  return sqrt_fast(__builtin_fma(e0x, e0x, e0y * e0y)) + sqrt_fast(__builtin_fma(e1x, e1x, e1y * e1y));
}

// FAST: method 2 with method 1 as fallback (consistent correspondences converge in 3-4 steps;
// whatever does not within 8 takes the Jacobi); !FAST: method 1 only.
// UNIT = false: X is only brought to max-norm [0.5, 1) by a power of two, neither normalised nor
// sign-canonicalised -- enough for the reprojection error, which is a ratio of components of P X.
template <bool FAST, bool UNIT = true>
__device__ __forceinline__ void dlt_solve(const Cameras &cam, double x0, double x1, double x2,
                                          double y0, double y1, double y2, double (&X)[4],
                                          double &u, double &v, double &up, double &vp) {
  double A[4][4];
  dlt_matrix(cam, x0, x1, x2, y0, y1, y2, A, u, v, up, vp);
  double xv[4];
  bool done = false;
  if (FAST) done = null_gs_inverse_iteration(A, xv);
  if (!done) null_jacobi(A, xv);
  if (UNIT) {
    dlt_finish(xv, X);
  } else {
    const double big = fmax(fmax(fabs(xv[0]), fabs(xv[1])), fmax(fabs(xv[2]), fabs(xv[3])));
    const int ex = -__builtin_amdgcn_frexp_exp(big);
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] = __builtin_amdgcn_ldexp(xv[i], ex);
  }
}

// Persistent, barrier-free: every lane walks the points p, p + stride, ... and loads the six
// input doubles of its next point before solving the current one, so the HBM stream (a wave
// covers 64 x 24 contiguous bytes per view, 64 x 32 on the way out) overlaps the fp64 work of the
// same wave instead of relying on other workgroups' phases (measured on 10M points: 0.208 ms with
// one LDS-staged tile per workgroup, 0.178 ms this way; memory pattern alone 0.143 ms, solve
// alone 0.127 ms).
template <bool WANT_ERROR>
__global__ __launch_bounds__(kDltThreads) void dlt_kernel(Cameras cam, long long npt,
                                                          const double *__restrict__ x,
                                                          const double *__restrict__ xp,
                                                          double *__restrict__ dst) {
  const long long stride = (long long)gridDim.x * kDltThreads;
  long long p = (long long)blockIdx.x * kDltThreads + threadIdx.x;
  double a0 = 1.0, a1 = 1.0, a2 = 1.0, b0 = 1.0, b1 = 1.0, b2 = 1.0;
  if (p < npt) {
    a0 = x[3 * p];
    a1 = x[3 * p + 1];
    a2 = x[3 * p + 2];
    b0 = xp[3 * p];
    b1 = xp[3 * p + 1];
    b2 = xp[3 * p + 2];
  }
  for (; p < npt; p += stride) {
    const double c0 = a0, c1 = a1, c2 = a2, d0 = b0, d1 = b1, d2 = b2;
    const long long q = p + stride;
    if (q < npt) {
      a0 = x[3 * q];
      a1 = x[3 * q + 1];
      a2 = x[3 * q + 2];
      b0 = xp[3 * q];
      b1 = xp[3 * q + 1];
      b2 = xp[3 * q + 2];
    }
    double X[4], u, v, up, vp;
    dlt_solve<true, !WANT_ERROR>(cam, c0, c1, c2, d0, d1, d2, X, u, v, up, vp);
    if (!WANT_ERROR) {
      double4 *o = reinterpret_cast<double4 *>(dst) + p;
      *o = make_double4(X[0], X[1], X[2], X[3]);
    } else {
      double z0, z1;
      dst[p] = reprojection_error(cam, X, u, v, up, vp, z0, z1);
    }
  }
}


// ---------------------------------------------------------------------------------
// RANSAC hypothesis scoring: the reference scores every
// candidate second camera by triangulating ALL correspondences and counting
// those with reprojection error <= threshold that lie in front of both cameras
// (src/RansacFitter.h:59-73, src/DltTriangulator.h:67-86).  grid = (point
// blocks, hypotheses); one lane = one (point, hypothesis); per-wave ballot
// counts feed one atomicAdd per wave.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double det3_left(const double *P) {
  return P[0] * (P[5] * P[10] - P[6] * P[9]) - P[1] * (P[4] * P[10] - P[6] * P[8]) +
         P[2] * (P[4] * P[9] - P[5] * P[8]);
}

// reprojection_error() <= max_error && is_infront_both_cameras() (src/DltTriangulator.h:67-86)
__device__ __forceinline__ bool score_inlier(const Cameras &cam, const double (&X)[4], double u, double v,
                                             double up, double vp, double max_error) {
  double z0, z1;
  const double err = reprojection_error(cam, X, u, v, up, vp, z0, z1);
  const double s0 = det3_left(cam.p0) < 0 ? -1.0 : 1.0, s1 = det3_left(cam.p1) < 0 ? -1.0 : 1.0;
  const double n0 = cam.p0[2] * cam.p0[2] + cam.p0[6] * cam.p0[6] + cam.p0[10] * cam.p0[10];
  const double n1 = cam.p1[2] * cam.p1[2] + cam.p1[6] * cam.p1[6] + cam.p1[10] * cam.p1[10];
  const double dc0 = s0 / n0 * z0 / X[3];
  const double dc1 = s1 / n1 * z1 / X[3];
  return (err <= max_error) && (dc0 > 0) && (dc1 > 0);
}

// Every (correspondence, hypothesis) pair is solved by method 2 (8 steps) with method 1 as its
// fallback, exactly as dlt_solve<true> does it for one point.  Most hypotheses of a RANSAC run are
// wrong for most correspondences, and about one solve in five does not converge within the 8 steps
// (sigma4 / sigma3 has a median of 0.02 but a long tail); a wave that ran the ~3100-instruction Jacobi
// for its slow lanes in place would pay for it with all 64.  With a work list (TWO_PASS) the slow
// lanes are appended to it instead -- one atomicAdd per wave, (hypothesis, point) packed in 8 bytes --
// and dlt_score_fallback_kernel solves them afterwards in full waves.  Lanes that find the list full,
// and the form without a work list, run the fallback in place: the results do not depend on the path.
// The list is cut into kListShards segments with a counter each (64 bytes apart): every wave with
// slow lanes does one atomicAdd, and 70 000 of them on ONE address serialise at ~12 ns each (measured:
// the first version of this kernel spent 0.83 of its 0.9 ms there).
constexpr unsigned int kListShards = 1024;
constexpr unsigned int kCountStride = 16;  // uints between two counters
struct WorkList {
  unsigned int *count;          // [kListShards * kCountStride]; may exceed `segment`: the excess ran in place
  unsigned long long *entries;  // [kListShards][segment], (hypothesis << 40) | point
  unsigned int segment;         // entries per shard; 0 = no list
};

// cand / ncand (RANSAC candidate processing): the candidates that passed the gate, in any order;
// row y then scores camera y % 4 of candidate cand[y / 4], and the grid has only as many rows
// (the caller's cap -- 2 048 for a RANSAC batch --, each workgroup walking y, y + gridDim.y, ...) as a batch plausibly needs: almost
// every candidate of a RANSAC batch is gated, and a grid row per gated camera is a row of workgroups
// that start only to find a NaN and leave -- 500 000 of them per batch of 5 461 tries, 1.2 of the
// batch's 2 ms in launch rate alone.  Without a list row y is hypothesis y.
template <bool TWO_PASS>
__global__ __launch_bounds__(kDltThreads) void dlt_score_kernel(
    Cameras cam0, const double *__restrict__ p1s, int nhyp, long long npt, const double *__restrict__ x,
    const double *__restrict__ xp, double max_error, int *__restrict__ counts,
    unsigned char *__restrict__ mask, WorkList wl, const int *__restrict__ cand, const int *__restrict__ ncand) {
  __shared__ double sx[kDltThreads * 3];
  __shared__ double sxp[kDltThreads * 3];
  const long long base = (long long)blockIdx.x * kDltThreads;
  const long long nblk = min((long long)kDltThreads, npt - base);
  // the workgroup's correspondences are staged once and serve every grid row it walks
  for (int e = threadIdx.x; e < nblk * 3; e += kDltThreads) {
    sx[e] = x[base * 3 + e];
    sxp[e] = xp[base * 3 + e];
  }
  __syncthreads();
  const int t = threadIdx.x;
  const bool live = t < nblk;
  const int nrows = cand ? 4 * *ncand : nhyp;  // workgroup-uniform
  for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
    const int h = cand ? 4 * cand[y >> 2] + (y & 3) : y;
    Cameras cam;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      cam.p0[i] = cam0.p0[i];
      cam.p1[i] = p1s[(size_t)h * 12 + i];  // wave-uniform
    }
    if (cam.p1[0] != cam.p1[0]) {  // NaN camera = a gated RANSAC candidate (workgroup-uniform): nothing is an inlier
      if (mask && live) mask[(size_t)h * npt + base + t] = 0;
      continue;
    }
    double A[4][4], xv[4], u = 0, v = 0, up = 0, vp = 0;
    bool solved = true;  // dead lanes have nothing to solve
    if (live) {
      dlt_matrix(cam, sx[3 * t], sx[3 * t + 1], sx[3 * t + 2], sxp[3 * t], sxp[3 * t + 1], sxp[3 * t + 2], A, u, v,
                 up, vp);
      solved = null_gs_inverse_iteration(A, xv);
    }
    bool deferred = false;
    if (TWO_PASS) {
      // slow lanes of the wave -> work list
      const unsigned long long slow = __ballot(!solved);
      if (slow) {
        const int lane = threadIdx.x & 63;
        const unsigned int shard = ((unsigned)y * gridDim.x + blockIdx.x) * (kDltThreads / 64) + (threadIdx.x >> 6);
        const unsigned int sh = shard % kListShards;
        unsigned int first = 0;
        if (lane == 0) first = atomicAdd(wl.count + sh * kCountStride, (unsigned int)__popcll(slow));
        first = __shfl(first, 0, 64);
        if (!solved) {
          const unsigned int slot = first + (unsigned int)__popcll(slow & ((1ull << lane) - 1ull));
          if (slot < wl.segment) {
            wl.entries[(size_t)sh * wl.segment + slot] = ((unsigned long long)h << 40) | (unsigned long long)(base + t);
            deferred = true;
          }
        }
      }
    }
    if (!solved && !deferred) null_jacobi(A, xv);  // in place: no work list, or it is full
    bool inlier = false;
    if (live && !deferred) {
      double X[4];
      dlt_finish(xv, X);
      inlier = score_inlier(cam, X, u, v, up, vp, max_error);
      if (mask) mask[(size_t)h * npt + base + t] = inlier ? 1 : 0;
    }
    const unsigned long long bal = __ballot(inlier);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[h], __popcll(bal));
  }
}

// Second pass: one lane per work-list entry, method 1.
__global__ __launch_bounds__(kDltThreads) void dlt_score_fallback_kernel(
    Cameras cam0, const double *__restrict__ p1s, long long npt, const double *__restrict__ x,
    const double *__restrict__ xp, double max_error, int *__restrict__ counts,
    unsigned char *__restrict__ mask, WorkList wl) {
  const unsigned int total = kListShards * wl.segment;
  for (unsigned int i = blockIdx.x * kDltThreads + threadIdx.x; i < total; i += gridDim.x * kDltThreads) {
    // slot-major over the shards: consecutive lanes take the same slot of consecutive shards, so the
    // filled part of the list (the first slots of every shard) is covered by full waves
    const unsigned int sh = i % kListShards, slot = i / kListShards;
    if (slot >= min(wl.count[sh * kCountStride], wl.segment)) continue;
    const unsigned long long e = wl.entries[(size_t)sh * wl.segment + slot];
    const int h = (int)(e >> 40);
    const long long p = (long long)(e & ((1ull << 40) - 1ull));
    Cameras cam;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      cam.p0[k] = cam0.p0[k];
      cam.p1[k] = p1s[(size_t)h * 12 + k];
    }
    double A[4][4], xv[4], X[4], u, v, up, vp;
    dlt_matrix(cam, x[3 * p], x[3 * p + 1], x[3 * p + 2], xp[3 * p], xp[3 * p + 1], xp[3 * p + 2], A, u, v, up, vp);
    null_jacobi(A, xv);
    dlt_finish(xv, X);
    const bool inlier = score_inlier(cam, X, u, v, up, vp, max_error);
    if (mask) mask[(size_t)h * npt + p] = inlier ? 1 : 0;
    if (inlier) atomicAdd(&counts[h], 1);
  }
}

// ---------------------------------------------------------------------------------
// RANSAC candidate processing (the rest of it): what the reference's
// process_fundamental_matrix does with ONE candidate F (src/RansacFitter.h:42-95), batched over
// many candidates:  SVD of F, singular-value-ratio gate (:49-53), E = U diag(1,1,0) V^T (:54-56),
// Essential2Cameras (src/Camera.h:31-46: a second SVD, t = U.col(2), Ra = U D V^T, Rb = U D^T V^T,
// cameras (Ra,t) (Ra,-t) (Rb,t) (Rb,-t)), every camera scored over all correspondences by
// dlt_score_kernel above, the best camera kept by the rule of :74-94.
//
// Which of the four cameras is "Ra, +t" depends on the signs Eigen's JacobiSVD happens to give the
// columns of U and V (sigma3 = 0 leaves u3 and v3 unpaired: det(U V^T) may be -1 and the reference
// does not correct it).  To stay a drop-in, the SVD here is the same published two-sided Jacobi
// iteration (Eigen 3.3 / 3.4 JacobiSVD.h, real_2x2_jacobi_svd, makeJacobi: scale by the largest
// entry, sweeps over (p,q) = (1,0) (2,0) (2,1) until every off-diagonal pair is below
// 2 eps * max|diagonal|, signs into U, selection sort), with products and sums rounded separately as
// the reference's build (no -march, no FMA) rounds them.
// ---------------------------------------------------------------------------------
struct JRot {  // J = [c s; -s c]
  double c, s;
};

template <int P, int Q>
__device__ __forceinline__ void rot_rows(double (&M)[3][3], JRot j) {  // B = J B on rows P, Q
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double xi = M[P][i], yi = M[Q][i];
    M[P][i] = j.c * xi + j.s * yi;
    M[Q][i] = -j.s * xi + j.c * yi;
  }
}

template <int P, int Q>
__device__ __forceinline__ void rot_cols(double (&M)[3][3], JRot j) {  // B = B J on columns P, Q
  const double c = j.c, s = -j.s;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double xi = M[i][P], yi = M[i][Q];
    M[i][P] = c * xi + s * yi;
    M[i][Q] = -s * xi + c * yi;
  }
}

__device__ __forceinline__ JRot make_jacobi(double x, double y, double z) {
  const double kMin = 2.2250738585072014e-308;
  const double deno = 2.0 * fabs(y);
  if (deno < kMin) return JRot{1.0, 0.0};
  const double tau = (x - z) / deno;
  const double w = sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / sqrt(t * t + 1.0);
  return JRot{n, -sign_t * (y / fabs(y)) * fabs(t) * n};
}

template <int P, int Q>
__device__ __forceinline__ void jacobi_pair(double (&W)[3][3], double (&U)[3][3], double (&V)[3][3],
                                            double &max_diag, bool &finished) {
  const double kMin = 2.2250738585072014e-308, kPrecision = 2.0 * 2.220446049250313e-16;
  const double threshold = fmax(kMin, kPrecision * max_diag);
  if (fabs(W[P][Q]) > threshold || fabs(W[Q][P]) > threshold) {
    finished = false;
    // 2x2 real Jacobi SVD of [[W(p,p) W(p,q)] [W(q,p) W(q,q)]]
    double m00 = W[P][P], m01 = W[P][Q], m10 = W[Q][P], m11 = W[Q][Q];
    JRot rot1;
    const double t = m00 + m11;
    const double d = m10 - m01;
    if (fabs(d) < kMin) {
      rot1 = JRot{1.0, 0.0};
    } else {
      const double u = t / d;
      const double tmp = sqrt(1.0 + u * u);
      rot1 = JRot{u / tmp, 1.0 / tmp};
    }
    {  // m = rot1 applied to its rows
      const double a0 = rot1.c * m00 + rot1.s * m10, a1 = rot1.c * m01 + rot1.s * m11;
      const double b1 = -rot1.s * m01 + rot1.c * m11;
      m00 = a0;
      m01 = a1;
      m11 = b1;
    }
    const JRot j_right = make_jacobi(m00, m01, m11);
    const JRot jrt{j_right.c, -j_right.s};
    const JRot j_left{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
    rot_rows<P, Q>(W, j_left);
    rot_cols<P, Q>(U, JRot{j_left.c, -j_left.s});
    rot_cols<P, Q>(W, j_right);
    rot_cols<P, Q>(V, j_right);
    max_diag = fmax(max_diag, fmax(fabs(W[P][P]), fabs(W[Q][Q])));
  }
}

template <int I, int J>
__device__ __forceinline__ void swap_cols_if(bool c, double (&S)[3], double (&U)[3][3], double (&V)[3][3]) {
  const double s = S[I];
  S[I] = c ? S[J] : S[I];
  S[J] = c ? s : S[J];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double u = U[r][I], v = V[r][I];
    U[r][I] = c ? U[r][J] : u;
    U[r][J] = c ? u : U[r][J];
    V[r][I] = c ? V[r][J] : v;
    V[r][J] = c ? v : V[r][J];
  }
}

// A = U diag(S) V^T, S descending.
__device__ __forceinline__ void jacobi_svd3(const double (&A)[3][3], double (&U)[3][3], double (&S)[3],
                                            double (&V)[3][3]) {
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) scale = fmax(scale, fabs(A[i][j]));
  if (scale == 0.0) scale = 1.0;
  double W[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      W[i][j] = A[i][j] / scale;
      U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  double max_diag = fmax(fabs(W[0][0]), fmax(fabs(W[1][1]), fabs(W[2][2])));
  bool finished = false;
  // a 3x3 converges in a handful of sweeps; the cap only bounds the loop for lanes fed nan / inf
  // (their comparisons are false, as in the reference, so they normally leave after one sweep)
  for (int sweep = 0; sweep < 60 && !finished; ++sweep) {
    finished = true;
    jacobi_pair<1, 0>(W, U, V, max_diag, finished);
    jacobi_pair<2, 0>(W, U, V, max_diag, finished);
    jacobi_pair<2, 1>(W, U, V, max_diag, finished);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a = W[i][i];
    S[i] = fabs(a) * scale;
    if (a < 0.0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) U[r][i] = -U[r][i];
    }
  }
  // selection sort, first maximum wins
  {
    int pos = 0;
    if (S[1] > S[0]) pos = 1;
    if (S[2] > (pos == 1 ? S[1] : S[0])) pos = 2;
    swap_cols_if<0, 1>(pos == 1, S, U, V);
    swap_cols_if<0, 2>(pos == 2, S, U, V);
    swap_cols_if<1, 2>(S[2] > S[1], S, U, V);
  }
}

__device__ __forceinline__ void mat3_mul(const double (&A)[3][3], const double (&B)[3][3], double (&C)[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) a += A[i][k] * B[k][j];
      C[i][j] = a;
    }
}

// One lane per candidate F: gate, E, the four cameras.  cams double[nF,4,12]; gated candidates (and
// candidates with a NaN entry) get NaN cameras (the scoring kernel skips them) and gated[f] = 1.
__global__ __launch_bounds__(kDltThreads) void essential_cameras_kernel(
    const double *__restrict__ Fs, int nF, double ratio_allowed, double *__restrict__ cams,
    double *__restrict__ ratio_out, double *__restrict__ E_out, int *__restrict__ gated,
    int *__restrict__ live, int *__restrict__ nlive) {
  const int f = blockIdx.x * kDltThreads + threadIdx.x;
  if (f >= nF) return;
  double F[3][3], U[3][3], S[3], V[3][3];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i / 3][i % 3] = Fs[(size_t)f * 9 + i];
  // A candidate with a NaN entry is no candidate (the empty root slots of the seven-point kernel
  // arrive this way; the reference has no such input): rejected like a gated one.
  bool has_nan = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) has_nan |= (F[i / 3][i % 3] != F[i / 3][i % 3]);
  jacobi_svd3(F, U, S, V);
  const double nan = __builtin_nan("");
  const double ratio = has_nan ? nan : fabs(S[0] - S[1]) / (fabs(S[0] + S[1]) / 2.);
  if (ratio_out) ratio_out[f] = ratio;
  double *out = cams + (size_t)f * 48;
  // src/RansacFitter.h:51-53 (a NaN ratio of finite entries, e.g. F = 0, is not gated there either)
  if (has_nan || ratio > ratio_allowed) {
    gated[f] = 1;
#pragma unroll
    for (int i = 0; i < 48; ++i) out[i] = nan;
    if (E_out)
#pragma unroll
      for (int i = 0; i < 9; ++i) E_out[(size_t)f * 9 + i] = nan;
    return;
  }
  gated[f] = 0;
  live[atomicAdd(nlive, 1)] = f;  // any order: every (camera, correspondence) result stands alone
  double Ud[3][3], Vt[3][3], E[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Ud[i][j] = U[i][j] * (j < 2 ? 1.0 : 0.0);
      Vt[i][j] = V[j][i];
    }
  mat3_mul(Ud, Vt, E);
  if (E_out)
#pragma unroll
    for (int i = 0; i < 9; ++i) E_out[(size_t)f * 9 + i] = E[i / 3][i % 3];
  // Essential2Cameras
  jacobi_svd3(E, U, S, V);
  const double D[3][3] = {{0, 1, 0}, {-1, 0, 0}, {0, 0, 1}};
  const double Dt[3][3] = {{0, -1, 0}, {1, 0, 0}, {0, 0, 1}};
  double UD[3][3], Ra[3][3], Rb[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Vt[i][j] = V[j][i];
  mat3_mul(U, D, UD);
  mat3_mul(UD, Vt, Ra);
  mat3_mul(U, Dt, UD);
  mat3_mul(UD, Vt, Rb);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double sg = (k & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[12 * k + 4 * r + c] = k < 2 ? Ra[r][c] : Rb[r][c];
      out[12 * k + 4 * r + 3] = sg * U[r][2];
    }
  }
}

// The choice among the four cameras, src/RansacFitter.h:74-84: in camera order, a camera becomes the
// best when it reaches required_percent (or find_best_even_in_failure) AND beats the best so far.
__global__ __launch_bounds__(kDltThreads) void select_camera_kernel(
    const int *__restrict__ counts, const int *__restrict__ gated, const double *__restrict__ cams, int nF,
    long long npt, double required_percent, int find_best, int *__restrict__ success,
    int *__restrict__ inlier_count, int *__restrict__ best_cam, double *__restrict__ best_P,
    int *__restrict__ counts4) {
  const int f = blockIdx.x * kDltThreads + threadIdx.x;
  if (f >= nF) return;
  int ok = 0, best = -1, nbest = 0;
  if (!gated[f]) {
    double best_percent = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ninlier = counts[4 * f + k];
      const double percent = ninlier / (double)npt;
      if ((percent >= required_percent || find_best) && percent > best_percent) {
        best_percent = percent;
        nbest = ninlier;
        best = k;
        ok = 1;
      }
    }
  }
  success[f] = ok;
  inlier_count[f] = nbest;
  best_cam[f] = best;
  if (counts4)
#pragma unroll
    for (int k = 0; k < 4; ++k) counts4[4 * f + k] = gated[f] ? -1 : counts[4 * f + k];
  if (best_P)
#pragma unroll
    for (int i = 0; i < 12; ++i) best_P[(size_t)f * 12 + i] = ok ? cams[(size_t)f * 48 + 12 * best + i] : 0.0;
}

// mask[f, p] = inlier flag of candidate f's best camera (all zero when it has none).
__global__ __launch_bounds__(kDltThreads) void best_mask_kernel(const unsigned char *__restrict__ mask4,
                                                                const int *__restrict__ best_cam, long long npt,
                                                                unsigned char *__restrict__ mask) {
  const int f = blockIdx.y;
  const int b = best_cam[f];
  for (long long p = (long long)blockIdx.x * kDltThreads + threadIdx.x; p < npt; p += (long long)gridDim.x * kDltThreads)
    mask[(size_t)f * npt + p] = b >= 0 ? mask4[((size_t)4 * f + b) * npt + p] : 0;
}

}  // namespace

size_t ransac_workspace_bytes(int nF, long long npt, bool want_mask) {
  size_t b = round_up((size_t)nF * 48 * sizeof(double), 256);      // cameras
  b += round_up((size_t)nF * 4 * sizeof(int), 256);                 // inlier counts
  b += round_up((size_t)nF * sizeof(int), 256);                     // gate flags
  b += round_up(((size_t)nF + 1) * sizeof(int), 256);               // the candidates that passed the gate + their count
  if (want_mask) b += round_up((size_t)nF * 4 * (size_t)npt, 256);  // per-camera inlier masks
  b += dlt_score_workspace_bytes(4 * nF, npt);                      // the scorer's work list
  return b;
}

int ransac_process_run(const double *d_Fs, int nF, long long npt, const double *d_x0, const double *d_x1,
                       double ratio_allowed, double required_percent, double max_error, int find_best,
                       int *d_success, int *d_inlier_count, int *d_best_cam, double *d_best_P, double *d_ratio,
                       double *d_E, int *d_counts4, unsigned char *d_mask, void *d_ws, size_t ws_bytes,
                       hipStream_t stream, int score_rows_cap) {
  if (nF < 0 || npt < 0) return set_error(SPV_ERR_INVALID, "negative count");
  if (nF == 0) return SPV_OK;
  if (!d_Fs || !d_success || !d_inlier_count || !d_best_cam || (npt > 0 && (!d_x0 || !d_x1)))
    return set_error(SPV_ERR_INVALID, "null device pointer");
  if (nF > 16383) return set_error(SPV_ERR_INVALID, "more than 16383 candidates per call");
  if (npt == 0) return set_error(SPV_ERR_INVALID, "no correspondences");
  const size_t need = ransac_workspace_bytes(nF, npt, d_mask != nullptr);
  if (!d_ws || ws_bytes < need) return set_error(SPV_ERR_INVALID, "workspace too small: %zu < %zu", ws_bytes, need);
  unsigned char *ws = static_cast<unsigned char *>(d_ws);
  double *cams = reinterpret_cast<double *>(ws);
  ws += round_up((size_t)nF * 48 * sizeof(double), 256);
  int *counts = reinterpret_cast<int *>(ws);
  ws += round_up((size_t)nF * 4 * sizeof(int), 256);
  int *gated = reinterpret_cast<int *>(ws);
  ws += round_up((size_t)nF * sizeof(int), 256);
  int *nlive = reinterpret_cast<int *>(ws);  // [0] = count, [1..] = candidate ids
  int *live = nlive + 1;
  ws += round_up(((size_t)nF + 1) * sizeof(int), 256);
  unsigned char *mask4 = d_mask ? ws : nullptr;
  if (d_mask) ws += round_up((size_t)nF * 4 * (size_t)npt, 256);
  const size_t score_ws = dlt_score_workspace_bytes(4 * nF, npt);
  const int fblocks = (nF + kDltThreads - 1) / kDltThreads;
  SPV_HIP_CHECK(hipMemsetAsync(nlive, 0, sizeof(int), stream));
  {
    ProfScope prof("ransac_cameras", stream);
    hipLaunchKernelGGL(essential_cameras_kernel, dim3(fblocks), dim3(kDltThreads), 0, stream, d_Fs, nF,
                       ratio_allowed, cams, d_ratio, d_E, gated, live, nlive);
    SPV_HIP_CHECK(hipGetLastError());
  }
  const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // Camera(): Identity(3,4), src/Camera.h:27
  // a mask row of a gated camera is never looked at (best_mask_kernel only reads the best camera's)
  SPV_TRY(dlt_score_run(P0, cams, 4 * nF, npt, d_x0, d_x1, max_error, counts, mask4, ws, score_ws, stream, live, nlive,
                        score_rows_cap));
  hipLaunchKernelGGL(select_camera_kernel, dim3(fblocks), dim3(kDltThreads), 0, stream, counts, gated, cams, nF, npt,
                     required_percent, find_best, d_success, d_inlier_count, d_best_cam, d_best_P, d_counts4);
  SPV_HIP_CHECK(hipGetLastError());
  if (d_mask) {
    const unsigned mb = (unsigned)std::min<long long>((npt + kDltThreads - 1) / kDltThreads, 1024);
    hipLaunchKernelGGL(best_mask_kernel, dim3(mb, (unsigned)nF), dim3(kDltThreads), 0, stream, mask4, d_best_cam, npt,
                       d_mask);
    SPV_HIP_CHECK(hipGetLastError());
  }
  return SPV_OK;
}

// Work list of the two-pass scorer: kListShards counters (64 bytes apart) + as many segments of
// deferred (hypothesis, point) pairs, up to 2^22 in all.
static unsigned int score_segment(int nhyp, long long npt) {
  const unsigned long long pairs = (unsigned long long)nhyp * (unsigned long long)npt;
  const unsigned long long cap = std::min<unsigned long long>(pairs, 1ull << 22);
  return (unsigned int)std::max<unsigned long long>((cap + kListShards - 1) / kListShards, 64);
}
constexpr size_t kScoreCountBytes = (size_t)kListShards * kCountStride * sizeof(unsigned int);

size_t dlt_score_workspace_bytes(int nhyp, long long npt) {
  if (nhyp <= 0 || npt <= 0) return 0;
  return kScoreCountBytes + (size_t)kListShards * score_segment(nhyp, npt) * sizeof(unsigned long long);
}

int dlt_score_run(const double *P0, const double *d_p1s, int nhyp, long long npt, const double *d_x,
                  const double *d_xp, double max_error, int *d_counts, unsigned char *d_mask, void *d_ws,
                  size_t ws_bytes, hipStream_t stream, const int *d_live, const int *d_nlive, int rows_cap) {
  if (npt < 0 || nhyp < 0) return set_error(SPV_ERR_INVALID, "negative count");
  if (!P0) return set_error(SPV_ERR_INVALID, "null camera pointer");
  if (nhyp == 0) return SPV_OK;
  if (!d_counts || !d_p1s) return set_error(SPV_ERR_INVALID, "null device pointer");
  SPV_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)nhyp * sizeof(int), stream));
  if (npt == 0) return SPV_OK;
  if (!d_x || !d_xp) return set_error(SPV_ERR_INVALID, "null device pointer");
  if (nhyp > 65535) return set_error(SPV_ERR_INVALID, "more than 65535 hypotheses per call");
  if (npt >= (1ll << 40)) return set_error(SPV_ERR_INVALID, "too many points");
  Cameras cam0;
  for (int i = 0; i < 12; ++i) cam0.p0[i] = cam0.p1[i] = P0[i];
  const long long blocks = (npt + kDltThreads - 1) / kDltThreads;
  if (blocks > 0x7FFFFFFFLL) return set_error(SPV_ERR_INVALID, "too many points");
  ProfScope prof("dlt_score", stream);
  const dim3 grid((unsigned)blocks, (unsigned)(d_live ? std::min(nhyp, std::max(rows_cap, 4)) : nhyp));
  // a workspace too small for the work list is not an error: the scorer then runs in one pass
  WorkList wl{nullptr, nullptr, 0};
  if (d_ws && ws_bytes > kScoreCountBytes && (reinterpret_cast<uintptr_t>(d_ws) & 7) == 0) {
    wl.count = static_cast<unsigned int *>(d_ws);
    wl.entries = reinterpret_cast<unsigned long long *>(static_cast<char *>(d_ws) + kScoreCountBytes);
    wl.segment = (unsigned int)std::min<size_t>((ws_bytes - kScoreCountBytes) / sizeof(unsigned long long) / kListShards,
                                                score_segment(nhyp, npt));
  }
  if (wl.segment == 0) {
    hipLaunchKernelGGL(dlt_score_kernel<false>, grid, dim3(kDltThreads), 0, stream, cam0, d_p1s, nhyp, npt, d_x, d_xp,
                       max_error, d_counts, d_mask, wl, d_live, d_nlive);
    SPV_HIP_CHECK(hipGetLastError());
    return SPV_OK;
  }
  SPV_HIP_CHECK(hipMemsetAsync(wl.count, 0, kScoreCountBytes, stream));
  hipLaunchKernelGGL(dlt_score_kernel<true>, grid, dim3(kDltThreads), 0, stream, cam0, d_p1s, nhyp, npt, d_x, d_xp,
                     max_error, d_counts, d_mask, wl, d_live, d_nlive);
  SPV_HIP_CHECK(hipGetLastError());
  // the list's length is only known on the device: a grid that covers a full list, strided
  const unsigned fb = (unsigned)std::min<unsigned long long>(((unsigned long long)kListShards * wl.segment + kDltThreads - 1) / kDltThreads,
                                                             (unsigned long long)device_cu_count() * 16);
  hipLaunchKernelGGL(dlt_score_fallback_kernel, dim3(std::max(fb, 1u)), dim3(kDltThreads), 0, stream, cam0, d_p1s, npt,
                     d_x, d_xp, max_error, d_counts, d_mask, wl);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

int dlt_run(const double *P0, const double *P1, long long npt, const double *d_x,
            const double *d_xp, double *d_dst, bool want_error, hipStream_t stream) {
  if (npt < 0) return set_error(SPV_ERR_INVALID, "negative point count");
  if (!P0 || !P1) return set_error(SPV_ERR_INVALID, "null camera pointer");
  if (npt == 0) return SPV_OK;
  if (!d_x || !d_xp || !d_dst) return set_error(SPV_ERR_INVALID, "null device pointer");
  Cameras cam;
  for (int i = 0; i < 12; ++i) {
    cam.p0[i] = P0[i];
    cam.p1[i] = P1[i];
  }
  // persistent grid: at most 32 workgroups per CU, each lane strides over its points
  const int cus = device_cu_count();
  const long long blocks = std::min<long long>((npt + kDltThreads - 1) / kDltThreads, (long long)std::max(cus, 1) * 32);
  ProfScope prof("dlt", stream);
  if (want_error)
    hipLaunchKernelGGL((dlt_kernel<true>), dim3((unsigned)blocks), dim3(kDltThreads), 0, stream,
                       cam, npt, d_x, d_xp, d_dst);
  else
    hipLaunchKernelGGL((dlt_kernel<false>), dim3((unsigned)blocks), dim3(kDltThreads), 0, stream,
                       cam, npt, d_x, d_xp, d_dst);
  SPV_HIP_CHECK(hipGetLastError());
  return SPV_OK;
}

}  // namespace spv
