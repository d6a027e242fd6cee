// oracle_dlt_mirror.cpp -- host-side MIRROR of the HIP DLT kernels' own operation sequence.
//
// TEST INFRASTRUCTURE ONLY (see oracle_l1k2.cpp header): never imported, linked
// or called from spectavi_amd/.
//
// THIS IS NOT THE ORACLE for the DLT rows: it executes, with std::fma, the sequence of operations
// spectavi_amd/csrc/dlt.hip executes (square-root-free Gram-Schmidt + division-free inverse
// iteration with a one-sided Hestenes Jacobi fallback; sign canonicalised to X[3] >= 0), with the
// exact 1/x and 1/sqrt(x) where the kernel uses v_rcp_f64 / v_rsq_f64 + two Newton steps (< 1 ulp
// apart).  "device == host to a few ulps times the point's conditioning, inf/nan in the same
// places" pins determinism of the device code (no miscompiled FMA contraction, no lane- or
// shape-dependent path), NOT that X is the right singular vector of the smallest singular value.
// (Until round 3 the kernel used IEEE divisions and the agreement was bit for bit; they were a
// third of its issue time.)  Correctness of the HIP path is judged against oracle_jacobisvd.cpp
// (the restatement of the reference's Eigen::JacobiSVD arithmetic, src/DltTriangulator.h:36-86)
// and against LAPACK in tests/test_dlt_gpu.py, tests/fuzz_gpu.py and tests/test_oracle.py.

#include <cmath>
#include <cstdint>
#include <utility>

namespace {

constexpr int kMaxSweeps = 30;

struct Solve {
  double X[4];   // unit norm, canonical sign
  double S[4];   // the same direction at max-norm [0.5, 1), sign as it fell: what the error kernel uses
  double u, v, up, vp;
};

// ---- null vector, method 1: one-sided (Hestenes) Jacobi, always converges ----------------
inline void null_jacobi(const double (&A0)[4][4], double (&xv)[4]) {
  double A[4][4], V[4][4];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) A[r][c] = A0[r][c];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;

  const double eps2 = 1e-30;  // (1e-15)^2
  for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 3; ++p) {
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int i = 0; i < 4; ++i) {
          alpha = std::fma(A[i][p], A[i][p], alpha);
          beta = std::fma(A[i][q], A[i][q], beta);
          gamma = std::fma(A[i][p], A[i][q], gamma);
        }
        // skip test |gamma| > eps*sqrt(alpha*beta), squared to avoid the square root
        if (gamma * gamma > eps2 * (alpha * beta)) {
          rotated = true;
          // tan of the rotation angle, t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with
          // zeta = (beta - alpha) / (2 gamma), rearranged to one sqrt and one division
          const double dd = beta - alpha;
          const double g2 = 2.0 * gamma;
          const double hh = std::sqrt(std::fma(dd, dd, g2 * g2));
          const double tn = g2 / (dd + (dd < 0.0 ? -hh : hh));
          const double cs = 1.0 / std::sqrt(std::fma(tn, tn, 1.0));
          const double sn = cs * tn;
          for (int i = 0; i < 4; ++i) {
            const double ap = A[i][p], aq = A[i][q];
            A[i][p] = std::fma(cs, ap, -(sn * aq));
            A[i][q] = std::fma(sn, ap, cs * aq);
            const double vp_ = V[i][p], vq_ = V[i][q];
            V[i][p] = std::fma(cs, vp_, -(sn * vq_));
            V[i][q] = std::fma(sn, vp_, cs * vq_);
          }
        }
      }
    }
    if (!rotated) break;
  }
  double best = 0.0;
  int kbest = 0;
  for (int c = 0; c < 4; ++c) {
    double nn = 0.0;
    for (int i = 0; i < 4; ++i) nn = std::fma(A[i][c], A[i][c], nn);
    if (c == 0 || nn < best) {
      best = nn;
      kbest = c;
    }
  }
  for (int i = 0; i < 4; ++i) xv[i] = V[i][kbest];
}

// ---- null vector, method 2: square-root-free Gram-Schmidt + inverse iteration ------------
// A = Q U with orthogonal (not normalised) columns q_j, d_j = |q_j|^2, U unit upper triangular
// (modified Gram-Schmidt, no pivoting), so A^T A = U^T D U; the smallest right singular vector
// of A by inverse iteration on U^T D U, the diagonal solve scaled by d_3 (y_j = z_j d_3 / d_j,
// y_3 = z_3), iterates left un-normalised, convergence = direction of consecutive iterates equal
// to 1e-13 (same schedule as the HIP kernel: steps 0, 1, 2, then tested steps 3..7).  Returns false
// when the last step still moved the direction or the iterate left the range -- the caller then
// falls back to method 1.
inline bool null_gs_inverse_iteration(const double (&A0)[4][4], double (&xv)[4]) {
  double col[4][4];  // col[c][r]
  double U[4][4] = {{0}};
  double g[3] = {0, 0, 0};
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) col[c][r] = A0[r][c];
  double tiny2 = 0.0;
  for (int j = 0; j < 4; ++j) {
    double d = 0.0;
    for (int r = 0; r < 4; ++r) d = std::fma(col[j][r], col[j][r], d);
    if (j == 0) tiny2 = 4.930380657631324e-32 * d;  // eps^2 |a_0|^2
    if (!(d > tiny2)) d = tiny2;
    if (j == 3) {
      for (int k = 0; k < 3; ++k) g[k] *= d;
      break;
    }
    const double idj = 1.0 / d;
    g[j] = idj;
    for (int k = j + 1; k < 4; ++k) {
      double s = 0.0;
      for (int r = 0; r < 4; ++r) s = std::fma(col[j][r], col[k][r], s);
      const double u = s * idj;
      U[j][k] = u;
      for (int r = 0; r < 4; ++r) col[k][r] = std::fma(-u, col[j][r], col[k][r]);
    }
  }
  double v[4];
  v[3] = 1.0;
  v[2] = -U[2][3];
  v[1] = std::fma(-U[1][2], v[2], -U[1][3]);
  v[0] = std::fma(-U[0][1], v[1], std::fma(-U[0][2], v[2], -U[0][3]));
  double w[4];
  auto step = [&]() {  // w = d_3 (U^T D U)^-1 v
    const double z0 = v[0];
    const double z1 = std::fma(-U[0][1], z0, v[1]);
    const double z2 = std::fma(-U[1][2], z1, std::fma(-U[0][2], z0, v[2]));
    const double z3 = std::fma(-U[2][3], z2, std::fma(-U[1][3], z1, std::fma(-U[0][3], z0, v[3])));
    const double y0 = z0 * g[0], y1 = z1 * g[1], y2 = z2 * g[2];
    w[3] = z3;
    w[2] = std::fma(-U[2][3], w[3], y2);
    w[1] = std::fma(-U[1][3], w[3], std::fma(-U[1][2], w[2], y1));
    w[0] = std::fma(-U[0][3], w[3], std::fma(-U[0][2], w[2], std::fma(-U[0][1], w[1], y0)));
  };
  auto maxabs = [](const double (&a)[4]) {
    return std::fmax(std::fmax(std::fabs(a[0]), std::fabs(a[1])), std::fmax(std::fabs(a[2]), std::fabs(a[3])));
  };
  for (int it = 1; it <= 2; ++it) {
    step();
    for (int c = 0; c < 4; ++c) v[c] = w[c];
  }
  double bv = maxabs(v);
  bool ok = false;
  for (int it = 3; it < 8; ++it) {
    step();
    const double bw = maxabs(w);
    double e[4];
    for (int c = 0; c < 4; ++c) e[c] = std::fabs(std::fma(w[c], bv, -(v[c] * bw)));
    const double bound = 1e-13 * (bw * bv);
    ok = (maxabs(e) <= bound) && (bound >= 1e-290) && (bound <= 1e290);
    for (int c = 0; c < 4; ++c) v[c] = w[c];
    bv = bw;
    if (ok) break;
  }
  if (!ok) return false;
  for (int c = 0; c < 4; ++c) xv[c] = v[c];
  return true;
}

// fast = true: method 2 with method 1 as fallback (dlt_triangulate / dlt_reprojection_error and,
// since round 2, RANSAC scoring: the kernel defers its slow lanes to a second pass, the result per
// (hypothesis, point) pair is this function's); fast = false: method 1 only.
inline void dlt_solve(const double *P0, const double *P1, const double *x, const double *xp,
                      Solve &out, bool fast = true) {
  const double ix = 1.0 / x[2], iy = 1.0 / xp[2];  // one reciprocal per view, as the kernel
  const double u = x[0] * ix, v = x[1] * ix;
  const double up = xp[0] * iy, vp = xp[1] * iy;
  double A[4][4];
  for (int c = 0; c < 4; ++c) {
    A[0][c] = std::fma(u, P0[8 + c], -P0[0 + c]);
    A[1][c] = std::fma(v, P0[8 + c], -P0[4 + c]);
    A[2][c] = std::fma(up, P1[8 + c], -P1[0 + c]);
    A[3][c] = std::fma(vp, P1[8 + c], -P1[4 + c]);
  }
  double xv[4];
  if (!(fast && null_gs_inverse_iteration(A, xv))) null_jacobi(A, xv);
  // max-norm into [0.5, 1) by an exact power of two, then the unit 2-norm and the canonical sign
  const double big = std::fmax(std::fmax(std::fabs(xv[0]), std::fabs(xv[1])), std::fmax(std::fabs(xv[2]), std::fabs(xv[3])));
  int ex = 0;
  if (std::isfinite(big) && big != 0.0) std::frexp(big, &ex);
  double sv[4];
  for (int i = 0; i < 4; ++i) sv[i] = std::ldexp(xv[i], -ex);
  double nrm2 = 0.0;
  for (int i = 0; i < 4; ++i) nrm2 = std::fma(sv[i], sv[i], nrm2);
  bool neg;
  if (xv[3] != 0.0)
    neg = xv[3] < 0.0;
  else if (xv[0] != 0.0)
    neg = xv[0] < 0.0;
  else if (xv[1] != 0.0)
    neg = xv[1] < 0.0;
  else
    neg = xv[2] < 0.0;
  const double q = 1.0 / std::sqrt(nrm2);
  const double scale = neg ? -q : q;
  for (int i = 0; i < 4; ++i) out.X[i] = sv[i] * scale;
  for (int i = 0; i < 4; ++i) out.S[i] = sv[i];
  out.u = u;
  out.v = v;
  out.up = up;
  out.vp = vp;
}

inline void reproject(const double *P, const double *X, double *r) {
  for (int k = 0; k < 3; ++k) {
    double a = 0.0;
    for (int c = 0; c < 4; ++c) a = std::fma(P[4 * k + c], X[c], a);
    r[k] = a;
  }
}

// sum of the two image-plane residual norms (reference src/DltTriangulator.h:67-74), one reciprocal
// per camera as the kernel; r0 / r1 = P X
inline double reprojection_error(const double *P0, const double *P1, const Solve &s, double *r0, double *r1,
                                 bool unit = true) {
  reproject(P0, unit ? s.X : s.S, r0);
  reproject(P1, unit ? s.X : s.S, r1);
  const double i0 = 1.0 / r0[2], i1 = 1.0 / r1[2];
  const double e0x = std::fma(r0[0], i0, -s.u), e0y = std::fma(r0[1], i0, -s.v);
  const double e1x = std::fma(r1[0], i1, -s.up), e1y = std::fma(r1[1], i1, -s.vp);
  return std::sqrt(std::fma(e0x, e0x, e0y * e0y)) + std::sqrt(std::fma(e1x, e1x, e1y * e1y));
}

inline double det3(const double *P) {
  // determinant of the left 3x3 block of a row-major 3x4
  return P[0] * (P[5] * P[10] - P[6] * P[9]) - P[1] * (P[4] * P[10] - P[6] * P[8]) +
         P[2] * (P[4] * P[9] - P[5] * P[8]);
}

}  // namespace

extern "C" {

// dst: double[npt,4].  Serial loop, as reference src/Spectavi.cpp:48-51.
void oracle_dlt_mirror_triangulate(const double *P0, const double *P1, int npt, const double *x,
                            const double *xp, double *dst) {
  for (int i = 0; i < npt; ++i) {
    Solve s;
    dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
    for (int k = 0; k < 4; ++k) dst[4 * (size_t)i + k] = s.X[k];
  }
}

// dst: double[npt].  As reference src/Spectavi.cpp:64-67.
void oracle_dlt_mirror_reprojection_error(const double *P0, const double *P1, int npt, const double *x,
                                   const double *xp, double *dst) {
  for (int i = 0; i < npt; ++i) {
    Solve s;
    dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
    double r0[3], r1[3];
    dst[i] = reprojection_error(P0, P1, s, r0, r1, /*unit=*/false);  // as dlt_kernel<true>: no normalisation
  }
}

// infront: uint8[npt], 1 iff the point is in front of both cameras
// (reference src/DltTriangulator.h:76-86).
void oracle_dlt_mirror_cheirality(const double *P0, const double *P1, int npt, const double *x,
                           const double *xp, uint8_t *infront) {
  const double s0 = det3(P0) < 0 ? -1.0 : 1.0, s1 = det3(P1) < 0 ? -1.0 : 1.0;
  const double n0 = P0[2] * P0[2] + P0[6] * P0[6] + P0[10] * P0[10];
  const double n1 = P1[2] * P1[2] + P1[6] * P1[6] + P1[10] * P1[10];
  for (int i = 0; i < npt; ++i) {
    Solve s;
    dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
    double r0[3], r1[3];
    reproject(P0, s.X, r0);
    reproject(P1, s.X, r1);
    const double dc0 = s0 / n0 * r0[2] / s.X[3];
    const double dc1 = s1 / n1 * r1[2] / s.X[3];
    infront[i] = (dc0 > 0 && dc1 > 0) ? 1 : 0;
  }
}

// RANSAC hypothesis scoring: counts int32[nhyp], mask uint8[nhyp,npt] (may be NULL).
// Restates the two scoring loops of reference src/RansacFitter.h:59-73 and :86-94
// (inlier iff reprojection_error() <= max_error && is_infront_both_cameras()).
void oracle_dlt_mirror_score_hypotheses(const double *P0, const double *P1s, int nhyp, int npt,
                                 const double *x, const double *xp, double max_error,
                                 int32_t *counts, uint8_t *mask) {
  for (int h = 0; h < nhyp; ++h) {
    const double *P1 = P1s + 12 * (size_t)h;
    const double s0 = det3(P0) < 0 ? -1.0 : 1.0, s1 = det3(P1) < 0 ? -1.0 : 1.0;
    const double n0 = P0[2] * P0[2] + P0[6] * P0[6] + P0[10] * P0[10];
    const double n1 = P1[2] * P1[2] + P1[6] * P1[6] + P1[10] * P1[10];
    int cnt = 0;
    for (int i = 0; i < npt; ++i) {
      Solve s;
      dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s, /*fast=*/true);
      double r0[3], r1[3];
      const double err = reprojection_error(P0, P1, s, r0, r1);
      const double dc0 = s0 / n0 * r0[2] / s.X[3];
      const double dc1 = s1 / n1 * r1[2] / s.X[3];
      const bool in = (err <= max_error) && (dc0 > 0) && (dc1 > 0);
      cnt += in ? 1 : 0;
      if (mask) mask[(size_t)h * npt + i] = in ? 1 : 0;
    }
    counts[h] = cnt;
  }
}

}  // extern "C"
