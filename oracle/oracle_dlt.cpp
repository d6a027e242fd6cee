// oracle_dlt.cpp -- CPU restatement of the reference's two-view DLT triangulation.
//
// TEST INFRASTRUCTURE ONLY (see oracle_l1k2.cpp header): never imported, linked
// or called from spectavi_amd/.
//
// Follows (read as text, restated, not copied):
//   reference src/DltTriangulator.h:36-65  solve: hnormalize both observations
//       (:38-45), A rows u*P[2]-P[0], v*P[2]-P[1] per view (:51-54), X = right
//       singular vector of the smallest singular value (:56-58), reprojections (:61-62)
//   reference src/DltTriangulator.h:67-74  reprojection_error: sum of the two
//       Euclidean pixel distances
//   reference src/DltTriangulator.h:76-86  cheirality (distance2camera*, not exported
//       through the C-ABI; provided for the RANSAC-scoring "next" row)
//   reference src/Spectavi.cpp:38-68        the serial per-point loops
//
// The SVD itself lives in Eigen (JacobiSVD, version unpinned by the reference
// build, absent here).  Only its result matters -- the unit right singular vector of the
// smallest singular value -- and it is restated two ways (column-pivoted QR + inverse
// iteration with a one-sided Hestenes Jacobi fallback; Jacobi alone for RANSAC scoring)
// in fp64 -- the same operation sequence the HIP kernel executes, fused
// multiply-adds written explicitly and implicit contraction disabled on both sides -- and pinned two ways in
// tests/test_oracle.py: (a) the reference's own test properties
// (test/test_mvg.py:94-125: reprojection error < 1e-3 and X == X0 up to scale on
// noise-free random cameras), (b) numpy.linalg.svd (LAPACK) null vectors of the
// same A.  PARITY UNPINNED: the sign of X (arbitrary in Eigen); canonicalised
// here and in the HIP path to X[3] >= 0.

#include <cmath>
#include <cstdint>
#include <utility>

namespace {

constexpr int kMaxSweeps = 30;

struct Solve {
  double X[4];
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

// ---- null vector, method 2: column-pivoted Gram-Schmidt QR + inverse iteration -----------
// A P = Q R; the smallest right singular vector of A is that of R (in pivoted order), found by
// inverse iteration on R^T R (two triangular solves per step, contraction (sigma4/sigma3)^2).
// 2 to 8 steps; returns false when the last step still moved the vector by more than
// 1e-12 (ill-separated sigma3, sigma4) -- the caller then falls back to method 1.
inline bool null_qr_inverse_iteration(const double (&A0)[4][4], double (&xv)[4]) {
  double col[4][4];  // col[c][r]
  int perm[4] = {0, 1, 2, 3};
  double R[4][4] = {{0}};
  double ri[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) col[c][r] = A0[r][c];
  double tiny = 0.0;
  for (int j = 0; j < 4; ++j) {
    double nn[4] = {0, 0, 0, 0};
    for (int k = j; k < 4; ++k)
      for (int r = 0; r < 4; ++r) nn[k] = std::fma(col[k][r], col[k][r], nn[k]);
    int best = j;
    for (int k = j + 1; k < 4; ++k)
      if (nn[k] > nn[best]) best = k;
    if (best != j) {
      for (int r = 0; r < 4; ++r) std::swap(col[j][r], col[best][r]);
      std::swap(perm[j], perm[best]);
      std::swap(nn[j], nn[best]);
      for (int r = 0; r < j; ++r) std::swap(R[r][j], R[r][best]);
    }
    double rjj = std::sqrt(nn[j]);
    if (j == 0) tiny = 2.220446049250313e-16 * rjj;
    if (!(rjj > tiny)) rjj = tiny;
    R[j][j] = rjj;
    ri[j] = 1.0 / rjj;
    if (j < 3) {
      double q[4];
      for (int r = 0; r < 4; ++r) q[r] = col[j][r] * ri[j];
      for (int k = j + 1; k < 4; ++k) {
        double rjk = 0.0;
        for (int r = 0; r < 4; ++r) rjk = std::fma(q[r], col[k][r], rjk);
        R[j][k] = rjk;
        for (int r = 0; r < 4; ++r) col[k][r] = std::fma(-rjk, q[r], col[k][r]);
      }
    }
  }
  double v[4] = {0.0, 0.0, 0.0, 1.0};
  double delta = 1.0;
  // at least 2, at most 8 steps; a point stops at the first step that moved its vector by no
  // more than 1e-12 (noise-free points after 2, pixel noise 1e-3 after 3)
  {
    for (int it = 0; it < 8; ++it) {
      // R^T z = v
      const double z0 = v[0] * ri[0];
      const double z1 = std::fma(-R[0][1], z0, v[1]) * ri[1];
      const double z2 = std::fma(-R[1][2], z1, std::fma(-R[0][2], z0, v[2])) * ri[2];
      const double z3 = std::fma(-R[2][3], z2, std::fma(-R[1][3], z1, std::fma(-R[0][3], z0, v[3]))) * ri[3];
      // R w = z
      const double w3 = z3 * ri[3];
      const double w2 = std::fma(-R[2][3], w3, z2) * ri[2];
      const double w1 = std::fma(-R[1][3], w3, std::fma(-R[1][2], w2, z1)) * ri[1];
      const double w0 = std::fma(-R[0][3], w3, std::fma(-R[0][2], w2, std::fma(-R[0][1], w1, z0))) * ri[0];
      const double nrm = std::sqrt(std::fma(w3, w3, std::fma(w2, w2, std::fma(w1, w1, w0 * w0))));
      const double inv = 1.0 / nrm;
      const double n0 = w0 * inv, n1 = w1 * inv, n2 = w2 * inv, n3 = w3 * inv;
      delta = std::fmax(std::fmax(std::fabs(n0 - v[0]), std::fabs(n1 - v[1])),
                        std::fmax(std::fabs(n2 - v[2]), std::fabs(n3 - v[3])));
      v[0] = n0;
      v[1] = n1;
      v[2] = n2;
      v[3] = n3;
      if (it >= 1 && delta <= 1e-12) break;
    }
  }
  if (!(delta <= 1e-12)) return false;
  for (int c = 0; c < 4; ++c) xv[perm[c]] = v[c];
  return true;
}

// fast = true: method 2 with method 1 as fallback (dlt_triangulate / dlt_reprojection_error);
// fast = false: method 1 only (RANSAC scoring, where most hypotheses are inconsistent and
// method 2 would rarely converge).
inline void dlt_solve(const double *P0, const double *P1, const double *x, const double *xp,
                      Solve &out, bool fast = true) {
  const double u = x[0] / x[2], v = x[1] / x[2];
  const double up = xp[0] / xp[2], vp = xp[1] / xp[2];
  double A[4][4];
  for (int c = 0; c < 4; ++c) {
    A[0][c] = std::fma(u, P0[8 + c], -P0[0 + c]);
    A[1][c] = std::fma(v, P0[8 + c], -P0[4 + c]);
    A[2][c] = std::fma(up, P1[8 + c], -P1[0 + c]);
    A[3][c] = std::fma(vp, P1[8 + c], -P1[4 + c]);
  }
  double xv[4];
  if (!(fast && null_qr_inverse_iteration(A, xv))) null_jacobi(A, xv);
  double nrm = 0.0;
  for (int i = 0; i < 4; ++i) nrm = std::fma(xv[i], xv[i], nrm);
  nrm = std::sqrt(nrm);
  bool neg;
  if (xv[3] != 0.0)
    neg = xv[3] < 0.0;
  else if (xv[0] != 0.0)
    neg = xv[0] < 0.0;
  else if (xv[1] != 0.0)
    neg = xv[1] < 0.0;
  else
    neg = xv[2] < 0.0;
  const double scale = neg ? -nrm : nrm;
  for (int i = 0; i < 4; ++i) out.X[i] = xv[i] / scale;
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

inline double det3(const double *P) {
  // determinant of the left 3x3 block of a row-major 3x4
  return P[0] * (P[5] * P[10] - P[6] * P[9]) - P[1] * (P[4] * P[10] - P[6] * P[8]) +
         P[2] * (P[4] * P[9] - P[5] * P[8]);
}

}  // namespace

extern "C" {

// dst: double[npt,4].  Serial loop, as reference src/Spectavi.cpp:48-51.
void oracle_dlt_triangulate(const double *P0, const double *P1, int npt, const double *x,
                            const double *xp, double *dst) {
  for (int i = 0; i < npt; ++i) {
    Solve s;
    dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
    for (int k = 0; k < 4; ++k) dst[4 * (size_t)i + k] = s.X[k];
  }
}

// dst: double[npt].  As reference src/Spectavi.cpp:64-67.
void oracle_dlt_reprojection_error(const double *P0, const double *P1, int npt, const double *x,
                                   const double *xp, double *dst) {
  for (int i = 0; i < npt; ++i) {
    Solve s;
    dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
    double r0[3], r1[3];
    reproject(P0, s.X, r0);
    reproject(P1, s.X, r1);
    const double e0x = r0[0] / r0[2] - s.u, e0y = r0[1] / r0[2] - s.v;
    const double e1x = r1[0] / r1[2] - s.up, e1y = r1[1] / r1[2] - s.vp;
    dst[i] = std::sqrt(std::fma(e0x, e0x, e0y * e0y)) + std::sqrt(std::fma(e1x, e1x, e1y * e1y));
  }
}

// infront: uint8[npt], 1 iff the point is in front of both cameras
// (reference src/DltTriangulator.h:76-86).
void oracle_dlt_cheirality(const double *P0, const double *P1, int npt, const double *x,
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
void oracle_dlt_score_hypotheses(const double *P0, const double *P1s, int nhyp, int npt,
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
      dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s, /*fast=*/false);
      double r0[3], r1[3];
      reproject(P0, s.X, r0);
      reproject(P1, s.X, r1);
      const double e0x = r0[0] / r0[2] - s.u, e0y = r0[1] / r0[2] - s.v;
      const double e1x = r1[0] / r1[2] - s.up, e1y = r1[1] / r1[2] - s.vp;
      const double err = std::sqrt(std::fma(e0x, e0x, e0y * e0y)) + std::sqrt(std::fma(e1x, e1x, e1y * e1y));
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
