// oracle_jacobisvd.cpp -- CPU restatement of the reference's DLT / essential-matrix arithmetic
// INCLUDING the third-party SVD it calls.
//
// TEST INFRASTRUCTURE ONLY (see oracle_l1k2.cpp header): never imported, linked or called from
// spectavi_amd/.  This is the oracle for SURVEY rows A8, A9 and (f)1; oracle_dlt_mirror.cpp is
// only a host-side mirror of the HIP kernel's own operation sequence (bit-reproducibility check).
//
// The reference computes (read as text, restated, not copied):
//   src/DltTriangulator.h:36-65   solve: hnormalize (:38-45), A (:51-54),
//                                 Eigen::JacobiSVD<MatrixType> svd(A, ComputeFullV) (:56),
//                                 X = V.col(3) (:57-58), reprojections (:61-62)
//   src/DltTriangulator.h:67-86   reprojection_error, distance2camera0/1, is_infront_both_cameras
//   src/Camera.h:31-46            Essential2Cameras: JacobiSVD(E, FullU|FullV); t = U.col(2);
//                                 Ra = U D V^T, Rb = U D^T V^T; cameras (Ra,t) (Ra,-t) (Rb,t) (Rb,-t)
//   src/RansacFitter.h:42-95      process_fundamental_matrix: JacobiSVD(F), singular-value-ratio
//                                 gate (:49-53), E = U diag(1,1,0) V^T (:54-56), score the four
//                                 cameras over all points (:59-73), keep the best (:74-94)
//   src/Spectavi.cpp:38-68        the serial per-point loops
//
// Third-party dependency absent from /root/reference: Eigen3, `find_package(Eigen3 REQUIRED)` with
// NO version pin (reference CMakeLists.txt:20; CI installs the distribution's libeigen3-dev,
// .travis.yml:4).  Its JacobiSVD is restated here from the published algorithm of Eigen 3.3.x / 3.4.x
// (Eigen/src/SVD/JacobiSVD.h `compute`, `real_2x2_jacobi_svd`; Eigen/src/Jacobi/Jacobi.h
// `makeJacobi`, `apply_rotation_in_the_plane`), for a square real matrix, which needs no QR
// preconditioner:
//   1. scale = max |a_ij| (1 if zero); W = A / scale; U = V = I.
//   2. sweeps over p = 1..n-1, q = 0..p-1 until a whole sweep finds nothing to do:
//        threshold = max(DBL_MIN, 2 eps * maxDiagEntry)
//        if |W(p,q)| > threshold or |W(q,p)| > threshold:
//            2x2 real Jacobi SVD of [[W(p,p) W(p,q)] [W(q,p) W(q,q)]]: a rotation rot1 that makes
//            the block symmetric (t = w00 + w11, d = w10 - w01, u = t/d, s = 1/sqrt(1+u^2),
//            c = u/sqrt(1+u^2)), then the symmetric Jacobi rotation j_right of the rotated block
//            (makeJacobi: tau = (x - z) / (2|y|), w = sqrt(tau^2+1), t = 1/(tau +- w),
//            n = 1/sqrt(t^2+1), s = -sign(t) (y/|y|) |t| n, c = n); j_left = rot1 * j_right^T.
//            W <- j_left applied to rows p,q; U <- U j_left^T (columns p,q);
//            W <- W j_right (columns p,q);    V <- V j_right (columns p,q);
//            maxDiagEntry = max(maxDiagEntry, |W(p,p)|, |W(q,q)|).
//   3. singular values = |diag W| * scale; columns of U with a negative diagonal entry are negated.
//   4. selection sort by descending singular value, swapping the columns of U and V alongside.
// Rotations are applied as Eigen applies them: x' = c x + s y, y' = -s x + c y, products and sums
// rounded separately (the reference is built without -march, so no FMA contraction: reference
// CMakeLists.txt:12,55-57); this file is compiled with -ffp-contract=off and without -mfma.
//
// PINNING (no reference-held vector exists for this path): tests/test_oracle.py checks
// this file against numpy.linalg.svd (LAPACK) -- singular values to 1e-13 relative, U S V^T = A,
// orthogonality, V.col(3) up to sign with a conditioning-scaled tolerance -- and against the
// reference's own test properties (test/test_mvg.py:94-125).  PARITY UNPINNED: which Eigen version
// the reference was built with, hence the sign of V.col(3) and of the columns of U, V of a 3x3 SVD.
// The results here carry the signs THIS restatement produces; every comparison with the HIP path
// is made up to those sign freedoms and says so.

#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <utility>

namespace {

struct Rot {  // J = [c s; -s c]
  double c, s;
};

// rows p,q of an n x n row-major matrix: x' = c x + s y, y' = -s x + c y  (B = J B)
template <int N>
inline void apply_left(double (&M)[N][N], int p, int q, Rot j) {
  for (int i = 0; i < N; ++i) {
    const double xi = M[p][i], yi = M[q][i];
    M[p][i] = j.c * xi + j.s * yi;
    M[q][i] = -j.s * xi + j.c * yi;
  }
}

// columns p,q: B = B J, i.e. the rotation (c, -s) applied to the pair (col p, col q)
template <int N>
inline void apply_right(double (&M)[N][N], int p, int q, Rot j) {
  const double c = j.c, s = -j.s;
  for (int i = 0; i < N; ++i) {
    const double xi = M[i][p], yi = M[i][q];
    M[i][p] = c * xi + s * yi;
    M[i][q] = -s * xi + c * yi;
  }
}

inline Rot transpose(Rot j) { return Rot{j.c, -j.s}; }
inline Rot compose(Rot a, Rot b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }

inline Rot make_jacobi(double x, double y, double z) {
  const double deno = 2.0 * std::fabs(y);
  if (deno < DBL_MIN) return Rot{1.0, 0.0};
  const double tau = (x - z) / deno;
  const double w = std::sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / std::sqrt(t * t + 1.0);
  return Rot{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}

template <int N>
inline void real_2x2_jacobi_svd(const double (&W)[N][N], int p, int q, Rot &j_left, Rot &j_right) {
  double m[2][2] = {{W[p][p], W[p][q]}, {W[q][p], W[q][q]}};
  Rot rot1;
  const double t = m[0][0] + m[1][1];
  const double d = m[1][0] - m[0][1];
  if (std::fabs(d) < DBL_MIN) {
    rot1 = Rot{1.0, 0.0};
  } else {
    const double u = t / d;
    const double tmp = std::sqrt(1.0 + u * u);
    rot1 = Rot{u / tmp, 1.0 / tmp};
  }
  apply_left<2>(m, 0, 1, rot1);
  j_right = make_jacobi(m[0][0], m[0][1], m[1][1]);
  j_left = compose(rot1, transpose(j_right));
}

// A (row-major n x n) = U diag(S) V^T, S descending.  Returns the number of sweeps.
template <int N>
int jacobi_svd(const double (&A)[N][N], double (&U)[N][N], double (&S)[N], double (&V)[N][N]) {
  const double precision = 2.0 * DBL_EPSILON;
  const double consider_as_zero = DBL_MIN;
  double scale = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) scale = std::fmax(scale, std::fabs(A[i][j]));
  if (scale == 0.0) scale = 1.0;
  double W[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      W[i][j] = A[i][j] / scale;
      U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  double max_diag = 0.0;
  for (int i = 0; i < N; ++i) max_diag = std::fmax(max_diag, std::fabs(W[i][i]));
  int sweeps = 0;
  bool finished = false;
  while (!finished) {
    finished = true;
    ++sweeps;
    for (int p = 1; p < N; ++p) {
      for (int q = 0; q < p; ++q) {
        const double threshold = std::fmax(consider_as_zero, precision * max_diag);
        // false whenever a NaN is involved, so NaNs cannot keep the sweeps going
        if (std::fabs(W[p][q]) > threshold || std::fabs(W[q][p]) > threshold) {
          finished = false;
          Rot j_left, j_right;
          real_2x2_jacobi_svd<N>(W, p, q, j_left, j_right);
          apply_left<N>(W, p, q, j_left);
          apply_right<N>(U, p, q, transpose(j_left));
          apply_right<N>(W, p, q, j_right);
          apply_right<N>(V, p, q, j_right);
          max_diag = std::fmax(max_diag, std::fmax(std::fabs(W[p][p]), std::fabs(W[q][q])));
        }
      }
    }
  }
  for (int i = 0; i < N; ++i) {
    const double a = W[i][i];
    S[i] = std::fabs(a);
    if (a < 0.0)
      for (int r = 0; r < N; ++r) U[r][i] = -U[r][i];
  }
  for (int i = 0; i < N; ++i) S[i] *= scale;
  for (int i = 0; i < N; ++i) {
    int pos = i;
    for (int k = i + 1; k < N; ++k)
      if (S[k] > S[pos]) pos = k;
    if (S[pos] == 0.0) break;
    if (pos != i) {
      std::swap(S[i], S[pos]);
      for (int r = 0; r < N; ++r) {
        std::swap(U[r][i], U[r][pos]);
        std::swap(V[r][i], V[r][pos]);
      }
    }
  }
  return sweeps;
}

struct Solve {
  double X[4];
  double u, v, up, vp;
  double rp0[3], rp1[3];
};

// DltTriangulator::solve, src/DltTriangulator.h:36-65
inline void dlt_solve(const double *P0, const double *P1, const double *x0, const double *x1, Solve &s) {
  s.u = x0[0] / x0[2];
  s.v = x0[1] / x0[2];
  s.up = x1[0] / x1[2];
  s.vp = x1[1] / x1[2];
  double A[4][4], U[4][4], S[4], V[4][4];
  for (int c = 0; c < 4; ++c) {
    A[0][c] = s.u * P0[8 + c] - P0[c];
    A[1][c] = s.v * P0[8 + c] - P0[4 + c];
    A[2][c] = s.up * P1[8 + c] - P1[c];
    A[3][c] = s.vp * P1[8 + c] - P1[4 + c];
  }
  jacobi_svd<4>(A, U, S, V);
  for (int i = 0; i < 4; ++i) s.X[i] = V[i][3];
  for (int r = 0; r < 3; ++r) {
    double a = 0.0, b = 0.0;
    for (int c = 0; c < 4; ++c) {
      a += P0[4 * r + c] * s.X[c];
      b += P1[4 * r + c] * s.X[c];
    }
    s.rp0[r] = a;
    s.rp1[r] = b;
  }
}

// src/DltTriangulator.h:67-74
inline double reprojection_error(const Solve &s) {
  const double e0x = s.rp0[0] / s.rp0[2] - s.u, e0y = s.rp0[1] / s.rp0[2] - s.v;
  const double e1x = s.rp1[0] / s.rp1[2] - s.up, e1y = s.rp1[1] / s.rp1[2] - s.vp;
  return std::sqrt(e0x * e0x + e0y * e0y) + std::sqrt(e1x * e1x + e1y * e1y);
}

inline double det3_left(const double *P) {  // determinant of the left 3x3 block of a row-major 3x4
  return P[0] * (P[5] * P[10] - P[6] * P[9]) - P[1] * (P[4] * P[10] - P[6] * P[8]) +
         P[2] * (P[4] * P[9] - P[5] * P[8]);
}

struct Cheirality {  // src/DltTriangulator.h:28-34
  double s0, s1, n0, n1;
  Cheirality(const double *P0, const double *P1) {
    s0 = det3_left(P0) < 0 ? -1.0 : 1.0;
    s1 = det3_left(P1) < 0 ? -1.0 : 1.0;
    n0 = P0[2] * P0[2] + P0[6] * P0[6] + P0[10] * P0[10];
    n1 = P1[2] * P1[2] + P1[6] * P1[6] + P1[10] * P1[10];
  }
  bool infront(const Solve &s) const {  // :76-86
    const double d0 = s0 / n0 * s.rp0[2] / s.X[3];
    const double d1 = s1 / n1 * s.rp1[2] / s.X[3];
    return d0 > 0 && d1 > 0;
  }
};

inline void mat3_mul(const double (&A)[3][3], const double (&B)[3][3], double (&C)[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += A[i][k] * B[k][j];
      C[i][j] = a;
    }
}

// Essential2Cameras, src/Camera.h:31-46.  cams: double[4][12] row-major 3x4.
void essential_to_cameras(const double (&E)[3][3], double *cams) {
  double U[3][3], S[3], V[3][3];
  jacobi_svd<3>(E, U, S, V);
  const double D[3][3] = {{0, 1, 0}, {-1, 0, 0}, {0, 0, 1}};
  const double Dt[3][3] = {{0, -1, 0}, {1, 0, 0}, {0, 0, 1}};
  double Vt[3][3], UD[3][3], Ra[3][3], Rb[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Vt[i][j] = V[j][i];
  mat3_mul(U, D, UD);
  mat3_mul(UD, Vt, Ra);
  mat3_mul(U, Dt, UD);
  mat3_mul(UD, Vt, Rb);
  for (int k = 0; k < 4; ++k) {
    const double(&R)[3][3] = k < 2 ? Ra : Rb;
    const double sg = (k & 1) ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) cams[12 * k + 4 * r + c] = R[r][c];
      cams[12 * k + 4 * r + 3] = sg * U[r][2];
    }
  }
}

}  // namespace

extern "C" {

// A row-major n x n (n = 3 or 4) -> U, S (descending), V as Eigen::JacobiSVD(A, FullU|FullV).
// Returns the number of sweeps, -1 for an unsupported n.
int oracle_jacobisvd(const double *A, int n, double *U, double *S, double *V) {
  if (n == 3) {
    double a[3][3], u[3][3], s[3], v[3][3];
    for (int i = 0; i < 9; ++i) a[i / 3][i % 3] = A[i];
    const int sw = jacobi_svd<3>(a, u, s, v);
    for (int i = 0; i < 9; ++i) {
      U[i] = u[i / 3][i % 3];
      V[i] = v[i / 3][i % 3];
    }
    for (int i = 0; i < 3; ++i) S[i] = s[i];
    return sw;
  }
  if (n == 4) {
    double a[4][4], u[4][4], s[4], v[4][4];
    for (int i = 0; i < 16; ++i) a[i / 4][i % 4] = A[i];
    const int sw = jacobi_svd<4>(a, u, s, v);
    for (int i = 0; i < 16; ++i) {
      U[i] = u[i / 4][i % 4];
      V[i] = v[i / 4][i % 4];
    }
    for (int i = 0; i < 4; ++i) S[i] = s[i];
    return sw;
  }
  return -1;
}

// dst: double[npt,4] = V.col(3), sign as this restatement of JacobiSVD leaves it.
// Serial loop, as reference src/Spectavi.cpp:48-51.
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
    dst[i] = reprojection_error(s);
  }
}

// infront: uint8[npt], 1 iff the point is in front of both cameras (src/DltTriangulator.h:76-86).
void oracle_dlt_cheirality(const double *P0, const double *P1, int npt, const double *x,
                           const double *xp, uint8_t *infront) {
  const Cheirality ch(P0, P1);
  for (int i = 0; i < npt; ++i) {
    Solve s;
    dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
    infront[i] = ch.infront(s) ? 1 : 0;
  }
}

// RANSAC hypothesis scoring: counts int32[nhyp], mask uint8[nhyp,npt] (may be NULL), err
// double[nhyp,npt] (may be NULL: the reprojection error of every (hypothesis, point), so a test
// can tell decisions that sit on the threshold from real disagreements).
// Restates the scoring loops of reference src/RansacFitter.h:59-73 and :86-94.
void oracle_dlt_score_hypotheses(const double *P0, const double *P1s, int nhyp, int npt,
                                 const double *x, const double *xp, double max_error,
                                 int32_t *counts, uint8_t *mask, double *err) {
  for (int h = 0; h < nhyp; ++h) {
    const double *P1 = P1s + 12 * (size_t)h;
    const Cheirality ch(P0, P1);
    int cnt = 0;
    for (int i = 0; i < npt; ++i) {
      Solve s;
      dlt_solve(P0, P1, x + 3 * (size_t)i, xp + 3 * (size_t)i, s);
      const double e = reprojection_error(s);
      const bool in = (e <= max_error) && ch.infront(s);
      cnt += in ? 1 : 0;
      if (mask) mask[(size_t)h * npt + i] = in ? 1 : 0;
      if (err) err[(size_t)h * npt + i] = e;
    }
    counts[h] = cnt;
  }
}

// Essential2Cameras (src/Camera.h:31-46): E double[9] row-major -> cams double[4,3,4].
void oracle_essential_to_cameras(const double *E, double *cams) {
  double e[3][3];
  for (int i = 0; i < 9; ++i) e[i / 3][i % 3] = E[i];
  essential_to_cameras(e, cams);
}

// process_fundamental_matrix (src/RansacFitter.h:42-95) for one candidate F (double[9]).
// Returns 1 on success (a camera reached required_percent_inliers, or find_best_even_in_failure,
// with more inliers than the cameras before it); then *inlier_count, best_P double[12],
// inlier_idx int32[>= npt] (first *inlier_count entries) are set.  *gate_ratio always receives the
// singular-value ratio of :49-50; E_out (may be NULL) the rank-2 essential matrix of :54-56;
// counts4 (may be NULL) the inlier count of each of the four cameras.
int oracle_process_fundamental_matrix(const double *F, double singular_value_ratio_allowed,
                                      const double *x0, const double *x1, int npt,
                                      double required_percent_inliers,
                                      double reprojection_error_allowed,
                                      int find_best_even_in_failure, int32_t *inlier_count,
                                      double *best_P, int32_t *inlier_idx, double *gate_ratio,
                                      double *E_out, int32_t *counts4) {
  double f[3][3], U[3][3], S[3], V[3][3];
  for (int i = 0; i < 9; ++i) f[i / 3][i % 3] = F[i];
  jacobi_svd<3>(f, U, S, V);
  const double ratio = std::fabs(S[0] - S[1]) / (std::fabs(S[0] + S[1]) / 2.);
  if (gate_ratio) *gate_ratio = ratio;
  if (counts4)
    for (int k = 0; k < 4; ++k) counts4[k] = -1;
  if (ratio > singular_value_ratio_allowed) return 0;
  // E = U diag(1,1,0) V^T
  double Ud[3][3], Vt[3][3], E[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Ud[i][j] = U[i][j] * (j < 2 ? 1.0 : 0.0);
      Vt[i][j] = V[j][i];
    }
  mat3_mul(Ud, Vt, E);
  if (E_out)
    for (int i = 0; i < 9; ++i) E_out[i] = E[i / 3][i % 3];
  double cams[48];
  essential_to_cameras(E, cams);
  const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // Camera(): Identity(3,4)
  int success = 0;
  double best_percent = 0;
  for (int k = 0; k < 4; ++k) {
    const double *P1 = cams + 12 * k;
    const Cheirality ch(P0, P1);
    int ninlier = 0;
    for (int i = 0; i < npt; ++i) {
      Solve s;
      dlt_solve(P0, P1, x0 + 3 * (size_t)i, x1 + 3 * (size_t)i, s);
      if (reprojection_error(s) <= reprojection_error_allowed && ch.infront(s)) ++ninlier;
    }
    if (counts4) counts4[k] = ninlier;
    const double percent = ninlier / (double)npt;
    if ((percent >= required_percent_inliers || find_best_even_in_failure) && percent > best_percent) {
      best_percent = percent;
      *inlier_count = ninlier;
      for (int i = 0; i < 12; ++i) best_P[i] = P1[i];
      success = 1;
      int n = 0;
      for (int i = 0; i < npt; ++i) {
        Solve s;
        dlt_solve(P0, P1, x0 + 3 * (size_t)i, x1 + 3 * (size_t)i, s);
        if (reprojection_error(s) <= reprojection_error_allowed && ch.infront(s)) inlier_idx[n++] = i;
      }
    }
  }
  return success;
}

}  // extern "C"
