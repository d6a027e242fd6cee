// oracle_ransac.cpp -- CPU restatement of the reference's seven-point solver and RANSAC loop.
//
// TEST INFRASTRUCTURE ONLY (see oracle_l1k2.cpp header): never imported, linked or called from
// spectavi_amd/.  Oracle for the callers of SURVEY row (f)1: `seven_point_algorithm` and
// `ransac_fitter` (reference src/Spectavi.cpp:14-36, :70-87).
//
// The reference computes (read as text, restated, not copied):
//   src/FundamentalMatrixFitter.h:108-123  add_putative_match: the row
//                                 [x'x, x'y, x', y'x, y'y, y', x, y, 1] of the 7 x 9 system
//   src/FundamentalMatrixFitter.h:127-141  solve: JacobiSVD(A, FullU|FullV); F0 = V.col(7),
//                                 F1 = V.col(8), each read as a row-major 3 x 3
//   :143-227                      the coefficients a, b, c, d of det(z F0 + (1-z) F1) (a 48-, 48-,
//                                 24- and 6-term expansion there; here through the four mixed
//                                 determinants m0..m3 of the pencil, which is the same polynomial:
//                                 a = m0 - m1 + m2 - m3, b = m1 - 2 m2 + 3 m3, c = m2 - 3 m3, d = m3)
//   :229-231                      |a| < 1e-14 -> no solution
//   :233-246                      roots of z^3 + (b/a) z^2 + (c/a) z + d/a by solve_cubic (:64-104:
//                                 the trigonometric form for three real roots, Cardano's otherwise;
//                                 a double root is reported when |x[2]| < 1e-14), F = z F0 + (1-z) F1
//   src/RansacFitter.h:152-272    fit_essential: per try a 7-subset (floyd_sample, :120-132),
//                                 hnormalized rows (:181-182), the seven-point solve, then every
//                                 root through process_fundamental_matrix (oracle_jacobisvd.cpp) and
//                                 the best-model update of :196-214
//
// Third-party arithmetic absent from /root/reference: Eigen's JacobiSVD of a 7 x 9 matrix.  With
// more columns than rows it runs its QR preconditioner first (Eigen 3.3.x / 3.4.x,
// Eigen/src/SVD/JacobiSVD.h `qr_preconditioner_impl<..., ColPivHouseholderQRPreconditioner,
// PreconditionIfMoreColsThanRows, true>::run`): ColPivHouseholderQR of the 9 x 7 adjoint of
// A / max|a_ij|, V = its Householder Q.  The Jacobi sweeps and the final sort afterwards only touch
// the first seven columns of V, so V.col(7) and V.col(8) are the last two columns of that Q.
// Restated here from the published algorithm (Eigen/src/QR/ColPivHouseholderQR.h `computeInPlace`:
// largest updated column norm pivots, LAPACK-working-note-176 norm downdating;
// Eigen/src/Householder/Householder.h `makeHouseholder`, `applyHouseholderOnTheLeft`;
// HouseholderSequence `evalTo`).  Eigen evaluates the norms and dot products inside with SSE2
// packets (two partial sums); this file sums sequentially, so results agree with a build of the
// reference to rounding, not to the bit -- PARITY UNPINNED in that sense, as is the Eigen version
// (reference CMakeLists.txt:20).  The basis (F0, F1) of the two-dimensional null space depends on
// that arithmetic; the solutions F do not, up to scale: a singular member of the pencil has exactly
// one representative whose two coefficients sum to one.
//
// PINNING (no reference-held vector exists): tests/test_oracle.py checks the results against the
// reference's own test properties (test/test_mvg.py:127-160: x'^T F x = 0 to 1e-10 for every
// returned F; the simulated F recovered to 1e-8) and against numpy (LAPACK null space + numpy.roots):
// same number of real roots away from a vanishing discriminant, every F parallel to a numpy one.
//
// Sampling: the reference draws each 7-subset from a fresh std::mt19937 seeded by
// std::random_device (:123-124) and walks an unordered_set -- neither the subsets nor the order of
// the seven rows is reproducible.  The fit below therefore takes the subsets as an argument, rows
// in the order given; the update rule is the reference's with nthread = 1 (tries in order).

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cfloat>
#include <utility>
#include <vector>

extern "C" int oracle_process_fundamental_matrix(const double *F, double singular_value_ratio_allowed,
                                                 const double *x0, const double *x1, int npt,
                                                 double required_percent_inliers,
                                                 double reprojection_error_allowed,
                                                 int find_best_even_in_failure, int32_t *inlier_count,
                                                 double *best_P, int32_t *inlier_idx, double *gate_ratio,
                                                 double *E_out, int32_t *counts4);

namespace {

constexpr int R = 9;  // rows of the adjoint
constexpr int C = 7;  // columns of the adjoint

// Householder vector of v[0..n): H v = beta e0, H = I - tau w w^T, w = (1, essential).
void make_householder(const double *v, int n, double *essential, double &tau, double &beta) {
  double tail = 0.0;
  for (int i = 1; i < n; ++i) tail += v[i] * v[i];
  const double c0 = v[0];
  if (n == 1 || tail <= DBL_MIN) {
    tau = 0.0;
    beta = c0;
    for (int i = 1; i < n; ++i) essential[i - 1] = 0.0;
  } else {
    beta = std::sqrt(c0 * c0 + tail);
    if (c0 >= 0.0) beta = -beta;
    for (int i = 1; i < n; ++i) essential[i - 1] = v[i] / (c0 - beta);
    tau = (beta - c0) / beta;
  }
}

// Last two columns of the Householder Q of the column-pivoted QR of M (9 x 7, row-major).
void null_space_basis(double (&M)[R][C], double (&q7)[R], double (&q8)[R]) {
  double norm_upd[C], norm_dir[C], hcoef[C], ess[C][R];
  for (int j = 0; j < C; ++j) {
    double s = 0.0;
    for (int i = 0; i < R; ++i) s += M[i][j] * M[i][j];
    norm_dir[j] = norm_upd[j] = std::sqrt(s);
  }
  const double downdate_threshold = std::sqrt(DBL_EPSILON);
  for (int k = 0; k < C; ++k) {
    int big = k;
    for (int j = k + 1; j < C; ++j)
      if (norm_upd[j] > norm_upd[big]) big = j;  // first maximum wins
    if (big != k) {
      for (int i = 0; i < R; ++i) std::swap(M[i][k], M[i][big]);
      std::swap(norm_upd[k], norm_upd[big]);
      std::swap(norm_dir[k], norm_dir[big]);
    }
    double col[R], beta;
    for (int i = k; i < R; ++i) col[i - k] = M[i][k];
    make_householder(col, R - k, ess[k], hcoef[k], beta);
    M[k][k] = beta;
    for (int i = k + 1; i < R; ++i) M[i][k] = ess[k][i - k - 1];
    if (hcoef[k] != 0.0) {  // the reflector on the block to the right of column k
      for (int j = k + 1; j < C; ++j) {
        double tmp = 0.0;
        for (int i = k + 1; i < R; ++i) tmp += ess[k][i - k - 1] * M[i][j];
        tmp += M[k][j];
        M[k][j] -= hcoef[k] * tmp;
        for (int i = k + 1; i < R; ++i) M[i][j] -= hcoef[k] * ess[k][i - k - 1] * tmp;
      }
    }
    for (int j = k + 1; j < C; ++j) {  // norm downdate
      if (norm_upd[j] != 0.0) {
        double temp = std::fabs(M[k][j]) / norm_upd[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double r = norm_upd[j] / norm_dir[j];
        const double temp2 = temp * (r * r);
        if (temp2 <= downdate_threshold) {
          double s = 0.0;
          for (int i = k + 1; i < R; ++i) s += M[i][j] * M[i][j];
          norm_dir[j] = norm_upd[j] = std::sqrt(s);
        } else {
          norm_upd[j] *= std::sqrt(temp);
        }
      }
    }
  }
  // Q = H0 H1 ... H6 applied to e7 and e8: reflectors from the last to the first
  for (int i = 0; i < R; ++i) {
    q7[i] = (i == 7) ? 1.0 : 0.0;
    q8[i] = (i == 8) ? 1.0 : 0.0;
  }
  for (int k = C - 1; k >= 0; --k) {
    if (hcoef[k] == 0.0) continue;
    double *cols[2] = {q7, q8};
    for (double *q : cols) {
      double tmp = 0.0;
      for (int i = k + 1; i < R; ++i) tmp += ess[k][i - k - 1] * q[i];
      tmp += q[k];
      q[k] -= hcoef[k] * tmp;
      for (int i = k + 1; i < R; ++i) q[i] -= hcoef[k] * ess[k][i - k - 1] * tmp;
    }
  }
}

inline double det3(const double *r0, const double *r1, const double *r2) {
  return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) +
         r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

// x^3 + a x^2 + b x + c = 0; src/FundamentalMatrixFitter.h:64-104
int solve_cubic(double *x, double a, double b, double c) {
  const double eps = 1e-14, two_pi = 6.28318530717958648;
  const double a2 = a * a;
  double q = (a2 - 3 * b) / 9;
  const double r = (a * (2 * a2 - 9 * b) + 27 * c) / 54;
  const double r2 = r * r;
  const double q3 = q * q * q;
  if (r2 < q3) {
    double t = r / std::sqrt(q3);
    if (t < -1) t = -1;
    if (t > 1) t = 1;
    t = std::acos(t);
    a /= 3;
    q = -2 * std::sqrt(q);
    x[0] = q * std::cos(t / 3) - a;
    x[1] = q * std::cos((t + two_pi) / 3) - a;
    x[2] = q * std::cos((t - two_pi) / 3) - a;
    return 3;
  }
  double A = -std::pow(std::fabs(r) + std::sqrt(r2 - q3), 1. / 3);
  if (r < 0) A = -A;
  const double B = A == 0 ? 0 : q / A;
  a /= 3;
  x[0] = (A + B) - a;
  x[1] = -0.5 * (A + B) - a;
  x[2] = 0.5 * std::sqrt(3.) * (A - B);
  if (std::fabs(x[2]) < eps) {
    x[2] = x[1];
    return 2;
  }
  return 1;
}

// Seven correspondences (euclidean, x[i] = (x, y), xp[i] = (x', y')) -> up to three F (row-major).
// basis (may be NULL): F0 then F1, 18 doubles.
int seven_point(const double *x, const double *xp, double *Fs, double *basis) {
  double A[C][R];
  double scale = 0.0;
  for (int i = 0; i < C; ++i) {
    const double px = x[2 * i], py = x[2 * i + 1], qx = xp[2 * i], qy = xp[2 * i + 1];
    const double row[R] = {qx * px, qx * py, qx, qy * px, qy * py, qy, px, py, 1.0};
    for (int j = 0; j < R; ++j) {
      A[i][j] = row[j];
      scale = std::fmax(scale, std::fabs(row[j]));
    }
  }
  if (scale == 0.0) scale = 1.0;
  double M[R][C];
  for (int i = 0; i < C; ++i)
    for (int j = 0; j < R; ++j) M[j][i] = A[i][j] / scale;
  double F0[R], F1[R];
  null_space_basis(M, F0, F1);
  if (basis)
    for (int i = 0; i < R; ++i) {
      basis[i] = F0[i];
      basis[R + i] = F1[i];
    }
  // det(z F0 + w F1) = m0 z^3 + m1 z^2 w + m2 z w^2 + m3 w^3
  const double m0 = det3(F0, F0 + 3, F0 + 6);
  const double m1 = det3(F1, F0 + 3, F0 + 6) + det3(F0, F1 + 3, F0 + 6) + det3(F0, F0 + 3, F1 + 6);
  const double m2 = det3(F0, F1 + 3, F1 + 6) + det3(F1, F0 + 3, F1 + 6) + det3(F1, F1 + 3, F0 + 6);
  const double m3 = det3(F1, F1 + 3, F1 + 6);
  const double a = m0 - m1 + m2 - m3;
  const double b = m1 - 2 * m2 + 3 * m3;
  const double c = m2 - 3 * m3;
  const double d = m3;
  if (std::fabs(a) < 1e-14) return 0;
  double alpha[3];
  const int nroots = solve_cubic(alpha, b / a, c / a, d / a);
  for (int k = 0; k < nroots; ++k)
    for (int i = 0; i < R; ++i) Fs[R * k + i] = alpha[k] * F0[i] + (1 - alpha[k]) * F1[i];
  return nroots;
}

}  // namespace

extern "C" {

// seven_point_algorithm (src/Spectavi.cpp:14-36): x, xp double[7,2]; Fs double[3,9] (the first
// *nroot filled); basis double[2,9] may be NULL.
void oracle_seven_point(const double *x, const double *xp, int *nroot, double *Fs, double *basis) {
  *nroot = seven_point(x, xp, Fs, basis);
}

// RansacFitter::fit_essential with nthread = 1 (src/RansacFitter.h:152-272) over the given
// 7-subsets (samples int32[ntries,7], rows of x0 / x1 in that order).
//   x0, x1 double[npt,3] homogeneous; outputs: *success, essential double[9] (the seven-point
//   solution F that won, as the reference returns it: m_best_fit_essential_matrix = F, :205),
//   camera double[12], *inlier_percent, inlier_idx int32[npt] / *n_inliers, *best_try / *best_root
//   (-1 when no model was kept; then essential and camera are left untouched).
void oracle_ransac_fit(const double *x0, const double *x1, int npt, double required_percent_inliers,
                       double reprojection_error_allowed, int find_best_even_in_failure,
                       double singular_value_ratio_allowed, const int32_t *samples, int ntries,
                       int *success, double *essential, double *camera, double *inlier_percent,
                       int32_t *inlier_idx, int *n_inliers, int *best_try, int *best_root) {
  bool ok = false;
  double best_percent = 0.0;
  *best_try = *best_root = -1;
  *n_inliers = 0;
  std::vector<int32_t> idx(npt > 0 ? npt : 1);
  for (int itry = 0; itry < ntries; ++itry) {
    if (ok) continue;
    double x[14], xp[14];
    for (int i = 0; i < 7; ++i) {
      const int s = samples[7 * (size_t)itry + i];
      x[2 * i] = x0[3 * (size_t)s] / x0[3 * (size_t)s + 2];
      x[2 * i + 1] = x0[3 * (size_t)s + 1] / x0[3 * (size_t)s + 2];
      xp[2 * i] = x1[3 * (size_t)s] / x1[3 * (size_t)s + 2];
      xp[2 * i + 1] = x1[3 * (size_t)s + 1] / x1[3 * (size_t)s + 2];
    }
    double Fs[27];
    const int nroot = seven_point(x, xp, Fs, nullptr);
    for (int k = 0; k < nroot && !ok; ++k) {  // once successful nothing is updated any more (:202)
      int32_t ninlier = 0;
      double cam[12];
      if (!oracle_process_fundamental_matrix(Fs + 9 * k, singular_value_ratio_allowed, x0, x1, npt,
                                             required_percent_inliers, reprojection_error_allowed,
                                             find_best_even_in_failure, &ninlier, cam, idx.data(), nullptr,
                                             nullptr, nullptr))
        continue;
      const double percent = (double)ninlier / (double)npt;
      if ((percent > required_percent_inliers || find_best_even_in_failure) && best_percent < percent) {
        if (!ok) {
          ok = percent > required_percent_inliers;
          best_percent = percent;
          for (int i = 0; i < 9; ++i) essential[i] = Fs[9 * k + i];
          for (int i = 0; i < 12; ++i) camera[i] = cam[i];
          for (int i = 0; i < ninlier; ++i) inlier_idx[i] = idx[i];
          *n_inliers = ninlier;
          *best_try = itry;
          *best_root = k;
        }
      }
    }
  }
  *success = ok ? 1 : 0;
  *inlier_percent = best_percent;
}

}  // extern "C"
