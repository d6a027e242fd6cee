"""Independent checks of a DLT result against the reference's DEFINITION, with LAPACK as the solver.

The reference computes X = V.col(3) of the SVD of the 4x4 DLT matrix A (src/DltTriangulator.h:51-58):
the unit right singular vector of the smallest singular value, sign arbitrary.  These helpers
restate that definition with numpy.linalg.svd (LAPACK, importable on the GPU box) so that the HIP
path -- which does not run an SVD at all on its fast path (Gram-Schmidt + inverse iteration) -- is
judged by something that shares no code and no algorithm with it:

  residual     ||A X||_2 <= sigma4 (1 + 1e-9) + 1e-13 ||A||_2          (every finite point)
  direction    |<X, v4>| >= 1 - 1e-9      where (sigma3 - sigma4) / sigma1 > 1e-6
               (below that gap v4 itself is ill-determined: any unit vector of the near-null plane
               with a small residual is as good an answer, and the residual test is what holds)
  unit norm    | ||X|| - 1 | <= 1e-12
  error        the reprojection error (src/DltTriangulator.h:67-74) agrees within 1e-6 relative
               (north_star's float tolerance) + a conditioning-scaled absolute term with the value
               recomputed from LAPACK's v4, on the well-separated points.

Used by tests/test_dlt_gpu.py, tests/fuzz_gpu.py (GPU) and tests/test_oracle.py (CPU, on the oracle
and on the host mirror of the kernel's operation sequence)."""
import numpy as np

GAP = 1e-6


def dlt_matrices(P0, P1, x, xp, return_formation_error=False):
    """A [npt,4,4] and the hnormalised observations (u, v, up, vp), as src/DltTriangulator.h:38-54.

    The entries u*P[2,c] - P[0,c] cancel heavily when a camera's translation column is large (both
    terms ~1e4, the difference ~1): a build without FMA (the reference's: no -march) rounds the
    product first and carries an absolute error eps*|u*P[2,c]|, a fused multiply-add (the HIP
    kernel) rounds once.  To judge X against the matrix the observations DEFINE rather than
    against one particular rounding of it, A is formed in extended precision from the
    float64-rounded u, v, up, vp and rounded once.  `formation` [npt] bounds the 2-norm of the
    difference between that A and the unfused float64 one."""
    P0, P1 = np.asarray(P0, np.float64), np.asarray(P1, np.float64)
    x, xp = np.atleast_2d(np.asarray(x, np.float64)), np.atleast_2d(np.asarray(xp, np.float64))
    L = np.longdouble
    with np.errstate(all="ignore"):
        u, v = x[:, 0] / x[:, 2], x[:, 1] / x[:, 2]
        up, vp = xp[:, 0] / xp[:, 2], xp[:, 1] / xp[:, 2]
        rows = []
        mags = []
        for s, P, a, b in ((u, P0, 2, 0), (v, P0, 2, 1), (up, P1, 2, 0), (vp, P1, 2, 1)):
            rows.append((s.astype(L)[:, None] * P[a].astype(L) - P[b].astype(L)).astype(np.float64))
            mags.append(np.abs(s)[:, None] * np.abs(P[a]))
        A = np.stack(rows, axis=1)
        formation = np.finfo(np.float64).eps * np.linalg.norm(np.stack(mags, axis=1), axis=(1, 2))
    if return_formation_error:
        return A, (u, v, up, vp), formation
    return A, (u, v, up, vp)


def reprojection_error(P0, P1, X, obs):
    """src/DltTriangulator.h:61-62, 67-74 for rows of X."""
    u, v, up, vp = obs
    with np.errstate(all="ignore"):
        r0, r1 = X @ np.asarray(P0).T, X @ np.asarray(P1).T
        e0 = np.hypot(r0[:, 0] / r0[:, 2] - u, r0[:, 1] / r0[:, 2] - v)
        e1 = np.hypot(r1[:, 0] / r1[:, 2] - up, r1[:, 1] / r1[:, 2] - vp)
    return e0 + e1, r0[:, 2], r1[:, 2]


def lapack_svd(A):
    """(S [n,4] descending, V4 [n,4]) for the finite matrices of A; rows of non-finite ones are nan."""
    n = A.shape[0]
    S = np.full((n, 4), np.nan)
    V4 = np.full((n, 4), np.nan)
    ok = np.isfinite(A).all(axis=(1, 2))
    if ok.any():
        _, s, vt = np.linalg.svd(A[ok])
        S[ok], V4[ok] = s, vt[:, 3, :]
    return S, V4, ok


def check_definition(X, P0, P1, x, xp, err=None, what="X", unfused=False):
    """Assert the definition above for rows of X (and, if given, the reprojection errors `err`).
    Non-finite inputs (w = 0, inf, nan observations) are skipped: the reference's JacobiSVD leaves
    them unspecified.  Every implementation solves a matrix that differs from the ideal A (formed in
    extended precision, dlt_matrices) by the order of one rounding of u P[2,c] -- the reference and
    the JacobiSVD oracle round that product (no FMA), the HIP path since round 3 takes u = x0 rcp(x2),
    up to 1.5 ulp from the quotient -- so twice that formation rounding is added to the residual
    bound and, divided by the gap, to the direction and error tolerances (`unfused` is kept for the
    callers that used to ask for it; it no longer changes anything).  Returns a dict of the worst
    margins seen."""
    X = np.asarray(X, np.float64)
    A, obs, formation = dlt_matrices(P0, P1, x, xp, return_formation_error=True)
    S, V4, ok = lapack_svd(A)
    ok &= np.isfinite(X).all(axis=1) & np.isfinite(formation)
    slack = 2.0 * formation
    stats = {"points": int(ok.sum()), "well_separated": 0, "worst_residual_excess": 0.0, "worst_direction": 0.0,
             "worst_err_rel": 0.0}
    if not ok.any():
        return stats
    Xo, Ao, So, Vo, slack = X[ok], A[ok], S[ok], V4[ok], slack[ok]
    nrm = np.linalg.norm(Xo, axis=1)
    assert np.max(np.abs(nrm - 1)) <= 1e-12, "%s: not unit norm (worst %.3e)" % (what, np.max(np.abs(nrm - 1)))
    resid = np.linalg.norm(np.einsum("nij,nj->ni", Ao, Xo), axis=1)
    bound = So[:, 3] * (1 + 1e-9) + 1e-13 * So[:, 0] + slack
    excess = resid - bound
    stats["worst_residual_excess"] = float(np.max(excess / np.maximum(So[:, 0], 1e-300)))
    bad = np.flatnonzero(excess > 0)
    assert bad.size == 0, ("%s: ||A X|| exceeds sigma4 at %d of %d points; worst: resid %.3e sigma %s"
                           % (what, bad.size, resid.size, resid[bad[np.argmax(excess[bad])]],
                              So[bad[np.argmax(excess[bad])]]))
    sep = (So[:, 2] - So[:, 3]) > GAP * So[:, 0]
    stats["well_separated"] = int(sep.sum())
    if sep.any():
        dots = np.abs(np.einsum("ni,ni->n", Xo[sep], Vo[sep]))
        need = 1 - 1e-9 - 2 * (slack[sep] / (So[sep, 2] - So[sep, 3])) ** 2   # 1 - cos(d) ~ d^2 / 2
        stats["worst_direction"] = float(np.max(1 - dots))
        assert np.all(dots >= need), "%s: |<X, v4>| = %.12f at a point with a sigma gap > 1e-6" % (what, np.min(dots))
    if err is not None:
        err = np.asarray(err, np.float64).reshape(-1)[ok]
        e_l, z0, z1 = reprojection_error(P0, P1, Vo, tuple(o[ok] for o in obs))
        # well-separated points whose reprojected depth is not near zero (the perspective division
        # amplifies the direction error of X by 1/depth)
        scale = np.maximum(1.0, np.max(np.abs(np.stack([o[ok] for o in obs])), axis=0))
        depth_ok = (np.abs(z0) > 1e-3) & (np.abs(z1) > 1e-3)
        use = sep & depth_ok & np.isfinite(e_l) & np.isfinite(err)
        if use.any():
            kappa = So[use, 0] / (So[use, 2] - So[use, 3])
            atol = (1e-13 + 4 * slack[use] / So[use, 0]) * kappa * scale[use] / np.minimum(np.abs(z0[use]), np.abs(z1[use]))
            diff = np.abs(err[use] - e_l[use])
            tol = 1e-6 * e_l[use] + atol
            stats["worst_err_rel"] = float(np.max(diff / np.maximum(tol, 1e-300)))
            assert np.all(diff <= tol), ("%s: reprojection error differs from the LAPACK value by %.3e (tol %.3e)"
                                         % (what, diff[np.argmax(diff - tol)], tol[np.argmax(diff - tol)]))
    return stats


def check_against_oracle(X, oX, P0, P1, x, xp, what="X"):
    """X (HIP, sign canonicalised) against the oracle's V.col(3) (Eigen-style JacobiSVD restatement,
    sign as it falls): equal up to sign within (64 eps sigma1 + 8 |dA|) / (sigma3 - sigma4) on the
    well-separated finite points, dA being one rounding of the formation of A (see dlt_matrices): the
    oracle rounds the product u P[2,c] before the subtraction (4 |dA| covered it through round 2), the
    kernel since round 3 forms u = x0 rcp(x2), up to 1.5 ulp from the quotient (fuzz seed 20261004,
    case 5249: 3.4e-14 against 3.1e-14 at 4 |dA|).  Returns the worst ratio to that tolerance."""
    X, oX = np.asarray(X, np.float64), np.asarray(oX, np.float64)
    A, _, formation = dlt_matrices(P0, P1, x, xp, return_formation_error=True)
    S, _, ok = lapack_svd(A)
    ok &= np.isfinite(X).all(axis=1) & np.isfinite(oX).all(axis=1) & np.isfinite(formation)
    ok &= (S[:, 2] - S[:, 3]) > GAP * S[:, 0]
    if not ok.any():
        return 0.0
    sgn = np.sign(np.einsum("ni,ni->n", X[ok], oX[ok]))
    diff = np.max(np.abs(X[ok] - sgn[:, None] * oX[ok]), axis=1)
    # 64 eps sigma1 / gap for the two solvers + both sides' rounding of the formation of A
    tol = (64 * np.finfo(np.float64).eps * S[ok, 0] + 8 * formation[ok]) / (S[ok, 2] - S[ok, 3]) + 1e-15
    # what the kernel's stopping rule leaves: a direction that moved by <= 1e-13 in the last step is
    # 1e-13 rho / (1 - rho) from the limit, rho = (sigma4 / sigma3)^2 the contraction of a step
    rho = (S[ok, 3] / S[ok, 2]) ** 2
    tol = tol + 1e-13 * rho / np.maximum(1 - rho, 1e-3)
    worst = float(np.max(diff / tol))
    assert worst <= 1.0, "%s: differs from the JacobiSVD oracle by %.3e (tol %.3e)" % (
        what, diff[np.argmax(diff / tol)], tol[np.argmax(diff / tol)])
    return worst


def check_against_mirror(X, mX, P0, P1, x, xp, E=None, mE=None, what="X"):
    """The HIP path against the host mirror of its own operation sequence (oracle_dlt_mirror.cpp).
    Until round 3 the two were bit-identical; since the kernel takes its reciprocals and reciprocal
    square roots from v_rcp_f64 / v_rsq_f64 + two Newton steps (< 1 ulp from the mirror's exact
    1/x, 1/sqrt(x)) they agree to a few ulps times the conditioning of the point -- a determinism
    aid (no lane- or shape-dependent path, inf/nan in the same rows), not a parity statement:
      * the rows with a non-finite entry coincide;
      * |X - mX| <= 1e-13 + 32 eps sigma1 / (sigma3 - sigma4) on the finite, well-separated rows
        (the verdict's "<= 1e-13 relative" with the same conditioning term as check_against_oracle);
      * reprojection errors within 1e-9 relative + a conditioning- and depth-scaled absolute term.
    Returns the worst |X - mX| seen on the compared rows."""
    X, mX = np.asarray(X, np.float64), np.asarray(mX, np.float64)
    badX, badM = ~np.isfinite(X).all(axis=1), ~np.isfinite(mX).all(axis=1)
    assert np.array_equal(badX, badM), "%s: non-finite rows differ from the mirror's (%d vs %d)" % (what, badX.sum(), badM.sum())
    A, obs, _ = dlt_matrices(P0, P1, x, xp, return_formation_error=True)
    S, _, ok = lapack_svd(A)
    ok &= ~badX
    sep = ok & ((S[:, 2] - S[:, 3]) > GAP * S[:, 0])
    worst = 0.0
    if sep.any():
        eps = np.finfo(np.float64).eps
        kappa = S[sep, 0] / (S[sep, 2] - S[sep, 3])
        diff = np.max(np.abs(X[sep] - mX[sep]), axis=1)
        tol = 1e-13 + 32 * eps * kappa
        worst = float(diff.max())
        assert np.all(diff <= tol), "%s: differs from the mirror by %.3e (tol %.3e)" % (
            what, diff[np.argmax(diff / tol)], tol[np.argmax(diff / tol)])
        if E is not None:
            E, mE = np.asarray(E, np.float64).reshape(-1)[sep], np.asarray(mE, np.float64).reshape(-1)[sep]
            fin = np.isfinite(E) & np.isfinite(mE)
            assert np.array_equal(np.isfinite(E), np.isfinite(mE)), "%s: non-finite errors differ from the mirror's" % what
            _, z0, z1 = reprojection_error(P0, P1, mX[sep], tuple(o[sep] for o in obs))
            scale = np.maximum(1.0, np.max(np.abs(np.stack([o[sep] for o in obs])), axis=0))
            depth = np.maximum(np.minimum(np.abs(z0), np.abs(z1)), 1e-300)
            atol = 64 * eps * kappa * scale / depth + 1e-15
            # like check_definition: points whose reprojected depth is near zero are left out (the
            # perspective division amplifies a direction difference of X by r / depth^2 there)
            fin &= depth > 1e-3
            d = np.abs(E - mE)[fin]
            t = (1e-9 * np.abs(mE) + atol)[fin]
            assert np.all(d <= t), "%s: reprojection error differs from the mirror's by %.3e (tol %.3e)" % (
                what, d[np.argmax(d / t)], t[np.argmax(d / t)])
    return worst
