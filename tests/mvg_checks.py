"""Independent (numpy / LAPACK) statements of the seven-point algorithm and scene builders shared by
the CPU oracle tests and the GPU parity tests of the RANSAC callers."""
import numpy as np


def skew(s):
    return np.array([[0, -s[2], s[1]], [s[2], 0, -s[0]], [-s[1], s[0], 0]], dtype=np.float64)


def seven_point_rows(x, xp):
    """The 7 x 9 system of reference src/FundamentalMatrixFitter.h:108-123 (x, xp euclidean [7,2])."""
    return np.array([[q[0] * p[0], q[0] * p[1], q[0], q[1] * p[0], q[1] * p[1], q[1], p[0], p[1], 1.0]
                     for p, q in zip(x, xp)])


def numpy_seven_point(x, xp):
    """Real solutions of the seven-point problem by LAPACK + numpy.roots, each normalised to unit
    Frobenius norm.  Also returns how close the cubic is to a root-count change: the smallest
    |imaginary part| among complex roots and the smallest gap between real roots."""
    A = seven_point_rows(x, xp)
    N = np.linalg.svd(A)[2][7:]
    G0, G1 = N[0].reshape(3, 3), N[1].reshape(3, 3)
    zs = np.array([0.0, 1.0, -1.0, 2.0])
    co = np.polyfit(zs, [np.linalg.det(z * G0 + (1 - z) * G1) for z in zs], 3)
    r = np.roots(co)
    real = np.sort(r[np.abs(r.imag) < 1e-12].real)
    cplx = r[np.abs(r.imag) >= 1e-12]
    margin = np.inf
    if len(cplx):
        margin = min(margin, np.abs(cplx.imag).min())
    if len(real) > 1:
        margin = min(margin, np.diff(real).min())
    Fs = []
    for z in real:
        G = z * G0 + (1 - z) * G1
        Fs.append(G / np.linalg.norm(G))
    return np.array(Fs).reshape(-1, 3, 3), margin, abs(co[0])


def parallel(F, G):
    """|cos| of the angle between two 3x3 matrices as 9-vectors."""
    return abs(float((F * G).sum())) / (np.linalg.norm(F) * np.linalg.norm(G))


def epipolar_residual(F, x, xp):
    """max |xp^T F x| over the seven pairs, x / xp euclidean (reference test/test_mvg.py:127-141)."""
    xh = np.c_[x, np.ones(len(x))]
    xph = np.c_[xp, np.ones(len(xp))]
    return float(np.abs(np.einsum('ij,jk,ik->i', xph, F, xh)).max())


def cubic_of_basis(basis):
    """(a, b, c, d) of det(z F0 + (1 - z) F1) for the null-space pair a solver worked in."""
    F0, F1 = np.asarray(basis, dtype=np.float64).reshape(2, 3, 3)
    zs = np.array([0.0, 1.0, -1.0, 2.0])
    return np.polyfit(zs, [np.linalg.det(z * F0 + (1 - z) * F1) for z in zs], 3)


def reference_cubic_is_ill_conditioned(basis):
    """The reference divides the cubic by its leading coefficient a = det(F0 - F1) and solves the
    monic form trigonometrically (src/FundamentalMatrixFitter.h:64-104, :229-233).  When a is tiny --
    a member of the pencil near infinity in the parameter, which depends on the arbitrary basis the
    SVD returned, not on the data -- one root is huge, acos is evaluated next to +-1 and the two
    ordinary roots lose digits by the square root of that ratio (seed 301, case 2191 of the fuzz run:
    a = -7e-11 against 6e-3, roots off by 8e-3, F off by 1e-6 .. 2e-5 in direction, det F = 1e-6
    instead of 1e-16).  That is the reference's arithmetic: oracle and device reproduce it, and the
    independent numpy statement is not asked to agree with it there."""
    a, b, c, d = np.abs(cubic_of_basis(basis))
    return a * 1e4 < max(b, c, d)


def check_seven_point(Fs, x, xp, what, rel=1e-9, basis=None):
    """Every F of a solver (rows of Fs [k,3,3]) is a seven-point solution by the independent
    statement: parallel to a numpy solution, and the counts agree unless the cubic sits at a
    root-count change.  With the solver's own null-space pair given, cases in which the reference's
    cubic solver is ill-conditioned in that basis are not compared."""
    if basis is not None and reference_cubic_is_ill_conditioned(basis):
        return
    ref, margin, lead = numpy_seven_point(x, xp)
    for F in Fs:
        assert np.all(np.isfinite(F)), what
        best = max((parallel(F, G) for G in ref), default=0.0)
        assert best >= 1 - rel, "%s: F is not a numpy solution (cos %.3e away from 1)" % (what, 1 - best)
    if margin > 1e-6 and lead > 1e-10:
        assert len(Fs) == len(ref), "%s: %d roots, numpy has %d (margin %.2e)" % (what, len(Fs), len(ref), margin)


def reference_ransac_scene(rng, npt=200):
    """The scene of the reference's RANSAC test (test/test_mvg.py:38-66): two cameras ~85 units from
    a unit cloud, exact correspondences in camera coordinates.  Returns x0, x1 and the true
    essential matrix scaled to a unit largest singular value."""
    C0 = (rng.standard_normal(3) + 1.0) * 50.0
    C1 = (rng.standard_normal(3) - 1.0) * 50.0

    def rot(a, b):
        sk = skew(np.cross(a, b))
        return np.eye(3) + sk + sk @ sk / (1 + a @ b)

    canon = np.array([1.0, 0.0, 0.0])
    R0 = rot(canon, -C0 / np.linalg.norm(C0))
    R1 = rot(canon, -C1 / np.linalg.norm(C1))
    P0 = np.hstack([R0, (R0 @ -C0)[:, None]])
    P1 = np.hstack([R1, (R1 @ -C1)[:, None]])
    X = np.hstack([rng.standard_normal((npt, 3)), np.ones((npt, 1))])
    e = P1 @ np.r_[C0, 1.0]
    E = skew(e) @ P1 @ (P0.T @ np.linalg.inv(P0 @ P0.T))
    return X @ P0.T, X @ P1.T, E / np.linalg.svd(E)[1][0]


def two_view_scene(rng, npt=200, outlier_fraction=0.0, noise=0.0, max_angle=0.3):
    """A well-posed calibrated pair: first camera [I | 0], second [R | t] with a small rotation and a
    unit baseline, points 4-8 units in front of both; `noise` (std, image units) on the second
    view, a fraction of the second view replaced by gross outliers.  Returns x0, x1 (homogeneous),
    the true essential matrix (unit largest singular value) and the sorted outlier rows."""
    a = rng.standard_normal(3)
    a /= np.linalg.norm(a)
    th = rng.uniform(-max_angle, max_angle)
    K = skew(a)
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    Xw = np.hstack([rng.uniform(-2, 2, (npt, 2)), rng.uniform(4, 8, (npt, 1)), np.ones((npt, 1))])
    x0 = Xw[:, :3].copy()
    x1 = Xw @ np.hstack([R, t[:, None]]).T
    if noise:
        x1[:, :2] += rng.normal(0, noise, (npt, 2)) * x1[:, 2:3]
    nout = int(round(outlier_fraction * npt))
    out_idx = np.sort(rng.choice(npt, nout, replace=False)) if nout else np.zeros(0, np.int64)
    if nout:
        x1[out_idx] = np.c_[rng.uniform(-0.6, 0.6, (nout, 2)), np.ones(nout)] * rng.uniform(4, 8, (nout, 1))
    E = skew(t) @ R
    return x0, x1, E / np.linalg.svd(E)[1][0], out_idx


def essential_agrees(rE, E):
    """The reference test's criterion (test/test_mvg.py:70-91): after scaling by the largest singular
    value the element-wise ratio is constant (std < 1e-2)."""
    rE = rE / np.linalg.svd(rE)[1][0]
    return float(np.std(rE / E))
