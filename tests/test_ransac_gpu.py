"""GPU parity: RANSAC candidate processing (SURVEY 8(f) row 1 in full: reference
src/RansacFitter.h:42-95 + src/Camera.h:31-46) through the C-ABI against the oracle
(oracle/oracle_jacobisvd.cpp, the reference's arithmetic incl. Eigen's JacobiSVD restated) and,
independently, against LAPACK for the properties an essential decomposition must have."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OPTS = {'required_percent_inliers': .5, 'reprojection_error_allowed': 1e-2, 'find_best_even_in_failure': False,
        'singular_value_ratio_allowed': 3e-2}


def _small_rotation(rng, max_angle=0.3):
    """Rotation by at most max_angle about a random axis (Rodrigues): keeps a scene in front of both cameras."""
    a = rng.standard_normal(3)
    a /= np.linalg.norm(a)
    th = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _scene(rng, npt=600, noise=2e-3, outliers=7):
    R = _small_rotation(rng)
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    Xw = np.hstack([rng.standard_normal((npt, 2)), rng.uniform(4, 8, (npt, 1)), np.ones((npt, 1))])
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = np.hstack([R, t[:, None]])
    x0, x1 = Xw @ P0.T, Xw @ P1.T
    x1[:, :2] += rng.normal(0, noise, (npt, 2)) * x1[:, 2:3]
    if outliers:
        x1[::outliers] = rng.standard_normal((len(x1[::outliers]), 3))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return x0, x1, tx @ R, P1


def _candidates(rng, E_true, n):
    """True E at arbitrary scale / sign, perturbed copies (some pass the gate, some do not), and
    arbitrary 3x3 matrices (gated)."""
    Fs = []
    for k in range(n):
        kind = k % 5
        if kind == 0:
            Fs.append(E_true * rng.uniform(0.1, 10) * rng.choice([-1, 1]))
        elif kind in (1, 2):
            Fs.append(E_true + 10.0 ** rng.uniform(-6, -2) * rng.standard_normal((3, 3)))
        elif kind == 3:
            Fs.append(E_true + 0.2 * rng.standard_normal((3, 3)))
        else:
            Fs.append(rng.standard_normal((3, 3)))
    return np.stack(Fs)


def _compare_with_oracle(oracle, Fs, x0, x1, opts, out):
    thr = opts['reprojection_error_allowed']
    for f in range(len(Fs)):
        o = oracle.process_fundamental_matrix(Fs[f], x0, x1, opts['singular_value_ratio_allowed'],
                                              opts['required_percent_inliers'], thr, opts['find_best_even_in_failure'])
        assert abs(out['singular_value_ratio'][f] - o['gate_ratio']) <= 1e-12 * max(1.0, o['gate_ratio'])
        gated = o['gate_ratio'] > opts['singular_value_ratio_allowed']
        assert np.array_equal(out['counts4'][f] == -1, o['counts4'] == -1) and (out['counts4'][f][0] == -1) == gated
        if gated:
            assert not out['success'][f] and out['best_camera'][f] == -1 and np.isnan(out['essential'][f]).all()
            continue
        # same Jacobi iteration on both sides: E and the cameras agree to rounding, signs included
        assert np.max(np.abs(out['essential'][f] - o['E'])) <= 1e-13
        ocams = oracle.essential_to_cameras(o['E'])
        # per-camera counts: equal up to the correspondences whose error sits on the threshold
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
        _, _, oerr = oracle.dlt_score_hypotheses(P0, ocams, x0, x1, thr, return_err=True)
        border = (np.abs(oerr - thr) <= 1e-9 * thr).sum(1)
        assert np.all(np.abs(out['counts4'][f] - o['counts4']) <= border), (f, out['counts4'][f], o['counts4'])
        if border.sum() == 0:
            assert bool(out['success'][f]) == o['success']
            assert out['inlier_count'][f] == o['inlier_count']
            if o['success']:
                assert np.max(np.abs(out['camera'][f] - o['best_P'])) <= 1e-13
                assert np.array_equal(np.flatnonzero(out['inlier_mask'][f]), o['inlier_idx'])
                assert np.max(np.abs(out['camera'][f] - ocams[out['best_camera'][f]])) <= 1e-13
            else:
                assert not out['inlier_mask'][f].any() and not out['camera'][f].any()


def test_candidates_match_oracle(oracle):
    from spectavi_amd import mvg
    rng = np.random.default_rng(2026)
    for trial in range(4):
        x0, x1, E_true, P1 = _scene(rng, npt=int(rng.integers(50, 900)), outliers=int(rng.integers(3, 12)))
        Fs = _candidates(rng, E_true, 40)
        for opts in (OPTS, dict(OPTS, find_best_even_in_failure=True, required_percent_inliers=.99)):
            out = mvg.process_fundamental_matrices(Fs, x0, x1, opts, return_mask=True)
            _compare_with_oracle(oracle, Fs, x0, x1, opts, out)
        out = mvg.process_fundamental_matrices(Fs, x0, x1, OPTS, return_mask=True)
        # the true essential matrix recovers the true camera (up to the scale of t) with most points
        k = 0
        assert out['success'][k] and out['inlier_count'][k] > 0.6 * len(x0)
        # ... as a projective camera: the reference does not force det(R) = +1, so -[R | t] (the same
        # camera, which its cheirality test handles through sign(det M)) is as likely as [R | t]
        cam = out['camera'][k]
        assert min(np.abs(cam - P1).max(), np.abs(cam + P1).max()) < 1e-6
        assert np.array_equal(out['inlier_count'], out['inlier_mask'].sum(1))


def test_essential_decomposition_properties_lapack():
    """Independent of the oracle: for every ungated candidate, E has singular values (1, 1, 0)
    (numpy.linalg.svd), its left null vector is +-t of the reported camera, the camera's rotation is
    orthogonal, and [t]x R = +-E: the four-camera construction of src/Camera.h:31-46."""
    from spectavi_amd import mvg
    rng = np.random.default_rng(7)
    x0, x1, E_true, _ = _scene(rng)
    Fs = _candidates(rng, E_true, 200)
    out = mvg.process_fundamental_matrices(Fs, x0, x1, dict(OPTS, find_best_even_in_failure=True))
    s = np.linalg.svd(Fs, compute_uv=False)
    ratio = np.abs(s[:, 0] - s[:, 1]) / (np.abs(s[:, 0] + s[:, 1]) / 2)
    assert np.allclose(out['singular_value_ratio'], ratio, rtol=1e-9, atol=1e-12)
    ok = ratio <= OPTS['singular_value_ratio_allowed'] * (1 - 1e-9)
    assert ok.sum() >= 40 and np.isnan(out['essential'][ratio > OPTS['singular_value_ratio_allowed'] * (1 + 1e-9)]).all()
    E = out['essential'][ok]
    assert np.allclose(np.linalg.svd(E, compute_uv=False), [1, 1, 0], atol=1e-12)
    won = ok & out['success']
    assert won.sum() >= 20
    for f in np.flatnonzero(won):
        P, Ef = out['camera'][f], out['essential'][f]
        R, t = P[:, :3], P[:, 3]
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-12) and abs(np.linalg.norm(t) - 1) < 1e-12  # det may be -1
        assert np.allclose(t @ Ef, 0, atol=1e-12)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        assert min(np.abs(tx @ R - Ef).max(), np.abs(tx @ R + Ef).max()) < 1e-12


def test_many_candidates_and_edge_cases(oracle):
    """More candidates than one launch takes (chunked), a candidate of zeros / NaN, a single
    correspondence; counts stay consistent with the standalone scorer."""
    from spectavi_amd import mvg
    rng = np.random.default_rng(11)
    x0, x1, E_true, _ = _scene(rng, npt=97)
    Fs = np.concatenate([_candidates(rng, E_true, 20000), np.zeros((1, 3, 3)), np.full((1, 3, 3), np.nan)])
    out = mvg.process_fundamental_matrices(Fs, x0, x1, OPTS)
    assert out['success'][0] and not out['success'][-1] and not out['success'][-2]
    sub = np.r_[0:50, 16380:16390, 19990:20002]
    out_sub = mvg.process_fundamental_matrices(Fs[sub], x0, x1, OPTS)
    for key in ('success', 'inlier_count', 'best_camera', 'counts4'):
        assert np.array_equal(out[key][sub], out_sub[key]), key
    assert np.array_equal(out['camera'][sub], out_sub['camera'], equal_nan=True)
    # the winners' counts are what the standalone scorer gives their cameras
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    w = np.flatnonzero(out['success'])[:50]
    c = mvg.dlt_score_hypotheses(P0, out['camera'][w], x0, x1, OPTS['reprojection_error_allowed'])
    assert np.array_equal(c, out['inlier_count'][w])
    one = mvg.process_fundamental_matrices(E_true, x0[:1], x1[:1], dict(OPTS, find_best_even_in_failure=True))
    o = oracle.process_fundamental_matrix(E_true, x0[:1], x1[:1], 3e-2, .5, 1e-2, True)
    assert bool(one['success'][0]) == o['success'] and one['inlier_count'][0] == o['inlier_count']
