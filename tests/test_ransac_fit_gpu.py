"""GPU parity: the callers of the RANSAC candidate processing -- `seven_point_algorithm` and
`ransac_fitter` (reference src/Spectavi.cpp:14-36, :70-87; src/FundamentalMatrixFitter.h:108-246;
src/RansacFitter.h:152-272) -- through the C-ABI against the oracle (oracle/oracle_ransac.cpp, the
reference's algorithm incl. Eigen's QR-preconditioned null space restated), against the independent
LAPACK + numpy.roots statement (tests/mvg_checks.py) and against the reference's own test properties
(test/test_mvg.py:38-91, :127-160)."""
import ctypes as ct

import numpy as np
import pytest

from tests import mvg_checks as mc

pytestmark = pytest.mark.gpu


def _euclid(rng, n, kind):
    if kind == "randn":  # the reference tests' inputs: hnormalized N(0,1) triples (heavy tails)
        x, xp = rng.standard_normal((n, 7, 3)), rng.standard_normal((n, 7, 3))
        return x[..., :2] / x[..., 2:], xp[..., :2] / xp[..., 2:]
    xs, xps = [], []
    while len(xs) < n:  # 7-subsets of consistent two-view scenes (one exact solution among the roots)
        x0, x1, _, _ = mc.two_view_scene(rng, npt=70, noise=1e-3 if kind == "noisy" else 0.0)
        for k in range(10):
            xs.append(x0[7 * k:7 * k + 7, :2] / x0[7 * k:7 * k + 7, 2:])
            xps.append(x1[7 * k:7 * k + 7, :2] / x1[7 * k:7 * k + 7, 2:])
    return np.array(xs[:n]), np.array(xps[:n])


@pytest.mark.parametrize("kind,n", [("randn", 1500), ("exact", 500), ("noisy", 500)])
def test_seven_point_batch_matches_oracle_and_numpy(oracle, kind, n):
    from spectavi_amd import mvg
    rng = np.random.default_rng({"randn": 11, "exact": 12, "noisy": 13}[kind])
    x, xp = _euclid(rng, n, kind)
    nroot, Fs, basis = mvg.seven_point_batch(x, xp, return_basis=True)
    assert nroot.min() >= 0 and nroot.max() <= 3
    same_basis = 0
    for i in range(n):
        k = int(nroot[i])
        assert np.all(np.isnan(Fs[i, k:])) and np.all(np.isfinite(Fs[i, :k]))
        oFs, ob = oracle.seven_point(x[i], xp[i], return_basis=True)
        # same plane (the kernel runs the oracle's algorithm; only libm's pow / acos / cos differ, after this)
        B, OB = basis[i].reshape(2, 9), ob.reshape(2, 9)
        assert np.abs(B.T @ B - OB.T @ OB).max() < 1e-10, i
        same_basis += np.array_equal(B, OB)
        _, margin, lead = mc.numpy_seven_point(x[i], xp[i])
        if margin > 1e-6 and lead > 1e-10:
            assert k == len(oFs), (i, k, len(oFs), margin)
        tol = 1e-6 if mc.reference_cubic_is_ill_conditioned(basis[i]) else 1e-9  # libm's acos next to +-1 is amplified there
        for F in Fs[i, :k]:
            assert max((mc.parallel(F, G) for G in oFs), default=0.0) >= 1 - tol, i
        mc.check_seven_point(Fs[i, :k], x[i], xp[i], "%s case %d" % (kind, i), basis=basis[i])
        scale = max(1.0, np.abs(mc.seven_point_rows(x[i], xp[i])).max())
        for F in Fs[i, :k]:
            assert mc.epipolar_residual(F, x[i], xp[i]) < 1e-10 * scale * max(1.0, np.abs(F).max())
    assert same_basis >= 0.99 * n  # unfused arithmetic on both sides: the null-space pair is the same bits


def test_seven_point_device_resident_equals_host_entry():
    import torch
    from spectavi_amd import device as spv
    from spectavi_amd import mvg
    rng = np.random.default_rng(21)
    x, xp = _euclid(rng, 777, "randn")
    nroot, Fs, basis = mvg.seven_point_batch(x, xp, return_basis=True)
    dn, dF, db = spv.seven_point(torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda(), want_basis=True)
    assert np.array_equal(dn.cpu().numpy(), nroot)
    assert np.array_equal(dF.cpu().numpy(), Fs, equal_nan=True)
    assert np.array_equal(db.cpu().numpy(), basis)


def test_seven_point_algorithm_reference_tests():
    """reference test/test_mvg.py:127-160 through the drop-in symbol."""
    from spectavi_amd import mvg
    np.random.seed(0xdeadbeef)
    for _ in range(100):
        x0, x1 = np.random.randn(7, 3), np.random.randn(7, 3)
        FF = mvg.seven_point_algorithm(x0, x1)
        assert FF.shape[0] % 3 == 0 and FF.shape[1:] == (3,)
        xe, xpe = mvg.hnormalize(x0), mvg.hnormalize(x1)
        scale = max(1.0, np.abs(mc.seven_point_rows(xe, xpe)).max())
        for i in range(FF.shape[0] // 3):
            F = FF[3 * i:3 * i + 3]
            # the reference asserts < 1e-10 on np.sum(np.dot(x1, F) * x0, axis=1) with homogeneous inputs
            assert np.max(np.abs(np.sum((x1 @ F) * x0, axis=1))) < 1e-10 * scale * max(1.0, np.abs(F).max())
    for _ in range(100):
        P0 = np.hstack((np.eye(3), np.zeros((3, 1))))
        P1 = np.random.randn(3, 4)
        F0 = mc.skew(P1.T[-1]) @ P1 @ (P0.T @ np.linalg.inv(P0 @ P0.T))
        X = np.random.randn(7, 4)
        FF = mvg.seven_point_algorithm(X @ P0.T, X @ P1.T)
        nF = FF.shape[0] // 3
        assert any(np.std(FF[3 * i:3 * (i + 1)] / F0) < 1e-8 for i in range(nF))
    with pytest.raises(TypeError):
        mvg.seven_point_algorithm(np.zeros((6, 2)), np.zeros((6, 2)))


def test_seven_point_degenerate_inputs_do_not_fault():
    from spectavi_amd import mvg
    x = np.zeros((6, 7, 2))
    xp = np.zeros((6, 7, 2))
    rng = np.random.default_rng(5)
    x[1], xp[1] = rng.standard_normal((7, 2)), rng.standard_normal((7, 2))
    x[1, 3:] = x[1, 0]  # repeated correspondences: the null space has more than two dimensions
    xp[1, 3:] = xp[1, 0]
    x[2], xp[2] = rng.standard_normal((7, 2)), np.nan
    x[3], xp[3] = rng.standard_normal((7, 2)) * 1e150, rng.standard_normal((7, 2)) * 1e150
    x[4] = xp[4] = rng.standard_normal((7, 2))  # identical views: F is skew, a one-parameter family
    x[5], xp[5] = rng.standard_normal((7, 2)), np.inf
    nroot, Fs = mvg.seven_point_batch(x, xp)
    assert np.all((nroot >= 0) & (nroot <= 3))
    for i in range(6):
        assert np.all(np.isnan(Fs[i, nroot[i]:]))


def _same_model(dev, o, exact_inliers=True):
    assert dev['success'] == o['success']
    assert dev['best_try'] == o['best_try'] and dev['best_root'] == o['best_root']
    assert dev['inlier_percent'] == o['inlier_percent']
    if exact_inliers:
        assert np.array_equal(dev['inlier_idx'], o['inlier_idx'])
    if o['best_try'] < 0:
        assert dev['essential'] is None and dev['camera'] is None
        return
    assert mc.parallel(dev['essential'], o['essential']) >= 1 - 1e-9
    assert np.allclose(dev['essential'], o['essential'], rtol=1e-8, atol=1e-10 * np.abs(o['essential']).max())
    # up to the sign of the whole matrix: E has two equal singular values, so the order Essential2Cameras'
    # SVD gives them (and with it the signs of t and R) hangs on the last bit of F; -P is the same camera
    assert min(np.abs(dev['camera'] - o['camera']).max(), np.abs(dev['camera'] + o['camera']).max()) < 1e-8


@pytest.mark.parametrize("npt,outliers,noise", [(200, 0.25, 0.0), (777, 0.3, 2e-4), (64, 0.1, 0.0), (2500, 0.35, 1e-4)])
def test_ransac_fit_matches_oracle_given_the_same_subsets(oracle, npt, outliers, noise):
    from spectavi_amd import mvg
    rng = np.random.default_rng(npt)
    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=npt, outlier_fraction=outliers, noise=noise)
    thr = 1e-3 if noise == 0 else 6 * noise
    samples = mvg.ransac_sample(1234 + npt, npt, 150 if npt < 2000 else 60)
    clean = 1.0 - outliers
    for req, find_best in [(clean - 0.1, False), (clean - 0.1, True), (0.995, False), (0.995, True)]:
        kw = dict(required_percent_inliers=req, reprojection_error_allowed=thr, find_best_even_in_failure=find_best,
                  singular_value_ratio_allowed=3e-2)
        o = oracle.ransac_fit(x0, x1, samples, **kw)
        dev = mvg.ransac_fit(x0, x1, samples=samples, **kw)
        _same_model(dev, o, exact_inliers=(noise == 0))
        if noise:  # decisions within rounding of the threshold may differ: none of them may be far from it
            diff = np.setxor1d(dev['inlier_idx'], o['inlier_idx'])
            assert len(diff) <= 2
        if req < clean and (noise == 0 or o['success']):  # with noise a clean subset need not reach the share
            assert dev['success'] and mc.essential_agrees(dev['essential'], E) < (1e-6 if noise == 0 else 0.25)
            if noise == 0:
                assert np.array_equal(dev['inlier_idx'], np.setdiff1d(np.arange(npt), out_idx))
            assert dev['tries_run'] >= dev['best_try'] + 1
        elif req < clean:
            assert not dev['success']
        elif find_best:
            assert not dev['success'] and dev['best_try'] >= 0
        else:
            assert not dev['success'] and dev['best_try'] == -1 and len(dev['inlier_idx']) == 0
            assert dev['tries_run'] == len(samples)


def test_ransac_fit_late_success_crosses_batches(oracle):
    """60 % outliers: a clean 7-subset turns up once in ~600 tries, so the winner lies beyond the
    first (256) and usually the second (1024) batch; the result is still the first success in try
    order, as the oracle's serial loop finds it, and nothing after its batch is evaluated."""
    from spectavi_amd import mvg
    rng = np.random.default_rng(99)
    npt = 300
    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=npt, outlier_fraction=0.6)
    samples = mvg.ransac_sample(77, npt, 6000)
    kw = dict(required_percent_inliers=0.35, reprojection_error_allowed=1e-3, find_best_even_in_failure=True,
              singular_value_ratio_allowed=3e-2)
    o = oracle.ransac_fit(x0, x1, samples, **kw)
    dev = mvg.ransac_fit(x0, x1, samples=samples, **kw)
    assert o['success'] and o['best_try'] >= 256
    _same_model(dev, o)
    assert dev['best_try'] < dev['tries_run'] <= 6000
    assert np.array_equal(dev['inlier_idx'], np.setdiff1d(np.arange(npt), out_idx))
    # the same through the seed: spv_ransac_fit(seed) draws what spv_ransac_sample(seed) returns
    dev2 = mvg.ransac_fit(x0, x1, maximum_tries=6000, seed=77, **kw)
    assert dev2['best_try'] == dev['best_try'] and np.array_equal(dev2['essential'], dev['essential'])
    assert np.array_equal(dev2['inlier_idx'], dev['inlier_idx'])
    # running out of tries before the first clean subset: the best contaminated model, not a success
    short = mvg.ransac_fit(x0, x1, samples=samples[:200], **kw)
    o_short = oracle.ransac_fit(x0, x1, samples[:200], **kw)
    _same_model(short, o_short)
    assert not short['success'] and short['tries_run'] == 200


def test_ransac_fitter_dropin(monkeypatch):
    """The reference's own RANSAC test (test/test_mvg.py:38-91) through the reference's symbol and
    front-end, and what the symbol returns when no model is kept."""
    from spectavi_amd import mvg
    monkeypatch.setenv("SPECTAVI_RANSAC_SEED", "20241004")
    rng = np.random.default_rng(0xdeadbeef)
    x0, x1, E = mc.reference_ransac_scene(rng)
    opts = {'required_percent_inliers': .9, 'reprojection_error_allowed': .5, 'maximum_tries': 200,
            'find_best_even_in_failure': False, 'singular_value_ratio_allowed': 3e-2, 'progressbar': False}
    r = mvg.ransac_fitter(x0, x1, options=opts)
    assert r['success'] is True
    assert r['essential'].shape == (3, 3) and r['camera'].shape == (3, 4)
    assert mc.essential_agrees(r['essential'], E) < 1e-2
    n = int(round(r['inlier_percent'] * 200))
    assert r['inlier_idx'].dtype == np.int32 and r['inlier_idx'].shape == (n, 1) and n > 180
    r_again = mvg.ransac_fitter(x0, x1, options=opts)  # same seed, same answer
    assert np.array_equal(r_again['essential'], r['essential'])
    # contaminated scene, defaults of the front-end apart from the threshold
    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=400, outlier_fraction=0.2)
    opts2 = dict(opts, reprojection_error_allowed=1e-3, required_percent_inliers=0.75, maximum_tries=500)
    r = mvg.ransac_fitter(x0, x1, options=opts2)
    assert r['success'] and mc.essential_agrees(r['essential'], E) < 1e-6
    assert np.array_equal(r['inlier_idx'][:, 0], np.setdiff1d(np.arange(400), out_idx))
    # the camera reprojects the inliers: it is one of the four decompositions of E, in front of both
    from spectavi_amd import mvg as m
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    err = m.dlt_reprojection_error(P0, r['camera'], x0[r['inlier_idx'][:, 0]], x1[r['inlier_idx'][:, 0]])
    assert err.max() <= 1e-3
    # nothing reaches 99 %: no model without find_best ...
    r = mvg.ransac_fitter(x0, x1, options=dict(opts2, required_percent_inliers=0.99, maximum_tries=64))
    assert r['success'] is False and r['inlier_percent'] == 0.0
    assert r['essential'].size == 0 and r['inlier_idx'].size == 0
    assert np.array_equal(r['camera'], P0)
    # ... and the best one with it
    r = mvg.ransac_fitter(x0, x1, options=dict(opts2, required_percent_inliers=0.99, maximum_tries=64,
                                               find_best_even_in_failure=True))
    assert r['success'] is False and 0 < r['inlier_percent'] <= 0.8 and r['essential'].shape == (3, 3)
    assert r['inlier_idx'].shape == (int(round(r['inlier_percent'] * 400)), 1)


def test_ransac_fit_argument_errors():
    from spectavi_amd import mvg, _lib
    rng = np.random.default_rng(3)
    x0, x1, _, _ = mc.two_view_scene(rng, npt=20)
    with pytest.raises(ValueError):
        mvg.ransac_fitter(x0[:9], x1[:9])
    with pytest.raises(TypeError):
        mvg.ransac_fitter(x0, x1[:19])
    bad = np.full((3, 7), 20, np.int32)  # row 20 of 20 correspondences
    with pytest.raises(Exception):
        mvg.ransac_fit(x0, x1, samples=bad)
    with pytest.raises(Exception):
        mvg.ransac_fit(x0, x1, samples=-bad)
    # the C symbol itself refuses fewer than 10 correspondences (the reference's constructor throws)
    ok = ct.c_bool(True)
    pct = ct.c_double(1.0)
    from spectavi_amd.ndarray import NdArray
    e, c, i = NdArray(), NdArray(), NdArray(dtype='int32')
    mvg._ransac_fitter(x0[:9].copy(), x1[:9].copy(), 9, .9, .5, 10, True, 3e-2, False, ct.byref(ok), ct.byref(e),
                       ct.byref(c), ct.byref(pct), ct.byref(i))
    assert _lib.clib.spv_last_status() != 0 and ok.value is False
    # zero tries: no model
    r = mvg.ransac_fit(x0, x1, maximum_tries=0, seed=1)
    assert not r['success'] and r['best_try'] == -1 and r['tries_run'] == 0
