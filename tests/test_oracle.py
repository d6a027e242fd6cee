"""CPU-only: pins the oracle (oracle/) against the reference's own test properties,
the independent numpy statements and the committed golden fixtures."""
import numpy as np
import pytest

from tests.conftest import uniform_u8


def test_l1k2_reference_test_property(oracle):
    """reference test/test_feature.py:102-121: 200x144 uniform uint8, distances equal
    the numpy brute-force L1 exactly (the reference compares distances only)."""
    np.random.seed(0xdeadbeef)
    x = np.random.uniform(low=0, high=256, size=(200, 144)).astype('uint8')
    y = np.random.uniform(low=0, high=256, size=(200, 144)).astype('uint8')
    _, nnd = oracle.nn_bruteforcel1k2(x, y)
    d = np.abs(x.astype(np.int32)[:, None, :] - y.astype(np.int32)[None, :, :]).sum(-1)
    gt = np.sort(d, axis=0)[:2].T
    assert np.sum(np.abs(gt - nnd) > 0) == 0


@pytest.mark.parametrize("seed,m,n,dim,hi", [(1, 64, 33, 16, 256), (2, 300, 200, 64, 2), (3, 1, 5, 32, 256),
                                            (4, 2, 5, 128, 4), (5, 513, 129, 128, 256)])
def test_l1k2_matches_numpy_indices(oracle, seed, m, n, dim, hi):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, hi, (m, dim), dtype=np.uint8)
    y = rng.integers(0, hi, (n, dim), dtype=np.uint8)
    idx, dist = oracle.nn_bruteforcel1k2(x, y, nthreads=4)
    nidx, ndist = oracle.numpy_l1_top2(x, y)
    assert np.array_equal(dist, ndist)
    assert np.array_equal(idx, nidx)  # lexicographic (dist, idx), lower index on ties


def test_l1k2_hand_checked_ties(oracle):
    """Streaming strict-< update (reference src/BruteForceNnL1K2.h:129-139) worked by
    hand on distance sequences [5,3,3,5], [3,5,5], [5,5,3]."""
    def rows(ds):
        x = np.zeros((len(ds), 16), np.uint8)
        x[:, 0] = ds
        return x
    y = np.zeros((1, 16), np.uint8)
    for ds, want_i, want_d in [([5, 3, 3, 5], [1, 2], [3, 3]), ([3, 5, 5], [0, 1], [3, 5]),
                               ([5, 5, 3], [2, 0], [3, 5])]:
        idx, dist = oracle.nn_bruteforcel1k2(rows(ds), y)
        assert idx[0].tolist() == want_i and dist[0].tolist() == want_d


def test_l1k2_sentinels_and_threads(oracle):
    y = uniform_u8(1, 7, 32)
    idx, dist = oracle.nn_bruteforcel1k2(np.zeros((0, 32), np.uint8), y)
    assert np.all(idx == np.iinfo(np.uint64).max) and np.all(dist == np.iinfo(np.int32).max)
    x = uniform_u8(2, 1, 32)
    idx, dist = oracle.nn_bruteforcel1k2(x, y)
    assert np.all(idx[:, 0] == 0) and np.all(idx[:, 1] == np.iinfo(np.uint64).max)
    assert np.all(dist[:, 1] == np.iinfo(np.int32).max)
    x = uniform_u8(3, 400, 128)
    y = uniform_u8(4, 300, 128)
    a = oracle.nn_bruteforcel1k2(x, y, nthreads=1)
    b = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(ValueError):
        oracle.nn_bruteforcel1k2(np.zeros((4, 24), np.uint8), np.zeros((4, 24), np.uint8))


@pytest.mark.parametrize("name", ["l1k2_200x144.npz", "l1k2_1kx1k_128.npz", "l1k2_ties_300x500_64.npz",
                                  "l1k2_dups_257x5_128.npz", "l1k2_m0.npz", "l1k2_m1.npz", "l1k2_m2.npz"])
def test_l1k2_golden(oracle, golden, name):
    g = golden(name)
    idx, dist = oracle.nn_bruteforcel1k2(g["x"], g["y"], nthreads=4)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])


def test_cascade_reference_test_bound(oracle, golden):
    """reference test/test_feature.py:123-151: 200x144 randn -> normalize, m=8 n=16 g=5,
    at most 2*round(.4*200) of the 400 index entries differ from exact L1."""
    from oracle.oracle import numpy_l1_top2
    g = golden("cascade_ref_test_inputs.npz")

    def normalize(x):  # reference spectavi/feature.py:384-407 restated in numpy
        x0 = x - np.mean(x, axis=0, keepdims=True)
        norm = np.max(np.stack([np.max(x0, 0, keepdims=True), -np.min(x0, 0, keepdims=True)]), axis=0)
        x0 = np.clip(np.round(x0 / norm * 128), -128, 127)
        return x0.astype(np.float32)
    x, y = normalize(g["x"]), normalize(g["y"])
    idx, dist, ncand, nset = oracle.nn_cascading_hash(x, y, 8, 16, 5, g["dict"])
    gt_idx, gt_dist = numpy_l1_top2((x + 128).astype(np.uint8), (y + 128).astype(np.uint8))
    assert np.sum(idx != gt_idx) <= 2 * round(.4 * 200)
    # every returned pair is a true L1 distance of that pair
    xi, yi = x.astype(np.int32), y.astype(np.int32)
    ok = idx != np.iinfo(np.uint64).max
    for q in range(200):
        for c in range(2):
            if ok[q, c]:
                assert dist[q, c] == np.abs(xi[int(idx[q, c])] - yi[q]).sum()


def test_cascade_candidates_closed_form(oracle, golden):
    """The bucket/multi-probe restatement (src/CascadingHashNn.h:150-227) equals the closed
    form ((xcode ^ ysign) & ~ymask) == 0, and the result is the lexicographic top-2 over it."""
    g = golden("cascade_2kx2k_m8n2g2.npz")
    x, y = g["x"].astype(np.float32), g["y"].astype(np.float32)
    m, n, gg = int(g["m"]), int(g["n"]), int(g["g"])
    idx, dist, ncand, nset, xcodes, ysign, ymask = oracle.nn_cascading_hash(x, y, m, n, gg, g["dict"], debug=True)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])
    assert np.array_equal(xcodes, g["xcodes"]) and np.array_equal(ymask, g["ymask"])
    cand = oracle.candidates_from_codes(xcodes, ysign, ymask)
    assert np.array_equal(cand.sum(1), nset)
    assert np.all([bin(int(v)).count("1") == gg for v in ymask.ravel()])
    ux, uy = (x + 128).astype(np.int32), (y + 128).astype(np.int32)
    for q in range(0, 2000, 37):
        ks = np.flatnonzero(cand[q])
        d = np.abs(ux[ks] - uy[q]).sum(1)
        order = np.lexsort((ks, d))[:2]
        want_i = [int(ks[o]) for o in order] + [np.iinfo(np.uint64).max] * (2 - len(order))
        want_d = [float(d[o]) for o in order] + [2147483648.0] * (2 - len(order))
        assert idx[q].tolist() == want_i and dist[q].tolist() == want_d


def _camera_pairs(seed=77, n=30, npt=2003):
    """The camera pairs of tests/test_dlt_gpu.py::test_many_camera_pairs: random 3x4 pairs, every fifth
    with almost no baseline, every seventh with a translation column scaled by 1e4; odd ones noisy."""
    rng = np.random.default_rng(seed)
    for k in range(n):
        P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
        if k % 5 == 0:
            P1 = P0 + 1e-6 * rng.standard_normal((3, 4))
        if k % 7 == 0:
            P0[:, 3] *= 1e4
        Xw = rng.standard_normal((npt, 4))
        x = Xw @ P0.T + (k % 2) * rng.normal(0, 1e-3, (npt, 3))
        xp = Xw @ P1.T + (k % 2) * rng.normal(0, 1e-3, (npt, 3))
        yield k, P0, P1, x, xp


def test_jacobisvd_restatement_matches_lapack(oracle):
    """oracle_jacobisvd.cpp (Eigen's two-sided JacobiSVD restated) against numpy.linalg.svd on 3x3
    and 4x4 matrices incl. nearly singular, badly scaled, rank-deficient and zero ones."""
    rng = np.random.default_rng(5)
    for n in (3, 4):
        for k in range(3000):
            A = rng.standard_normal((n, n))
            if k % 5 == 0:
                A[:, 0] = A[:, 1] * (1 + 1e-9 * rng.standard_normal())
            if k % 7 == 0:
                A *= 10.0 ** rng.integers(-8, 9)
            if k % 11 == 0:
                A[n - 1] = 0
            if k == 1:
                A[:] = 0
            U, S, V, sweeps = oracle.jacobisvd(A)
            s = np.linalg.svd(A, compute_uv=False)
            scale = max(s[0], 1e-300)
            assert np.all(np.diff(S) <= 0) and np.all(S >= 0) and 1 <= sweeps <= 12
            assert np.max(np.abs(S - s)) <= 1e-14 * scale
            assert np.max(np.abs(U @ np.diag(S) @ V.T - A)) <= 1e-14 * scale
            assert np.max(np.abs(U.T @ U - np.eye(n))) <= 1e-14 and np.max(np.abs(V.T @ V - np.eye(n))) <= 1e-14


def test_dlt_reference_test_properties(oracle):
    """reference test/test_mvg.py:94-125 on randn cameras and points, on the oracle (JacobiSVD
    restatement) and on the host mirror of the HIP kernel's operation sequence."""
    rng = np.random.default_rng(0xdeadbeef)
    for _ in range(100):
        P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
        X0 = rng.standard_normal(4)
        x, xp = P0 @ X0, P1 @ X0
        for tri, rep, canonical in ((oracle.dlt_triangulate, oracle.dlt_reprojection_error, False),
                                    (oracle.dlt_mirror_triangulate, oracle.dlt_mirror_reprojection_error, True)):
            err = rep(P0, P1, x, xp)
            assert abs(err[0, 0]) < 1e-3
            X = tri(P0, P1, x, xp)[0]
            assert np.allclose(X / X[3], X0 / X0[3])
            assert np.allclose(np.cross(P0 @ X, x), 0, atol=1e-8)
            assert abs(np.linalg.norm(X) - 1) < 1e-12 and (X[3] >= 0 or not canonical)


def test_dlt_matches_lapack_and_golden(oracle, golden):
    g = golden("dlt_1000.npz")
    X = oracle.dlt_triangulate(g["P0"], g["P1"], g["x"], g["xp"])
    assert np.array_equal(X, g["X"])
    assert np.max(np.abs(oracle.canonical_sign(X) - g["X_lapack"])) < 1e-12
    err = oracle.dlt_reprojection_error(g["P0"], g["P1"], g["x"], g["xp"])
    assert np.array_equal(err, g["err"])
    assert np.all(err[:500] < 1e-9) and np.all(err[500:] < 1e-1)
    assert np.all(oracle.dlt_cheirality(g["P0"], g["P1"], g["x"], g["xp"]))
    Xm = oracle.dlt_mirror_triangulate(g["P0"], g["P1"], g["x"], g["xp"])
    assert np.array_equal(Xm, g["X_mirror"]) and np.max(np.abs(Xm - g["X_lapack"])) < 1e-12
    assert np.array_equal(oracle.dlt_mirror_reprojection_error(g["P0"], g["P1"], g["x"], g["xp"]), g["err_mirror"])
    assert np.all(oracle.dlt_mirror_cheirality(g["P0"], g["P1"], g["x"], g["xp"]))


def test_dlt_definition_on_ill_conditioned_pairs(oracle):
    """The reference's definition of X (smallest right singular vector of A, LAPACK as the solver,
    tests/dlt_checks.py) on 30 camera pairs incl. near-zero baseline (sigma3 tiny) and 1e4-scaled
    translations: for the oracle, and for the host mirror of the HIP fast path (Gram-Schmidt +
    inverse iteration, no SVD), which the kernel reproduces bit for bit on the GPU."""
    from tests import dlt_checks as dc
    for k, P0, P1, x, xp in _camera_pairs():
        oX = oracle.dlt_triangulate(P0, P1, x, xp)
        dc.check_definition(oX, P0, P1, x, xp, err=oracle.dlt_reprojection_error(P0, P1, x, xp), what="oracle pair %d" % k,
                            unfused=True)
        mX = oracle.dlt_mirror_triangulate(P0, P1, x, xp)
        dc.check_definition(mX, P0, P1, x, xp, err=oracle.dlt_mirror_reprojection_error(P0, P1, x, xp),
                            what="mirror pair %d" % k)
        dc.check_against_oracle(mX, oX, P0, P1, x, xp, what="mirror pair %d" % k)


def test_dlt_scoring_mirror_agrees_with_oracle(oracle):
    """RANSAC scoring (src/RansacFitter.h:59-73): the mirror's inlier decisions equal the oracle's
    (JacobiSVD per point and hypothesis) except where the oracle's own reprojection error sits
    within 1e-9 relative of the threshold."""
    rng = np.random.default_rng(12)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    t = rng.standard_normal((3, 1))
    npt = 1500
    Xw = np.hstack([rng.standard_normal((npt, 2)), rng.uniform(4, 8, (npt, 1)), np.ones((npt, 1))])
    x, xp = Xw @ P0.T, Xw @ np.hstack([R, t]).T
    xp[:, :2] += rng.normal(0, 2e-3, (npt, 2)) * xp[:, 2:3]
    P1s = np.stack([np.hstack([R, t]), np.hstack([R, -t])] + [rng.standard_normal((3, 4)) for _ in range(6)])
    for thr in (1e-3, 1e-2, 0.5):
        oc, om, oe = oracle.dlt_score_hypotheses(P0, P1s, x, xp, thr, return_err=True)
        mc, mm = oracle.dlt_mirror_score_hypotheses(P0, P1s, x, xp, thr)
        clear = np.abs(oe - thr) > 1e-9 * thr
        assert np.array_equal(om[clear], mm[clear])
        assert np.all(np.abs(oc - mc) <= (~clear).sum(1))


def test_dlt_fuzz_slice_on_the_mirror(oracle):
    """The GPU fuzz's DLT and scoring cases (tests/fuzz_gpu.py: nearly identical cameras, points at
    infinity, 1e6 homogeneous scale, inconsistent pairs, noise up to 10, inf/nan observations) with
    the mirror standing in for the kernel it mirrors: definition (LAPACK) + oracle on every case."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(os.path.dirname(__file__), "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    try:
        n = fz.fuzz_dlt(4, 4.0, impl=(oracle.dlt_mirror_triangulate, oracle.dlt_mirror_reprojection_error))
        m = fz.fuzz_score(4, 3.0, impl=oracle.dlt_mirror_score_hypotheses)
    except SystemExit as e:
        pytest.fail(str(e))
    assert n >= 20 and m >= 10


def _small_rotation(rng, max_angle=0.3):
    """Rotation by at most max_angle about a random axis (Rodrigues): keeps a scene in front of both cameras."""
    a = rng.standard_normal(3)
    a /= np.linalg.norm(a)
    th = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def test_essential_to_cameras_and_candidate_processing_oracle(oracle):
    """The oracle for SURVEY 8(f) row 1 (oracle_jacobisvd.cpp: src/Camera.h:31-46,
    src/RansacFitter.h:42-95) pinned by LAPACK and by the geometry it must satisfy: the four
    cameras of an essential matrix are (R, +-t) pairs with R orthogonal, t its unit left null vector
    and [t]x R = +-E; on noise-free correspondences of a known motion the true camera (as a
    projective camera: +-[R | t]) wins with every point an inlier; matrices whose two largest
    singular values differ by more than the gate are rejected with the LAPACK ratio."""
    rng = np.random.default_rng(3)
    for _ in range(50):
        R = _small_rotation(rng)
        t = rng.standard_normal(3)
        t /= np.linalg.norm(t)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R
        cams = oracle.essential_to_cameras(E)
        for k in range(4):
            Rk, tk = cams[k][:, :3], cams[k][:, 3]
            assert np.allclose(Rk.T @ Rk, np.eye(3), atol=1e-12) and abs(abs(tk @ t) - 1) < 1e-12
            tkx = np.array([[0, -tk[2], tk[1]], [tk[2], 0, -tk[0]], [-tk[1], tk[0], 0]])
            assert min(np.abs(tkx @ Rk - E).max(), np.abs(tkx @ Rk + E).max()) < 1e-12
        assert np.allclose(cams[0][:, :3], cams[1][:, :3]) and np.allclose(cams[0][:, 3], -cams[1][:, 3])
        assert np.allclose(cams[2][:, :3], cams[3][:, :3]) and np.allclose(cams[2][:, 3], -cams[3][:, 3])
        npt = 120
        Xw = np.hstack([rng.standard_normal((npt, 2)), rng.uniform(4, 8, (npt, 1)), np.ones((npt, 1))])
        P1 = np.hstack([R, t[:, None]])
        x0, x1 = Xw[:, :3].copy(), Xw @ P1.T
        r = oracle.process_fundamental_matrix(E * rng.uniform(0.2, 5) * rng.choice([-1, 1]), x0, x1, 3e-2, .9, 1e-3, False)
        assert r["success"] and r["inlier_count"] == npt and r["gate_ratio"] < 1e-12
        assert min(np.abs(r["best_P"] - P1).max(), np.abs(r["best_P"] + P1).max()) < 1e-9
        assert np.array_equal(r["inlier_idx"], np.arange(npt)) and sorted(r["counts4"])[:3] == [0, 0, 0]
        assert np.allclose(np.linalg.svd(r["E"], compute_uv=False), [1, 1, 0], atol=1e-12)
        F = rng.standard_normal((3, 3))
        s = np.linalg.svd(F, compute_uv=False)
        g = oracle.process_fundamental_matrix(F, x0, x1, 3e-2, .9, 1e-3, True)
        assert abs(g["gate_ratio"] - abs(s[0] - s[1]) / (abs(s[0] + s[1]) / 2)) < 1e-12
        assert g["success"] == (False if g["gate_ratio"] > 3e-2 else g["success"]) and (g["gate_ratio"] <= 3e-2 or (g["counts4"] == -1).all())


def test_seven_point_oracle_reference_properties_and_numpy(oracle):
    """Pins oracle/oracle_ransac.cpp::seven_point: the reference's own test properties
    (test/test_mvg.py:127-160) and the independent LAPACK + numpy.roots statement."""
    from tests import mvg_checks as mc
    rng = np.random.default_rng(0x7b7)
    for it in range(300):
        x, xp = rng.standard_normal((7, 3)), rng.standard_normal((7, 3))
        xe, xpe = x[:, :2] / x[:, 2:], xp[:, :2] / xp[:, 2:]
        Fs, basis = oracle.seven_point(xe, xpe, return_basis=True)
        assert len(Fs) in (0, 1, 2, 3)
        # the basis is orthonormal and lies in the null space of the 7 x 9 system
        B = basis.reshape(2, 9)
        A = mc.seven_point_rows(xe, xpe)
        assert np.abs(B @ B.T - np.eye(2)).max() < 1e-13
        assert np.abs(A @ B.T).max() <= 1e-12 * np.abs(A).max()
        scale = max(1.0, np.abs(A).max())
        for F in Fs:
            assert mc.epipolar_residual(F, xe, xpe) < 1e-10 * scale  # test_mvg.py:139-141 (there unscaled)
        mc.check_seven_point(Fs, xe, xpe, "oracle case %d" % it, basis=basis)
    # reconstruction, test_mvg.py:143-160
    for it in range(100):
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
        P1 = rng.standard_normal((3, 4))
        F0 = mc.skew(P1[:, 3]) @ P1 @ (P0.T @ np.linalg.inv(P0 @ P0.T))
        X = rng.standard_normal((7, 4))
        x0, x1 = X @ P0.T, X @ P1.T
        Fs = oracle.seven_point(x0[:, :2] / x0[:, 2:], x1[:, :2] / x1[:, 2:])
        assert any(np.std(F / F0) < 1e-8 for F in Fs), it


def test_ransac_fit_oracle(oracle):
    """oracle_ransac_fit: the reference's own RANSAC test (test/test_mvg.py:38-91: success, essential
    recovered to std(rE / E) < 1e-2 with the reference's options), then a contaminated scene: the
    inliers are the uncontaminated correspondences, the winning subset is clean, and the serial
    update rule holds (nothing after the first success is looked at)."""
    from tests import mvg_checks as mc
    rng = np.random.default_rng(0xdeadbeef)
    x0, x1, E = mc.reference_ransac_scene(rng)
    samples = np.stack([rng.choice(np.arange(1, 200), 7, replace=False) for _ in range(200)]).astype(np.int32)
    r = oracle.ransac_fit(x0, x1, samples, required_percent_inliers=0.9, reprojection_error_allowed=0.5,
                          find_best_even_in_failure=False, singular_value_ratio_allowed=3e-2)
    assert r["success"] and mc.essential_agrees(r["essential"], E) < 1e-2

    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=200, outlier_fraction=0.25)
    kw = dict(reprojection_error_allowed=1e-3, find_best_even_in_failure=False)
    r = oracle.ransac_fit(x0, x1, samples[:80], required_percent_inliers=0.7, **kw)
    assert r["success"] and r["best_try"] >= 0
    assert mc.essential_agrees(r["essential"], E) < 1e-6
    assert r["inlier_percent"] == len(r["inlier_idx"]) / 200.0 > 0.7
    assert np.array_equal(r["inlier_idx"], np.setdiff1d(np.arange(200), out_idx))
    assert not np.intersect1d(samples[r["best_try"]], out_idx).size
    r2 = oracle.ransac_fit(x0, x1, samples[:r["best_try"] + 1], required_percent_inliers=0.7, **kw)
    assert r2["best_try"] == r["best_try"] and r2["best_root"] == r["best_root"]
    assert np.array_equal(r2["inlier_idx"], r["inlier_idx"])
    # an unreachable requirement: nothing is kept without find_best, the best is kept with it
    r3 = oracle.ransac_fit(x0, x1, samples[:40], required_percent_inliers=0.99, **kw)
    assert not r3["success"] and r3["best_try"] == -1 and r3["essential"] is None and r3["inlier_percent"] == 0
    kw["find_best_even_in_failure"] = True
    r4 = oracle.ransac_fit(x0, x1, samples[:40], required_percent_inliers=0.99, **kw)
    assert not r4["success"] and r4["best_try"] >= 0 and 0 < r4["inlier_percent"] <= 0.75


def test_seven_point_reference_cubic_loses_digits_when_its_leading_coefficient_is_tiny(oracle):
    """Found by tests/fuzz_gpu.py (seed 301, case 2191, item 30): in the null-space pair the SVD happens to
    return, det(F0 - F1) = -7e-11 against coefficients of 6e-3.  The reference divides by it and solves
    the monic cubic trigonometrically (src/FundamentalMatrixFitter.h:64-104, :229-233): one root is 4e7,
    acos is taken at -1 + 4e-15, and the two ordinary roots come out 8e-3 off.  Every returned F still
    satisfies the epipolar constraints to rounding (any member of the pencil does: the reference's own
    test, test/test_mvg.py:127-141, cannot see this), but two of the three are not singular: det = 1e-6
    of a unit-norm F instead of 1e-16.  The oracle restates the reference's arithmetic and the device
    follows the oracle, so parity means reproducing this; the independent numpy check is skipped for
    such pairs (mvg_checks.reference_cubic_is_ill_conditioned)."""
    from tests import mvg_checks as mc
    rng = np.random.default_rng([301, 6, 2191])
    nb = int(rng.choice([1, 63, 64, 65, 200]))
    kind = int(rng.integers(0, 4))
    assert (nb, kind) == (63, 3)
    for item in range(31):
        x0, x1, _, _ = mc.two_view_scene(rng, npt=7, noise=10.0 ** rng.uniform(-8, -2))
    xe, xpe = x0[:, :2] / x0[:, 2:], x1[:, :2] / x1[:, 2:]
    Fs, basis = oracle.seven_point(xe, xpe, return_basis=True)
    a, b, c, d = np.abs(mc.cubic_of_basis(basis))
    assert a < 1e-9 < 1e-3 < max(b, c, d) and mc.reference_cubic_is_ill_conditioned(basis)
    assert len(Fs) == 3
    dets = sorted(abs(np.linalg.det(F / np.linalg.norm(F))) for F in Fs)
    for F in Fs:
        assert mc.epipolar_residual(F / np.linalg.norm(F), xe, xpe) < 1e-15
    assert dets[0] < 1e-15 and dets[1] > 1e-7          # one root accurate, two visibly not singular
    ref, margin, lead = mc.numpy_seven_point(xe, xpe)   # the well-conditioned statement of the same problem
    assert len(ref) == 3 and all(abs(np.linalg.det(G)) < 1e-14 for G in ref)
    worst = max(1 - max(mc.parallel(F, G) for G in ref) for F in Fs)
    assert 1e-7 < worst < 1e-3
