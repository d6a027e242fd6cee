"""Randomised differential run on the GPU box: the C-ABI host paths against the CPU oracle on
many random shapes / parameters per path, for a time budget.  Prints one line per path and
exits non-zero on the first mismatch (with the failing configuration).

    python tests/fuzz_gpu.py --seconds 60 --seed 1
    python tests/fuzz_gpu.py --only cascade --case 17 --seed 1     # re-run one reported case
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from oracle import oracle as o  # noqa: E402
from spectavi_amd import feature, mvg  # noqa: E402


def fuzz_l1k2(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 0, n])  # every case is reproducible on its own
        dim = int(rng.choice([16, 32, 48, 64, 96, 128, 144, 192, 256, 272, 512, 1024]))
        xrows = int(rng.choice([0, 1, 2, 3, rng.integers(4, 300), rng.integers(300, 6000), rng.integers(60000, 70000)]))
        yrows = int(rng.choice([1, 2, rng.integers(3, 200), rng.integers(200, 3000)]))
        if dim > 256:
            xrows, yrows = min(xrows, 3000), min(yrows, 500)
        hi = int(rng.choice([2, 3, 16, 256]))        # small alphabets force distance ties
        x = rng.integers(0, hi, (xrows, dim), dtype=np.uint8)
        y = rng.integers(0, hi, (yrows, dim), dtype=np.uint8)
        if xrows and rng.random() < 0.3:               # exact duplicates of database rows
            k = min(yrows, 8)
            y[:k] = x[rng.integers(0, xrows, k)]
        idx, dist = feature.nn_bruteforcel1k2(x, y)
        oi, od = o.nn_bruteforcel1k2(x, y, nthreads=o.max_threads())
        if not (np.array_equal(idx, oi) and np.array_equal(dist, od)):
            raise SystemExit("L1K2 MISMATCH case=%d xrows=%d yrows=%d dim=%d hi=%d" % (n, xrows, yrows, dim, hi))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_cascade(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 1, n])  # every case is reproducible on its own
        dim = int(rng.choice([16, 32, 64, 128, 144, 256, 128, 128, 400, 512, 1040, 2048]))
        m = int(rng.choice([1, 2, 3, 4, 6, 8, 11, 13, 17, 20, 22, 23, 27, 31]))
        nt = int(rng.choice([1, 2, 3, 4, 7, 16]))
        g = int(rng.integers(0, min(m, 6) + 1))
        xrows = int(rng.choice([0, 1, 2, rng.integers(3, 200), rng.integers(200, 5000)]))
        yrows = int(rng.choice([1, 2, rng.integers(3, 100), rng.integers(100, 1500)]))
        span = int(rng.choice([2, 20, 128]))           # narrow ranges: many zero projections / ties
        x = rng.integers(-span, span, (xrows, dim)).astype(np.float32)
        y = rng.integers(-span, span, (yrows, dim)).astype(np.float32)
        if xrows and rng.random() < 0.5:
            k = min(yrows, max(1, xrows // 2))
            y[:k] = np.clip(x[rng.integers(0, xrows, k)] + rng.integers(-2, 3, (k, dim)), -128, 127)
        d = rng.standard_normal((nt, dim, m)).astype(np.float32)
        if rng.random() < 0.2:
            d = np.round(d)                            # integer hyperplanes: exact-zero projections
        # the probe's query order: library default (input order at these sizes), forced per-table
        # sign-code order, forced input order -- read per call, same results
        mode = (None, "1", "0")[n % 3]
        if mode is None:
            os.environ.pop("SPECTAVI_CASCADE_SORT", None)
        else:
            os.environ["SPECTAVI_CASCADE_SORT"] = mode
        idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
        os.environ.pop("SPECTAVI_CASCADE_SORT", None)
        oi, od, onc, _ = o.nn_cascading_hash(x, y, m, nt, g, d)
        if not (np.array_equal(ncand, onc) and np.array_equal(dist, od) and np.array_equal(idx, oi)):
            bad = np.flatnonzero((ncand != onc) | (dist != od).any(1) | (idx != oi).any(1))
            raise SystemExit("CASCADE MISMATCH case=%d xrows=%d yrows=%d dim=%d m=%d n=%d g=%d span=%d; %d queries differ, first %s: "
                             "gpu idx %s dist %s ncand %d, oracle idx %s dist %s ncand %d" %
                             (n, xrows, yrows, dim, m, nt, g, span, len(bad), bad[:5], idx[bad[0]], dist[bad[0]], ncand[bad[0]],
                              oi[bad[0]], od[bad[0]], onc[bad[0]]))
        n += 1
        if only_case is not None:
            break
    return n


def dlt_case(seed, n):
    """Case n of the DLT fuzz: (P0, P1, x, xp, kind, noise), reproducible on its own."""
    rng = np.random.default_rng([seed, 2, n])
    npt = int(rng.choice([1, 2, 63, 64, 65, 255, 257, rng.integers(300, 20000)]))
    kind = int(rng.integers(0, 6))
    P0 = rng.standard_normal((3, 4))
    P1 = rng.standard_normal((3, 4))
    if kind == 1:
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    if kind == 2:
        P1 = P0 + 1e-9 * rng.standard_normal((3, 4))       # nearly identical cameras
    Xw = rng.standard_normal((npt, 4))
    if kind == 3:
        Xw[:, 3] = 0.0                                       # points at infinity
    x, xp = Xw @ P0.T, Xw @ P1.T
    noise = float(rng.choice([0.0, 1e-6, 1e-3, 0.1, 10.0]))
    x[:, :2] += noise * rng.standard_normal((npt, 2))
    xp[:, :2] += noise * rng.standard_normal((npt, 2))
    if kind == 4:
        x *= 1e6                                             # large homogeneous scale
    if kind == 5:
        x[: max(1, npt // 10)] = xp[: max(1, npt // 10)]    # inconsistent pairs
    if rng.random() < 0.15:                                  # w = 0 / non-finite observations: inf and nan
        x[rng.integers(0, npt), 2] = 0.0                     # must come out in the same places on both sides
        xp[rng.integers(0, npt), 2] = 0.0
        x[rng.integers(0, npt), 0] = np.inf if rng.random() < 0.5 else np.nan
    return P0, P1, x, xp, kind, noise


def fuzz_dlt(seed, budget, only_case=None, impl=None):
    """Every case three ways: (a) against the host mirror of the kernel's operation sequence, to a
    few ulps times the conditioning (dlt_checks.check_against_mirror -- determinism: inf/nan in the
    same rows, no lane- or shape-dependent path); (b) the
    reference's DEFINITION with LAPACK as the solver (tests/dlt_checks.py: residual <= sigma4,
    direction where the gap allows, reprojection error within 1e-6 relative); (c) up to sign
    against the oracle (oracle_jacobisvd.cpp, the reference's JacobiSVD arithmetic restated).
    `impl` = (triangulate, reprojection_error) replaces the HIP entry points (the CPU suite runs the
    same cases on the mirror itself)."""
    from tests import dlt_checks as dc
    tri, rep = impl or (mvg.dlt_triangulate, mvg.dlt_reprojection_error)
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        P0, P1, x, xp, kind, noise = dlt_case(seed, n)
        npt = x.shape[0]
        X = tri(P0, P1, x, xp)
        E = rep(P0, P1, x, xp)
        mX, mE = o.dlt_mirror_triangulate(P0, P1, x, xp), o.dlt_mirror_reprojection_error(P0, P1, x, xp)
        try:
            dc.check_against_mirror(X, mX, P0, P1, x, xp, E=E, mE=mE, what="case %d" % n)
        except AssertionError as e:
            raise SystemExit("DLT MISMATCH vs mirror case=%d npt=%d kind=%d noise=%g: %s" % (n, npt, kind, noise, e))
        try:
            dc.check_definition(X, P0, P1, x, xp, err=E, what="case %d" % n)
            dc.check_against_oracle(X, o.dlt_triangulate(P0, P1, x, xp), P0, P1, x, xp, what="case %d" % n)
        except AssertionError as e:
            raise SystemExit("DLT DEFINITION FAILURE npt=%d kind=%d noise=%g: %s" % (npt, kind, noise, e))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_ratio(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 3, n])
        yrows = int(rng.choice([1, 2, 255, 256, 257, rng.integers(3, 5000), rng.integers(5000, 300000)]))
        idx = rng.integers(0, 1 << 20, (yrows, 2)).astype(np.uint64)
        is_float = bool(rng.integers(0, 2))
        d0 = rng.integers(0, int(rng.choice([2, 50, 32641])), yrows)
        d1 = d0 + rng.integers(0, int(rng.choice([1, 3, 20000])), yrows)
        dist = np.stack([d0, d1], axis=1)
        none = rng.random(yrows) < 0.05                 # no neighbour at all / only one
        one = rng.random(yrows) < 0.05
        idx[none] = np.iinfo(np.uint64).max
        idx[one, 1] = np.iinfo(np.uint64).max
        if is_float:
            dist = dist.astype(np.float32)
            dist[none] = 2147483648.0
            dist[one, 1] = 2147483648.0
        else:
            dist = dist.astype(np.int32)
            dist[none] = np.iinfo(np.int32).max
            dist[one, 1] = np.iinfo(np.int32).max
        ratio = float(rng.choice([0.0, 1.0, 1.25, 1.75, 3.0, 1e9]))
        got = feature.ratio_test_matches(idx, dist, ratio)
        want = o.ratio_test_matches(idx, dist, ratio)
        if not np.array_equal(got, want):
            raise SystemExit("RATIO MISMATCH case=%d yrows=%d float=%d ratio=%g got %d want %d rows" %
                             (n, yrows, is_float, ratio, len(got), len(want)))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_score(seed, budget, only_case=None, impl=None):
    score = impl or (lambda *a: mvg.dlt_score_hypotheses(*a, return_mask=True))
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 4, n])
        npt = int(rng.choice([1, 63, 64, 65, 256, 257, rng.integers(2, 3000)]))
        nhyp = int(rng.choice([1, 2, 4, rng.integers(5, 200)]))
        P0 = np.hstack([np.eye(3), np.zeros((3, 1))]) if rng.random() < 0.5 else rng.standard_normal((3, 4))
        R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        t = rng.standard_normal((3, 1))
        Ptrue = np.hstack([R, t]) if rng.random() < 0.7 else rng.standard_normal((3, 4))
        Xw = rng.standard_normal((npt, 4))
        Xw[:, 2] += 4.0
        Xw[:, 3] = 1.0
        x, xp = Xw @ P0.T, Xw @ Ptrue.T
        x[:, :2] += 1e-3 * rng.standard_normal((npt, 2))
        P1s = rng.standard_normal((nhyp, 3, 4))
        P1s[0] = Ptrue
        if nhyp > 1:
            P1s[1] = np.hstack([R, -t])                 # behind-the-camera twin
        thr = float(rng.choice([1e-4, 1e-2, 0.5, 1e3]))
        c, mk = score(P0, P1s, x, xp, thr)
        mk = np.asarray(mk, bool)
        mc, mmk = o.dlt_mirror_score_hypotheses(P0, P1s, x, xp, thr)
        # the oracle (JacobiSVD per point and hypothesis): same decisions wherever its own error is
        # not within 1e-9 relative of the threshold and the solve is well separated enough for the
        # cheirality sign to be determined (|error| finite); the host mirror of the kernel's own
        # operation sequence (a few ulps from the kernel) is held to the same rule
        oc, omk, oe = o.dlt_score_hypotheses(P0, P1s, x, xp, thr, return_err=True)
        clear = np.isfinite(oe) & (np.abs(oe - thr) > 1e-9 * thr)
        if not np.array_equal(mk[clear], np.asarray(mmk, bool)[clear]):
            hh, pp = np.nonzero((mk != np.asarray(mmk, bool)) & clear)
            raise SystemExit("SCORE MISMATCH vs mirror case=%d npt=%d nhyp=%d thr=%g first (hyp %d, point %d) err %g" %
                             (n, npt, nhyp, thr, hh[0], pp[0], oe[hh[0], pp[0]]))
        if not np.array_equal(mk[clear], omk[clear]):
            hh, pp = np.nonzero((mk != omk) & clear)
            raise SystemExit("SCORE MISMATCH vs oracle case=%d npt=%d nhyp=%d thr=%g first (hyp %d, point %d) err %g" %
                             (n, npt, nhyp, thr, hh[0], pp[0], oe[hh[0], pp[0]]))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_normalize(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 5, n])
        rows = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 511, 512, 513, 1023, 1025, 65535, 65536, 65537, rng.integers(2, 40000),
                               rng.integers(65536, 300000)]))   # 65536 rows and up take the folded column sums
        dim = int(rng.choice([2, 4, 15, 16, 17, 128, 132, 144, rng.integers(2, 300)]))   # dim 1 is refused
        kind = int(rng.integers(0, 3))
        if kind == 0:
            x = (rng.standard_normal((rows, dim)) * rng.uniform(0.1, 100, (1, dim)) + rng.uniform(-50, 50, (1, dim)))
        elif kind == 1:
            x = rng.integers(0, 256, (rows, dim)).astype(np.float64)       # SIFT-like integer values
        else:
            x = rng.standard_normal((rows, dim)) * 1e4 + 1e6               # large offsets: rounding in the chain
        x = x.astype(np.float32)
        if rows > 1:
            x[0, 0] += 1.0                               # keep every column non-constant where possible
        with np.errstate(divide="ignore", invalid="ignore"):
            want = feature.normalize_to_ubyte_and_multiple_16_dim(x)
        got, u8 = feature.normalize_to_ubyte_and_multiple_16_dim_gpu(x, want_ubyte=True)
        ok = got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
        if ok and not np.isnan(want).any():
            ok = np.array_equal(u8, (want + 128).astype('uint8'))
        if not ok:
            raise SystemExit("NORMALIZE MISMATCH case=%d rows=%d dim=%d kind=%d" % (n, rows, dim, kind))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_seven_point(seed, budget, only_case=None):
    """Batches of random 7-subsets (heavy-tailed hnormalized Gaussians, consistent two-view subsets,
    near-degenerate ones): every F parallel to an oracle solution and a LAPACK + numpy.roots solution,
    root counts equal away from a vanishing discriminant, the null-space pair spanning the oracle's."""
    from tests import mvg_checks as mc
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 6, n])
        nb = int(rng.choice([1, 63, 64, 65, 200]))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            a, b = rng.standard_normal((nb, 7, 3)), rng.standard_normal((nb, 7, 3))
            x, xp = a[..., :2] / a[..., 2:], b[..., :2] / b[..., 2:]
        elif kind == 1:
            x, xp = rng.standard_normal((nb, 7, 2)) * 10.0 ** rng.uniform(-3, 3), rng.standard_normal((nb, 7, 2))
        else:
            xs, xps = [], []
            for _ in range(nb):
                x0, x1, _, _ = mc.two_view_scene(rng, npt=7, noise=0.0 if kind == 2 else 10.0 ** rng.uniform(-8, -2))
                xs.append(x0[:, :2] / x0[:, 2:])
                xps.append(x1[:, :2] / x1[:, 2:])
            x, xp = np.array(xs), np.array(xps)
        nroot, Fs, basis = mvg.seven_point_batch(x, xp, return_basis=True)
        for i in range(nb):
            k = int(nroot[i])
            oFs, ob = o.seven_point(x[i], xp[i], return_basis=True)
            B, OB = basis[i].reshape(2, 9), ob.reshape(2, 9)
            bad = None
            if not np.abs(B.T @ B - OB.T @ OB).max() < 1e-9:
                bad = "null-space plane differs from the oracle's"
            _, margin, lead = mc.numpy_seven_point(x[i], xp[i])
            clear = margin > 1e-6 and lead > 1e-10
            if bad is None and clear and k != len(oFs):
                bad = "%d roots, oracle %d" % (k, len(oFs))
            if bad is None and clear:
                # where the reference's cubic is ill-conditioned, the last bit of libm's acos / cos is amplified too
                tol = 1e-6 if mc.reference_cubic_is_ill_conditioned(basis[i]) else 1e-9
                for F in Fs[i, :k]:
                    if max((mc.parallel(F, G) for G in oFs), default=0.0) < 1 - tol:
                        bad = "an F is not an oracle solution"
                try:
                    mc.check_seven_point(Fs[i, :k], x[i], xp[i], "fuzz", basis=basis[i])
                except AssertionError as e:
                    bad = str(e)
            if bad:
                raise SystemExit("SEVEN-POINT MISMATCH case=%d kind=%d item=%d of %d: %s" % (n, kind, i, nb, bad))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_ransac_fit(seed, budget, only_case=None):
    """Whole fits on random contaminated scenes with the subsets handed to both sides: the winning
    try, root, inlier list and share equal the oracle's serial loop."""
    from tests import mvg_checks as mc
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 7, n])
        npt = int(rng.choice([10, 11, 64, 65, 300, rng.integers(12, 1500)]))
        frac = float(rng.choice([0.0, 0.1, 0.3, 0.5]))
        x0, x1, E, out_idx = mc.two_view_scene(rng, npt=npt, outlier_fraction=frac)
        tries = int(rng.choice([1, 7, 255, 256, 257, 700]))
        samples = mvg.ransac_sample(int(rng.integers(1, 1 << 30)), npt, tries)
        kw = dict(required_percent_inliers=float(rng.choice([0.3, 1.0 - frac - 0.05, 0.999])),
                  reprojection_error_allowed=1e-3, find_best_even_in_failure=bool(rng.integers(0, 2)),
                  singular_value_ratio_allowed=float(rng.choice([3e-2, 1e-3, 0.5])))
        oo = o.ransac_fit(x0, x1, samples, **kw)
        dd = mvg.ransac_fit(x0, x1, samples=samples, **kw)
        same = (dd['success'] == oo['success'] and dd['best_try'] == oo['best_try'] and dd['best_root'] == oo['best_root']
                and dd['inlier_percent'] == oo['inlier_percent'])
        if same and oo['best_try'] >= 0:
            same = mc.parallel(dd['essential'], oo['essential']) >= 1 - 1e-9
            cam_same = min(np.abs(dd['camera'] - oo['camera']).max(), np.abs(dd['camera'] + oo['camera']).max()) < 1e-8  # -P is the same camera
            if same and not (cam_same and np.array_equal(dd['inlier_idx'], oo['inlier_idx'])):
                # Two of the four cameras with the same inlier count (junk models with one or two inliers): which
                # of them comes first hangs on the signs of an SVD with two equal singular values, i.e. on the
                # last bit of F, where device and oracle differ through libm's cos.  Then the oracle must give the
                # device's answer when it is handed the device's F.
                r = o.process_fundamental_matrix(dd['essential'], x0, x1, kw['singular_value_ratio_allowed'],
                                                 kw['required_percent_inliers'], kw['reprojection_error_allowed'],
                                                 kw['find_best_even_in_failure'])
                same = (r['success'] and np.array_equal(r['inlier_idx'], dd['inlier_idx']) and
                        min(np.abs(dd['camera'] - r['best_P']).max(), np.abs(dd['camera'] + r['best_P']).max()) < 1e-8 and
                        sorted(r['counts4'])[-1] == sorted(r['counts4'])[-2])
        elif same:
            same = len(dd['inlier_idx']) == 0
        if not same:
            raise SystemExit("RANSAC FIT MISMATCH case=%d npt=%d outliers=%g tries=%d %r: device (try %d root %d, %d inliers) "
                             "oracle (try %d root %d, %d inliers)" % (n, npt, frac, tries, kw, dd['best_try'], dd['best_root'],
                                                                      len(dd['inlier_idx']), oo['best_try'], oo['best_root'],
                                                                      len(oo['inlier_idx'])))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_pipeline(seed, budget, only_case=None):
    """Steps 2-4 of the reference's example on random synthetic SIFT table pairs: the device-resident
    pipeline (every intermediate in HBM) against the same steps through the host front-end -- same
    matches, and with the same RANSAC seed the same model, inliers and triangulated points to rounding --
    and both against the scene (true pairs, essential matrix)."""
    import torch
    from examples import essential_from_sift_tables as ex
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 8, n])
        nc = int(rng.choice([60, 200, 1000, rng.integers(60, 4000)]))
        ne = int(rng.choice([0, 10, nc // 2, nc]))
        wrong = float(rng.choice([0.0, 0.1, 0.3]))
        ta, tb, K, truth = ex.synthetic_sift_pair(int(rng.integers(0, 1 << 30)), nc, ne, wrong_fraction=wrong)
        rs = int(rng.integers(1, 1 << 30))
        os.environ["SPECTAVI_RANSAC_SEED"] = str(rs)
        host = ex.host_pipeline(ta, tb, K, descriptor_only=True, maximum_tries=20000)
        dev = ex.device_pipeline(torch.from_numpy(ta).cuda(), torch.from_numpy(tb).cuda(), K, maximum_tries=20000, seed=rs)
        ok = np.array_equal(host['matches'], dev['matches']) and host['ransac']['success'] == dev['ransac']['success']
        if ok and dev['ransac']['success']:
            # the calibration x K^-T is a numpy product on one side and a torch matmul on the other: the
            # correspondences may differ in the last bit, hence F too, hence (two equal singular values of E)
            # the sign of the camera; everything is compared to rounding, the camera up to its sign
            hP, dP = host['ransac']['camera'], dev['ransac']['camera']
            # (a minimal solver amplifies a last-bit difference of its seven correspondences by up to ~1e8)
            hF, dF = host['ransac']['essential'], dev['ransac']['essential']
            ok = (np.abs(hF - dF).max() <= 1e-6 * np.abs(hF).max() and
                  min(np.abs(hP - dP).max(), np.abs(hP + dP).max()) < 1e-6 and
                  np.array_equal(host['ransac']['inlier_idx'][:, 0], dev['ransac']['inlier_idx']) and
                  np.allclose(host['points'], dev['points'], rtol=1e-5, atol=1e-6))
            m = dev['matches']
            good = (truth['true_row0'][m[:, 0]] == m[:, 1]) & truth['consistent'][m[:, 0]]
            # against the scene: the true consistent pairs are inliers (a wrongly placed keypoint may satisfy the
            # epipolar constraint by chance: a few extras are allowed), E is the scene's up to scale
            inl = dev['ransac']['inlier_idx']
            missed, extra = np.setdiff1d(np.flatnonzero(good), inl), np.setdiff1d(inl, np.flatnonzero(good))
            from tests import mvg_checks as mc
            par = mc.parallel(dev['ransac']['essential'], truth['E'])
            if ok and not (len(missed) <= 0.3 * good.sum() and len(extra) <= 3 + 0.01 * len(m) and par >= 1 - 1e-3):  # the search stops at 70 % inliers  # one minimal sample on float32 pixel coordinates, no refit
                raise SystemExit("PIPELINE vs SCENE case=%d common=%d extra=%d wrong=%g ransac seed %d: missed %d extra %d of %d, "
                                 "1 - cos(E, E_true) = %.2e" % (n, nc, ne, wrong, rs, len(missed), len(extra), good.sum(), 1 - par))
        if not ok:
            raise SystemExit("PIPELINE MISMATCH (device-resident vs host front-end) case=%d common=%d extra=%d wrong=%g ransac seed %d" %
                             (n, nc, ne, wrong, rs))
        n += 1
        if only_case is not None:
            break
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0, help="budget per path")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="comma list of l1k2,cascade,dlt,ratio,score,normalize,seven_point,ransac_fit,pipeline")
    ap.add_argument("--case", type=int, default=None, help="re-run one case number of the --only path")
    a = ap.parse_args()
    want = set(filter(None, a.only.split(",")))
    for name, fn in (("l1k2", fuzz_l1k2), ("cascade", fuzz_cascade), ("dlt", fuzz_dlt), ("ratio", fuzz_ratio),
                     ("score", fuzz_score), ("normalize", fuzz_normalize), ("seven_point", fuzz_seven_point),
                     ("ransac_fit", fuzz_ransac_fit), ("pipeline", fuzz_pipeline)):
        if want and name not in want:
            continue
        cases = fn(a.seed, a.seconds, a.case)
        if name == "pipeline":
            print("pipeline: %d random scenes, device-resident == host front-end (to rounding), both consistent with the scene" % cases, flush=True)
            continue
        how = {"seven_point": "agree with", "ransac_fit": "agree with",
               "dlt": "within tolerance of the LAPACK definition, the JacobiSVD oracle (up to sign) and the host mirror; decisions as",
               "score": "decide as (away from the threshold)"}.get(name, "bit-identical to")
        print("%s: %d random cases %s the oracle" % (name, cases, how), flush=True)
