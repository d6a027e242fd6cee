"""GPU parity: batched DLT triangulation through the C-ABI, judged three ways:

  * the ORACLE (oracle/oracle_jacobisvd.cpp: the reference's arithmetic with Eigen's two-sided
    JacobiSVD restated), up to the sign of X, within a conditioning-scaled tolerance;
  * the reference's DEFINITION with LAPACK as the solver (tests/dlt_checks.py): ||A X|| <= sigma4,
    |<X, v4>| >= 1 - 1e-9 where the singular-value gap allows, reprojection error within 1e-6
    relative (north_star's float tolerance);
  * the host MIRROR of the kernel's own operation sequence (oracle/oracle_dlt_mirror.cpp), to a
    few ulps times the point's conditioning (dlt_checks.check_against_mirror; bit for bit until
    round 3 replaced the kernel's IEEE divisions by v_rcp_f64 + Newton steps): determinism of the
    device code, not correctness.
"""
import numpy as np
import pytest

from tests import dlt_checks as dc

pytestmark = pytest.mark.gpu

# fp64 tolerance vs the mirror on well-conditioned batches.  The HIP kernel and the mirror execute the
# same operation sequence with contraction disabled, except that the kernel's reciprocals and
# reciprocal square roots are v_rcp_f64 / v_rsq_f64 + two Newton steps (< 1 ulp from exact).
RTOL = 1e-12


def test_golden(golden):
    from spectavi_amd import mvg
    g = golden("dlt_1000.npz")
    X = mvg.dlt_triangulate(g["P0"], g["P1"], g["x"], g["xp"])
    assert X.shape == (1000, 4)
    assert np.max(np.abs(X - g["X_mirror"])) <= RTOL
    assert np.max(np.abs(X - g["X_lapack"])) < 1e-12
    sgn = np.sign(np.einsum("ni,ni->n", X, g["X"]))           # oracle: sign as JacobiSVD leaves it
    assert np.max(np.abs(X - sgn[:, None] * g["X"])) < 1e-12
    err = mvg.dlt_reprojection_error(g["P0"], g["P1"], g["x"], g["xp"])
    assert err.shape == (1000, 1)
    dc.check_against_mirror(X, g["X_mirror"], g["P0"], g["P1"], g["x"], g["xp"], E=err, mE=g["err_mirror"], what="golden")
    assert np.allclose(err, g["err"], rtol=1e-6, atol=1e-12)   # oracle, north_star's float tolerance
    dc.check_definition(X, g["P0"], g["P1"], g["x"], g["xp"], err=err, what="golden")


def test_reference_test_properties():
    """reference test/test_mvg.py:94-125."""
    from spectavi_amd import mvg
    rng = np.random.default_rng(0xdeadbeef)
    for _ in range(20):
        P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
        X0 = rng.standard_normal(4)
        x, xp = P0 @ X0, P1 @ X0
        assert abs(mvg.dlt_reprojection_error(P0, P1, x, xp)[0, 0]) < 1e-3
        X = mvg.dlt_triangulate(P0, P1, x, xp)[0]
        assert np.allclose(X / X[3], X0 / X0[3])
        assert np.allclose(np.cross(P0 @ X, x), 0, atol=1e-8)


def test_batch_matches_oracle(oracle):
    from spectavi_amd import mvg
    rng = np.random.default_rng(3)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((100003, 4))
    x = Xw @ P0.T + rng.normal(0, 1e-3, (100003, 3))
    xp = Xw @ P1.T + rng.normal(0, 1e-3, (100003, 3))
    X = mvg.dlt_triangulate(P0, P1, x, xp)
    assert np.max(np.abs(np.linalg.norm(X, axis=1) - 1)) < 1e-12 and np.all(X[:, 3] >= 0)
    e = mvg.dlt_reprojection_error(P0, P1, x, xp)
    dc.check_against_mirror(X, oracle.dlt_mirror_triangulate(P0, P1, x, xp), P0, P1, x, xp, E=e,
                            mE=oracle.dlt_mirror_reprojection_error(P0, P1, x, xp), what="batch")
    # the oracle (JacobiSVD restatement) and the LAPACK statement of the definition
    dc.check_against_oracle(X, oracle.dlt_triangulate(P0, P1, x, xp), P0, P1, x, xp)
    st = dc.check_definition(X, P0, P1, x, xp, err=e, what="batch")
    assert st["well_separated"] > 99000
    oe = oracle.dlt_reprojection_error(P0, P1, x, xp)
    assert np.allclose(e, oe, rtol=1e-6, atol=1e-9)


def test_device_path_10m_properties():
    """BASELINE config 4 (10M point pairs) resident in HBM: noise-free points reproject to ~0
    and X equals the planted point up to scale."""
    import torch
    from spectavi_amd import device
    npt = 10_000_000
    rng = np.random.default_rng(1)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    P1 = np.hstack([R, rng.standard_normal((3, 1))])
    g = torch.Generator(device="cuda").manual_seed(7)
    Xw = torch.randn((npt, 4), dtype=torch.float64, device="cuda", generator=g)
    Xw[:, 2] += 5.0
    Xw[:, 3] = 1.0
    def project(P):  # elementwise on purpose: no dependency on a BLAS library being loaded
        Pt = torch.from_numpy(P).cuda()
        return torch.stack([(Xw * Pt[r]).sum(1) for r in range(3)], dim=1).contiguous()

    x, xp = project(P0), project(P1)
    X = device.dlt_triangulate(P0, P1, x, xp)
    err = device.dlt_reprojection_error(P0, P1, x, xp)
    torch.cuda.synchronize()
    assert float(err.max()) < 1e-8
    assert float((X / X[:, 3:4] - Xw).abs().max()) < 1e-6
    assert float(((X * X).sum(1) - 1).abs().max()) < 1e-12


def test_device_path_10m_noisy_config4(oracle):
    """BASELINE configs[3] at its stated inputs: P0 = [I|0], P1 = [R|t] from a
    seed, X ~ N(0,1)^3 + (0,0,5), w = 1, x = P0 X, xp = P1 X plus N(0, 1e-3) pixel noise, 10M point
    pairs resident in HBM -- plus 0.1 % gross outliers (an unrelated xp) so that the fast path's
    non-convergence exit and the Jacobi fallback run at scale.  Over all 10M points: finite, unit
    norm, canonical sign; the inliers reproject within the noise and sit at the planted point.  On a
    4096-point subsample (every outlier class included): the reference's definition with LAPACK as
    the solver (dlt_checks.check_definition), the JacobiSVD oracle up to sign, and the oracle's
    reprojection error at north_star's 1e-6 relative.  Reference: src/DltTriangulator.h:36-74,
    src/Spectavi.cpp:38-68."""
    import torch
    from spectavi_amd import device
    npt = 10_000_000
    rng = np.random.default_rng(4)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(R) < 0:
        R = -R
    P1 = np.hstack([R, rng.standard_normal((3, 1))])
    g = torch.Generator(device="cuda").manual_seed(44)
    Xw = torch.randn((npt, 4), dtype=torch.float64, device="cuda", generator=g)
    Xw[:, 2] += 5.0
    Xw[:, 3] = 1.0

    def project(P):  # elementwise on purpose: no dependency on a BLAS library being loaded
        Pt = torch.from_numpy(P).cuda()
        return torch.stack([(Xw * Pt[r]).sum(1) for r in range(3)], dim=1).contiguous()

    x, xp = project(P0), project(P1)
    x[:, :2] += 1e-3 * torch.randn((npt, 2), dtype=torch.float64, device="cuda", generator=g) * x[:, 2:3]
    xp[:, :2] += 1e-3 * torch.randn((npt, 2), dtype=torch.float64, device="cuda", generator=g) * xp[:, 2:3]
    outl = torch.arange(0, npt, 1000, device="cuda") + 7            # 0.1 %: every 1000th point
    xp[outl] = torch.randn((len(outl), 3), dtype=torch.float64, device="cuda", generator=g)
    inl = torch.ones(npt, dtype=torch.bool, device="cuda")
    inl[outl] = False

    X = device.dlt_triangulate(P0, P1, x, xp)
    err = device.dlt_reprojection_error(P0, P1, x, xp)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(X).all()) and bool(torch.isfinite(err[inl]).all())
    assert float(((X * X).sum(1) - 1).abs().max()) < 1e-12
    assert bool((X[:, 3] >= 0).all())
    # inliers: the error is a sum of two 2-D residual norms of N(0, 1e-3) noise (a few 1e-3), and the
    # triangulated point is the planted one up to ~ noise x depth^2 / baseline (1e-3 x 25 / 1)
    e_in = err[inl]
    assert float(e_in.median()) < 5e-3 and float((e_in < 2e-2).double().mean()) > 0.999
    behind = (Xw[:, 2] > 1.0) & inl                                   # well in front of camera 0
    dX = (X[behind] / X[behind, 3:4] - Xw[behind]).abs().max(1).values
    assert float(dX.median()) < 5e-2
    # outliers are inconsistent correspondences: their error is large, which is what RANSAC keys on
    assert float((err[outl] > 2e-2).double().mean()) > 0.9

    sub = np.unique(np.concatenate([np.arange(0, npt, 2563), outl[::50].cpu().numpy()]))
    assert len(sub) >= 4096
    ts = torch.from_numpy(sub).cuda()
    hx, hxp, hX, herr = x[ts].cpu().numpy(), xp[ts].cpu().numpy(), X[ts].cpu().numpy(), err[ts].cpu().numpy()
    st = dc.check_definition(hX, P0, P1, hx, hxp, err=herr, what="config 4, noisy")
    assert st["well_separated"] > 0.9 * len(sub)
    dc.check_against_oracle(hX, oracle.dlt_triangulate(P0, P1, hx, hxp), P0, P1, hx, hxp, what="config 4, noisy")
    oe = oracle.dlt_reprojection_error(P0, P1, hx, hxp).reshape(-1)
    sane = np.isfinite(oe) & (oe < 1e3)
    assert np.allclose(herr.reshape(-1)[sane], oe[sane], rtol=1e-6, atol=1e-9)


def test_host_call_10m_fresh_output_is_race_free(oracle):
    """The reference symbol on config 4's full size with a FRESH output array: dlt_triangulate's
    chunked host path downloads finished chunks while other threads are still pre-touching the
    caller's untouched pages (320 MB); a pre-touch store landing after a chunk's copy would zero the
    low byte of one double per 4 KB page.  Every row must equal the device-resident result bit for
    bit, and specifically the first double of every page."""
    import torch
    from spectavi_amd import device, mvg
    npt = 10_000_000
    rng = np.random.default_rng(8)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((npt, 4))
    x, xp = Xw @ P0.T, Xw @ P1.T
    del Xw
    for _ in range(2):  # a fresh np.empty inside mvg.dlt_triangulate each time
        X = mvg.dlt_triangulate(P0, P1, x, xp)
        dX = device.dlt_triangulate(P0, P1, torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda()).cpu().numpy()
        assert np.array_equal(X.view(np.uint64), dX.view(np.uint64))
        del X, dX


def _essential_cameras(rng):
    """Four candidate second cameras of an essential matrix (the set the reference scores,
    src/Camera.h:31-46), built directly from a random rotation and baseline."""
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(R) < 0:
        R = -R
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    W, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(W) < 0:
        W = -W
    return [np.hstack([R, t[:, None]]), np.hstack([R, -t[:, None]]),
            np.hstack([W @ R, t[:, None]]), np.hstack([W @ R, -t[:, None]])]


def test_score_hypotheses_matches_oracle(oracle):
    """RANSAC scoring (reference src/RansacFitter.h:59-95): counts and masks equal to the oracle's
    wherever the error is not on the threshold; the true camera wins."""
    from spectavi_amd import mvg
    rng = np.random.default_rng(12)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    cams = _essential_cameras(rng)
    npt = 5003
    Xw = np.hstack([rng.standard_normal((npt, 2)), rng.uniform(4, 8, (npt, 1)), np.ones((npt, 1))])
    x = Xw @ P0.T
    xp = Xw @ cams[0].T
    x[:, :2] += rng.normal(0, 2e-3, (npt, 2)) * x[:, 2:3]
    xp[:, :2] += rng.normal(0, 2e-3, (npt, 2)) * xp[:, 2:3]
    xp[::7] = rng.standard_normal((len(xp[::7]), 3))          # outliers
    P1s = np.stack(cams + [rng.standard_normal((3, 4)) for _ in range(3)])
    counts, mask = mvg.dlt_score_hypotheses(P0, P1s, x, xp, 1e-2, return_mask=True)
    assert np.array_equal(counts, mask.sum(1))
    # the oracle (a JacobiSVD solve per point and hypothesis, src/RansacFitter.h:59-73): identical
    # decisions except where its own error is within 1e-9 relative of the threshold; the same for the
    # host mirror of the kernel's operation sequence (a few ulps from the kernel)
    ocounts, omask, oerr = oracle.dlt_score_hypotheses(P0, P1s, x, xp, 1e-2, return_err=True)
    clear_o = np.abs(oerr - 1e-2) > 1e-11
    mcounts, mmask = oracle.dlt_mirror_score_hypotheses(P0, P1s, x, xp, 1e-2)
    assert np.array_equal(np.asarray(mask, bool)[clear_o], np.asarray(mmask, bool)[clear_o])
    assert np.all(np.abs(counts - mcounts) <= (~clear_o).sum(1))
    assert np.array_equal(np.asarray(mask, bool)[clear_o], omask[clear_o])
    assert np.all(np.abs(counts - ocounts) <= (~clear_o).sum(1))
    assert counts.argmax() == 0 and counts[0] > 0.8 * npt * 6 / 7
    assert np.array_equal(mvg.dlt_score_hypotheses(P0, P1s, x, xp, 1e-2), counts)
    # the inlier definition, recomputed from the exported pieces
    # (the exported error uses the QR + inverse-iteration solve, the scorer the Jacobi solve: equal
    # to ~1e-15, so only points within 1e-9 of the threshold are excluded from this cross-check)
    err = mvg.dlt_reprojection_error(P0, P1s[0], x, xp)[:, 0]
    front = oracle.dlt_cheirality(P0, P1s[0], x, xp)
    clear = np.abs(err - 1e-2) > 1e-9
    assert np.array_equal(mask[0][clear], ((err <= 1e-2) & front)[clear])


def test_many_camera_pairs(oracle):
    """30 random camera pairs (every fifth with almost no baseline, every seventh with a translation
    column scaled by 1e4), noisy and noise-free points.  The HIP kernel agrees with the host mirror
    of its operation sequence to a few ulps times the conditioning, and -- the check that matters --
    satisfies the reference's definition (LAPACK) and agrees with the JacobiSVD oracle up to sign."""
    from spectavi_amd import mvg
    rng = np.random.default_rng(77)
    worst = 0.0
    for k in range(30):
        P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
        if k % 5 == 0:
            P1 = P0 + 1e-6 * rng.standard_normal((3, 4))   # almost no baseline
        if k % 7 == 0:
            P0[:, 3] *= 1e4                                 # badly scaled translation
        Xw = rng.standard_normal((2003, 4))
        x = Xw @ P0.T + (k % 2) * rng.normal(0, 1e-3, (2003, 3))
        xp = Xw @ P1.T + (k % 2) * rng.normal(0, 1e-3, (2003, 3))
        X = mvg.dlt_triangulate(P0, P1, x, xp)
        mX = oracle.dlt_mirror_triangulate(P0, P1, x, xp)
        e = mvg.dlt_reprojection_error(P0, P1, x, xp)
        worst = max(worst, dc.check_against_mirror(X, mX, P0, P1, x, xp, E=e,
                                                   mE=oracle.dlt_mirror_reprojection_error(P0, P1, x, xp), what="pair %d" % k))
        dc.check_definition(X, P0, P1, x, xp, err=e, what="pair %d" % k)
        dc.check_against_oracle(X, oracle.dlt_triangulate(P0, P1, x, xp), P0, P1, x, xp, what="pair %d" % k)
    assert worst < 1e-6


def test_large_host_call_is_chunked_and_pipelined(oracle):
    """Results of 16 MB and more leave through the chunked upload / solve / download pipeline
    (pinned bounce buffers drained by host threads): 2 300 017 points = three chunks, the last one
    ragged; fresh and reused output arrays; both entry points; bit for bit the rows of one
    device-resident launch over all points (and those within the mirror tolerance of the mirror's)."""
    from spectavi_amd import mvg
    from spectavi_amd.mvg import _dlt_triangulate
    rng = np.random.default_rng(41)
    npt = 2_300_017
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((npt, 4))
    x = Xw @ P0.T + rng.normal(0, 1e-3, (npt, 3))
    xp = Xw @ P1.T + rng.normal(0, 1e-3, (npt, 3))
    import torch
    from spectavi_amd import device
    dx, dxp = torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda()
    want = device.dlt_triangulate(P0, P1, dx, dxp).cpu().numpy()   # one launch, no chunks, no host pipeline
    X = mvg.dlt_triangulate(P0, P1, x, xp)
    assert np.array_equal(X, want)
    dst = np.full((npt, 4), np.nan)
    _dlt_triangulate(P0, P1, npt, x, xp, dst)          # into an array that already has its pages
    assert np.array_equal(dst, want)
    e = mvg.dlt_reprojection_error(P0, P1, x, xp)       # 18 MB of errors: pipelined too
    assert np.array_equal(e.reshape(-1), device.dlt_reprojection_error(P0, P1, dx, dxp).cpu().numpy().reshape(-1))
    sub0 = np.arange(0, npt, 23)
    dc.check_against_mirror(X[sub0], oracle.dlt_mirror_triangulate(P0, P1, x[sub0], xp[sub0]), P0, P1, x[sub0], xp[sub0],
                            E=e[sub0], mE=oracle.dlt_mirror_reprojection_error(P0, P1, x[sub0], xp[sub0]), what="large host call")
    sub = rng.integers(0, npt, 20000)
    dc.check_definition(X[sub], P0, P1, x[sub], xp[sub], err=e[sub], what="large host call")


def test_two_pass_scorer_paths_agree(oracle):
    """The RANSAC scorer defers the solves that do not converge on its fast path to a second kernel
    (work list in the caller's workspace).  Whatever the work list holds -- everything, the first few
    entries of an undersized one, or nothing (no workspace: all in place) -- counts and masks are the
    same bits (and the host mirror's away from the threshold)."""
    import ctypes as ct
    import torch
    from spectavi_amd import device
    from spectavi_amd._lib import clib, check
    rng = np.random.default_rng(21)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    cams = _essential_cameras(rng)
    npt = 3001
    Xw = np.hstack([rng.standard_normal((npt, 2)), rng.uniform(4, 8, (npt, 1)), np.ones((npt, 1))])
    x, xp = Xw @ P0.T, Xw @ cams[0].T
    xp[:, :2] += rng.normal(0, 2e-3, (npt, 2)) * xp[:, 2:3]
    xp[::4] = rng.standard_normal((len(xp[::4]), 3))
    P1s = np.stack(cams + [c + 0.05 * rng.standard_normal((3, 4)) for c in cams] + [rng.standard_normal((3, 4)) for _ in range(9)])
    dP, dx, dxp = torch.from_numpy(P1s).cuda(), torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda()
    nh = P1s.shape[0]
    # the reference for "same bits whatever the path": the one-pass form (no work list, every solve in place)
    c1 = torch.zeros((nh,), dtype=torch.int32, device="cuda")
    m1 = torch.zeros((nh, npt), dtype=torch.uint8, device="cuda")
    check(clib.spv_dlt_score_hypotheses_device_ws(P0, dP.data_ptr(), nh, npt, dx.data_ptr(), dxp.data_ptr(), 1e-2,
                                                  c1.data_ptr(), m1.data_ptr(), None, 0, None))
    torch.cuda.synchronize()
    want_c, want_m = c1.cpu().numpy(), m1.cpu().numpy().astype(bool)
    # ... which agrees with the host mirror and with the oracle away from the threshold
    mir_c, mir_m = oracle.dlt_mirror_score_hypotheses(P0, P1s, x, xp, 1e-2)
    _, _, oerr = oracle.dlt_score_hypotheses(P0, P1s, x, xp, 1e-2, return_err=True)
    clear = np.isfinite(oerr) & (np.abs(oerr - 1e-2) > 1e-11)
    assert np.array_equal(want_m[clear], np.asarray(mir_m, bool)[clear])
    full = clib.spv_dlt_score_workspace_bytes(nh, npt)
    for ws_bytes in (full, 65536 + 8 * 1024 * 3, 65536 + 8 * 1024, 4096, 0):   # full list, 3 / 1 entries per shard, too small, none
        counts = torch.full((nh,), -7, dtype=torch.int32, device="cuda")
        mask = torch.full((nh, npt), 9, dtype=torch.uint8, device="cuda")
        ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device="cuda")
        check(clib.spv_dlt_score_hypotheses_device_ws(P0, dP.data_ptr(), nh, npt, dx.data_ptr(), dxp.data_ptr(), 1e-2,
                                                      counts.data_ptr(), mask.data_ptr(),
                                                      ws.data_ptr() if ws_bytes else None, ws_bytes, None))
        torch.cuda.synchronize()
        assert np.array_equal(counts.cpu().numpy(), want_c), ws_bytes
        assert np.array_equal(mask.cpu().numpy().astype(bool), want_m), ws_bytes
    c2, m2 = device.dlt_score_hypotheses(P0, dP, dx, dxp, 1e-2, want_mask=True)
    assert np.array_equal(c2.cpu().numpy(), want_c) and np.array_equal(m2.cpu().numpy().astype(bool), want_m)
    # some of these solves really do take the second pass
    assert (want_c < npt).any()
