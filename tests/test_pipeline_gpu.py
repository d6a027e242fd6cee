"""GPU: the reference's example pipeline, steps 2-4 (example/ex01_essential_estimation.py:89-187:
match keypoints, estimate the essential matrix by RANSAC, triangulate the inliers), on synthetic SIFT
tables -- through the reference front-end's functions, with every intermediate resident in HBM, and
the same steps on the CPU oracle.  All rows of SURVEY 8 compose here: normalisation, L1 2-NN /
cascade hash, ratio test, SIFT-table split, match coordinates, RANSAC fit, DLT."""
import numpy as np
import pytest

from examples import essential_from_sift_tables as ex

pytestmark = pytest.mark.gpu


def _oracle_pipeline(oracle, t0, t1, K, samples, descriptor_only):
    """Steps 2-4 with the CPU oracle for every library call and numpy for the glue."""
    from spectavi_amd import feature  # numpy-only function, identical to the reference's (tested in test_abi)
    fx, fy = (t0[:, 4:], t1[:, 4:]) if descriptor_only else (t0, t1)
    _x = feature.normalize_to_ubyte_and_multiple_16_dim(fx)
    _y = feature.normalize_to_ubyte_and_multiple_16_dim(fy)
    nn_idx, nn_dist = oracle.nn_bruteforcel1k2((_x + 128).astype('uint8'), (_y + 128).astype('uint8'), nthreads=8)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = nn_dist[:, 1] / nn_dist[:, 0].astype('float64')
    pass_idx = ratio >= 1.75
    matches = np.c_[np.flatnonzero(pass_idx), nn_idx[pass_idx, 0].astype(np.int64)].astype(np.int32)
    iK = np.linalg.inv(K)
    x0 = ex.homogeneous(t0[matches[:, 1], :2].astype(np.float64)) @ iK.T
    x1 = ex.homogeneous(t1[matches[:, 0], :2].astype(np.float64)) @ iK.T
    fit = oracle.ransac_fit(x0, x1, samples(len(x0)), required_percent_inliers=ex.RANSAC_QUALITY['medium'],
                            reprojection_error_allowed=3.35e-4, find_best_even_in_failure=False,
                            singular_value_ratio_allowed=1e-3)
    P0 = np.hstack((np.eye(3), np.zeros((3, 1))))
    X = oracle.dlt_triangulate(P0, fit['camera'], x0[fit['inlier_idx']], x1[fit['inlier_idx']])
    return matches, fit, X / X[:, 3:], x0, x1


def _check_against_truth(out, truth, t1):
    m, r = out['matches'], out['ransac']
    true_pair = truth['true_row0'][m[:, 0]] == m[:, 1]
    assert true_pair.mean() > 0.98 and len(m) > 0.8 * (truth['true_row0'] >= 0).sum()
    assert r['success']
    rE = r['essential'] / np.linalg.svd(r['essential'])[1][0]
    assert np.std(rE / truth['E']) < 1e-3  # a 7-point fit to float32 pixel coordinates
    inl = np.asarray(r['inlier_idx']).reshape(-1)
    # the inliers are exactly the matches that are true pairs at geometrically consistent positions
    good = true_pair & truth['consistent'][m[:, 0]]
    assert np.array_equal(inl, np.flatnonzero(good))
    # the triangulated points are the scene points (the true baseline has unit length, as the fit's)
    X_true = truth['X_of_row1'][m[inl, 0]]
    # (to the float32 rounding of the pixel coordinates in the tables, ~6e-5 px)
    assert np.abs(out['points'][:, :3] - X_true).max() < 1e-4


@pytest.mark.parametrize("method,descriptor_only", [("bruteforce", False), ("bruteforce", True),
                                                    ("cascading-hash", True)])
def test_example_pipeline_host_front_end(monkeypatch, method, descriptor_only):
    monkeypatch.setenv("SPECTAVI_RANSAC_SEED", "7")
    monkeypatch.setenv("SPECTAVI_HASH_SEED", "7")
    t0, t1, K, truth = ex.synthetic_sift_pair(seed=3, n_common=3000, n_extra=1500)
    out = ex.host_pipeline(t0, t1, K, matching_method=method, descriptor_only=descriptor_only, maximum_tries=20000)
    if method == "cascading-hash":
        # an approximate matcher: it may miss pairs, never invent distances; the model is still found
        m = out['matches']
        assert (truth['true_row0'][m[:, 0]] == m[:, 1]).mean() > 0.98 and len(m) > 0.5 * 3000
        assert out['ransac']['success']
        rE = out['ransac']['essential'] / np.linalg.svd(out['ransac']['essential'])[1][0]
        assert np.std(rE / truth['E']) < 1e-3  # a 7-point fit to float32 pixel coordinates
    else:
        _check_against_truth(out, truth, t1)


def test_example_pipeline_device_resident_equals_host_and_oracle(oracle, monkeypatch):
    import torch
    from spectavi_amd import mvg
    monkeypatch.setenv("SPECTAVI_RANSAC_SEED", "11")
    t0, t1, K, truth = ex.synthetic_sift_pair(seed=5, n_common=2500, n_extra=2000, wrong_fraction=0.2)
    host = ex.host_pipeline(t0, t1, K, descriptor_only=True, maximum_tries=20000)
    dev = ex.device_pipeline(torch.from_numpy(t0).cuda(), torch.from_numpy(t1).cuda(), K, maximum_tries=20000, seed=11)
    _check_against_truth(host, truth, t1)
    _check_against_truth(dev, truth, t1)
    assert np.array_equal(host['matches'], dev['matches'])
    # same seed, same subsets, hence the same model -- to rounding: the calibration x K^-T is a numpy
    # product on one side and a torch matmul on the other, so the correspondences may differ in the last
    # bit, F with them, and (E has two equal singular values) the camera in its sign
    # (a minimal solver amplifies a last-bit difference of its seven correspondences by up to ~1e8)
    hF, dF = host['ransac']['essential'], dev['ransac']['essential']
    assert np.abs(hF - dF).max() <= 1e-6 * np.abs(hF).max()
    hP, dP = host['ransac']['camera'], dev['ransac']['camera']
    assert min(np.abs(hP - dP).max(), np.abs(hP + dP).max()) < 1e-6
    assert np.array_equal(host['ransac']['inlier_idx'][:, 0], dev['ransac']['inlier_idx'])
    assert np.allclose(host['points'], dev['points'], rtol=1e-5, atol=1e-6)
    # the CPU oracle on the same subsets
    om, ofit, oX, x0, x1 = _oracle_pipeline(oracle, t0, t1, K, lambda n: mvg.ransac_sample(11, n, dev['ransac']['tries_run']),
                                            descriptor_only=True)
    assert np.array_equal(om, dev['matches'])
    assert ofit['success'] and ofit['best_try'] == dev['ransac']['best_try'] and ofit['best_root'] == dev['ransac']['best_root']
    assert np.array_equal(ofit['inlier_idx'], dev['ransac']['inlier_idx'])
    assert np.allclose(ofit['essential'], dev['ransac']['essential'], rtol=1e-8, atol=1e-12)
    # points: the oracle's camera may be -P (same camera), its X is the same point
    assert np.abs(oX - dev['points']).max() < 1e-7


def test_example_pipeline_device_resident_cascade():
    """The same steps with the cascade hash as the matcher, device-resident: an approximate matcher
    (it may miss pairs, it never invents one), the model is still the scene's."""
    import torch
    t0, t1, K, truth = ex.synthetic_sift_pair(seed=8, n_common=3000, n_extra=1500)
    dev = ex.device_pipeline(torch.from_numpy(t0).cuda(), torch.from_numpy(t1).cuda(), K, maximum_tries=20000, seed=3,
                             matching_method='cascading-hash')
    m = dev['matches']
    assert (truth['true_row0'][m[:, 0]] == m[:, 1]).mean() > 0.98 and len(m) > 0.5 * 3000
    assert dev['ransac']['success']
    rE = dev['ransac']['essential'] / np.linalg.svd(dev['ransac']['essential'])[1][0]
    assert np.std(rE / truth['E']) < 1e-3
    X_true = truth['X_of_row1'][m[dev['ransac']['inlier_idx'], 0]]
    assert np.abs(dev['points'][:, :3] - X_true).max() < 1e-4
