"""Steps 2-4 of the reference's example pipeline on the gfx950 library.

The reference's example/ex01_essential_estimation.py runs SIFT detection (step 1, vlfeat: out of
scope here), keypoint matching (step 2, :89-131), RANSAC estimation of the essential matrix (step 3,
:134-164), triangulation of the inliers (step 4, :167-187) and rectification (step 5, out of scope).
This file is steps 2-4 with the same calls and options, fed with SIFT tables (rows of 132 float32:
x, y, sigma, angle, 128 descriptor values, reference src/Sift.h:13,115-123) -- synthetic ones by
default, since extracting them is not part of this library -- in two forms:

  * `host_pipeline`: the reference front-end's functions (`feature.*`, `mvg.*`), numpy in / numpy out,
    exactly the example's sequence of calls;
  * `device_pipeline`: the same steps with every intermediate resident in HBM (`spectavi_amd.device`),
    only the RANSAC result and the triangulated points come back.

    python examples/essential_from_sift_tables.py [--features 20000] [--matching-method cascading-hash]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RANSAC_QUALITY = {'low': .6, 'medium': .7, 'high': .75, 'ultra': .8, 'uber': .9}  # reference example :142-143


def homogeneous(x):
    return np.hstack([x, np.ones((x.shape[0], 1), dtype=x.dtype)])


def synthetic_sift_pair(seed=0, n_common=3000, n_extra=1500, wrong_fraction=0.15, pixel_noise=0.0,
                        desc_noise=6.0):
    """Two SIFT tables of one scene seen by two calibrated cameras.

    n_common scene points appear in both tables (their descriptors differ by N(0, desc_noise) per
    component), n_extra unrelated keypoints are added to each; for a `wrong_fraction` of the common
    points the second view's keypoint sits at an unrelated position (a descriptor match that is a
    geometric outlier).  Returns (table0, table1, K, truth): truth holds R, t, the essential matrix
    (unit largest singular value), the scene points and, per row of table1, the row of table0 it
    truly matches (-1: none) and whether that match is geometrically consistent."""
    rng = np.random.default_rng(seed)
    K = np.array([[1000.0, 0.0, 640.0], [0.0, 1000.0, 480.0], [0.0, 0.0, 1.0]])
    a = rng.standard_normal(3)
    a /= np.linalg.norm(a)
    th = rng.uniform(0.05, 0.25) * rng.choice([-1, 1])
    S = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * S + (1 - np.cos(th)) * (S @ S)
    t = rng.standard_normal(3)
    t /= np.linalg.norm(t)
    X = np.c_[rng.uniform(-2.5, 2.5, n_common), rng.uniform(-1.8, 1.8, n_common), rng.uniform(4, 8, n_common)]
    p0 = (X / X[:, 2:]) @ K.T
    Xc1 = X @ R.T + t
    p1 = (Xc1 / Xc1[:, 2:]) @ K.T
    p1[:, :2] += pixel_noise * rng.standard_normal((n_common, 2))
    consistent = np.ones(n_common, bool)
    wrong = rng.choice(n_common, int(round(wrong_fraction * n_common)), replace=False)
    p1[wrong, :2] = np.c_[rng.uniform(0, 1280, len(wrong)), rng.uniform(0, 960, len(wrong))]
    consistent[wrong] = False

    def descriptors(n):  # SIFT-like: non-negative, unit norm, stored as uint8(512 d) (src/Sift.h:118-121)
        d = np.abs(rng.standard_normal((n, 128))) ** 2
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return np.minimum(np.floor(512.0 * d), 255.0)

    d_common = descriptors(n_common)
    d1 = np.clip(np.round(d_common + desc_noise * rng.standard_normal(d_common.shape)), 0, 255)

    def table(pix, desc):
        n = len(pix)
        geom = np.c_[pix[:, :2], rng.uniform(1, 8, n), rng.uniform(-np.pi, np.pi, n)]
        return np.hstack([geom, desc]).astype(np.float32)

    extra0 = np.c_[rng.uniform(0, 1280, n_extra), rng.uniform(0, 960, n_extra)]
    extra1 = np.c_[rng.uniform(0, 1280, n_extra), rng.uniform(0, 960, n_extra)]
    t0 = np.vstack([table(p0, d_common), table(extra0, descriptors(n_extra))])
    t1 = np.vstack([table(p1, d1), table(extra1, descriptors(n_extra))])
    perm0, perm1 = rng.permutation(len(t0)), rng.permutation(len(t1))
    inv0 = np.empty_like(perm0)
    inv0[perm0] = np.arange(len(perm0))
    true_row0 = np.full(len(t1), -1, np.int64)
    ok = np.zeros(len(t1), bool)
    for new1, old1 in enumerate(perm1):
        if old1 < n_common:
            true_row0[new1] = inv0[old1]
            ok[new1] = consistent[old1]
    Xrow1 = np.full((len(t1), 3), np.nan)
    Xrow1[np.flatnonzero(perm1 < n_common)] = X[perm1[perm1 < n_common]]
    E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    truth = {'R': R, 't': t, 'E': E / np.linalg.svd(E)[1][0], 'true_row0': true_row0, 'consistent': ok,
             'X_of_row1': Xrow1}
    return t0[perm0], t1[perm1], K, truth


# ---------------------------------------------------------------------------------------------
# the reference front-end's sequence of calls (numpy in, numpy out)
# ---------------------------------------------------------------------------------------------
def step2_match_keypoints(x, y, matching_method='bruteforce', min_ratio=1.75, descriptor_only=False):
    """reference example :89-106.  x, y: SIFT tables.  Returns (xd, yd, matches): the matched rows of
    both tables and int32 [n,2] (row of y, row of x).  descriptor_only: match on the 128 descriptor
    columns instead of the example's whole 132-column table."""
    from spectavi_amd import feature
    fx, fy = (x[:, 4:], y[:, 4:]) if descriptor_only else (x, y)
    _x = feature.normalize_to_ubyte_and_multiple_16_dim(fx)
    _y = feature.normalize_to_ubyte_and_multiple_16_dim(fy)
    if matching_method == 'bruteforce':
        nn_idx, nn_dist = feature.nn_bruteforcel1k2((_x + 128).astype('uint8'), (_y + 128).astype('uint8'))
    elif matching_method == 'cascading-hash':
        nn_idx, nn_dist = feature.nn_cascading_hash(_x, _y)
    else:
        raise ValueError(matching_method)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = nn_dist[:, 1] / nn_dist[:, 0].astype('float64')
    pass_idx = ratio >= min_ratio
    idx0 = nn_idx[:, 0]
    # rows without a neighbour carry the (size_t)-1 sentinel; their ratio is never a pass in practice,
    # but do not index with it
    pass_idx &= idx0 != np.uint64(0xFFFFFFFFFFFFFFFF)
    matches = np.c_[np.flatnonzero(pass_idx), idx0[pass_idx].astype(np.int64)].astype(np.int32)
    return x[matches[:, 1]], y[matches[:, 0]], matches


def step3_estimate_essential_matrix(xd, yd, K, ransac_quality='medium', maximum_tries=10000000):
    """reference example :134-164 (same options)."""
    from spectavi_amd import mvg
    iK = np.linalg.inv(K)
    x0 = np.dot(homogeneous(xd[..., :2].astype(np.float64)), iK.T)
    x1 = np.dot(homogeneous(yd[..., :2].astype(np.float64)), iK.T)
    ransac_options = {'required_percent_inliers': RANSAC_QUALITY[ransac_quality],
                      'reprojection_error_allowed': 3.35e-4,
                      'maximum_tries': maximum_tries,
                      'find_best_even_in_failure': False,
                      'singular_value_ratio_allowed': 1e-3,
                      'progressbar': False}
    return mvg.ransac_fitter(x0, x1, options=ransac_options), x0, x1


def step4_triangulate_points(ransac, x0, x1):
    """reference example :167-176."""
    from spectavi_amd import mvg
    idx = ransac['inlier_idx'][:, 0]
    P0 = np.hstack((np.eye(3), np.zeros((3, 1))))
    RX = mvg.dlt_triangulate(P0, ransac['camera'], x0[idx], x1[idx])
    return RX[..., :] / RX[..., -1].reshape(-1, 1)


def host_pipeline(table0, table1, K, matching_method='bruteforce', min_ratio=1.75, ransac_quality='medium',
                  descriptor_only=False, maximum_tries=10000000):
    xd, yd, matches = step2_match_keypoints(table0, table1, matching_method, min_ratio, descriptor_only)
    ransac, x0, x1 = step3_estimate_essential_matrix(xd, yd, K, ransac_quality, maximum_tries)
    RX = step4_triangulate_points(ransac, x0, x1) if ransac['success'] else np.zeros((0, 4))
    return {'matches': matches, 'ransac': ransac, 'points': RX}


# ---------------------------------------------------------------------------------------------
# the same steps with every intermediate resident in HBM
# ---------------------------------------------------------------------------------------------
def device_pipeline(table0, table1, K, min_ratio=1.75, ransac_quality='medium', maximum_tries=10000000, seed=0,
                    matching_method='bruteforce', hash_seed=0x5eed):
    """SIFT tables (CUDA float32 [n,132]) -> split -> normalise the descriptor columns -> exact L1 2-NN
    (or the cascade hash + L1 refine) -> ratio test + compaction -> matched coordinates -> calibration
    -> RANSAC fit -> triangulation of the inliers.  One upload (the tables), one small download (the
    model), one download of the points."""
    import torch
    from spectavi_amd import device as spv
    from spectavi_amd import feature
    geom0, desc0 = spv.split_sift_table(table0)
    geom1, desc1 = spv.split_sift_table(table1)
    if matching_method == 'bruteforce':
        u0 = spv.normalize(desc0.to(torch.float32), want_float=False, want_ubyte=True)
        u1 = spv.normalize(desc1.to(torch.float32), want_float=False, want_ubyte=True)
        idx, dist = spv.l1k2(u0, u1)
    elif matching_method == 'cascading-hash':
        f0 = spv.normalize(desc0.to(torch.float32))
        f1 = spv.normalize(desc1.to(torch.float32))
        m = feature.auto_hash_bit_rate(f0.shape[0], f1.shape[0])  # the reference front-end's rule
        if m < 4:
            raise ValueError("tables too small for the cascade hash (the front-end falls back to brute force)")
        d = torch.from_numpy(feature.generate_hash_dict(hash_seed, f0.shape[1], m, 2)).to(table0.device)
        idx, dist = spv.cascade(f0, f1, d, g=2)
    else:
        raise ValueError(matching_method)
    matches, count = spv.ratio_test(idx, dist, min_ratio)
    p0, p1 = spv.match_coordinates(geom0, geom1, matches, count)
    n = int(count.item())
    iKt = torch.from_numpy(np.linalg.inv(K).T.copy()).to(table0.device)
    x0 = (p0[:n] @ iKt).contiguous()
    x1 = (p1[:n] @ iKt).contiguous()
    fit = spv.ransac_fit(x0, x1, required_percent_inliers=RANSAC_QUALITY[ransac_quality],
                         reprojection_error_allowed=3.35e-4, maximum_tries=maximum_tries,
                         find_best_even_in_failure=False, singular_value_ratio_allowed=1e-3, seed=seed)
    RX = np.zeros((0, 4))
    if fit['success']:
        sel = torch.from_numpy(fit['inlier_idx'].astype(np.int64)).to(table0.device)
        P0 = np.hstack((np.eye(3), np.zeros((3, 1))))
        X = spv.dlt_triangulate(P0, fit['camera'], x0[sel].contiguous(), x1[sel].contiguous())
        RX = (X / X[:, 3:]).cpu().numpy()
    return {'matches': matches[:n].cpu().numpy(), 'ransac': fit, 'points': RX}


def report(name, out, truth, dt):
    r = out['ransac']
    m = out['matches']
    true_pairs = truth['true_row0'][m[:, 0]] == m[:, 1]
    line = "%-28s %7.1f ms  matches %5d (%.1f %% true pairs)  success %s  inliers %.3f" % (
        name, dt * 1e3, len(m), 100.0 * true_pairs.mean() if len(m) else 0.0, r['success'], r['inlier_percent'])
    if r['success']:
        rE = r['essential'] / np.linalg.svd(r['essential'])[1][0]
        line += "  std(rE/E) %.2e  points %d" % (np.std(rE / truth['E']), len(out['points']))
    print(line)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--features", type=int, default=20000, help="scene points seen in both views")
    ap.add_argument("--extra", type=int, default=10000, help="unrelated keypoints per view")
    ap.add_argument("--matching-method", default="bruteforce", choices=["bruteforce", "cascading-hash"])
    ap.add_argument("--min-ratio", type=float, default=1.75)
    ap.add_argument("--ransac-quality", default="medium", choices=sorted(RANSAC_QUALITY))
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import torch
    t0, t1, K, truth = synthetic_sift_pair(a.seed, a.features, a.extra)
    for rep in range(2):  # the second round is the warm one
        s = time.perf_counter()
        out = host_pipeline(t0, t1, K, a.matching_method, a.min_ratio, a.ransac_quality)
        report("host front-end (%s)" % a.matching_method, out, truth, time.perf_counter() - s)
        s = time.perf_counter()
        out = host_pipeline(t0, t1, K, a.matching_method, a.min_ratio, a.ransac_quality, descriptor_only=True)
        report("  descriptor columns only", out, truth, time.perf_counter() - s)
        d0, d1 = torch.from_numpy(t0).cuda(), torch.from_numpy(t1).cuda()
        torch.cuda.synchronize()
        s = time.perf_counter()
        out = device_pipeline(d0, d1, K, a.min_ratio, a.ransac_quality, matching_method=a.matching_method)
        torch.cuda.synchronize()
        report("device-resident (%s)" % a.matching_method, out, truth, time.perf_counter() - s)


if __name__ == "__main__":
    main()
