"""Generates the golden fixtures in this directory.

The reference holds no stored vectors for the hot path (its tests draw from one
np.random stream whose state depends on nose's method order, reference
test/test_feature.py:7, test/test_mvg.py:8) and it cannot be built or imported in
the build container (Eigen3 / cndarray / nose absent).  The fixtures are
therefore produced by the CPU oracle (oracle/) *after* it has been checked
against the independent numpy statements in oracle/oracle.py, and every
fixture stores the numpy result next to the oracle result when the case is
small enough for numpy.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def u8(seed, rows, dim, hi=256):
    return np.random.default_rng(seed).integers(0, hi, (rows, dim), dtype=np.uint8)


def l1k2_case(name, x, y):
    idx, dist = o.nn_bruteforcel1k2(x, y, nthreads=8)
    nidx, ndist = o.numpy_l1_top2(x, y)
    assert np.array_equal(idx, nidx) and np.array_equal(dist, ndist), name
    np.savez_compressed(os.path.join(OUT, name), x=x, y=y, idx=idx, dist=dist)


def main():
    # the reference's own test shape: 200 x 144 uniform uint8 (test/test_feature.py:108-115)
    l1k2_case("l1k2_200x144.npz", u8(0xdeadbeef, 200, 144), u8(0xdeadbef0, 200, 144))
    # BASELINE config 1: 1k x 1k, D=128
    l1k2_case("l1k2_1kx1k_128.npz", u8(1, 1000, 128), u8(2, 1000, 128))
    # tie-heavy: values in {0,1} -> many equal distances, exercises the (dist, idx) rule
    l1k2_case("l1k2_ties_300x500_64.npz", u8(3, 300, 64, hi=2), u8(4, 500, 64, hi=2))
    # duplicates of the query in the database: distance 0 ties
    x = u8(5, 257, 128)
    y = x[[5, 5, 100, 256, 0]].copy()
    x[17] = x[5]
    x[200] = x[5]
    l1k2_case("l1k2_dups_257x5_128.npz", x, y)
    # sentinel cases: 0, 1, 2 database rows
    for m in (0, 1, 2):
        l1k2_case("l1k2_m%d.npz" % m, u8(6, m, 32).reshape(m, 32), u8(7, 9, 32))

    # cascade: 2k x 2k, fixed hyperplanes (numpy-generated, stored)
    rng = np.random.default_rng(0x5eed)
    xf = (u8(8, 2000, 128).astype(np.float32) - 128)
    yf = (u8(9, 2000, 128).astype(np.float32) - 128)
    # planted neighbours: half of the queries are noisy copies of database rows
    perm = rng.integers(0, 2000, 1000)
    yf[:1000] = np.clip(xf[perm] + rng.integers(-3, 4, (1000, 128)), -128, 127)
    m, n, g = 8, 2, 2
    d = rng.standard_normal((n, 128, m)).astype(np.float32)
    idx, dist, ncand, nset, xcodes, ysign, ymask = o.nn_cascading_hash(xf, yf, m, n, g, d, debug=True)
    np.savez_compressed(os.path.join(OUT, "cascade_2kx2k_m8n2g2.npz"), x=xf.astype(np.int8),
                        y=yf.astype(np.int8), dict=d, m=m, n=n, g=g, idx=idx, dist=dist,
                        ncand=ncand, nset=nset, xcodes=xcodes, ysign=ysign, ymask=ymask)
    # the reference's own cascade test configuration (test/test_feature.py:134-143)
    rng = np.random.default_rng(0xfeed)
    xs = rng.standard_normal((200, 144)).astype(np.float32)
    ys = rng.standard_normal((200, 144)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "cascade_ref_test_inputs.npz"), x=xs, y=ys,
                        dict=rng.standard_normal((16, 144, 8)).astype(np.float32))

    # DLT: 1000 points, noise-free and noisy
    rng = np.random.default_rng(0xd17)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(R) < 0:
        R = -R
    P1 = np.hstack([R, rng.standard_normal((3, 1))])
    X = np.hstack([rng.standard_normal((1000, 3)) + np.array([0, 0, 5.0]), np.ones((1000, 1))])
    x = X @ P0.T
    xp = X @ P1.T
    x[500:, :2] += rng.normal(0, 1e-3, (500, 2)) * x[500:, 2:3]
    xp[500:, :2] += rng.normal(0, 1e-3, (500, 2)) * xp[500:, 2:3]
    # X / err: the oracle (oracle_jacobisvd.cpp: the reference's arithmetic with Eigen's two-sided
    # JacobiSVD restated; sign of X as that SVD leaves it).  X_mirror / err_mirror: the host mirror
    # of the HIP kernel's operation sequence (sign canonicalised to X[3] >= 0), which the kernel must
    # reproduce bit for bit.  X_lapack: numpy.linalg.svd null vectors, canonical sign.
    tri = o.dlt_triangulate(P0, P1, x, xp)
    err = o.dlt_reprojection_error(P0, P1, x, xp)
    mir = o.dlt_mirror_triangulate(P0, P1, x, xp)
    mir_err = o.dlt_mirror_reprojection_error(P0, P1, x, xp)
    ref = o.numpy_dlt_null_vector(P0, P1, x, xp)
    assert np.allclose(o.canonical_sign(tri), ref, rtol=0, atol=1e-12)
    assert np.allclose(mir, ref, rtol=0, atol=1e-12)
    np.savez_compressed(os.path.join(OUT, "dlt_1000.npz"), P0=P0, P1=P1, x=x, xp=xp, X=tri, err=err,
                        X_mirror=mir, err_mirror=mir_err, X_lapack=ref)
    # the reference's own golden SIFT table (test/test_feature.py:38-41 loads it with np.loadtxt):
    # a data file, stored here as float32 npz so that GPU tests have real-distribution descriptors
    ref_table = "/root/reference/data/sift-test/sur-ogre.sift"
    if os.path.exists(ref_table):
        np.savez_compressed(os.path.join(OUT, "sift_sur_ogre_table.npz"),
                            table=np.loadtxt(ref_table).astype(np.float32))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
