"""ctypes binding of the CPU oracle (oracle/liboracle.so) + independent numpy checks.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from spectavi_amd/ (the product path has no
CPU fallback).  Each C function cites the reference lines it restates in its
own header (oracle_l1k2.cpp, oracle_cascade.cpp, oracle_jacobisvd.cpp).  The DLT rows have two
CPU sides: `dlt_*`, `essential_to_cameras`, `process_fundamental_matrix`, `jacobisvd` are the
ORACLE (oracle_jacobisvd.cpp: the reference's arithmetic including Eigen's two-sided JacobiSVD);
`dlt_mirror_*` (oracle_dlt_mirror.cpp) is a host mirror of the HIP kernel's own operation
sequence, good only for bit-reproducibility checks.

The numpy helpers at the bottom are a *second*, independent statement of the
same results (the same role `brute_force_nn_batched` plays in the reference's
test/test_feature.py:10-26) used to pin the C oracle itself.
"""
import ctypes as ct
import os
import subprocess

import numpy as np
from numpy.ctypeslib import ndpointer

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("SPECTAVI_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only, no GPU)."""
    if force or not os.path.exists(_LIB_PATH) or _stale():
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def _stale():
    """liboracle.so older than one of its sources (a checkout that moved on): rebuild."""
    if os.environ.get("SPECTAVI_ORACLE_LIB"):
        return False
    t = os.path.getmtime(_LIB_PATH)
    return any(os.path.getmtime(os.path.join(_HERE, f)) > t for f in os.listdir(_HERE)
               if f.endswith(".cpp") or f == "Makefile")


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ct.CDLL(_LIB_PATH)
        u8 = ndpointer(np.uint8, flags="C_CONTIGUOUS")
        f32 = ndpointer(np.float32, flags="C_CONTIGUOUS")
        f64 = ndpointer(np.float64, flags="C_CONTIGUOUS")
        u64 = ndpointer(np.uint64, flags="C_CONTIGUOUS")
        i32 = ndpointer(np.int32, flags="C_CONTIGUOUS")
        i64 = ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.oracle_nn_bruteforcel1k2.restype = ct.c_int
        L.oracle_nn_bruteforcel1k2.argtypes = [u8, u8, ct.c_int, ct.c_int, ct.c_int, ct.c_int, u64, i32]
        L.oracle_l1k2_candidates.restype = ct.c_int
        L.oracle_l1k2_candidates.argtypes = [u8, u8, ct.c_int, ct.c_int, i64, i32, ct.c_int, u64, i32]
        L.oracle_nn_cascading_hash.restype = ct.c_int
        L.oracle_nn_cascading_hash.argtypes = [
            f32, f32, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, f32, u64, f32,
            ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]
        for name in ("oracle_dlt_triangulate", "oracle_dlt_reprojection_error",
                     "oracle_dlt_mirror_triangulate", "oracle_dlt_mirror_reprojection_error"):
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = [f64, f64, ct.c_int, f64, f64, f64]
        for name in ("oracle_dlt_cheirality", "oracle_dlt_mirror_cheirality"):
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = [f64, f64, ct.c_int, f64, f64, u8]
        L.oracle_dlt_mirror_score_hypotheses.restype = None
        L.oracle_dlt_mirror_score_hypotheses.argtypes = [f64, f64, ct.c_int, ct.c_int, f64, f64, ct.c_double, i32, u8]
        L.oracle_dlt_score_hypotheses.restype = None
        L.oracle_dlt_score_hypotheses.argtypes = [f64, f64, ct.c_int, ct.c_int, f64, f64, ct.c_double, i32, u8,
                                                  ct.c_void_p]
        L.oracle_jacobisvd.restype = ct.c_int
        L.oracle_jacobisvd.argtypes = [f64, ct.c_int, f64, f64, f64]
        L.oracle_essential_to_cameras.restype = None
        L.oracle_essential_to_cameras.argtypes = [f64, f64]
        L.oracle_process_fundamental_matrix.restype = ct.c_int
        L.oracle_process_fundamental_matrix.argtypes = [
            f64, ct.c_double, f64, f64, ct.c_int, ct.c_double, ct.c_double, ct.c_int,
            ct.POINTER(ct.c_int32), f64, i32, ct.POINTER(ct.c_double), f64, i32]
        L.oracle_seven_point.restype = None
        L.oracle_seven_point.argtypes = [f64, f64, ct.POINTER(ct.c_int), f64, f64]
        L.oracle_ransac_fit.restype = None
        L.oracle_ransac_fit.argtypes = [f64, f64, ct.c_int, ct.c_double, ct.c_double, ct.c_int, ct.c_double, i32,
                                        ct.c_int, ct.POINTER(ct.c_int), f64, f64, ct.POINTER(ct.c_double), i32,
                                        ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
        L.oracle_max_threads.restype = ct.c_int
        _lib = L
    return _lib


def max_threads():
    """Host threads OpenMP would use by default (captured at load time: set_threads changes the team size)."""
    global _max_threads
    if _max_threads is None:
        _max_threads = int(lib().oracle_max_threads())
    return _max_threads


_max_threads = None


def set_threads(n):
    """Team size of the oracle's parallel regions that take no explicit count (the cascade)."""
    max_threads()
    L = lib()
    L.oracle_set_threads.restype = None
    L.oracle_set_threads.argtypes = [ct.c_int]
    L.oracle_set_threads(int(n))


def nn_bruteforcel1k2(x, y, nthreads=1):
    """(uint64[N,2], int32[N,2]) -- restates reference src/BruteForceNnL1K2.h:84-145."""
    x = np.ascontiguousarray(x, dtype=np.uint8)
    y = np.ascontiguousarray(y, dtype=np.uint8)
    xrows, dim = x.shape if x.ndim == 2 else (0, y.shape[1])
    yrows, ydim = y.shape
    assert xrows == 0 or dim == ydim
    idx = np.empty((yrows, 2), np.uint64)
    dist = np.empty((yrows, 2), np.int32)
    rc = lib().oracle_nn_bruteforcel1k2(x.reshape(-1, ydim), y, xrows, yrows, ydim, nthreads, idx, dist)
    if rc != 0:
        raise ValueError("Input matrix inner dimensions must be 16-byte aligned.")
    return idx, dist


def nn_cascading_hash(x, y, m, n, g, dict_, debug=False):
    """(uint64[N,2], float32[N,2], ncand int32[N], nset int32[N]) -- restates
    reference src/CascadingHashNn.h:86-245 with explicit hyperplanes dict_[n,dim,m]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    dict_ = np.ascontiguousarray(dict_, dtype=np.float32)
    xrows, dim = x.shape
    yrows = y.shape[0]
    assert dict_.shape == (n, dim, m)
    idx = np.empty((yrows, 2), np.uint64)
    dist = np.empty((yrows, 2), np.float32)
    ncand = np.zeros(yrows, np.int32)
    nset = np.zeros(yrows, np.int32)
    xcodes = np.zeros((n, xrows), np.uint32)
    ysign = np.zeros((n, yrows), np.uint32)
    ymask = np.zeros((n, yrows), np.uint32)
    rc = lib().oracle_nn_cascading_hash(
        x, y, xrows, yrows, dim, m, n, g, dict_, idx, dist,
        ncand.ctypes.data, nset.ctypes.data,
        xcodes.ctypes.data if debug else None,
        ysign.ctypes.data if debug else None,
        ymask.ctypes.data if debug else None)
    if rc != 0:
        raise ValueError("bad cascade arguments")
    if debug:
        return idx, dist, ncand, nset, xcodes, ysign, ymask
    return idx, dist, ncand, nset


def _dlt_args(P0, P1, x, xp):
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
    xp = np.ascontiguousarray(np.atleast_2d(xp), dtype=np.float64)
    assert P0.shape == (3, 4) and P1.shape == (3, 4)
    assert x.shape == xp.shape and x.shape[1] == 3
    return P0, P1, x, xp


def dlt_triangulate(P0, P1, x, xp):
    """X = V.col(3) of JacobiSVD(A) as the reference computes it (src/DltTriangulator.h:36-58),
    with the sign this restatement of Eigen's two-sided Jacobi leaves (NOT canonicalised)."""
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    dst = np.empty((x.shape[0], 4))
    lib().oracle_dlt_triangulate(P0, P1, x.shape[0], x, xp, dst)
    return dst


def dlt_reprojection_error(P0, P1, x, xp):
    """Reference src/DltTriangulator.h:67-74 on top of `dlt_triangulate`."""
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    dst = np.empty((x.shape[0], 1))
    lib().oracle_dlt_reprojection_error(P0, P1, x.shape[0], x, xp, dst)
    return dst


def dlt_cheirality(P0, P1, x, xp):
    """Reference src/DltTriangulator.h:76-86 (is_infront_both_cameras)."""
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    out = np.empty(x.shape[0], np.uint8)
    lib().oracle_dlt_cheirality(P0, P1, x.shape[0], x, xp, out)
    return out.astype(bool)


def dlt_score_hypotheses(P0, P1s, x, xp, max_error, return_err=False):
    """(counts int32[H], mask bool[H,npt][, err float64[H,npt]]) -- restates reference
    src/RansacFitter.h:59-95 (JacobiSVD solve per point and hypothesis)."""
    P1s = np.ascontiguousarray(P1s, dtype=np.float64).reshape(-1, 3, 4)
    P0, _, x, xp = _dlt_args(P0, P1s[0] if len(P1s) else np.zeros((3, 4)), x, xp)
    counts = np.zeros(P1s.shape[0], np.int32)
    mask = np.zeros((P1s.shape[0], x.shape[0]), np.uint8)
    err = np.zeros((P1s.shape[0], x.shape[0]))
    lib().oracle_dlt_score_hypotheses(P0, P1s.reshape(-1, 12), P1s.shape[0], x.shape[0], x, xp,
                                      float(max_error), counts, mask.reshape(-1), err.ctypes.data)
    return (counts, mask.astype(bool), err) if return_err else (counts, mask.astype(bool))


def jacobisvd(A):
    """(U, S, V, sweeps): Eigen::JacobiSVD(A, ComputeFullU | ComputeFullV) restated, A 3x3 or 4x4."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    assert A.shape == (n, n) and n in (3, 4)
    U, S, V = np.empty((n, n)), np.empty(n), np.empty((n, n))
    sweeps = lib().oracle_jacobisvd(A, n, U, S, V)
    return U, S, V, sweeps


def essential_to_cameras(E):
    """Reference src/Camera.h:31-46: the four candidate second cameras [4,3,4] of E."""
    E = np.ascontiguousarray(E, dtype=np.float64)
    assert E.shape == (3, 3)
    cams = np.empty((4, 3, 4))
    lib().oracle_essential_to_cameras(E, cams)
    return cams


def process_fundamental_matrix(F, x0, x1, singular_value_ratio_allowed=3e-2, required_percent_inliers=0.5,
                               reprojection_error_allowed=1e-2, find_best_even_in_failure=False):
    """Reference src/RansacFitter.h:42-95 for one candidate F.  Returns a dict: success,
    inlier_count, best_P [3,4], inlier_idx, gate_ratio, E, counts4 (per camera, -1 if gated)."""
    F = np.ascontiguousarray(F, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    assert F.shape == (3, 3) and x0.shape == x1.shape and x0.shape[1] == 3
    npt = x0.shape[0]
    cnt, ratio = ct.c_int32(0), ct.c_double(0.0)
    best_P, E = np.zeros((3, 4)), np.zeros((3, 3))
    idx, counts4 = np.zeros(max(npt, 1), np.int32), np.zeros(4, np.int32)
    ok = lib().oracle_process_fundamental_matrix(
        F, float(singular_value_ratio_allowed), x0, x1, npt, float(required_percent_inliers),
        float(reprojection_error_allowed), int(bool(find_best_even_in_failure)), ct.byref(cnt), best_P, idx,
        ct.byref(ratio), E, counts4)
    return {"success": bool(ok), "inlier_count": int(cnt.value) if ok else 0, "best_P": best_P if ok else None,
            "inlier_idx": idx[:cnt.value].copy() if ok else np.zeros(0, np.int32), "gate_ratio": float(ratio.value),
            "E": E, "counts4": counts4}


def seven_point(x, xp, return_basis=False):
    """Reference seven_point_algorithm (src/Spectavi.cpp:14-36): x, xp [7,2] euclidean.
    Returns Fs [nroot,3,3] (and the null-space basis [2,3,3] the roots were taken in)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    assert x.shape == (7, 2) and xp.shape == (7, 2)
    nroot = ct.c_int(0)
    Fs, basis = np.zeros((3, 3, 3)), np.zeros((2, 3, 3))
    lib().oracle_seven_point(x, xp, ct.byref(nroot), Fs.reshape(-1), basis.reshape(-1))
    return (Fs[:nroot.value].copy(), basis) if return_basis else Fs[:nroot.value].copy()


def ransac_fit(x0, x1, samples, required_percent_inliers=0.9, reprojection_error_allowed=0.5,
               find_best_even_in_failure=True, singular_value_ratio_allowed=3e-2):
    """RansacFitter::fit_essential with nthread = 1 (reference src/RansacFitter.h:152-272) over the given
    7-subsets (int32 [ntries,7]).  Returns the dict the reference front-end returns plus best_try / best_root."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    assert x0.shape == x1.shape and x0.shape[1] == 3 and samples.ndim == 2 and samples.shape[1] == 7
    npt = x0.shape[0]
    ok, n, bt, br = ct.c_int(0), ct.c_int(0), ct.c_int(-1), ct.c_int(-1)
    pct = ct.c_double(0.0)
    F, P, idx = np.zeros(9), np.zeros(12), np.zeros(max(npt, 1), np.int32)
    lib().oracle_ransac_fit(x0, x1, npt, float(required_percent_inliers), float(reprojection_error_allowed),
                            int(bool(find_best_even_in_failure)), float(singular_value_ratio_allowed),
                            samples.reshape(-1), samples.shape[0], ct.byref(ok), F, P, ct.byref(pct), idx,
                            ct.byref(n), ct.byref(bt), ct.byref(br))
    found = bt.value >= 0
    return {"success": bool(ok.value), "essential": F.reshape(3, 3) if found else None,
            "camera": P.reshape(3, 4) if found else None, "inlier_percent": float(pct.value),
            "inlier_idx": idx[:n.value].copy(), "best_try": bt.value, "best_root": br.value}


# ---- host mirror of the HIP kernels' operation sequence (bit-reproducibility only) -------------
def dlt_mirror_triangulate(P0, P1, x, xp):
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    dst = np.empty((x.shape[0], 4))
    lib().oracle_dlt_mirror_triangulate(P0, P1, x.shape[0], x, xp, dst)
    return dst


def dlt_mirror_reprojection_error(P0, P1, x, xp):
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    dst = np.empty((x.shape[0], 1))
    lib().oracle_dlt_mirror_reprojection_error(P0, P1, x.shape[0], x, xp, dst)
    return dst


def dlt_mirror_cheirality(P0, P1, x, xp):
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    out = np.empty(x.shape[0], np.uint8)
    lib().oracle_dlt_mirror_cheirality(P0, P1, x.shape[0], x, xp, out)
    return out.astype(bool)


def dlt_mirror_score_hypotheses(P0, P1s, x, xp, max_error):
    P1s = np.ascontiguousarray(P1s, dtype=np.float64).reshape(-1, 3, 4)
    P0, _, x, xp = _dlt_args(P0, P1s[0] if len(P1s) else np.zeros((3, 4)), x, xp)
    counts = np.zeros(P1s.shape[0], np.int32)
    mask = np.zeros((P1s.shape[0], x.shape[0]), np.uint8)
    lib().oracle_dlt_mirror_score_hypotheses(P0, P1s.reshape(-1, 12), P1s.shape[0], x.shape[0], x, xp,
                                             float(max_error), counts, mask.reshape(-1))
    return counts, mask.astype(bool)


def canonical_sign(X):
    """The HIP path's sign convention (X[3] >= 0, else first nonzero component > 0) applied to
    rows of X: the reference leaves the sign of V.col(3) to Eigen (parity unpinned)."""
    X = np.array(X, dtype=np.float64, copy=True)
    lead = np.where(X[:, 3] != 0, X[:, 3], np.where(X[:, 0] != 0, X[:, 0], np.where(X[:, 1] != 0, X[:, 1], X[:, 2])))
    X[lead < 0] *= -1
    return X


# ----------------------------------------------------------------------------------
# independent numpy statements (pin the C oracle; small sizes only)
# ----------------------------------------------------------------------------------
def numpy_l1_top2(x, y, batch=256):
    """Exact L1 two smallest (dist, idx) pairs per query row, lexicographic.
    Distances follow reference test/test_feature.py:10-26 (full |x-y| sum); the
    tie rule (lower index first) is made explicit with a stable argsort."""
    x = x.astype(np.int32)
    y = y.astype(np.int32)
    n, m = y.shape[0], x.shape[0]
    idx = np.full((n, 2), np.iinfo(np.uint64).max, np.uint64)
    dist = np.full((n, 2), np.iinfo(np.int32).max, np.int32)
    if m == 0:
        return idx, dist
    for i in range(0, n, batch):
        d = np.abs(y[i:i + batch, None, :] - x[None, :, :]).sum(-1)  # [b, m]
        order = np.argsort(d, axis=1, kind="stable")[:, :2]
        k = order.shape[1]
        idx[i:i + batch, :k] = order.astype(np.uint64)
        dist[i:i + batch, :k] = np.take_along_axis(d, order, axis=1).astype(np.int32)
    return idx, dist


def candidates_from_codes(xcodes, ysign, ymask):
    """[N, M] bool: k is a candidate of i iff some table j has
    ((xcode_j[k] ^ ysign_j[i]) & ~ymask_j[i]) == 0."""
    n = xcodes.shape[0]
    out = np.zeros((ysign.shape[1], xcodes.shape[1]), bool)
    for j in range(n):
        diff = (xcodes[j][None, :] ^ ysign[j][:, None]) & ~ymask[j][:, None]
        out |= diff == 0
    return out


def ratio_test_matches(nn_idx, nn_dist, min_ratio):
    """numpy statement of reference example/ex01_essential_estimation.py:102-106:
    ratio = d1 / d0.astype('float64'); pass = ratio >= min_ratio; rows (query, nn_idx[:,0])
    of the passing queries, ascending.  Queries without a neighbour (idx0 == uint64 max, an
    IndexError in the reference's fancy indexing) never pass."""
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = nn_dist[:, 1].astype(np.float64) / nn_dist[:, 0].astype(np.float64)
    ok = (ratio >= min_ratio) & (nn_idx[:, 0] != np.iinfo(np.uint64).max)
    q = np.flatnonzero(ok)
    return np.stack([q.astype(np.int32), nn_idx[q, 0].astype(np.int32)], axis=1)


def numpy_dlt_null_vector(P0, P1, x, xp):
    """LAPACK SVD null vector of the DLT matrix (reference src/DltTriangulator.h:51-58),
    sign-canonicalised like the oracle (X[3] >= 0)."""
    P0, P1, x, xp = _dlt_args(P0, P1, x, xp)
    out = np.empty((x.shape[0], 4))
    for i in range(x.shape[0]):
        u, v = x[i, 0] / x[i, 2], x[i, 1] / x[i, 2]
        up, vp = xp[i, 0] / xp[i, 2], xp[i, 1] / xp[i, 2]
        A = np.stack([u * P0[2] - P0[0], v * P0[2] - P0[1], up * P1[2] - P1[0], vp * P1[2] - P1[1]])
        X = np.linalg.svd(A)[2][3]
        nz = X[3] if X[3] != 0 else X[np.flatnonzero(X)[0]]
        out[i] = -X if nz < 0 else X
    return out
