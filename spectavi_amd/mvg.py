"""
``spectavi_amd.mvg``
====================
Multi-view-geometry front-end: the batched DLT triangulator of the reference's
``spectavi.mvg`` (reference spectavi/mvg.py:259-306) and its callers, the seven-point
algorithm and the RANSAC fitter (reference spectavi/mvg.py:112-248), bound to the
gfx950 build of ``libspectavi.so``.
"""
import ctypes as ct

import numpy as np
from numpy.ctypeslib import ndpointer

from spectavi_amd._lib import clib, check
from spectavi_amd.ndarray import NdArray


def hnormalize(x):
    """Homogeneous -> euclidean (reference spectavi/mvg.py:14-18)."""
    return x[..., :-1] / np.expand_dims(x[..., -1], axis=-1)


_dlt_triangulate = clib.dlt_triangulate
_dlt_triangulate.restype = None
_dlt_triangulate.argtypes = [ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                             ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                             ct.c_int,
                             ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                             ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                             ndpointer(ct.c_double, flags="C_CONTIGUOUS"), ]

_dlt_reprojection_error = clib.dlt_reprojection_error
_dlt_reprojection_error.restype = None
_dlt_reprojection_error.argtypes = list(_dlt_triangulate.argtypes)


def dlt_triangulate(P0, P1, x, xp, ret_error=False):
    """
    Triangulate `npt` two-view correspondences (reference spectavi/mvg.py:282-302).

    P0, P1 : float64 [3,4] camera matrices; x, xp : float64 [npt,3] (or [3])
    homogeneous image points.  Returns float64 [npt,4] unit-norm homogeneous
    points (sign canonicalised to X[3] >= 0), or [npt,1] reprojection errors
    with `ret_error=True`.
    """
    if not (P0.shape == (3, 4) and P1.shape == (3, 4)):
        raise TypeError('P0,P1 must be camera matrices.')
    if len(x.shape) == 1:
        x = np.expand_dims(x, axis=0)
    if len(xp.shape) == 1:
        xp = np.expand_dims(xp, axis=0)
    if not (x.shape[0] == xp.shape[0]):
        raise TypeError('Must be same # points or shape.')
    if not (len(x.shape) == 2 and len(xp.shape) == 2):
        raise TypeError('Wrong dimensionality of input.')
    if not (x.shape[1] == 3 and xp.shape[1] == 3):
        raise TypeError('Coords must be homogenous.')
    npt = x.shape[0]
    if ret_error:
        dst = np.empty((npt, 1))
        _dlt_reprojection_error(P0, P1, npt, x, xp, dst)
    else:
        dst = np.empty((npt, 4))
        _dlt_triangulate(P0, P1, npt, x, xp, dst)
    check()
    return dst


def dlt_reprojection_error(P0, P1, x, xp):
    return dlt_triangulate(P0, P1, x, xp, ret_error=True)


# ==================================================================================
# RANSAC hypothesis scoring (the inner loops of reference src/RansacFitter.h:59-95)
# ==================================================================================
_spv_dlt_score_hypotheses = clib.spv_dlt_score_hypotheses
_spv_dlt_score_hypotheses.restype = ct.c_int
_spv_dlt_score_hypotheses.argtypes = [ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                                      ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                                      ct.c_int, ct.c_int,
                                      ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                                      ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                                      ct.c_double,
                                      ndpointer(ct.c_int32, flags="C_CONTIGUOUS"),
                                      ct.c_void_p]


def dlt_score_hypotheses(P0, P1s, x, xp, max_error, return_mask=False):
    """
    Score candidate second cameras the way the reference's RANSAC does
    (reference src/RansacFitter.h:59-73): for every camera P1s[h] triangulate all
    correspondences against P0 and count the points whose reprojection error is
    <= `max_error` and which lie in front of both cameras.

    P0 float64 [3,4]; P1s float64 [H,3,4]; x, xp float64 [npt,3].
    Returns counts int32 [H] (and the inlier mask bool [H,npt] with `return_mask`).
    """
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1s = np.ascontiguousarray(P1s, dtype=np.float64)
    if P1s.ndim == 2:
        P1s = P1s[None]
    if not (P0.shape == (3, 4) and P1s.shape[1:] == (3, 4)):
        raise TypeError('P0,P1s must be camera matrices.')
    x = np.ascontiguousarray(x, dtype=np.float64)
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    if not (x.ndim == 2 and xp.ndim == 2 and x.shape == xp.shape and x.shape[1] == 3):
        raise TypeError('Coords must be homogenous [npt,3] pairs.')
    nhyp, npt = P1s.shape[0], x.shape[0]
    counts = np.zeros(nhyp, np.int32)
    mask = np.zeros((nhyp, npt), np.uint8) if return_mask else None
    check(_spv_dlt_score_hypotheses(P0, P1s.reshape(-1), nhyp, npt, x, xp, float(max_error), counts,
                                    mask.ctypes.data if return_mask else None))
    return (counts, mask.astype(bool)) if return_mask else counts


# ==================================================================================
# RANSAC candidate processing (reference src/RansacFitter.h:42-95 + src/Camera.h:31-46)
# ==================================================================================
_spv_ransac_process = clib.spv_ransac_process_candidates
_spv_ransac_process.restype = ct.c_int
_f64 = ndpointer(ct.c_double, flags="C_CONTIGUOUS")
_i32 = ndpointer(ct.c_int32, flags="C_CONTIGUOUS")
_spv_ransac_process.argtypes = [_f64, ct.c_int, _f64, _f64, ct.c_int, ct.c_double, ct.c_double, ct.c_double, ct.c_int,
                                _i32, _i32, _i32, _f64, _f64, _f64, _i32, ct.c_void_p]


def process_fundamental_matrices(Fs, x0, x1, options={'required_percent_inliers': .9,
                                                     'reprojection_error_allowed': .5,
                                                     'find_best_even_in_failure': True,
                                                     'singular_value_ratio_allowed': 3e-2}, return_mask=False):
    """
    What the reference's `ransac_fitter` does with every candidate fundamental matrix of a trial
    (`RansacFitter::process_fundamental_matrix`, reference src/RansacFitter.h:42-95), for a batch of
    candidates at once: singular-value-ratio gate, E = U diag(1,1,0) V^T, the four candidate
    second cameras of E (`Essential2Cameras`, src/Camera.h:31-46), every camera scored over all
    correspondences against [I | 0], the best camera kept.  The option names and defaults are the
    reference front-end's (spectavi/mvg.py:138-143).

    Fs float64 [nF,3,3] (or [3,3]); x0, x1 float64 [npt,3] homogeneous.
    Returns a dict of arrays over the candidates: success bool [nF], inlier_count int32 [nF],
    inlier_percent float64 [nF], camera float64 [nF,3,4] (zeros where no camera qualified),
    best_camera int32 [nF] (0..3, -1 = none), essential float64 [nF,3,3] (NaN where gated),
    singular_value_ratio float64 [nF], counts4 int32 [nF,4] (-1 where gated) and, with
    `return_mask`, inlier_mask bool [nF,npt] (np.flatnonzero of a row = the reference's inlier_idx).
    """
    Fs = np.ascontiguousarray(Fs, dtype=np.float64)
    if Fs.ndim == 2:
        Fs = Fs[None]
    if Fs.ndim != 3 or Fs.shape[1:] != (3, 3):
        raise TypeError('Fs must be [nF,3,3] fundamental matrices.')
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    if not (x0.ndim == 2 and x0.shape == x1.shape and x0.shape[1] == 3):
        raise TypeError('Coords must be homogenous [npt,3] pairs.')
    nF, npt = Fs.shape[0], x0.shape[0]
    if npt < 1:
        raise ValueError('Supplied no point matches.')
    success = np.zeros(nF, np.int32)
    count = np.zeros(nF, np.int32)
    best = np.full(nF, -1, np.int32)
    best_P = np.zeros((nF, 3, 4))
    ratio = np.zeros(nF)
    E = np.zeros((nF, 3, 3))
    counts4 = np.zeros((nF, 4), np.int32)
    mask = np.zeros((nF, npt), np.uint8) if return_mask else None
    check(_spv_ransac_process(Fs.reshape(-1), nF, x0, x1, npt, float(options['singular_value_ratio_allowed']),
                              float(options['required_percent_inliers']), float(options['reprojection_error_allowed']),
                              int(bool(options['find_best_even_in_failure'])), success, count, best,
                              best_P.reshape(-1), ratio, E.reshape(-1), counts4.reshape(-1),
                              mask.ctypes.data if return_mask else None))
    ret = {'success': success.astype(bool), 'inlier_count': count, 'inlier_percent': count / float(npt),
           'camera': best_P, 'best_camera': best, 'essential': E, 'singular_value_ratio': ratio, 'counts4': counts4}
    if return_mask:
        ret['inlier_mask'] = mask.astype(bool)
    return ret


# ==================================================================================
# seven_point_algorithm / ransac_fitter (reference spectavi/mvg.py:112-248)
# ==================================================================================
_seven_point_algorithm = clib.seven_point_algorithm
_seven_point_algorithm.restype = None
_seven_point_algorithm.argtypes = [ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                                   ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                                   ct.POINTER(ct.c_int),
                                   ndpointer(ct.c_double, flags="C_CONTIGUOUS")]


def seven_point_algorithm(x, xp):
    """
    The fundamental matrices through seven correspondences (reference spectavi/mvg.py:237-248).

    x, xp : float64 [7,2] euclidean or [7,3] homogeneous image points.
    Returns float64 [3*nroot, 3]: the nroot (0..3) solutions stacked, each with xp^T F x = 0.
    """
    if not (x.shape[0] == 7 and xp.shape[0] == 7):
        raise TypeError('Must be 7 points.')
    if not (x.shape[1] == 2 and xp.shape[1] == 2):
        x, xp = hnormalize(x), hnormalize(xp)
    x = np.ascontiguousarray(x, dtype=np.float64)
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    dst = np.empty((3, 3, 3))
    nroot = ct.c_int()
    _seven_point_algorithm(x, xp, ct.byref(nroot), dst)
    check()
    nroot = nroot.value
    return np.vstack(dst[:nroot]) if nroot else np.empty((0, 3))


_spv_seven_point = clib.spv_seven_point
_spv_seven_point.restype = ct.c_int
_spv_seven_point.argtypes = [_f64, _f64, ct.c_int, _i32, _f64, ct.c_void_p]


def seven_point_batch(x, xp, return_basis=False):
    """
    Seven-point algorithm for n independent 7-subsets at once.

    x, xp : float64 [n,7,2] euclidean.  Returns (nroot int32 [n], Fs float64 [n,3,3,3]) -- slots
    of missing roots are NaN -- and, with `return_basis`, the null-space pair [n,2,3,3].
    """
    x = np.ascontiguousarray(x, dtype=np.float64)
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    if not (x.ndim == 3 and x.shape[1:] == (7, 2) and xp.shape == x.shape):
        raise TypeError('x, xp must be [n,7,2].')
    n = x.shape[0]
    nroot = np.zeros(n, np.int32)
    Fs = np.zeros((n, 3, 3, 3))
    basis = np.zeros((n, 2, 3, 3)) if return_basis else None
    check(_spv_seven_point(x.reshape(-1), xp.reshape(-1), n, nroot, Fs.reshape(-1),
                           basis.ctypes.data if return_basis else None))
    return (nroot, Fs, basis) if return_basis else (nroot, Fs)


_ransac_fitter = clib.ransac_fitter
_ransac_fitter.restype = None
_ransac_fitter.argtypes = [ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                           ndpointer(ct.c_double, flags="C_CONTIGUOUS"),
                           ct.c_int,
                           ct.c_double,
                           ct.c_double,
                           ct.c_int,
                           ct.c_bool,
                           ct.c_double,
                           ct.c_bool,
                           ct.POINTER(ct.c_bool),
                           ct.POINTER(NdArray),
                           ct.POINTER(NdArray),
                           ct.POINTER(ct.c_double),
                           ct.POINTER(NdArray), ]


def ransac_fitter(x0, x1, options={'required_percent_inliers': .9,
                                   'reprojection_error_allowed': .5,
                                   'maximum_tries': 500,
                                   'find_best_even_in_failure': True,
                                   'singular_value_ratio_allowed': 3e-2,
                                   'progressbar': False}):
    """
    Fit a two-view geometry to tentatively corresponding points by RANSAC
    (reference spectavi/mvg.py:131-221; same option names and defaults).

    x0, x1 : float64 [npt,3] homogeneous normalised coordinates, npt >= 10.
    Returns the reference's dict: success, essential [3,3] (the winning seven-point solution),
    camera [3,4] (second camera; the first is [I | 0]), inlier_percent, inlier_idx int32 [n,1].
    If no model was kept essential and inlier_idx are empty and camera is [I | 0].
    Set SPECTAVI_RANSAC_SEED for reproducible subsets; `progressbar` is ignored.
    """
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    if not (x0.ndim == 2 and x0.shape == x1.shape and x0.shape[1] == 3):
        raise TypeError('Coords must be homogenous [npt,3] pairs.')
    npt = x0.shape[0]
    if npt < 10:
        raise ValueError('Supplied less than 10 point matches, unsupported.')
    success = ct.c_bool()
    essential = NdArray()
    camera = NdArray()
    inlier_idx = NdArray(dtype='int32')
    inlier_percent = ct.c_double()
    _ransac_fitter(x0, x1, npt, options['required_percent_inliers'],
                   options['reprojection_error_allowed'],
                   options['maximum_tries'],
                   options['find_best_even_in_failure'],
                   options['singular_value_ratio_allowed'],
                   options.get('progressbar', False),
                   ct.byref(success),
                   ct.byref(essential),
                   ct.byref(camera),
                   ct.byref(inlier_percent),
                   ct.byref(inlier_idx))
    check()
    return {'success': success.value,
            'essential': essential.asarray(),
            'camera': camera.asarray(),
            'inlier_percent': inlier_percent.value,
            'inlier_idx': inlier_idx.asarray(), }


_spv_ransac_sample = clib.spv_ransac_sample
_spv_ransac_sample.restype = ct.c_int
_spv_ransac_sample.argtypes = [ct.c_ulonglong, ct.c_int, ct.c_int, _i32]


def ransac_sample(seed, npt, ntries):
    """The 7-subsets int32 [ntries,7] that `ransac_fit(seed=seed)` evaluates (drawn as the reference's
    floyd_sample draws them, reference src/RansacFitter.h:120-132, from one mt19937)."""
    samples = np.zeros((ntries, 7), np.int32)
    check(_spv_ransac_sample(int(seed), int(npt), int(ntries), samples.reshape(-1)))
    return samples


_fit_out = [ct.POINTER(ct.c_int32), _f64, _f64, ct.POINTER(ct.c_double), _i32, ct.POINTER(ct.c_int32),
            ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]
_spv_ransac_fit = clib.spv_ransac_fit
_spv_ransac_fit.restype = ct.c_int
_spv_ransac_fit.argtypes = [_f64, _f64, ct.c_int, ct.c_double, ct.c_double, ct.c_int, ct.c_int, ct.c_double,
                            ct.c_ulonglong] + _fit_out
_spv_ransac_fit_samples = clib.spv_ransac_fit_samples
_spv_ransac_fit_samples.restype = ct.c_int
_spv_ransac_fit_samples.argtypes = [_f64, _f64, ct.c_int, ct.c_double, ct.c_double, _i32, ct.c_int, ct.c_int,
                                    ct.c_double] + _fit_out


def ransac_fit(x0, x1, required_percent_inliers=.9, reprojection_error_allowed=.5, maximum_tries=500,
               find_best_even_in_failure=True, singular_value_ratio_allowed=3e-2, seed=0, samples=None):
    """
    `ransac_fitter` with plain outputs and explicit control of the subsets: `samples` int32 [ntries,7]
    (then `maximum_tries` and `seed` are ignored) or `seed` (0 = SPECTAVI_RANSAC_SEED / random).

    Returns the `ransac_fitter` dict (essential / camera None and inlier_idx empty when no model was
    kept) plus best_try, best_root (which try and which of its roots won; -1 if none) and tries_run.
    """
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    if not (x0.ndim == 2 and x0.shape == x1.shape and x0.shape[1] == 3):
        raise TypeError('Coords must be homogenous [npt,3] pairs.')
    npt = x0.shape[0]
    if npt < 10:
        raise ValueError('Supplied less than 10 point matches, unsupported.')
    ok, n, bt, br, ran = ct.c_int32(0), ct.c_int32(0), ct.c_int32(-1), ct.c_int32(-1), ct.c_int32(0)
    pct = ct.c_double(0.0)
    F, P, idx = np.zeros(9), np.zeros(12), np.zeros(npt, np.int32)
    outs = (ct.byref(ok), F, P, ct.byref(pct), idx, ct.byref(n), ct.byref(bt), ct.byref(br), ct.byref(ran))
    if samples is not None:
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        if not (samples.ndim == 2 and samples.shape[1] == 7):
            raise TypeError('samples must be [ntries,7].')
        check(_spv_ransac_fit_samples(x0, x1, npt, float(required_percent_inliers), float(reprojection_error_allowed),
                                      samples.reshape(-1), samples.shape[0], int(bool(find_best_even_in_failure)),
                                      float(singular_value_ratio_allowed), *outs))
    else:
        check(_spv_ransac_fit(x0, x1, npt, float(required_percent_inliers), float(reprojection_error_allowed),
                              int(maximum_tries), int(bool(find_best_even_in_failure)),
                              float(singular_value_ratio_allowed), int(seed), *outs))
    found = bt.value >= 0
    return {'success': bool(ok.value), 'essential': F.reshape(3, 3) if found else None,
            'camera': P.reshape(3, 4) if found else None, 'inlier_percent': float(pct.value),
            'inlier_idx': idx[:n.value].copy(), 'best_try': bt.value, 'best_root': br.value,
            'tries_run': ran.value}
