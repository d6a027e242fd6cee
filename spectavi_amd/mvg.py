"""
``spectavi_amd.mvg``
====================
Multi-view-geometry front-end: the batched DLT triangulator of the reference's
``spectavi.mvg`` (reference spectavi/mvg.py:259-306), bound to the gfx950 build
of ``libspectavi.so``.
"""
import ctypes as ct

import numpy as np
from numpy.ctypeslib import ndpointer

from spectavi_amd._lib import clib, check


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
