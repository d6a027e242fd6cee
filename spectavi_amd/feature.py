"""
``spectavi_amd.feature``
========================
Descriptor matching front-end, the counterpart of the hot-path functions of the
reference's ``spectavi.feature`` (reference spectavi/feature.py:234-243,
292-304, 346-407): same names, arguments, return types and error behaviour,
bound with ctypes to the gfx950 build of ``libspectavi.so``.
"""
import ctypes as ct

import numpy as np
from numpy.ctypeslib import ndpointer

from spectavi_amd._lib import clib, check
from spectavi_amd.ndarray import NdArray

# ==================================================================================
# brute-force L1, k = 2       (reference spectavi/feature.py:234-243)
# ==================================================================================
_nn_bruteforcel1k2 = clib.nn_bruteforcel1k2
_nn_bruteforcel1k2.restype = None
_nn_bruteforcel1k2.argtypes = [ndpointer(ct.c_ubyte, flags="C_CONTIGUOUS"),
                               ndpointer(ct.c_ubyte, flags="C_CONTIGUOUS"),
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.POINTER(NdArray),
                               ct.POINTER(NdArray), ]


def nn_bruteforcel1k2(x, y, nthreads=1):
    """
    Exact L1 nearest neighbours with k=2 of every row of `y` (queries) among the
    rows of `x` (database); inputs are unsigned bytes with a row length that is
    a multiple of 16 (reference spectavi/feature.py:292-304).

    Returns
    -------
    nn_idx : uint64 ndarray [yrows, 2]   index into `x`, nearest first
    nn_dist : int32 ndarray [yrows, 2]   L1 distances, ascending

    `nthreads` is accepted for signature compatibility; the GPU path ignores it.
    """
    xrows, xdim = x.shape
    yrows, ydim = y.shape
    assert ydim == xdim
    dim = xdim
    if dim % 16 != 0:
        # the reference throws std::runtime_error through extern "C" here
        # (src/BruteForceNnL1K2.h:77-81), which aborts the interpreter
        raise ValueError("Input matrix inner dimensions must be 16-byte aligned.")
    nn_idx = NdArray(dtype='uint64')
    nn_dist = NdArray(dtype='int32')
    _nn_bruteforcel1k2(x, y, xrows, yrows, dim, nthreads, ct.byref(nn_idx), ct.byref(nn_dist))
    check()
    return nn_idx.asarray(), nn_dist.asarray()


# ==================================================================================
# cascading hash              (reference spectavi/feature.py:346-376)
# ==================================================================================
_nn_cascading_hash = clib.nn_cascading_hash
_nn_cascading_hash.restype = None
_nn_cascading_hash.argtypes = [ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                               ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.c_int,
                               ct.POINTER(NdArray),
                               ct.POINTER(NdArray), ]

_spv_nn_cascading_hash = clib.spv_nn_cascading_hash
_spv_nn_cascading_hash.restype = ct.c_int
_spv_nn_cascading_hash.argtypes = [ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                                   ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                                   ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int,
                                   ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                                   ndpointer(ct.c_uint64, flags="C_CONTIGUOUS"),
                                   ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                                   ct.c_void_p]

_spv_generate_hash_dict = clib.spv_generate_hash_dict
_spv_generate_hash_dict.restype = ct.c_int
_spv_generate_hash_dict.argtypes = [ct.c_uint32, ct.c_int, ct.c_int, ct.c_int,
                                    ndpointer(ct.c_float, flags="C_CONTIGUOUS")]


def auto_hash_bit_rate(xrows, yrows):
    """`m` auto-tune of the reference: ~6 points per hash code
    (reference spectavi/feature.py:364-367)."""
    mrows = max([xrows, yrows])
    return int(np.floor(np.log2(mrows / 6.)))


def nn_cascading_hash(x, y, k=2, m=None, n=2, g=2):
    """
    Approximate L1 2-NN through a cascade of `n` random-hyperplane hash tables of
    `m` bits probed at the `g` least-confident bits, then exact L1 over the
    candidates (reference spectavi/feature.py:360-376).  `x`, `y` are float32,
    integer-valued in [-128,127] (see `normalize_to_ubyte_and_multiple_16_dim`).

    Returns (uint64 [yrows,k], float32 [yrows,k]); with `m=None` and fewer than
    ~96 rows the reference falls back to exact brute force on the +128 shifted
    bytes and returns int32 distances -- reproduced here.
    """
    xrows, xdim = x.shape
    yrows, ydim = y.shape
    assert ydim == xdim
    if m is None:  # auto-tune `m` if specified with None
        m = auto_hash_bit_rate(xrows, yrows)
        if m < 4:
            # using hashes is not appropriate:
            return nn_bruteforcel1k2((x + 128).astype('uint8'),
                                     (y + 128).astype('uint8'), nthreads=8)
    dim = xdim
    if k != 2:
        raise ValueError("nn_cascading_hash: only k=2 is defined (the reference writes two columns)")
    if dim % 16 != 0:
        raise ValueError("Input matrix inner dimensions must be 16-byte aligned.")
    if not (1 <= m <= 31):
        raise ValueError("hash bit rate m must be in [1, 31]")
    if n < 1 or not (0 <= g <= min(m, 16)):
        raise ValueError("need n >= 1 and 0 <= g <= min(m, 16)")
    cashash_idx = NdArray(dtype='uint64')
    cashash_dist = NdArray(dtype='float32')
    _nn_cascading_hash(x, y, xrows, yrows, dim, k, m, n, g, ct.byref(cashash_idx), ct.byref(cashash_dist))
    check()
    return cashash_idx.asarray(), cashash_dist.asarray()


def generate_hash_dict(seed, dim, m, n):
    """float32 [n, dim, m] hyperplanes exactly as `nn_cascading_hash` draws them
    for a given std::mt19937 seed (reference src/CascadingHashNn.h:86-100)."""
    d = np.empty((n, dim, m), np.float32)
    check(_spv_generate_hash_dict(int(seed) & 0xFFFFFFFF, dim, m, n, d))
    return d


def nn_cascading_hash_with_dict(x, y, hash_dict, g=2, return_ncand=False):
    """`nn_cascading_hash` with explicit hyperplanes `hash_dict` float32 [n, dim, m]
    (reproducible: the reference seeds from std::random_device)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    hash_dict = np.ascontiguousarray(hash_dict, dtype=np.float32)
    xrows, dim = x.shape
    yrows, ydim = y.shape
    assert ydim == dim
    n, ddim, m = hash_dict.shape
    assert ddim == dim
    idx = np.empty((yrows, 2), np.uint64)
    dist = np.empty((yrows, 2), np.float32)
    ncand = np.zeros(yrows, np.int32)
    check(_spv_nn_cascading_hash(x, y, xrows, yrows, dim, m, n, g, hash_dict, idx, dist,
                                 ncand.ctypes.data if return_ncand else None))
    if return_ncand:
        return idx, dist, ncand
    return idx, dist


# ==================================================================================
# normalization               (reference spectavi/feature.py:384-407)
# ==================================================================================
def normalize_to_ubyte_and_multiple_16_dim(x, dtype='float32'):
    """
    Normalize a data matrix to:
    - have zero mean for each column
    - be in the range [-128,127]
    - have a column count that is a multiple of 16 (zero padded)
    - the required `dtype`
    for use with `nn_cascading_hash` (expects [-128,127]) and, after `+128`,
    `nn_bruteforcel1k2` (expects [0,255]).
    """
    x0 = x - np.mean(x, axis=0, keepdims=True)  # de-mean
    max_per_col = np.max(x0, axis=0, keepdims=True)
    min_per_col = np.min(x0, axis=0, keepdims=True)
    norm = np.max(np.stack([max_per_col, -min_per_col]), axis=0)
    x0 = np.round(x0 / norm * 128)
    x0[x0 > 127] = 127
    x0[x0 < -128] = -128
    xrows, dim = x0.shape
    new_dim = int(np.ceil(dim / 16.) * 16)
    xx = np.zeros([xrows, new_dim])
    xx[:, :dim] = x0
    return xx.astype(dtype)


# ==================================================================================
# ratio test + match compaction (reference example/ex01_essential_estimation.py:102-106)
# ==================================================================================
_spv_ratio_test = clib.spv_ratio_test
_spv_ratio_test.restype = ct.c_int
_spv_ratio_test.argtypes = [ndpointer(ct.c_uint64, flags="C_CONTIGUOUS"), ct.c_void_p, ct.c_int, ct.c_int,
                            ct.c_double, ndpointer(ct.c_int32, flags="C_CONTIGUOUS"),
                            ct.POINTER(ct.c_int32)]


def ratio_test_matches(nn_idx, nn_dist, min_ratio):
    """
    The match filter of the reference's pipeline, on the GPU:
    ``pass = nn_dist[:,1] / nn_dist[:,0].astype('float64') >= min_ratio`` and the
    compaction ``(where(pass), nn_idx[pass, 0])``.  Returns int32 [nmatch, 2] rows
    (query row, database row) in ascending query order.  Queries without any
    neighbour never pass.
    """
    nn_idx = np.ascontiguousarray(nn_idx, dtype=np.uint64)
    if nn_dist.dtype == np.float32:
        is_float, nn_dist = 1, np.ascontiguousarray(nn_dist)
    else:
        is_float, nn_dist = 0, np.ascontiguousarray(nn_dist, dtype=np.int32)
    yrows = nn_idx.shape[0]
    assert nn_idx.shape == (yrows, 2) and nn_dist.shape == (yrows, 2)
    matches = np.empty((yrows, 2), np.int32)
    count = ct.c_int32(0)
    check(_spv_ratio_test(nn_idx, nn_dist.ctypes.data, is_float, yrows, float(min_ratio), matches,
                          ct.byref(count)))
    return matches[:count.value].copy()


# ==================================================================================
# SIFT table adapter (reference src/Sift.h:13,115-123: rows of 132 floats)
# ==================================================================================
_spv_sift_split = clib.spv_sift_split
_spv_sift_split.restype = ct.c_int
_spv_sift_split.argtypes = [ndpointer(ct.c_float, flags="C_CONTIGUOUS"), ct.c_int,
                            ndpointer(ct.c_float, flags="C_CONTIGUOUS"),
                            ndpointer(ct.c_ubyte, flags="C_CONTIGUOUS")]


def split_sift_table(table):
    """Split the reference's SIFT table (float32 [n,132] = x, y, sigma, angle + 128 descriptor
    values that are uint8(512*d) stored as float) into (geometry float32 [n,4], descriptors
    uint8 [n,128]) so that `nn_bruteforcel1k2` runs on the true 128-D bytes."""
    table = np.ascontiguousarray(table, dtype=np.float32)
    if table.ndim != 2 or table.shape[1] != 132:
        raise TypeError("SIFT table must be [n, 132]")
    n = table.shape[0]
    geom = np.empty((n, 4), np.float32)
    desc = np.empty((n, 128), np.uint8)
    check(_spv_sift_split(table, n, geom, desc))
    return geom, desc


# ==================================================================================
# normalisation on device (same result as normalize_to_ubyte_and_multiple_16_dim above)
# ==================================================================================
_spv_normalize = clib.spv_normalize
_spv_normalize.restype = ct.c_int
_spv_normalize.argtypes = [ndpointer(ct.c_float, flags="C_CONTIGUOUS"), ct.c_int, ct.c_int, ct.c_void_p,
                           ct.c_void_p]


def normalize_to_ubyte_and_multiple_16_dim_gpu(x, want_ubyte=False):
    """`normalize_to_ubyte_and_multiple_16_dim(x)` for a float32 `x`, computed on the GPU and
    bit-identical to the numpy version; with `want_ubyte` also returns the
    `(out + 128).astype('uint8')` image that `nn_bruteforcel1k2` takes.  A single-column
    table (which numpy sums pairwise instead of row by row) is refused with SpectaviError."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, dim = x.shape
    dim16 = int(np.ceil(dim / 16.) * 16)
    out = np.empty((rows, dim16), np.float32)
    u8 = np.empty((rows, dim16), np.uint8) if want_ubyte else None
    check(_spv_normalize(x, rows, dim, out.ctypes.data, u8.ctypes.data if want_ubyte else None))
    return (out, u8) if want_ubyte else out
