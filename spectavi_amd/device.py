"""Device-resident entry points: torch tensors in HBM in, torch tensors out.

Thin binding of section 3 of include/spectavi_amd.h (spv_*_device).  torch is
used only for device memory and the current HIP stream; every kernel launched
here is hand-written HIP inside libspectavi.so and is enqueued on torch's
current stream, so `torch.cuda.Event` timing and stream ordering apply.
"""
import ctypes as ct

import numpy as np
import torch

from spectavi_amd._lib import clib, check

_vp = ct.c_void_p

clib.spv_l1k2_workspace_bytes.restype = ct.c_size_t
clib.spv_l1k2_workspace_bytes.argtypes = [ct.c_int, ct.c_int, ct.c_int]
clib.spv_l1k2_device.restype = ct.c_int
clib.spv_l1k2_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, _vp, _vp, _vp, ct.c_size_t, _vp]
clib.spv_cascade_workspace_bytes.restype = ct.c_size_t
clib.spv_cascade_workspace_bytes.argtypes = [ct.c_int] * 6
clib.spv_cascade_device.restype = ct.c_int
clib.spv_cascade_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int,
                                    _vp, _vp, _vp, _vp, _vp, ct.c_size_t, _vp]
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
clib.spv_dlt_triangulate_device.restype = ct.c_int
clib.spv_dlt_triangulate_device.argtypes = [_f64p, _f64p, ct.c_longlong, _vp, _vp, _vp, _vp]
clib.spv_dlt_reprojection_error_device.restype = ct.c_int
clib.spv_dlt_reprojection_error_device.argtypes = [_f64p, _f64p, ct.c_longlong, _vp, _vp, _vp, _vp]
clib.spv_profile_enable.restype = None
clib.spv_profile_enable.argtypes = [ct.c_int]
clib.spv_profile_read.restype = ct.c_int
clib.spv_profile_read.argtypes = [ct.c_char_p, ct.POINTER(ct.c_longlong), ct.POINTER(ct.c_double)]
clib.spv_profile_reset.restype = None
clib.spv_profile_reset.argtypes = []


def _stream(dev=None):
    return ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _on_device_of:
    """All tensors of one call must live on ONE GPU; the call then runs with that GPU as the current
    device and on torch's current stream OF THAT GPU, whatever the process-wide current device is
    (the library launches on the calling thread's current HIP device), and the previous device is
    restored afterwards."""

    def __init__(self, *tensors):
        devs = {t.device for t in tensors if t is not None}
        if len(devs) != 1:
            raise ValueError("all tensors of one call must live on the same GPU, got %s" % sorted(map(str, devs)))
        self.device = devs.pop()
        self._ctx = torch.cuda.device(self.device)

    def __enter__(self):
        self._ctx.__enter__()
        return _stream(self.device)

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def _need(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError("%s must be a contiguous %s tensor on the GPU" % (name, dtype))


class Workspace:
    """Grow-only scratch buffer reused across calls (one per device)."""

    def __init__(self):
        self._buf = None

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        if self._buf is None or self._buf.numel() < nbytes or self._buf.device != device:
            self._buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._buf


_default_ws = Workspace()


def l1k2(x, y, workspace=None):
    """Exact L1 2-NN on device: x uint8 [M,D], y uint8 [N,D] (CUDA tensors).
    Returns (idx int64 [N,2] -- the ABI's size_t bits, -1 = no neighbour --,
    dist int32 [N,2]).  Asynchronous on the current stream."""
    _need(x, torch.uint8, "x")
    _need(y, torch.uint8, "y")
    xrows, dim = x.shape
    yrows, ydim = y.shape
    assert dim == ydim
    if dim % 16 != 0:
        raise ValueError("Input matrix inner dimensions must be 16-byte aligned.")
    idx = torch.empty((yrows, 2), dtype=torch.int64, device=y.device)
    dist = torch.empty((yrows, 2), dtype=torch.int32, device=y.device)
    nbytes = clib.spv_l1k2_workspace_bytes(xrows, yrows, dim)
    with _on_device_of(x, y) as stream:
        ws = (workspace or _default_ws).get(nbytes, y.device)
        check(clib.spv_l1k2_device(x.data_ptr(), y.data_ptr(), xrows, yrows, dim, idx.data_ptr(),
                                   dist.data_ptr(), ws.data_ptr(), ws.numel(), stream))
    return idx, dist


def shard_bounds(total, shards):
    """[lo_0, lo_1, ..., total]: the contiguous balanced shards the library itself uses."""
    clib.spv_shard_lo.restype = ct.c_longlong
    clib.spv_shard_lo.argtypes = [ct.c_longlong, ct.c_int, ct.c_int]
    return [int(clib.spv_shard_lo(int(total), int(shards), r)) for r in range(int(shards) + 1)]


def l1k2_gathered(xs, ys, transport="rccl"):
    """The sharded query loop inside ONE process (reference loop
    src/BruteForceNnL1K2.h:92-93): xs[r] = replica of the database on GPU r, ys[r] = query shard r
    (contiguous balanced shards, `shard_bounds`), one tensor per listed GPU.  Every GPU matches its
    shard; the 16-byte (idx0, idx1, d0, d1) records are gathered on ys[0]'s GPU by ncclGather on the
    library's ncclCommInitAll clique ("rccl") or by peer copies ("copy") and widened there.
    Returns (idx int64 [N,2], dist int32 [N,2]) on that GPU.  Synchronous."""
    if len(xs) != len(ys) or not xs:
        raise ValueError("one database replica and one query shard per GPU")
    G = len(xs)
    for x, y in zip(xs, ys):
        _need(x, torch.uint8, "x")
        _need(y, torch.uint8, "y")
        if x.device != y.device:
            raise ValueError("replica and shard of one rank must share a GPU")
    xrows, dim = xs[0].shape
    if any(tuple(x.shape) != (xrows, dim) for x in xs) or any(y.shape[1] != dim for y in ys):
        raise ValueError("database replicas must be identical in shape; all rows %d wide" % dim)
    total = sum(int(y.shape[0]) for y in ys)
    if [int(y.shape[0]) for y in ys] != [b - a for a, b in zip(shard_bounds(total, G)[:-1], shard_bounds(total, G)[1:])]:
        raise ValueError("query shards must be the contiguous balanced split of the %d queries" % total)
    root = ys[0].device
    idx = torch.empty((total, 2), dtype=torch.int64, device=root)
    dist = torch.empty((total, 2), dtype=torch.int32, device=root)
    devs = (ct.c_int * G)(*[y.device.index for y in ys])
    px = (ct.c_void_p * G)(*[x.data_ptr() for x in xs])
    py = (ct.c_void_p * G)(*[y.data_ptr() for y in ys])
    for y in ys:
        torch.cuda.synchronize(y.device)  # the library runs on streams of its own
    clib.spv_l1k2_gathered_device.restype = ct.c_int
    clib.spv_l1k2_gathered_device.argtypes = [ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.c_void_p),
                                              ct.POINTER(ct.c_void_p), ct.c_int, ct.c_longlong, ct.c_int, _vp, _vp, ct.c_int]
    mode = {"rccl": 1, "copy": 2}[transport]
    check(clib.spv_l1k2_gathered_device(G, devs, px, py, xrows, total, dim, idx.data_ptr(), dist.data_ptr(), mode))
    return idx, dist


def _gather_lists(G, per_rank, total):
    if [int(t.shape[0]) for t in per_rank] != [b - a for a, b in zip(shard_bounds(total, G)[:-1], shard_bounds(total, G)[1:])]:
        raise ValueError("the per-GPU shards must be the contiguous balanced split of the %d rows" % total)
    for t in per_rank:
        torch.cuda.synchronize(t.device)  # the library runs on streams of its own


def cascade_gathered(xs, ys, dicts, g=2, transport="rccl", want_ncand=False):
    """nn_cascading_hash sharded inside ONE process with everything resident: per GPU a
    replica of the database xs[r] and of the hyperplanes dicts[r] (float32 [n,D,m]) and the query shard
    ys[r]; results gathered (16-byte records, ncand as int32) and widened on ys[0]'s GPU."""
    G = len(xs)
    if not (G == len(ys) == len(dicts)) or not xs:
        raise ValueError("one database replica, one dictionary replica and one query shard per GPU")
    for x, y, d in zip(xs, ys, dicts):
        _need(x, torch.float32, "x")
        _need(y, torch.float32, "y")
        _need(d, torch.float32, "hash_dict")
        if not (x.device == y.device == d.device):
            raise ValueError("the tensors of one rank must share a GPU")
    xrows, dim = xs[0].shape
    n, ddim, m = dicts[0].shape
    total = sum(int(y.shape[0]) for y in ys)
    _gather_lists(G, ys, total)
    root = ys[0].device
    idx = torch.empty((total, 2), dtype=torch.int64, device=root)
    dist = torch.empty((total, 2), dtype=torch.float32, device=root)
    ncand = torch.empty((total,), dtype=torch.int32, device=root) if want_ncand else None
    devs = (ct.c_int * G)(*[y.device.index for y in ys])
    vp = ct.c_void_p * G
    clib.spv_cascade_gathered_device.restype = ct.c_int
    clib.spv_cascade_gathered_device.argtypes = [ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p),
                                                 ct.c_int, ct.c_longlong, ct.c_int, ct.c_int, ct.c_int, ct.c_int,
                                                 ct.POINTER(ct.c_void_p), _vp, _vp, _vp, ct.c_int]
    check(clib.spv_cascade_gathered_device(G, devs, vp(*[x.data_ptr() for x in xs]), vp(*[y.data_ptr() for y in ys]), xrows,
                                           total, dim, m, n, g, vp(*[d.data_ptr() for d in dicts]), idx.data_ptr(),
                                           dist.data_ptr(), ncand.data_ptr() if want_ncand else None,
                                           {"rccl": 1, "copy": 2}[transport]))
    return (idx, dist, ncand) if want_ncand else (idx, dist)


def dlt_gathered(P0, P1, xs, xps, want_error=False, transport="rccl"):
    """dlt_triangulate / dlt_reprojection_error sharded inside ONE process with the point shards
    resident (xs[r], xps[r] float64 [cnt_r,3] on GPU r); rows gathered on xs[0]'s GPU."""
    G = len(xs)
    if G != len(xps) or not xs:
        raise ValueError("one shard of each view per GPU")
    for x, xp in zip(xs, xps):
        _need(x, torch.float64, "x")
        _need(xp, torch.float64, "xp")
        if x.device != xp.device or x.shape != xp.shape:
            raise ValueError("the two views of one rank must share a GPU and a shape")
    total = sum(int(x.shape[0]) for x in xs)
    _gather_lists(G, xs, total)
    out = torch.empty((total, 1 if want_error else 4), dtype=torch.float64, device=xs[0].device)
    devs = (ct.c_int * G)(*[x.device.index for x in xs])
    vp = ct.c_void_p * G
    clib.spv_dlt_gathered_device.restype = ct.c_int
    clib.spv_dlt_gathered_device.argtypes = [ct.c_int, ct.POINTER(ct.c_int), _f64p, _f64p, ct.c_longlong,
                                             ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p), _vp, ct.c_int, ct.c_int]
    check(clib.spv_dlt_gathered_device(G, devs, np.ascontiguousarray(P0, np.float64), np.ascontiguousarray(P1, np.float64), total,
                                       vp(*[x.data_ptr() for x in xs]), vp(*[xp.data_ptr() for xp in xps]), out.data_ptr(),
                                       int(bool(want_error)), {"rccl": 1, "copy": 2}[transport]))
    return out


def cascade(x, y, hash_dict, g=2, workspace=None, want_ncand=False):
    """Cascade-hash 2-NN on device: x,y float32 [rows,D]; hash_dict float32 [n,D,m]."""
    _need(x, torch.float32, "x")
    _need(y, torch.float32, "y")
    _need(hash_dict, torch.float32, "hash_dict")
    xrows, dim = x.shape
    yrows, ydim = y.shape
    n, ddim, m = hash_dict.shape
    assert dim == ydim == ddim
    idx = torch.empty((yrows, 2), dtype=torch.int64, device=y.device)
    dist = torch.empty((yrows, 2), dtype=torch.float32, device=y.device)
    ncand = torch.empty((yrows,), dtype=torch.int32, device=y.device) if want_ncand else None
    nbytes = clib.spv_cascade_workspace_bytes(xrows, yrows, dim, m, n, g)
    with _on_device_of(x, y, hash_dict) as stream:
        ws = (workspace or _default_ws).get(nbytes, y.device)
        check(clib.spv_cascade_device(x.data_ptr(), y.data_ptr(), xrows, yrows, dim, m, n, g,
                                      hash_dict.data_ptr(), idx.data_ptr(), dist.data_ptr(),
                                      ncand.data_ptr() if want_ncand else None, ws.data_ptr(),
                                      ws.numel(), stream))
    return (idx, dist, ncand) if want_ncand else (idx, dist)


def _dlt(fn, P0, P1, x, xp, cols):
    _need(x, torch.float64, "x")
    _need(xp, torch.float64, "xp")
    assert x.shape == xp.shape and x.shape[1] == 3
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    assert P0.shape == (3, 4) and P1.shape == (3, 4)
    npt = x.shape[0]
    dst = torch.empty((npt, cols), dtype=torch.float64, device=x.device)
    with _on_device_of(x, xp) as stream:
        check(fn(P0, P1, npt, x.data_ptr(), xp.data_ptr(), dst.data_ptr(), stream))
    return dst


def dlt_triangulate(P0, P1, x, xp):
    """P0,P1 host float64 [3,4]; x,xp CUDA float64 [npt,3] -> CUDA float64 [npt,4]."""
    return _dlt(clib.spv_dlt_triangulate_device, P0, P1, x, xp, 4)


def dlt_reprojection_error(P0, P1, x, xp):
    return _dlt(clib.spv_dlt_reprojection_error_device, P0, P1, x, xp, 1)


clib.spv_ratio_test_workspace_bytes.restype = ct.c_size_t
clib.spv_ratio_test_workspace_bytes.argtypes = [ct.c_int]
clib.spv_ratio_test_device.restype = ct.c_int
clib.spv_ratio_test_device.argtypes = [_vp, _vp, ct.c_int, ct.c_int, ct.c_double, _vp, _vp, _vp, ct.c_size_t, _vp]
clib.spv_dlt_score_hypotheses_device.restype = ct.c_int
clib.spv_dlt_score_hypotheses_device.argtypes = [_f64p, _vp, ct.c_int, ct.c_longlong, _vp, _vp, ct.c_double,
                                                 _vp, _vp, _vp]
clib.spv_dlt_score_workspace_bytes.restype = ct.c_size_t
clib.spv_dlt_score_workspace_bytes.argtypes = [ct.c_int, ct.c_longlong]
clib.spv_dlt_score_hypotheses_device_ws.restype = ct.c_int
clib.spv_dlt_score_hypotheses_device_ws.argtypes = [_f64p, _vp, ct.c_int, ct.c_longlong, _vp, _vp, ct.c_double,
                                                    _vp, _vp, _vp, ct.c_size_t, _vp]


def ratio_test(idx, dist, min_ratio, workspace=None):
    """Ratio test + ordered compaction on device.  idx int64 [N,2], dist int32/float32 [N,2]
    (outputs of l1k2 / cascade).  Returns (matches int32 [N,2] capacity, count int32 [1]);
    rows [0, count) are (query row, database row).  Asynchronous."""
    _need(idx, torch.int64, "idx")
    if dist.dtype == torch.float32:
        is_float = 1
    else:
        _need(dist, torch.int32, "dist")
        is_float = 0
    n = idx.shape[0]
    matches = torch.empty((n, 2), dtype=torch.int32, device=idx.device)
    count = torch.zeros((1,), dtype=torch.int32, device=idx.device)
    with _on_device_of(idx, dist) as stream:
        ws = (workspace or _default_ws).get(clib.spv_ratio_test_workspace_bytes(n), idx.device)
        check(clib.spv_ratio_test_device(idx.data_ptr(), dist.data_ptr(), is_float, n, float(min_ratio),
                                         matches.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                         stream))
    return matches, count


def dlt_score_hypotheses(P0, P1s, x, xp, max_error, want_mask=False, workspace=None):
    """RANSAC scoring on device: P0 host [3,4]; P1s CUDA float64 [H,3,4]; x,xp CUDA [npt,3].
    Returns counts int32 [H] (and mask uint8 [H,npt])."""
    _need(P1s, torch.float64, "P1s")
    _need(x, torch.float64, "x")
    _need(xp, torch.float64, "xp")
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    nh, npt = P1s.shape[0], x.shape[0]
    counts = torch.empty((nh,), dtype=torch.int32, device=x.device)
    mask = torch.empty((nh, npt), dtype=torch.uint8, device=x.device) if want_mask else None
    with _on_device_of(P1s, x, xp) as stream:
        ws = (workspace or _default_ws).get(clib.spv_dlt_score_workspace_bytes(nh, npt), x.device)
        check(clib.spv_dlt_score_hypotheses_device_ws(P0, P1s.data_ptr(), nh, npt, x.data_ptr(), xp.data_ptr(),
                                                      float(max_error), counts.data_ptr(),
                                                      mask.data_ptr() if want_mask else None, ws.data_ptr(),
                                                      ws.numel(), stream))
    return (counts, mask) if want_mask else counts


clib.spv_sift_split_device.restype = ct.c_int
clib.spv_sift_split_device.argtypes = [_vp, ct.c_int, _vp, _vp, _vp]
clib.spv_gather_match_coords_device.restype = ct.c_int
clib.spv_gather_match_coords_device.argtypes = [_vp, _vp, _vp, _vp, ct.c_int, _vp, _vp, _vp]


def split_sift_table(table):
    """CUDA float32 [n,132] -> (geom float32 [n,4], desc uint8 [n,128])."""
    _need(table, torch.float32, "table")
    assert table.shape[1] == 132
    n = table.shape[0]
    geom = torch.empty((n, 4), dtype=torch.float32, device=table.device)
    desc = torch.empty((n, 128), dtype=torch.uint8, device=table.device)
    with _on_device_of(table) as stream:
        check(clib.spv_sift_split_device(table.data_ptr(), n, geom.data_ptr(), desc.data_ptr(), stream))
    return geom, desc


def match_coordinates(geom_x, geom_y, matches, count):
    """Homogeneous float64 coordinates of matched keypoints: (x0 [cap,3] from geom_x[database row],
    x1 [cap,3] from geom_y[query row]); rows [0, count) are valid."""
    _need(geom_x, torch.float32, "geom_x")
    _need(geom_y, torch.float32, "geom_y")
    _need(matches, torch.int32, "matches")
    cap = matches.shape[0]
    x0 = torch.zeros((cap, 3), dtype=torch.float64, device=matches.device)
    x1 = torch.zeros((cap, 3), dtype=torch.float64, device=matches.device)
    with _on_device_of(geom_x, geom_y, matches, count) as stream:
        check(clib.spv_gather_match_coords_device(geom_x.data_ptr(), geom_y.data_ptr(), matches.data_ptr(),
                                                  count.data_ptr(), cap, x0.data_ptr(), x1.data_ptr(), stream))
    return x0, x1


def profile_enable(on=True):
    """Bracket the hot kernels with HIP events on their launch stream."""
    clib.spv_profile_enable(1 if on else 0)


def profile_reset():
    clib.spv_profile_reset()


def profile_read(name):
    """(launch count, total milliseconds) of the named kernel since the last reset.
    Synchronises with the recorded events."""
    n = ct.c_longlong(0)
    ms = ct.c_double(0.0)
    check(clib.spv_profile_read(name.encode(), ct.byref(n), ct.byref(ms)))
    return int(n.value), float(ms.value)


clib.spv_ransac_workspace_bytes.restype = ct.c_size_t
clib.spv_ransac_workspace_bytes.argtypes = [ct.c_int, ct.c_longlong, ct.c_int]
clib.spv_ransac_process_candidates_device.restype = ct.c_int
clib.spv_ransac_process_candidates_device.argtypes = [_vp, ct.c_int, ct.c_longlong, _vp, _vp, ct.c_double, ct.c_double,
                                                      ct.c_double, ct.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                      _vp, ct.c_size_t, _vp]


def ransac_process_candidates(Fs, x0, x1, singular_value_ratio_allowed=3e-2, required_percent_inliers=.9,
                              reprojection_error_allowed=.5, find_best_even_in_failure=True, want_mask=False,
                              workspace=None):
    """RANSAC candidate processing on device (reference src/RansacFitter.h:42-95 for a batch of
    candidates): Fs CUDA float64 [nF,3,3] (nF <= 16383), x0, x1 CUDA float64 [npt,3].  Returns a dict of
    CUDA tensors: success int32 [nF], inlier_count int32 [nF], best_camera int32 [nF], camera float64
    [nF,3,4], singular_value_ratio float64 [nF], essential float64 [nF,3,3], counts4 int32 [nF,4] and,
    with want_mask, inlier_mask uint8 [nF,npt].  Asynchronous on the current stream."""
    _need(Fs, torch.float64, "Fs")
    _need(x0, torch.float64, "x0")
    _need(x1, torch.float64, "x1")
    nF, npt = Fs.shape[0], x0.shape[0]
    assert Fs.shape[1:] == (3, 3) and x0.shape == x1.shape and x0.shape[1] == 3
    dev = x0.device
    out = {"success": torch.empty(nF, dtype=torch.int32, device=dev),
           "inlier_count": torch.empty(nF, dtype=torch.int32, device=dev),
           "best_camera": torch.empty(nF, dtype=torch.int32, device=dev),
           "camera": torch.empty((nF, 3, 4), dtype=torch.float64, device=dev),
           "singular_value_ratio": torch.empty(nF, dtype=torch.float64, device=dev),
           "essential": torch.empty((nF, 3, 3), dtype=torch.float64, device=dev),
           "counts4": torch.empty((nF, 4), dtype=torch.int32, device=dev)}
    mask = torch.empty((nF, npt), dtype=torch.uint8, device=dev) if want_mask else None
    with _on_device_of(Fs, x0, x1) as stream:
        ws = (workspace or _default_ws).get(clib.spv_ransac_workspace_bytes(nF, npt, 1 if want_mask else 0), dev)
        check(clib.spv_ransac_process_candidates_device(
            Fs.data_ptr(), nF, npt, x0.data_ptr(), x1.data_ptr(), float(singular_value_ratio_allowed),
            float(required_percent_inliers), float(reprojection_error_allowed), int(bool(find_best_even_in_failure)),
            out["success"].data_ptr(), out["inlier_count"].data_ptr(), out["best_camera"].data_ptr(),
            out["camera"].data_ptr(), out["singular_value_ratio"].data_ptr(), out["essential"].data_ptr(),
            out["counts4"].data_ptr(), mask.data_ptr() if want_mask else None, ws.data_ptr(), ws.numel(), stream))
    if want_mask:
        out["inlier_mask"] = mask
    return out


_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
clib.spv_ransac_fit_device.restype = ct.c_int
clib.spv_ransac_fit_device.argtypes = [_vp, _vp, ct.c_int, ct.c_double, ct.c_double, ct.c_int, ct.c_int, ct.c_double,
                                       ct.c_ulonglong, _vp, ct.POINTER(ct.c_int32), _f64p, _f64p,
                                       ct.POINTER(ct.c_double), _i32p, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32),
                                       ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), _vp]


def ransac_fit(x0, x1, required_percent_inliers=.9, reprojection_error_allowed=.5, maximum_tries=500,
               find_best_even_in_failure=True, singular_value_ratio_allowed=3e-2, seed=0, samples=None):
    """`mvg.ransac_fit` with the correspondences resident in HBM (reference RansacFitter::fit_essential,
    src/RansacFitter.h:152-272): x0, x1 CUDA float64 [npt,3] (e.g. the output of `match_coordinates` after
    calibration).  The small results come back as numpy: the `mvg.ransac_fit` dict.  Synchronises the
    current stream once per batch of tries."""
    _need(x0, torch.float64, "x0")
    _need(x1, torch.float64, "x1")
    assert x0.shape == x1.shape and x0.dim() == 2 and x0.shape[1] == 3
    npt = x0.shape[0]
    if npt < 10:
        raise ValueError('Supplied less than 10 point matches, unsupported.')
    if samples is not None:
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        if not (samples.ndim == 2 and samples.shape[1] == 7):
            raise TypeError('samples must be [ntries,7].')
        maximum_tries = samples.shape[0]
    ok, n, bt, br, ran = ct.c_int32(0), ct.c_int32(0), ct.c_int32(-1), ct.c_int32(-1), ct.c_int32(0)
    pct = ct.c_double(0.0)
    F, P, idx = np.zeros(9), np.zeros(12), np.zeros(npt, np.int32)
    with _on_device_of(x0, x1) as stream:
        check(clib.spv_ransac_fit_device(
            x0.data_ptr(), x1.data_ptr(), npt, float(required_percent_inliers), float(reprojection_error_allowed),
            int(maximum_tries), int(bool(find_best_even_in_failure)), float(singular_value_ratio_allowed), int(seed),
            samples.ctypes.data if samples is not None else None, ct.byref(ok), F, P, ct.byref(pct), idx, ct.byref(n),
            ct.byref(bt), ct.byref(br), ct.byref(ran), stream))
    found = bt.value >= 0
    return {'success': bool(ok.value), 'essential': F.reshape(3, 3) if found else None,
            'camera': P.reshape(3, 4) if found else None, 'inlier_percent': float(pct.value),
            'inlier_idx': idx[:n.value].copy(), 'best_try': bt.value, 'best_root': br.value, 'tries_run': ran.value}


clib.spv_normalize_workspace_bytes_rows.restype = ct.c_size_t
clib.spv_normalize_workspace_bytes_rows.argtypes = [ct.c_int, ct.c_int]
clib.spv_normalize_device.restype = ct.c_int
clib.spv_normalize_device.argtypes = [_vp, ct.c_int, ct.c_int, _vp, _vp, _vp, ct.c_size_t, _vp]


def normalize(x, want_float=True, want_ubyte=False, workspace=None):
    """`normalize_to_ubyte_and_multiple_16_dim` (reference spectavi/feature.py:384-407) on a CUDA float32
    [rows, dim] table, bit-identical to the numpy function: returns the float32 [rows, dim16] table
    and / or its `(out + 128).astype('uint8')` image (what `l1k2` takes).  Asynchronous."""
    _need(x, torch.float32, "x")
    rows, dim = x.shape
    dim16 = (dim + 15) // 16 * 16
    out = torch.empty((rows, dim16), dtype=torch.float32, device=x.device) if want_float else None
    u8 = torch.empty((rows, dim16), dtype=torch.uint8, device=x.device) if want_ubyte else None
    with _on_device_of(x) as stream:
        ws = (workspace or _default_ws).get(clib.spv_normalize_workspace_bytes_rows(rows, dim), x.device)
        check(clib.spv_normalize_device(x.data_ptr(), rows, dim, out.data_ptr() if want_float else None,
                                        u8.data_ptr() if want_ubyte else None, ws.data_ptr(), ws.numel(), stream))
    if want_float and want_ubyte:
        return out, u8
    return out if want_float else u8


clib.spv_seven_point_device.restype = ct.c_int
clib.spv_seven_point_device.argtypes = [_vp, _vp, ct.c_int, _vp, _vp, _vp, _vp]


def seven_point(x, xp, want_basis=False):
    """Seven-point algorithm for n 7-subsets resident in HBM (reference src/FundamentalMatrixFitter.h:108-246):
    x, xp CUDA float64 [n,7,2] euclidean.  Returns (nroot int32 [n], Fs float64 [n,3,3,3], NaN in the slots of
    missing roots) and, with want_basis, the null-space pair [n,2,3,3].  Asynchronous on the current stream."""
    _need(x, torch.float64, "x")
    _need(xp, torch.float64, "xp")
    assert x.dim() == 3 and x.shape[1:] == (7, 2) and xp.shape == x.shape
    n = x.shape[0]
    nroot = torch.empty(n, dtype=torch.int32, device=x.device)
    Fs = torch.empty((n, 3, 3, 3), dtype=torch.float64, device=x.device)
    basis = torch.empty((n, 2, 3, 3), dtype=torch.float64, device=x.device) if want_basis else None
    with _on_device_of(x, xp) as stream:
        check(clib.spv_seven_point_device(x.data_ptr(), xp.data_ptr(), n, Fs.data_ptr(), nroot.data_ptr(),
                                          basis.data_ptr() if want_basis else None, stream))
    return (nroot, Fs, basis) if want_basis else (nroot, Fs)
