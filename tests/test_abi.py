"""CPU-only: the C-ABI library loads, exports every symbol include/*.h declares, and the
front-end's argument validation mirrors the reference's.  No compute calls (no GPU here)."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for hdr in ("spectavi_amd.h", "NdArray.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?(?:void|int|size_t|char|long long)\s*\*?\s*(\w+)\s*\(", text, flags=re.M):
            names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from spectavi_amd._lib import clib
    names = declared_symbols()
    assert {"nn_bruteforcel1k2", "nn_cascading_hash", "dlt_triangulate", "dlt_reprojection_error",
            "spv_l1k2_device", "spv_cascade_device", "spv_dlt_triangulate_device",
            "ndarray_set_size", "ndarray_alloc"} <= set(names)
    for n in names:
        assert hasattr(clib, n), "libspectavi.so does not export %s" % n


def test_ndarray_struct_matches_header():
    from spectavi_amd._lib import clib
    from spectavi_amd.ndarray import NdArray
    assert ct.sizeof(NdArray) == 8 + 4 * 8 + 4 + 4
    a = NdArray(dtype='int32')
    clib.ndarray_set_size.argtypes = [ct.POINTER(NdArray), ct.c_size_t, ct.c_size_t]
    clib.ndarray_alloc.argtypes = [ct.POINTER(NdArray)]
    clib.ndarray_alloc.restype = ct.c_int
    clib.ndarray_set_size(ct.byref(a), 3, 2)
    assert clib.ndarray_alloc(ct.byref(a)) == 0
    ct.memmove(a.m_data, (ct.c_int32 * 6)(1, 2, 3, 4, 5, 6), 24)
    arr = a.asarray()
    assert arr.dtype == np.int32 and arr.shape == (3, 2) and arr.tolist() == [[1, 2], [3, 4], [5, 6]]


def test_status_api_without_gpu():
    from spectavi_amd import _lib
    assert _lib.clib.spv_version().startswith(b"spectavi_amd")
    assert _lib.device_count() >= 0
    # invalid arguments are rejected before any device work
    from spectavi_amd import feature
    st = feature._spv_generate_hash_dict(1, 0, 4, 2, np.zeros(1, np.float32))
    assert st == _lib.SPV_ERR_INVALID
    with pytest.raises(_lib.SpectaviError):
        _lib.check(st)


def test_hash_dict_is_mt19937_normal_stream():
    from spectavi_amd import feature
    d = feature.generate_hash_dict(123, 16, 5, 3)
    assert d.shape == (3, 16, 5) and d.dtype == np.float32
    d2 = feature.generate_hash_dict(123, 16, 5, 3)
    assert np.array_equal(d, d2)
    assert not np.array_equal(d, feature.generate_hash_dict(124, 16, 5, 3))
    big = feature.generate_hash_dict(7, 128, 17, 2)
    assert abs(float(big.mean())) < 0.05 and abs(float(big.std()) - 1) < 0.05


def test_frontend_validation_matches_reference():
    from spectavi_amd import feature, mvg
    with pytest.raises(AssertionError):  # reference feature.py:297-299
        feature.nn_bruteforcel1k2(np.zeros((4, 16), np.uint8), np.zeros((4, 32), np.uint8))
    with pytest.raises(ValueError):      # reference throws runtime_error (BruteForceNnL1K2.h:77-81)
        feature.nn_bruteforcel1k2(np.zeros((4, 24), np.uint8), np.zeros((4, 24), np.uint8))
    P = np.zeros((3, 4))
    with pytest.raises(TypeError):       # reference mvg.py:283-294
        mvg.dlt_triangulate(np.zeros((4, 4)), P, np.zeros((2, 3)), np.zeros((2, 3)))
    with pytest.raises(TypeError):
        mvg.dlt_triangulate(P, P, np.zeros((2, 3)), np.zeros((3, 3)))
    with pytest.raises(TypeError):
        mvg.dlt_triangulate(P, P, np.zeros((2, 2)), np.zeros((2, 2)))
    assert feature.auto_hash_bit_rate(1_000_000, 1_000_000) == 17  # reference feature.py:366-367
    assert feature.auto_hash_bit_rate(200, 200) == 5 and feature.auto_hash_bit_rate(90, 90) == 3


def test_normalize_properties():
    """reference spectavi/feature.py:384-407: integer-valued, in [-128,127], zero column mean
    before scaling, width padded to a multiple of 16."""
    from spectavi_amd import feature
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 132)).astype(np.float32) * rng.uniform(0.1, 10, (1, 132))
    out = feature.normalize_to_ubyte_and_multiple_16_dim(x)
    assert out.dtype == np.float32 and out.shape == (50, 144)
    assert np.all(out == np.round(out)) and out.min() >= -128 and out.max() <= 127
    assert np.all(out[:, 132:] == 0)
    assert np.all(np.abs(out[:, :132]).max(0) >= 127)
    assert np.all(np.abs(out[:, :132].mean(0)) < 1.0)


def test_no_gpu_is_a_loud_error():
    """Without a device the product path must fail, never fall back to a CPU computation."""
    from spectavi_amd import _lib, feature
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.zeros((4, 16), np.uint8)
    with pytest.raises(_lib.SpectaviError):
        feature.nn_bruteforcel1k2(x, x)


def test_frontend_signatures_match_reference_source():
    """Where the reference tree is mounted (the build container; not the GPU box), check that the
    hot-path front-end functions keep the reference's parameter names and defaults and that the
    ctypes argtypes lists have the same length and scalar/pointer pattern.  Source is parsed as
    text (ast), nothing is imported or executed from the reference."""
    import ast
    import inspect
    ref_dir = "/root/reference/spectavi"
    if not os.path.isdir(ref_dir):
        pytest.skip("reference tree not present")
    from spectavi_amd import feature, mvg

    def ref_functions(path):
        tree = ast.parse(open(path).read())
        out = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                names = [a.arg for a in node.args.args]
                defaults = [ast.literal_eval(d) for d in node.args.defaults]
                out[node.name] = (names, defaults)
        return tree, out

    def ref_argtypes_len(tree, var):
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute):
                t = node.targets[0]
                if t.attr == "argtypes" and isinstance(t.value, ast.Name) and t.value.id == var:
                    return len(node.value.elts)
        return None

    ftree, ffun = ref_functions(os.path.join(ref_dir, "feature.py"))
    mtree, mfun = ref_functions(os.path.join(ref_dir, "mvg.py"))
    for mod, funs, names in ((feature, ffun, ["nn_bruteforcel1k2", "nn_cascading_hash",
                                              "normalize_to_ubyte_and_multiple_16_dim"]),
                             (mvg, mfun, ["dlt_triangulate", "dlt_reprojection_error", "hnormalize",
                                          "ransac_fitter", "seven_point_algorithm"])):
        for name in names:
            sig = inspect.signature(getattr(mod, name))
            ours = list(sig.parameters)
            our_defaults = [p.default for p in sig.parameters.values() if p.default is not inspect._empty]
            assert ours == funs[name][0], name
            assert our_defaults == funs[name][1], name
    assert len(feature._nn_bruteforcel1k2.argtypes) == ref_argtypes_len(ftree, "_nn_bruteforcel1k2") == 8
    assert len(feature._nn_cascading_hash.argtypes) == ref_argtypes_len(ftree, "_nn_cascading_hash") == 11
    assert len(mvg._dlt_triangulate.argtypes) == ref_argtypes_len(mtree, "_dlt_triangulate") == 6
    assert len(mvg._dlt_reprojection_error.argtypes) == ref_argtypes_len(mtree, "_dlt_reprojection_error") == 6
    assert len(mvg._ransac_fitter.argtypes) == ref_argtypes_len(mtree, "_ransac_fitter") == 14
    assert len(mvg._seven_point_algorithm.argtypes) == ref_argtypes_len(mtree, "_seven_point_algorithm") == 4


def test_only_the_declared_api_is_exported():
    """libspectavi.so is built -fvisibility=hidden: its dynamic symbol table holds the declared
    extern "C" API and nothing else of its own (no C++ internals another library could interpose)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "spectavi_amd", "libspectavi.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert exported == set(declared_symbols()), exported ^ set(declared_symbols())


def test_record_pack_and_widen_arithmetic():
    """The 16-byte (idx0, idx1, d0, d1) record that the RCCL gather moves (spectavi_amd/csrc/records.h,
    the same inline functions the device kernels call): (size_t)-1 travels as -1 and comes back as
    (size_t)-1, indices up to 2^31-1 and both distance types survive bit for bit, ragged shards map
    back to their rows.  Host functions only; no GPU involved."""
    from spectavi_amd._lib import clib, SPV_ERR_INVALID
    u64p, i32p = ct.POINTER(ct.c_uint64), ct.POINTER(ct.c_int32)
    clib.spv_records_pack.restype = ct.c_int
    clib.spv_records_pack.argtypes = [u64p, ct.c_void_p, ct.c_longlong, i32p]
    clib.spv_records_unpack.restype = ct.c_int
    clib.spv_records_unpack.argtypes = [i32p, ct.c_longlong, ct.c_int, ct.c_longlong, u64p, ct.c_void_p]
    rng = np.random.default_rng(9)
    none = np.iinfo(np.uint64).max
    for total, G in ((1, 1), (7, 1), (7, 3), (8, 8), (1001, 3), (1000, 8), (5, 4), (64, 7)):
        idx = rng.integers(0, 2**31, (total, 2)).astype(np.uint64)
        idx[0, 0] = 2**31 - 1
        idx[rng.random(total) < 0.2, 1] = none               # one neighbour only
        idx[rng.random(total) < 0.1] = none                  # none at all
        for dist in (rng.integers(0, 2**31, (total, 2)).astype(np.int32),
                     rng.standard_normal((total, 2)).astype(np.float32)):
            dist[idx == none] = 2147483647 if dist.dtype == np.int32 else 2147483648.0
            # what each rank would pack from its contiguous shard, padded to the largest shard
            base, extra = divmod(total, G)
            max_cnt = base + (1 if extra else 0)
            rec = np.full((G, max_cnt, 4), 0x5A5A5A5A, np.int32)
            lo = 0
            for r in range(G):
                cnt = base + (1 if r < extra else 0)
                part_i = np.ascontiguousarray(idx[lo:lo + cnt])
                part_d = np.ascontiguousarray(dist[lo:lo + cnt])
                out = np.empty((cnt, 4), np.int32)
                assert clib.spv_records_pack(part_i.ctypes.data_as(u64p), part_d.ctypes.data, cnt,
                                             out.ctypes.data_as(i32p)) == 0
                assert np.array_equal(out[:, :2] == -1, part_i == none)
                assert np.array_equal(out[:, 2:].view(dist.dtype), part_d)
                rec[r, :cnt] = out
                lo += cnt
            gi, gd = np.empty((total, 2), np.uint64), np.empty((total, 2), dist.dtype)
            assert clib.spv_records_unpack(rec.ctypes.data_as(i32p), total, G, max_cnt, gi.ctypes.data_as(u64p),
                                           gd.ctypes.data) == 0
            assert np.array_equal(gi, idx) and np.array_equal(gd.view(np.uint32), dist.view(np.uint32))
    # python statement of the same record (sharded.py) agrees
    import torch
    from spectavi_amd.sharded import pack_records, unpack_records
    idx = np.array([[5, none], [none, none], [2**31 - 1, 0]], np.uint64)
    dist = np.array([[7, 2147483647], [2147483647, 2147483647], [0, 65280]], np.int32)
    rec = np.empty((3, 4), np.int32)
    clib.spv_records_pack(idx.ctypes.data_as(u64p), dist.ctypes.data, 3, rec.ctypes.data_as(i32p))
    trec = pack_records(torch.from_numpy(idx.view(np.int64)), torch.from_numpy(dist))
    assert np.array_equal(trec.numpy(), rec)
    ti, td = unpack_records(trec)
    assert np.array_equal(ti.numpy().view(np.uint64), idx) and np.array_equal(td.numpy(), dist)
    # a max_cnt smaller than the largest shard is refused
    assert clib.spv_records_unpack(rec.ctypes.data_as(i32p), 3, 2, 1, idx.ctypes.data_as(u64p),
                                   dist.ctypes.data) == SPV_ERR_INVALID


def test_gather_mode_api_without_gpu():
    from spectavi_amd._lib import clib, SPV_ERR_INVALID
    clib.spv_set_gather_mode.restype = ct.c_int
    clib.spv_set_gather_mode.argtypes = [ct.c_int]
    assert clib.spv_set_gather_mode(7) == SPV_ERR_INVALID
    for mode in (1, 0, -1):
        assert clib.spv_set_gather_mode(mode) == 0


def test_ransac_sample_is_the_reference_draw():
    """floyd_sample (reference src/RansacFitter.h:120-132): seven distinct rows per try, values in
    [1, npt-1] (the reference's range starts at 1), reproducible per seed.  Host-only logic: runs without a GPU."""
    from spectavi_amd import mvg
    for npt in (10, 11, 50, 100000):
        s = mvg.ransac_sample(5, npt, 400)
        assert s.shape == (400, 7) and s.min() >= 1 and s.max() <= npt - 1
        assert all(len(set(r)) == 7 for r in s.tolist())
        assert np.array_equal(s, mvg.ransac_sample(5, npt, 400))
        assert not np.array_equal(s, mvg.ransac_sample(6, npt, 400))
    with pytest.raises(Exception):
        mvg.ransac_sample(1, 9, 1)


def test_ransac_fit_argument_checks_without_gpu():
    """Argument checks of the RANSAC entry points happen before any device is touched."""
    from spectavi_amd import mvg
    from spectavi_amd._lib import SPV_ERR_INVALID
    x = np.zeros((12, 3))
    ok, n = ct.c_int32(0), ct.c_int32(0)
    pct = ct.c_double(0)
    F, P, idx = np.zeros(9), np.zeros(12), np.zeros(12, np.int32)

    def call(npt, tries):
        return mvg._spv_ransac_fit(x, x, npt, .9, .5, tries, 1, 3e-2, 1, ct.byref(ok), F, P, ct.byref(pct), idx, ct.byref(n),
                                   None, None, None)
    assert call(9, 10) == SPV_ERR_INVALID        # fewer than 10 correspondences (the reference's constructor throws)
    assert call(12, -1) == SPV_ERR_INVALID
    assert call(12, 800000000) == SPV_ERR_INVALID
