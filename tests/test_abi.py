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
        for m in re.finditer(r"^\s*(?:const\s+)?(?:void|int|size_t|char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M):
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
                             (mvg, mfun, ["dlt_triangulate", "dlt_reprojection_error", "hnormalize"])):
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
