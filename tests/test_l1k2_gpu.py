"""GPU parity: the HIP L1 2-NN path, called through the C-ABI, against the CPU oracle,
the golden fixtures and size-independent properties at BASELINE sizes."""
import numpy as np
import pytest

from tests.conftest import uniform_u8

pytestmark = pytest.mark.gpu

U64MAX = np.iinfo(np.uint64).max
I32MAX = np.iinfo(np.int32).max


def _raw(x, y):
    """spv_nn_bruteforcel1k2: host pointers, caller-allocated outputs."""
    import ctypes as ct
    from spectavi_amd._lib import clib, check
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y)
    idx = np.empty((y.shape[0], 2), np.uint64)
    dist = np.empty((y.shape[0], 2), np.int32)
    clib.spv_nn_bruteforcel1k2.restype = ct.c_int
    clib.spv_nn_bruteforcel1k2.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int,
                                           ct.c_void_p, ct.c_void_p]
    check(clib.spv_nn_bruteforcel1k2(x.ctypes.data, y.ctypes.data, x.shape[0], y.shape[0], y.shape[1],
                                     idx.ctypes.data, dist.ctypes.data))
    return idx, dist


@pytest.mark.parametrize("name", ["l1k2_200x144.npz", "l1k2_1kx1k_128.npz", "l1k2_ties_300x500_64.npz",
                                  "l1k2_dups_257x5_128.npz", "l1k2_m0.npz", "l1k2_m1.npz", "l1k2_m2.npz"])
def test_golden_through_reference_symbol(golden, name):
    """nn_bruteforcel1k2 (NdArray out-params), the symbol the reference front-end binds."""
    from spectavi_amd import feature
    g = golden(name)
    idx, dist = feature.nn_bruteforcel1k2(g["x"], g["y"])
    assert idx.dtype == np.uint64 and dist.dtype == np.int32 and idx.shape == (g["y"].shape[0], 2)
    assert np.array_equal(dist, g["dist"])  # what the reference's own test checks
    assert np.array_equal(idx, g["idx"])    # bit-exact indices, lexicographic ties


@pytest.mark.parametrize("m,n,dim,hi", [
    (1000, 1000, 128, 256),   # BASELINE config 1
    (200, 200, 144, 256),     # the reference's test shape (dim 144)
    (777, 333, 128, 256),     # ragged vs tile/block sizes
    (65, 1, 128, 256), (63, 257, 128, 3), (1, 1, 16, 256), (2, 3, 32, 256),
    (130, 70, 64, 2), (500, 300, 48, 256), (300, 100, 80, 256), (300, 100, 160, 256),
    (300, 100, 192, 256), (300, 100, 208, 256), (300, 100, 256, 256),
    (3000, 700, 96, 256), (2500, 900, 112, 256), (1200, 1300, 224, 256), (5000, 520, 160, 3), (700, 4100, 80, 256),   # every instantiated width
    (300, 100, 272, 256), (200, 130, 512, 256), (150, 70, 2048, 256),   # wide-row kernel: whole chunks and a ragged end
    (4100, 700, 400, 256), (900, 333, 1040, 7), (513, 257, 2032, 256), (77, 600, 288, 256),
    (70000, 300, 128, 256),   # more than one 65536-row slice limit worth of rows
    (4096, 5000, 128, 2),     # tie-heavy at a size with several slices
])
def test_matches_oracle(oracle, m, n, dim, hi):
    rng = np.random.default_rng(m * 131 + n * 7 + dim)
    x = rng.integers(0, hi, (m, dim), dtype=np.uint8)
    y = rng.integers(0, hi, (n, dim), dtype=np.uint8)
    idx, dist = _raw(x, y)
    oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_max_distance_and_extremes(oracle):
    x = np.zeros((300, 256), np.uint8)
    y = np.full((5, 256), 255, np.uint8)  # 256 * 255 = 65280: largest key that must fit 16 bits
    x[7, :3] = 1
    idx, dist = _raw(x, y)
    oidx, odist = oracle.nn_bruteforcel1k2(x, y)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist) and dist[0, 1] == 65280


def test_empty_and_sentinels():
    y = uniform_u8(1, 9, 32)
    idx, dist = _raw(np.zeros((0, 32), np.uint8), y)
    assert np.all(idx == U64MAX) and np.all(dist == I32MAX)
    idx, dist = _raw(uniform_u8(2, 1, 32), y)
    assert np.all(idx[:, 0] == 0) and np.all(idx[:, 1] == U64MAX) and np.all(dist[:, 1] == I32MAX)
    idx, dist = _raw(uniform_u8(2, 5, 32), np.zeros((0, 32), np.uint8))
    assert idx.shape == (0, 2)


def test_errors_are_reported_not_thrown():
    from spectavi_amd._lib import SpectaviError
    with pytest.raises(SpectaviError):
        _raw(np.zeros((4, 24), np.uint8), np.zeros((4, 24), np.uint8))
    with pytest.raises(SpectaviError):
        _raw(np.zeros((4, 2064), np.uint8), np.zeros((4, 2064), np.uint8))


def test_device_path_and_variants(oracle, monkeypatch):
    """spv_l1k2_device on resident tensors; every queries-per-lane variant gives the same bits."""
    import torch
    from spectavi_amd import device
    x = uniform_u8(5, 3000, 128)
    y = uniform_u8(6, 2500, 128)
    oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    idx, dist = device.l1k2(xd, yd)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist.cpu().numpy(), odist)


def test_full_size_properties_256k(oracle):
    """BASELINE config 2 (256k x 256k, D=128) through size-independent properties:
    planted exact copies are found at distance 0, results are sorted, and a query
    subsample agrees with the oracle bit for bit."""
    import torch
    from spectavi_amd import device
    n = 262144
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    planted = torch.arange(0, n, 1021, device="cuda")
    src = (planted * 7919) % n
    y[planted] = x[src]
    idx, dist = device.l1k2(x, y)
    torch.cuda.synchronize()
    assert bool((dist[:, 0] <= dist[:, 1]).all())
    assert bool((dist[planted, 0] == 0).all()) and bool((idx[planted, 0] == src).all())
    assert bool((idx >= 0).all()) and bool((idx < n).all()) and bool((idx[:, 0] != idx[:, 1]).all())
    # the distance reported for (query, idx) is the true L1 distance of that pair
    sel = torch.arange(0, n, 257, device="cuda")
    for c in range(2):
        d = (x[idx[sel, c]].to(torch.int32) - y[sel].to(torch.int32)).abs().sum(1)
        assert bool((d == dist[sel, c]).all())
    # oracle on a 512-query subsample against the full database
    sub = np.arange(0, n, 512)
    oidx, odist = oracle.nn_bruteforcel1k2(x.cpu().numpy(), y[sub].cpu().numpy(), nthreads=oracle.max_threads())
    assert np.array_equal(idx[sub].cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist[sub].cpu().numpy(), odist)


def test_randomized_shapes(oracle):
    """40 seeded random shapes / widths / value ranges, host-pointer path, bit-exact vs the oracle."""
    rng = np.random.default_rng(20261004)
    for _ in range(40):
        dim = 16 * int(rng.integers(1, 21))
        m = int(rng.integers(0, 3000))
        n = int(rng.integers(1, 1500))
        hi = int(rng.choice([2, 4, 256]))
        x = rng.integers(0, hi, (m, dim), dtype=np.uint8)
        y = rng.integers(0, hi, (n, dim), dtype=np.uint8)
        idx, dist = _raw(x, y)
        oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
        assert np.array_equal(dist, odist), (m, n, dim, hi)
        assert np.array_equal(idx, oidx), (m, n, dim, hi)


def test_repeated_calls_reuse_cached_buffers(oracle):
    """The host path caches device buffers between calls; results must not depend on it."""
    import ctypes as ct
    from spectavi_amd._lib import clib
    rng = np.random.default_rng(5)
    for k in range(6):
        m, n = int(rng.integers(10, 4000)), int(rng.integers(10, 4000))
        x = rng.integers(0, 256, (m, 128), dtype=np.uint8)
        y = rng.integers(0, 256, (n, 128), dtype=np.uint8)
        idx, dist = _raw(x, y)
        oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        if k == 3:
            clib.spv_release_cached_memory.restype = None
            clib.spv_release_cached_memory()


def test_workspace_too_small_is_an_error():
    import torch
    import ctypes as ct
    from spectavi_amd._lib import clib, SPV_ERR_INVALID
    from spectavi_amd import device  # noqa: F401  (declares argtypes)
    x = torch.zeros((1000, 128), dtype=torch.uint8, device="cuda")
    idx = torch.empty((1000, 2), dtype=torch.int64, device="cuda")
    dist = torch.empty((1000, 2), dtype=torch.int32, device="cuda")
    ws = torch.empty(256, dtype=torch.uint8, device="cuda")
    st = clib.spv_l1k2_device(x.data_ptr(), x.data_ptr(), 1000, 1000, 128, idx.data_ptr(), dist.data_ptr(),
                              ws.data_ptr(), 16, None)
    assert st == SPV_ERR_INVALID and b"workspace" in clib.spv_last_error()


def test_extreme_aspect_ratios(oracle):
    """One query against millions of rows (2048 slices, 64-lane merge) and a million queries
    against a handful of rows."""
    import torch
    from spectavi_amd import device
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randint(0, 256, (3_000_001, 128), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.randint(0, 256, (3, 128), dtype=torch.uint8, device="cuda", generator=g)
    x[2_999_999] = y[1]            # exact match in the very last slice
    x[5] = y[1]                    # and an earlier duplicate: lower index must win
    idx, dist = device.l1k2(x, y)
    torch.cuda.synchronize()
    oidx, odist = oracle.nn_bruteforcel1k2(x.cpu().numpy(), y.cpu().numpy(), nthreads=oracle.max_threads())
    assert np.array_equal(idx.cpu().numpy().view(np.uint64), oidx) and np.array_equal(dist.cpu().numpy(), odist)
    assert idx[1].tolist() == [5, 2_999_999] and dist[1].tolist() == [0, 0]
    xs = torch.randint(0, 256, (3, 128), dtype=torch.uint8, device="cuda", generator=g)
    ys = torch.randint(0, 256, (1_000_003, 128), dtype=torch.uint8, device="cuda", generator=g)
    idx, dist = device.l1k2(xs, ys)
    torch.cuda.synchronize()
    sub = np.arange(0, 1_000_003, 977)
    oidx, odist = oracle.nn_bruteforcel1k2(xs.cpu().numpy(), ys[sub].cpu().numpy(), nthreads=8)
    assert np.array_equal(idx[sub].cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist[sub].cpu().numpy(), odist)
    assert bool((idx >= 0).all()) and bool((idx < 3).all())


def test_config5_full_shard(oracle):
    """BASELINE configs[4] shards 4M queries over 8 GPUs against a replicated 4M-row database:
    ONE GPU's full share of it, 4,000,000 database rows x 500,000 query rows (2e12 pairs, ~1.8 s),
    through size-independent properties -- planted exact copies spread over all 64 database
    slices come back at distance 0 with the right index, the lower index wins between two exact
    copies, every row is sorted, in range and distinct, the reported distance of a strided
    sample of (query, idx) pairs is the true L1 distance -- plus a 1024-query subsample of the
    shard compared bit for bit with the oracle against the full database."""
    import torch
    from spectavi_amd import device
    m, n = 4_000_000, 500_000
    g = torch.Generator(device="cuda").manual_seed(2026)
    x = torch.randint(0, 256, (m, 128), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    planted = torch.arange(0, 4096, device="cuda") * 122 + 7          # query rows
    src = (torch.arange(0, 4096, device="cuda") * 976_553 + 11) % m    # database rows, all slices
    y[planted] = x[src]
    x[3_999_999] = y[499_999]   # two exact copies of the last query: the last database row ...
    x[17] = y[499_999]          # ... and an early one; the lower index must come first
    idx, dist = device.l1k2(x, y)
    torch.cuda.synchronize()
    assert bool((dist[planted, 0] == 0).all()) and bool((idx[planted, 0] == src).all())
    assert idx[499_999].tolist() == [17, 3_999_999] and dist[499_999].tolist() == [0, 0]
    assert bool((dist[:, 0] <= dist[:, 1]).all())
    assert bool((idx >= 0).all()) and bool((idx < m).all()) and bool((idx[:, 0] != idx[:, 1]).all())
    sel = torch.arange(0, n, 97, device="cuda")
    for c in range(2):
        d = (x[idx[sel, c]].to(torch.int32) - y[sel].to(torch.int32)).abs().sum(1)
        assert bool((d == dist[sel, c]).all())
    sub = np.concatenate([np.arange(0, n, 489), [499_999]])
    oidx, odist = oracle.nn_bruteforcel1k2(x.cpu().numpy(), y[sub].cpu().numpy(), nthreads=oracle.max_threads())
    assert np.array_equal(idx[sub].cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist[sub].cpu().numpy(), odist)


def test_north_star_shape_1m_x_1m(oracle):
    """north_star's target shape, the one bench.py's headline line is quoted on: 1,000,000 database
    rows x 1,000,000 query rows, D = 128 (1e12 pairs, ~0.9 s), with the reference's test distribution
    (uniform uint8, test/test_feature.py:112-115).  Size-independent properties over all rows --
    planted exact copies in every database slice found at distance 0, the lower index first between
    two exact copies, rows sorted / in range / distinct, reported distances equal to the true L1
    distance on a strided sample (the reference's own test property, test/test_feature.py:116-121,
    here per returned index) -- plus a 1024-query subsample compared bit for bit with the oracle
    against the full database."""
    import torch
    from spectavi_amd import device
    m = n = 1_000_000
    g = torch.Generator(device="cuda").manual_seed(0xdeadbeef)
    x = torch.randint(0, 256, (m, 128), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    planted = torch.arange(0, 4096, device="cuda") * 244 + 3           # query rows
    src = (torch.arange(0, 4096, device="cuda") * 244_141 + 5) % m      # database rows, all 16 slices
    y[planted] = x[src]
    x[999_999] = y[999_999]     # two exact copies of the last query: the last database row ...
    x[41] = y[999_999]          # ... and an early one; the lower index must come first
    idx, dist = device.l1k2(x, y)
    torch.cuda.synchronize()
    assert bool((dist[planted, 0] == 0).all()) and bool((idx[planted, 0] == src).all())
    assert idx[999_999].tolist() == [41, 999_999] and dist[999_999].tolist() == [0, 0]
    assert bool((dist[:, 0] <= dist[:, 1]).all())
    assert bool((idx >= 0).all()) and bool((idx < m).all()) and bool((idx[:, 0] != idx[:, 1]).all())
    sel = torch.arange(0, n, 61, device="cuda")
    for c in range(2):
        d = (x[idx[sel, c]].to(torch.int32) - y[sel].to(torch.int32)).abs().sum(1)
        assert bool((d == dist[sel, c]).all())
    sub = np.concatenate([np.arange(0, n, 978), [999_999]])
    assert len(sub) >= 1024
    oidx, odist = oracle.nn_bruteforcel1k2(x.cpu().numpy(), y[sub].cpu().numpy(), nthreads=oracle.max_threads())
    assert np.array_equal(idx[sub].cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist[sub].cpu().numpy(), odist)


def test_large_host_output_uses_the_pinned_pipeline(oracle):
    """1.2M queries: the 19 MB index array comes back through the pinned, threaded download path
    (results of 16 MB and more); a strided sample against the oracle, every row against the
    device-resident path."""
    import torch
    from spectavi_amd import device
    rng = np.random.default_rng(99)
    x = rng.integers(0, 256, (513, 128), dtype=np.uint8)
    y = rng.integers(0, 256, (1_200_003, 128), dtype=np.uint8)
    idx, dist = _raw(x, y)
    didx, ddist = device.l1k2(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    assert np.array_equal(idx, didx.cpu().numpy().view(np.uint64)) and np.array_equal(dist, ddist.cpu().numpy())
    sub = np.arange(0, y.shape[0], 1013)
    oidx, odist = oracle.nn_bruteforcel1k2(x, y[sub], nthreads=8)
    assert np.array_equal(idx[sub], oidx) and np.array_equal(dist[sub], odist)
