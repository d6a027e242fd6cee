"""GPU parity: cascade-hash prefilter + L1 refine through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_golden_2k(golden):
    from spectavi_amd import feature
    g = golden("cascade_2kx2k_m8n2g2.npz")
    x, y = g["x"].astype(np.float32), g["y"].astype(np.float32)
    idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, g["dict"], g=int(g["g"]), return_ncand=True)
    assert idx.dtype == np.uint64 and dist.dtype == np.float32
    assert np.array_equal(dist, g["dist"])
    assert np.array_equal(idx, g["idx"])
    assert np.array_equal(ncand, g["ncand"])


@pytest.mark.parametrize("m_rows,n_rows,dim,m,n,g", [
    (2000, 1500, 128, 8, 2, 2), (5000, 3000, 128, 10, 2, 2), (3000, 1000, 128, 6, 3, 0),
    (1000, 777, 144, 8, 16, 5),   # the reference test's parameters (m=8, n=16, g=5), dim 144
    (4000, 500, 128, 3, 1, 1),    # huge buckets: overflows the LDS candidate list
    (3000, 600, 128, 25, 2, 3),   # m > bucket bits: full-code check path
    (50, 40, 16, 4, 2, 2), (1, 5, 32, 4, 2, 1), (0, 5, 32, 4, 2, 1), (900, 300, 256, 7, 2, 2),
    (700, 200, 384, 6, 2, 2), (600, 150, 512, 6, 3, 1), (500, 120, 1024, 5, 2, 2), (300, 90, 2048, 5, 2, 3),  # wide rows
])
def test_matches_oracle(oracle, m_rows, n_rows, dim, m, n, g):
    from spectavi_amd import feature
    rng = np.random.default_rng(m_rows + 3 * n_rows + dim + m)
    x = rng.integers(-128, 128, (m_rows, dim)).astype(np.float32)
    y = rng.integers(-128, 128, (n_rows, dim)).astype(np.float32)
    k = min(m_rows, n_rows) // 2
    if k:
        y[:k] = np.clip(x[rng.integers(0, m_rows, k)] + rng.integers(-2, 3, (k, dim)), -128, 127)
    d = rng.standard_normal((n, dim, m)).astype(np.float32)
    idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
    oidx, odist, oncand, onset = oracle.nn_cascading_hash(x, y, m, n, g, d)
    assert np.array_equal(ncand, oncand)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


@pytest.mark.parametrize("dim,m,n,g", [
    # n*m = 16 CT + left, left in 1..8: the left-over columns run on 4x4x1 MFMAs (project_mfma4_kernel<CT,NG>)
    (128, 2, 1, 1),    # CT 0, one group
    (64, 6, 1, 2),     # CT 0, two groups
    (128, 9, 2, 2),    # 18 = 16 + 2
    (256, 11, 2, 4),   # 22 = 16 + 6: two groups, g > 2 (the 16-entry selection network)
    (128, 17, 2, 2),   # 34 = 32 + 2: the default configuration of a 1M-row database
    (512, 13, 3, 2),   # 39 = 32 + 7, widest rows the kernel takes
    (32, 25, 2, 3),    # 50 = 48 + 2, m > bucket bits
    (128, 14, 4, 0),   # 56 = 48 + 8, no probing at all
    # the same column counts where that kernel does not apply: ragged chunk, rows wider than 512
    (144, 17, 2, 2), (544, 9, 2, 2),
])
def test_left_over_hyperplane_columns(oracle, dim, m, n, g):
    """Column counts that are not a whole number of 16-column MFMA tiles, on every (tiles, groups)
    instantiation of the 4x4x1 path and on its fall-backs: codes, masks and therefore candidates,
    indices and distances bit-identical to the oracle; non-integer inputs included (the uint8 image
    truncates, the projections do not)."""
    from spectavi_amd import feature
    rng = np.random.default_rng([dim, m, n, g])
    x = rng.integers(-128, 128, (3000, dim)).astype(np.float32)
    y = rng.integers(-128, 128, (1100, dim)).astype(np.float32)
    y[:500] = np.clip(x[rng.integers(0, 3000, 500)] + rng.integers(-2, 3, (500, dim)), -128, 127)
    x[100:200] += rng.uniform(-0.49, 0.49, (100, dim)).astype(np.float32)
    y[600:700] += rng.uniform(-0.49, 0.49, (100, dim)).astype(np.float32)
    d = rng.standard_normal((n, dim, m)).astype(np.float32)
    idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(x, y, m, n, g, d)
    assert np.array_equal(ncand, oncand)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


@pytest.mark.parametrize("m_rows,n_rows,dim,m,n,g,span", [
    (2, 1350, 128, 6, 3, 4, 20), (300, 900, 64, 8, 2, 5, 2), (1000, 500, 32, 11, 3, 3, 20), (64, 64, 16, 4, 1, 2, 2),
])
def test_tied_projection_magnitudes(oracle, m_rows, n_rows, dim, m, n, g, span):
    """Integer hyperplanes on small-integer data make |projection| ties and exact zeros common
    (plus all-zero rows, whose projections all tie at +0): the g probe bits must be the g
    smallest (|proj|, bit) PAIRS, lower bit first on equal magnitude, as the reference's heap of
    pairs keeps them (src/CascadingHashNn.h:153-159) -- found by tests/fuzz_gpu.py."""
    from spectavi_amd import feature
    rng = np.random.default_rng([m_rows, n_rows, dim, m])
    x = rng.integers(-span, span, (m_rows, dim)).astype(np.float32)
    y = rng.integers(-span, span, (n_rows, dim)).astype(np.float32)
    y[::17] = 0.0
    x[::5] = 0.0
    d = np.round(rng.standard_normal((n, dim, m))).astype(np.float32)
    idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(x, y, m, n, g, d)
    assert np.array_equal(ncand, oncand)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_reference_symbol_and_fallback(oracle):
    """nn_cascading_hash (NdArray path, library-drawn hyperplanes from a fixed seed) and the
    m<4 brute-force fallback of the front-end (reference spectavi/feature.py:364-371)."""
    import spectavi_amd
    from spectavi_amd import feature
    rng = np.random.default_rng(5)
    x = rng.integers(-128, 128, (3000, 128)).astype(np.float32)
    y = rng.integers(-128, 128, (1000, 128)).astype(np.float32)
    y[:500] = np.clip(x[rng.integers(0, 3000, 500)] + rng.integers(-2, 3, (500, 128)), -128, 127)
    spectavi_amd.set_hash_seed(42)
    try:
        idx, dist = feature.nn_cascading_hash(x, y)          # m auto = 8, n = 2, g = 2
        d = feature.generate_hash_dict(42, 128, 8, 2)
    finally:
        spectavi_amd.set_hash_seed(None)
    oidx, odist, _, _ = oracle.nn_cascading_hash(x, y, 8, 2, 2, d)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    xs, ys = x[:60], y[:50]                                   # m = 3 < 4 -> brute force, int32 dists
    idx, dist = feature.nn_cascading_hash(xs, ys)
    oidx, odist = oracle.nn_bruteforcel1k2((xs + 128).astype(np.uint8), (ys + 128).astype(np.uint8))
    assert dist.dtype == np.int32 and np.array_equal(idx, oidx) and np.array_equal(dist, odist)


def test_reference_test_recall_bound(golden):
    """reference test/test_feature.py:123-151 on the HIP path."""
    from oracle.oracle import numpy_l1_top2
    from spectavi_amd import feature
    g = golden("cascade_ref_test_inputs.npz")
    x = feature.normalize_to_ubyte_and_multiple_16_dim(g["x"])
    y = feature.normalize_to_ubyte_and_multiple_16_dim(g["y"])
    nni, nnd = feature.nn_cascading_hash(x, y, m=8, n=16, g=5)
    gt_nni, _ = numpy_l1_top2((x + 128).astype(np.uint8), (y + 128).astype(np.uint8))
    assert np.sum(gt_nni != nni) <= 2 * round(.4 * 200)


def test_randomized_parameters(oracle):
    """Seeded random (rows, dim, m, n, g) settings, bit-exact vs the oracle incl. candidate counts."""
    from spectavi_amd import feature
    rng = np.random.default_rng(424242)
    for _ in range(16):
        dim = 16 * int(rng.integers(1, 13))
        mr, nr = int(rng.integers(1, 3000)), int(rng.integers(1, 1200))
        m = int(rng.integers(2, 14))
        n = int(rng.integers(1, 5))
        g = int(rng.integers(0, min(m, 5) + 1))
        x = rng.integers(-128, 128, (mr, dim)).astype(np.float32)
        y = rng.integers(-128, 128, (nr, dim)).astype(np.float32)
        k = min(mr, nr) // 2
        if k:
            y[:k] = np.clip(x[rng.integers(0, mr, k)] + rng.integers(-2, 3, (k, dim)), -128, 127)
        d = rng.standard_normal((n, dim, m)).astype(np.float32)
        idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
        oidx, odist, oncand, _ = oracle.nn_cascading_hash(x, y, m, n, g, d)
        assert np.array_equal(ncand, oncand), (mr, nr, dim, m, n, g)
        assert np.array_equal(dist, odist) and np.array_equal(idx, oidx), (mr, nr, dim, m, n, g)


def test_full_size_properties_1m(oracle):
    """BASELINE config 3 (1M x 1M, m=17 n=2 g=2) resident in HBM: planted noisy copies are
    recovered, reported distances are true L1 distances, and a 512-query subsample agrees
    with the oracle (which hashes the full 1M-row database on the CPU) bit for bit."""
    import torch
    from spectavi_amd import device, feature
    rows = 1_000_000
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.randint(0, 256, (rows, 128), dtype=torch.uint8, device=dev, generator=g).float() - 128
    perm = torch.randperm(rows, device=dev, generator=g)
    y = torch.clamp(x[perm] + torch.randint(-3, 4, (rows, 128), device=dev, generator=g).float(), -128, 127)
    m = feature.auto_hash_bit_rate(rows, rows)
    assert m == 17
    d_host = feature.generate_hash_dict(0x5eed, 128, m, 2)
    idx, dist, ncand = device.cascade(x, y, torch.from_numpy(d_host).to(dev), g=2, want_ncand=True)
    torch.cuda.synchronize()
    assert float((idx[:, 0] == perm).float().mean()) > 0.999
    assert bool((dist[:, 0] <= dist[:, 1]).all())
    sel = torch.arange(0, rows, 997, device=dev)
    ok = idx[sel, 1] >= 0
    for c in range(2):
        rows_c = idx[sel, c].clamp(min=0)
        dtrue = ((x[rows_c] + 128).to(torch.int32) - (y[sel] + 128).to(torch.int32)).abs().sum(1).float()
        good = ok if c == 1 else torch.ones_like(ok)
        assert bool((dtrue[good] == dist[sel, c][good]).all())
    sub = np.arange(0, rows, rows // 512)[:512]
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(x.cpu().numpy(), y[sub].cpu().numpy(), m, 2, 2, d_host)
    assert np.array_equal(idx[sub].cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist[sub].cpu().numpy(), odist)
    assert np.array_equal(ncand[sub].cpu().numpy(), oncand)


@pytest.mark.parametrize("m_rows,n_rows,dim,m,n,g", [
    (5000, 3001, 128, 10, 2, 2),    # the default form (two tables, 8 probes), ragged last query block
    (3000, 1000, 128, 6, 3, 0),     # no probing: one bucket per table, three passes
    (4000, 517, 128, 3, 1, 1),      # one table, huge buckets: several list windows per pass
    (2000, 33, 144, 8, 4, 3),       # 4 tables x 8 probes, dim 144 (two chunks per lane), fewer blocks than XCDs
    (6000, 2500, 256, 9, 2, 4),     # 16 probes per table: two rounds of 8 inside a pass
    (1000, 777, 144, 8, 16, 5),     # the reference test's parameters through 16 passes (VALU projection + query_rank_kernel)
    (1, 300, 32, 4, 2, 1), (50, 1, 16, 4, 2, 2),
])
def test_sorted_probe_passes_match_oracle(oracle, monkeypatch, m_rows, n_rows, dim, m, n, g):
    """The probe walks the queries table by table in the order of that table's sign code (large
    inputs by default; SPECTAVI_CASCADE_SORT=1 forces it at any size, =0 forces the one-pass form).
    Whatever the order and the number of passes, indices, distances and candidate counts are the
    oracle's bits -- planted neighbours reached through several tables (the keys carried from pass to
    pass, the skip of rows that already hold a place) included.  Reference:
    src/CascadingHashNn.h:208-245."""
    from spectavi_amd import feature
    rng = np.random.default_rng([m_rows, n_rows, dim, m, n, g])
    x = rng.integers(-128, 128, (m_rows, dim)).astype(np.float32)
    y = rng.integers(-128, 128, (n_rows, dim)).astype(np.float32)
    k = min(m_rows, n_rows) // 2
    if k:
        src = rng.integers(0, m_rows, k)
        y[:k] = np.clip(x[src] + rng.integers(-1, 2, (k, dim)), -128, 127)
        y[: k // 4] = x[src[: k // 4]]               # exact copies: distance 0 through every table
    d = rng.standard_normal((n, dim, m)).astype(np.float32)
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(x, y, m, n, g, d)
    for mode in ("1", "0"):
        monkeypatch.setenv("SPECTAVI_CASCADE_SORT", mode)
        for ru8 in (None, "0"):   # eight rows per round with a candidate per lane (default) / four, every lane reducing all
            if ru8 is None:
                monkeypatch.delenv("SPECTAVI_CASCADE_RU8", raising=False)
            else:
                monkeypatch.setenv("SPECTAVI_CASCADE_RU8", ru8)
            idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
            assert np.array_equal(ncand, oncand), (mode, ru8)
            assert np.array_equal(dist, odist), (mode, ru8)
            assert np.array_equal(idx, oidx), (mode, ru8)
