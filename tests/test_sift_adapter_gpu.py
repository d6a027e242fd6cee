"""GPU: SIFT table adapter + the whole device-side match pipeline on REAL descriptors.
Input fixture: the reference's own golden SIFT table (data/sift-test/sur-ogre.sift, 1168 x 132,
produced by vlfeat's sift binary; stored as float32 npz in tests/golden)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_split_matches_numpy(golden):
    from spectavi_amd import feature
    table = golden("sift_sur_ogre_table.npz")["table"]
    geom, desc = feature.split_sift_table(table)
    assert np.array_equal(geom, table[:, :4])
    assert np.array_equal(desc, table[:, 4:].astype(np.uint8))
    assert desc.max() <= 255 and (desc == 0).mean() > 0.1  # real SIFT: many zero bins


def test_real_descriptors_l1k2_and_pipeline(oracle, golden):
    """Real-distribution descriptors through L1 2-NN (vs oracle), ratio test and coordinate
    gather, all on device."""
    import torch
    from spectavi_amd import device
    table = golden("sift_sur_ogre_table.npz")["table"]
    rng = np.random.default_rng(3)
    # second 'image': a shuffled, slightly perturbed copy of 800 keypoints + 300 unrelated rows
    perm = rng.permutation(1168)[:800]
    t2 = table[perm].copy()
    t2[:, 4:] = np.clip(t2[:, 4:] + rng.integers(-2, 3, (800, 128)), 0, 255)
    t2[:, :2] += 5.0
    extra = table[rng.permutation(1168)[:300]].copy()
    extra[:, 4:] = rng.integers(0, 60, (300, 128))
    t2 = np.vstack([t2, extra]).astype(np.float32)

    gx, dx = device.split_sift_table(torch.from_numpy(table).cuda())
    gy, dy = device.split_sift_table(torch.from_numpy(t2).cuda())
    idx, dist = device.l1k2(dx, dy)
    matches, count = device.ratio_test(idx, dist, 1.75)
    x0, x1 = device.match_coordinates(gx, gy, matches, count)
    torch.cuda.synchronize()

    oidx, odist = oracle.nn_bruteforcel1k2(table[:, 4:].astype(np.uint8), t2[:, 4:].astype(np.uint8), nthreads=8)
    assert np.array_equal(idx.cpu().numpy().view(np.uint64), oidx)
    assert np.array_equal(dist.cpu().numpy(), odist)
    want = oracle.ratio_test_matches(oidx, odist, 1.75)
    n = int(count.item())
    assert n == len(want) and np.array_equal(matches[:n].cpu().numpy(), want)
    # planted copies are recovered: >= 95 % of the 800 perturbed keypoints match their source
    m = matches[:n].cpu().numpy()
    planted = m[m[:, 0] < 800]
    assert len(planted) >= 760 and np.mean(perm[planted[:, 0]] == planted[:, 1]) > 0.99
    x0h, x1h = x0[:n].cpu().numpy(), x1[:n].cpu().numpy()
    assert np.array_equal(x0h[:, :2], table[m[:, 1], :2].astype(np.float64)) and np.all(x0h[:, 2] == 1)
    assert np.array_equal(x1h[:, :2], t2[m[:, 0], :2].astype(np.float64)) and np.all(x1h[:, 2] == 1)


@pytest.mark.parametrize("rows,dim,seed", [(1168, 132, 0), (200, 144, 1), (5000, 128, 2), (33, 7, 3), (100003, 132, 4), (1576, 20, 5), (1047, 16, 6)])
def test_normalize_bit_identical_to_numpy(golden, rows, dim, seed):
    """Device normalisation == the numpy front-end function (reference spectavi/feature.py:384-407)
    bit for bit on float32 input, including the real SIFT table (all 132 columns, as the
    reference's example feeds it, example/ex01_essential_estimation.py:92-93)."""
    from spectavi_amd import feature
    if seed == 0:
        x = golden("sift_sur_ogre_table.npz")["table"]
    else:
        rng = np.random.default_rng(seed)
        x = (rng.standard_normal((rows, dim)) * rng.uniform(0.5, 40, (1, dim)) + rng.uniform(-5, 5, (1, dim))).astype(np.float32)
    want = feature.normalize_to_ubyte_and_multiple_16_dim(x)
    got, u8 = feature.normalize_to_ubyte_and_multiple_16_dim_gpu(x, want_ubyte=True)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.array_equal(u8, (want + 128).astype('uint8'))


def test_normalize_single_column_is_refused():
    """numpy reduces one contiguous column pairwise, not row by row; the device path refuses
    that shape instead of returning a differently rounded mean."""
    import spectavi_amd
    from spectavi_amd import feature
    with pytest.raises(spectavi_amd.SpectaviError):
        feature.normalize_to_ubyte_and_multiple_16_dim_gpu(np.arange(100, dtype=np.float32)[:, None])
    one = feature.normalize_to_ubyte_and_multiple_16_dim_gpu(np.array([[3.0, 5.0]], np.float32))  # one row is fine
    assert one.shape == (1, 16)


def _adversarial_table(rng, rows):
    """One column per way a float32 running sum can behave: monotone integer data crossing binades
    (ties to even once the sum passes 2^24), odd integers and x.5 values (every add a tie), pixel
    coordinates, angles (a random walk around zero: binade changes and cancellation), symmetric
    mixed signs, huge / tiny mixtures, exact cancellation, subnormals, signed zeros, negative drift."""
    cols = [rng.integers(0, 256, rows), rng.integers(0, 65536, rows), 2 * rng.integers(0, 5000, rows) + 1,
            rng.integers(0, 2000, rows) + 0.5, rng.uniform(0, 1280, rows), rng.uniform(1, 8, rows),
            rng.uniform(-np.pi, np.pi, rows), rng.integers(-1000, 1001, rows),
            np.where(rng.random(rows) < 0.01, 1e7, 1e-3), 2.0 ** rng.integers(-10, 20, rows),
            np.tile(np.array([1e6, -1e6, 3.25, -3.0]), rows // 4 + 1)[:rows], rng.integers(0, 1000, rows) * 1e-45,
            np.where(rng.random(rows) < 0.5, 0.0, -0.0), -rng.uniform(0, 300, rows),
            rng.standard_normal(rows) * 1e4 + 1e6, rng.standard_normal(rows) * 3e-2 - 1e-2,
            np.full(rows, 16777216.0), np.full(rows, 3.0), rng.standard_normal(rows) * 100 + 50]
    x = np.stack(cols, axis=1).astype(np.float32)
    x[0, -1] += 1.0
    return x


@pytest.mark.parametrize("rows,seed", [(65536, 1), (65537, 2), (100003, 3), (262144, 4), (1000000, 5), (70001, 6)])
def test_normalize_folded_column_sums_are_numpys_bits(rows, seed):
    """Tables of 65536 rows and more take the folded column sums (adapter.hip: per 1024-row chunk a
    parity -> (increment, parity) function, applied only where the true running sum provably stays
    in one binade).  Every column type that could break the argument, against numpy -- the reference
    function itself (spectavi/feature.py:384-407) -- bit for bit; and against the walking kernel."""
    import os
    from spectavi_amd import feature
    rng = np.random.default_rng(seed)
    x = _adversarial_table(rng, rows)
    with np.errstate(all='ignore'):
        want = feature.normalize_to_ubyte_and_multiple_16_dim(x)
    got, u8 = feature.normalize_to_ubyte_and_multiple_16_dim_gpu(x, want_ubyte=True)
    bad = np.flatnonzero(~np.all((got == want) | (np.isnan(got) & np.isnan(want)), axis=0))
    assert bad.size == 0, "columns %s differ from numpy" % bad.tolist()
    finite = ~np.isnan(want).any(axis=0)
    assert np.array_equal(u8[:, finite], (want[:, finite] + 128).astype('uint8'))
    # the means themselves (a column's normalised values could hide a last-bit difference of its mean)
    import torch
    from spectavi_amd import device as spv
    xt = torch.from_numpy(x).cuda()
    dev = spv.normalize(xt).cpu().numpy()
    assert np.array_equal(dev, got, equal_nan=True)


def test_normalize_with_inf_and_nan_columns():
    from spectavi_amd import feature
    rng = np.random.default_rng(9)
    x = rng.uniform(0, 100, (90000, 6)).astype(np.float32)
    x[10000, 1] = np.inf
    x[7000, 2] = np.nan
    x[89999, 3] = -np.inf
    x[0, 4] = np.nan
    with np.errstate(all='ignore'):
        want = feature.normalize_to_ubyte_and_multiple_16_dim(x)
    got = feature.normalize_to_ubyte_and_multiple_16_dim_gpu(x)
    assert np.array_equal(got, want, equal_nan=True)
