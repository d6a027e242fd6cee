"""GPU parity: ratio test + ordered match compaction vs the numpy statement of the
reference's pipeline step (example/ex01_essential_estimation.py:102-106)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U64MAX = np.iinfo(np.uint64).max


@pytest.mark.parametrize("n,seed", [(0, 1), (1, 2), (255, 3), (256, 4), (257, 5), (100_003, 6), (1_500_000, 7)])
def test_int_distances(oracle, n, seed):
    from spectavi_amd import feature
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 1 << 20, (n, 2)).astype(np.uint64)
    dist = np.sort(rng.integers(0, 40, (n, 2)).astype(np.int32), axis=1)  # many zeros and ties
    if n > 10:
        dist[3] = (0, 0)            # 0/0 -> NaN -> fails
        dist[4] = (0, 7)            # 7/0 -> inf -> passes
        dist[5] = (9, 2 ** 31 - 1)  # second neighbour missing -> passes
        idx[6, 0] = U64MAX          # no neighbour at all -> never passes
        dist[6] = (2 ** 31 - 1, 2 ** 31 - 1)
    for ratio in (1.0, 1.75, 3.0):
        got = feature.ratio_test_matches(idx, dist, ratio)
        want = oracle.ratio_test_matches(idx, dist, ratio)
        assert got.dtype == np.int32 and np.array_equal(got, want)


def test_float_distances_and_device_path(oracle):
    import torch
    from spectavi_amd import device, feature
    rng = np.random.default_rng(9)
    n = 70_001
    idx = rng.integers(0, 1 << 20, (n, 2)).astype(np.uint64)
    dist = np.sort(rng.integers(0, 3000, (n, 2)), axis=1).astype(np.float32)
    dist[10] = (5.0, 2147483648.0)
    want = oracle.ratio_test_matches(idx, dist, 1.75)
    assert np.array_equal(feature.ratio_test_matches(idx, dist, 1.75), want)
    m, c = device.ratio_test(torch.from_numpy(idx.view(np.int64)).cuda(), torch.from_numpy(dist).cuda(), 1.75)
    torch.cuda.synchronize()
    assert int(c.item()) == len(want) and np.array_equal(m[:len(want)].cpu().numpy(), want)


def test_pipeline_l1k2_then_ratio(oracle):
    """NN path -> ratio test entirely on device, against oracle + numpy."""
    import torch
    from spectavi_amd import device
    rng = np.random.default_rng(21)
    x = rng.integers(0, 256, (5000, 128), dtype=np.uint8)
    y = rng.integers(0, 256, (3000, 128), dtype=np.uint8)
    y[:1000] = np.clip(x[rng.integers(0, 5000, 1000)].astype(np.int32) + rng.integers(-4, 5, (1000, 128)), 0, 255)
    idx, dist = device.l1k2(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    m, c = device.ratio_test(idx, dist, 1.75)
    torch.cuda.synchronize()
    oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    want = oracle.ratio_test_matches(oidx, odist, 1.75)
    assert int(c.item()) == len(want) >= 900
    assert np.array_equal(m[:len(want)].cpu().numpy(), want)
