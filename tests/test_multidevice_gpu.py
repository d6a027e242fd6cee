"""GPU: in-process sharding of the host-array entry points over a device list.  The box has
one GPU, so the list names device 0 three times (allowed): this exercises the shard bounds,
the per-shard threads and the direct writes into output slices; results must not change."""
import os

import numpy as np
import pytest

from tests import dlt_checks as dc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def three_shards():
    import spectavi_amd
    spectavi_amd.set_devices([0, 0, 0])
    yield
    spectavi_amd.set_devices([0])


def test_l1k2_cascade_dlt_sharded(oracle, three_shards):
    from spectavi_amd import feature, mvg
    rng = np.random.default_rng(77)
    x = rng.integers(0, 256, (3000, 128), dtype=np.uint8)
    y = rng.integers(0, 256, (1001, 128), dtype=np.uint8)  # ragged over 3 shards
    idx, dist = feature.nn_bruteforcel1k2(x, y)
    oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    idx, dist = feature.nn_bruteforcel1k2(x, y[:2])         # fewer rows than devices
    assert np.array_equal(idx, oidx[:2]) and np.array_equal(dist, odist[:2])

    xf, yf = x.astype(np.float32) - 128, y.astype(np.float32) - 128
    d = rng.standard_normal((2, 128, 8)).astype(np.float32)
    cidx, cdist, ncand = feature.nn_cascading_hash_with_dict(xf, yf, d, g=2, return_ncand=True)
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(xf, yf, 8, 2, 2, d)
    assert np.array_equal(cidx, oidx) and np.array_equal(cdist, odist) and np.array_equal(ncand, oncand)

    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((10007, 4))
    X = mvg.dlt_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T)
    assert np.max(np.abs(X - oracle.dlt_mirror_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T))) <= 1e-12
    e = mvg.dlt_reprojection_error(P0, P1, Xw @ P0.T, Xw @ P1.T)
    assert e.shape == (10007, 1) and float(e.max()) < 1e-6


def test_bad_device_is_reported():
    import spectavi_amd
    from spectavi_amd import feature
    spectavi_amd.set_devices([0, 99])
    try:
        with pytest.raises(spectavi_amd.SpectaviError):
            feature.nn_bruteforcel1k2(np.zeros((8, 16), np.uint8), np.zeros((8, 16), np.uint8))
    finally:
        spectavi_amd.set_devices([0])


def test_concurrent_callers(oracle):
    """ctypes releases the GIL: four Python threads inside libspectavi at once (per-thread
    streams, shared device-buffer cache) must each get their own correct result."""
    import threading
    from spectavi_amd import feature, mvg
    rng = np.random.default_rng(31)
    jobs = []
    for k in range(4):
        x = rng.integers(0, 256, (2000 + 300 * k, 128), dtype=np.uint8)
        y = rng.integers(0, 256, (1500 + 200 * k, 128), dtype=np.uint8)
        jobs.append((x, y, oracle.nn_bruteforcel1k2(x, y, nthreads=4)))
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((50000, 4))
    xa, xb = Xw @ P0.T, Xw @ P1.T  # once: a threaded BLAS need not reproduce its own bits
    want_X = oracle.dlt_mirror_triangulate(P0, P1, xa, xb)
    errors = []

    def work(k):
        try:
            for _ in range(5):
                x, y, (oi, od) = jobs[k]
                i, d = feature.nn_bruteforcel1k2(x, y)
                assert np.array_equal(i, oi) and np.array_equal(d, od)
                X = mvg.dlt_triangulate(P0, P1, xa, xb)
                assert np.max(np.abs(X - want_X)) <= 1e-12
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_distributed_wrappers_single_rank_rccl(oracle):
    """spectavi_amd.sharded with its default (HIP) local functions and an RCCL process group of
    one rank on the box's GPU: the shard / pack / gather / unpack path on device tensors.  (Two
    and more ranks run the same code; the gather itself is covered with gloo on CPU.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from spectavi_amd import sharded
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(5)
        x = rng.integers(0, 256, (700, 128), dtype=np.uint8)
        y = rng.integers(0, 256, (333, 128), dtype=np.uint8)
        idx, d = sharded.nn_bruteforcel1k2_sharded(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), 333)
        oi, od = oracle.nn_bruteforcel1k2(x, y)
        assert np.array_equal(idx.cpu().numpy().view(np.uint64), oi) and np.array_equal(d.cpu().numpy(), od)

        xf, yf = x.astype(np.float32) - 128, y.astype(np.float32) - 128
        hd = rng.standard_normal((2, 128, 7)).astype(np.float32)
        ci, cd = sharded.nn_cascading_hash_sharded(torch.from_numpy(xf).to(dev), torch.from_numpy(yf).to(dev),
                                                   torch.from_numpy(hd).to(dev), 333, g=2)
        wi, wd, _, _ = oracle.nn_cascading_hash(xf, yf, 7, 2, 2, hd)
        assert np.array_equal(ci.cpu().numpy().view(np.uint64), wi) and np.array_equal(cd.cpu().numpy(), wd)

        P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
        Xw = rng.standard_normal((501, 4))
        px, pxp = torch.from_numpy(Xw @ P0.T).to(dev), torch.from_numpy(Xw @ P1.T).to(dev)
        X = sharded.dlt_sharded(P0, P1, px, pxp, 501)
        E = sharded.dlt_sharded(P0, P1, px, pxp, 501, want_error=True)
        assert np.max(np.abs(X.cpu().numpy() - oracle.dlt_mirror_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T))) <= 1e-12
        assert E.shape == (501, 1) and float(E.max()) < 1e-6
    finally:
        dist.destroy_process_group()


@pytest.fixture
def rccl_gather():
    """Force the in-library RCCL gather for a clique of one (this box has one GPU): the shards'
    16-byte records are packed on the device, gathered with ncclGather on the root, widened by the
    root's kernel and copied out once -- the code path that runs unchanged over 8 GPUs."""
    import spectavi_amd
    spectavi_amd.set_devices([0])
    spectavi_amd.set_gather_mode("rccl")
    yield
    spectavi_amd.set_gather_mode("auto")


def test_in_library_rccl_gather_clique_of_one(oracle, rccl_gather):
    """nn_bruteforcel1k2, nn_cascading_hash, dlt_triangulate and dlt_reprojection_error through the
    unchanged reference-style front-end with SPV_GATHER_RCCL: results identical to the oracle, the
    collective really ran (spv_profile_read("gather") counts launches), sentinels survive the
    32-bit records, and no PyTorch is involved in the exchange."""
    from spectavi_amd import device, feature, mvg
    device.profile_reset()
    device.profile_enable(True)
    rng = np.random.default_rng(2026)
    x = rng.integers(0, 256, (3000, 128), dtype=np.uint8)
    y = rng.integers(0, 256, (1001, 128), dtype=np.uint8)
    idx, dist = feature.nn_bruteforcel1k2(x, y)
    oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    # fewer than two database rows: (size_t)-1 / INT_MAX travel through the records as -1
    idx1, dist1 = feature.nn_bruteforcel1k2(x[:1], y[:5])
    oidx1, odist1 = oracle.nn_bruteforcel1k2(x[:1], y[:5])
    assert np.array_equal(idx1, oidx1) and np.array_equal(dist1, odist1)
    assert idx1[0, 1] == np.iinfo(np.uint64).max and dist1[0, 1] == np.iinfo(np.int32).max

    xf, yf = x.astype(np.float32) - 128, y.astype(np.float32) - 128
    d = rng.standard_normal((2, 128, 8)).astype(np.float32)
    cidx, cdist, ncand = feature.nn_cascading_hash_with_dict(xf, yf, d, g=2, return_ncand=True)
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(xf, yf, 8, 2, 2, d)
    assert np.array_equal(cidx, oidx) and np.array_equal(cdist, odist) and np.array_equal(ncand, oncand)

    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((10007, 4))
    X = mvg.dlt_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T)
    mX = oracle.dlt_mirror_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T)
    oX = oracle.dlt_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T)
    assert np.max(np.abs(X - np.sign(np.einsum("ni,ni->n", X, oX))[:, None] * oX)) < 1e-9
    e = mvg.dlt_reprojection_error(P0, P1, Xw @ P0.T, Xw @ P1.T)
    assert e.shape == (10007, 1)
    dc.check_against_mirror(X, mX, P0, P1, Xw @ P0.T, Xw @ P1.T, E=e,
                            mE=oracle.dlt_mirror_reprojection_error(P0, P1, Xw @ P0.T, Xw @ P1.T), what="clique of one")
    device.profile_enable(False)
    launches, ms = device.profile_read("gather")
    assert launches == 6 and ms > 0.0          # 2 x L1 + cascade records + cascade ncand + 2 x DLT
    assert device.profile_read("gather_widen")[0] == 3


def test_rccl_gather_rejects_a_device_listed_twice(rccl_gather):
    import spectavi_amd
    from spectavi_amd import feature
    spectavi_amd.set_devices([0, 0])
    try:
        with pytest.raises(spectavi_amd.SpectaviError, match="listed twice"):
            feature.nn_bruteforcel1k2(np.zeros((8, 16), np.uint8), np.zeros((8, 16), np.uint8))
    finally:
        spectavi_amd.set_devices([0])


def test_misaligned_device_pointers_are_rejected():
    """spv_l1k2_device / spv_cascade_device read rows as 16-byte vectors: a tensor view that starts
    off a 16-byte boundary must come back as SPV_ERR_INVALID, not as a fault."""
    import torch
    from spectavi_amd._lib import clib, SPV_ERR_INVALID
    from spectavi_amd import device  # noqa: F401  (declares argtypes)
    buf = torch.zeros(1000 * 128 + 64, dtype=torch.uint8, device="cuda")
    x = buf[:1000 * 128]
    idx = torch.empty((1000, 2), dtype=torch.int64, device="cuda")
    dist = torch.empty((1000, 2), dtype=torch.int32, device="cuda")
    n = clib.spv_l1k2_workspace_bytes(1000, 1000, 128)
    ws = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    ok = clib.spv_l1k2_device(x.data_ptr(), x.data_ptr(), 1000, 1000, 128, idx.data_ptr(), dist.data_ptr(),
                              ws.data_ptr(), n, None)
    assert ok == 0
    for dx, dy, dw in ((4, 0, 0), (0, 8, 0), (0, 0, 4)):
        st = clib.spv_l1k2_device(x.data_ptr() + dx, x.data_ptr() + dy, 1000, 1000, 128, idx.data_ptr(),
                                  dist.data_ptr(), ws.data_ptr() + dw, n, None)
        assert st == SPV_ERR_INVALID and b"16-byte aligned" in clib.spv_last_error()
    st = clib.spv_l1k2_device(x.data_ptr(), x.data_ptr(), 1000, 1000, 128, idx.data_ptr() + 4, dist.data_ptr(),
                              ws.data_ptr(), n, None)
    assert st == SPV_ERR_INVALID
    xf = torch.zeros(200 * 128 + 16, dtype=torch.float32, device="cuda")
    hd = torch.zeros(2 * 128 * 6 + 4, dtype=torch.float32, device="cuda")
    cn = clib.spv_cascade_workspace_bytes(200, 200, 128, 6, 2, 2)
    cws = torch.empty(cn, dtype=torch.uint8, device="cuda")
    fd = torch.empty((200, 2), dtype=torch.float32, device="cuda")
    for off_x, off_d in ((4, 0), (0, 4)):
        st = clib.spv_cascade_device(xf.data_ptr() + off_x, xf.data_ptr(), 200, 200, 128, 6, 2, 2, hd.data_ptr() + off_d,
                                     idx.data_ptr(), fd.data_ptr(), None, cws.data_ptr(), cn, None)
        assert st == SPV_ERR_INVALID and b"16-byte aligned" in clib.spv_last_error()
    # rows wider than the refine kernels take are refused before anything is enqueued
    st = clib.spv_cascade_device(xf.data_ptr(), xf.data_ptr(), 2, 2, 2064, 6, 2, 2, hd.data_ptr(), idx.data_ptr(),
                                 fd.data_ptr(), None, cws.data_ptr(), cn, None)
    assert st == SPV_ERR_INVALID and b"2048" in clib.spv_last_error()


def test_host_entry_points_restore_the_callers_device():
    """A host-array call selects its device(s) with hipSetDevice; the caller's current device --
    torch's, here -- must be what it was when the call returns (with one GPU: still device 0, and
    the call must not have been confused by a caller stream / device context either)."""
    import torch
    from spectavi_amd import feature
    torch.cuda.set_device(0)
    before = torch.cuda.current_device()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        feature.nn_bruteforcel1k2(np.zeros((64, 16), np.uint8), np.ones((8, 16), np.uint8))
    assert torch.cuda.current_device() == before
    t = torch.ones(4, device="cuda") * 2            # torch still works on its device afterwards
    assert float(t.sum()) == 8.0


@pytest.fixture
def three_ranks_copy_transport():
    """The gathered multi-rank layout on a one-GPU box: device 0 listed three times, records moved into
    the root's [rank][max_cnt] buffer by the peer-copy transport (RCCL refuses a duplicated device).
    Everything but the collective itself is the code that runs over RCCL on eight GPUs: ragged shard
    bounds, padding to the largest shard, rank order, the widening kernel, the ncand and DLT segments."""
    import spectavi_amd
    spectavi_amd.set_devices([0, 0, 0])
    spectavi_amd.set_gather_mode("copy")
    yield
    spectavi_amd.set_gather_mode("auto")
    spectavi_amd.set_devices([0])


def test_gathered_layout_three_ranks(oracle, three_ranks_copy_transport):
    from spectavi_amd import device, feature, mvg
    device.profile_reset()
    device.profile_enable(True)
    rng = np.random.default_rng(303)
    x = rng.integers(0, 256, (3000, 128), dtype=np.uint8)
    for nq in (1001, 1000, 2, 4):                           # ragged, even, fewer rows than ranks, one extra
        y = rng.integers(0, 256, (nq, 128), dtype=np.uint8)
        idx, dist = feature.nn_bruteforcel1k2(x, y)
        oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist), nq
    idx1, dist1 = feature.nn_bruteforcel1k2(x[:1], y)      # sentinels through three ranks' records
    oidx1, odist1 = oracle.nn_bruteforcel1k2(x[:1], y)
    assert np.array_equal(idx1, oidx1) and np.array_equal(dist1, odist1)

    y = rng.integers(0, 256, (1001, 128), dtype=np.uint8)
    xf, yf = x.astype(np.float32) - 128, y.astype(np.float32) - 128
    d = rng.standard_normal((2, 128, 8)).astype(np.float32)
    cidx, cdist, ncand = feature.nn_cascading_hash_with_dict(xf, yf, d, g=2, return_ncand=True)
    oidx, odist, oncand, _ = oracle.nn_cascading_hash(xf, yf, 8, 2, 2, d)
    assert np.array_equal(cidx, oidx) and np.array_equal(cdist, odist) and np.array_equal(ncand, oncand)

    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((10007, 4))
    X = mvg.dlt_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T)
    e = mvg.dlt_reprojection_error(P0, P1, Xw @ P0.T, Xw @ P1.T)
    dc.check_against_mirror(X, oracle.dlt_mirror_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T), P0, P1, Xw @ P0.T, Xw @ P1.T, E=e,
                            mE=oracle.dlt_mirror_reprojection_error(P0, P1, Xw @ P0.T, Xw @ P1.T), what="three ranks")
    device.profile_enable(False)
    assert device.profile_read("gather")[0] == 5 + 2 + 2    # five L1 calls, cascade records + ncand, two DLT calls


def test_gathered_device_resident_form(oracle):
    """spv_l1k2_gathered_device (what `bench.py --mode inlib` times): replicas and query shards
    already in HBM, one rank per listed device, records gathered and widened on the first.  On a
    one-GPU box: a real RCCL clique of one, and three ranks on device 0 over the peer-copy transport
    with ragged shards -- both bit-equal to the oracle; bad shard splits are refused."""
    import torch
    import spectavi_amd
    from spectavi_amd import device
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.integers(0, 256, (4099, 128), dtype=np.uint8)).cuda()
    for nq, G, transport in ((1001, 1, "rccl"), (1001, 3, "copy"), (1002, 3, "copy"), (5, 4, "copy")):
        y = rng.integers(0, 256, (nq, 128), dtype=np.uint8)
        b = device.shard_bounds(nq, G)
        assert b[0] == 0 and b[-1] == nq and all(0 <= b[i + 1] - b[i] - nq // G <= 1 for i in range(G))
        ys = [torch.from_numpy(y[b[r]:b[r + 1]]).cuda() for r in range(G)]
        idx, dist = device.l1k2_gathered([x] * G, ys, transport=transport)
        oidx, odist = oracle.nn_bruteforcel1k2(x.cpu().numpy(), y, nthreads=8)
        assert np.array_equal(idx.cpu().numpy().view(np.uint64), oidx) and np.array_equal(dist.cpu().numpy(), odist), (nq, G)
    # sentinels through the records
    idx, dist = device.l1k2_gathered([x[:1]] * 2, [torch.from_numpy(y[:3]).cuda(), torch.from_numpy(y[3:5]).cuda()], transport="copy")
    assert bool((idx[:, 1] == -1).all()) and bool((dist[:, 1] == np.iinfo(np.int32).max).all())
    with pytest.raises(ValueError, match="balanced"):
        device.l1k2_gathered([x, x], [torch.from_numpy(y[:1]).cuda(), torch.from_numpy(y[1:5]).cuda()], transport="copy")
    with pytest.raises(spectavi_amd.SpectaviError, match="listed twice"):
        device.l1k2_gathered([x, x], [torch.from_numpy(y[:3]).cuda(), torch.from_numpy(y[3:5]).cuda()], transport="rccl")


def test_gathered_device_resident_cascade_and_dlt(oracle):
    """spv_cascade_gathered_device / spv_dlt_gathered_device: the cascade hash and the DLT sharded inside
    one process with replicas and shards resident -- a real RCCL clique of one and three ranks on
    device 0 over peer copies with ragged shards; idx / dist / ncand bit-equal to the oracle, DLT rows
    bit-equal to the one-device call."""
    import torch
    from spectavi_amd import device
    rng = np.random.default_rng(78)
    xf = rng.integers(-128, 128, (3001, 128)).astype(np.float32)
    d = rng.standard_normal((2, 128, 8)).astype(np.float32)
    tx, td = torch.from_numpy(xf).cuda(), torch.from_numpy(d).cuda()
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    for nq, G, transport in ((1001, 1, "rccl"), (1001, 3, "copy"), (4, 3, "copy")):
        yf = rng.integers(-128, 128, (nq, 128)).astype(np.float32)
        yf[: nq // 2] = np.clip(xf[rng.integers(0, 3001, nq // 2)] + rng.integers(-2, 3, (nq // 2, 128)), -128, 127)
        b = device.shard_bounds(nq, G)
        ys = [torch.from_numpy(yf[b[r]:b[r + 1]].copy()).cuda() for r in range(G)]
        idx, dist, ncand = device.cascade_gathered([tx] * G, ys, [td] * G, g=2, transport=transport, want_ncand=True)
        oidx, odist, oncand, _ = oracle.nn_cascading_hash(xf, yf, 8, 2, 2, d)
        assert np.array_equal(idx.cpu().numpy().view(np.uint64), oidx) and np.array_equal(dist.cpu().numpy(), odist)
        assert np.array_equal(ncand.cpu().numpy(), oncand)
        Xw = rng.standard_normal((nq, 4))
        x, xp = Xw @ P0.T, Xw @ P1.T
        xs = [torch.from_numpy(x[b[r]:b[r + 1]].copy()).cuda() for r in range(G)]
        xps = [torch.from_numpy(xp[b[r]:b[r + 1]].copy()).cuda() for r in range(G)]
        one = device.dlt_triangulate(P0, P1, torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda())
        onee = device.dlt_reprojection_error(P0, P1, torch.from_numpy(x).cuda(), torch.from_numpy(xp).cuda())
        assert torch.equal(device.dlt_gathered(P0, P1, xs, xps, transport=transport), one)
        assert torch.equal(device.dlt_gathered(P0, P1, xs, xps, want_error=True, transport=transport).reshape(-1), onee.reshape(-1))
        dc.check_against_oracle(one.cpu().numpy(), oracle.dlt_triangulate(P0, P1, x, xp), P0, P1, x, xp, what="gathered dlt")


def _hip_rank(rank, world, port, nq, out_path):
    """One of `world` processes sharing GPU 0: HIP local compute on its query shard, the records
    gathered over gloo (RCCL cannot span ranks that share one device)."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spectavi_amd import device
    from spectavi_amd.sharded import nn_bruteforcel1k2_sharded, nn_cascading_hash_sharded, dlt_sharded, shard_bounds
    rng = np.random.default_rng(123)
    x = rng.integers(0, 256, (5000, 128), dtype=np.uint8)
    y = rng.integers(0, 256, (nq, 128), dtype=np.uint8)
    d = rng.standard_normal((2, 128, 9)).astype(np.float32)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((nq, 4))
    lo, hi = shard_bounds(nq, world, rank)
    dev = torch.device("cuda", 0)

    def l1(xt, yt):  # HIP on the device, records back on the CPU for the gloo gather
        i, dd = device.l1k2(xt.to(dev), yt.to(dev))
        return i.cpu(), dd.cpu()

    def casc(xt, yt, dt, g):
        i, dd = device.cascade(xt.to(dev), yt.to(dev), dt.to(dev), g=g)
        return i.cpu(), dd.cpu()

    def tri(a, b, c, e):
        return device.dlt_triangulate(a, b, c.to(dev), e.to(dev)).cpu()

    idx, dist_ = nn_bruteforcel1k2_sharded(torch.from_numpy(x), torch.from_numpy(y[lo:hi]), nq, local_fn=l1)
    xf, yf = torch.from_numpy(x.astype(np.float32) - 128), torch.from_numpy(y[lo:hi].astype(np.float32) - 128)
    cidx, cdist = nn_cascading_hash_sharded(xf, yf, torch.from_numpy(d), nq, g=2, local_fn=casc)
    X = dlt_sharded(P0, P1, torch.from_numpy((Xw @ P0.T)[lo:hi].copy()), torch.from_numpy((Xw @ P1.T)[lo:hi].copy()), nq,
                    local_fn=tri)
    if rank == 0:
        np.savez(out_path, idx=idx.numpy(), dist=dist_.numpy(), cidx=cidx.numpy(), cdist=cdist.numpy(), X=X.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_ranks_hip_compute_and_gather(oracle, tmp_path):
    """Two ranks (processes) with the HIP kernels as their local compute and a real N > 1 gather of
    the 16-byte records (gloo: both processes share the box's one GPU): the sharded wrappers of
    spectavi_amd/sharded.py end to end, ragged shards included."""
    import torch.multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    nq = 2001
    out = str(tmp_path / "ranks.npz")
    mp.spawn(_hip_rank, args=(2, port, nq, out), nprocs=2, join=True)
    got = np.load(out)
    rng = np.random.default_rng(123)
    x = rng.integers(0, 256, (5000, 128), dtype=np.uint8)
    y = rng.integers(0, 256, (nq, 128), dtype=np.uint8)
    d = rng.standard_normal((2, 128, 9)).astype(np.float32)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((nq, 4))
    oidx, odist = oracle.nn_bruteforcel1k2(x, y, nthreads=8)
    assert np.array_equal(got["idx"].view(np.uint64), oidx) and np.array_equal(got["dist"], odist)
    ci, cd, _, _ = oracle.nn_cascading_hash(x.astype(np.float32) - 128, y.astype(np.float32) - 128, 9, 2, 2, d)
    assert np.array_equal(got["cidx"].view(np.uint64), ci) and np.array_equal(got["cdist"], cd)
    assert np.max(np.abs(got["X"] - oracle.dlt_mirror_triangulate(P0, P1, Xw @ P0.T, Xw @ P1.T))) <= 1e-12


def _hip_fit_rank(rank, world, port, out_path):
    """One of `world` processes sharing GPU 0: the HIP RANSAC fit on its block of tries, the per-rank
    winners ranked over gloo."""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spectavi_amd import mvg
    from spectavi_amd.sharded import ransac_fit_sharded
    from tests import mvg_checks as mc
    rng = np.random.default_rng(5)
    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=400, outlier_fraction=0.45)
    samples = mvg.ransac_sample(99, 400, 900)
    out = {}
    for name, kw in (("success", dict(required_percent_inliers=0.5, find_best_even_in_failure=False)),
                     ("best", dict(required_percent_inliers=0.99, find_best_even_in_failure=True))):
        kw.update(reprojection_error_allowed=1e-3, singular_value_ratio_allowed=3e-2)
        r = ransac_fit_sharded(x0, x1, samples, **kw)
        out[name + "_try"] = r['best_try']
        out[name + "_root"] = r['best_root']
        out[name + "_ok"] = r['success']
        out[name + "_F"] = r['essential']
        out[name + "_P"] = r['camera']
        out[name + "_idx"] = r['inlier_idx']
    if rank == world - 1:
        np.savez(out_path, **out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_ranks_shard_the_ransac_tries(tmp_path):
    """Three processes sharing GPU 0, each fitting its block of the 900 tries with the HIP path; the ranked
    result is what one process gets from all 900 (the first success in try order / the earliest best model)."""
    import socket
    import torch.multiprocessing as mp
    from spectavi_amd import mvg
    from tests import mvg_checks as mc
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "fit.npz")
    mp.spawn(_hip_fit_rank, args=(3, port, out), nprocs=3, join=True)
    got = np.load(out)
    rng = np.random.default_rng(5)
    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=400, outlier_fraction=0.45)
    samples = mvg.ransac_sample(99, 400, 900)
    for name, kw in (("success", dict(required_percent_inliers=0.5, find_best_even_in_failure=False)),
                     ("best", dict(required_percent_inliers=0.99, find_best_even_in_failure=True))):
        kw.update(reprojection_error_allowed=1e-3, singular_value_ratio_allowed=3e-2)
        one = mvg.ransac_fit(x0, x1, samples=samples, **kw)
        assert one['best_try'] >= 0
        assert int(got[name + "_try"]) == one['best_try'] and int(got[name + "_root"]) == one['best_root']
        assert bool(got[name + "_ok"]) == one['success']
        assert np.array_equal(got[name + "_F"], one['essential']) and np.array_equal(got[name + "_P"], one['camera'])
        assert np.array_equal(got[name + "_idx"], one['inlier_idx'])
    assert bool(got["success_ok"]) and not bool(got["best_ok"])
