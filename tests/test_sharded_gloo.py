"""CPU-only: the query-sharding + gather path with world_size 2 over gloo.  The local
compute is injected (the CPU oracle) because there is no GPU in this container; the HIP
path is the default `local_fn` and is covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    from spectavi_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 1000, 4_000_000):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_pack_roundtrip():
    from spectavi_amd.sharded import pack_records, unpack_records
    idx = torch.tensor([[3, -1], [0, 2 ** 31 - 2]], dtype=torch.int64)
    d = torch.tensor([[5, 2 ** 31 - 1], [0, 32640]], dtype=torch.int32)
    i2, d2 = unpack_records(pack_records(idx, d))
    assert torch.equal(i2, idx) and torch.equal(d2, d)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from spectavi_amd.sharded import nn_bruteforcel1k2_sharded, shard_bounds
    rng = np.random.default_rng(11)
    x = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    y = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    lo, hi = shard_bounds(nq, world, rank)

    def local_fn(xt, yt):
        idx, d = o.nn_bruteforcel1k2(xt.numpy(), yt.numpy())
        return torch.from_numpy(idx.view(np.int64)), torch.from_numpy(d)

    idx, d = nn_bruteforcel1k2_sharded(torch.from_numpy(x), torch.from_numpy(y[lo:hi]), nq,
                                       local_fn=local_fn)
    if rank == 0:
        want_i, want_d = o.nn_bruteforcel1k2(x, y)
        ok = np.array_equal(idx.numpy().view(np.uint64), want_i) and np.array_equal(d.numpy(), want_d)
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        assert idx is None and d is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gather_world2_gloo(tmp_path):
    for nq in (101, 64):  # ragged and even shards
        out = tmp_path / ("res%d.txt" % nq)
        mp.spawn(_worker, args=(2, _free_port(), nq, str(out)), nprocs=2, join=True)
        assert out.read_text() == "ok"


def _worker_cascade_dlt(rank, world, port, nq, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from spectavi_amd.sharded import nn_cascading_hash_sharded, dlt_sharded, shard_bounds
    rng = np.random.default_rng(23)
    x = (rng.integers(0, 256, (400, 32)).astype(np.float32) - 128)
    y = (rng.integers(0, 256, (nq, 32)).astype(np.float32) - 128)
    hd = rng.standard_normal((3, 32, 6)).astype(np.float32)
    lo, hi = shard_bounds(nq, world, rank)

    def cascade_fn(xt, yt, dt, g):
        idx, d, _, _ = o.nn_cascading_hash(xt.numpy(), yt.numpy(), dt.shape[2], dt.shape[0], g, dt.numpy())
        return torch.from_numpy(idx.view(np.int64)), torch.from_numpy(d)

    idx, d = nn_cascading_hash_sharded(torch.from_numpy(x), torch.from_numpy(y[lo:hi]), torch.from_numpy(hd), nq,
                                       g=2, local_fn=cascade_fn)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((nq, 4))
    px, pxp = Xw @ P0.T, Xw @ P1.T

    def tri_fn(a, b, c, e):
        return torch.from_numpy(o.dlt_triangulate(a, b, c.numpy(), e.numpy()))

    def err_fn(a, b, c, e):
        return torch.from_numpy(o.dlt_reprojection_error(a, b, c.numpy(), e.numpy()))

    X = dlt_sharded(P0, P1, torch.from_numpy(px[lo:hi]), torch.from_numpy(pxp[lo:hi]), nq, local_fn=tri_fn)
    E = dlt_sharded(P0, P1, torch.from_numpy(px[lo:hi]), torch.from_numpy(pxp[lo:hi]), nq, want_error=True,
                    local_fn=err_fn)
    if rank == 0:
        wi, wd, _, _ = o.nn_cascading_hash(x, y, 6, 3, 2, hd)
        ok = np.array_equal(idx.numpy().view(np.uint64), wi) and np.array_equal(d.numpy(), wd)
        ok = ok and np.array_equal(X.numpy(), o.dlt_triangulate(P0, P1, px, pxp))
        ok = ok and np.array_equal(E.numpy().reshape(-1), o.dlt_reprojection_error(P0, P1, px, pxp).reshape(-1))
        ok = ok and X.shape == (nq, 4) and E.shape == (nq, 1)
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        assert idx is None and d is None and X is None and E is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_cascade_and_dlt_world2_gloo(tmp_path):
    """Row (e) for the other two paths: cascade queries and DLT correspondences sharded over
    two ranks, results gathered on rank 0 and compared with the unsharded oracle."""
    for nq in (77, 40):
        out = tmp_path / ("cd%d.txt" % nq)
        mp.spawn(_worker_cascade_dlt, args=(2, _free_port(), nq, str(out)), nprocs=2, join=True)
        assert out.read_text() == "ok"


def _worker_fit(rank, world, port, out_path):
    import torch.distributed as dist
    from oracle import oracle as o
    from spectavi_amd import sharded
    from tests import mvg_checks as mc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    rng = np.random.default_rng(77)
    x0, x1, E, out_idx = mc.two_view_scene(rng, npt=150, outlier_fraction=0.3)
    samples = np.stack([rng.choice(np.arange(1, 150), 7, replace=False) for _ in range(90)]).astype(np.int32)

    def local(x0_, x1_, s, **kw):
        return o.ransac_fit(x0_, x1_, s, **kw)

    for name, kw in (("success", dict(required_percent_inliers=0.6, reprojection_error_allowed=1e-3,
                                      find_best_even_in_failure=False, singular_value_ratio_allowed=3e-2)),
                     ("best", dict(required_percent_inliers=0.99, reprojection_error_allowed=1e-3,
                                   find_best_even_in_failure=True, singular_value_ratio_allowed=3e-2)),
                     ("none", dict(required_percent_inliers=0.99, reprojection_error_allowed=1e-3,
                                   find_best_even_in_failure=False, singular_value_ratio_allowed=3e-2))):
        r = sharded.ransac_fit_sharded(x0, x1, samples, local_fn=local, **kw)
        serial = o.ransac_fit(x0, x1, samples, **kw)
        ok = (r['success'] == serial['success'] and r['best_try'] == serial['best_try'] and
              r['best_root'] == serial['best_root'] and np.array_equal(r['inlier_idx'], serial['inlier_idx']) and
              r['inlier_percent'] == serial['inlier_percent'])
        if serial['best_try'] >= 0:
            ok = ok and np.array_equal(r['essential'], serial['essential']) and np.array_equal(r['camera'], serial['camera'])
        else:
            ok = ok and r['essential'] is None
        res[name] = bool(ok)
    if rank == 1:  # every rank holds the answer: let a non-zero rank report
        np.save(out_path, np.array([res["success"], res["best"], res["none"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ransac_tries_sharded_gloo(tmp_path, world):
    """RANSAC tries sharded over ranks (the loop the reference spreads over OpenMP threads): with the CPU
    oracle as the local fit, every rank gets what the serial loop over all tries gives -- first success in
    try order; else, with find_best_even_in_failure, the earliest model with the most inliers; else nothing."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "fit.npy")
    mp.spawn(_worker_fit, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out).all()
