"""The hot path sharded across the GPUs of one node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI).  L1 2-NN and cascade hash shard the
queries and collect packed (idx0, idx1, d0, d1) records with one gather; DLT shards the
correspondences and gathers 32 bytes per point.

Every query row is independent (the reference parallelises exactly this loop
with OpenMP, src/BruteForceNnL1K2.h:92-93), so the only exchange step is the
result gather; the database is replicated on every GPU.  Shard r owns the
contiguous query rows [lo, hi) of `shard_bounds`.
"""
import torch
import torch.distributed as dist


def shard_bounds(nrows, world_size, rank):
    """Contiguous, balanced shards: the first (nrows % world) shards get one extra row."""
    base, extra = divmod(int(nrows), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def pack_records(idx, dist_):
    """(int64 [n,2], int32 [n,2]) -> int32 [n,4] record (idx0, idx1, d0, d1).
    Database indices fit 31 bits (the ABI passes xrows as int); the no-neighbour
    sentinel (size_t)-1 travels as -1."""
    return torch.cat([idx.to(torch.int32), dist_.to(torch.int32)], dim=1).contiguous()


def unpack_records(rec):
    """int32 [n,4] -> (int64 [n,2] with -1 = (size_t)-1, int32 [n,2])."""
    return rec[:, 0:2].to(torch.int64), rec[:, 2:4].contiguous()


def _default_local_fn(x, y):
    from spectavi_amd import device
    return device.l1k2(x, y)


def gather_rows(local, total_rows, group=None, dst=0):
    """Gather row-sharded results on rank `dst`: every rank passes its [hi - lo, k] tensor
    (shard_bounds order); ragged shards are padded to the largest so all ranks send the same
    count (one collective, no size exchange).  Returns the [total_rows, k] tensor on `dst`,
    None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total_rows, world, rank)
    assert local.shape[0] == hi - lo, "local rows do not match shard_bounds"
    max_rows = shard_bounds(total_rows, world, 0)[1]
    send = local.contiguous()
    if send.shape[0] < max_rows:
        pad = torch.zeros((max_rows - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=bufs, dst=dst, group=group)
        parts = []
        for r in range(world):
            rlo, rhi = shard_bounds(total_rows, world, r)
            parts.append(bufs[r][: rhi - rlo])
        return torch.cat(parts, dim=0)
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None


def nn_bruteforcel1k2_sharded(x, y_shard, total_queries, group=None, dst=0, local_fn=None):
    """Run the local shard and gather all shards' results on rank `dst`.

    x            database, replicated on every rank (uint8 [M,D])
    y_shard      this rank's query rows, shard_bounds(total_queries, world, rank)
    local_fn     (x, y) -> (idx int64 [n,2], dist int32 [n,2]); defaults to the HIP
                 kernel (spectavi_amd.device.l1k2).  Tests inject a CPU checker.
    Returns (idx, dist) for all `total_queries` rows on rank dst, (None, None) elsewhere.
    """
    local_fn = local_fn or _default_local_fn
    idx, d = local_fn(x, y_shard)
    rec = gather_rows(pack_records(idx, d), total_queries, group, dst)
    return unpack_records(rec) if rec is not None else (None, None)


def nn_cascading_hash_sharded(x, y_shard, hash_dict, total_queries, g=2, group=None, dst=0, local_fn=None):
    """Cascade-hash 2-NN with the queries sharded like `nn_bruteforcel1k2_sharded`: database,
    hyperplanes (and therefore codes and bucket tables, rebuilt identically on every rank) are
    replicated, each rank probes and refines its own query rows, one gather of 16-byte records
    (the float32 distances travel bit-cast in the int32 record).

    local_fn     (x, y, hash_dict, g) -> (idx int64 [n,2], dist float32 [n,2]); defaults to
                 spectavi_amd.device.cascade.
    Returns (idx int64 [N,2] with -1 = no neighbour, dist float32 [N,2]) on rank dst."""
    if local_fn is None:
        from spectavi_amd import device
        local_fn = lambda a, b, d, gg: device.cascade(a, b, d, g=gg)[:2]  # noqa: E731
    idx, d = local_fn(x, y_shard, hash_dict, g)
    rec = gather_rows(pack_records(idx, d.contiguous().view(torch.int32)), total_queries, group, dst)
    if rec is None:
        return None, None
    i, di = unpack_records(rec)
    return i, di.view(torch.float32)


def dlt_sharded(P0, P1, x_shard, xp_shard, total_points, want_error=False, group=None, dst=0, local_fn=None):
    """Two-view triangulation (or its reprojection error) with the correspondences sharded over
    the ranks; cameras replicated; one gather of 32 (8) bytes per point.

    local_fn     (P0, P1, x, xp) -> float64 [n,4] (or [n,1]); defaults to
                 spectavi_amd.device.dlt_triangulate / dlt_reprojection_error.
    Returns float64 [total_points, 4] (or [total_points, 1]) on rank dst, None elsewhere."""
    if local_fn is None:
        from spectavi_amd import device
        local_fn = device.dlt_reprojection_error if want_error else device.dlt_triangulate
    out = local_fn(P0, P1, x_shard, xp_shard)
    if out.dim() == 1:
        out = out[:, None]
    return gather_rows(out, total_points, group, dst)


def _default_fit_fn(x0, x1, samples, **kw):
    from spectavi_amd import mvg
    return mvg.ransac_fit(x0, x1, samples=samples, **kw)


def ransac_fit_sharded(x0, x1, samples, group=None, local_fn=None, device=None, **kw):
    """RANSAC tries sharded over the ranks (the reference spreads exactly this loop over OpenMP threads,
    src/RansacFitter.h:163): the correspondences x0, x1 (numpy float64 [npt,3]) are replicated, rank r
    evaluates the contiguous tries shard_bounds(len(samples), world, r) of `samples` (int32 [T,7], the same
    array on every rank, e.g. mvg.ransac_sample(seed, npt, T)), and two small collectives rank the
    per-rank results the way the serial loop would: the success with the lowest try index wins; without a
    success (find_best_even_in_failure) the model with the most inliers, the earliest among equals.  Every
    rank returns the winner's dict (mvg.ransac_fit's keys; best_try counts over all of `samples`).

    local_fn(x0, x1, samples, **kw) defaults to the HIP path (mvg.ransac_fit); tests inject the CPU oracle.
    `device`: where the exchanged tensors live (default: cuda for the nccl backend, else cpu)."""
    import numpy as np
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    local_fn = local_fn or _default_fit_fn
    npt = x0.shape[0]
    samples = np.asarray(samples)
    if samples.ndim != 2 or samples.shape[1] != 7 or samples.dtype != np.int32 or samples.shape[0] < 1:
        raise ValueError("samples must be int32 [T,7] with T >= 1 (T < world size leaves some ranks without tries), got %s %s"
                         % (samples.dtype, samples.shape))
    # the same array on every rank, or best_try would index different subsets: a 2-word digest, compared
    digest = torch.tensor([samples.shape[0], int(np.bitwise_xor.reduce(samples.astype(np.int64).reshape(-1) *
                           (np.arange(samples.size, dtype=np.int64) % 8191 + 1)))], dtype=torch.int64, device=device)
    digests = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(digests, digest, group=group)
    if any(not torch.equal(dg, digest) for dg in digests):
        raise ValueError("ransac_fit_sharded: `samples` differs between ranks")
    lo, hi = shard_bounds(len(samples), world, rank)
    r = local_fn(x0, x1, samples[lo:hi], **kw)
    found = r['best_try'] >= 0
    count = len(r['inlier_idx']) if found else 0
    head = torch.tensor([int(r['success']), (lo + r['best_try']) if found else -1, count], dtype=torch.int64, device=device)
    heads = [torch.empty_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    heads = torch.stack(heads).cpu().numpy()
    winners = [k for k in range(world) if heads[k, 0]]
    if winners:
        win = min(winners, key=lambda k: heads[k, 1])
    else:
        have = [k for k in range(world) if heads[k, 1] >= 0]
        win = min(have, key=lambda k: (-heads[k, 2], heads[k, 1])) if have else -1
    if win < 0:
        return {'success': False, 'essential': None, 'camera': None, 'inlier_percent': 0.0,
                'inlier_idx': np.zeros(0, np.int32), 'best_try': -1, 'best_root': -1}
    model = torch.zeros(9 + 12 + 2, dtype=torch.float64, device=device)
    idx = torch.full((npt,), -1, dtype=torch.int32, device=device)
    if rank == win:
        model[:9] = torch.from_numpy(np.ascontiguousarray(r['essential']).reshape(-1)).to(device)
        model[9:21] = torch.from_numpy(np.ascontiguousarray(r['camera']).reshape(-1)).to(device)
        model[21] = float(r['inlier_percent'])
        model[22] = float(r['best_root'])
        idx[:count] = torch.from_numpy(np.asarray(r['inlier_idx'], dtype=np.int32)).to(device)
    src = win if group is None else dist.get_global_rank(group, win)
    dist.broadcast(model, src=src, group=group)
    dist.broadcast(idx, src=src, group=group)
    model, idx = model.cpu().numpy(), idx.cpu().numpy()
    return {'success': bool(heads[win, 0]), 'essential': model[:9].reshape(3, 3), 'camera': model[9:21].reshape(3, 4),
            'inlier_percent': float(model[21]), 'inlier_idx': idx[:int(heads[win, 2])].copy(),
            'best_try': int(heads[win, 1]), 'best_root': int(model[22])}
