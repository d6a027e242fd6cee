"""Query-sharded L1 2-NN across the GPUs of one node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI), results collected with
one gather of packed (idx0, idx1, d0, d1) records.

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


def nn_bruteforcel1k2_sharded(x, y_shard, total_queries, group=None, dst=0, local_fn=None):
    """Run the local shard and gather all shards' results on rank `dst`.

    x            database, replicated on every rank (uint8 [M,D])
    y_shard      this rank's query rows, shard_bounds(total_queries, world, rank)
    local_fn     (x, y) -> (idx int64 [n,2], dist int32 [n,2]); defaults to the HIP
                 kernel (spectavi_amd.device.l1k2).  Tests inject a CPU checker.
    Returns (idx, dist) for all `total_queries` rows on rank dst, (None, None) elsewhere.
    """
    local_fn = local_fn or _default_local_fn
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total_queries, world, rank)
    assert y_shard.shape[0] == hi - lo, "y_shard does not match shard_bounds"
    idx, d = local_fn(x, y_shard)
    rec = pack_records(idx, d)
    # ragged shards: pad to the largest shard so every rank sends the same count
    max_rows = shard_bounds(total_queries, world, 0)[1]
    if rec.shape[0] < max_rows:
        pad = torch.zeros((max_rows - rec.shape[0], 4), dtype=rec.dtype, device=rec.device)
        rec = torch.cat([rec, pad], dim=0)
    if rank == dst:
        bufs = [torch.empty_like(rec) for _ in range(world)]
        dist.gather(rec, gather_list=bufs, dst=dst, group=group)
        parts = []
        for r in range(world):
            rlo, rhi = shard_bounds(total_queries, world, r)
            parts.append(bufs[r][: rhi - rlo])
        return unpack_records(torch.cat(parts, dim=0))
    dist.gather(rec, gather_list=None, dst=dst, group=group)
    return None, None
