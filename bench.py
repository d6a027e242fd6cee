#!/usr/bin/env python3
"""bench.py -- headline benchmark of the L1 2-NN hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: every rank matches its query shard (uint8 [yrows, 128]) against the
replicated database (uint8 [xrows, 128]) with the hand-written HIP kernels of
libspectavi.so (exact L1 2-NN), then (N > 1) the packed (idx0, idx1, d0, d1) records are
gathered on rank 0 over RCCL.

Workloads (BASELINE.json):
  --gpus 1   1,000,000 x 1,000,000, D=128 -- the shape north_star's target is quoted on
             ("1 M x 1 M SIFT-128 L1 2-NN"); configs[1] (256k x 256k) is a parity-test case
             (tests/test_l1k2_gpu.py::test_full_size_properties_256k) and stays reachable with
             --xrows 262144 --yrows 262144.
  --gpus N>1 database 4,000,000 rows replicated on every GPU, 500,000 query rows per rank:
             at N = 8 exactly configs[4] (4M x 4M, query set sharded 8 ways, RCCL gather of
             (idx0, idx1, d0, d1)).  Weak scaling: per-GPU work is fixed, the global query set
             grows with N.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline`
and `cpu_baseline` objects.  The CPU baseline leg is the only place the oracle is used (its
output also checks the GPU result on the sample's first queries).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "descriptor-pairs/sec, L1 2-NN, 128-D SIFT; achieved HBM GB/s vs roofline"

# MI355X constants (/opt/skills/guides/MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, 2.4 GHz max clock,
# HBM3E 8 TB/s.  v_sad_hi_u8 (like every 3-source VOP3 integer op measured: v_sad_u8/u16,
# v_med3_u32, v_dot4_u32_u8) issues one wave64 instruction per 4 cycles per SIMD on gfx950
# (tools/microbench.py: 4.13-4.19 cycles at saturation, profiles/r01_microbench.txt), i.e.
# 16 lanes/clk/SIMD = 64 lanes/clk/CU.  Peak = 256 x 64 x 2.4e9 = 3.93e13 SAD lane-ops/s.
CUS, SAD_LANES_PER_CU_CLK, CLK_HZ = 256, 64, 2.4e9
VALU_LANE_OPS_PEAK = CUS * SAD_LANES_PER_CU_CLK * CLK_HZ
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--xrows", type=int, default=None,
                    help="database rows (replicated); default 1,000,000 at --gpus 1, 4,000,000 otherwise")
    ap.add_argument("--yrows", type=int, default=None,
                    help="query rows PER GPU; default 1,000,000 at --gpus 1, 500,000 otherwise")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target wall time of the CPU baseline sample (0 disables it)")
    ap.add_argument("--verify", type=int, default=64,
                    help="queries of the CPU-baseline sample also compared with the GPU result (0 = none)")
    a = ap.parse_args()
    if a.xrows is None:
        a.xrows = 1_000_000 if a.gpus == 1 else 4_000_000
    if a.yrows is None:
        a.yrows = 1_000_000 if a.gpus == 1 else 500_000
    return a


def workload_name(xrows, yrows, dim, world):
    if (xrows, yrows, dim, world) == (1_000_000, 1_000_000, 128, 1):
        tag = "north_star target shape 1M x 1M"
    elif (xrows, yrows, dim) == (262144, 262144, 128):
        tag = "BASELINE configs[1]"
    elif (xrows, yrows, dim) == (4_000_000, 500_000, 128):
        tag = ("BASELINE configs[4]: 4M x 4M over 8 GPUs" if world == 8 else
               "one-eighth shards of BASELINE configs[4] (4M-row database, 500k queries per GPU), %d of 8 shards" % world)
    else:
        tag = "custom shape"
    return "L1 2-NN all-pairs, %d database rows x %d query rows per GPU, D=%d uint8 (%s)" % (xrows, yrows, dim, tag)


def cpu_baseline(x_host, y_host, target_s, gpu_idx=None, gpu_dist=None, nverify=0):
    """Times the oracle (port of the reference loop nest: SSE2 SAD + early-exit prune +
    OpenMP over queries) on a bounded query sample against the full database.  The same oracle
    output doubles as the check of the GPU result on the sample's first `nverify` queries; this
    function is the only place bench.py touches oracle/."""
    from oracle import oracle as o
    import numpy as np
    threads = o.max_threads()
    o.nn_bruteforcel1k2(x_host, y_host[:min(256, y_host.shape[0])], nthreads=threads)  # thread start-up
    # fixed-size chunks of queries until the time target is reached: robust against the rate
    # changing with the sample size
    chunk = max(4096, threads * 32)
    nq, dt, oidx, odist = 0, 0.0, None, None
    while nq < y_host.shape[0] and dt < target_s:
        hi = min(y_host.shape[0], nq + chunk)
        t0 = time.perf_counter()
        ci, cd = o.nn_bruteforcel1k2(x_host, np.ascontiguousarray(y_host[nq:hi]), nthreads=threads)
        dt += time.perf_counter() - t0
        if oidx is None:
            oidx, odist = ci, cd
        nq = hi
    verified = None
    nv = min(nverify, nq, len(oidx))
    if nv > 0 and gpu_idx is not None:
        verified = bool(np.array_equal(gpu_idx[:nv].view(np.uint64), oidx[:nv]) and np.array_equal(gpu_dist[:nv], odist[:nv]))
    pairs = float(nq) * x_host.shape[0]
    return {
        "value": pairs / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": "%d queries (in chunks of %d) x %d database rows (D=%d), %.1f s, OpenMP threads=%d; the reference "
                  "itself is not buildable here (Eigen3 absent)" % (nq, chunk, x_host.shape[0], x_host.shape[1], dt, threads),
    }, verified


def load_traffic(xrows, yrows, dim):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass
    (profiles/l1k2_pmc.json, written by tools/pmc_summary.py), if it matches this workload."""
    path = os.path.join(ROOT, "profiles", "l1k2_pmc.json")
    try:
        doc = json.load(open(path))
        for rec in doc.get("records", [doc]):
            if (rec.get("xrows"), rec.get("yrows"), rec.get("dim")) == (xrows, yrows, dim):
                return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Rehearsal on a 1-GPU box (SPECTAVI_BENCH_REHEARSE=1): every rank shares cuda:0 and the
    # gather runs over gloo on host copies.  It exercises the sharding / gather / timing code;
    # its numbers are not benchmark results (printed with "rehearsal": true).
    rehearse = os.environ.get("SPECTAVI_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from spectavi_amd import device as spv
    from spectavi_amd.sharded import pack_records

    # synthetic descriptors, the reference's test distribution (uniform uint8); database
    # identical on every rank, query shard seeded by rank
    gx = torch.Generator(device=dev).manual_seed(0xdeadbeef)
    x = torch.randint(0, 256, (args.xrows, args.dim), dtype=torch.uint8, device=dev, generator=gx)
    gy = torch.Generator(device=dev).manual_seed(0xdeadbeef + 1 + rank)
    y = torch.randint(0, 256, (args.yrows, args.dim), dtype=torch.uint8, device=dev, generator=gy)
    gather_bufs = None
    gdev = torch.device("cpu") if rehearse else dev
    if world > 1 and rank == 0:
        gather_bufs = [torch.empty((args.yrows, 4), dtype=torch.int32, device=gdev) for _ in range(world)]

    def step():
        idx, d = spv.l1k2(x, y)
        if world > 1:
            rec = pack_records(idx, d)
            dist.gather(rec.cpu() if rehearse else rec, gather_list=gather_bufs, dst=0)
        return idx, d

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    spv.profile_reset()
    spv.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx, d = step()
    fence()
    elapsed = time.perf_counter() - t0
    spv.profile_enable(False)
    launches, tile_ms = spv.profile_read("l1k2_tile")
    _, merge_ms = spv.profile_read("l1k2_merge")

    # the exchange step on its own ("report its time as a separate line"): K packs +
    # gathers of the last result, outside the timed region
    gather_s = 0.0
    if world > 1:
        fence()
        g0 = time.perf_counter()
        for _ in range(args.steps):
            rec = pack_records(idx, d)
            dist.gather(rec.cpu() if rehearse else rec, gather_list=gather_bufs, dst=0)
        fence()
        gather_s = time.perf_counter() - g0

    t = torch.tensor([elapsed, gather_s], dtype=torch.float64, device=gdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, gather_s = float(t[0].item()), float(t[1].item())

    pairs_per_step = float(args.xrows) * args.yrows * world
    value = pairs_per_step * args.steps / elapsed

    if rank == 0:
        groups = args.dim // 4  # v_sad lane-ops per pair (4 bytes each)
        tile_s = tile_ms / 1e3 / max(launches, 1)
        pairs_launch = float(args.xrows) * args.yrows
        kpairs = pairs_launch / tile_s if tile_s > 0 else 0.0
        compulsory = (args.xrows + args.yrows) * args.dim + 24 * args.yrows
        roofline = {
            # Sum-of-absolute-differences is not a contraction, so neither MFMA nor HBM bounds
            # this kernel: the binding unit is the integer VALU's v_sad issue rate (4 bytes/lane-op,
            # one wave64 instruction per 4 cycles per SIMD at the nominal 2.4 GHz).
            "bound": "valu",
            "kernel": "l1k2_tile_kernel",
            "achieved": kpairs * groups / 1e12,
            "peak": VALU_LANE_OPS_PEAK / 1e12,
            "unit": "Tlane-op/s",
            "frac": kpairs * groups / VALU_LANE_OPS_PEAK,
            "avg_launch_ms": tile_s * 1e3,
            "merge_avg_launch_ms": merge_ms / max(launches, 1),
            "algorithmic": "%d v_sad lane-ops per pair x %.4g pairs per launch" % (groups, pairs_launch),
            "traffic": load_traffic(args.xrows, args.yrows, args.dim),
            # BASELINE.json's reading: the reference streams one 128-byte database row per pair
            "hbm_streaming_equiv": {"achieved": kpairs * args.dim / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": kpairs * args.dim / 1e9 / HBM_PEAK_GBS},
            "hbm_compulsory": {"bytes": compulsory, "achieved": compulsory / tile_s / 1e9 if tile_s > 0 else 0.0,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s"},
        }
        verified = None
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            nv = min(args.verify, args.yrows)
            cpu, verified = cpu_baseline(x.cpu().numpy(), y.cpu().numpy(), args.cpu_seconds,
                                         idx[:nv].cpu().numpy() if nv else None,
                                         d[:nv].cpu().numpy() if nv else None, nv)
        out = {
            "metric": METRIC, "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_name(args.xrows, args.yrows, args.dim, world),
                       "xrows": args.xrows, "yrows_per_gpu": args.yrows, "dim": args.dim,
                       "parallelism": "query-shard x%d, database replicated, RCCL gather of 16 B records" % world},
            "roofline": roofline, "cpu_baseline": cpu, "verified_vs_oracle": verified,
        }
        if world > 1:
            out["gather_ms_per_step"] = gather_s / args.steps * 1e3  # pack + RCCL gather alone, max over ranks
            # shard 0 of the gathered records must be rank 0's own result
            i0, d0 = gather_bufs[0][:, 0:2].to(dev), gather_bufs[0][:, 2:4].to(dev)
            out["gather_consistent"] = bool(torch.equal(i0, idx.to(torch.int32)) and torch.equal(d0, d))
        if rehearse:
            out["rehearsal"] = True
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
