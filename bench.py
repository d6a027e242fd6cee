#!/usr/bin/env python3
"""bench.py -- headline benchmark of the L1 2-NN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W              (any N: starts its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W (the driver's N > 1 form)
    python bench.py --gpus N --mode inlib                      (one process, ncclCommInitAll clique)

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: every GPU matches its query shard (uint8 [yrows, 128]) against its replica of
the database (uint8 [xrows, 128]) with the hand-written HIP kernels of libspectavi.so (exact
L1 2-NN), then (N > 1) the packed (idx0, idx1, d0, d1) records are gathered on GPU 0 over RCCL.
The loop being sharded is the reference's OpenMP loop over queries, src/BruteForceNnL1K2.h:92-93.

Two forms of the N-GPU job:
  --mode ranks  one process per GPU, torch.distributed (backend nccl = RCCL), spectavi_amd/sharded.py's
                record format.  Started by torch.distributed.run, or -- when WORLD_SIZE is not in
                the environment -- by this script itself: the parent spawns N children with
                RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set BEFORE anything touches a GPU (it
                never imports torch), relays rank 0's JSON line and exits with the worst child rc.
  --mode inlib  north_star's literal form: ONE process, the library's own ncclCommInitAll clique
                and ncclGather (spv_l1k2_gathered_device in include/spectavi_amd.h), no
                torch.distributed.

Workloads (BASELINE.json):
  --gpus 1   1,000,000 x 1,000,000, D=128 -- the shape north_star's target is quoted on
             ("1 M x 1 M SIFT-128 L1 2-NN"); configs[1] (256k x 256k) is a parity-test case
             (tests/test_l1k2_gpu.py::test_full_size_properties_256k) and stays reachable with
             --xrows 262144 --yrows 262144.
  --gpus N>1 database 4,000,000 rows replicated on every GPU, 500,000 query rows per GPU:
             at N = 8 exactly configs[4] (4M x 4M, query set sharded 8 ways, RCCL gather of
             (idx0, idx1, d0, d1)).  Weak scaling: per-GPU work is fixed, the global query set
             grows with N.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline`
and `cpu_baseline` objects, for every N.  The CPU baseline leg is the only place the oracle is
used (its output also checks the GPU result on the sample's first queries).

SPECTAVI_BENCH_REHEARSE=1 (one-GPU box): every rank shares cuda:0, the exchange runs over gloo
on host copies (ranks) / peer copies (inlib); exercises the launcher, sharding, gather and
timing code -- its numbers are not benchmark results (printed with "rehearsal": true).
"""
import argparse
import json
import os
import socket
import subprocess
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("ranks", "inlib"), default="ranks",
                    help="ranks: one process per GPU over torch.distributed/RCCL; inlib: one process, the "
                         "library's own ncclCommInitAll clique + ncclGather")
    ap.add_argument("--xrows", type=int, default=None,
                    help="database rows (replicated); default 1,000,000 at --gpus 1, 4,000,000 otherwise")
    ap.add_argument("--yrows", type=int, default=None,
                    help="query rows PER GPU; default 1,000,000 at --gpus 1, 500,000 otherwise")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target wall time of the CPU baseline sample, split over the thread counts (0 disables it)")
    ap.add_argument("--verify", type=int, default=64,
                    help="queries of the CPU-baseline sample also compared with the GPU result (0 = none)")
    a = ap.parse_args(argv)
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.xrows is None:
        a.xrows = 1_000_000 if a.gpus == 1 else 4_000_000
    if a.yrows is None:
        a.yrows = 1_000_000 if a.gpus == 1 else 500_000
    return a


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment
# ---------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """Environment of child `rank`: what torch.distributed.run would set."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SPECTAVI_BENCH_CHILD": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))  # the CPU-baseline leg sets its own counts
    return env


def launch_ranks(world, argv, script=None, python=None, base_env=None, out=None):
    """Spawns one child per rank (no GPU call, no torch import in this process), relays rank 0's
    stdout, waits for all; a failing child ends the others.  Returns the worst return code."""
    script = script or os.path.abspath(__file__)
    python = python or sys.executable
    out = out or sys.stdout
    port = free_port()
    try:
        err = sys.stderr.fileno()
    except (AttributeError, OSError, ValueError):  # a captured stderr (pytest): inherit the real one
        err = None
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([python, script] + list(argv), env=rank_env(r, world, port, base_env),
                                      stdout=subprocess.PIPE if r == 0 else (err if err is not None else subprocess.DEVNULL),
                                      stderr=err))
    rcs = [None] * world
    line0 = []
    import threading

    def pump():
        for raw in procs[0].stdout:
            line0.append(raw.decode("utf-8", "replace"))
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):  # one rank died: the others would wait in a collective forever
            time.sleep(2.0)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()  # exactly the PIDs started above
            for r, p in enumerate(procs):
                rcs[r] = p.wait()
            break
        time.sleep(0.05)
    t.join(timeout=5.0)
    out.write("".join(line0))
    out.flush()
    return max([0] + [(rc if rc and rc > 0 else 1) for rc in rcs if rc != 0])


# ---------------------------------------------------------------------------------------------
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


def host_cpu():
    """(model name, physical cores, logical cpus) of this box from /proc/cpuinfo."""
    model, cores = "?", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or None), os.cpu_count()


def cpu_baseline(x_host, y_host, target_s, gpu_idx=None, gpu_dist=None, nverify=0):
    """Times the oracle (port of the reference loop nest: SSE2 SAD + early-exit prune + OpenMP over
    queries) on a bounded query sample against the full database at the reference's thread counts:
    1 (Python default, spectavi/feature.py:292), 8 (example default,
    example/ex01_essential_estimation.py:292) and all host threads, a third of the budget each.
    `value` is the best of them.  The same oracle output doubles as the check of the GPU result on
    the sample's first `nverify` queries; this function is the only place bench.py touches oracle/."""
    from oracle import oracle as o
    import numpy as np
    allthreads = o.max_threads()
    model, physical, logical = host_cpu()
    counts = sorted({1, min(8, allthreads), allthreads})
    budget = target_s / len(counts)
    by_threads, legs, verified = {}, [], None
    for threads in counts:
        o.nn_bruteforcel1k2(x_host, y_host[:min(max(threads, 8), y_host.shape[0])], nthreads=threads)  # thread start-up
        # fixed-size chunks of queries until the time target is reached: robust against the rate
        # changing with the sample size
        chunk = max(64, threads * 32)
        nq, dt, oidx, odist = 0, 0.0, None, None
        while nq < y_host.shape[0] and dt < budget:
            hi = min(y_host.shape[0], nq + chunk)
            t0 = time.perf_counter()
            ci, cd = o.nn_bruteforcel1k2(x_host, np.ascontiguousarray(y_host[nq:hi]), nthreads=threads)
            dt += time.perf_counter() - t0
            if oidx is None:
                oidx, odist = ci, cd
            nq = hi
        nv = min(nverify, nq, len(oidx))
        if nv > 0 and gpu_idx is not None:
            ok = bool(np.array_equal(gpu_idx[:nv].view(np.uint64), oidx[:nv]) and np.array_equal(gpu_dist[:nv], odist[:nv]))
            verified = ok if verified is None else (verified and ok)
        rate = float(nq) * x_host.shape[0] / dt
        by_threads[str(threads)] = rate
        legs.append("%d thr: %d queries in %.1f s" % (threads, nq, dt))
    best = max(by_threads, key=lambda k: by_threads[k])
    return {
        "value": by_threads[best], "unit": "pairs/s", "cores": int(best), "kind": "port",
        "by_threads": by_threads, "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
        "omp_max_threads": allthreads,
        "sample": "query samples (chunks of 32 x threads) x %d database rows (D=%d): %s; value = the best thread count; "
                  "the reference itself is not buildable here (Eigen3 absent)" % (x_host.shape[0], x_host.shape[1], "; ".join(legs)),
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


def roofline_of(args, launches, tile_ms, merge_ms):
    groups = args.dim // 4  # v_sad lane-ops per pair (4 bytes each)
    tile_s = tile_ms / 1e3 / max(launches, 1)
    pairs_launch = float(args.xrows) * args.yrows
    kpairs = pairs_launch / tile_s if tile_s > 0 else 0.0
    compulsory = (args.xrows + args.yrows) * args.dim + 24 * args.yrows
    return {
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
        "launches_timed": launches,
        "merge_avg_launch_ms": merge_ms / max(launches, 1),
        "algorithmic": "%d v_sad lane-ops per pair x %.4g pairs per launch" % (groups, pairs_launch),
        "traffic": load_traffic(args.xrows, args.yrows, args.dim),
        "traffic_source": "profiles/l1k2_pmc.json: separate rocprofv3 --pmc passes of this command on this shape (null: shape not profiled)",
        # BASELINE.json's reading: the reference streams one 128-byte database row per pair
        "hbm_streaming_equiv": {"achieved": kpairs * args.dim / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": kpairs * args.dim / 1e9 / HBM_PEAK_GBS},
        "hbm_compulsory": {"bytes": compulsory, "achieved": compulsory / tile_s / 1e9 if tile_s > 0 else 0.0,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s"},
    }


def synthetic(torch, dev, args, rank):
    """The reference's test distribution (uniform uint8, test/test_feature.py:112-115): database
    identical on every rank, query shard seeded by rank."""
    gx = torch.Generator(device=dev).manual_seed(0xdeadbeef)
    x = torch.randint(0, 256, (args.xrows, args.dim), dtype=torch.uint8, device=dev, generator=gx)
    gy = torch.Generator(device=dev).manual_seed(0xdeadbeef + 1 + rank)
    y = torch.randint(0, 256, (args.yrows, args.dim), dtype=torch.uint8, device=dev, generator=gy)
    return x, y


def line_of(args, world, value, elapsed, roofline, cpu, verified, mode, extra):
    par = ("query-shard x%d, database replicated, RCCL gather of 16 B records" % world) + (
        " (one process per GPU, torch.distributed)" if mode == "ranks" else
        " (one process, in-library ncclCommInitAll clique + ncclGather)")
    out = {
        "metric": METRIC, "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload_name(args.xrows, args.yrows, args.dim, world),
                   "xrows": args.xrows, "yrows_per_gpu": args.yrows, "dim": args.dim, "mode": mode,
                   "parallelism": par},
        "roofline": roofline, "cpu_baseline": cpu, "verified_vs_oracle": verified,
    }
    out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------
# --mode ranks: one process per GPU
# ---------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    rehearse = os.environ.get("SPECTAVI_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: only %d GPUs visible (SPECTAVI_BENCH_REHEARSE=1 shares cuda:0)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from spectavi_amd import device as spv
    from spectavi_amd.sharded import pack_records

    x, y = synthetic(torch, dev, args, rank)
    gather_bufs = None
    gdev = torch.device("cpu") if rehearse else dev
    if world > 1 and rank == 0:
        gather_bufs = [torch.empty((args.yrows, 4), dtype=torch.int32, device=gdev) for _ in range(world)]

    def exchange(idx, d):
        rec = pack_records(idx, d)
        dist.gather(rec.cpu() if rehearse else rec, gather_list=gather_bufs, dst=0)

    def step():
        idx, d = spv.l1k2(x, y)
        if world > 1:
            exchange(idx, d)
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
            exchange(idx, d)
        fence()
        gather_s = time.perf_counter() - g0

    t = torch.tensor([elapsed, gather_s], dtype=torch.float64, device=gdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, gather_s = float(t[0].item()), float(t[1].item())

    value = float(args.xrows) * args.yrows * world * args.steps / elapsed
    if rank == 0:
        extra = {}
        if world > 1:
            extra["gather_ms_per_step"] = gather_s / args.steps * 1e3  # pack + RCCL gather alone, max over ranks
            # shard 0 of the gathered records must be rank 0's own result
            i0, d0 = gather_bufs[0][:, 0:2].to(dev), gather_bufs[0][:, 2:4].to(dev)
            extra["gather_consistent"] = bool(torch.equal(i0, idx.to(torch.int32)) and torch.equal(d0, d))
        if rehearse:
            extra["rehearsal"] = True
        cpu = verified = None
        if args.cpu_seconds > 0:  # rank 0's host cores, for every N; the other ranks wait at the barrier below
            nv = min(args.verify, args.yrows)
            cpu, verified = cpu_baseline(x.cpu().numpy(), y.cpu().numpy(), args.cpu_seconds,
                                         idx[:nv].cpu().numpy() if nv else None,
                                         d[:nv].cpu().numpy() if nv else None, nv)
        print(json.dumps(line_of(args, world, value, elapsed, roofline_of(args, launches, tile_ms, merge_ms),
                                 cpu, verified, "ranks", extra)), flush=True)
    if world > 1:
        # not a device-side barrier: the other ranks would spin on their GPUs for the whole CPU leg
        _store_barrier(dist, rank, world)
        dist.destroy_process_group()


def _store_barrier(dist, rank, world):
    """Host-side barrier over the rendezvous store (no GPU work while rank 0 times the CPU leg).
    Rank 0 hosts the store, so it leaves last: the others acknowledge with their final store call."""
    store = dist.distributed_c10d._get_default_store()
    store.add("spectavi_bench_done", 1)
    while int(store.add("spectavi_bench_done", 0)) < world:
        time.sleep(0.2)
    if rank != 0:
        store.add("spectavi_bench_ack", 1)
    else:
        while int(store.add("spectavi_bench_ack", 0)) < world - 1:
            time.sleep(0.05)


# ---------------------------------------------------------------------------------------------
# --mode inlib: one process, the library's own clique
# ---------------------------------------------------------------------------------------------
def run_inlib(args):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    rehearse = os.environ.get("SPECTAVI_BENCH_REHEARSE", "") == "1"
    world = args.gpus
    if not rehearse and world > torch.cuda.device_count():
        raise SystemExit("--gpus %d but only %d GPUs visible (SPECTAVI_BENCH_REHEARSE=1 lists cuda:0 %d times)" %
                         (world, torch.cuda.device_count(), world))
    from spectavi_amd import device as spv

    devs = [torch.device("cuda", 0 if rehearse else r) for r in range(world)]
    transport = "copy" if rehearse else "rccl"  # a clique needs distinct devices
    xs, ys = [], []
    for r, dev in enumerate(devs):
        x, y = synthetic(torch, dev, args, r)
        if rehearse and r > 0:
            x = xs[0]  # one physical GPU: one replica
        xs.append(x)
        ys.append(y)

    def fence():
        for dev in set(devs):
            torch.cuda.synchronize(dev)

    def step():
        return spv.l1k2_gathered(xs, ys, transport=transport)  # synchronous

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
    gl, gather_ms = spv.profile_read("gather")
    wl, widen_ms = spv.profile_read("gather_widen")

    value = float(args.xrows) * args.yrows * world * args.steps / elapsed
    # every shard of the gathered result against that GPU's own un-gathered result
    consistent = True
    for r, dev in enumerate(devs):
        ir, dr = spv.l1k2(xs[r], ys[r])
        lo = r * args.yrows
        consistent = consistent and bool(torch.equal(ir.to(idx.device), idx[lo:lo + args.yrows]) and
                                         torch.equal(dr.to(d.device), d[lo:lo + args.yrows]))
    extra = {"gather_ms_per_step": (gather_ms / max(gl, 1)) + (widen_ms / max(wl, 1)),
             "gather_transport": "ncclGather (ncclCommInitAll clique)" if transport == "rccl" else "hipMemcpyPeerAsync",
             "gather_consistent": consistent}
    if rehearse:
        extra["rehearsal"] = True
    cpu = verified = None
    if args.cpu_seconds > 0:
        nv = min(args.verify, args.yrows)
        cpu, verified = cpu_baseline(xs[0].cpu().numpy(), ys[0].cpu().numpy(), args.cpu_seconds,
                                     idx[:nv].cpu().numpy() if nv else None, d[:nv].cpu().numpy() if nv else None, nv)
    print(json.dumps(line_of(args, world, value, elapsed, roofline_of(args, launches, tile_ms, merge_ms),
                             cpu, verified, "inlib", extra)), flush=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.mode == "inlib":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--mode inlib is one process: do not wrap it in torch.distributed.run")
        return run_inlib(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing in this process has touched (or will touch) a GPU, and torch is not imported
        sys.exit(launch_ranks(args.gpus, argv))
    return run_rank(args)


if __name__ == "__main__":
    main()
