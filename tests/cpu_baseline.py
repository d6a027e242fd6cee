"""CPU baseline table: the oracle (port of the reference loop nest) at
nthreads = 1, 8 and all host threads on a bounded query sample against the full database,
plus the serial DLT loop.  Run on the GPU box's host; prints JSON lines."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as o  # noqa: E402


def l1k2(xrows, dim, threads, target_s=6.0):
    rng = np.random.default_rng(0xdeadbeef)
    x = rng.integers(0, 256, (xrows, dim), dtype=np.uint8)
    y = rng.integers(0, 256, (4096 * 8, dim), dtype=np.uint8)
    probe = 64 * threads
    t0 = time.perf_counter()
    o.nn_bruteforcel1k2(x, y[:probe], nthreads=threads)
    dt = time.perf_counter() - t0
    nq = int(min(len(y), max(probe, probe * target_s / dt)))
    t0 = time.perf_counter()
    o.nn_bruteforcel1k2(x, np.ascontiguousarray(y[:nq]), nthreads=threads)
    dt = time.perf_counter() - t0
    return {"path": "l1k2", "xrows": xrows, "queries_sampled": nq, "threads": threads, "seconds": dt,
            "pairs_per_s": nq * xrows / dt}


def cascade(rows=1_000_000, queries=1_000_000):
    """the cascade restatement (oracle_cascade.cpp: hashing + bucket build over the
    full database, then probe + L1 refine) at BASELINE configs[2] in full -- 1M x 1M, m=17, n=2, g=2 (the
    Python front-end's defaults at this size) -- at the reference's hard-wired 8 threads
    (src/CascadingHashNn.h:230) and at all host threads; the setup (one query) is reported beside it."""
    rng = np.random.default_rng(0xdeadbeef)
    x = (rng.integers(0, 256, (rows, 128), dtype=np.uint8).astype(np.float32) - 128)
    y = (rng.integers(0, 256, (queries, 128), dtype=np.uint8).astype(np.float32) - 128)
    d = np.random.default_rng(0x5eed).standard_normal((2, 128, 17)).astype(np.float32)
    out = []
    for threads in sorted({8, o.max_threads()}):
        o.set_threads(threads)
        o.nn_cascading_hash(x, y[:1], 17, 2, 2, d)           # warm-up: threads, page faults
        t0 = time.perf_counter()
        o.nn_cascading_hash(x, y[:1], 17, 2, 2, d)           # one query: the setup (codes + buckets)
        setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, _, ncand, _ = o.nn_cascading_hash(x, y, 17, 2, 2, d)
        full = time.perf_counter() - t0
        out.append({"path": "nn_cascading_hash (oracle_cascade.cpp: codes + buckets + probe + L1 refine)", "xrows": rows,
                    "queries": queries, "threads": threads, "setup_s": setup, "seconds": full,
                    "mean_candidates": float(ncand.mean()), "queries_per_s": queries / full})
    o.set_threads(o.max_threads())
    return out


def dlt(npt=1_000_000):
    rng = np.random.default_rng(1)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    Xw = rng.standard_normal((npt, 4))
    x, xp = Xw @ P0.T, Xw @ P1.T
    t0 = time.perf_counter()
    o.dlt_triangulate(P0, P1, x, xp)
    dt = time.perf_counter() - t0
    return {"path": "dlt_triangulate (serial loop, as reference src/Spectavi.cpp:48-51; per point the "
                    "two-sided JacobiSVD of oracle_jacobisvd.cpp, the reference's own algorithm restated)",
            "points": npt, "threads": 1, "seconds": dt, "points_per_s": npt / dt}


def ransac():
    """The oracle's serial RANSAC loop (reference fit_essential with nthread = 1) on the scenes of
    tools/bench_paths.py --only fit: tries/s with no try succeeding (a try costs microseconds when the
    gate rejects its candidates, 4 x npt SVDs per candidate when not)."""
    from tests import mvg_checks as mc
    rng = np.random.default_rng(21)
    for npt in (500, 2000, 20000):
        x0, x1, E, out_idx = mc.two_view_scene(rng, npt=npt, outlier_fraction=0.4, noise=1e-4)
        tries = 2000 if npt <= 2000 else 400
        srng = np.random.default_rng(2)
        samples = np.stack([srng.choice(np.arange(1, npt), 7, replace=False) for _ in range(tries)]).astype(np.int32)
        t0 = time.perf_counter()
        o.ransac_fit(x0, x1, samples, required_percent_inliers=0.999, reprojection_error_allowed=1e-3,
                     singular_value_ratio_allowed=3e-2)
        dt = time.perf_counter() - t0
        print(json.dumps({"path": "ransac_fit (serial loop of oracle_ransac.cpp: seven-point + process_fundamental_matrix "
                                  "per root, reference src/RansacFitter.h:152-272 with nthread = 1)",
                          "correspondences": npt, "outliers": 0.4, "tries": tries, "threads": 1, "seconds": dt,
                          "tries_per_s": tries / dt}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ransac":
        ransac()
        sys.exit(0)
    allc = o.max_threads()
    cpu = [l for l in open("/proc/cpuinfo") if l.startswith("model name")]
    print(json.dumps({"cpu": cpu[0].split(":", 1)[1].strip() if cpu else "?", "logical_cpus": os.cpu_count(),
                      "omp_max_threads": allc}), flush=True)
    for xrows in (1000, 262144, 1_000_000):
        for th in sorted({1, 8, allc}):
            print(json.dumps(l1k2(xrows, 128, th)), flush=True)
    for rec in cascade():
        print(json.dumps(rec), flush=True)
    print(json.dumps(dlt()), flush=True)
