"""Measures the non-headline rows of the hot path on one GPU (BASELINE configs[2], [3]) and
prints one JSON line each: cascade 1M x 1M (m=17, n=2, g=2) and 10M-point DLT.
Inputs are resident in HBM; kernel times come from the library's hipEvent brackets."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from spectavi_amd import device as spv  # noqa: E402
from spectavi_amd import feature  # noqa: E402

HBM_PEAK = 8000.0


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    spv.profile_reset()
    spv.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    spv.profile_enable(False)
    return out, dt


def cascade(rows, steps, warmup, planted):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0xdeadbeef)
    x = torch.randint(0, 256, (rows, 128), dtype=torch.uint8, device=dev, generator=g).float() - 128
    if planted:
        perm = torch.randperm(rows, device=dev, generator=g)
        noise = torch.randint(-3, 4, (rows, 128), device=dev, generator=g).float()
        y = torch.clamp(x[perm] + noise, -128, 127)
    else:
        y = torch.randint(0, 256, (rows, 128), dtype=torch.uint8, device=dev, generator=g).float() - 128
    m = feature.auto_hash_bit_rate(rows, rows)
    n, gg = 2, 2
    d = torch.from_numpy(feature.generate_hash_dict(0x5eed, 128, m, n)).to(dev)
    (idx, dist, ncand), dt = timed(lambda: spv.cascade(x, y, d, g=gg, want_ncand=True), steps, warmup)
    k = {name: spv.profile_read(name) for name in ("cascade_project", "cascade_buckets", "cascade_probe_refine")}
    cbar = float(ncand.float().mean())
    probe_s = k["cascade_probe_refine"][1] / max(k["cascade_probe_refine"][0], 1) / 1e3
    proj_s = k["cascade_project"][1] / max(k["cascade_project"][0], 1) / 1e3
    probe_bytes = rows * (128 + 128 * cbar + 24 + n * (1 << gg) * 8 + cbar * 4)
    setup_bytes = 2 * rows * (512 + 128 + 4 * n)
    rec = {
        "metric": "cascade-hash L1 2-NN, queries/s (1M x 1M, m=%d n=%d g=%d)" % (m, n, gg),
        "value": rows / dt, "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f32 hash + u8 refine", "data": "synthetic",
        "config": {"workload": "cascade %d x %d D=128, %s queries" % (rows, rows, "planted" if planted else "uniform"),
                   "m": m, "n": n, "g": gg},
        "mean_candidates": cbar, "equiv_bruteforce_pairs_per_s": float(rows) * rows / dt,
        "evaluated_pairs_per_s": rows * cbar / dt,
        "found_fraction": float((idx[:, 0] >= 0).float().mean()),
        "kernels_ms": {kk: v[1] / max(v[0], 1) for kk, v in k.items()},
        "roofline": {"bound": "hbm", "kernel": "probe_refine_kernel", "achieved": probe_bytes / probe_s / 1e9,
                     "peak": HBM_PEAK, "unit": "GB/s", "frac": probe_bytes / probe_s / 1e9 / HBM_PEAK,
                     "algorithmic": "%.0f B per query (128 + 128*C + 24 + bucket look-ups + 4*C index reads), C=%.1f" % (
                         probe_bytes / rows, cbar), "traffic": None},
        "roofline_project": {"bound": "hbm", "kernel": "project_kernel", "achieved": setup_bytes / proj_s / 1e9,
                             "peak": HBM_PEAK, "unit": "GB/s", "frac": setup_bytes / proj_s / 1e9 / HBM_PEAK},
    }
    if planted:
        rec["recall_planted_top1"] = float((idx[:, 0] == perm).float().mean())
    print(json.dumps(rec), flush=True)


def dlt(npt, steps, warmup):
    dev = torch.device("cuda")
    rng = np.random.default_rng(1)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    P1 = np.hstack([R, rng.standard_normal((3, 1))])
    g = torch.Generator(device=dev).manual_seed(7)
    Xw = torch.randn((npt, 4), dtype=torch.float64, device=dev, generator=g)
    Xw[:, 2] += 5.0
    Xw[:, 3] = 1.0
    x = Xw @ torch.from_numpy(P0).to(dev).T
    xp = Xw @ torch.from_numpy(P1).to(dev).T
    x[:, :2] += 1e-3 * torch.randn((npt, 2), dtype=torch.float64, device=dev, generator=g) * x[:, 2:3]
    xp[:, :2] += 1e-3 * torch.randn((npt, 2), dtype=torch.float64, device=dev, generator=g) * xp[:, 2:3]
    for name, fn, nbytes in (("dlt_triangulate", spv.dlt_triangulate, 80), ("dlt_reprojection_error", spv.dlt_reprojection_error, 56)):
        _, dt = timed(lambda: fn(P0, P1, x, xp), steps, warmup)
        n, ms = spv.profile_read("dlt")
        ks = ms / max(n, 1) / 1e3
        print(json.dumps({
            "metric": "%s, points/s (10M point pairs)" % name, "value": npt / dt, "unit": "points/s", "n_gpus": 1,
            "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f64",
            "data": "synthetic", "config": {"workload": "%s %d points, noisy observations" % (name, npt)},
            "roofline": {"bound": "hbm", "kernel": "dlt_kernel", "achieved": nbytes * npt / ks / 1e9, "peak": HBM_PEAK,
                         "unit": "GB/s", "frac": nbytes * npt / ks / 1e9 / HBM_PEAK, "avg_launch_ms": ks * 1e3,
                         "algorithmic": "%d B per point" % nbytes, "traffic": None},
        }), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--npt", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    cascade(a.rows, a.steps, a.warmup, planted=False)
    cascade(a.rows, a.steps, a.warmup, planted=True)
    dlt(a.npt, a.steps, a.warmup)
