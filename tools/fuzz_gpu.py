"""Randomised differential run on the GPU box: the C-ABI host paths against the CPU oracle on
many random shapes / parameters per path, for a time budget.  Prints one line per path and
exits non-zero on the first mismatch (with the failing configuration).

    python tools/fuzz_gpu.py --seconds 60 --seed 1
    python tools/fuzz_gpu.py --only cascade --case 17 --seed 1     # re-run one reported case
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from oracle import oracle as o  # noqa: E402
from spectavi_amd import feature, mvg  # noqa: E402


def fuzz_l1k2(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 0, n])  # every case is reproducible on its own
        dim = int(rng.choice([16, 32, 48, 64, 96, 128, 144, 192, 256, 272, 512, 1024]))
        xrows = int(rng.choice([0, 1, 2, 3, rng.integers(4, 300), rng.integers(300, 6000), rng.integers(60000, 70000)]))
        yrows = int(rng.choice([1, 2, rng.integers(3, 200), rng.integers(200, 3000)]))
        if dim > 256:
            xrows, yrows = min(xrows, 3000), min(yrows, 500)
        hi = int(rng.choice([2, 3, 16, 256]))        # small alphabets force distance ties
        x = rng.integers(0, hi, (xrows, dim), dtype=np.uint8)
        y = rng.integers(0, hi, (yrows, dim), dtype=np.uint8)
        if xrows and rng.random() < 0.3:               # exact duplicates of database rows
            k = min(yrows, 8)
            y[:k] = x[rng.integers(0, xrows, k)]
        idx, dist = feature.nn_bruteforcel1k2(x, y)
        oi, od = o.nn_bruteforcel1k2(x, y, nthreads=o.max_threads())
        if not (np.array_equal(idx, oi) and np.array_equal(dist, od)):
            raise SystemExit("L1K2 MISMATCH case=%d xrows=%d yrows=%d dim=%d hi=%d" % (n, xrows, yrows, dim, hi))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_cascade(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 1, n])  # every case is reproducible on its own
        dim = int(rng.choice([16, 32, 64, 128, 144, 256]))
        m = int(rng.choice([1, 2, 3, 4, 6, 8, 11, 13, 17, 20, 22, 23, 27, 31]))
        nt = int(rng.choice([1, 2, 3, 4, 7, 16]))
        g = int(rng.integers(0, min(m, 6) + 1))
        xrows = int(rng.choice([0, 1, 2, rng.integers(3, 200), rng.integers(200, 5000)]))
        yrows = int(rng.choice([1, 2, rng.integers(3, 100), rng.integers(100, 1500)]))
        span = int(rng.choice([2, 20, 128]))           # narrow ranges: many zero projections / ties
        x = rng.integers(-span, span, (xrows, dim)).astype(np.float32)
        y = rng.integers(-span, span, (yrows, dim)).astype(np.float32)
        if xrows and rng.random() < 0.5:
            k = min(yrows, max(1, xrows // 2))
            y[:k] = np.clip(x[rng.integers(0, xrows, k)] + rng.integers(-2, 3, (k, dim)), -128, 127)
        d = rng.standard_normal((nt, dim, m)).astype(np.float32)
        if rng.random() < 0.2:
            d = np.round(d)                            # integer hyperplanes: exact-zero projections
        idx, dist, ncand = feature.nn_cascading_hash_with_dict(x, y, d, g=g, return_ncand=True)
        oi, od, onc, _ = o.nn_cascading_hash(x, y, m, nt, g, d)
        if not (np.array_equal(ncand, onc) and np.array_equal(dist, od) and np.array_equal(idx, oi)):
            bad = np.flatnonzero((ncand != onc) | (dist != od).any(1) | (idx != oi).any(1))
            raise SystemExit("CASCADE MISMATCH case=%d xrows=%d yrows=%d dim=%d m=%d n=%d g=%d span=%d; %d queries differ, first %s: "
                             "gpu idx %s dist %s ncand %d, oracle idx %s dist %s ncand %d" %
                             (n, xrows, yrows, dim, m, nt, g, span, len(bad), bad[:5], idx[bad[0]], dist[bad[0]], ncand[bad[0]],
                              oi[bad[0]], od[bad[0]], onc[bad[0]]))
        n += 1
        if only_case is not None:
            break
    return n


def fuzz_dlt(seed, budget, only_case=None):
    t0, n = time.time(), 0 if only_case is None else only_case
    while time.time() - t0 < budget:
        rng = np.random.default_rng([seed, 2, n])  # every case is reproducible on its own
        npt = int(rng.choice([1, 2, 63, 64, 65, 255, 257, rng.integers(300, 20000)]))
        kind = int(rng.integers(0, 6))
        P0 = rng.standard_normal((3, 4))
        P1 = rng.standard_normal((3, 4))
        if kind == 1:
            P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
        if kind == 2:
            P1 = P0 + 1e-9 * rng.standard_normal((3, 4))       # nearly identical cameras
        Xw = rng.standard_normal((npt, 4))
        if kind == 3:
            Xw[:, 3] = 0.0                                       # points at infinity
        x, xp = Xw @ P0.T, Xw @ P1.T
        noise = float(rng.choice([0.0, 1e-6, 1e-3, 0.1, 10.0]))
        x[:, :2] += noise * rng.standard_normal((npt, 2))
        xp[:, :2] += noise * rng.standard_normal((npt, 2))
        if kind == 4:
            x *= 1e6                                             # large homogeneous scale
        if kind == 5:
            x[: max(1, npt // 10)] = xp[: max(1, npt // 10)]    # inconsistent pairs
        X = mvg.dlt_triangulate(P0, P1, x, xp)
        E = mvg.dlt_reprojection_error(P0, P1, x, xp)
        oX, oE = o.dlt_triangulate(P0, P1, x, xp), o.dlt_reprojection_error(P0, P1, x, xp)
        if not (np.array_equal(X, oX, equal_nan=True) and np.array_equal(E, oE, equal_nan=True)):
            bad = int(np.sum(~((X == oX) | (np.isnan(X) & np.isnan(oX))).all(axis=1)))
            raise SystemExit("DLT MISMATCH case=%d npt=%d kind=%d noise=%g rows_differing=%d maxabs=%g" %
                             (n, npt, kind, noise, bad, float(np.nanmax(np.abs(X - oX)))))
        n += 1
        if only_case is not None:
            break
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0, help="budget per path")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="comma list of l1k2,cascade,dlt")
    ap.add_argument("--case", type=int, default=None, help="re-run one case number of the --only path")
    a = ap.parse_args()
    want = set(filter(None, a.only.split(",")))
    for name, fn in (("l1k2", fuzz_l1k2), ("cascade", fuzz_cascade), ("dlt", fuzz_dlt)):
        if want and name not in want:
            continue
        cases = fn(a.seed, a.seconds, a.case)
        print("%s: %d random cases bit-identical to the oracle" % (name, cases), flush=True)
