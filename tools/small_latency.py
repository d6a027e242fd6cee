import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from spectavi_amd import feature, mvg
rng = np.random.default_rng(0)
for n in (1000, 10000, 50000):
    x = rng.integers(0, 256, (n, 128), dtype=np.uint8); y = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    xf, yf = x.astype(np.float32) - 128, y.astype(np.float32) - 128
    m = max(4, feature.auto_hash_bit_rate(n, n)); d = feature.generate_hash_dict(1, 128, m, 2)
    P0, P1 = rng.standard_normal((3, 4)), rng.standard_normal((3, 4)); Xw = rng.standard_normal((n, 4))
    px, pxp = np.ascontiguousarray(Xw @ P0.T), np.ascontiguousarray(Xw @ P1.T)
    for name, fn in (("l1k2", lambda: feature.nn_bruteforcel1k2(x, y)), ("cascade", lambda: feature.nn_cascading_hash_with_dict(xf, yf, d, g=2)),
                     ("dlt_triangulate", lambda: mvg.dlt_triangulate(P0, P1, px, pxp)), ("dlt_error", lambda: mvg.dlt_reprojection_error(P0, P1, px, pxp))):
        for _ in range(3): fn()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        print("%-16s n=%6d  %.3f ms per host call" % (name, n, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
