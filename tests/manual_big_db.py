"""One-off: L1 2-NN against a 20M-row database (2.56 GB, row offsets beyond 2^31 bytes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spectavi_amd import device as spv
from oracle import oracle as o
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
xrows, yrows = 20_000_000, 4096
x = torch.randint(0, 256, (xrows, 128), dtype=torch.uint8, device=dev, generator=g)
y = torch.randint(0, 256, (yrows, 128), dtype=torch.uint8, device=dev, generator=g)
y[:100] = x[torch.arange(100, device=dev) * 199_999 + 17]          # planted exact copies, incl. rows > 2^24
t0 = time.time(); idx, d = spv.l1k2(x, y); torch.cuda.synchronize(); print("gpu s", time.time() - t0, flush=True)
assert torch.equal(idx[:100, 0].cpu(), torch.arange(100) * 199_999 + 17) and int(d[:100, 0].abs().max()) == 0
xh = x.cpu().numpy(); yh = y[:48].cpu().numpy()
t0 = time.time(); oi, od = o.nn_bruteforcel1k2(xh, yh, nthreads=o.max_threads()); print("oracle s", time.time() - t0, flush=True)
assert np.array_equal(idx[:48].cpu().numpy().view(np.uint64), oi) and np.array_equal(d[:48].cpu().numpy(), od)
assert bool((d[:, 0] <= d[:, 1]).all())
print("20M-row database: planted rows found, 48 queries identical to the oracle", flush=True)
