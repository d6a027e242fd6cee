"""Sweeps the experiment knobs of the L1 tile kernel on one GPU (each setting in its own
process because the library reads the knobs once).  Usage: python tools/l1k2_sweep.py"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import torch
from spectavi_amd import device as spv
n = int(os.environ.get("SWEEP_ROWS", "262144"))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
y = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
spv.l1k2(x, y); torch.cuda.synchronize()
spv.profile_enable(True)
for _ in range(3): spv.l1k2(x, y)
torch.cuda.synchronize()
k, ms = spv.profile_read("l1k2_tile")
print(json.dumps({"ms": ms / k, "pairs_per_s": float(n) * n / (ms / k * 1e-3)}))
''' % ROOT


def main():
    rows = []
    feeds = ["lds"]  # the scalar-feed variant left the library (tools/exp/l1k2_sfeed_exp.hip)
    qs = [int(v) for v in os.environ.get("SWEEP_QS", "1,2,4").split(",")]
    blks = [int(v) for v in os.environ.get("SWEEP_BLOCKS", "1024,2048,4096").split(",")]
    for feed, q, blocks in itertools.product(feeds, qs, blks):
        env = dict(os.environ, SPECTAVI_L1K2_Q=str(q), SPECTAVI_L1K2_BLOCKS=str(blocks))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        try:
            rec = json.loads(line)
        except Exception:
            rec = {"error": line}
        rec.update(feed=feed, q=q, blocks=blocks)
        rows.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
