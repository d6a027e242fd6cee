"""Per-kernel timing probe for the normalisation (run under rocprofv3 --kernel-trace --stats)."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spectavi_amd import device as spv  # noqa: E402

rows = 1_000_000
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
table = (torch.rand((rows, 132), dtype=torch.float32, device=dev, generator=g) * 200).floor().contiguous()
if len(sys.argv) > 1 and sys.argv[1] == "sift":
    table[:, 0] = torch.rand(rows, device=dev, generator=g) * 1280
    table[:, 1] = torch.rand(rows, device=dev, generator=g) * 960
    table[:, 2] = torch.rand(rows, device=dev, generator=g) * 7 + 1
    table[:, 3] = (torch.rand(rows, device=dev, generator=g) * 2 - 1) * 3.14159
for _ in range(3):
    out = spv.normalize(table)
torch.cuda.synchronize()
