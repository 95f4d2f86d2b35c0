"""Scatter-add probe: n random ids over a rows x d table (defaults: 16.8 M ids, 4 M x 64); a sixth argument "zipf"
draws the ids Zipf(1.05) over the item ranks (ranks spread over the table by a multiplicative hash)."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_recommenders_amd import hmx, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_001
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
ids = torch.randint(1, rows, (n,), device=dev, generator=g)
if len(sys.argv) > 5 and sys.argv[5] == "zipf":
    u = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
    ranks = ((float(rows - 1) ** -0.05 - 1.0) * u + 1.0) ** (1.0 / -0.05)
    ids = ranks.long().clamp_(1, rows - 1) * 2654435761 % (rows - 1) + 1
    del u, ranks
grad = torch.randn(n, d, device=dev, generator=g)
out = torch.zeros(rows, d, device=dev)
ops.scatter_add_rows(grad, ids, rows, 0, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ops.scatter_add_rows(grad, ids, rows, 0, out=out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
u = int(torch.unique(ids).numel())
alg = n * (8 + 4 * d) + 2 * u * 4 * d
print(f"scatter_add n={n} rows={rows} d={d}: {ms:.3f} ms per call, unique {u}, algorithmic {alg / 1e9:.3f} GB -> {alg / ms / 1e9:.3f} TB/s "
      f"({alg / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s)")
