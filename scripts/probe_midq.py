"""Whole-table top-10 for launches between 16 K and 262 K queries (where the item-split count is chosen)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from generative_recommenders_amd import ops      # noqa: E402

g = torch.Generator().manual_seed(0)
table = torch.randn(4_000_001, 64, generator=g).cuda()
prep = ops.PreparedTable(table)
for nq in (16384, 40000, 65536, 100000, 131072, 140000, 180000, 213568, 262144):
    q = torch.randn(nq, 64, generator=g).cuda()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.score_topk(q, prep, 10)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    print(f"{nq:7d} queries: {ms:7.2f} ms  ({nq / ms / 1e3:.3f} M queries/s)")
