"""Differential sweep of the bf16 top-K path against the exact fp32 kernel over launch shapes around the plan's
switch points (item splits 1 / 2-4 / 5-7 / >= 8, sampled / seeded / cold starts).  Prints one line per shape;
exits non-zero on the first mismatch."""
import itertools
import sys

import torch

sys.path.insert(0, ".")
from generative_recommenders_amd import ops      # noqa: E402

g = torch.Generator().manual_seed(1)
bad = 0
for d, rows in itertools.product((64, 128, 48), (5000, 70_001, 300_007)):
    table = torch.randn(rows, d, generator=g).cuda()
    # a block of near-duplicates so that some queries need the exact re-run
    table[100:140] = table[100] + 1e-4 * torch.randn(40, d, generator=g).cuda()
    prep = ops.PreparedTable(table)
    for n_q, k in itertools.product((4096, 20_000, 40_000, 70_000, 100_000, 131_072, 150_000), (1, 10, 16, 50)):
        q = torch.randn(n_q, d, generator=g).cuda()
        q[:7] = table[100] * 3
        ids, sc, st = ops.score_topk(q, prep, k, return_stats=True)
        want_ids, want_sc = ops.score_topk_exact(q, table, k)
        ok = torch.equal(ids, want_ids) and torch.equal(sc, want_sc)
        print(f"d={d:3d} rows={rows:6d} n_q={n_q:6d} k={k:2d} uncertain={st['uncertain']:5d} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
        if bad:
            sys.exit(1)
print("all equal")
