"""Diagnostic: few-query shared-stream shapes vs the exact kernel (mismatch counts)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
for nq, rows, k in ((1024, 300_017, 50), (1024, 300_017, 10), (640, 262_144, 10), (4100, 270_001, 10), (4096, 262_200, 50)):
    t = torch.randn(rows, 64, device="cuda", generator=g)
    q = torch.randn(nq, 64, device="cuda", generator=g)
    ids, sc, st = ops.score_topk(q, ops.PreparedTable(t), k, return_stats=True)
    ide, sce = ops.score_topk_exact(q, t, k)
    bad = (ids != ide).any(dim=1)
    print(nq, rows, k, "bad queries", int(bad.sum()), "uncertain", st["uncertain"], "first bad", bad.nonzero().flatten()[:8].tolist())
    if bad.any():
        i = int(bad.nonzero()[0])
        print("  got ", ids[i, :12].tolist()); print("  want", ide[i, :12].tolist())
print("with exclusions")
for nq, rows, k in ((1024, 300_017, 50), (4096, 262_200, 50), (40, 30000, 50)):
    t = torch.randn(rows, 64, device="cuda", generator=g)
    q = torch.randn(nq, 64, device="cuda", generator=g)
    excl = torch.randint(0, rows, (nq, 20), device="cuda", generator=g)
    top, _ = ops.score_topk_exact(q[:64], t, 5)
    excl[:min(64, nq), :5] = top[:min(64, nq)]
    ids, sc, st = ops.score_topk(q, ops.PreparedTable(t), k, excl, return_stats=True)
    ide, sce = ops.score_topk_exact(q, t, k, excl)
    bad = (ids != ide).any(dim=1)
    print(nq, rows, k, "bad queries", int(bad.sum()), "uncertain", st["uncertain"], "first bad", bad.nonzero().flatten()[:8].tolist())
    if bad.any():
        i = int(bad.nonzero()[0])
        print("  got ", ids[i, :12].tolist()); print("  want", ide[i, :12].tolist()); print("  excl", excl[i, :8].tolist())
