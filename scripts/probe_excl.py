"""Diagnostic: bf16 top-K with exclusion lists vs the exact kernel over a grid of shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
for nq, rows, k, ne, knock in ((40, 300_017, 50, 20, 5), (1024, 30_000, 50, 20, 5), (1024, 300_017, 50, 20, 0), (1024, 300_017, 50, 1, 0),
                               (1024, 300_017, 10, 20, 5), (64, 70_000, 50, 20, 5), (64, 40_000, 50, 20, 5), (1024, 300_017, 50, 0, 0)):
    t = torch.randn(rows, 64, device="cuda", generator=g)
    q = torch.randn(nq, 64, device="cuda", generator=g)
    excl = torch.randint(0, rows, (nq, max(ne, 1)), device="cuda", generator=g) if ne else None
    if knock:
        top, _ = ops.score_topk_exact(q[:64], t, knock)
        excl[:min(64, nq), :knock] = top[:min(64, nq)]
    ids, sc, st = ops.score_topk(q, ops.PreparedTable(t), k, excl, return_stats=True)
    ide, sce = ops.score_topk_exact(q, t, k, excl)
    full = q @ t.T
    if excl is not None:
        full.scatter_(1, excl, float("-inf"))
    idt = full.topk(k, dim=1).indices
    print("   exact-vs-torch bad", int((ide != idt).any(dim=1).sum()), " bf16-vs-torch bad", int((ids != idt).any(dim=1).sum()))
    bad = (ids != ide).any(dim=1)
    print(nq, rows, k, ne, knock, "bad queries", int(bad.sum()), "uncertain", st["uncertain"])
    if bad.any():
        i = int(bad.nonzero()[0])
        missing = [x for x in ide[i].tolist() if x not in ids[i].tolist()]
        extra = [x for x in ids[i].tolist() if x not in ide[i].tolist()]
        print("  query", i, "missing", missing[:6], "extra", extra[:6], "extra in excl:", [x in excl[i].tolist() for x in extra[:6]])
