"""Retrieval over one item-range shard: python scripts/probe_shard.py [rows] [n_q] (defaults 500001 rows, 262144 queries)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_001
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 262_144
d = 64
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(rows, d, device="cuda", generator=g)
q = torch.randn(nq, d, device="cuda", generator=g)
prep = ops.PreparedTable(t)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    ids, sc, st = ops.score_topk(q, prep, 10, return_stats=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    fl = 2.0 * nq * rows * d
    print(f"rows={rows} nq={nq} {dt*1e3:.2f} ms  {fl/dt/1e12:.1f} TFLOP/s  ideal at 1385 TF: {fl/1385e12*1e3:.2f} ms  uncertain={st['uncertain']}", flush=True)
