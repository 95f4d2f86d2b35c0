"""Quick timing probe for the retrieval kernels (not the bench): python scripts/probe_topk.py [n_q]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows, d = 4_000_001, 64
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(rows, d, device="cuda", generator=g)
q = torch.randn(nq, d, device="cuda", generator=g)
prep = ops.PreparedTable(t)
for k in (10, 50):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        ids, sc, st = ops.score_topk(q, prep, k, return_stats=True)
        torch.cuda.synchronize(); dt = time.time() - t0
        fl = 2.0 * nq * rows * d
        print(f"k={k} nq={nq} {dt*1e3:.1f} ms  {nq/dt:.0f} q/s  {fl/dt/1e12:.1f} TFLOP/s  uncertain={st['uncertain']}", flush=True)
nqe = 2048
torch.cuda.synchronize(); t0 = time.time()
ops.score_topk_exact(q[:nqe], t, 10)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"exact nq={nqe} {dt*1e3:.1f} ms {2.0*nqe*rows*d/dt/1e12:.1f} TFLOP/s")
