"""Kernel-span timing of the two top-K passes: python scripts/probe_topk_spans.py [n_q] [k]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops, hmx
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_001
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(rows, 64, device="cuda", generator=g)
q = torch.randn(nq, 64, device="cuda", generator=g)
prep = ops.PreparedTable(t)
ops.score_topk(q, prep, k)
hmx.profile_enable(True)
for tag in ("topk_seed_bf16", "topk_warm_bf16", "topk_coarse_bf16"):
    hmx.profile_collect(tag)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.score_topk(q, prep, k)
e1.record()
torch.cuda.synchronize()
print(f"rows={rows} nq={nq} k={k} whole {e0.elapsed_time(e1) / 3:.2f} ms")
for tag in ("topk_seed_bf16", "topk_warm_bf16", "topk_coarse_bf16"):
    ms, n = hmx.profile_collect(tag)
    print(f"  {tag}: {ms / max(n, 1):.2f} ms x {n}")
