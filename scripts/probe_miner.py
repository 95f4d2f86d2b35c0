"""Miner-shaped top-50 launch with and without exclusion lists (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops
n_excl = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(4_000_001, 64, device="cuda", generator=g)
q = torch.randn(1024, 64, device="cuda", generator=g)
prep = ops.PreparedTable(t)
excl = torch.randint(1, 4_000_001, (1024, n_excl), device="cuda", generator=g) if n_excl else None
for _ in range(4):
    ops.score_topk(q, prep, 50, excl)
torch.cuda.synchronize()
