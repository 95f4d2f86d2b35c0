"""One launch of the retrieval coarse kernel for counter collection (run under rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(4_000_001, 64, device="cuda", generator=g)
q = torch.randn(nq, 64, device="cuda", generator=g)
prep = ops.PreparedTable(t)
ops.score_topk(q, prep, 10)
torch.cuda.synchronize()
