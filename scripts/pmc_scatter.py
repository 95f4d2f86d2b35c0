"""One scatter-add of 16,777,216 uniform random ids into a 4,000,001 x 64 table for counter collection (run under
rocprofv3 --pmc); the table is zero on entry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
rows, d, n = 4_000_001, 64, 1 << 24
ids = torch.randint(1, rows, (n,), device="cuda", generator=g)
grad = torch.randn(n, d, device="cuda", generator=g)
out = torch.zeros(rows, d, device="cuda")
ops.scatter_add_rows(grad, ids, rows, 0, out=out)
torch.cuda.synchronize()
