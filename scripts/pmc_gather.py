"""One launch of the row-gather kernel on uniform random ids for counter collection (run under rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.randn(4_000_001, 64, device="cuda", generator=g)
ids = torch.randint(1, 4_000_001, (1 << 23,), device="cuda", generator=g)
ops.gather_rows(table, ids)
torch.cuda.synchronize()
