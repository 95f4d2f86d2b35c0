"""Row-gather / scatter-add micro-benchmark on uniform random ids over a 4,000,001 x 64 fp32 table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import hmx, ops

n_rows, d = 4_000_001, 64
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.randn(n_rows, d, device="cuda", generator=g)
for n in (1 << 20, 1 << 23, 1 << 24):
    ids = torch.randint(1, n_rows, (n,), device="cuda", generator=g)
    out = ops.gather_rows(table, ids)
    hmx.profile_enable(True)
    hmx.profile_collect("gather_rows")
    for _ in range(5):
        out = ops.gather_rows(table, ids)
    torch.cuda.synchronize()
    ms, cnt = hmx.profile_collect("gather_rows")
    gbs = n * (8 + 2 * d * 4) * cnt / (ms * 1e-3) / 1e9
    print(f"gather n={n} {ms / cnt:.3f} ms  {gbs:.0f} GB/s algorithmic ({gbs / 8000:.1%} of 8 TB/s)")
    grad = torch.randn(n, d, device="cuda", generator=g)
    hmx.profile_collect("scatter_segment_sum")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gt = torch.zeros_like(table)
    ops.scatter_add_rows(grad, ids, n_rows, out=gt)
    e0.record()
    for _ in range(3):
        ops.scatter_add_rows(grad, ids, n_rows, out=gt)
    e1.record()
    torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 3
    sms, scnt = hmx.profile_collect("scatter_segment_sum")
    u = int(torch.unique(ids).numel())
    b = n * (8 + 4 * d) + 2 * u * 4 * d
    print(f"scatter n={n} unique={u} whole {tot:.3f} ms ({b / tot / 1e6:.0f} GB/s algorithmic), segment-sum kernel {sms / max(scnt,1):.3f} ms")
    hmx.profile_enable(False)
