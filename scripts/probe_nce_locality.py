"""How much of the fused loss's time is the cache level its key rows come from?  Same launch, in-batch draws either
uniform over all M rows (26 MB of keys: Infinity Cache) or confined to a window of rows that fits an XCD's L2; also
draws sorted within each row (every wave walks the key matrix front to back).  PYTHONPATH=. on the GPU box."""
import torch
from generative_recommenders_amd import hmx, ops

dev = "cuda"
M, d, B, K = 102400, 64, 1024, 50
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(M, d, device=dev, generator=g)
k = torch.randn(M, d, device=dev, generator=g)
table = torch.randn(400_000, d, device=dev, generator=g)
lists = torch.randint(1, 400_000, (B, K), device=dev, generator=g)
hard = lists.repeat_interleave(M // B, dim=0)
starts = torch.arange(0, M + 1, M // B, device=dev)


def run(label, inb):
    qq, kk, tt = (t.clone().requires_grad_() for t in (q, k, table))
    for tag in ("nce_fwd", "nce_bwd", "scatter_segment_sum"):
        hmx.profile_collect(tag)
    hmx.profile_enable(True)
    for _ in range(3):
        loss = ops.fused_nce_rows(qq, kk, tt, hard, inb, 0.05, True, hard_groups=(starts, K), n_hard=153,
                                  inb_skip_self=True).mean()
        loss.backward()
    torch.cuda.synchronize()
    f = hmx.profile_collect("nce_fwd")
    b = hmx.profile_collect("nce_bwd")
    hmx.profile_enable(False)
    print(f"{label:46s} forward {f[0] / 3:.3f} ms   backward spans {b[0] / 3:.3f} ms", flush=True)


uni = torch.randint(0, M - 1, (M, 359), device=dev, dtype=torch.int32, generator=g)
run("uniform draws over all rows", uni)
run("draws sorted within each row", uni.sort(dim=1).values.contiguous())
for w in (4096, 16384, 65536):
    run(f"draws confined to the first {w} rows", torch.randint(0, w, (M, 359), device=dev, dtype=torch.int32, generator=g))
