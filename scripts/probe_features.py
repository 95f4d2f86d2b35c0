"""Early-fusion feature path at the benchmark shape with pre-packed feature tensors: forward + backward time of
SharedEmbeddingModule.get_complete_embeddings + hydra_projection (python scripts/probe_features.py [B] [L] [d])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import hydra as H
from generative_recommenders_amd.shared_embedding import SharedEmbeddingModule

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = "cuda"
if os.environ.get("PROBE_DIRECT", "1") == "1":     # the training loop's gradient modes (bench_train.py uses them too)
    from generative_recommenders_amd import enc_ops, ops
    ops.set_fused_table_grad(True)
    enc_ops.set_direct_param_grad(True)
ft = {"user_sparse": ["103", "104", "105", "109"], "item_sparse": ["100", "117", "111", "118", "101", "102"],
      "user_array": ["106", "107", "108", "110"], "item_array": [], "user_continual": [], "item_continual": [], "item_emb": ["81"]}
stats = {k: 1000 + 37 * i for i, k in enumerate(ft["user_sparse"] + ft["item_sparse"] + ft["user_array"])}
args = H.make_args(d, L, device=dev)
torch.manual_seed(0)
se = SharedEmbeddingModule(100_000, 400_000, stats, ft, args, None)
m = H.Hydra(100_000, 400_000, stats, ft, args, se).to(dev).train()
g = torch.Generator(device=dev).manual_seed(1)
feats = {}
for k in ft["user_sparse"] + ft["item_sparse"]:
    feats[k] = torch.randint(0, stats[k] + 1, (B, L), device=dev, generator=g)
for k in ft["user_array"]:
    feats[k] = torch.randint(0, stats[k] + 1, (B, L, 5), device=dev, generator=g)
feats["81"] = torch.randn(B, L, 32, device=dev, generator=g)
seq = torch.randint(1, 400_001, (B, L), device=dev, generator=g)
tok = torch.ones(B, L, dtype=torch.long, device=dev)
ts = torch.cumsum(torch.randint(1, 1000, (B, L), device=dev, generator=g), 1)
z = torch.zeros(B, L, dtype=torch.long, device=dev)


def step():
    m.zero_grad(set_to_none=True)
    x, pm = m.assemble_inputs(seq, tok, ts, z, z, feats)
    x.sum().backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"input assembly with {sum(len(v) for v in ft.values())} features, B={B} L={L} d={d}: "
      f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms per forward + backward")
