"""Which Python lines launch torch's own small kernels (fills, copies, adds, cats) inside one train step: torch profiler
with stacks over the bench's train step (configs[1] sizes unless overridden)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_recommenders_amd import enc_ops, hydra as H, ops, synth  # noqa: E402
from generative_recommenders_amd.optim import HmxAdamW  # noqa: E402

B, L, d = 1024, 100, int(os.environ.get("PROBE_DIM", "64"))
dev = torch.device("cuda", 0)
model = H.build_model(1_000_000, 4_000_000, d=d, L=L, dropout=0.2, device=dev, seed=1234).train()
ops.set_fused_table_grad(True)
enc_ops.set_direct_param_grad(True)
opt = HmxAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
bt = synth.train_batch(B, L, 1_000_000, 4_000_000, dev, seed=99)
feat = [[] for _ in range(B)]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"], bt["act"],
                 bt["dwell"])
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()

# torch-level calls made from the package, by calling line (dispatch mode: every aten op with the Python line that asked)
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

cnt = collections.Counter()
WANT = ("zeros", "zero_", "fill_", "copy_", "add", "cat", "clone", "mul", "index_select", "index_copy_", "sum", "mean",
        "empty_strided", "_to_copy", "sub", "div", "neg", "where", "eq", "ne", "gt", "arange", "cumsum", "slice_backward",
        "select_backward", "constant_pad_nd", "stack", "masked_fill", "nonzero", "repeat", "expand")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        big = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
        if big or name in ("zeros", "arange", "empty_strided"):
            here = [f for f in traceback.extract_stack() if "generative-recommenders_amd" in f.filename or "generative_recommenders_amd" in f.filename]
            where = f"{os.path.basename(here[-1].filename)}:{here[-1].lineno}" if here else "autograd/other"
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), tuple(args[0]) if args and isinstance(args[0], (list, tuple)) else ())
            if not here:
                where = "|".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[-6:-2])[-70:]
            cnt[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step()
torch.cuda.synchronize()
for (name, where, shp), n in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0][1])):
    print(f"{n:3d} {name:22s} {where:28s} {shp}")
