"""Host-side time of a train step: how long the Python / autograd / ctypes code of one step takes to enqueue its work,
against how long the GPU takes to run it.  Steps are issued back to back without synchronising; while the host runs
ahead of the GPU, the interval between two returns of step() is the host's own time."""
import argparse
import time

import torch

from generative_recommenders_amd import enc_ops, hydra as H, ops, synth
from generative_recommenders_amd.optim import HmxAdamW

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--seq-len", type=int, default=100)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = H.build_model(1_000_000, 4_000_000, d=a.dim, L=a.seq_len, dropout=0.2, device=dev, seed=1234).train()
ops.set_fused_table_grad(True)
enc_ops.set_direct_param_grad(True)
opt = HmxAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
bt = synth.train_batch(a.batch, a.seq_len, 1_000_000, 4_000_000, dev, seed=99)
feat = [[] for _ in range(a.batch)]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"], bt["act"],
                 bt["dwell"])
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
marks = [time.perf_counter()]
for _ in range(a.steps):
    step()
    marks.append(time.perf_counter())
torch.cuda.synchronize()
end = time.perf_counter()
gaps = [1e3 * (b - c) for b, c in zip(marks[1:], marks[:-1])]
print("host ms per step (no sync between steps):", " ".join(f"{g:.2f}" for g in gaps))
print(f"wall ms per step incl. final sync: {1e3 * (end - marks[0]) / a.steps:.2f}")
# phases of the host time of one step, GPU idle at the start of each
torch.cuda.synchronize()
t0 = time.perf_counter()
opt.zero_grad(set_to_none=True)
loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"], bt["act"], bt["dwell"])
t1 = time.perf_counter()
loss.backward()
t2 = time.perf_counter()
opt.step()
t3 = time.perf_counter()
torch.cuda.synchronize()
print(f"host forward {1e3 * (t1 - t0):.2f} ms, backward {1e3 * (t2 - t1):.2f} ms, optimizer {1e3 * (t3 - t2):.2f} ms")
