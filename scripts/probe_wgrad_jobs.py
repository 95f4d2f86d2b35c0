"""Lists the weight-gradient products of one train step as the batched launches see them (PYTHONPATH=. on the GPU box)."""
import argparse
import collections

import torch

from generative_recommenders_amd import enc_ops, hydra as H, ops, synth
from generative_recommenders_amd.optim import HmxAdamW

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--seq-len", type=int, default=200)
ap.add_argument("--dim", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = H.build_model(1_000_000, 4_000_000, d=a.dim, L=a.seq_len, dropout=0.2, device=dev, seed=1234).train()
ops.set_fused_table_grad(True)
enc_ops.set_direct_param_grad(True)
opt = HmxAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
bt = synth.train_batch(a.batch, a.seq_len, 1_000_000, 4_000_000, dev, seed=99)
feat = [[] for _ in range(a.batch)]
launches = []
real_flush = enc_ops._WgradQueue.flush


kept = []
in_step = []
step_marks = []


def flush(self):
    if self.jobs:
        kept.append((list(self.jobs), list(self.keep)))
        del kept[:-16]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_flush(self)
        e1.record()
        in_step.append((e0, e1))
        return
        launches.append([(j.T, j.N, j.K, j.ldy, j.ldx, (j.dY or 0) % 16, (j.X or 0) % 16, bool(j.dbias), bool(j.in_rowmask),
                          j.n_group_padded) for j in self.jobs])
    real_flush(self)


enc_ops._WgradQueue.flush = flush
import time
for it in range(int(__import__("os").environ.get("PROBE_STEPS", "2"))):
    step_marks.append(len(in_step))
    launches.clear()
    opt.zero_grad(set_to_none=True)
    loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"], bt["act"],
                 bt["dwell"])
    loss.backward()
    opt.step()
torch.cuda.synchronize()
step_marks.append(len(in_step))
for i in range(len(step_marks) - 1):
    print("step", i, "launches (events):", " ".join(f"{a.elapsed_time(b) * 1000:.0f}" for a, b in in_step[step_marks[i]:step_marks[i + 1]]),
          "us")
print("in-step launch times (events):", " ".join(f"{a.elapsed_time(b) * 1000:.0f}" for a, b in in_step[-len(launches):]), "us")
for i, jobs in enumerate(launches):
    flop = sum(2.0 * T * N * K for (T, N, K, *_r) in jobs)
    print(f"launch {i}: {len(jobs)} products, {flop / 1e9:.2f} GFLOP")
    for job, cnt in collections.Counter(jobs).items():
        print("   x%d  T=%d N=%d K=%d ldy=%d ldx=%d dY%%16=%d X%%16=%d bias=%d mask=%d grouped=%d" % ((cnt,) + job))

# the same descriptor arrays again, one launch at a time under events (operands kept alive above)
from generative_recommenders_amd import hmx
lib = hmx.lib()
big = torch.zeros(512 * 1024 * 1024, device=dev)
for i, (jobs, _keep) in enumerate(kept[-len(launches):]):
    n = len(jobs)
    arr = (hmx.WgradDesc * n)(*jobs)
    ws = torch.empty(lib.hmx_linear_wgrad_batch_workspace_bytes(arr, n), dtype=torch.uint8, device=dev)
    for variant in ("as queued", "after a 2 GB fill", "without bias and mask"):
        if variant == "without bias and mask":
            for j in arr:
                j.dbias = None
                j.in_rowmask = None
        times = []
        for _ in range(5):
            if variant == "after a 2 GB fill":
                big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.hmx_linear_wgrad_batch_f32(arr, n, ws.data_ptr(), ws.numel(), hmx.current_stream())
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1000)
        print(f"launch {i} again, {variant}: " + " ".join(f"{t:.0f}" for t in times) + " us")
