"""infer.py's file hand-off at full size through ann_files.retrieve_files: embedding.fbin (4,000,001 x 64),
id.u64bin, query.fbin (1,000,000 x 64) -> id100.u64bin.  Times the whole call (file reads from page cache, host ->
device copies over PCIe, scoring + top-10, device -> host, id mapping, result write) beside the device-resident
search it contains."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from generative_recommenders_amd import ann_files, ops      # noqa: E402

rng = np.random.default_rng(0)
rows, nq, d = 4_000_001, 1_000_000, 64
table = rng.standard_normal((rows, d), dtype=np.float32)
queries = rng.standard_normal((nq, d), dtype=np.float32)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    p = lambda n: os.path.join(tmp, n)
    ann_files.save_emb(table, p("embedding.fbin"))
    ann_files.save_emb(np.arange(rows, dtype=np.uint64).reshape(-1, 1) + 1000, p("id.u64bin"))
    ann_files.save_emb(queries, p("query.fbin"))
    for rep in range(2):
        t0 = time.perf_counter()
        ann_files.retrieve_files(p("embedding.fbin"), p("id.u64bin"), p("query.fbin"), p("id100.u64bin"), 10)
        whole = time.perf_counter() - t0
    tq, qq = torch.from_numpy(table).cuda(), torch.from_numpy(queries).cuda()
    prep = ops.PreparedTable(tq)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.score_topk(qq, prep, 10)
        torch.cuda.synchronize()
        dev = time.perf_counter() - t0
    out = ann_files.read_result_ids(p("id100.u64bin"))
    print(f"retrieve_files, files to file: {whole * 1e3:.0f} ms = {nq / whole / 1e6:.2f} M queries/s "
          f"(device-resident search inside it: {dev * 1e3:.0f} ms = {nq / dev / 1e6:.2f} M queries/s); result {out.shape}")
