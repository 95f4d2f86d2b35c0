"""Phase times of ann_files.retrieve_files at full size (see probe_files.py)."""
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
torch.zeros(1).cuda()
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    p = lambda n: os.path.join(tmp, n)
    ann_files.save_emb(table, p("embedding.fbin"))
    ann_files.save_emb(np.arange(rows, dtype=np.uint64).reshape(-1, 1) + 1000, p("id.u64bin"))
    ann_files.save_emb(queries, p("query.fbin"))
    for rep in range(2):
        marks = [("start", time.perf_counter())]

        def mark(name):
            torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))
        t = ann_files.load_fbin(p("embedding.fbin")); ids = ann_files.load_u64bin(p("id.u64bin")); q = ann_files.load_fbin(p("query.fbin"))
        mark("read files")
        tt = torch.from_numpy(t).cuda(); qq = torch.from_numpy(q).cuda()
        mark("host -> device")
        prep = ops.PreparedTable(tt)
        mark("prepare table")
        r = ops.score_topk(qq, prep, 10)[0]
        mark("search")
        rn = r.cpu().numpy()
        mark("device -> host")
        out = ids[np.where(rn < 0, 0, rn)]
        mark("map ids")
        ann_files.write_result_ids(out, p("id100.u64bin"))
        mark("write result")
    print(", ".join(f"{b[0]} {1e3 * (b[1] - a[1]):.0f} ms" for a, b in zip(marks, marks[1:])))
