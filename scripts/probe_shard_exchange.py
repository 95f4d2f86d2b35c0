"""One item range of an S-way sharded table with the threshold exchange, emulated on one GPU:
python scripts/probe_shard_exchange.py [S] [rows_total] [n_q].  The S ranges are sampled one after the other (on S
GPUs these run side by side), the largest threshold per query is taken, and the seeded pass of range 0 is timed
against the stand-alone pass of the same range."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_001
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 262_144
d, k = 64, 10
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(rows, d, device="cuda", generator=g)
q = torch.randn(nq, d, device="cuda", generator=g)
per = -(-rows // S)
preps = [ops.PreparedTable(t[s * per:min((s + 1) * per, rows)]) for s in range(S)]


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.time()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.time() - t0) * 1e3


for it in range(3):
    m = min(5, max(2, -(-3 * k // (4 * S))))          # retrieval.ShardedRetriever._sample_values
    nv = int(os.environ.get("PROBE_VALUES", str(2 * m)))
    thr, ms_sample = timed(lambda: ops.sample_thresholds(q, preps[0], k, S, nv))
    vals = [thr] + [ops.sample_thresholds(q, preps[s], k, S, nv) for s in range(1, S)]
    (start), ms_comb = timed(lambda: torch.stack(vals).amax(0) if nv == 1 else
                             ops.kth_largest(torch.stack(vals), k))
    (ids, sc, bounds, failed), ms_pass = timed(lambda: ops.score_topk_seeded(q, preps[0], k, start))
    _, ms_alone = timed(lambda: ops.score_topk(q, preps[0], k))
    fl = 2.0 * nq * preps[0].table.shape[0] * d
    print(f"S={S} range rows={preps[0].table.shape[0]} nq={nq}: sample {ms_sample:.2f} ms + seeded pass {ms_pass:.2f} ms "
          f"+ combine {ms_comb:.2f} ms = {ms_sample + ms_pass + ms_comb:.2f} ms; stand-alone {ms_alone:.2f} ms; ideal at 1385 TF {fl / 1385e12 * 1e3:.2f} ms; "
          f"failed {int(failed.sum())}, short lists {int((ids[:, -1] < 0).sum())}", flush=True)

# the merged result equals the unsharded one
lists = [ops.score_topk_seeded(q, preps[s], k, start, id_offset=s * per) for s in range(S)]
ids, sc = ops.topk_merge(torch.stack([l[0] for l in lists]), torch.stack([l[1] for l in lists]))
bound = torch.stack([l[2] for l in lists]).amax(0)
flagged = ~(sc[:, k - 1] > bound) | (torch.stack([l[3] for l in lists]).amax(0) > 0)
want_ids, want_sc = ops.score_topk(q, ops.PreparedTable(t), k)
ok = flagged | ((ids == want_ids).all(1) & (sc == want_sc).all(1))
print(f"flagged {int(flagged.sum())} of {nq}; unflagged queries equal to the unsharded result: {bool(ok.all())}")
