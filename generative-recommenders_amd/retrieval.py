"""Exact top-K retrieval over an item table sharded by contiguous id range (one shard per rank).

Replaces the step infer.py:405-420 delegates to an external FAISS-HNSW process (query.fbin ->
top-10 ids) and the scoring of HardNegativeMiner (hydra3.py:993-1006) when the table is sharded.

Every rank scores the same queries against its own rows (HIP scoring + fused top-K), the per-shard
lists are exchanged with one all_gather per query chunk (RCCL on GPUs), and each rank merges them
with the same (score desc, id asc) order, so the result is identical to the single-GPU one.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Rows [lo, hi) of the table owned by `rank` (contiguous, ceil-divided; trailing ranks may be empty)."""
    per = -(-n_rows // world_size)
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


class ShardedRetriever:
    """search(queries, k) -> (ids int64 (n_q, k), scores fp32 (n_q, k)), identical on every rank.

    local_topk / merge default to the HIP kernels; they are injectable so that the exchange logic can be
    exercised without a GPU (CPU tests pass the oracle's functions with the gloo backend).
    """

    def __init__(self, table_shard, id_offset=0, group=None, local_topk=None, merge=None, chunk=262144):
        self.id_offset = int(id_offset)
        self.group = group
        self.chunk = int(chunk)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if local_topk is None:
            from . import ops
            self.prepared = ops.PreparedTable(table_shard) if table_shard.shape[0] > 0 else None
            local_topk = self._hip_local
            merge = ops.topk_merge
        self.local_topk, self.merge = local_topk, merge
        self.last_stats = {}

    def refresh(self, table_shard=None):
        """Call after the table's weights changed (re-derives the bf16 copy)."""
        if self.prepared is not None:
            self.prepared.refresh(table_shard)

    def _hip_local(self, queries, k):
        from . import ops
        if self.prepared is None:          # empty shard: nothing eligible
            ids = torch.full((queries.shape[0], k), -1, dtype=torch.int64, device=queries.device)
            return ids, torch.full((queries.shape[0], k), float("-inf"), device=queries.device)
        ids, sc, st = ops.score_topk(queries, self.prepared, k, id_offset=self.id_offset, return_stats=True)
        self.last_stats["uncertain"] = self.last_stats.get("uncertain", 0) + st["uncertain"]
        return ids, sc

    def search(self, queries, k):
        self.last_stats = {}
        if self.world == 1:
            return self.local_topk(queries, k)
        out_ids, out_sc, pending = [], [], None
        for q0 in range(0, queries.shape[0], self.chunk):
            ids, sc = self.local_topk(queries[q0:q0 + self.chunk], k)
            ids, sc = ids.contiguous(), sc.contiguous()
            g_ids = [torch.empty_like(ids) for _ in range(self.world)]
            g_sc = [torch.empty_like(sc) for _ in range(self.world)]
            works = (dist.all_gather(g_ids, ids, group=self.group, async_op=True),
                     dist.all_gather(g_sc, sc, group=self.group, async_op=True))
            if pending is not None:                      # merge the previous chunk while this exchange runs
                self._finish(pending, out_ids, out_sc)
            pending = (works, g_ids, g_sc)
        if pending is not None:
            self._finish(pending, out_ids, out_sc)
        return torch.cat(out_ids), torch.cat(out_sc)

    def _finish(self, pending, out_ids, out_sc):
        works, g_ids, g_sc = pending
        for w in works:
            w.wait()
        ids, sc = self.merge(torch.stack(g_ids), torch.stack(g_sc))
        out_ids.append(ids)
        out_sc.append(sc)
