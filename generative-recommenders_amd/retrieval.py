"""Exact top-K retrieval over an item table sharded by contiguous id range (one shard per rank).

Replaces the step infer.py:405-420 delegates to an external FAISS-HNSW process (query.fbin ->
top-10 ids) and the scoring of HardNegativeMiner (hydra3.py:993-1006) when the table is sharded.

Ranks score queries against their own rows (HIP scoring + fused top-K), the per-shard lists are
exchanged with one all_gather per query chunk (RCCL on GPUs), and each rank merges them with the
same (score desc, id asc) order, so the result is identical to the single-GPU one.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Rows [lo, hi) of the table owned by `rank` (contiguous, ceil-divided; trailing ranks may be empty)."""
    per = -(-n_rows // world_size)
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def grid_layout(world_size, rank, item_shards=None):
    """2-D decomposition of `world_size` ranks: S item-range shards x G = world/S query groups.
    Returns (shard_index, S, query_group, G); rank = query_group * S + shard_index.  item_shards=None -> S = world
    (every rank owns a different slice of the table and scores all queries)."""
    S = world_size if not item_shards else int(item_shards)
    if S < 1 or world_size % S:
        raise ValueError(f"item_shards={S} does not divide world_size={world_size}")
    return rank % S, S, rank // S, world_size // S


class ShardedRetriever:
    """search(queries, k) -> (ids int64 (n_q, k), scores fp32 (n_q, k)), identical on every rank.

    The table is cut into S contiguous item-id ranges; G = world / S ranks hold a copy of each range and split the
    queries between them (grid_layout).  A rank scores its slice of the queries against its range, one all_gather
    per query chunk brings every rank's (ids, scores) to every rank, and the S lists of each query are merged with
    the (score desc, id asc) order -- the same result as one GPU scoring everything.  S = world is plain item-range
    sharding; S < world trades table replication (1.5 GB per copy at 4 M x 64) for longer item ranges per wave,
    which the scoring kernel turns into higher MFMA utilisation (its per-query start-up cost is per range).

    local_topk / merge default to the HIP kernels; they are injectable so that the exchange logic can be
    exercised without a GPU (CPU tests pass the oracle's functions with the gloo backend).
    """

    def __init__(self, table_shard, id_offset=0, group=None, local_topk=None, merge=None, chunk=262144,
                 item_shards=None):
        self.id_offset = int(id_offset)
        self.group = group
        self.chunk = int(chunk)
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.shard_index, self.S, self.qgroup, self.G = grid_layout(self.world, self.rank, item_shards)
        self.prepared = None
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
        if self.prepared is None or queries.shape[0] == 0:          # empty shard / empty slice: nothing eligible
            ids = torch.full((queries.shape[0], k), -1, dtype=torch.int64, device=queries.device)
            return ids, torch.full((queries.shape[0], k), float("-inf"), device=queries.device)
        ids, sc, st = ops.score_topk(queries, self.prepared, k, id_offset=self.id_offset, return_stats=True)
        self.last_stats["uncertain"] = self.last_stats.get("uncertain", 0) + st["uncertain"]
        return ids, sc

    def _all_gather(self, t):
        """(n, k) per rank -> (world, n, k); one collective, launched asynchronously."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        try:
            return out, dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)
        except (RuntimeError, NotImplementedError):      # backend without the flat form
            parts = list(out.unbind(0))
            return out, dist.all_gather(parts, t, group=self.group, async_op=True)

    def search(self, queries, k):
        self.last_stats = {}
        if self.world == 1:
            return self.local_topk(queries, k)
        n_q = queries.shape[0]
        per_group = -(-n_q // self.G)                       # queries per query group (last group may be short)
        lo = min(self.qgroup * per_group, n_q)
        mine = queries[lo:min(lo + per_group, n_q)]
        out_ids = torch.empty(n_q, k, dtype=torch.int64, device=queries.device)
        out_sc = torch.empty(n_q, k, dtype=torch.float32, device=queries.device)
        pending = None
        for c0 in range(0, per_group, self.chunk):
            qc = min(self.chunk, per_group - c0)            # the same chunk length on every rank
            part = mine[c0:c0 + qc]
            ids, sc = self.local_topk(part, k)
            if part.shape[0] < qc:                          # short last group: pad so the collective is uniform
                pad = qc - part.shape[0]
                ids = torch.cat([ids, torch.full((pad, k), -1, dtype=ids.dtype, device=ids.device)])
                sc = torch.cat([sc, torch.full((pad, k), float("-inf"), dtype=sc.dtype, device=sc.device)])
            g_ids, w1 = self._all_gather(ids.contiguous())
            g_sc, w2 = self._all_gather(sc.contiguous())
            if pending is not None:                         # merge the previous chunk while this exchange runs
                self._finish(pending, out_ids, out_sc, per_group)
            pending = ((w1, w2), g_ids, g_sc, c0, qc)
        if pending is not None:
            self._finish(pending, out_ids, out_sc, per_group)
        return out_ids, out_sc

    def _finish(self, pending, out_ids, out_sc, per_group):
        works, g_ids, g_sc, c0, qc = pending
        for w in works:
            w.wait()
        k = g_ids.shape[-1]
        # (world, qc, k) with rank = group * S + shard  ->  (S, G * qc, k): the S lists of every query side by side
        ci = g_ids.view(self.G, self.S, qc, k).permute(1, 0, 2, 3).reshape(self.S, self.G * qc, k).contiguous()
        cs = g_sc.view(self.G, self.S, qc, k).permute(1, 0, 2, 3).reshape(self.S, self.G * qc, k).contiguous()
        ids, sc = self.merge(ci, cs)
        n_q = out_ids.shape[0]
        for g in range(self.G):
            q0 = g * per_group + c0
            n = max(0, min(qc, n_q - q0))
            if n:
                out_ids[q0:q0 + n] = ids[g * qc:g * qc + n]
                out_sc[q0:q0 + n] = sc[g * qc:g * qc + n]
