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

    Threshold exchange (S > 1, on by default with the HIP kernels): the scoring pass of a range starts from
    thresholds sampled on that range, and a short range gives a loose sample (the start-up work of a pass depends on
    the absolute number of rows seen, so a 500 K-row range paid almost the start-up of the 4 M-row table).  The S
    ranks that score the same queries therefore exchange their sampled thresholds first (one all_gather of one
    float per query) and all start from the largest: k sampled rows of ANY range above a value make it a valid
    start for every range.  A range's own list may then hold fewer than k rows, so completeness is decided after
    the merge (k-th merged score > every range's drop bound); queries that fail it are re-scored exactly on every
    range at the end of the search (one host synchronisation per search instead of one per chunk).

    local_topk / merge (and `seeded` = (sample, seeded_topk, exact_topk)) default to the HIP kernels; they are
    injectable so that the exchange logic can be exercised without a GPU (CPU tests pass oracle functions with the
    gloo backend).
    """

    def __init__(self, table_shard, id_offset=0, group=None, local_topk=None, merge=None, chunk=262144,
                 item_shards=None, exchange_thresholds=True, seeded=None):
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
            if exchange_thresholds and seeded is None:
                seeded = (self._hip_sample, self._hip_seeded, self._hip_exact)
        self.local_topk, self.merge = local_topk, merge
        self.seeded = seeded if (seeded is not None and self.S > 1) else None
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

    def _nothing(self, n, k, device):
        return (torch.full((n, k), -1, dtype=torch.int64, device=device),
                torch.full((n, k), float("-inf"), dtype=torch.float32, device=device))

    def _sample_values(self, k):
        """Sampled scores per query that a range contributes: 2 m (m per half of its sampled rows) with about 1.5 k
        values over all ranges, so that the k-th largest of the union is rarely cut short by one range's share."""
        m = min(5 if k <= 10 else 8, max(2, -(-3 * k // (4 * self.S))))
        return 2 * m if 2 * m * self.S >= k else 1

    def _hip_sample(self, queries, k):
        from . import ops
        if self.prepared is None:
            nv = self._sample_values(k)
            return torch.full((queries.shape[0],) if nv == 1 else (queries.shape[0], nv), float("-inf"),
                              device=queries.device)
        return ops.sample_thresholds(queries, self.prepared, k, n_ranges=self.S, n_values=self._sample_values(k))

    def _hip_seeded(self, queries, k, thr):
        from . import ops
        if self.prepared is None:
            n = queries.shape[0]
            return (*self._nothing(n, k, queries.device), torch.full((n,), float("-inf"), device=queries.device),
                    torch.zeros(n, dtype=torch.int32, device=queries.device))
        return ops.score_topk_seeded(queries, self.prepared, k, thr, id_offset=self.id_offset)

    def _hip_exact(self, queries, k):
        from . import ops
        if self.prepared is None or queries.shape[0] == 0:
            return self._nothing(queries.shape[0], k, queries.device)
        return ops.score_topk_exact(queries, self.prepared.table, k, id_offset=self.id_offset)

    class _Done:
        def wait(self):
            return True

    def _all_gather(self, t):
        """(n, k) per rank -> (world, n, k); one collective.  RCCL: launched asynchronously on its own stream, wait()
        orders the current stream behind it.  gloo with GPU tensors (tests and rehearsals with several ranks on one
        GPU): the tensor is copied to the host, gathered there and copied back, all before returning -- gloo's own
        handling of GPU tensors (helper threads that copy and synchronise streams) was seen to lock up now and then
        with two processes on one GPU, asynchronous or not."""
        staged = dist.get_backend(self.group) == "gloo" and t.is_cuda
        src = t.cpu() if staged else t
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        try:
            w = dist.all_gather_into_tensor(out, src, group=self.group, async_op=not staged)
        except (RuntimeError, NotImplementedError):      # backend without the flat form
            w = dist.all_gather(list(out.unbind(0)), src, group=self.group, async_op=not staged)
        if staged:
            return out.to(t.device), self._Done()
        return out, w

    def search(self, queries, k):
        self.last_stats = {}
        if self.world == 1:
            return self.local_topk(queries, k)
        if self.seeded is not None:
            return self._search_exchanged(queries, k)
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

    # ---- threshold exchange between the item ranges -----------------------------------------------
    def _search_exchanged(self, queries, k):
        sample, seeded_topk, exact_topk = self.seeded
        n_q, dev = queries.shape[0], queries.device
        per_group = -(-n_q // self.G)
        lo = min(self.qgroup * per_group, n_q)
        mine = queries[lo:min(lo + per_group, n_q)]
        out_ids = torch.empty(n_q, k, dtype=torch.int64, device=dev)
        out_sc = torch.empty(n_q, k, dtype=torch.float32, device=dev)
        flags, flag_queries, pending = [], [], None      # per chunk: 'not proven complete' flags and their query numbers

        def start_sampling(c0):
            """Samples chunk c0 and starts the exchange of the samples (queued ahead of the previous chunk's scoring
            pass, so that under RCCL the exchange runs while that pass computes)."""
            qc = min(self.chunk, per_group - c0)
            part = mine[c0:c0 + qc]
            if part.shape[0] < qc:       # short last group: zero queries keep the collectives uniform (results dropped)
                part = torch.cat([part, torch.zeros(qc - part.shape[0], queries.shape[1], dtype=queries.dtype, device=dev)])
            g_thr, w = self._all_gather(sample(part, k).contiguous())
            return c0, qc, part, g_thr, w

        nxt = start_sampling(0) if per_group > 0 else None
        while nxt is not None:
            c0, qc, part, g_thr, w = nxt
            nxt = start_sampling(c0 + self.chunk) if c0 + self.chunk < per_group else None
            w.wait()
            start = self._combine_samples(g_thr, qc, k)
            ids, sc, bounds, failed = seeded_topk(part, k, start)
            packed = torch.cat([sc, bounds.unsqueeze(1), failed.to(sc.dtype).unsqueeze(1)], dim=1).contiguous()
            g_ids, w1 = self._all_gather(ids.contiguous())
            g_pk, w2 = self._all_gather(packed)
            if pending is not None:
                flags.append(self._finish_exchanged(pending, out_ids, out_sc, per_group, flag_queries))
            pending = ((w1, w2), g_ids, g_pk, c0, qc)
        if pending is not None:
            flags.append(self._finish_exchanged(pending, out_ids, out_sc, per_group, flag_queries))
        redo = torch.cat(flags).nonzero().flatten() if flags else torch.empty(0, dtype=torch.int64, device=dev)
        # the flags derive from all-gathered data: every rank holds the same list, so the collectives below agree
        self.last_stats["uncertain"] = int(redo.numel())
        if redo.numel():
            qidx = torch.cat(flag_queries).to(dev)[redo]
            ids, sc = exact_topk(queries[qidx], k)
            g_ids, w1 = self._all_gather(ids.contiguous())
            g_sc, w2 = self._all_gather(sc.contiguous())
            w1.wait()
            w2.wait()
            n = qidx.numel()
            ci = g_ids.view(self.G, self.S, n, k)[self.qgroup].contiguous()
            cs = g_sc.view(self.G, self.S, n, k)[self.qgroup].contiguous()
            ids, sc = self.merge(ci, cs)
            out_ids[qidx], out_sc[qidx] = ids, sc
        return out_ids, out_sc

    def _combine_samples(self, g_thr, qc, k):
        """Start thresholds of one chunk from the samples of the S ranges that score it.  One value per range: the
        largest (k sampled rows of that range beat it).  Several sampled scores per range: the k-th largest of them
        all (k sampled rows of the table beat it) -- tighter, because the best sampled rows are spread over the
        ranges."""
        vals = g_thr.view(self.G, self.S, qc, -1)[self.qgroup]              # (S, qc, values per range)
        if vals.shape[-1] == 1:
            return vals[..., 0].amax(0)
        if vals.is_cuda and k <= 16:
            from . import ops
            return ops.kth_largest(vals, k)
        union = vals.permute(1, 0, 2).reshape(qc, -1)
        return union.topk(k, dim=1).values[:, k - 1].contiguous()

    def _finish_exchanged(self, pending, out_ids, out_sc, per_group, flag_queries):
        """Merge one chunk; returns the per-query 'not proven complete' flags of its valid queries (the matching
        global query numbers are appended to flag_queries)."""
        works, g_ids, g_pk, c0, qc = pending
        for w in works:
            w.wait()
        k = g_ids.shape[-1]
        ci = g_ids.view(self.G, self.S, qc, k).permute(1, 0, 2, 3).reshape(self.S, self.G * qc, k).contiguous()
        pk = g_pk.view(self.G, self.S, qc, k + 2).permute(1, 0, 2, 3).reshape(self.S, self.G * qc, k + 2)
        ids, sc = self.merge(ci, pk[..., :k].contiguous())
        bound = pk[..., k].amax(0)
        failed = pk[..., k + 1].amax(0) > 0
        flagged = ~(sc[:, k - 1] > bound) | failed
        n_q = out_ids.shape[0]
        keep = []
        for g in range(self.G):
            q0 = g * per_group + c0
            n = max(0, min(qc, n_q - q0))
            if n:
                out_ids[q0:q0 + n] = ids[g * qc:g * qc + n]
                out_sc[q0:q0 + n] = sc[g * qc:g * qc + n]
                keep.append(flagged[g * qc:g * qc + n])
                flag_queries.append(torch.arange(q0, q0 + n))
        return torch.cat(keep) if keep else flagged[:0]
