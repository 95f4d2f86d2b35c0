"""Exact top-K retrieval over an item table sharded by contiguous id range (one shard per rank).

Replaces the step infer.py:405-420 delegates to an external FAISS-HNSW process (query.fbin ->
top-10 ids) and the scoring of HardNegativeMiner (hydra3.py:993-1006) when the table is sharded.

Ranks score queries against their own rows (HIP scoring + fused top-K); per query chunk one
all_to_all hands every rank the lists of ITS slice of the queries (RCCL on GPUs), the rank merges
them with the (score desc, id asc) order, and one all_gather spreads the merged results, so the
result is identical to the single-GPU one.
"""
import torch
import torch.distributed as dist


def _gloo_on_device():
    from . import parallel
    return parallel.GLOO_ON_DEVICE_TENSORS


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
    queries between them (grid_layout).  A rank scores its group's queries against its range; the S lists of each
    query meet on one rank of the group (see search), are merged with the (score desc, id asc) order, and the merged
    results are gathered everywhere -- the same result as one GPU scoring everything.  S = world is plain item-range
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
        staged = dist.get_backend(self.group) == "gloo" and t.is_cuda and not _gloo_on_device()
        src = t.cpu() if staged else t
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        try:
            w = dist.all_gather_into_tensor(out, src, group=self.group, async_op=not staged)
        except (RuntimeError, NotImplementedError):      # backend without the flat form
            w = dist.all_gather(list(out.unbind(0)), src, group=self.group, async_op=not staged)
        if staged:
            return out.to(t.device), self._Done()
        return out, w

    def _group_exchange(self, send):
        """(S, n, w) -> (S, n, w): block j of the result is what the j-th rank of this rank's query group sent in ITS
        block number shard_index.  One all_to_all_single over the whole group of ranks with zero-length messages
        between ranks of different query groups (no sub-communicators to create).  Same staging rule as _all_gather."""
        staged = dist.get_backend(self.group) == "gloo" and send.is_cuda and not _gloo_on_device()
        src = send.cpu() if staged else send
        out = torch.empty_like(src)
        S, n, w = src.shape
        if self.S == self.world:
            work = dist.all_to_all_single(out, src, group=self.group, async_op=not staged)
        else:
            splits = [n if r // self.S == self.qgroup else 0 for r in range(self.world)]
            work = dist.all_to_all_single(out.view(S * n, w), src.view(S * n, w), splits, splits, group=self.group,
                                          async_op=not staged)
        if staged:
            return out.to(send.device), self._Done()
        return out, work

    def _range_offsets(self, device):
        """First item id of every rank's range (exchanged once): lists travel with 32-bit rows relative to it."""
        if getattr(self, "_offsets", None) is None or self._offsets.device != device:
            mine = torch.tensor([self.id_offset], dtype=torch.int64, device=device)
            got, w = self._all_gather(mine)
            w.wait()
            self._offsets = got.reshape(self.world).clone()
        return self._offsets

    def search(self, queries, k):
        """Every chunk of a query group's queries is cut into S slices, one per rank of the group.  A rank scores the
        whole chunk against its range, sends each group-mate the lists of that mate's slice in ONE exchange
        (_group_exchange; a record per query: k 32-bit range-relative rows, k scores and, with the threshold exchange,
        the pass's drop bound and give-up flag), merges the S lists of its own slice, and one all_gather of the merged
        (slice, k) results gives every rank the whole answer.  Against gathering every list on every rank this moves
        1 / S of the list volume per rank and merges 1 / S of the queries per rank."""
        self.last_stats = {}
        if self.world == 1:
            return self.local_topk(queries, k)
        sample, seeded_topk, exact_topk = self.seeded if self.seeded is not None else (None, None, None)
        n_q, dev = queries.shape[0], queries.device
        per_group = -(-n_q // self.G)                       # queries per query group (last group may be short)
        lo = min(self.qgroup * per_group, n_q)
        mine = queries[lo:min(lo + per_group, n_q)]
        out_ids = torch.empty(n_q, k, dtype=torch.int64, device=dev)
        out_sc = torch.empty(n_q, k, dtype=torch.float32, device=dev)
        offsets = self._range_offsets(dev)
        flags, flag_queries = [], []                        # per chunk: 'not proven complete' flags and their query numbers
        in_flight = []                                      # chunks whose exchange / result gather has been started

        def chunk_of(c0):
            qc = min(self.chunk, per_group - c0)            # the same chunk length on every rank
            part = mine[c0:c0 + qc]
            if part.shape[0] < qc:       # short last group: zero queries keep the collectives uniform (results dropped)
                part = torch.cat([part, torch.zeros(qc - part.shape[0], queries.shape[1], dtype=queries.dtype, device=dev)])
            return qc, part

        def start_sampling(c0):
            """Samples chunk c0 and starts the exchange of the samples (queued ahead of the previous chunk's scoring
            pass, so that under RCCL the exchange runs while that pass computes)."""
            qc, part = chunk_of(c0)
            g_thr, w = self._all_gather(sample(part, k).contiguous())
            return c0, qc, part, g_thr, w

        def advance(keep_newest=True):
            """Moves every chunk in flight (but the newest, whose exchange has just been started) one stage on, oldest
            first: gathered results are written out, exchanged lists are merged and their results' gather started --
            so a chunk's two collectives run while the next chunks are scored and merged."""
            i = 0
            while i < len(in_flight) - (1 if keep_newest else 0):
                st = in_flight[i]
                if st[0] == "lists":
                    in_flight[i] = self._merge_slice(st, k, offsets)
                    i += 1
                else:
                    fl = self._store_results(in_flight.pop(i), k, out_ids, out_sc, per_group, flag_queries)
                    if self.seeded is not None:
                        flags.append(fl)

        nxt = start_sampling(0) if (self.seeded is not None and per_group > 0) else None
        for c0 in range(0, per_group, self.chunk):
            if self.seeded is not None:
                c0, qc, part, g_thr, w = nxt
                nxt = start_sampling(c0 + self.chunk) if c0 + self.chunk < per_group else None
                w.wait()
                ids, sc, bounds, failed = seeded_topk(part, k, self._combine_samples(g_thr, qc, k))
                extra = torch.stack([bounds.to(torch.float32), failed.to(torch.float32)], dim=1)
            else:
                qc, part = chunk_of(c0)
                ids, sc = self.local_topk(part, k)
                extra = None
            in_flight.append(self._send_lists(ids, sc, extra, c0, qc))
            advance()
        while in_flight:
            advance(keep_newest=False)
        if self.seeded is None:
            return out_ids, out_sc
        redo = torch.cat(flags).nonzero().flatten() if flags else torch.empty(0, dtype=torch.int64, device=dev)
        # the flags come out of an all_gather: every rank holds the same list, so the collectives below agree
        self.last_stats["uncertain"] = int(redo.numel())
        if redo.numel():
            qidx = torch.cat(flag_queries).to(dev)[redo]
            ids, sc = exact_topk(queries[qidx], k)          # every group repairs all of them: same lists in every group
            n = qidx.numel()
            st = self._merge_slice(self._send_lists(ids, sc, None, 0, n), k, offsets)
            r_ids, r_sc, _ = self._gathered_results(st, k)
            out_ids[qidx], out_sc[qidx] = r_ids[self.qgroup, :n], r_sc[self.qgroup, :n]
        return out_ids, out_sc

    # ---- the per-slice exchange -------------------------------------------------------------------
    def _send_lists(self, ids, sc, extra, c0, qc):
        """Starts the exchange of one chunk's lists: (qc, k) global ids (-1: none), scores, optional (qc, 2) extras."""
        k = ids.shape[1]
        qs = max(1, -(-qc // self.S))                       # queries per slice
        local = torch.where(ids >= 0, ids - self.id_offset, ids).to(torch.int32)
        cols = [local, sc.to(torch.float32).contiguous().view(torch.int32)]
        if extra is not None:
            cols.append(extra.contiguous().view(torch.int32))
        rec = torch.cat(cols, dim=1)
        pad = self.S * qs - rec.shape[0]
        if pad:                                             # slices of equal length: empty lists, nothing dropped
            fill = torch.cat([torch.full((pad, k), -1, dtype=torch.int32, device=rec.device),
                              torch.full((pad, k), float("-inf"), device=rec.device).view(torch.int32)] +
                             ([torch.tensor([[float("-inf"), 0.0]], device=rec.device).expand(pad, 2).contiguous()
                               .view(torch.int32)] if extra is not None else []), dim=1)
            rec = torch.cat([rec, fill])
        got, work = self._group_exchange(rec.view(self.S, qs, rec.shape[1]).contiguous())
        return "lists", work, got, c0, qc, qs, extra is not None

    def _merge_slice(self, st, k, offsets):
        """Merges the S lists of this rank's slice and starts the all_gather of the merged results."""
        _, work, got, c0, qc, qs, has_extra = st
        work.wait()
        base = offsets[self.qgroup * self.S:(self.qgroup + 1) * self.S].view(self.S, 1, 1)
        local = got[..., :k].to(torch.int64)
        ci = torch.where(local >= 0, local + base, local).contiguous()
        cs = got[..., k:2 * k].contiguous().view(torch.float32)
        ids, sc = self.merge(ci, cs)
        if has_extra:
            ex = got[..., 2 * k:2 * k + 2].contiguous().view(torch.float32)
            bound, failed = ex[..., 0].amax(0), ex[..., 1].amax(0) > 0
            flagged = (~(sc[:, k - 1] > bound) | failed).to(torch.int32)
        else:
            flagged = torch.zeros(qs, dtype=torch.int32, device=got.device)
        res = torch.cat([ids.contiguous().view(torch.int32), sc.to(torch.float32).contiguous().view(torch.int32),
                         flagged.unsqueeze(1)], dim=1)       # (qs, 3 k + 1): one message per rank
        g_res, w = self._all_gather(res.contiguous())
        return "results", w, g_res, c0, qc, qs, has_extra

    def _gathered_results(self, st, k):
        """(G, S * qs, k) ids, scores and (G, S * qs) flags of one chunk, query order within every group."""
        _, work, g_res, c0, qc, qs, _ = st
        work.wait()
        r = g_res.view(self.G, self.S * qs, 3 * k + 1)
        ids = r[..., :2 * k].contiguous().view(torch.int64)
        sc = r[..., 2 * k:3 * k].contiguous().view(torch.float32)
        return ids, sc, r[..., 3 * k] != 0

    def _store_results(self, st, k, out_ids, out_sc, per_group, flag_queries):
        """Writes one chunk's results; returns the 'not proven complete' flags of its valid queries (their global query
        numbers are appended to flag_queries)."""
        ids, sc, flagged = self._gathered_results(st, k)
        c0, qc = st[3], st[4]
        n_q = out_ids.shape[0]
        keep = []
        for g in range(self.G):
            q0 = g * per_group + c0
            n = max(0, min(qc, min(per_group - c0, n_q - q0)))
            if n:
                out_ids[q0:q0 + n] = ids[g, :n]
                out_sc[q0:q0 + n] = sc[g, :n]
                keep.append(flagged[g, :n])
                flag_queries.append(torch.arange(q0, q0 + n))
        return torch.cat(keep) if keep else flagged.reshape(-1)[:0]

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
