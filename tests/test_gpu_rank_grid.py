"""ShardedRetriever's own search code for whole rank grids (up to eight ranges) on ONE GPU in ONE process: every rank is
a thread, the all_gather is a hand-over between the threads, and only one thread runs at a time (the library's
workspace is shared).  Odd shapes: ranges shorter than k, empty trailing ranges, ragged last query group, one query,
k = 1 and k = 16, d = 128.  The result must equal the unsharded kernel's (itself pinned to the C oracle), and the
oracle's on a slice."""
import threading

import numpy as np
import pytest
import torch

from generative_recommenders_amd import ops
from generative_recommenders_amd.retrieval import ShardedRetriever, grid_layout, shard_bounds

pytestmark = pytest.mark.gpu


class _Grid:
    """Lock-step stand-in for the process group: rank threads meet in gather()."""

    def __init__(self, world):
        self.world = world
        self.turn = threading.Lock()
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = [0] * world

    def gather(self, rank, t):
        self.slots[rank] = t
        self.calls[rank] += 1
        self._meet()
        assert len(set(self.calls)) == 1, "ranks disagree on the number of collectives"
        assert len({(tuple(s.shape), s.dtype) for s in self.slots}) == 1, "ranks disagree on a collective's shape"
        out = torch.stack(self.slots)
        self._meet()                            # nobody overwrites a slot before all have read it
        return out, ShardedRetriever._Done()

    def _meet(self):
        self.turn.release()
        try:
            self.barrier.wait(timeout=100)
        finally:
            self.turn.acquire()


def _run_grid(table, queries, k, world, item_shards, chunk, exchange=True):
    grid = _Grid(world)
    results, errors = [None] * world, []

    def rank_main(rank):
        grid.turn.acquire()
        try:
            shard_index, S, qgroup, G = grid_layout(world, rank, item_shards)
            lo, hi = shard_bounds(table.shape[0], S, shard_index)
            r = ShardedRetriever(table[lo:hi], id_offset=lo, chunk=chunk, exchange_thresholds=exchange)
            r.world, r.rank = world, rank                       # a one-process retriever made to believe in the grid
            r.shard_index, r.S, r.qgroup, r.G = shard_index, S, qgroup, G
            if exchange and S > 1:
                r.seeded = (r._hip_sample, r._hip_seeded, r._hip_exact)
            r._all_gather = lambda t, rank=rank: grid.gather(rank, t)
            ids, sc = r.search(queries, k)
            results[rank] = (ids, sc, dict(r.last_stats))
        except BaseException as e:                               # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))
            grid.barrier.abort()
        finally:
            grid.turn.release()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(r is not None for r in results)
    return results


CASES = [
    # rows, n_q, k, d, world, item_shards, chunk
    (60_003, 262_144 + 77, 10, 64, 8, 8, 262_144),     # eight ranges, sampled + exchanged, ragged second chunk
    (60_003, 300_011, 10, 64, 4, 2, 262_144),          # 2 ranges x 2 query groups, ragged last group
    (25, 1000, 10, 64, 4, 4, 512),                     # ranges shorter than k (7 rows), last range 4 rows
    (5, 300, 10, 64, 8, 8, 128),                       # fewer rows than ranks: empty trailing ranges, fewer rows than k
    (40_001, 1, 10, 64, 2, 2, 128),                    # one query
    (40_001, 20_000, 1, 64, 3, 3, 8192),               # k = 1
    (40_001, 20_000, 16, 64, 3, 3, 8192),              # k = 16
    (30_001, 262_144, 10, 128, 4, 4, 262_144),         # d = 128, sampled + exchanged
    (40_001, 5000, 50, 64, 2, 2, 2048),                # the miner's k on ranges
]


@pytest.mark.parametrize("rows,n_q,k,d,world,item_shards,chunk", CASES)
def test_rank_grid_in_one_process(rows, n_q, k, d, world, item_shards, chunk):
    from oracle import c_oracle as C
    g = torch.Generator().manual_seed(rows + n_q + k)
    table = torch.randn(rows, d, generator=g)
    q = torch.randn(n_q, d, generator=g)
    tq, qq = table.cuda(), q.cuda()
    results = _run_grid(tq, qq, k, world, item_shards, chunk)
    if rows >= k:
        want_ids, want_sc = ops.score_topk(qq, ops.PreparedTable(tq), k)
    else:                                                   # fewer rows than k: the oracle's padding (-1 / -inf)
        o_ids, o_sc = C.score_topk(q.numpy(), table.numpy(), k)
        want_ids, want_sc = torch.from_numpy(o_ids).cuda(), torch.from_numpy(o_sc).cuda()
    for rank, (ids, sc, stats) in enumerate(results):
        assert torch.equal(ids, want_ids), f"rank {rank} ids"
        assert torch.equal(sc, want_sc), f"rank {rank} scores"
    sl = slice(0, min(n_q, 16))
    o_ids, o_sc = C.score_topk(q[sl].numpy(), table.numpy(), k)
    assert np.array_equal(results[0][0][sl].cpu().numpy(), o_ids) and np.array_equal(results[0][1][sl].cpu().numpy(), o_sc)
    if n_q >= 262_144:
        assert results[0][2]["uncertain"] < n_q // 50
