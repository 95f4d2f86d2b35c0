"""ShardedRetriever's own search code for whole rank grids (up to eight ranges) on ONE GPU in ONE process: every rank is
a thread, the collectives are hand-overs between the threads, and only one thread runs at a time (the library's
workspace is shared).  Odd shapes: ranges shorter than k, empty trailing ranges, ragged last query group, one query,
k = 1 and k = 16, d = 128.  The result must equal the unsharded kernel's (itself pinned to the C oracle), and the
oracle's on a slice.  GradientSync's exchange runs the same way for 4 and 8 ranks."""
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

    def exchange(self, rank, send, S):
        """all_to_all within a query group: block j of the result is what the group's j-th rank sent in its block
        number (rank % S)."""
        self.slots[rank] = send
        self.calls[rank] += 1
        self._meet()
        assert len(set(self.calls)) == 1, "ranks disagree on the number of collectives"
        assert len({(tuple(s.shape), s.dtype) for s in self.slots}) == 1, "ranks disagree on a collective's shape"
        g, me = rank // S, rank % S
        out = torch.stack([self.slots[g * S + j][me] for j in range(S)])
        self._meet()
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
            r._group_exchange = lambda t, rank=rank, S=S: grid.exchange(rank, t, S)
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
    (60_003, 100_000, 10, 64, 2, 2, 131_072),          # launches cut into two item splits: still sampled + exchanged
    (60_003, 50_000, 10, 64, 3, 3, 65_536),            # five item splits: seeded from table prefixes, nothing to exchange
    (4_000_001, 1_000_000, 10, 64, 8, 8, 262_144),     # the full sharded retrieval: 4 M items in eight ranges, 1 M queries
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


class _FakeDist:
    """What parallel.py uses of torch.distributed, carried by a _Grid (the calling rank is the thread's)."""

    class ReduceOp:
        SUM = "sum"

    def __init__(self, grid):
        self.grid, self.local = grid, threading.local()

    def is_initialized(self):
        return True

    def get_world_size(self, group=None):
        return self.grid.world

    def get_backend(self, group=None):
        return "threads"

    def all_gather_into_tensor(self, out, t, group=None, async_op=False):
        got, done = self.grid.gather(self.local.rank, t)
        out.copy_(got.reshape(out.shape))
        return done

    def all_reduce(self, t, op=None, group=None, async_op=False):
        got, done = self.grid.gather(self.local.rank, t.clone())
        total = got[0].clone()
        for r in range(1, self.grid.world):            # rank order: every rank ends with the same bits
            total += got[r]
        t.copy_(total)
        return done


@pytest.mark.parametrize("world", [4, 8])
def test_gradient_sync_on_a_rank_grid_in_one_process(world, monkeypatch):
    """GradientSync's own code for 4 and 8 ranks on one GPU (ranks are threads, collectives are hand-overs): the sparse
    (row id, row) exchange on the HIP row gather + deterministic scatter-add equals the dense all-reduce, every rank
    ends with identical bits, and ranks that touched no row, one row, or the same rows as a neighbour take part."""
    from generative_recommenders_amd import parallel
    grid = _Grid(world)
    fake = _FakeDist(grid)
    monkeypatch.setattr(parallel, "dist", fake)
    g = torch.Generator().manual_seed(world)
    rows, d = 20_000, 64
    table0 = torch.randn(rows, d, generator=g).cuda()
    w0 = torch.randn(d, d, generator=g).cuda()
    counts = [3000, 0, 1, 3000, 17, 2500, 1, 999][:world]
    id_sets = [torch.randint(0, rows, (n,), generator=g).cuda() for n in counts]
    id_sets[3] = id_sets[0].clone()                         # two ranks touching exactly the same rows
    results, errors = [None] * world, []

    def rank_main(rank):
        grid.turn.acquire()
        fake.local.rank = rank
        try:
            out = {}
            for name, sparse in (("sparse", True), ("dense", False)):
                table, w = torch.nn.Parameter(table0.clone()), torch.nn.Parameter(w0.clone())
                ids = id_sets[rank]
                if ids.numel():
                    loss = ((ops.gather_rows(table, ids) @ w) ** 2).mean() * (rank + 1)
                    loss.backward()
                sync = parallel.GradientSync([table, w], big_bytes=1 << 20, sparse_tables=sparse)
                sync()
                out[name] = (table.grad.clone(), w.grad.clone())
            results[rank] = out
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
    for name in ("sparse", "dense"):
        for r in range(1, world):
            assert torch.equal(results[r][name][0], results[0][name][0]), (name, r)
            assert torch.equal(results[r][name][1], results[0][name][1]), (name, r)
    torch.testing.assert_close(results[0]["sparse"][0], results[0]["dense"][0], rtol=1e-5, atol=1e-7)
    assert torch.equal(results[0]["sparse"][1], results[0]["dense"][1])
    touched = torch.cat(id_sets).unique()
    assert int((results[0]["sparse"][0].abs().sum(1) > 0).sum()) == touched.numel()


def test_data_parallel_train_step_equals_the_average_of_the_ranks_own_steps(monkeypatch):
    """Two ranks (threads) run Hydra's train step on their own batches and meet in GradientSync: every rank must end
    with the mean of the gradients that the two batches give on their own (each rank draws its negatives from its own
    batch, as one process of the reference does), tables included."""
    from generative_recommenders_amd import hydra as H
    from generative_recommenders_amd import parallel, synth
    world, U, I, d, L, B = 2, 300, 5000, 64, 20, 6
    batches = [synth.train_batch(B, L, U, I, "cuda", seed=90 + r) for r in range(world)]
    feat = [[] for _ in range(B)]

    def step(model, bt, seed):
        torch.manual_seed(seed)                                  # the step's draws (negatives) follow this seed
        loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"],
                     bt["act"], bt["dwell"])
        loss.backward()
        return {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}

    alone = [step(H.build_model(U, I, d=d, L=L, dropout=0.0, device="cuda", seed=5).train(), batches[r], 100 + r)
             for r in range(world)]
    grid = _Grid(world)
    fake = _FakeDist(grid)
    monkeypatch.setattr(parallel, "dist", fake)
    results, errors = [None] * world, []

    def rank_main(rank):
        grid.turn.acquire()
        fake.local.rank = rank
        try:
            model = H.build_model(U, I, d=d, L=L, dropout=0.0, device="cuda", seed=5).train()
            grads = step(model, batches[rank], 100 + rank)
            del grads
            parallel.GradientSync(model, big_bytes=1 << 18)()    # both tables take the sparse row exchange
            results[rank] = {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}
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
    keys = set(alone[0]) | set(alone[1])
    # parameters that no step reaches (a module the model never calls) take part in the exchange as zeros
    assert set(results[0]) == set(results[1]) and keys <= set(results[0])
    for k in set(results[0]) - keys:
        assert float(results[0][k].abs().sum()) == 0.0, k
    for k in keys:
        assert torch.equal(results[0][k], results[1][k]), k
        zero = torch.zeros_like(results[0][k])
        want = (alone[0].get(k, zero) + alone[1].get(k, zero)) * 0.5
        torch.testing.assert_close(results[0][k], want, rtol=1e-5, atol=1e-7, msg=k)
