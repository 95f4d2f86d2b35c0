"""world_size-2 gloo test of the sharded-retrieval exchange logic (no GPU): the local scorer and the merge are
the oracle's, injected into the same ShardedRetriever the GPU path uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pytest

from generative_recommenders_amd.retrieval import ShardedRetriever, grid_layout, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, item_shards=None, n_q=70, exchange=None):
    from oracle import c_oracle as C
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((2501, 16)).astype(np.float32)
    q = torch.from_numpy(rng.standard_normal((n_q, 16)).astype(np.float32))
    shard_index, S, _, _ = grid_layout(world, rank, item_shards)
    lo, hi = shard_bounds(table.shape[0], S, shard_index)
    shard = table[lo:hi]

    def local(queries, k):
        ids, sc = C.score_topk(queries.numpy(), shard, k, id_offset=lo)
        return torch.from_numpy(ids), torch.from_numpy(sc)

    def merge(ci, cs):
        ids, sc = C.topk_merge(ci.numpy(), cs.numpy())
        return torch.from_numpy(ids), torch.from_numpy(sc)

    # stand-ins for the threshold-exchange kernels: thresholds sampled from every third row of the shard, a pass
    # that drops every row at or below its start threshold (bound = that threshold), and the exact local scorer
    def sample(queries, k):
        sub = shard[::3]
        if sub.shape[0] < k:
            return torch.full((queries.shape[0],), float("-inf"))
        srt = np.sort(queries.numpy() @ sub.T, axis=1)
        if exchange == "union":          # the three best sampled scores of the range: the k-th of all ranges' is used
            return torch.from_numpy(srt[:, -3:].copy())
        thr = torch.from_numpy(srt[:, -k].copy())
        if exchange == "too_high":       # every seventh query starts above all scores: the re-run must repair it
            thr[::7] = 1e9
        return thr

    def seeded_topk(queries, k, thr):
        ids, sc = local(queries, k)
        drop = sc <= thr.unsqueeze(1)
        ids[drop], sc[drop] = -1, float("-inf")
        return ids, sc, thr.clone(), torch.zeros(queries.shape[0], dtype=torch.int32)

    seeded = (sample, seeded_topk, local) if exchange else None
    r = ShardedRetriever(None, id_offset=lo, local_topk=local, merge=merge, chunk=32, item_shards=item_shards,
                         seeded=seeded)
    ids, sc = r.search(q, 10)
    if exchange == "too_high" and S > 1:
        assert r.last_stats["uncertain"] >= -(-n_q // 7)
    elif exchange and S > 1:
        assert r.last_stats["uncertain"] == 0
    want_ids, want_sc = C.score_topk(q.numpy(), table, 10)
    ok = np.array_equal(ids.numpy(), want_ids) and np.array_equal(sc.numpy(), want_sc)
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.destroy_process_group()


def test_two_rank_sharded_retrieval_equals_unsharded(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]


@pytest.mark.parametrize("item_shards,n_q", [(2, 70), (1, 70), (2, 67), (4, 5)])
def test_four_rank_grid_equals_unsharded(tmp_path, item_shards, n_q):
    """4 ranks as S item shards x 4/S query groups, incl. a query count the groups do not divide."""
    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), item_shards, n_q), nprocs=4, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(4)] == ["1"] * 4


@pytest.mark.parametrize("item_shards,n_q,mode", [(2, 70, "plain"), (4, 67, "plain"), (2, 67, "too_high"),
                                                  (4, 70, "too_high"), (1, 70, "plain"), (4, 70, "union")])
def test_threshold_exchange_equals_unsharded(tmp_path, item_shards, n_q, mode):
    """The ranges start from the largest of their sampled thresholds, local lists may be short, completeness is
    decided after the merge; start values above every score force the exact re-run, which must repair them."""
    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), item_shards, n_q, mode), nprocs=4, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(4)] == ["1"] * 4


def test_grid_layout():
    assert [grid_layout(8, r, 2)[:3:2] for r in range(8)] == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3)]
    assert grid_layout(8, 5, None) == (5, 8, 0, 1)
    with pytest.raises(ValueError):
        grid_layout(8, 0, 3)


def test_shard_bounds_cover_and_do_not_overlap():
    for n, w in ((4_000_001, 8), (10, 3), (5, 8), (1, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
