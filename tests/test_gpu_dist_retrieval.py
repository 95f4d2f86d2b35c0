"""Two ranks sharing one GPU (gloo carries the CUDA tensors): the HIP scoring + merge kernels behind the rank-grid
exchange of ShardedRetriever must give the oracle's ids and scores, bit for bit, for both grid shapes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, item_shards, n_q=333):
    from oracle import c_oracle as C
    from generative_recommenders_amd.retrieval import ShardedRetriever, grid_layout, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((40_001, 64)).astype(np.float32)
    q = rng.standard_normal((n_q, 64)).astype(np.float32)
    shard_index, S, _, _ = grid_layout(world, rank, item_shards)
    lo, hi = shard_bounds(table.shape[0], S, shard_index)
    big = n_q > 100_000          # launches of 262,144 queries: the shape whose start thresholds are sampled and exchanged
    r = ShardedRetriever(torch.from_numpy(table[lo:hi]).cuda(), id_offset=lo, chunk=262_144 if big else 128,
                         item_shards=item_shards)
    ids, sc = r.search(torch.from_numpy(q).cuda(), 10)
    if big:                      # oracle on a slice, the unsharded HIP kernel (itself pinned to the oracle) on everything
        from generative_recommenders_amd import ops
        sl = slice(5000, 5064)
        want_ids, want_sc = C.score_topk(q[sl], table, 10)
        ok = np.array_equal(ids[sl].cpu().numpy(), want_ids) and np.array_equal(sc[sl].cpu().numpy(), want_sc)
        all_ids, all_sc = ops.score_topk(torch.from_numpy(q).cuda(), ops.PreparedTable(torch.from_numpy(table).cuda()), 10)
        ok = ok and torch.equal(ids, all_ids) and torch.equal(sc, all_sc) and (S == 1 or r.last_stats["uncertain"] < n_q // 100)
    else:
        want_ids, want_sc = C.score_topk(q, table, 10)
        ok = np.array_equal(ids.cpu().numpy(), want_ids) and np.array_equal(sc.cpu().numpy(), want_sc)
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("item_shards,n_q", [(2, 333), (1, 333), (2, 300_011)])
def test_two_ranks_on_one_gpu(tmp_path, item_shards, n_q):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), item_shards, n_q), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]
