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


def _worker(rank, world, port, out_dir, item_shards):
    from oracle import c_oracle as C
    from generative_recommenders_amd.retrieval import ShardedRetriever, grid_layout, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((40_001, 64)).astype(np.float32)
    q = rng.standard_normal((333, 64)).astype(np.float32)
    shard_index, S, _, _ = grid_layout(world, rank, item_shards)
    lo, hi = shard_bounds(table.shape[0], S, shard_index)
    r = ShardedRetriever(torch.from_numpy(table[lo:hi]).cuda(), id_offset=lo, chunk=128, item_shards=item_shards)
    ids, sc = r.search(torch.from_numpy(q).cuda(), 10)
    want_ids, want_sc = C.score_topk(q, table, 10)
    ok = np.array_equal(ids.cpu().numpy(), want_ids) and np.array_equal(sc.cpu().numpy(), want_sc)
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("item_shards", [2, 1])
def test_two_ranks_on_one_gpu(tmp_path, item_shards):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), item_shards), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]
