"""Two ranks sharing one GPU (gloo): the sparse table-gradient exchange of GradientSync on the HIP row gather and
deterministic scatter-add must reproduce the dense all-reduce, and leave both ranks with identical bits."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    from generative_recommenders_amd import ops
    from generative_recommenders_amd.parallel import GradientSync
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(3)
    table = torch.nn.Parameter(torch.randn(20_000, 64, generator=g).cuda())
    w = torch.nn.Parameter(torch.randn(64, 64, generator=g).cuda())
    ids = torch.randint(0, 20_000, (2, 3000), generator=g)[rank].cuda()      # overlapping and disjoint rows
    results = {}
    for name, sparse in (("sparse", True), ("dense", False)):
        table.grad = w.grad = None
        loss = ((ops.gather_rows(table, ids) @ w) ** 2).mean() * (rank + 1)
        loss.backward()
        GradientSync([table, w], big_bytes=1 << 20, sparse_tables=sparse)()
        results[name] = (table.grad.clone().cpu(), w.grad.clone().cpu())
    torch.save(results, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_sparse_exchange_equals_dense_all_reduce(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt", weights_only=True) for r in range(2))
    for name in ("sparse", "dense"):
        assert torch.equal(r0[name][0], r1[name][0]) and torch.equal(r0[name][1], r1[name][1]), name
    torch.testing.assert_close(r0["sparse"][0], r0["dense"][0], rtol=1e-6, atol=1e-7)
    assert torch.equal(r0["sparse"][1], r0["dense"][1])
    assert int((r0["sparse"][0].abs().sum(1) > 0).sum()) > 3000          # rows of both ranks are present
