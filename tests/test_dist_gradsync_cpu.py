"""world_size-2 gloo test of the data-parallel gradient exchange: two ranks on two half batches must end with the
gradient of one process on the whole batch (mean-reduced loss), for in-place ("big") and packed ("small") tensors,
including a parameter that received no gradient on one rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generative_recommenders_amd.parallel import GradientSync, broadcast_parameters


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Toy(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.table = torch.nn.Parameter(torch.randn(500, 8, generator=g))   # "big": reduced in place
        self.table2 = torch.nn.Parameter(torch.randn(600, 8, generator=g))  # a second one: both share the count exchange
        self.w = torch.nn.Parameter(torch.randn(8, 8, generator=g))
        self.b = torch.nn.Parameter(torch.randn(8, generator=g))
        self.unused_on_rank1 = torch.nn.Parameter(torch.randn(3, generator=g))

    def forward(self, ids, use_extra):
        h = (self.table[ids] + self.table2[ids // 2 + 7]) @ self.w + self.b
        loss = (h * h).mean()
        if use_extra:
            loss = loss + self.unused_on_rank1.sum() * 0.5
        return loss


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randint(0, 500, (64,), generator=g)


def _torch_row_ops(with_flags=False):
    """CPU stand-ins for the HIP gather / scatter-add (same contract: ids < 0 give zero rows / are ignored)."""
    def gather(table, ids):
        return table[ids.clamp_min(0)] * (ids >= 0).unsqueeze(1)

    def scatter_add(rows, ids, out):
        keep = ids >= 0
        out.index_add_(0, ids[keep], rows[keep])

    def touched(g):
        return (g != 0).any(dim=1).nonzero().flatten()

    def touched_flags(g):
        return (g != 0).any(dim=1).to(torch.uint8)
    return (gather, scatter_add, touched, touched_flags) if with_flags else (gather, scatter_add)


def _worker(rank, world, port, out_dir, sparse=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _Toy(seed=100 + rank)           # different init per rank: broadcast must fix that
    broadcast_parameters(model, src=0)
    sync = GradientSync(model, big_bytes=500 * 8 * 4, sparse_tables=bool(sparse),
                        row_ops=_torch_row_ops(with_flags=sparse == "flags") if sparse else None)
    assert [tuple(p.shape) for p in sync.big] == [(500, 8), (600, 8)] and len(sync.small) == 3
    ids = _data()[rank * 32:(rank + 1) * 32]
    for _ in range(2):                      # second pass reuses the flat bucket
        model.zero_grad(set_to_none=True)
        model(ids, use_extra=(rank == 0)).backward()
        sync()
    torch.save({k: p.grad for k, p in model.named_parameters()}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("sparse", [False, True, "flags"])
def test_two_rank_gradients_equal_whole_batch(tmp_path, sparse):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), sparse), nprocs=2, join=True)
    ref = _Toy(seed=100)
    ids = _data()
    # mean over ranks of per-rank mean losses == mean over the whole batch (equal halves); the extra term is on
    # rank 0 only, so its averaged gradient is half of 0.5
    (0.5 * (ref(ids[:32], True) + ref(ids[32:], False))).backward()
    for r in range(2):
        got = torch.load(tmp_path / f"g{r}.pt", weights_only=True)
        for k, p in ref.named_parameters():
            assert torch.allclose(got[k], p.grad, rtol=1e-5, atol=1e-6), k
    a, b = (torch.load(tmp_path / f"g{r}.pt", weights_only=True) for r in range(2))
    assert all(torch.equal(a[k], b[k]) for k in a)          # every rank ends with the same bits


def test_single_process_is_a_no_op():
    m = _Toy(seed=1)
    m(_data(), True).backward()
    before = m.w.grad.clone()
    GradientSync(m)()
    assert torch.equal(m.w.grad, before)
