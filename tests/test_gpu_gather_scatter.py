"""GPU parity: HIP gather / scatter-add through the C ABI vs the C oracle (bit-exact)."""
import numpy as np
import pytest
import torch

from generative_recommenders_amd import hmx, ops
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _table(rows, d, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(rows, d, generator=g)
    t[0] = 0
    return t


@pytest.mark.parametrize("rows,d,shape", [(1000, 64, (37, 100)), (257, 60, (5, 7)), (100, 16, (3,)), (50, 128, (2, 3, 4)),
                                          (64, 6, (9, 5)), (4001, 64, (0,)), (300, 64, (1,))])
def test_gather_bit_exact(rows, d, shape):
    t = _table(rows, d, 1)
    ids = torch.randint(0, rows, shape, generator=torch.Generator().manual_seed(2))
    out = ops.gather_rows(t.to(DEV), ids.to(DEV))
    assert out.shape == shape + (d,)
    assert np.array_equal(out.cpu().numpy(), C.gather_rows(t.numpy(), ids.numpy()))


@pytest.mark.parametrize("rows,d,shape", [(1000, 64, (37, 100)), (50, 128, (2, 3, 4)), (100, 16, (3,)), (300, 20, (11,))])
def test_gather_bf16_is_the_rounded_exact_copy(rows, d, shape):
    """bf16 output = round-to-nearest-even of the gathered fp32 rows (PyTorch's CPU conversion as the rounding
    reference), bit for bit; out-of-range ids read as zero rows."""
    t = _table(rows, d, 11)
    ids = torch.randint(-2, rows + 2, shape, generator=torch.Generator().manual_seed(12))
    out = ops.gather_rows_bf16(t.to(DEV), ids.to(DEV))
    assert out.dtype == torch.bfloat16 and out.shape == shape + (d,)
    want = torch.from_numpy(C.gather_rows(t.numpy(), ids.numpy())).bfloat16()
    assert torch.equal(out.cpu().view(torch.int16), want.view(torch.int16))


def test_gather_out_of_range_ids_read_zero():
    t = _table(10, 64, 3)
    ids = torch.tensor([1, -1, 10, 9, 2 ** 40])
    out = ops.gather_rows(t.to(DEV), ids.to(DEV)).cpu()
    assert np.array_equal(out.numpy(), C.gather_rows(t.numpy(), ids.numpy()))
    assert not out[1].any() and not out[2].any() and not out[4].any()


def test_gather_routed_bit_exact():
    it, ut = _table(500, 64, 4), _table(80, 64, 5)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 80, (16, 33), generator=g)
    tt = torch.randint(0, 4, (16, 33), generator=g)
    out = ops.gather_routed(it.to(DEV), ut.to(DEV), ids.to(DEV), tt.to(DEV))
    assert np.array_equal(out.cpu().numpy(), C.gather_routed(it.numpy(), ut.numpy(), ids.numpy(), tt.numpy()))


@pytest.mark.parametrize("rows,d,n,hot", [(1000, 64, 5000, False), (50, 64, 20000, True), (300, 60, 777, False),
                                          (64, 6, 100, True), (2000, 128, 4096, False), (10, 64, 1, False)])
def test_scatter_add_bit_exact(rows, d, n, hot):
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, rows, (n,), generator=g)
    if hot:
        ids[: n // 2] = 3                                  # one very long run
    grad = torch.randn(n, d, generator=g)
    got = ops.scatter_add_rows(grad.to(DEV), ids.to(DEV), rows, padding_idx=0).cpu().numpy()
    want = C.scatter_add_rows(grad.numpy(), ids.numpy(), rows, padding_idx=0)
    assert np.array_equal(got, want)
    assert not got[0].any()


@pytest.mark.parametrize("n,distinct,d", [(200_000, 5, 64), (70_000, 1, 64), (5000, 3, 16), (1537, 2, 128), (513, 1, 64),
                                          (100_000, 2, 16), (60_000, 3, 128), (50_000, 1, 32), (40_000, 2, 256),
                                          (20_000, 1, 60), (4608 + 513, 1, 64), (4608 + 511, 1, 64), (300_000, 40, 64)])
def test_scatter_add_extreme_skew_bit_exact(n, distinct, d):
    """Runs of tens of thousands of equal ids (every user mining the same hard negatives): pieces of 512 entries
    are summed in parallel and joined in order (runs of more than eight further pieces by a whole wave, through LDS;
    shorter ones by one lane group; widths that are not a multiple of four always by one group); the oracle follows
    the same association."""
    g = torch.Generator().manual_seed(n)
    ids = torch.randint(1, distinct + 1, (n,), generator=g) * 7
    grad = torch.randn(n, d, generator=g)
    got = ops.scatter_add_rows(grad.to(DEV), ids.to(DEV), 100, padding_idx=0).cpu().numpy()
    want = C.scatter_add_rows(grad.numpy(), ids.numpy(), 100, padding_idx=0)
    assert np.array_equal(got, want)
    np.testing.assert_allclose(got[7], grad[ids == 7].double().sum(0).float().numpy(), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("rows,n,d,dup", [(3_000_000, 600_000, 16, 0.5), (70_000, 300_000, 8, 0.0), (200, 270_000, 4, 0.0),
                                          (4_000_001, 280_000, 64, 0.3), (65_536, 4097, 12, 0.0),
                                          (200_000, 300_000, 8, 0.4), (100_000, 5000, 16, 0.3), (262_143, 270_000, 4, 0.2)])
def test_scatter_add_multi_pass_sort_bit_exact(rows, n, d, dup):
    """Enough entries for the sort's scanned tile prefixes (more than 64 tiles of 4096) and key widths of one, two
    and three digits, including the 17- and 18-bit widths that are sorted as two 9-bit digits; `dup` of the entries repeat a few thousand ids, so equal keys sit in many tiles and any loss of
    their original order would change the fp32 sums."""
    g = torch.Generator().manual_seed(rows % 1000 + n)
    ids = torch.randint(0, rows, (n,), generator=g)
    n_dup = int(n * dup)
    if n_dup:
        pool = torch.randint(1, rows, (3000,), generator=g)
        where = torch.randperm(n, generator=g)[:n_dup]
        ids[where] = pool[torch.randint(0, 3000, (n_dup,), generator=g)]
    grad = torch.randn(n, d, generator=g)
    got = ops.scatter_add_rows(grad.to(DEV), ids.to(DEV), rows, padding_idx=0).cpu().numpy()
    want = C.scatter_add_rows(grad.numpy(), ids.numpy(), rows, padding_idx=0)
    assert np.array_equal(got, want)


def test_scatter_add_is_deterministic_and_accumulates():
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(0, 40, (30000,), generator=g).to(DEV)
    grad = torch.randn(30000, 64, generator=g).to(DEV)
    a = ops.scatter_add_rows(grad, ids, 40)
    b = ops.scatter_add_rows(grad, ids, 40)
    assert torch.equal(a, b)
    c = ops.scatter_add_rows(grad, ids, 40, out=a.clone())
    assert torch.equal(c, a + a)


def test_embedding_autograd_matches_torch_cpu():
    torch.manual_seed(9)
    emb = torch.nn.Embedding(200, 64, padding_idx=0)
    ids = torch.randint(0, 200, (8, 50))
    ids[:, :5] = 0
    out_ref = emb(ids)
    gout = torch.randn_like(out_ref)
    out_ref.backward(gout)
    w = emb.weight.detach().to(DEV).requires_grad_()
    out = ops.gather_rows(w, ids.to(DEV), 0)
    out.backward(gout.to(DEV))
    assert torch.equal(out.detach().cpu(), out_ref.detach())
    torch.testing.assert_close(w.grad.cpu(), emb.weight.grad, rtol=1e-6, atol=1e-6)
    assert not w.grad[0].any()


def test_routed_autograd():
    it = _table(100, 64, 10).to(DEV).requires_grad_()
    ut = _table(30, 64, 11).to(DEV).requires_grad_()
    g = torch.Generator().manual_seed(12)
    ids = torch.randint(0, 30, (6, 20), generator=g)
    tt = torch.randint(0, 3, (6, 20), generator=g)
    out = ops.gather_routed(it, ut, ids.to(DEV), tt.to(DEV))
    gout = torch.randn(6, 20, 64, generator=g)
    out.backward(gout.to(DEV))
    want_i = C.scatter_add_rows(gout.numpy(), ids.numpy(), 100, 0, tt.numpy(), 1)
    want_u = C.scatter_add_rows(gout.numpy(), ids.numpy(), 30, 0, tt.numpy(), 2)
    assert np.array_equal(it.grad.cpu().numpy(), want_i)
    assert np.array_equal(ut.grad.cpu().numpy(), want_u)


def test_large_gather_roundtrip_property():
    """Full-size property (4M x 64 table): gather(ids) then scatter-add of ones counts every id."""
    rows, d, n = 4_000_001, 64, 1 << 20
    t = torch.randn(rows, d, device=DEV)
    ids = torch.randint(1, rows, (n,), device=DEV)
    out = ops.gather_rows(t, ids)
    assert torch.equal(out, t[ids])
    cnt = ops.scatter_add_rows(torch.ones(n, d, device=DEV), ids, rows)
    assert torch.equal(cnt[:, 0], torch.bincount(ids, minlength=rows).float())


@pytest.mark.parametrize("rows,d", [(1, 64), (1000, 64), (100_003, 64), (777, 128), (513, 6), (300, 1), (64, 260), (0, 64)])
def test_touched_rows_of_a_dense_gradient(rows, d):
    """Rows holding any non-zero value (NaN and -inf included, -0.0 not), as torch's (g != 0).any(1).nonzero()."""
    from generative_recommenders_amd import ops
    g = torch.Generator().manual_seed(rows + d)
    x = torch.zeros(rows, d)
    if rows:
        hit = torch.randperm(rows, generator=g)[:max(1, rows // 7)]
        x[hit, torch.randint(0, d, (hit.numel(),), generator=g)] = 1e-30           # one tiny value somewhere in the row
        x[hit[0], d - 1] = float("nan")
        x[hit[-1], 0] = float("-inf")
        x[(hit[0] + 1) % rows, d // 2] += -0.0
    want = (x != 0).any(dim=1).nonzero().flatten()
    got = ops.touched_rows(x.cuda())
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    if rows > 2:                                   # a view that starts one row in (offset base pointer)
        assert torch.equal(ops.touched_rows(x.cuda()[1:]).cpu(), (x[1:] != 0).any(dim=1).nonzero().flatten())
