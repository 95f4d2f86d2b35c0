"""The scatter-add entry point that allocates nothing (its scratch is a fixed array inside the library) must give
the bits of the workspace-taking entry point, also when a call follows a larger one on the same GPU array."""
import numpy as np
import pytest
import torch

from generative_recommenders_amd import hmx
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(n, n_rows, d, seed, with_types=False):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, n_rows, (n,), generator=g)
    ids[::7] = ids[0]                                   # a long run of one id
    rows = torch.randn(n, d, generator=g)
    tt = torch.randint(0, 3, (n,), generator=g) if with_types else None
    L = hmx.lib()
    s = hmx.current_stream()
    ids_d, rows_d = ids.to(DEV), rows.to(DEV)
    tt_d = tt.to(DEV) if tt is not None else None
    want = torch.zeros(n_rows, d, device=DEV)
    ws = torch.empty(L.hmx_scatter_add_rows_workspace_bytes(n, n_rows, d), dtype=torch.uint8, device=DEV)
    hmx.check(L.hmx_scatter_add_rows_f32(hmx.ptr(rows_d), hmx.ptr(ids_d), hmx.ptr(tt_d), 1, n, d, hmx.ptr(want), n_rows, 0,
                                         None, hmx.ptr(ws), ws.numel(), s), "hmx_scatter_add_rows_f32")
    got = torch.zeros(n_rows, d, device=DEV)
    hmx.check(L.hmx_scatter_add_rows_f32_static(hmx.ptr(rows_d), hmx.ptr(ids_d), hmx.ptr(tt_d), 1, n, d, hmx.ptr(got),
                                                n_rows, 0, s), "hmx_scatter_add_rows_f32_static")
    torch.cuda.synchronize()
    return got.cpu(), want.cpu(), ids, rows, tt


def test_static_scratch_matches_workspace_entry_point_twice_in_a_row():
    got, want, ids, rows, _ = _both(200_000, 50_001, 64, 1)
    assert torch.equal(got, want)
    assert np.array_equal(got.numpy(), C.scatter_add_rows(rows.numpy(), ids.numpy(), 50_001, 0))
    got2, want2, ids2, rows2, tt2 = _both(3_001, 977, 64, 2, with_types=True)    # smaller input, same GPU array
    assert torch.equal(got2, want2)
    keep = tt2.numpy() == 1
    assert np.array_equal(got2.numpy(), C.scatter_add_rows(rows2.numpy()[keep], ids2.numpy()[keep], 977, 0))
    got3, want3, *_ = _both(10, 5, 20, 3)                                         # 4-byte lanes (d % 4 == 0 but tiny)
    assert torch.equal(got3, want3)


def test_static_scratch_rejects_oversize_input_and_enqueues_nothing():
    L = hmx.lib()
    n, n_rows, d = 4_000_000, 4_000_001, 64
    assert L.hmx_scatter_add_rows_workspace_bytes(n, n_rows, d) > L.hmx_scatter_add_rows_static_scratch_bytes()
    one = torch.zeros(1, device=DEV)
    rc = L.hmx_scatter_add_rows_f32_static(hmx.ptr(one), hmx.ptr(one), None, 0, n, d, hmx.ptr(one), n_rows, 0,
                                           hmx.current_stream())
    assert rc == -4 and b"fixed scratch" in L.hmx_error_string(rc)
    torch.cuda.synchronize()
