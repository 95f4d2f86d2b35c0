"""Pin the C oracle: against the reference-captured top-K golden and against PyTorch's own
nn.Embedding forward/backward on CPU (the ATen ops the reference calls)."""
import numpy as np
import torch

import _golden as G
from oracle import c_oracle as C


def test_topk_matches_reference_golden():
    s, p = G.load("g9_topk"), "G9_topk"
    table, users = s[p + "/table"], s[p + "/users"]
    ids10, sc10 = C.score_topk(users, table, 10)
    assert np.array_equal(ids10, s[p + "/ids10"])
    np.testing.assert_allclose(sc10, s[p + "/val10"], rtol=1e-5, atol=1e-5)
    excl = np.concatenate([s[p + "/pos"], s[p + "/hist"]], axis=1)
    ids50, _ = C.score_topk(users, table, 50, exclude=excl)
    assert np.array_equal(ids50, s[p + "/ids50"])


def test_topk_tie_break_and_short_lists():
    table = np.zeros((6, 4), np.float32)
    table[:, 0] = [1, 3, 3, 2, 3, 0]
    q = np.array([[1, 0, 0, 0]], np.float32)
    ids, sc = C.score_topk(q, table, 3)
    assert ids.tolist() == [[1, 2, 4]] and sc.tolist() == [[3, 3, 3]]          # score desc, id asc
    ids, sc = C.score_topk(q, table, 4, exclude=np.array([[1, 2, 4, 0]]), id_offset=0)
    assert ids.tolist() == [[3, 5, -1, -1]] and sc[0, 2] == -np.inf
    ids, _ = C.score_topk(q, table, 2, id_offset=100)
    assert ids.tolist() == [[101, 102]]


def test_topk_merge_equals_unsharded():
    rng = np.random.default_rng(0)
    table = rng.standard_normal((1001, 8)).astype(np.float32)
    q = rng.standard_normal((5, 8)).astype(np.float32)
    full_ids, full_sc = C.score_topk(q, table, 10)
    parts = [C.score_topk(q, table[a:b], 10, id_offset=a) for a, b in ((0, 334), (334, 668), (668, 1001))]
    ids, sc = C.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(ids, full_ids) and np.array_equal(sc, full_sc)


def test_gather_and_scatter_match_torch_embedding():
    torch.manual_seed(0)
    emb = torch.nn.Embedding(50, 12, padding_idx=0)
    ids = torch.randint(0, 50, (7, 9))
    ids[0, :3] = 0
    ids[1, :] = 5                                        # heavy repeats
    out = emb(ids)
    assert np.array_equal(C.gather_rows(emb.weight.detach().numpy(), ids.numpy()), out.detach().numpy())
    g = torch.randn_like(out)
    out.backward(g)
    mine = C.scatter_add_rows(g.numpy(), ids.numpy(), 50, padding_idx=0)
    np.testing.assert_allclose(mine, emb.weight.grad.numpy(), rtol=1e-6, atol=1e-6)
    assert not mine[0].any()


def test_routed_gather():
    rng = np.random.default_rng(1)
    it, ut = rng.standard_normal((10, 4)).astype(np.float32), rng.standard_normal((6, 4)).astype(np.float32)
    ids = np.array([[3, 2, 9, 0, 7]])
    tt = np.array([[2, 1, 1, 0, 3]])
    out = C.gather_routed(it, ut, ids, tt)
    assert np.array_equal(out[0, 0], ut[3]) and np.array_equal(out[0, 1], it[2]) and np.array_equal(out[0, 2], it[9])
    assert not out[0, 3].any() and not out[0, 4].any()
