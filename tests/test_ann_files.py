"""File formats of the retrieval hand-off (dataset.py:653-666, infer.py:145-159) and the file-level exact top-K."""
import struct

import numpy as np
import pytest
import torch

from generative_recommenders_amd import ann_files as A


def _case(tmp_path, n=300, d=16, q=17, seed=0):
    rng = np.random.default_rng(seed)
    table = rng.standard_normal((n, d)).astype(np.float32)
    ids = rng.permutation(np.arange(10_000_000_000, 10_000_000_000 + n, dtype=np.uint64))
    queries = rng.standard_normal((q, d)).astype(np.float32)
    A.save_emb(table, tmp_path / "embedding.fbin")
    A.save_emb(ids.reshape(-1, 1), tmp_path / "id.u64bin")
    A.save_emb(queries, tmp_path / "query.fbin")
    return table, ids, queries


def test_save_emb_bytes_are_header_then_raw_rows(tmp_path):
    emb = np.arange(12, dtype=np.float32).reshape(3, 4)
    A.save_emb(emb, tmp_path / "e.fbin")
    raw = (tmp_path / "e.fbin").read_bytes()
    assert raw == struct.pack("II", 3, 4) + emb.tobytes()
    assert np.array_equal(A.load_fbin(tmp_path / "e.fbin"), emb)
    ids = np.array([[7], [2 ** 40 + 5]], dtype=np.uint64)
    A.save_emb(ids, tmp_path / "id.u64bin")
    assert (tmp_path / "id.u64bin").read_bytes() == struct.pack("II", 2, 1) + ids.tobytes()
    assert np.array_equal(A.load_u64bin(tmp_path / "id.u64bin"), ids.reshape(-1))


def test_result_file_round_trip_matches_reader_layout(tmp_path):
    ids = np.array([[5, 6, 7], [2 ** 50, 1, 0]], dtype=np.uint64)
    A.write_result_ids(ids, tmp_path / "id100.u64bin")
    raw = (tmp_path / "id100.u64bin").read_bytes()
    # what infer.py:145-159 parses: two uint32, then n*k uint64
    n, k = struct.unpack("I", raw[:4])[0], struct.unpack("I", raw[4:8])[0]
    assert (n, k) == (2, 3)
    assert np.array_equal(np.frombuffer(raw[8:], dtype=np.uint64).reshape(n, k), ids)
    assert np.array_equal(A.read_result_ids(tmp_path / "id100.u64bin"), ids)


def test_truncated_or_inconsistent_files_raise(tmp_path):
    (tmp_path / "bad.fbin").write_bytes(struct.pack("II", 4, 4) + b"\0" * 10)
    with pytest.raises(ValueError):
        A.load_fbin(tmp_path / "bad.fbin")
    (tmp_path / "short.fbin").write_bytes(b"\0\0")
    with pytest.raises(ValueError):
        A.load_fbin(tmp_path / "short.fbin")
    with pytest.raises(ValueError):
        A.save_emb(np.zeros(3, dtype=np.float32), tmp_path / "x.fbin")


def _oracle_search(q, t, k):
    from oracle import c_oracle as C
    return torch.from_numpy(C.score_topk(q.numpy(), t.numpy(), k)[0])


def test_retrieve_files_with_injected_scorer(tmp_path):
    table, ids, queries = _case(tmp_path)
    A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id.u64bin", tmp_path / "query.fbin",
                     tmp_path / "id100.u64bin", query_ann_top_k=10, search=_oracle_search)
    got = A.read_result_ids(tmp_path / "id100.u64bin")
    want_rows = torch.topk(torch.from_numpy(queries) @ torch.from_numpy(table).T, 10, dim=1).indices.numpy()
    assert got.shape == (17, 10) and np.array_equal(got, ids[want_rows])


def test_retrieve_files_rejects_mismatched_inputs(tmp_path):
    table, ids, queries = _case(tmp_path)
    A.save_emb(ids[:-1].reshape(-1, 1), tmp_path / "id_short.u64bin")
    with pytest.raises(ValueError):
        A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id_short.u64bin", tmp_path / "query.fbin",
                         tmp_path / "r.u64bin", search=_oracle_search)
    A.save_emb(queries[:, :8], tmp_path / "q8.fbin")
    with pytest.raises(ValueError):
        A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id.u64bin", tmp_path / "q8.fbin",
                         tmp_path / "r.u64bin", search=_oracle_search)


def test_fewer_candidates_than_k_pads_with_last_id(tmp_path):
    table, ids, queries = _case(tmp_path, n=4)
    A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id.u64bin", tmp_path / "query.fbin",
                     tmp_path / "r.u64bin", query_ann_top_k=6, search=_oracle_search)
    got = A.read_result_ids(tmp_path / "r.u64bin")
    assert got.shape == (17, 6)
    assert all(set(row[:4]) == set(ids) and (row[4:] == row[3]).all() for row in got)


@pytest.mark.gpu
def test_retrieve_files_on_gpu_matches_oracle(tmp_path):
    from oracle import c_oracle as C
    table, ids, queries = _case(tmp_path, n=50_000, d=64, q=300, seed=3)
    A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id.u64bin", tmp_path / "query.fbin",
                     tmp_path / "id100.u64bin", query_ann_top_k=10)
    got = A.read_result_ids(tmp_path / "id100.u64bin")
    want_rows, _ = C.score_topk(queries, table, 10)
    assert np.array_equal(got, ids[want_rows])


@pytest.mark.gpu
def test_retrieve_files_on_gpu_large_ids_and_fewer_candidates_than_k(tmp_path):
    """uint64 ids above 2^63 survive the device-side row -> id mapping bit for bit, and with fewer candidates than k
    the tail repeats the last valid id (the reader has no "missing" value)."""
    from oracle import c_oracle as C
    rng = np.random.default_rng(5)
    table = rng.standard_normal((7, 32)).astype(np.float32)
    ids = (np.uint64(2 ** 63) + rng.permutation(7).astype(np.uint64) * np.uint64(3)).astype(np.uint64)
    ids[0] = np.uint64(2 ** 64 - 1)
    queries = rng.standard_normal((40, 32)).astype(np.float32)
    A.save_emb(table, tmp_path / "embedding.fbin")
    A.save_emb(ids.reshape(-1, 1), tmp_path / "id.u64bin")
    A.save_emb(queries, tmp_path / "query.fbin")
    rows = A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id.u64bin", tmp_path / "query.fbin",
                            tmp_path / "id100.u64bin", query_ann_top_k=10)
    got = A.read_result_ids(tmp_path / "id100.u64bin")
    want_rows, _ = C.score_topk(queries, table, 7)
    want_rows = np.concatenate([want_rows, np.repeat(want_rows[:, -1:], 3, axis=1)], axis=1)
    assert got.dtype == np.uint64 and got.shape == (40, 10)
    assert np.array_equal(rows, want_rows) and np.array_equal(got, ids[want_rows])


@pytest.mark.gpu
def test_candidate_table_generation_matches_reference_golden(tmp_path):
    """SharedEmbeddingModule.get_item_embeddings / save_item_embeddings / get_complete_embeddings on the fused
    gather+GELU kernel against outputs captured from the reference (embedding.py:613-735, ids-only)."""
    import _golden as G
    from generative_recommenders_amd import hydra as H
    s = G.load("g11_item_embs")
    args = H.make_args(d=16, L=10, device="cuda")
    m = H.Hydra(20, 60, {}, H.IDS_ONLY_FEAT_TYPES, args).cuda()
    se = m.shared_embeddings
    with torch.no_grad():
        se.item_emb.weight.copy_(G.t(s, "G11/item_w"))
    full = se.get_item_embeddings({})
    assert full.shape == (61, 16) and bool((full[0] == 0).all())
    G.assert_close(full.cpu(), G.t(s, "G11/full"), 2e-6, 2e-7, "full")
    seq = G.t(s, "G11/seq").cuda()
    G.assert_close(se.get_complete_embeddings(seq, None, fusion_style="late").cpu(), G.t(s, "G11/late"), 2e-6, 2e-7, "late")
    G.assert_close(se.get_complete_embeddings(seq, None, fusion_style="early").cpu(), G.t(s, "G11/early"), 0, 0, "early")
    # through the model-level hook that infer.py:328 calls (it forwards to the shared module's writer)
    m.save_item_emb(G.t(s, "G11/item_ids").tolist(), G.t(s, "G11/retrieval_ids").tolist(), {}, tmp_path)
    G.assert_close(torch.from_numpy(A.load_fbin(tmp_path / "embedding.fbin")), G.t(s, "G11/saved/embedding.fbin"),
                   2e-6, 2e-7, "embedding.fbin")
    assert np.array_equal(A.load_u64bin(tmp_path / "id.u64bin"), s["G11/saved/id.u64bin"].reshape(-1))
