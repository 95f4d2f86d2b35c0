"""infer.py's hand-off in one flow on the drop-in model: user vectors from `predict` -> query.fbin (infer.py:384-402),
candidate files from `save_item_emb` (infer.py:328), the exact top-10 over them (in place of the ANN command of
infer.py:405-417) -> id100.u64bin -> `read_result_ids` (infer.py:420).

(1) The result file is exactly the C oracle's top-10 over the very files the flow wrote.  (2) Those files hold what
the CPU oracle model computes for the same weights and batch (encoder tolerance).  (3) Against the oracle's own
top-10 from ITS vectors the lists agree except where two candidates score within the vectors' rounding."""
import numpy as np
import pytest
import torch

from generative_recommenders_amd import ann_files as A
from generative_recommenders_amd import hydra as H
from generative_recommenders_amd import synth
from oracle import c_oracle as C
from oracle import hydra_oracle as O

pytestmark = pytest.mark.gpu


def test_predict_save_retrieve_read(tmp_path):
    users, items, d, L, B = 120, 2000, 32, 16, 40
    ref = O.build_model(users, items, d=d, L=L, dropout=0.0, seed=8).eval()
    m = H.build_model(users, items, d=d, L=L, dropout=0.0, device="cuda", seed=1)
    m.load_state_dict(ref.state_dict(), strict=True)
    m.eval()
    bt = synth.train_batch(B, L, users, items, "cpu", seed=81, pad_frac=0.4)
    feat = [[] for _ in range(B)]
    with torch.no_grad():
        q = m.predict(bt["seq"].cuda(), feat, bt["tok"].cuda(), bt["ts"].cuda(), bt["act"].cuda(), bt["dwell"].cuda())
        q_ref = ref.log2feats(bt["seq"], feat, bt["tok"], bt["ts"], bt["act"], bt["dwell"])[:, -1, :]
    A.save_emb(q.cpu().numpy().astype(np.float32), tmp_path / "query.fbin")
    item_ids = list(range(1, items + 1))
    retrieval_ids = [7_000_000_000 + 3 * i for i in range(items)]            # the reader's ids are not row numbers
    m.save_item_emb(item_ids, retrieval_ids, {}, tmp_path)
    A.retrieve_files(tmp_path / "embedding.fbin", tmp_path / "id.u64bin", tmp_path / "query.fbin",
                     tmp_path / "id100.u64bin", query_ann_top_k=10)
    got = A.read_result_ids(tmp_path / "id100.u64bin")
    # (1) exact over the written files
    table = A.load_fbin(tmp_path / "embedding.fbin")
    ids = A.load_u64bin(tmp_path / "id.u64bin")
    rows, _ = C.score_topk(A.load_fbin(tmp_path / "query.fbin"), table, 10)
    assert got.shape == (B, 10) and np.array_equal(got, ids[rows])
    assert np.array_equal(ids, np.array(retrieval_ids, dtype=np.uint64))
    # (2) the vectors are the oracle's
    with torch.no_grad():
        seq = torch.tensor(item_ids).unsqueeze(1)
        t_ref = ref.shared_embeddings.get_complete_embeddings(seq, [[[]] for _ in item_ids], include_user=False,
                                                              fusion_style='late').squeeze(1)
    torch.testing.assert_close(torch.from_numpy(table), t_ref, rtol=2e-6, atol=2e-7)
    err = (q.cpu() - q_ref).abs().max().item()
    assert err <= 2e-3 * q_ref.abs().max().item() + 1e-4, err
    # (3) the oracle's own lists
    rows_ref, sc_ref = C.score_topk(q_ref.numpy(), t_ref.numpy(), 10)
    same = (rows == rows_ref)
    assert same.mean() > 0.9
    for b, j in zip(*np.nonzero(~same)):                       # a differing entry sits on a near-tie of the oracle's scores
        s_mine = float(q_ref[b] @ t_ref[rows[b, j]])
        assert abs(s_mine - sc_ref[b, j]) <= 4e-3 * abs(sc_ref[b, j]) + 1e-4, (b, j)
