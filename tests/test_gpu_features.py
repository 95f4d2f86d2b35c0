"""GPU parity of the feature-fusion paths (embedding.py:216-264, 486-652) through the drop-in SharedEmbeddingModule,
i.e. through the C ABI, against (1) what the reference itself produced (tests/golden/g13_features.npz) and (2) torch
fp32 for the row kernels on their own.

Tolerance: row gathers, bag sums, activations, gates and LayerNorm are fp32 (1e-5 class); the projections run on
split-bf16 MFMA, so module outputs and gradients are asserted at the encoder tolerances of test_gpu_encoder.py
(max 2e-3 / rms 1e-3 forward, 4x / 6.5x for parameter gradients)."""
import numpy as np
import pytest
import torch

import _golden as G
from generative_recommenders_amd import enc_ops as E
from generative_recommenders_amd import feature_ops as FO
from generative_recommenders_amd import hydra as H
from generative_recommenders_amd import shared_embedding as SE
from test_gpu_encoder import close
from test_oracle_golden import feature_module_from, feature_runs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n,bag,d,rows", [(37, 3, 16, 50), (1000, 5, 64, 4000), (9, 1, 6, 7), (4, 0, 8, 5), (0, 4, 8, 5)])
def test_bag_sum_forward_backward(n, bag, d, rows):
    g = torch.Generator().manual_seed(n + bag)
    w = torch.randn(rows, d, generator=g)
    ids = torch.randint(0, rows, (n, bag), generator=g)
    if n and bag:
        ids[0, 0] = 0
    G_ = torch.randn(n, d, generator=g)
    wr = w.clone().requires_grad_()
    want = torch.nn.functional.embedding(ids, wr, padding_idx=0).sum(1)
    want.backward(G_)
    wd = w.to(DEV).requires_grad_()
    got = FO.gather_sum_rows(wd, ids.to(DEV), 0)
    got.backward(G_.to(DEV))
    G.assert_close(got.detach().cpu(), want.detach(), 1e-6, 1e-6, "bag sum")
    G.assert_close(wd.grad.cpu(), wr.grad, 1e-5, 1e-6, "bag sum table grad")
    assert float(wd.grad[0].abs().sum()) == 0.0            # padding row


@pytest.mark.parametrize("n", [1, 7, 4096, 100003])
def test_activations_and_gate(n):
    g = torch.Generator().manual_seed(n)
    x, q, G_ = (torch.randn(n, generator=g) * 2 for _ in range(3))
    for fn, ref in ((FO.relu, torch.relu), (FO.gelu, torch.nn.functional.gelu)):
        xr = x.clone().requires_grad_()
        ref(xr).backward(G_)
        xd = x.to(DEV).requires_grad_()
        y = fn(xd)
        y.backward(G_.to(DEV))
        G.assert_close(y.detach().cpu(), ref(x), 1e-5, 1e-6, fn.__name__)
        G.assert_close(xd.grad.cpu(), xr.grad, 1e-4, 1e-6, fn.__name__ + " grad")
    pr, qr = x.clone().requires_grad_(), q.clone().requires_grad_()
    (pr * torch.sigmoid(qr)).backward(G_)
    pd, qd = x.to(DEV).requires_grad_(), q.to(DEV).requires_grad_()
    y = FO.sigmoid_gate(pd, qd)
    y.backward(G_.to(DEV))
    G.assert_close(y.detach().cpu(), x * torch.sigmoid(q), 1e-5, 1e-6, "gate")
    G.assert_close(pd.grad.cpu(), pr.grad, 1e-4, 1e-6, "gate dp")
    G.assert_close(qd.grad.cpu(), qr.grad, 1e-4, 1e-6, "gate dq")


@pytest.mark.parametrize("T,d", [(1, 16), (37, 32), (1000, 256), (5000, 64), (3, 512), (0, 16)])
def test_layernorm(T, d):
    g = torch.Generator().manual_seed(T + d)
    x = torch.randn(T, d, generator=g) * 3 + 1
    w, b, G_ = torch.randn(d, generator=g), torch.randn(d, generator=g), torch.randn(T, d, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    torch.nn.functional.layer_norm(xr, (d,), wr, br).backward(G_)
    xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = FO.layernorm(xd, wd, bd)
    y.backward(G_.to(DEV))
    G.assert_close(y.detach().cpu(), torch.nn.functional.layer_norm(x, (d,), w, b), 1e-4, 1e-5, "layernorm")
    G.assert_close(xd.grad.cpu(), xr.grad, 1e-3, 2e-5, "layernorm dx")
    G.assert_close(wd.grad.cpu(), wr.grad, 1e-3, 1e-4 * max(1, T) ** 0.5, "layernorm dw")
    G.assert_close(bd.grad.cpu(), br.grad, 1e-3, 1e-4 * max(1, T) ** 0.5, "layernorm db")


@pytest.mark.parametrize("B,Lq,Lk,d,H,with_drop", [(2, 5, 5, 16, 2, False), (3, 100, 100, 64, 2, True),
                                                  (1, 7, 130, 32, 4, True), (2, 1, 3, 8, 1, False)])
def test_attention_core(B, Lq, Lk, d, H, with_drop):
    """softmax(q k^T / sqrt(hd)) (times a dropout multiplier) v per head, against the same formula in torch."""
    g = torch.Generator().manual_seed(B * 100 + Lq)
    q, kv, G_ = torch.randn(B * Lq, d, generator=g), torch.randn(B * Lk, 2 * d, generator=g), torch.randn(B * Lq, d, generator=g)
    drop = ((torch.rand(B, H, Lq, Lk, generator=g) > 0.3).float() / 0.7) if with_drop else None
    hd = d // H

    def ref(q, kv):
        qh = q.view(B, Lq, H, hd).transpose(1, 2)
        kh = kv[:, :d].reshape(B, Lk, H, hd).transpose(1, 2)
        vh = kv[:, d:].reshape(B, Lk, H, hd).transpose(1, 2)
        w = torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, dim=-1)
        if drop is not None:
            w = w * drop
        return (w @ vh).transpose(1, 2).reshape(B * Lq, d)

    qr, kvr = q.clone().requires_grad_(), kv.clone().requires_grad_()
    want = ref(qr, kvr)
    want.backward(G_)
    qd, kvd = q.to(DEV).requires_grad_(), kv.to(DEV).requires_grad_()
    got = FO.attention(qd, kvd, None if drop is None else drop.to(DEV), B, Lq, Lk, H)
    got.backward(G_.to(DEV))
    G.assert_close(got.detach().cpu(), want.detach(), 1e-4, 1e-5, "attention")
    G.assert_close(qd.grad.cpu(), qr.grad, 1e-3, 2e-5, "attention dq")
    G.assert_close(kvd.grad.cpu(), kvr.grad, 1e-3, 2e-5, "attention dkv")


def test_projection_column_blocks_and_unit_width_input():
    """Column blocks of one wide weight accumulated launch by launch equal the projection of the concatenation, and
    the Linear(1, d) of a continual feature (K = 1) runs on the same kernel."""
    g = torch.Generator().manual_seed(5)
    T, d = 77, 16
    parts = [torch.randn(T, d, generator=g) for _ in range(3)]
    W, b, G_ = torch.randn(d, 3 * d, generator=g), torch.randn(d, generator=g), torch.randn(T, d, generator=g)
    Wr = W.clone().requires_grad_()
    pr = [p.clone().requires_grad_() for p in parts]
    torch.nn.functional.linear(torch.cat(pr, 1), Wr, b).backward(G_)
    Wd = W.to(DEV).requires_grad_()
    pd = [p.to(DEV).requires_grad_() for p in parts]
    acc = None
    for i, p in enumerate(pd):
        acc = E.linear(p, Wd, b.to(DEV) if i == 0 else None, r1=acc, c1=1.0 if acc is not None else 0.0, col_off=i * d)
    acc.backward(G_.to(DEV))
    close(acc, torch.nn.functional.linear(torch.cat(parts, 1), W, b), 2e-3, 1e-3, "blocks")
    close(Wd.grad, Wr.grad, 8e-3, 6.5e-3, "blocks dW")
    for a, r in zip(pd, pr):
        close(a.grad, r.grad, 4e-3, 2.5e-3, "blocks dx")
    x1, W1, b1 = torch.randn(T, 1, generator=g), torch.randn(d, 1, generator=g), torch.randn(d, generator=g)
    W1r, b1r = W1.clone().requires_grad_(), b1.clone().requires_grad_()
    torch.nn.functional.linear(x1, W1r, b1r).backward(G_)
    W1d, b1d = W1.to(DEV).requires_grad_(), b1.to(DEV).requires_grad_()
    y = E.linear(x1.to(DEV), W1d, b1d)
    y.backward(G_.to(DEV))
    close(y, torch.nn.functional.linear(x1, W1, b1), 2e-3, 1e-3, "K = 1")
    close(W1d.grad, W1r.grad, 8e-3, 6.5e-3, "K = 1 dW")
    close(b1d.grad, b1r.grad, 8e-3, 6.5e-3, "K = 1 db")


@pytest.mark.parametrize("T,K,N,scaled", [(77, 1024, 128, True), (300, 3584, 64, True), (33, 4096, 256, False),
                                          (50, 300, 96, False), (19, 32, 64, True), (5, 1027, 32, True)])
def test_wide_projection_in_column_chunks(T, K, N, scaled):
    """nn.Linear with more than 256 input columns (raw multimodal item vectors): chunks of 256 columns added launch
    by launch, the F.normalize of the rows carried as a row scale; autograd accumulation and the direct gradient
    mode give the same parameter gradients."""
    g = torch.Generator().manual_seed(T + K)
    x = torch.randn(T, K, generator=g) * 2
    if scaled:
        x[0] = 0                                                  # a zero row stays zero (eps clamp)
    W, b, G_ = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(T, N, generator=g)
    Wr, br, xr = W.double().requires_grad_(), b.double().requires_grad_(), x.double().requires_grad_()
    xin = torch.nn.functional.normalize(xr, p=2, dim=-1) if scaled else xr
    want = torch.nn.functional.linear(xin, Wr, br)
    want.backward(G_.double())
    if scaled:
        inv = FO.row_inv_norm(x.to(DEV))
        G.assert_close(inv.cpu(), 1.0 / x.norm(dim=1).clamp_min(1e-12), 1e-5, 0, "row scale")
    for direct in (False, True):
        Wd, bd = W.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        xd = x.to(DEV).requires_grad_(not scaled)
        E.set_direct_param_grad(direct)
        try:
            y = FO.linear_any(xd, Wd, bd, in_scale=FO.row_inv_norm(xd.detach()) if scaled else None)
            y.backward(G_.to(DEV))
        finally:
            E.set_direct_param_grad(False)
        close(y, want.float(), 2e-3, 1e-3, "wide")
        close(Wd.grad, Wr.grad.float(), 8e-3, 6.5e-3, f"wide dW direct={direct}")
        close(bd.grad, br.grad.float(), 8e-3, 6.5e-3, f"wide db direct={direct}")
        if not scaled:
            close(xd.grad, xr.grad.float(), 4e-3, 2.5e-3, "wide dx")


@pytest.mark.parametrize("n,p", [(1, 0.5), (7, 0.2), (4096, 0.2), (1000003, 0.3), (65536, 0.0), (1001, 1.0)])
def test_dropout_multiplier(n, p):
    """Values are 0 or 1 / (1 - p), the kept fraction is 1 - p within five standard deviations, a draw repeats under
    the same torch seed and differs under another, and consecutive draws differ."""
    torch.manual_seed(11)
    a = FO.dropout_multiplier((n,), p, DEV)
    b = FO.dropout_multiplier((n,), p, DEV)
    torch.manual_seed(11)
    a2 = FO.dropout_multiplier((n,), p, DEV)
    torch.manual_seed(12)
    c = FO.dropout_multiplier((n,), p, DEV)
    assert torch.equal(a, a2)
    keep = 1.0 - p
    scale = np.float32(1.0) / np.float32(keep) if keep > 0 else 0.0
    vals = set(a.unique().tolist())
    assert vals <= {0.0, float(scale)}
    if p in (0.0, 1.0):
        assert float(a.sum()) == (n if p == 0.0 else 0.0)
        return
    if n >= 4096:
        frac = float((a > 0).float().mean())
        assert abs(frac - keep) < 5 * (keep * p / n) ** 0.5
        assert not torch.equal(a, b) and not torch.equal(a, c)
        # neighbours are independent: the kept fraction among elements that follow a kept one is 1 - p as well
        k = a > 0
        both = float((k[1:] & k[:-1]).float().mean())
        assert abs(both - keep * keep) < 6 * (keep * keep * (1 - keep * keep) / n) ** 0.5
    # an unaligned view (odd element offset) takes the scalar store path and holds the same stream
    buf = torch.zeros(n + 1, device=DEV)
    from generative_recommenders_amd import hmx as HM
    HM.check(HM.lib().hmx_dropout_multiplier_f32(123, 5, n, p, buf[1:].data_ptr(), None), "dropout")
    ref = torch.zeros(n, device=DEV)
    HM.check(HM.lib().hmx_dropout_multiplier_f32(123, 5, n, p, ref.data_ptr(), None), "dropout")
    torch.cuda.synchronize()
    assert torch.equal(buf[1:], ref) and float(buf[0]) == 0.0


def test_dropout_rows_forward_backward():
    x = torch.randn(300, 64, device=DEV, requires_grad=True)
    assert FO.dropout(x, 0.2, False) is x and FO.dropout(x, 0.0, True) is x
    torch.manual_seed(3)
    y = FO.dropout(x, 0.2, True)
    torch.manual_seed(3)
    m = FO.dropout_multiplier(x.shape, 0.2, DEV)
    g = torch.randn_like(x)
    y.backward(g)
    assert torch.equal(y.detach(), x.detach() * m) and torch.equal(x.grad, g * m)


def _module(s, p):
    se, args = feature_module_from(s, p, SE.SharedEmbeddingModule, device=DEV)
    return se, args


@pytest.mark.parametrize("run", ["early_user", "early_item", "late_item", "late_user", "gated_user"])
def test_feature_fusion_styles_against_reference_golden(run):
    s, p = G.load("g13_features"), "G13_mix"
    se, _ = _module(s, p)
    se.load_state_dict(G.params(s, p), strict=True)        # every reference key, nothing else
    se.to(DEV).eval()
    y = feature_runs(se, s, p, to=lambda x: x.to(DEV))[run]()
    close(y, G.t(s, f"{p}/{run}/y"), 2e-3, 1e-3, run)
    y.backward(G.t(s, f"{p}/{run}/G").to(DEV))
    gold = G.grads(s, f"{p}/{run}")
    got = {k: q.grad for k, q in se.named_parameters() if q.grad is not None}
    assert set(gold) == set(got), set(gold) ^ set(got)
    for k in gold:
        close(got[k], gold[k], 8e-3, 6.5e-3, f"{run} grad {k}")


def test_prepacked_tensor_features_equal_the_list_format():
    """A dict {feature id: tensor} (ids (B, L), bags (B, L, A), values (B, L), vectors (B, L, dim)) skips the host
    packing and must give the same bits as the reference's list-of-dicts format."""
    s, p = G.load("g13_features"), "G13_mix"
    se, _ = _module(s, p)
    se.load_state_dict(G.params(s, p), strict=True)
    se.to(DEV).eval()
    feats = G.js(s, p + "/feats_json")
    packed = {}
    for kind, k in se._groups(include_user=True, interaction=True):
        if kind == 'item_emb':
            packed[k] = torch.from_numpy(SE.pack_dense_feature(feats, k, se.ITEM_EMB_FEAT[k]))
        elif kind == 'array':
            packed[k] = torch.from_numpy(SE.pack_array_feature(feats, k))
        else:
            packed[k] = torch.from_numpy(SE.pack_scalar_feature(feats, k, kind == 'continual'))
    with torch.no_grad():
        for style in ("early", "late", "gated"):
            a = se.get_feature_embeddings(feats, include_user=True, fusion_style=style)
            b = se.get_feature_embeddings(packed, include_user=True, fusion_style=style)
            assert torch.equal(a, b), style


def test_candidate_table_with_features_against_reference_golden():
    s, p = G.load("g13_features"), "G13_mix"
    se, _ = _module(s, p)
    se.load_state_dict(G.params(s, p), strict=True)
    se.to(DEV).eval()
    item_feats = {int(k): v for k, v in G.js(s, p + "/table/item_feats_json").items()}
    full = se.get_item_embeddings(item_feats, batch_size=16)
    close(full, G.t(s, p + "/table/full"), 2e-3, 1e-3, "candidate table")
    assert float(full[0].abs().sum()) == 0.0


def test_train_step_with_features_against_reference_golden():
    """Hydra.forward + backward with sparse / array / continual features on all three feature arguments, on the
    reference's own draws: loss to 1e-4, every parameter gradient at the encoder tolerances."""
    s, p = G.load("g13_features"), "G13_train"
    se, args = _module(s, p)
    m = H.Hydra(int(s[p + "/U"]), int(s[p + "/I"]), {}, {}, args, se)
    m.load_state_dict(G.params(s, p), strict=True)
    m.to(DEV).train()
    b = {k: G.t(s, f"{p}/{k}").to(DEV) for k in ("seq", "pos", "neg", "tok", "next_tok", "ts", "act", "dwell")}
    f_seq, f_pos, f_neg = (G.js(s, f"{p}/f_{k}_json") for k in ("seq", "pos", "neg"))
    hard, inb = G.assemble_train_draws(s, p, G.t(s, p + "/next_tok") == 1)
    loss = m(b["seq"], b["pos"], b["neg"], b["tok"], b["next_tok"], None, f_seq, f_pos, f_neg, b["ts"], b["act"],
             b["dwell"], predrawn=(hard.to(DEV), inb.to(DEV)))
    loss.backward()
    assert torch.equal(m.last_mined_ids.cpu(), G.t(s, p + "/mined_ids"))
    ref_loss = float(s[p + "/loss"])
    assert abs(loss.item() - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    gold = G.grads(s, p)
    got = {k: q.grad for k, q in m.named_parameters() if q.grad is not None}
    assert set(gold) == set(got), set(gold) ^ set(got)
    for k in gold:
        close(got[k], gold[k], 8e-3, 6.5e-3, f"grad {k}")


def test_feature_train_step_direct_gradient_modes_match_autograd_accumulation():
    """The two opt-in gradient modes of the training loop (table gradients added straight into .grad, projection
    weight gradients queued into batched launches) on a step WITH features: column-block projections, bag sums and
    feature tables must give the gradients of the standard autograd path."""
    from generative_recommenders_amd import ops
    s, p = G.load("g13_features"), "G13_train"
    b = {k: G.t(s, f"{p}/{k}").to(DEV) for k in ("seq", "pos", "neg", "tok", "next_tok", "ts", "act", "dwell")}
    f_seq, f_pos, f_neg = (G.js(s, f"{p}/f_{k}_json") for k in ("seq", "pos", "neg"))
    hard, inb = G.assemble_train_draws(s, p, G.t(s, p + "/next_tok") == 1)
    grads, losses = [], []
    try:
        for direct in (False, True):
            E.set_direct_param_grad(direct)
            ops.set_fused_table_grad(direct)
            se, args = _module(s, p)
            m = H.Hydra(int(s[p + "/U"]), int(s[p + "/I"]), {}, {}, args, se)
            m.load_state_dict(G.params(s, p), strict=True)
            m.to(DEV).train()
            for rep in range(2 if direct else 1):        # second pass: gradient buffers are recycled after zero_grad
                m.zero_grad(set_to_none=True)
                loss = m(b["seq"], b["pos"], b["neg"], b["tok"], b["next_tok"], None, f_seq, f_pos, f_neg, b["ts"],
                         b["act"], b["dwell"], predrawn=(hard.to(DEV), inb.to(DEV)))
                loss.backward()
            losses.append(loss.item())
            grads.append({k: (None if q.grad is None else q.grad.clone()) for k, q in m.named_parameters()})
    finally:
        E.set_direct_param_grad(False)
        ops.set_fused_table_grad(False)
    assert losses[0] == losses[1]
    assert grads[0].keys() == grads[1].keys()
    for k in grads[0]:
        a, c = grads[0][k], grads[1][k]
        assert (a is None) == (c is None), k
        if a is not None:
            close(c, a, 2e-5, 2e-5, "direct grad " + k)


def test_saved_item_embeddings_with_features(tmp_path):
    """embedding.fbin / id.u64bin written from late-fusion rows with features (embedding.py:654-684; feat_dict is
    indexed by position): the rows equal get_complete_embeddings on the same ids and features."""
    from generative_recommenders_amd.ann_files import load_fbin, load_u64bin
    s, p = G.load("g13_features"), "G13_mix"
    se, _ = _module(s, p)
    se.load_state_dict(G.params(s, p), strict=True)
    se.to(DEV).eval()
    item_feats = {int(k): v for k, v in G.js(s, p + "/table/item_feats_json").items()}
    item_ids = [5, 1, 40, 33, 2, 17, 9]
    feat_by_pos = {i: item_feats.get(item, {}) for i, item in enumerate(item_ids)}
    se.save_item_embeddings(item_ids, [100 + i for i in item_ids], feat_by_pos, str(tmp_path), batch_size=3)
    rows = load_fbin(tmp_path / "embedding.fbin")
    ids = load_u64bin(tmp_path / "id.u64bin")
    assert list(ids.reshape(-1)) == [100 + i for i in item_ids]
    # same batches of three as the writer: with a non-zero padding row (this fixture's tables are fully random) an
    # absent array feature contributes (longest bag of the batch) x row 0, in the reference as well
    with torch.no_grad():
        want = torch.cat([se.get_complete_embeddings(torch.tensor(item_ids[i:i + 3], device=DEV).unsqueeze(1),
                                                     [[feat_by_pos[j]] for j in range(i, min(i + 3, len(item_ids)))],
                                                     include_user=False, fusion_style="late").squeeze(1)
                          for i in range(0, len(item_ids), 3)])
    assert np.array_equal(rows, want.cpu().numpy())
