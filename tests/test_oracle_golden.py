"""Pin the CPU oracle against golden vectors produced by the reference's own code.

Tolerances: the oracle runs the same fp32 math in a different association order
(sequential recurrence vs Blelloch sweep), so values agree to a few fp32 ulps of the
largest intermediate; 2e-5 relative / 2e-5 absolute is asserted.
"""
import json
import os

import pytest
import torch

import _golden as G
from oracle import hydra_oracle as O

RT, AT = 2e-5, 2e-5


@pytest.mark.parametrize("L", [1, 2, 3, 10, 25, 64, 100, 101, 512])
def test_pscan(L):
    s, p = G.load("g1_pscan"), f"G1_pscan_L{L}"
    A = G.t(s, p + "/A").requires_grad_()
    X = G.t(s, p + "/X").requires_grad_()
    H = O.pscan(A, X)
    H.backward(G.t(s, p + "/G"))
    G.assert_close(H, G.t(s, p + "/H"), RT, AT, "H")
    G.assert_close(A.grad, G.t(s, p + "/dA"), 1e-4, 1e-4, "dA")
    G.assert_close(X.grad, G.t(s, p + "/dX"), 1e-4, 1e-4, "dX")


@pytest.mark.parametrize("case", ["a", "b"])
def test_selective_scan(case):
    s, p = G.load("g2_selective_scan"), f"G2_sscan_{case}"
    names = ["x", "delta", "A", "B", "C", "D"]
    ins = [G.t(s, f"{p}/{n}").requires_grad_() for n in names]
    y = O.selective_scan(*ins)
    y.backward(G.t(s, p + "/G"))
    G.assert_close(y, G.t(s, p + "/y"), RT, AT, "y")
    for n, v in zip(names, ins):
        G.assert_close(v.grad, G.t(s, f"{p}/d{n}"), 1e-4, 1e-4, "d" + n)


def test_rmsnorm():
    s, p = G.load("g3_rmsnorm"), "G3_rmsnorm"
    m = O.RMSNorm(16)
    m.scale.data = G.t(s, p + "/scale")
    x = G.t(s, p + "/x").requires_grad_()
    y = m(x)
    y.backward(G.t(s, p + "/G"))
    G.assert_close(y, G.t(s, p + "/y"), RT, AT)
    G.assert_close(x.grad, G.t(s, p + "/dx"), 1e-4, 1e-4)
    G.assert_close(m.scale.grad, G.t(s, p + "/dscale"), 1e-4, 1e-4)


def _run_module_case(store, prefix, module, n_in, grad_inputs=(0,)):
    module.load_state_dict(G.params(store, prefix), strict=True)
    module.eval()
    ins = []
    for i in range(n_in):
        v = G.t(store, f"{prefix}/in{i}")
        ins.append(v.requires_grad_() if i in grad_inputs else v)
    y = module(*ins)
    y.backward(G.t(store, prefix + "/G"))
    G.assert_close(y, G.t(store, prefix + "/y"), RT, AT, prefix + " y")
    for i in grad_inputs:
        G.assert_close(ins[i].grad, G.t(store, f"{prefix}/din{i}"), 2e-4, 2e-4, f"{prefix} din{i}")
    gold = G.grads(store, prefix)
    got = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
    assert set(gold) == set(got), (set(gold) ^ set(got))
    for k in gold:
        G.assert_close(got[k], gold[k], 3e-4, 3e-4, f"{prefix} grad {k}")


@pytest.mark.parametrize("hd", [8, 32])
def test_mamba_block(hd):
    _run_module_case(G.load("g4_mamba"), f"G4_mamba_hd{hd}", O.MambaBlock(hd, 16, 4 * hd, 16), 1)


def test_multi_head_mamba():
    m = O.MultiHeadMamba(16, 16, 64, 16, O.make_args(16, 12))
    _run_module_case(G.load("g5_mhm"), "G5_mhm", m, 3, grad_inputs=(0, 1))


@pytest.mark.parametrize("name,cls", [("first", O.HydraBlockFirst), ("regular", O.HydraBlock)])
def test_hydra_blocks(name, cls):
    _run_module_case(G.load("g6_blocks"), f"G6_block_{name}", cls(16, 16, 64, 16, O.make_args(16, 14)), 2)


@pytest.mark.parametrize("L", [2, 7, 10, 25, 26, 40])
def test_multiscale(L):
    m = O.MultiScaleUserEncoder(16, 16, 64, 16, O.make_args(16, L))
    _run_module_case(G.load("g7_multiscale"), f"G7_multiscale_L{L}", m, 2)


def _model_from(store, prefix, dropout=0.2):
    U, I, d, L = (int(store[f"{prefix}/{k}"]) for k in "UIdL")
    args = O.make_args(d, L, dropout=dropout)
    m = O.Hydra(U, I, {}, O.IDS_ONLY_FEAT_TYPES, args, O.SharedEmbeddingModule(U, I, {}, O.IDS_ONLY_FEAT_TYPES, args))
    m.load_state_dict(G.params(store, prefix), strict=True)
    return m


@pytest.mark.parametrize("d", [16, 64])
def test_log2feats(d):
    s, p = G.load("g8_log2feats"), f"G8_log2feats_d{d}"
    m = _model_from(s, p).eval()
    b = {k: G.t(s, f"{p}/{k}") for k in ("seq", "tok", "ts", "act", "dwell")}
    feat = [[] for _ in range(b["seq"].shape[0])]
    with torch.no_grad():
        out = m.log2feats(b["seq"], feat, b["tok"], b["ts"], b["act"], b["dwell"])
        ue = m.get_user_embedding(b["seq"], feat, b["tok"], b["ts"], b["act"], b["dwell"])
    G.assert_close(out, G.t(s, p + "/out"), 5e-5, 5e-5, "log2feats")
    G.assert_close(ue, G.t(s, p + "/user_emb"), 5e-5, 5e-5, "user_emb")
    assert torch.all(out[b["seq"] == 0] == 0)          # padded positions are exactly zero


def test_state_dict_keys_match_reference():
    with open(os.path.join(G.GOLDEN_DIR, "state_dict_keys_d64_L20.json")) as f:
        gold = json.load(f)
    m = O.build_model(30, 120, d=64, L=20)
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == gold
    assert len(got) == 190


def test_topk_and_miner():
    s, p = G.load("g9_topk"), "G9_topk"
    table, users = G.t(s, p + "/table"), G.t(s, p + "/users")
    assert float(s[p + "/min_gap50"]) > 3e-4 and float(s[p + "/min_gap10"]) > 3e-4   # goldens are tie-free
    ids50, _ = O.mine_hard_negatives_similarity(users, table, 50, G.t(s, p + "/pos"), G.t(s, p + "/hist"))
    assert torch.equal(ids50, G.t(s, p + "/ids50"))
    ids10, val10 = O.mine_hard_negatives_similarity(users, table, 10)
    assert torch.equal(ids10, G.t(s, p + "/ids10"))
    G.assert_close(val10, G.t(s, p + "/val10"), 1e-6, 1e-6)


@pytest.mark.parametrize("name", ["small", "chunky"])
def test_train_step_loss_and_grads(name):
    s, p = G.load("g10_train"), f"G10_train_{name}"
    m = _model_from(s, p, dropout=0.0).train()
    b = {k: G.t(s, f"{p}/{k}") for k in ("seq", "pos", "neg", "tok", "next_tok", "ts", "act", "dwell")}
    feat = [[] for _ in range(b["seq"].shape[0])]
    torch.manual_seed(int(s[p + "/rng_seed"]))
    loss = m(b["seq"], b["pos"], b["neg"], b["tok"], b["next_tok"], None, feat, feat, feat,
             b["ts"], b["act"], b["dwell"])
    loss.backward()
    G.assert_close(loss, G.t(s, p + "/loss"), 2e-5, 2e-5, "loss")
    gold = G.grads(s, p)
    got = {k: q.grad for k, q in m.named_parameters() if q.grad is not None}
    tables = {k for k in got if k.endswith(("item_emb.weight", "user_emb.weight"))}
    assert set(gold) == set(got) - tables
    for k in gold:
        G.assert_close(got[k], gold[k], 1e-3, 2e-5, "grad " + k)
    for k in tables:
        rows = G.t(s, f"{p}/grad_rows/{k}")
        vals = G.t(s, f"{p}/grad_vals/{k}")
        g = got[k]
        assert torch.equal(g.abs().sum(1).nonzero().flatten(), rows), k
        G.assert_close(g[rows], vals, 1e-3, 2e-5, "table grad " + k)
        assert float(g[0].abs().sum()) == 0.0


def _train_step_matches(s, p, m):
    b = {k: G.t(s, f"{p}/{k}") for k in ("seq", "pos", "neg", "tok", "next_tok", "ts", "act", "dwell")}
    feat = [[] for _ in range(b["seq"].shape[0])]
    torch.manual_seed(int(s[p + "/rng_seed"]))
    loss = m(b["seq"], b["pos"], b["neg"], b["tok"], b["next_tok"], None, feat, feat, feat,
             b["ts"], b["act"], b["dwell"])
    loss.backward()
    G.assert_close(loss, G.t(s, p + "/loss"), 2e-5, 2e-5, "loss")
    gold = G.grads(s, p)
    got = {k: q.grad for k, q in m.named_parameters() if q.grad is not None}
    tables = {k for k in got if k.endswith(("item_emb.weight", "user_emb.weight"))}
    assert set(gold) == set(got) - tables
    for k in gold:
        G.assert_close(got[k], gold[k], 1e-3, 2e-5, "grad " + k)
    for k in tables:
        rows = G.t(s, f"{p}/grad_rows/{k}")
        g = got[k]
        assert torch.equal(g.abs().sum(1).nonzero().flatten(), rows), k
        G.assert_close(g[rows], G.t(s, f"{p}/grad_vals/{k}"), 1e-3, 2e-5, "table grad " + k)


@pytest.mark.parametrize("drop", [False, True])
def test_train_step_at_the_benchmark_width(drop):
    """d = 64 (head_dim 32, inner width 128): the reference's train step without dropout, and with the blocks'
    dropout applied through the multipliers the fixture recorded (attention-weight dropout off)."""
    s = G.load("g14_train_d64")
    base = "G14_train_d64"
    p = base + ("_drop" if drop else "")
    U, I, d, L = (int(s[f"{base}/{k}"]) for k in "UIdL")
    args = O.make_args(d, L, dropout=0.0)
    m = O.Hydra(U, I, {}, O.IDS_ONLY_FEAT_TYPES, args, O.SharedEmbeddingModule(U, I, {}, O.IDS_ONLY_FEAT_TYPES, args))
    m.load_state_dict(G.params(s, base), strict=True)
    m.train()
    if drop:
        mults = G.block_drop_multipliers(s, p)
        assert set(mults) == set(G.BLOCK_NAMES)
        mods = dict(m.named_modules())
        for name, mult in mults.items():
            mods[name].dropout = G.FixedDropout(mult)
    _train_step_matches(s, p, m)


@pytest.mark.parametrize("name", ["small", "chunky"])
def test_recorded_draws_reproduce_the_reference_loss(name):
    """The fixture also holds what the reference drew (its torch.randint outputs and mined ids).  Assembled into
    index tensors (tests/_golden.py) and fed to the index form of the loss, they must give the reference's loss:
    this pins the assembly that the GPU train-step test relies on."""
    s, p = G.load("g10_train"), f"G10_train_{name}"
    m = _model_from(s, p, dropout=0.0).train()
    b = {k: G.t(s, f"{p}/{k}") for k in ("seq", "pos", "neg", "tok", "next_tok", "ts", "act", "dwell")}
    feat = [[] for _ in range(b["seq"].shape[0])]
    with torch.no_grad():
        lf = m.log2feats(b["seq"], feat, b["tok"], b["ts"], b["act"], b["dwell"])
        table = m.shared_embeddings.item_emb.weight
        pos = m.hydra_projection(m.shared_embeddings.get_complete_embeddings(b["pos"], feat, b["tok"], False, 'early', 'hydra'))
        mined = O.HardNegativeMiner(m, table, 50).mine_hard_negatives_similarity(lf[:, -1], b["pos"], b["seq"])
        assert torch.equal(mined, G.t(s, p + "/mined_ids"))
        mask = b["next_tok"] == 1
        hard, inb = G.assemble_train_draws(s, p, mask)
        loss = O.nce_rows(lf[mask], pos[mask], table, hard, inb).mean()
    G.assert_close(loss, G.t(s, p + "/loss"), 2e-5, 2e-5, "loss from recorded draws")


def test_d128_widths_match_reference():
    """MambaBlock at head_dim 64 (inner width 256) and a regular HydraBlock at d = 128: the long-history widths."""
    s = G.load("g12_d128")
    _run_module_case(s, "G12_mamba_hd64", O.MambaBlock(64, 16, 256, 16), 1)
    _run_module_case(s, "G12_block_regular_d128", O.HydraBlock(128, 16, 512, 16, O.make_args(128, 11)), 2)


def test_candidate_embedding_generation_matches_reference():
    """embedding.py:654-735 (ids-only): the oracle's late-fusion rows against what the reference produced."""
    s = G.load("g11_item_embs")
    w = G.t(s, "G11/item_w")
    G.assert_close(O.item_embedding_table(w), G.t(s, "G11/full"), 1e-6, 1e-7, "full table")
    G.assert_close(O.late_item_rows(w, G.t(s, "G11/seq")), G.t(s, "G11/late"), 1e-6, 1e-7, "late")
    G.assert_close(O.late_item_rows(w, G.t(s, "G11/item_ids")), G.t(s, "G11/saved/embedding.fbin"), 1e-6, 1e-7, "saved")


# ---- feature-fusion paths (embedding.py:216-264, 486-652) ---------------------------------------------------
def feature_module_from(store, prefix, cls, device='cpu'):
    U, I, d, L = (int(store[f"{prefix}/{k}"]) for k in "UIdL")
    args = O.make_args(d, L, dropout=0.0)
    args.device = device
    inter = G.js(store, prefix + "/interaction_json") if prefix + "/interaction_json" in store else None
    se = cls(U, I, G.js(store, prefix + "/feat_stats_json"), G.js(store, prefix + "/feat_types_json"), args, inter)
    return se, args


def feature_runs(se, s, p, to=lambda x: x):
    feats = G.js(s, p + "/feats_json")
    seq, tok, pos, poss = (to(G.t(s, f"{p}/{k}")) for k in ("seq", "tok", "pos", "poss"))
    return {"early_user": lambda: se.get_complete_embeddings(seq, feats, mask=tok, include_user=True, fusion_style="early",
                                                             model_style="hydra", pos_seq=poss),
            "early_item": lambda: se.get_complete_embeddings(pos, feats, mask=tok, include_user=False,
                                                             fusion_style="early", model_style="hydra"),
            "late_item": lambda: se.get_complete_embeddings(pos, feats, include_user=False, fusion_style="late"),
            "late_user": lambda: se.get_feature_embeddings(feats, include_user=True, fusion_style="late"),
            "gated_user": lambda: se.get_feature_embeddings(feats, include_user=True, fusion_style="gated")}


@pytest.mark.parametrize("run", ["early_user", "early_item", "late_item", "late_user", "gated_user"])
def test_feature_fusion_styles(run):
    s, p = G.load("g13_features"), "G13_mix"
    se, _ = feature_module_from(s, p, O.SharedEmbeddingModule)
    se.load_state_dict(G.params(s, p), strict=True)
    se.eval()
    y = feature_runs(se, s, p)[run]()
    y.backward(G.t(s, f"{p}/{run}/G"))
    G.assert_close(y, G.t(s, f"{p}/{run}/y"), RT, AT, run)
    gold = G.grads(s, f"{p}/{run}")
    got = {k: q.grad for k, q in se.named_parameters() if q.grad is not None}
    assert set(gold) == set(got), set(gold) ^ set(got)
    for k in gold:
        G.assert_close(got[k], gold[k], 3e-4, 3e-5, f"{run} grad {k}")


def test_train_step_with_features():
    s, p = G.load("g13_features"), "G13_train"
    se, args = feature_module_from(s, p, O.SharedEmbeddingModule)
    m = O.Hydra(int(s[p + "/U"]), int(s[p + "/I"]), {}, {}, args, se)
    m.load_state_dict(G.params(s, p), strict=True)
    m.train()
    b = {k: G.t(s, f"{p}/{k}") for k in ("seq", "pos", "neg", "tok", "next_tok", "ts", "act", "dwell")}
    f_seq, f_pos, f_neg = (G.js(s, f"{p}/f_{k}_json") for k in ("seq", "pos", "neg"))
    torch.manual_seed(int(s[p + "/rng_seed"]))
    loss = m(b["seq"], b["pos"], b["neg"], b["tok"], b["next_tok"], None, f_seq, f_pos, f_neg, b["ts"], b["act"],
             b["dwell"])
    loss.backward()
    G.assert_close(loss, G.t(s, p + "/loss"), 2e-5, 2e-5, "loss")
    gold = G.grads(s, p)
    got = {k: q.grad for k, q in m.named_parameters() if q.grad is not None}
    assert set(gold) == set(got), set(gold) ^ set(got)
    for k in gold:
        G.assert_close(got[k], gold[k], 1e-3, 2e-5, "grad " + k)
