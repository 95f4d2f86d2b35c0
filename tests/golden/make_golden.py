"""Generate golden vectors by running the REFERENCE's own hydra3.py / embedding.py on CPU.

Runs only in the build container (needs /root/reference).  The reference source
never leaves that directory: what is written here is data only -- weights, inputs,
outputs and gradients as .npz arrays plus a JSON list of state_dict keys/shapes.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json

The absent sibling module `dataset2` (imported by the reference for two names that
the hot path never calls) is stubbed before import.
"""
import contextlib
import io
import json
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REF = os.environ.get("HYDRA_REFERENCE_DIR", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
_stub = types.ModuleType("dataset2")
_stub.save_emb = lambda *a, **k: None
_stub.MyDataset = type("MyDataset", (), {})
sys.modules["dataset2"] = _stub
sys.path.insert(0, REF)
import embedding as ref_emb  # noqa: E402
import hydra3 as ref  # noqa: E402

FEAT_TYPES = {k: [] for k in ("user_sparse", "item_sparse", "user_array", "item_array",
                              "user_continual", "item_continual", "item_emb")}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def make_args(d, L, heads=2, blocks=2, dropout=0.2):
    a = Namespace(device="cpu", hydra_hidden_units=d, hstu_hidden_units=d, hydra_num_heads=heads,
                  hydra_num_blocks=blocks, hydra_dropout_rate=dropout, maxlen=L - 1, norm_first=False,
                  num_actions=5, time_span=512)
    ref_emb.SharedEmbeddingConfig(a)
    return a


def xavier_(module):
    for p in module.parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
        else:                      # make biases / scales / D non-trivial so they are actually tested
            p.data.add_(0.1 * torch.randn_like(p))


def put(store, prefix, **arrays):
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        store[f"{prefix}/{k}"] = np.asarray(v)


def put_params(store, prefix, module):
    for k, v in module.state_dict().items():
        store[f"{prefix}/param/{k}"] = v.detach().cpu().numpy()


def g1_pscan(store):
    for L in (1, 2, 3, 10, 25, 64, 100, 101, 512):
        torch.manual_seed(100 + L)
        B, ED, N = (2, 3, 4) if L < 512 else (1, 2, 2)
        A = torch.rand(B, L, ED, N).mul_(0.9).add_(0.05).requires_grad_()
        X = torch.randn(B, L, ED, N, requires_grad=True)
        G = torch.randn(B, L, ED, N)
        H = ref.pscan(A, X)
        H.backward(G)
        put(store, f"G1_pscan_L{L}", A=A, X=X, G=G, H=H, dA=A.grad, dX=X.grad)


def g2_selective_scan(store):
    for name, (B, L, ED, N) in {"a": (2, 7, 6, 4), "b": (3, 33, 8, 16)}.items():
        torch.manual_seed(200 + L)
        x = torch.randn(B, L, ED, requires_grad=True)
        delta = torch.rand(B, L, ED).add_(0.01).requires_grad_()
        A = (-torch.rand(ED, N) * 3).requires_grad_()
        Bm = torch.randn(B, L, N, requires_grad=True)
        C = torch.randn(B, L, N, requires_grad=True)
        D = torch.randn(ED, requires_grad=True)
        G = torch.randn(B, L, ED)
        y = ref.selective_scan(x, delta, A, Bm, C, D)
        y.backward(G)
        put(store, f"G2_sscan_{name}", x=x, delta=delta, A=A, B=Bm, C=C, D=D, G=G, y=y,
            dx=x.grad, ddelta=delta.grad, dA=A.grad, dB=Bm.grad, dC=C.grad, dD=D.grad)


def g3_rmsnorm(store):
    torch.manual_seed(300)
    m = ref.RMSNorm(16)
    m.scale.data = torch.randn(16)
    x = torch.randn(3, 5, 16, requires_grad=True)
    x.data[1, 2] = 0.0                                   # an all-zero row (padded position)
    G = torch.randn(3, 5, 16)
    y = m(x)
    y.backward(G)
    put(store, "G3_rmsnorm", x=x, scale=m.scale, G=G, y=y, dx=x.grad, dscale=m.scale.grad)


def module_case(store, prefix, module, inputs, grad_inputs=(0,)):
    """Run module(*inputs) in eval mode, record outputs and grads wrt inputs[i] and all params."""
    module.eval()
    ins = [t.clone().requires_grad_(i in grad_inputs) if torch.is_floating_point(t) else t
           for i, t in enumerate(inputs)]
    y = module(*ins)
    G = torch.randn_like(y)
    y.backward(G)
    put_params(store, prefix, module)
    put(store, prefix, y=y, G=G)
    for i, t in enumerate(ins):
        put(store, prefix, **{f"in{i}": t})
        if torch.is_floating_point(t) and t.grad is not None:
            put(store, prefix, **{f"din{i}": t.grad})
    for k, p in module.named_parameters():
        if p.grad is not None:
            store[f"{prefix}/grad/{k}"] = p.grad.detach().numpy()


def g4_mamba(store):
    for hd, L, B in ((8, 9, 2), (32, 21, 2)):
        torch.manual_seed(400 + hd)
        m = ref.MambaBlock(hd, 16, 4 * hd, 16)
        xavier_(m)
        module_case(store, f"G4_mamba_hd{hd}", m, [torch.randn(B, L, hd)])


def g12_d128(store):
    """The long-history configuration's widths (d = 128: head_dim 64, inner width 256): one MambaBlock and one
    regular HydraBlock."""
    torch.manual_seed(1264)
    m = ref.MambaBlock(64, 16, 256, 16)
    xavier_(m)
    module_case(store, "G12_mamba_hd64", m, [torch.randn(2, 19, 64)])
    torch.manual_seed(1228)
    a = make_args(128, 11)
    blk = ref.HydraBlock(128, 16, 512, 16, a)
    xavier_(blk)
    pm = pad_mask(2, 11, (0, 4))
    module_case(store, "G12_block_regular_d128", blk, [torch.randn(2, 11, 128) * pm, pm])


def pad_mask(B, L, pads):
    m = torch.ones(B, L, 1)
    for b, p in enumerate(pads):
        m[b, :p] = 0
    return m


def g5_mhm(store):
    torch.manual_seed(500)
    a = make_args(16, 12)
    m = ref.MultiHeadMamba(16, 16, 64, 16, a)
    xavier_(m)
    x, xo = torch.randn(3, 12, 16), torch.randn(3, 12, 16)
    module_case(store, "G5_mhm", m, [x, xo, pad_mask(3, 12, (0, 4, 11))], grad_inputs=(0, 1))


def g6_blocks(store):
    for name, cls in (("first", ref.HydraBlockFirst), ("regular", ref.HydraBlock)):
        torch.manual_seed(600 + len(name))
        a = make_args(16, 14)
        m = cls(16, 16, 64, 16, a)
        xavier_(m)
        pm = pad_mask(3, 14, (0, 5, 13))
        module_case(store, f"G6_block_{name}", m, [torch.randn(3, 14, 16) * pm, pm])


def g7_multiscale(store):
    for L in (2, 7, 10, 25, 26, 40):
        torch.manual_seed(700 + L)
        a = make_args(16, L)
        m = ref.MultiScaleUserEncoder(16, 16, 64, 16, a)
        xavier_(m)
        pm = pad_mask(2, L, (0, L // 2))
        module_case(store, f"G7_multiscale_L{L}", m, [torch.randn(2, L, 16) * pm, pm])


def build_ref_model(U, I, d, L, dropout, seed):
    a = make_args(d, L, dropout=dropout)
    torch.manual_seed(seed)
    with quiet():
        se = ref_emb.SharedEmbeddingModule(U, I, {}, FEAT_TYPES, a, None)
        m = ref.Hydra(U, I, {}, FEAT_TYPES, a, se)
    for _, p in m.named_parameters():                                  # main.py:603-607
        try:
            torch.nn.init.xavier_normal_(p.data)
        except Exception:
            pass
    for t in (se.pos_emb, se.item_emb, se.user_emb, se.action_emb):    # main.py:619-622
        t.weight.data[0, :] = 0
    return m, a


def synth_batch(B, L, U, I, pads, seed, user_slot=True):
    g = torch.Generator().manual_seed(seed)
    seq = torch.randint(1, I + 1, (B, L), generator=g)
    tok = torch.ones(B, L, dtype=torch.long)
    ts = torch.cumsum(torch.randint(1, 100000, (B, L), generator=g), dim=1)
    for b, p in enumerate(pads):
        seq[b, :p] = 0
        tok[b, :p] = 0
        ts[b, :p] = 0
        if user_slot and p < L:
            seq[b, p] = int(torch.randint(1, U + 1, (1,), generator=g))
            tok[b, p] = 2
    # dt edge cases: a zero interval, a huge interval (bin clamp is never hit below e^511; keeps log large)
    ts[0, -1] = ts[0, -2]
    ts[-1, -1] = ts[-1, -2] + 10 ** 12
    act = torch.randint(0, 6, (B, L), generator=g) * (seq != 0)
    dwell = torch.randint(0, 7, (B, L), generator=g) * (seq != 0)
    pos = torch.randint(1, I + 1, (B, L), generator=g) * (seq != 0)
    neg = torch.randint(1, I + 1, (B, L), generator=g) * (seq != 0)
    next_tok = torch.ones(B, L, dtype=torch.long) * (seq != 0)
    return dict(seq=seq, tok=tok, ts=ts, act=act, dwell=dwell, pos=pos, neg=neg, next_tok=next_tok)


def g8_log2feats(store, keys_out):
    for d, L, B, U, I in ((16, 30, 4, 40, 200), (64, 20, 3, 30, 120)):
        m, a = build_ref_model(U, I, d, L, 0.2, seed=800 + d)
        m.eval()
        bt = synth_batch(B, L, U, I, pads=(0, 3, L - 2, L - 1)[:B], seed=810 + d)
        feat = [[] for _ in range(B)]
        with torch.no_grad():
            out = m.log2feats(bt["seq"], feat, bt["tok"], bt["ts"], bt["act"], bt["dwell"])
            ue = m.get_user_embedding(bt["seq"], feat, bt["tok"], bt["ts"], bt["act"], bt["dwell"])
        prefix = f"G8_log2feats_d{d}"
        put_params(store, prefix, m)
        put(store, prefix, out=out, user_emb=ue, U=U, I=I, d=d, L=L, **bt)
        if d == 64:
            keys_out.update({k: list(v.shape) for k, v in m.state_dict().items()})


def g9_topk(store):
    """Seeds are searched until every consecutive score gap inside the top-51 / top-11 exceeds 3e-4, i.e. more
    than 100x the fp32 rounding noise of a 16-term dot product with |score| ~ 16: the ids are then unambiguous."""
    B, I, d, L = 6, 3000, 16, 9
    for seed in range(900, 5000):
        torch.manual_seed(seed)
        table = torch.randn(I + 1, d)
        table[0] = 0
        users = torch.randn(B, d)
        pos = torch.randint(1, I + 1, (B, L))
        hist = torch.randint(0, I + 1, (B, L))
        full = users @ table.t()
        # make sure some excluded ids would otherwise be in the top-50
        top = full.topk(5, dim=1).indices
        pos[:, 0] = top[:, 0]
        hist[:, 0] = top[:, 2]
        masked = full.clone()
        for b in range(B):
            masked[b, pos[b]] = float("-inf")
            masked[b, hist[b]] = float("-inf")
        gap50 = (-masked.topk(51, dim=1).values.diff(dim=1)).min()
        gap10 = (-full.topk(11, dim=1).values.diff(dim=1)).min()
        if gap50 > 3e-4 and gap10 > 3e-4:
            break
    else:
        raise RuntimeError("no tie-free seed found")
    miner = ref.HardNegativeMiner(None, table, top_k=50)
    ids50 = miner.mine_hard_negatives_similarity(users, positive_items=pos, exclude_items=hist)
    val10, ids10 = torch.topk(users @ table.t(), 10, dim=1)
    put(store, "G9_topk", table=table, users=users, pos=pos, hist=hist, ids50=ids50, ids10=ids10, val10=val10,
        min_gap50=gap50, min_gap10=gap10, seed=seed)


def train_step_case(store, prefix, m, bt, B, put_model=True, block_dropout=None):
    """One Hydra.forward + backward of the reference model `m` (train mode) on batch `bt`; records the loss, every
    gradient, the reference's own torch.randint outputs and its mined ids (data, in call order: torch.randint is
    wrapped for the duration of the forward pass, which does not touch the generator's stream).
    block_dropout = p: every HydraBlock's nn.Dropout is replaced for this call by a module that multiplies by a
    recorded multiplier tensor (Bernoulli(1 - p) / (1 - p), exactly what nn.Dropout applies; drawn from a separate
    generator) so that the step can be repeated elsewhere on the same multipliers; the attention-weight dropout inside
    nn.MultiheadAttention cannot be recorded and is switched off for such a case."""
    feat = [[] for _ in range(B)]
    draws, mined, masks = [], [], {}
    real_randint, real_mine = torch.randint, ref.HardNegativeMiner.mine_hard_negatives_similarity

    def rec_randint(*a_, **k_):
        out = real_randint(*a_, **k_)
        draws.append(out.clone())
        return out

    def rec_mine(self, *a_, **k_):
        out = real_mine(self, *a_, **k_)
        mined.append(out.clone())
        return out

    class RecordedDropout(torch.nn.Module):
        def __init__(self, p, name, gen):
            super().__init__()
            self.p, self.name, self.gen = p, name, gen

        def forward(self, x):
            mult = (torch.rand(x.shape, generator=self.gen) >= self.p).float() / (1.0 - self.p)
            masks[self.name] = mult
            return x * mult

    swapped = []
    if block_dropout is not None:
        gen = torch.Generator().manual_seed(77)
        for name, mod in m.named_modules():
            if isinstance(getattr(mod, "dropout", None), torch.nn.Dropout) and hasattr(mod, "multiheadMamba"):
                swapped.append((mod, mod.dropout))
                mod.dropout = RecordedDropout(block_dropout, name, gen)
        mha_p, m.multi_scale_encoder.cross_attention.dropout = m.multi_scale_encoder.cross_attention.dropout, 0.0
    m.zero_grad(set_to_none=True)
    torch.manual_seed(4242)
    torch.randint, ref.HardNegativeMiner.mine_hard_negatives_similarity = rec_randint, rec_mine
    try:
        with quiet():
            loss = m(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat,
                     bt["ts"], bt["act"], bt["dwell"])
    finally:
        torch.randint, ref.HardNegativeMiner.mine_hard_negatives_similarity = real_randint, real_mine
        for mod, dr in swapped:
            mod.dropout = dr
        if block_dropout is not None:
            m.multi_scale_encoder.cross_attention.dropout = mha_p
    loss.backward()
    if put_model:
        put_params(store, prefix, m)
    put(store, prefix, loss=loss, rng_seed=4242, mined_ids=mined[0], n_draws=len(draws), **bt)
    for i, dr in enumerate(draws):
        store[f"{prefix}/draw{i}"] = dr.numpy()
    for name, mult in masks.items():
        store[f"{prefix}/dropmask/{name}"] = (mult != 0).numpy()          # multiplier = mask / (1 - p)
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        if k.endswith(("item_emb.weight", "user_emb.weight")):
            rows = g.abs().sum(1).nonzero().flatten()
            store[f"{prefix}/grad_rows/{k}"] = rows.numpy()
            store[f"{prefix}/grad_vals/{k}"] = g[rows].numpy()
            assert float(g[0].abs().sum()) == 0.0
        else:
            store[f"{prefix}/grad/{k}"] = g.numpy()


def g10_train(store):
    for name, (B, L, U, I) in {"small": (3, 12, 20, 400), "chunky": (8, 60, 30, 500)}.items():
        d = 16
        m, a = build_ref_model(U, I, d, L, 0.0, seed=1000 + B)
        m.train()
        bt = synth_batch(B, L, U, I, pads=(0, 2, L - 3, 5, 0, 0, 1, 7)[:B], seed=1010 + B)
        prefix = f"G10_train_{name}"
        train_step_case(store, prefix, m, bt, B)
        put(store, prefix, U=U, I=I, d=d, L=L)


def g14_train_d64(store):
    """Train steps at d = 64 (head_dim 32, inner width 128: the width of the benchmark model): one model, the same
    batch, once without dropout and once with the blocks' dropout (p = 0.2) applied through recorded multipliers."""
    B, L, U, I, d = 4, 24, 30, 600, 64
    m, a = build_ref_model(U, I, d, L, 0.0, seed=1404)
    m.train()
    bt = synth_batch(B, L, U, I, pads=(0, 2, L - 3, 5), seed=1414)
    train_step_case(store, "G14_train_d64", m, bt, B)
    put(store, "G14_train_d64", U=U, I=I, d=d, L=L)
    train_step_case(store, "G14_train_d64_drop", m, bt, B, put_model=False, block_dropout=0.2)
    put(store, "G14_train_d64_drop", U=U, I=I, d=d, L=L, p_drop=0.2)


def g11_item_embs(store):
    """Candidate-table generation (embedding.py:654-735) in the ids-only configuration."""
    m, a = build_ref_model(20, 60, 16, 10, 0.0, seed=21)
    se = m.shared_embeddings
    m.eval()
    with torch.no_grad(), quiet(), contextlib.redirect_stderr(io.StringIO()):
        full = se.get_item_embeddings({}, batch_size=16)
        saved = []
        _stub.save_emb = lambda arr, path: saved.append((os.path.basename(str(path)), np.array(arr)))
        ref_emb.save_emb = _stub.save_emb
        item_ids = [5, 1, 60, 33, 2, 17, 9]
        retrieval_ids = [105, 101, 160, 133, 102, 117, 109]
        se.save_item_embeddings(item_ids, retrieval_ids, {i: {} for i in range(len(item_ids))}, "/nonexistent", batch_size=3)
        seq = torch.tensor([[3, 0, 7], [60, 1, 1]])
        late = se.get_complete_embeddings(seq, [[{}] * 3] * 2, include_user=False, fusion_style="late")
        early = se.get_complete_embeddings(seq, [[{}] * 3] * 2, include_user=False, fusion_style="early")
    put(store, "G11", item_w=se.item_emb.weight, full=full, seq=seq, late=late, early=early,
        item_ids=torch.tensor(item_ids), retrieval_ids=torch.tensor(retrieval_ids))
    for name, arr in saved:
        store["G11/saved/" + name] = arr


FEATURE_TYPES_MIX = {"user_sparse": ["103", "104"], "item_sparse": ["100", "117"], "user_array": ["106"],
                     "item_array": ["150"], "user_continual": ["200"], "item_continual": ["201"],
                     "item_emb": ["81", "82"]}
FEATURE_STATS_MIX = {"103": 9, "104": 5, "100": 11, "117": 7, "106": 13, "150": 6}
FEATURE_TYPES_TRAIN = {"user_sparse": ["103"], "item_sparse": ["100"], "user_array": ["106"], "item_array": [],
                       "user_continual": [], "item_continual": ["201"], "item_emb": []}
FEATURE_STATS_TRAIN = {"103": 9, "100": 11, "106": 13}
EMB_DIMS = {"81": 32, "82": 1024}


def synth_features(B, L, ftypes, stats, interaction, seed, lengths=None, p_have=0.8):
    """The reference's feature format: per sequence a list of per-position dicts {feature id: value}; a feature is
    present at a position with probability p_have."""
    rng = np.random.RandomState(seed)
    out = []
    for b in range(B):
        seq = []
        for _ in range(lengths[b] if lengths else L):
            it = {}
            for k in ftypes["user_sparse"] + ftypes["item_sparse"]:
                if rng.rand() < p_have:
                    it[k] = int(rng.randint(0, stats[k] + 1))
            for k in ftypes["user_array"] + ftypes["item_array"]:
                if rng.rand() < p_have:
                    it[k] = [int(v) for v in rng.randint(0, stats[k] + 1, size=rng.randint(1, 5))]
            for k in ftypes["user_continual"] + ftypes["item_continual"]:
                if rng.rand() < p_have:
                    it[k] = float(np.float32(rng.randn()))
            for k in ftypes["item_emb"]:
                if rng.rand() < p_have:
                    it[k] = [float(v) for v in rng.randn(EMB_DIMS[k]).astype(np.float32)]
            for fid in interaction:
                if rng.rand() < p_have:
                    it[f"interaction_user_{fid}"] = int(rng.randint(0, 64))
            seq.append(it)
        out.append(seq)
    return out


def grads_of(store, prefix, module):
    for k, p in module.named_parameters():
        if p.grad is not None:
            store[f"{prefix}/grad/{k}"] = p.grad.detach().numpy().copy()
    module.zero_grad(set_to_none=True)


def g13_features(store):
    """Feature-fusion paths of embedding.py (:216-264, :486-652) with every feature group populated, and a full
    train step of hydra3.py with sparse / array / continual features (no modality encoders: their dropout is not
    configurable)."""
    d, L, B, U, I = 16, 6, 3, 12, 40
    a = make_args(d, L, dropout=0.0)
    torch.manual_seed(1300)
    inter = {"7": {}}
    with quiet():
        se = ref_emb.SharedEmbeddingModule(U, I, FEATURE_STATS_MIX, FEATURE_TYPES_MIX, a, inter)
    xavier_(se)
    se.sparse_emb["interaction_user_7"].weight.data[64:] = 0      # 50,001-row table (embedding.py:267): keeps the file small
    se.eval()
    feats = synth_features(B, L, FEATURE_TYPES_MIX, FEATURE_STATS_MIX, inter, seed=1301, lengths=(L, L - 1, L))
    bt = synth_batch(B, L, U, I, pads=(0, 2, 0), seed=1302)
    poss = torch.arange(1, L + 1).unsqueeze(0).expand(B, -1) * (bt["seq"] != 0)
    put_params(store, "G13_mix", se)
    put(store, "G13_mix", seq=bt["seq"], tok=bt["tok"], poss=poss, U=U, I=I, d=d, L=L,
        feats_json=np.array(json.dumps(feats)), feat_types_json=np.array(json.dumps(FEATURE_TYPES_MIX)),
        feat_stats_json=np.array(json.dumps(FEATURE_STATS_MIX)), interaction_json=np.array(json.dumps(inter)))
    runs = {"early_user": lambda: se.get_complete_embeddings(bt["seq"], feats, mask=bt["tok"], include_user=True,
                                                             fusion_style="early", model_style="hydra", pos_seq=poss),
            "early_item": lambda: se.get_complete_embeddings(bt["pos"], feats, mask=bt["tok"], include_user=False,
                                                             fusion_style="early", model_style="hydra"),
            "late_item": lambda: se.get_complete_embeddings(bt["pos"], feats, include_user=False, fusion_style="late"),
            "late_user": lambda: se.get_feature_embeddings(feats, include_user=True, fusion_style="late"),
            "gated_user": lambda: se.get_feature_embeddings(feats, include_user=True, fusion_style="gated")}
    put(store, "G13_mix", pos=bt["pos"])
    for name, fn in runs.items():
        torch.manual_seed(1310)
        with quiet():
            y = fn()
        G = torch.randn_like(y)
        y.backward(G)
        put(store, f"G13_mix/{name}", y=y, G=G)
        grads_of(store, f"G13_mix/{name}", se)
    # candidate table with features (embedding.py:686-735) and the saved files (:654-684)
    with torch.no_grad(), quiet(), contextlib.redirect_stderr(io.StringIO()):
        item_feats = {i: synth_features(1, 1, FEATURE_TYPES_MIX, FEATURE_STATS_MIX, {}, seed=1400 + i)[0][0]
                      for i in range(1, I + 1) if i % 3}
        full = se.get_item_embeddings(item_feats, batch_size=16)
    put(store, "G13_mix/table", full=full, item_feats_json=np.array(json.dumps({str(k): v for k, v in item_feats.items()})))

    # ---- full train step with features -----------------------------------------------------------
    d, L, B, U, I = 16, 12, 3, 20, 400
    a = make_args(d, L, dropout=0.0)
    torch.manual_seed(1350)
    with quiet():
        se = ref_emb.SharedEmbeddingModule(U, I, FEATURE_STATS_TRAIN, FEATURE_TYPES_TRAIN, a, None)
        m = ref.Hydra(U, I, FEATURE_STATS_TRAIN, FEATURE_TYPES_TRAIN, a, se)
    for _, p in m.named_parameters():
        try:
            torch.nn.init.xavier_normal_(p.data)
        except Exception:
            pass
    for t in (se.pos_emb, se.item_emb, se.user_emb, se.action_emb):
        t.weight.data[0, :] = 0
    m.train()
    bt = synth_batch(B, L, U, I, pads=(0, 2, L - 3), seed=1351)
    f_seq = synth_features(B, L, FEATURE_TYPES_TRAIN, FEATURE_STATS_TRAIN, {}, seed=1352)
    f_pos = synth_features(B, L, FEATURE_TYPES_TRAIN, FEATURE_STATS_TRAIN, {}, seed=1353)
    f_neg = synth_features(B, L, FEATURE_TYPES_TRAIN, FEATURE_STATS_TRAIN, {}, seed=1354)
    draws, mined = [], []
    real_randint, real_mine = torch.randint, ref.HardNegativeMiner.mine_hard_negatives_similarity

    def rec_randint(*a_, **k_):
        out = real_randint(*a_, **k_)
        draws.append(out.clone())
        return out

    def rec_mine(self, *a_, **k_):
        out = real_mine(self, *a_, **k_)
        mined.append(out.clone())
        return out

    torch.manual_seed(4343)
    torch.randint, ref.HardNegativeMiner.mine_hard_negatives_similarity = rec_randint, rec_mine
    try:
        with quiet():
            loss = m(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, f_seq, f_pos, f_neg,
                     bt["ts"], bt["act"], bt["dwell"])
    finally:
        torch.randint, ref.HardNegativeMiner.mine_hard_negatives_similarity = real_randint, real_mine
    loss.backward()
    prefix = "G13_train"
    put_params(store, prefix, m)
    put(store, prefix, loss=loss, rng_seed=4343, U=U, I=I, d=d, L=L, mined_ids=mined[0], n_draws=len(draws),
        f_seq_json=np.array(json.dumps(f_seq)), f_pos_json=np.array(json.dumps(f_pos)),
        f_neg_json=np.array(json.dumps(f_neg)), feat_types_json=np.array(json.dumps(FEATURE_TYPES_TRAIN)),
        feat_stats_json=np.array(json.dumps(FEATURE_STATS_TRAIN)), **bt)
    for i, dr in enumerate(draws):
        store[f"{prefix}/draw{i}"] = dr.numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            store[f"{prefix}/grad/{k}"] = p.grad.detach().numpy()


def main():
    only = set(sys.argv[1:])
    groups = {"g1_pscan": g1_pscan, "g2_selective_scan": g2_selective_scan, "g3_rmsnorm": g3_rmsnorm,
              "g4_mamba": g4_mamba, "g5_mhm": g5_mhm, "g6_blocks": g6_blocks, "g7_multiscale": g7_multiscale,
              "g9_topk": g9_topk, "g10_train": g10_train, "g11_item_embs": g11_item_embs, "g12_d128": g12_d128,
              "g13_features": g13_features, "g14_train_d64": g14_train_d64}
    for fname, fn in groups.items():
        if only and fname not in only:
            continue
        store = {}
        fn(store)
        np.savez_compressed(os.path.join(OUT, fname + ".npz"), **store)
        print(fname, len(store), "arrays")
    if only and "g8_log2feats" not in only:
        return
    store, keys = {}, {}
    g8_log2feats(store, keys)
    np.savez_compressed(os.path.join(OUT, "g8_log2feats.npz"), **store)
    with open(os.path.join(OUT, "state_dict_keys_d64_L20.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("g8_log2feats", len(store), "arrays;", len(keys), "state_dict keys")


if __name__ == "__main__":
    main()
