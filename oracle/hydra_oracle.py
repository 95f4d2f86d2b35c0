"""CPU oracle for the Hydra hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch fp32 restatement of the reference algorithm, written from the
math (sequential recurrence instead of the reference's Blelloch sweep, explicit
attention instead of nn.MultiheadAttention.forward).  Only tests/, the smoke
check and the CPU-baseline leg of the benchmark may import this file; the
product package never does.

Parity is pinned: tests/test_oracle_golden.py checks every function here against
golden vectors captured by running the reference's own hydra3.py / embedding.py
on CPU (tests/golden/make_golden.py).

Module, parameter and argument names follow the reference so that a reference
state_dict loads with strict=True.
Reference citations are file:line into the upstream repository.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

D_STATE = 16   # hydra3.py:593,603,612
DT_RANK = 16   # hydra3.py:591,601,610
D_CONV = 4     # hydra3.py:233


# ----------------------------------------------------------------------------
# A5: scan  h[t] = a[t] * h[t-1] + x[t],  h[-1] = 0      (hydra3.py:16-104)
# ----------------------------------------------------------------------------
def pscan(A, X):
    """A, X: (B, L, ED, N) -> H (B, L, ED, N).  Same contract as hydra3.py:104."""
    h = torch.zeros_like(X[:, 0])
    out = []
    for t in range(X.shape[1]):
        h = A[:, t] * h + X[:, t]
        out.append(h)
    return torch.stack(out, dim=1)


def selective_scan(x, delta, A, B, C, D):
    """hydra3.py:107-123.  x, delta (B,L,ED); A (ED,N); B, C (B,L,N); D (ED)."""
    dA = torch.exp(delta.unsqueeze(-1) * A)
    bx = (delta.unsqueeze(-1) * B.unsqueeze(2)) * x.unsqueeze(-1)
    hs = pscan(dA, bx)
    y = (hs * C.unsqueeze(2)).sum(-1)
    return y + D * x


class SSM(nn.Module):
    """hydra3.py:158-218."""

    def __init__(self, in_features, dt_rank, dim_inner, d_state):
        super().__init__()
        self.dt_rank, self.dim_inner, self.d_state = dt_rank, dim_inner, d_state
        self.deltaBC_layer = nn.Linear(in_features, dt_rank + 2 * d_state, bias=False)
        self.dt_proj_layer = nn.Linear(dt_rank, dim_inner, bias=True)
        self.A_log = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32)).repeat(dim_inner, 1))
        self.D = nn.Parameter(torch.ones(dim_inner))

    def forward(self, x, pscan=True):
        A = -torch.exp(self.A_log.float())
        dbc = self.deltaBC_layer(x)
        dr, Bm, Cm = dbc.split([self.dt_rank, self.d_state, self.d_state], dim=-1)
        delta = F.softplus(self.dt_proj_layer(dr))
        return selective_scan(x, delta, A, Bm, Cm, self.D.float())


class RMSNorm(nn.Module):
    """hydra3.py:221-229 -- eps sits inside the square root."""

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.scale = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return x / torch.sqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.scale


class MambaBlock(nn.Module):
    """hydra3.py:232-277."""

    def __init__(self, dim, dt_rank, dim_inner, d_state, d_conv=D_CONV):
        super().__init__()
        self.dim, self.dim_inner, self.d_conv = dim, dim_inner, d_conv
        self.in_proj = nn.Linear(dim, 2 * dim_inner, bias=False)
        self.conv1d = nn.Conv1d(dim_inner, dim_inner, d_conv, groups=dim_inner, padding=d_conv - 1, bias=True)
        self.ssm = SSM(dim_inner, dt_rank, dim_inner, d_state)
        self.out_proj = nn.Linear(dim_inner, dim, bias=False)

    def forward(self, x):
        L = x.shape[1]
        xi, z = self.in_proj(x).chunk(2, dim=-1)
        # causal depthwise conv: out[t] = b + sum_j w[j] * x[t - (K-1) + j]
        w = self.conv1d.weight[:, 0, :]                       # (ED, K)
        xp = F.pad(xi, (0, 0, self.d_conv - 1, 0))            # left-pad time
        xc = self.conv1d.bias + sum(w[:, j] * xp[:, j:j + L] for j in range(self.d_conv))
        y = self.ssm(F.silu(xc))
        return self.out_proj(y * F.silu(z))


def rope_tables(L, head_dim, base=10000.0):
    """hydra3.py:300-330: angle[t, i] = t * base**(-2i/head_dim), t = 0..L-1."""
    theta = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = torch.arange(L, dtype=torch.float32)[:, None] * theta[None, :]
    return torch.cos(ang), torch.sin(ang)


def apply_rope(x, cos, sin):
    """x (B,L,H,hd); rotates interleaved pairs (2i, 2i+1)  (hydra3.py:332-357)."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    out = torch.empty_like(x)
    out[..., 0::2] = xe * c - xo * s
    out[..., 1::2] = xe * s + xo * c
    return out


class MultiHeadMamba(nn.Module):
    """hydra3.py:280-389."""

    def __init__(self, dim, dt_rank, dim_inner, d_state, args):
        super().__init__()
        self.H = args.hydra_num_heads
        self.head_dim = dim // self.H
        assert self.head_dim % 2 == 0, "Head dimension must be even for RoPE"   # hydra3.py:315
        self.mamba = nn.ModuleList(
            [MambaBlock(self.head_dim, dt_rank, 4 * self.head_dim, d_state) for _ in range(self.H)])
        self.linear_layer1 = nn.Linear(dim, dim)
        self.linear_layer2 = nn.Linear(dim, dim)

    def forward(self, x, x_org, padding_mask):
        B, L, d = x.shape
        h1 = x * padding_mask if padding_mask is not None else x
        s = F.silu(self.linear_layer1(h1))
        heads = s.view(B, L, self.H, self.head_dim)
        y1 = torch.cat([m(heads[:, :, i]) for i, m in enumerate(self.mamba)], dim=-1)
        cos, sin = rope_tables(L, self.head_dim)
        y2 = apply_rope(heads, cos, sin).reshape(B, L, d)
        return x_org + self.linear_layer2(y1 * y2 / math.sqrt(self.H))


class _HydraBlockBase(nn.Module):
    def __init__(self, dim, dt_rank, dim_inner, d_state, args):
        super().__init__()
        self.multiheadMamba = MultiHeadMamba(dim, dt_rank, dim_inner, d_state, args)
        self.gate = nn.Linear(dim, dim, bias=False)
        self.up = nn.Linear(dim, dim, bias=False)
        self.down = nn.Linear(dim, dim, bias=False)
        self.dropout = nn.Dropout(p=args.hydra_dropout_rate)
        self.norm1 = RMSNorm(dim)
        self.norm2 = RMSNorm(dim)

    def _ffn(self, n2):
        return self.down(F.silu(self.gate(n2)) * self.up(n2))


class HydraBlockFirst(_HydraBlockBase):
    """hydra3.py:392-423: residuals are taken from the *normalised* tensors."""

    def forward(self, x, padding_mask):
        n = self.norm1(self.dropout(x))
        m = n + self.multiheadMamba(n, x, padding_mask)
        n2 = self.norm2(m)
        out = self._ffn(n2) + n2
        return out * padding_mask if padding_mask is not None else out


class HydraBlock(_HydraBlockBase):
    """hydra3.py:426-457: x enters the mixer output twice (x + (x_org + ...))."""

    def forward(self, x, padding_mask):
        n = self.norm1(self.dropout(x))
        m = x + self.multiheadMamba(n, x, padding_mask)
        out = self._ffn(self.norm2(m)) + m
        return out * padding_mask if padding_mask is not None else out


class MultiScaleUserEncoder(nn.Module):
    """hydra3.py:460-548."""

    def __init__(self, dim, dt_rank, dim_inner, d_state, args):
        super().__init__()
        self.H = args.hydra_num_heads
        self.short_term_encoder = HydraBlock(dim, dt_rank, dim_inner, d_state, args)
        self.mid_term_encoder = HydraBlock(dim, dt_rank, dim_inner, d_state, args)
        self.long_term_encoder = HydraBlock(dim, dt_rank, dim_inner, d_state, args)
        # parameter container only; the attention itself is spelled out in forward()
        self.cross_attention = nn.MultiheadAttention(dim, self.H, dropout=args.hydra_dropout_rate, batch_first=True)
        self.fusion_gate = nn.Sequential(nn.Linear(3 * dim, dim), nn.Sigmoid())
        self.output_proj = nn.Linear(dim, dim)
        self.norm = RMSNorm(dim)

    def _one_query_attention(self, q_in, kv_in):
        """q_in (B,1,d), kv_in (B,L,d): H-head softmax attention, no key mask."""
        B, L, d = kv_in.shape
        hd = d // self.H
        Wq, Wk, Wv = self.cross_attention.in_proj_weight.chunk(3, dim=0)
        bq, bk, bv = self.cross_attention.in_proj_bias.chunk(3, dim=0)
        q = F.linear(q_in, Wq, bq).view(B, 1, self.H, hd).transpose(1, 2)
        k = F.linear(kv_in, Wk, bk).view(B, L, self.H, hd).transpose(1, 2)
        v = F.linear(kv_in, Wv, bv).view(B, L, self.H, hd).transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
        if self.training and self.cross_attention.dropout > 0:
            p = F.dropout(p, self.cross_attention.dropout)
        o = (p @ v).transpose(1, 2).reshape(B, 1, d)
        return self.cross_attention.out_proj(o)

    def forward(self, sequence, padding_mask=None):
        L = sequence.shape[1]
        sw, mw = min(10, L), min(25, L)
        sm = padding_mask[:, -sw:] if padding_mask is not None else None
        mm = padding_mask[:, -mw:] if padding_mask is not None else None
        short = self.short_term_encoder(sequence[:, -sw:], sm)
        mid = self.mid_term_encoder(sequence[:, -mw:], mm)
        long_ = self.long_term_encoder(sequence, padding_mask)
        sf, mf, lf = short[:, -1:], mid[:, -1:], long_[:, -1:]
        att = self._one_query_attention(sf, long_)
        g = self.fusion_gate(torch.cat([sf, mf, lf], dim=-1))
        fused = self.output_proj(self.norm(g * att + (1 - g) * lf))
        return long_ + 0.5 * fused


class PositionAwareEncoder(nn.Module):
    """hydra3.py:1044-1082 -- constructed but never called; kept for state_dict keys."""

    def __init__(self, hidden_dim, max_len=512):
        super().__init__()
        self.position_weights = nn.Parameter(torch.ones(max_len))
        self.position_attention = nn.MultiheadAttention(hidden_dim, 2, dropout=0.2, batch_first=True)


class AdaptiveHardNegativeSampler(nn.Module):
    """hydra3.py:1024-1041."""

    def __init__(self, initial_ratio=0.3, max_ratio=0.7):
        super().__init__()
        self.initial_ratio, self.max_ratio = initial_ratio, max_ratio
        self.register_buffer('training_step', torch.tensor(0.0))


# ----------------------------------------------------------------------------
# A1: tables + routed gather   (embedding.py:157-164, 429-454, 613-652)
# ----------------------------------------------------------------------------
EMB_DIMS = {"81": 32, "82": 1024, "83": 3584, "84": 4096, "85": 3584, "86": 3584}       # embedding.py:279


def modality_encoder(input_dim, h, kind):
    """embedding.py:97-136 as a bare nn.Sequential (registered under `.encoder` by the caller)."""
    if kind == "visual":
        layers = [nn.Linear(input_dim, 2 * h), nn.LayerNorm(2 * h), nn.GELU(), nn.Dropout(0.2), nn.Linear(2 * h, h),
                  nn.LayerNorm(h), nn.GELU(), nn.Linear(h, h)]
    elif kind == "categorical":
        layers = [nn.Linear(input_dim, h), nn.LayerNorm(h), nn.ReLU(), nn.Dropout(0.2), nn.Linear(h, h)]
    else:
        layers = [nn.Linear(input_dim, h), nn.LayerNorm(h), nn.ReLU(), nn.Linear(h, h)]
    holder = nn.Module()
    holder.encoder = nn.Sequential(*layers)
    return holder


class CrossModalFusion(nn.Module):
    """embedding.py:12-89."""

    def __init__(self, d, n, heads):
        super().__init__()
        self.n = n
        self.cross_attentions = nn.ModuleDict()
        self.fusion_gates = nn.ModuleList()
        if n > 1:
            for i in range(n):
                self.fusion_gates.append(nn.Sequential(nn.Linear(2 * d, d), nn.Sigmoid()))
                for j in range(n):
                    if i != j:
                        self.cross_attentions[f"{i}_{j}"] = nn.MultiheadAttention(d, heads, dropout=0.3, batch_first=True)
        self.norm = nn.LayerNorm(d)

    def forward(self, feats):
        if self.n <= 1:
            return feats[0] if feats else None
        out = 0
        for i, src in enumerate(feats):
            got = [self.cross_attentions[f"{i}_{j}"](src, tgt, tgt)[0] for j, tgt in enumerate(feats) if j != i]
            infl = sum(got) / len(got)
            out = out + src + self.fusion_gates[i](torch.cat([src, infl], dim=-1)) * infl
        return self.norm(out)


class SharedEmbeddingModule(nn.Module):
    """embedding.py:142-214, 216-264, 397-652: id tables plus the feature groups in the three fusion styles."""

    def __init__(self, user_num, item_num, feat_statistics, feat_types, args, interaction_vocab=None):
        super().__init__()
        d = args.shared_hidden_units
        self.args = args
        self.item_emb = nn.Embedding(item_num + 1, d, padding_idx=0)
        self.user_emb = nn.Embedding(user_num + 1, d, padding_idx=0)
        self.pos_emb = nn.Embedding(2 * args.maxlen + 1, d, padding_idx=0)
        self.action_emb = nn.Embedding(args.num_actions + 1, d, padding_idx=0)
        self.time_span_emb = nn.Embedding(getattr(args, 'time_span', 512), d, padding_idx=0)
        self.dwell_time_emb = nn.Embedding(7, d, padding_idx=0)
        ft = lambda k: list(feat_types.get(k, []))
        self.groups = {      # name -> (kind, keys); insertion order = embedding.py:465-477
            'item_sparse': ('sparse', ft('item_sparse')), 'item_array': ('array', ft('item_array')),
            'item_continual': ('continual', ft('item_continual')), 'item_emb': ('dense', ft('item_emb')),
            'user_sparse': ('sparse', ft('user_sparse')), 'user_array': ('array', ft('user_array')),
            'user_continual': ('continual', ft('user_continual')),
            'interaction_sparse': ('sparse', [f'interaction_user_{f}' for f in (interaction_vocab or {})])}
        vocab = dict(feat_statistics)
        vocab.update({k: 50000 for k in self.groups['interaction_sparse'][1]})        # embedding.py:267,277
        self.sparse_emb, self.emb_transform = nn.ModuleDict(), nn.ModuleDict()
        self.feature_projections, self.feature_gates = nn.ModuleDict(), nn.ModuleDict()
        for name, (kind, keys) in self.groups.items():
            for k in keys:
                if kind in ('sparse', 'array'):
                    self.sparse_emb[k] = nn.Embedding(vocab[k] + 1, d, padding_idx=0)
                    self.feature_projections[k] = nn.Linear(d, d)
                elif kind == 'continual':
                    self.feature_projections[k] = nn.Linear(1, d)
                else:
                    self.emb_transform[k] = modality_encoder(EMB_DIMS[k], d, "categorical" if k == "81" else "visual")
                self.feature_gates[k] = nn.Sequential(nn.Linear(d, d), nn.Sigmoid())
        n_dense = len(self.groups['item_emb'][1])
        self.cross_modal_fusion = CrossModalFusion(d, n_dense, 2)                    # embedding.py:181-185
        n_item = sum(len(self.groups[g][1]) for g in ('item_sparse', 'item_array', 'item_continual')) + (n_dense > 0)
        n_user = sum(len(self.groups[g][1]) for g in ('user_sparse', 'user_array', 'user_continual',
                                                      'interaction_sparse'))
        self.early_fusion_projection_no_user = nn.Linear(n_item * d, d) if n_item else nn.Identity()
        self.early_fusion_projection_with_user = nn.Linear((n_item + n_user) * d, d) if n_item + n_user else nn.Identity()

    def get_base_embeddings(self, seq, mask=None, include_user=False):
        """embedding.py:429-454."""
        if include_user and mask is not None:
            item_rows = self.item_emb(torch.where(mask == 1, seq, torch.zeros_like(seq)))
            user_rows = self.user_emb(torch.where(mask == 2, seq, torch.zeros_like(seq)))
            return item_rows * (mask == 1).unsqueeze(-1) + user_rows * (mask == 2).unsqueeze(-1)
        return self.item_emb(seq)

    # one feature of the batch as a tensor (embedding.py:397-427, :511-518)
    def _tensor(self, feats, kind, k):
        B, Lm = len(feats), max(len(s) for s in feats)
        if kind == 'array':
            present = [[it[k] for it in s if k in it] for s in feats]          # packed to the front (:411-414)
            A = max([len(v) for s in present for v in s] or [1])
            t = torch.zeros(B, Lm, A, dtype=torch.long)
            for b, s in enumerate(present):
                for j, v in enumerate(s):
                    t[b, j, :len(v)] = torch.as_tensor(v, dtype=torch.long)
            return t
        if kind == 'dense':
            t = torch.zeros(B, Lm, EMB_DIMS[k])
        else:
            t = torch.zeros(B, Lm, dtype=torch.float32 if kind == 'continual' else torch.long)
        for b, s in enumerate(feats):
            for l, it in enumerate(s):
                if k in it:
                    t[b, l] = torch.as_tensor(it[k], dtype=t.dtype)
        return t

    def _walk(self, include_user, interaction):
        names = ['item_sparse', 'item_array', 'item_continual', 'item_emb']
        names += ['user_sparse', 'user_array', 'user_continual'] if include_user else []
        names += ['interaction_sparse'] if interaction else []
        return [(self.groups[n][0], k) for n in names for k in self.groups[n][1]]

    def _raw(self, feats, kind, k):
        t = self._tensor(feats, kind, k)
        if kind == 'sparse':
            return self.sparse_emb[k](t)
        if kind == 'array':
            return self.sparse_emb[k](t).sum(2)
        return t.unsqueeze(2)

    def get_feature_embeddings(self, feats, include_user=False, fusion_style='late'):
        """embedding.py:456-611, :216-264."""
        d = self.args.shared_hidden_units
        if fusion_style != 'gated' and (len(feats) == 0 or len(feats[0]) == 0):
            return torch.zeros(len(feats), 0, d)
        B, Lm = len(feats), max(len(s) for s in feats)
        if fusion_style in ('late', 'gated'):
            total = torch.zeros(B, Lm, d)
            for kind, k in self._walk(include_user, True):
                p = self.emb_transform[k].encoder(self._tensor(feats, kind, k)) if kind == 'dense' \
                    else self.feature_projections[k](self._raw(feats, kind, k))
                total = total + (p * self.feature_gates[k](p) if fusion_style == 'gated' else p)
            return total
        pieces, dense = [], []
        for kind, k in self._walk(include_user, include_user):
            if kind == 'dense':
                dense.append(self.emb_transform[k].encoder(F.normalize(self._tensor(feats, kind, k), p=2, dim=-1)))
            elif kind == 'continual':
                pieces.append(self.feature_projections[k](self._raw(feats, kind, k)))
            else:
                pieces.append(self._raw(feats, kind, k))
        if dense:
            pieces.append(self.cross_modal_fusion(dense))
        if not pieces:
            return torch.zeros(B, Lm, d)
        proj = self.early_fusion_projection_with_user if include_user else self.early_fusion_projection_no_user
        return F.relu(proj(torch.cat(pieces, dim=2)))

    def get_complete_embeddings(self, seq, feature_array, mask=None, include_user=False,
                                fusion_style='late', model_style='hstu', pos_seq=None):
        """embedding.py:613-652 (a batch without features gets the zero block of :549-554, :636-639)."""
        base = self.get_base_embeddings(seq, mask, include_user)
        if pos_seq is not None:
            base = base + self.pos_emb(pos_seq)
        feat = self.get_feature_embeddings(feature_array, include_user, fusion_style)
        if feat.shape[1] == 0 and seq.shape[1] > 0:
            feat = torch.zeros_like(base)
        if fusion_style == 'late':
            return F.gelu(base + feat)
        return torch.cat([base, feat], dim=2)


def late_item_rows(item_weight, ids):
    """embedding.py:641-644 with the ids-only zero feature block: F.gelu(item_emb[ids] + 0)."""
    return F.gelu(item_weight[ids])


def item_embedding_table(item_weight):
    """embedding.py:686-735: row 0 zeros, row i = late-fusion embedding of item i."""
    n = item_weight.shape[0]
    return torch.cat([torch.zeros(1, item_weight.shape[1]), late_item_rows(item_weight, torch.arange(1, n))], dim=0)


def time_bins_from_timestamps(log_seqs, seq_timestamp, time_span=512):
    """hydra3.py:650-659: bin = min(floor(ln(dt + 1)), time_span-1) * (id != 0)."""
    dt = torch.zeros_like(seq_timestamp)
    dt[:, 1:] = seq_timestamp[:, 1:] - seq_timestamp[:, :-1]
    prev_pad = torch.zeros_like(log_seqs, dtype=torch.bool)
    prev_pad[:, 1:] = log_seqs[:, :-1] == 0
    dt[prev_pad] = 0
    bins = torch.clamp(torch.log(dt.float() + 1.0).long(), max=time_span - 1)
    return bins * (log_seqs != 0).long()


# ----------------------------------------------------------------------------
# A12 / A14: masked full-table scoring + top-k  (hydra3.py:984-1007)
# ----------------------------------------------------------------------------
def mine_hard_negatives_similarity(user_embeddings, item_embeddings, top_k, positive_items=None, exclude_items=None):
    """Exactly the reference's ops: mm, -inf at listed ids, topk.  Returns (ids, scores)."""
    sim = torch.mm(user_embeddings, item_embeddings.t())
    for b in range(sim.shape[0]):
        if positive_items is not None:
            sim[b, positive_items[b]] = float('-inf')
        if exclude_items is not None:
            sim[b, exclude_items[b]] = float('-inf')
    val, idx = torch.topk(sim, top_k, dim=1)
    return idx, val


class HardNegativeMiner:
    """hydra3.py:978-1007."""

    def __init__(self, model, item_embeddings, top_k=100):
        self.model, self.item_embeddings, self.top_k = model, item_embeddings, top_k

    def mine_hard_negatives_similarity(self, user_embeddings, positive_items, exclude_items=None):
        return mine_hard_negatives_similarity(user_embeddings, self.item_embeddings, self.top_k,
                                              positive_items, exclude_items)[0]


# ----------------------------------------------------------------------------
# loss (hydra3.py:817-974, 1085-1151).  Same RNG call sequence as the reference
# so that, under one torch.manual_seed on CPU, the sampled indices coincide.
# ----------------------------------------------------------------------------
def nce_rows(q, k, table, hard_ids, inb_idx, temperature=0.05, normalize=True):
    """Index form of hydra3.py:845-974: the per-row loss for GIVEN negatives.  Keys of row i are
    [k_i | table[hard_ids[i]] | k[inb_idx[i]]]; logits = <q^, key^> / temperature (hydra3.py:958-962);
    loss_i = -log_softmax(logits)[0] (hydra3.py:965-967).  compute_mixed_contrastive_loss below is this function
    applied to the indices it samples (checked in tests/test_oracle_golden.py)."""
    if normalize:
        q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    keys = [k.unsqueeze(1)]
    if hard_ids is not None and hard_ids.shape[1] > 0:
        hn = table[hard_ids]
        keys.append(F.normalize(hn, dim=-1) if normalize else hn)
    if inb_idx is not None and inb_idx.shape[1] > 0:
        keys.append(k[inb_idx])
    logits = (q.unsqueeze(1) * torch.cat(keys, dim=1)).sum(-1) / temperature
    return -F.log_softmax(logits, dim=1)[:, 0]


def compute_mixed_contrastive_loss(log_feats, pos_embs, loss_mask, item_table=None, hard_negatives=None,
                                   T=512, hard_ratio=0.3, temperature=0.05, normalize=True, q_chunk=64, device=None,
                                   record=None):
    """record: optional dict that receives the sampled indices ('hard' (M, .) item-table rows incl. the random
    fill rows of the short-batch case, 'inb' (M, .) rows of the masked positives) for use with nce_rows."""
    mask = loss_mask.bool()
    q, k = log_feats[mask], pos_embs[mask]
    M = q.shape[0]
    if M == 0:
        return torch.tensor(0.0, requires_grad=True)
    if normalize:
        q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    num_hard = int(T * hard_ratio)
    num_inb = T - num_hard
    hard_ids = hard_negatives[mask] if hard_negatives is not None else None
    rows = torch.arange(M)
    losses = []
    rec_hard, rec_inb = [], []
    for s in range(0, M, q_chunk):
        e = min(s + q_chunk, M)
        n = e - s
        negs = []
        if hard_ids is not None and num_hard > 0:
            ch = hard_ids[s:e]
            K = ch.shape[1]
            if K >= num_hard:
                ch = torch.gather(ch, 1, torch.randint(0, K, (n, num_hard)))
            else:
                ch = ch.repeat(1, (num_hard + K - 1) // K)[:, :num_hard]
            hn = item_table[ch]
            negs.append(F.normalize(hn, dim=-1) if normalize else hn)
            rec_hard.append([ch])
        else:
            rec_hard.append([])
        if num_inb > 0:
            if M - 1 >= num_inb:
                r = torch.randint(0, M - 1, (n, num_inb))
                rec_inb.append(r + (r >= rows[s:e].unsqueeze(1)).long())
                inb = k[rec_inb[-1]]
            else:
                base = torch.arange(M).unsqueeze(0).repeat(M, 1)
                allidx = base[base != torch.arange(M).unsqueeze(1)].view(M, M - 1)[s:e]
                inb = k[allidx]
                rec_inb.append(allidx)
                short = num_inb - inb.shape[1]
                if short > 0:
                    if item_table is not None:
                        rid = torch.randint(1, item_table.shape[0], (n, short))
                        re_ = item_table[rid]
                        inb = torch.cat([inb, F.normalize(re_, dim=-1) if normalize else re_], dim=1)
                        rec_hard[-1].append(rid)
                    else:
                        inb = torch.cat([inb, inb[:, -1:].expand(-1, short, -1)], dim=1)
                        rec_inb[-1] = torch.cat([allidx, allidx[:, -1:].expand(-1, short)], dim=1)
            negs.append(inb)
        keys = torch.cat([k[s:e].unsqueeze(1)] + negs, dim=1)
        logits = (q[s:e].unsqueeze(1) * keys).sum(-1) / temperature
        losses.append(-F.log_softmax(logits, dim=1)[:, 0])
    if record is not None:
        hard_parts = [torch.cat(h, dim=1) for h in rec_hard if h]
        record['hard'] = torch.cat(hard_parts, dim=0) if hard_parts else None
        record['inb'] = torch.cat(rec_inb, dim=0) if rec_inb else None
    return torch.cat(losses).mean()


class EnhancedTrainingLoss(nn.Module):
    """hydra3.py:1085-1151 with its default weights (ranking 0, contrastive 1, diversity 0)."""

    def __init__(self, args):
        super().__init__()
        self.ranking_loss_weight = getattr(args, 'ranking_loss_weight', 0.0)
        self.contrastive_loss_weight = getattr(args, 'contrastive_loss_weight', 1.0)
        self.diversity_loss_weight = getattr(args, 'diversity_loss_weight', 0.0)
        self.adaptive_hard_sampler = AdaptiveHardNegativeSampler()

    def forward(self, user_embs, pos_embs, neg_embs, hard_negatives, loss_mask, item_table):
        total = self.contrastive_loss_weight * compute_mixed_contrastive_loss(
            user_embs, pos_embs, loss_mask, item_table, hard_negatives,
            hard_ratio=0.3, T=512, temperature=0.05, q_chunk=256)
        self.adaptive_hard_sampler.training_step += 1
        if neg_embs is not None:
            u, p, n, lm = user_embs[:, -1], pos_embs[:, -1], neg_embs[:, -1], loss_mask[:, -1]
            rl = -F.logsigmoid((u * p).sum(-1) - (u * n).sum(-1))
            total = total + self.ranking_loss_weight * ((rl * lm).sum() / lm.sum())
        un = F.normalize(user_embs[:, -1], dim=1)
        sim = un @ un.t()
        div = (sim * (1 - torch.eye(sim.shape[0]))).abs().mean()
        return total + self.diversity_loss_weight * div


class Hydra(nn.Module):
    """hydra3.py:553-720 (ids-only)."""

    def __init__(self, user_num, item_num, feat_statistics, feat_types, args, shared_embeddings=None):
        super().__init__()
        self.args, self.maxlen = args, args.maxlen
        if not hasattr(args, 'shared_hidden_units'):
            args.shared_hidden_units = max(getattr(args, 'hstu_hidden_units', 60), args.hydra_hidden_units)
        self.shared_embeddings = shared_embeddings if shared_embeddings is not None else \
            SharedEmbeddingModule(user_num, item_num, feat_statistics, feat_types, args)
        d = args.hydra_hidden_units
        self.embedding_projection = nn.Linear(args.shared_hidden_units, d) \
            if args.shared_hidden_units != d else nn.Identity()
        self.hydra_projection = nn.Linear(2 * args.shared_hidden_units, d)
        if not hasattr(self.shared_embeddings, 'hydra_projection'):
            self.shared_embeddings.hydra_projection = self.hydra_projection      # hydra3.py:577-578 alias
        self.time_span_emb = nn.Embedding(getattr(args, 'time_span', 512), d, padding_idx=0)  # unused (:580)
        self.hydra = nn.ModuleList()
        self.multi_scale_encoder = MultiScaleUserEncoder(d, DT_RANK, 4 * d, D_STATE, args)
        for _ in range(args.hydra_num_blocks):
            self.hydra.append(HydraBlock(d, DT_RANK, 4 * d, D_STATE, args))
        self.hydra_first = HydraBlockFirst(d, DT_RANK, 4 * d, D_STATE, args)
        self.position_aware_encoder = PositionAwareEncoder(d, max_len=args.maxlen + 1)
        self.enchanced_loss = EnhancedTrainingLoss(args)

    def assemble_inputs(self, log_seqs, seq_feature, mask, seq_timestamp, action_type, dwell_bins):
        """hydra3.py:621-669 -> (seqs (B,L,d), padding_mask (B,L,1))."""
        padding_mask = (log_seqs != 0).unsqueeze(-1).float()
        L = log_seqs.shape[1]
        poss = torch.arange(1, L + 1).unsqueeze(0).expand_as(log_seqs) * (log_seqs != 0)
        se = self.shared_embeddings
        cat = se.get_complete_embeddings(log_seqs, seq_feature, mask=mask, include_user=True,
                                         fusion_style='early', model_style='hydra', pos_seq=poss)
        seqs = self.hydra_projection(cat) * (self.args.hydra_hidden_units ** 0.5)
        tb = time_bins_from_timestamps(log_seqs, seq_timestamp, getattr(self.args, 'time_span', 512))
        seqs = seqs + self.embedding_projection(se.time_span_emb(tb))
        seqs = seqs + self.embedding_projection(se.action_emb(action_type))
        seqs = seqs + self.embedding_projection(se.dwell_time_emb(dwell_bins))
        return seqs * padding_mask, padding_mask

    def log2feats(self, log_seqs, seq_feature, mask, seq_timestamp, action_type, dwell_bins):
        """hydra3.py:618-681."""
        seqs, pm = self.assemble_inputs(log_seqs, seq_feature, mask, seq_timestamp, action_type, dwell_bins)
        seqs = self.hydra_first(seqs, pm)
        for blk in self.hydra:
            seqs = blk(seqs, pm)
        return self.multi_scale_encoder(seqs, pm) * pm

    def get_user_embedding(self, seq, seq_feat, token_type, seq_timestamp, action_type, dwell_bins):
        """hydra3.py:808-815."""
        return self.log2feats(seq, seq_feat, token_type, seq_timestamp, action_type, dwell_bins)[:, -1, :]

    def forward(self, user_item, pos_seqs, neg_seqs, mask, next_mask, next_action_type, seq_feature, pos_feature,
                neg_feature, timestamp, action_type, dwell_bins):
        """hydra3.py:683-720."""
        log_feats = self.log2feats(user_item, seq_feature, mask, timestamp, action_type, dwell_bins)
        se = self.shared_embeddings
        table = se.item_emb.weight
        pos = se.get_complete_embeddings(pos_seqs, pos_feature, mask, False, 'early', 'hydra')
        neg = se.get_complete_embeddings(neg_seqs, neg_feature, mask, False, 'early', 'hydra')
        hard = HardNegativeMiner(self, table, top_k=50).mine_hard_negatives_similarity(
            log_feats[:, -1, :], positive_items=pos_seqs, exclude_items=user_item)
        hard = hard.unsqueeze(1).expand(-1, log_feats.shape[1], -1)
        return self.enchanced_loss(user_embs=log_feats, pos_embs=self.hydra_projection(pos),
                                   neg_embs=self.hydra_projection(neg), hard_negatives=hard,
                                   loss_mask=(next_mask == 1), item_table=table)


def make_args(d=64, L=100, heads=2, blocks=2, dropout=0.2, **kw):
    """Namespace with the fields the hot path reads (main.py:45-83 defaults otherwise)."""
    a = SimpleNamespace(device='cpu', hydra_hidden_units=d, hstu_hidden_units=d, shared_hidden_units=d,
                        hydra_num_heads=heads, hydra_num_blocks=blocks, hydra_dropout_rate=dropout,
                        maxlen=L - 1, norm_first=False, num_actions=5, time_span=512)
    a.__dict__.update(kw)
    return a


IDS_ONLY_FEAT_TYPES = {k: [] for k in ('user_sparse', 'item_sparse', 'user_array', 'item_array',
                                        'user_continual', 'item_continual', 'item_emb')}


def build_model(U, I, d=64, L=100, heads=2, blocks=2, dropout=0.2, seed=0):
    """Ids-only model, initialised as main.py:603-607,619-622 (xavier on >=2-D, zero padding rows)."""
    args = make_args(d, L, heads, blocks, dropout)
    torch.manual_seed(seed)
    se = SharedEmbeddingModule(U, I, {}, IDS_ONLY_FEAT_TYPES, args, None)
    m = Hydra(U, I, {}, IDS_ONLY_FEAT_TYPES, args, se)
    for p in m.parameters():
        if p.dim() >= 2:
            nn.init.xavier_normal_(p.data)
    for t in (se.pos_emb, se.item_emb, se.user_emb, se.action_emb):
        t.weight.data[0].zero_()
    return m
