"""Drop-in for embedding.py's SharedEmbeddingModule on the HIP row kernels.

Constructor, attribute / parameter names (`item_emb.weight`, `sparse_emb.<k>.weight`, `feature_projections.<k>`,
`early_fusion_projection_with_user`, `emb_transform.<k>.encoder.<i>`, `cross_modal_fusion...`) and method signatures
follow embedding.py:12-214, 429-454, 486-652 so that reference checkpoints load and main.py / infer.py call sites work
unchanged.  Id tables: HIP gather / deterministic scatter-add.  Feature groups (sparse, array, continual,
interaction, multimodal item embeddings) in the three fusion styles: HIP row gathers, bag sums, projections,
activations, gates and LayerNorm; the projections of one concatenated input are column blocks of the wide weight
accumulated in place, so no concatenated tensor is built.  The raw multimodal inputs (1024 to 4096 columns) enter
their modality encoder through the same projection kernel in column chunks of 256, their L2 normalisation riding as
the row scale of that product; dropout multipliers come from the library's Philox kernel.
"""
import numpy as np
import torch
from torch import nn

from . import enc_ops as E
from . import feature_ops as FO
from . import ops

EMB_SHAPE_DICT = {"81": 32, "82": 1024, "83": 3584, "84": 4096, "85": 3584, "86": 3584}     # embedding.py:279
MODALITY_KIND = {"81": "categorical", "82": "visual", "83": "visual", "84": "visual", "85": "visual", "86": "visual"}
MAX_INTERACTION_VOCAB = 50000                                                                # embedding.py:267


class HipEmbedding(nn.Module):
    """nn.Embedding-compatible table (same `weight`, `padding_idx`, state_dict key) whose forward is the HIP row
    gather and whose backward is the deterministic HIP scatter-add into a dense gradient."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)
        if padding_idx is not None:
            with torch.no_grad():
                self.weight[padding_idx].zero_()

    def forward(self, ids):
        return ops.gather_rows(self.weight, ids, self.padding_idx)

    def bag_sum(self, ids):
        """sum over the last axis of ids of the gathered rows (`self(ids).sum(-2)` in one pass)."""
        return FO.gather_sum_rows(self.weight, ids, self.padding_idx)

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}"


class SharedEmbeddingConfig:
    """embedding.py:738-749."""

    def __init__(self, args):
        self.shared_hidden_units = max(getattr(args, 'hstu_hidden_units', 60), getattr(args, 'hydra_hidden_units', 60))
        args.shared_hidden_units = self.shared_hidden_units


def _run_layers(layers, x2, training, in_scale=None):
    """An nn.Sequential of Linear / LayerNorm / GELU / ReLU / Dropout holders applied to (T, K) rows through the HIP
    ops.  in_scale (T): row scale of the input, applied inside the first projection."""
    for i, layer in enumerate(layers):
        if isinstance(layer, nn.Linear):
            x2 = FO.linear_any(x2, layer.weight, layer.bias, in_scale=in_scale if i == 0 else None)
        elif isinstance(layer, nn.LayerNorm):
            x2 = FO.layernorm(x2, layer.weight, layer.bias, layer.eps)
        elif isinstance(layer, nn.GELU):
            x2 = FO.gelu(x2)
        elif isinstance(layer, nn.ReLU):
            x2 = FO.relu(x2)
        elif isinstance(layer, nn.Dropout):
            x2 = FO.dropout(x2, layer.p, training)
        else:
            raise TypeError(f"unexpected layer {type(layer).__name__}")
    return x2


class ModalitySpecificEncoder(nn.Module):
    """embedding.py:92-139: per-modality MLP; the layer list (and so every state_dict key) is the reference's."""

    def __init__(self, input_dim, hidden_dim, modality_type="visual"):
        super().__init__()
        self.modality_type = modality_type
        h = hidden_dim
        if modality_type == "visual":
            layers = [nn.Linear(input_dim, 2 * h), nn.LayerNorm(2 * h), nn.GELU(), nn.Dropout(0.2), nn.Linear(2 * h, h),
                      nn.LayerNorm(h), nn.GELU(), nn.Linear(h, h)]
        elif modality_type == "categorical":
            layers = [nn.Linear(input_dim, h), nn.LayerNorm(h), nn.ReLU(), nn.Dropout(0.2), nn.Linear(h, h)]
        elif modality_type == "behavioral":
            layers = [nn.Linear(input_dim, h), nn.LayerNorm(h), nn.GELU(), nn.Linear(h, h)]
        else:
            layers = [nn.Linear(input_dim, h), nn.LayerNorm(h), nn.ReLU(), nn.Linear(h, h)]
        self.encoder = nn.Sequential(*layers)

    def forward(self, x, normalize_input=False):
        """normalize_input: encode x / max(|x|, 1e-12) (F.normalize over the last axis) without writing it: the row
        scale rides in the first projection."""
        x2 = x.reshape(-1, x.shape[-1])
        scale = FO.row_inv_norm(x2) if normalize_input else None
        y = _run_layers(self.encoder, x2, self.training, scale)
        return y.view(*x.shape[:-1], y.shape[-1])


def _weighted_sum(tensors, coef=1.0):
    """coef * sum(tensors) over (T, d) rows on the HIP combine kernel (four operands per launch)."""
    acc = E.combine_ad(tensors[:4], [coef] * len(tensors[:4]))
    for i in range(4, len(tensors), 3):
        more = tensors[i:i + 3]
        acc = E.combine_ad([acc] + more, [1.0] + [coef] * len(more))
    return acc


class CrossModalFusionLayer(nn.Module):
    """embedding.py:12-89: every modality attends to every other one, the mean of what it receives is added through
    a learned gate, the enhanced modalities are summed and LayerNorm-ed."""

    def __init__(self, hidden_dim, num_modalities, num_heads=4):
        super().__init__()
        self.hidden_dim, self.num_modalities, self.num_heads = hidden_dim, num_modalities, num_heads
        self.cross_attentions = nn.ModuleDict()
        self.fusion_gates = nn.ModuleList()
        if num_modalities > 1:
            for i in range(num_modalities):
                self.fusion_gates.append(nn.Sequential(nn.Linear(2 * hidden_dim, hidden_dim), nn.Sigmoid()))
                for j in range(num_modalities):
                    if i != j:
                        self.cross_attentions[f"{i}_{j}"] = nn.MultiheadAttention(hidden_dim, num_heads, dropout=0.3,
                                                                                  batch_first=True)
        self.norm = nn.LayerNorm(hidden_dim)

    def _attend(self, mha, src2, tgt2, B, L):
        """nn.MultiheadAttention(query=src, key=value=tgt) on (B * L, d) rows."""
        d, H = self.hidden_dim, self.num_heads
        q = E.linear(src2, mha.in_proj_weight[:d], mha.in_proj_bias[:d])
        kv = E.linear(tgt2, mha.in_proj_weight[d:], mha.in_proj_bias[d:])
        drop = None
        if self.training and mha.dropout > 0:      # the multiplier nn.MultiheadAttention applies to its weights
            drop = FO.dropout_multiplier((B, H, L, L), mha.dropout, q.device)
        ctxv = FO.attention(q, kv, drop, B, L, L, H)
        return E.linear(ctxv, mha.out_proj.weight, mha.out_proj.bias)

    def forward(self, modality_features):
        if self.num_modalities <= 1:
            return modality_features[0] if modality_features else None
        B, L, d = modality_features[0].shape
        rows = [m.reshape(B * L, d) for m in modality_features]
        n = len(rows)
        terms = []
        for i, src in enumerate(rows):
            got = [self._attend(self.cross_attentions[f"{i}_{j}"], src, tgt, B, L)
                   for j, tgt in enumerate(rows) if j != i]
            infl = _weighted_sum(got, 1.0 / (n - 1))                  # torch.stack(...).mean(0)
            glin = self.fusion_gates[i][0]
            # gate = sigmoid(W [src | infl] + b): the two halves of the weight applied to the two inputs
            q = E.linear(src, glin.weight, glin.bias)
            q = E.linear(infl, glin.weight, None, r1=q, c1=1.0, col_off=d)
            terms += [src, FO.sigmoid_gate(infl, q)]                  # enhanced_i = src + gate * influence
        fused = _weighted_sum(terms)
        return FO.layernorm(fused, self.norm.weight, self.norm.bias, self.norm.eps).view(B, L, d)


# ---- host-side packing of the reference's list-of-lists-of-dicts feature format --------------------------------
def _seq_len(feature_array):
    return max(len(s) for s in feature_array)


def pack_scalar_feature(feature_array, k, continual):
    """(B, Lmax) array of one sparse / continual feature: missing keys and short sequences read 0
    (embedding.py:417-427)."""
    out = np.zeros((len(feature_array), _seq_len(feature_array)), dtype=np.float32 if continual else np.int64)
    for b, seq in enumerate(feature_array):
        for l, item in enumerate(seq):
            if k in item:
                out[b, l] = item[k]
    return out


def pack_array_feature(feature_array, k):
    """(B, Lmax, Amax) id bags of one array feature.  As embedding.py:401-415 does, the bags of the positions that
    carry the feature are stored at the FRONT of the sequence axis in order (positions without it are skipped, not
    left empty), and Amax is the longest bag of the batch (1 when there is none)."""
    bags = [[item[k] for item in seq if k in item] for seq in feature_array]
    amax = max((len(v) for seq in bags for v in seq), default=1)
    out = np.zeros((len(feature_array), _seq_len(feature_array), amax), dtype=np.int64)
    for b, seq in enumerate(bags):
        for j, v in enumerate(seq):
            out[b, j, :len(v)] = v
    return out


def pack_dense_feature(feature_array, k, dim):
    """(B, Lmax, dim) fp32 block of one precomputed item-embedding feature, zeros where absent
    (embedding.py:511-516)."""
    out = np.zeros((len(feature_array), _seq_len(feature_array), dim), dtype=np.float32)
    for b, seq in enumerate(feature_array):
        for l, item in enumerate(seq):
            if k in item:
                out[b, l] = item[k]
    return out


class SharedEmbeddingModule(nn.Module):
    def __init__(self, user_num, item_num, feat_statistics, feat_types, args, interaction_vocab=None):
        super().__init__()
        if not hasattr(args, 'shared_hidden_units'):
            SharedEmbeddingConfig(args)
        self.user_num, self.item_num, self.args, self.dev = user_num, item_num, args, args.device
        self.interaction_vocab_dict = interaction_vocab if interaction_vocab is not None else {}
        d = args.shared_hidden_units
        self.item_emb = HipEmbedding(item_num + 1, d, padding_idx=0)
        self.user_emb = HipEmbedding(user_num + 1, d, padding_idx=0)
        self.pos_emb = HipEmbedding(2 * args.maxlen + 1, d, padding_idx=0)
        self.action_emb = HipEmbedding(args.num_actions + 1, d, padding_idx=0)
        self.time_span_emb = HipEmbedding(getattr(args, 'time_span', 512), d, padding_idx=0)
        self.dwell_time_emb = HipEmbedding(7, d, padding_idx=0)
        self.sparse_emb = nn.ModuleDict()
        self.emb_transform = nn.ModuleDict()
        self.feature_projections = nn.ModuleDict()

        # feature groups (embedding.py:265-280)
        ft = {k: list(feat_types.get(k, [])) for k in ('user_sparse', 'item_sparse', 'user_array', 'item_array',
                                                       'user_continual', 'item_continual', 'item_emb')}
        self.USER_SPARSE_FEAT = {k: feat_statistics[k] for k in ft['user_sparse']}
        self.USER_CONTINUAL_FEAT = ft['user_continual']
        self.ITEM_SPARSE_FEAT = {k: feat_statistics[k] for k in ft['item_sparse']}
        self.ITEM_CONTINUAL_FEAT = ft['item_continual']
        self.USER_ARRAY_FEAT = {k: feat_statistics[k] for k in ft['user_array']}
        self.ITEM_ARRAY_FEAT = {k: feat_statistics[k] for k in ft['item_array']}
        self.INTERACTION_SPARSE_FEAT = {f'interaction_user_{fid}': MAX_INTERACTION_VOCAB
                                        for fid in self.interaction_vocab_dict}
        self.ITEM_EMB_FEAT = {k: EMB_SHAPE_DICT[k] for k in ft['item_emb']}
        self.num_item_emb_features = len(ft['item_emb'])
        self.cross_modal_fusion = CrossModalFusionLayer(d, self.num_item_emb_features, num_heads=2)

        # tables, modality encoders, per-feature projections (embedding.py:282-358)
        for group in (self.USER_SPARSE_FEAT, self.ITEM_SPARSE_FEAT, self.ITEM_ARRAY_FEAT, self.USER_ARRAY_FEAT):
            for k, vocab in group.items():
                self.sparse_emb[k] = HipEmbedding(vocab + 1, d, padding_idx=0)
        for k, vocab in self.INTERACTION_SPARSE_FEAT.items():
            self.sparse_emb[k] = HipEmbedding(vocab + 1, d, padding_idx=0)
        for k, dim in self.ITEM_EMB_FEAT.items():
            self.emb_transform[k] = ModalitySpecificEncoder(dim, d, MODALITY_KIND.get(k, "generic"))
        for k in (list(self.USER_SPARSE_FEAT) + list(self.ITEM_SPARSE_FEAT) + list(self.USER_ARRAY_FEAT) +
                  list(self.ITEM_ARRAY_FEAT) + list(self.INTERACTION_SPARSE_FEAT)):
            self.feature_projections[k] = nn.Linear(d, d)
        for k in self.USER_CONTINUAL_FEAT + self.ITEM_CONTINUAL_FEAT:
            self.feature_projections[k] = nn.Linear(1, d)

        # early-fusion projections over the concatenation of all pieces (embedding.py:360-395)
        n_item = (len(self.ITEM_SPARSE_FEAT) + len(self.ITEM_ARRAY_FEAT) + len(self.ITEM_CONTINUAL_FEAT) +
                  (1 if self.ITEM_EMB_FEAT else 0))
        n_user = (len(self.USER_SPARSE_FEAT) + len(self.USER_ARRAY_FEAT) + len(self.USER_CONTINUAL_FEAT) +
                  len(self.INTERACTION_SPARSE_FEAT))
        self.early_fusion_projection_no_user = nn.Linear(n_item * d, d) if n_item else nn.Identity()
        self.early_fusion_projection_with_user = nn.Linear((n_item + n_user) * d, d) if n_item + n_user else nn.Identity()

        self.feature_gates = nn.ModuleDict()
        for k in (list(self.USER_SPARSE_FEAT) + list(self.ITEM_SPARSE_FEAT) + list(self.USER_ARRAY_FEAT) +
                  list(self.ITEM_ARRAY_FEAT) + list(self.INTERACTION_SPARSE_FEAT) + list(self.USER_CONTINUAL_FEAT) +
                  list(self.ITEM_CONTINUAL_FEAT) + list(self.ITEM_EMB_FEAT)):
            self.feature_gates[k] = nn.Sequential(nn.Linear(d, d), nn.Sigmoid())
        self.num_item_features = (len(ft['item_sparse']) + len(ft['item_array']) + len(ft['item_continual']) +
                                  len(ft['item_emb']))
        self.num_user_features = len(ft['user_sparse']) + len(ft['user_array']) + len(ft['user_continual'])

    # ---- id embeddings ------------------------------------------------------------------------
    def get_base_embeddings(self, seq, mask=None, include_user=False):
        """embedding.py:429-454."""
        if include_user and mask is not None:
            return ops.gather_routed(self.item_emb.weight, self.user_emb.weight, seq, mask)
        return self.item_emb(seq)

    # ---- features -----------------------------------------------------------------------------
    @property
    def has_features(self):
        return bool(self.feature_gates)

    def _device(self):
        return self.item_emb.weight.device

    def feat2tensor(self, seq_feature, k):
        """embedding.py:397-427: one feature of a batch as a device tensor."""
        if k in self.ITEM_ARRAY_FEAT or k in self.USER_ARRAY_FEAT:
            return torch.from_numpy(pack_array_feature(seq_feature, k)).to(self._device())
        continual = k in self.USER_CONTINUAL_FEAT or k in self.ITEM_CONTINUAL_FEAT
        return torch.from_numpy(pack_scalar_feature(seq_feature, k, continual)).to(self._device())

    def _groups(self, include_user, interaction):
        """(group kind, key) pairs in the reference's iteration order (embedding.py:465-477, :500, :557-558)."""
        order = [('sparse', self.ITEM_SPARSE_FEAT), ('array', self.ITEM_ARRAY_FEAT),
                 ('continual', self.ITEM_CONTINUAL_FEAT), ('item_emb', self.ITEM_EMB_FEAT)]
        if include_user:
            order += [('sparse', self.USER_SPARSE_FEAT), ('array', self.USER_ARRAY_FEAT),
                      ('continual', self.USER_CONTINUAL_FEAT)]
        if interaction:
            order.append(('sparse', self.INTERACTION_SPARSE_FEAT))
        return [(kind, k) for kind, keys in order for k in keys]

    def _raw_rows(self, kind, k, feature_array):
        """(T, d) rows of one sparse / array feature (table rows, bag sums) or (T, 1) values of a continual one."""
        t = feature_array[k].to(self._device()) if isinstance(feature_array, dict) else self.feat2tensor(feature_array, k)
        if kind == 'sparse':
            return self.sparse_emb[k](t.reshape(-1))
        if kind == 'array':
            return self.sparse_emb[k].bag_sum(t.reshape(-1, t.shape[-1]))
        return t.reshape(-1, 1).float()

    def _dense_input(self, k, feature_array):
        if isinstance(feature_array, dict):
            return feature_array[k].to(self._device()).float()
        return torch.from_numpy(pack_dense_feature(feature_array, k, self.ITEM_EMB_FEAT[k])).to(self._device())

    @staticmethod
    def _shape_of(feature_array):
        if isinstance(feature_array, dict):
            t = next(iter(feature_array.values()))
            return t.shape[0], t.shape[1]
        return len(feature_array), _seq_len(feature_array)

    @staticmethod
    def _is_empty(feature_array):
        if feature_array is None:
            return True
        if isinstance(feature_array, dict):
            return len(feature_array) == 0
        return len(feature_array) == 0 or len(feature_array[0]) == 0

    def _zeros(self, B, L):
        return torch.zeros(B, L, self.args.shared_hidden_units, device=self._device())

    def _summed_features(self, feature_array, include_user, gated):
        """Late (embedding.py:486-542) and gated (:216-264) fusion: every feature projected on its own and summed;
        gated multiplies each projection by sigmoid(gate(projection)) first."""
        if not gated and self._is_empty(feature_array):
            return self._zeros(len(feature_array), 0)
        B, L = self._shape_of(feature_array)
        total = None
        for kind, k in self._groups(include_user, interaction=True):
            if kind == 'item_emb':
                x = self._dense_input(k, feature_array)
                p = self.emb_transform[k](x).reshape(B * L, -1)
            else:
                lin = self.feature_projections[k]
                raw = self._raw_rows(kind, k, feature_array)
                if not gated:        # the running sum rides along as the projection's residual input
                    total = E.linear(raw, lin.weight, lin.bias, r1=total, c1=1.0 if total is not None else 0.0)
                    continue
                p = E.linear(raw, lin.weight, lin.bias)
            if gated:
                g = self.feature_gates[k][0]
                p = FO.sigmoid_gate(p, E.linear(p, g.weight, g.bias))
            total = p if total is None else E.combine_ad([total, p], [1.0, 1.0])
        return total.view(B, L, -1) if total is not None else self._zeros(B, L)

    def _early_features(self, feature_array, include_user):
        """Early fusion (embedding.py:544-611): ReLU(W [f_1 | f_2 | ... | fused modalities] + b) with W applied one
        column block per piece, each launch adding onto the one before."""
        if self._is_empty(feature_array):
            return self._zeros(len(feature_array), 0)
        B, L = self._shape_of(feature_array)
        d = self.args.shared_hidden_units
        pieces, modalities = [], []
        for kind, k in self._groups(include_user, interaction=include_user):
            if kind == 'item_emb':
                modalities.append(self.emb_transform[k](self._dense_input(k, feature_array), normalize_input=True))
            elif kind == 'continual':
                lin = self.feature_projections[k]
                pieces.append(E.linear(self._raw_rows(kind, k, feature_array), lin.weight, lin.bias))
            else:
                pieces.append(self._raw_rows(kind, k, feature_array))
        if modalities:
            pieces.append(self.cross_modal_fusion(modalities).reshape(B * L, d))
        if not pieces:
            return self._zeros(B, L)
        proj = self.early_fusion_projection_with_user if include_user else self.early_fusion_projection_no_user
        acc = None
        for i, piece in enumerate(pieces):
            acc = E.linear(piece, proj.weight, proj.bias if i == 0 else None, r1=acc, c1=1.0 if acc is not None else 0.0,
                           col_off=i * d)
        return FO.relu(acc).view(B, L, d)

    def get_feature_embeddings(self, feature_array, include_user=False, fusion_style='late'):
        """embedding.py:456-484."""
        if fusion_style == 'late':
            return self._summed_features(feature_array, include_user, gated=False)
        if fusion_style == 'gated':
            return self._summed_features(feature_array, include_user, gated=True)
        return self._early_features(feature_array, include_user)

    def embedding_parts(self, seq, feature_array, mask=None, include_user=False, fusion_style='early', pos_seq=None):
        """(base, features): the two halves that get_complete_embeddings adds (late) or concatenates (otherwise);
        features is None when the batch carries none (a zero block in the reference)."""
        base = self.get_base_embeddings(seq, mask, include_user)
        if pos_seq is not None:
            base = E.combine_ad([base.reshape(-1, base.shape[-1]), self.pos_emb(pos_seq.reshape(-1))],
                                [1.0, 1.0]).view(base.shape)
        if not self.has_features or self._is_empty(feature_array):
            return base, None
        return base, self.get_feature_embeddings(feature_array, include_user, fusion_style)

    def get_complete_embeddings(self, seq, feature_array, mask=None, include_user=False, fusion_style='late',
                                model_style='hstu', pos_seq=None):
        """embedding.py:613-652."""
        base, feat = self.embedding_parts(seq, feature_array, mask, include_user, fusion_style, pos_seq)
        if fusion_style == 'late':
            return FO.gelu(base if feat is None else E.combine_ad(
                [base.reshape(-1, base.shape[-1]), feat.reshape(-1, feat.shape[-1])], [1.0, 1.0]).view(base.shape))
        return torch.cat([base, torch.zeros_like(base) if feat is None else feat], dim=2)

    # ---- candidate table ----------------------------------------------------------------------
    @torch.no_grad()
    def _late_item_rows(self, ids, feat_rows=None, batch_size=1024):
        """get_complete_embeddings(ids[:, None], [[f] for f in feat_rows], include_user=False,
        fusion_style='late').squeeze(1).  Without feature groups: GELU fused into the row gather, one pass."""
        if not self.has_features:
            return ops.gather_gelu_rows(self.item_emb.weight, ids)
        out = []
        for s in range(0, ids.numel(), batch_size):
            chunk = ids[s:s + batch_size]
            feats = [[f] for f in feat_rows[s:s + batch_size]]
            out.append(self.get_complete_embeddings(chunk.unsqueeze(1), feats, include_user=False,
                                                    fusion_style='late').squeeze(1))
        return torch.cat(out, dim=0) if out else torch.empty(0, self.item_emb.embedding_dim, device=ids.device)

    @torch.no_grad()
    def get_item_embeddings(self, feat_dict=None, batch_size=1024):
        """embedding.py:686-735: (item_num + 1, d) candidate table, row 0 zeros, row i = late-fusion embedding of
        item i.  In the ids-only configuration `batch_size` only bounded the reference's memory and one fused pass
        covers every row."""
        ids = torch.arange(1, self.item_num + 1, device=self._device())
        feat_rows = None
        if self.has_features:
            feat_dict = feat_dict or {}
            feat_rows = [feat_dict.get(i, {}) for i in range(1, self.item_num + 1)]
        out = torch.empty(self.item_num + 1, self.item_emb.embedding_dim, device=ids.device)
        out[0].zero_()
        out[1:] = self._late_item_rows(ids, feat_rows, batch_size)
        return out

    @torch.no_grad()
    def save_item_embeddings(self, item_ids, retrieval_ids, feat_dict, save_path, batch_size=1024):
        """embedding.py:654-684: embedding.fbin (late-fusion rows of `item_ids`) and id.u64bin (`retrieval_ids`).
        As in the reference, feat_dict is indexed by POSITION in item_ids (feat_dict[i] for the i-th id)."""
        from pathlib import Path
        from .ann_files import save_emb
        ids = torch.as_tensor(list(item_ids), dtype=torch.int64, device=self._device())
        feat_rows = [feat_dict[i] for i in range(len(ids))] if self.has_features else None
        embs = self._late_item_rows(ids, feat_rows, batch_size).cpu().numpy().astype(np.float32)
        save_emb(embs, Path(save_path, 'embedding.fbin'))
        save_emb(np.array(list(retrieval_ids), dtype=np.uint64).reshape(-1, 1), Path(save_path, 'id.u64bin'))
