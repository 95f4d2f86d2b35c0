"""Drop-in for the id-embedding part of embedding.py's SharedEmbeddingModule, on the HIP gather / scatter-add.

Constructor, attribute names (`item_emb.weight`, ...) and method signatures follow embedding.py:142-214,
429-454, 613-652 so that reference checkpoints load and main.py / infer.py call sites work unchanged.
Only the ids-only configuration (all feature groups empty -- the configuration the benchmark and the
synthetic data use) is implemented; a non-empty feature group raises NotImplementedError.
"""
import torch
from torch import nn

from . import ops


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

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}"


class SharedEmbeddingConfig:
    """embedding.py:738-749."""

    def __init__(self, args):
        self.shared_hidden_units = max(getattr(args, 'hstu_hidden_units', 60), getattr(args, 'hydra_hidden_units', 60))
        args.shared_hidden_units = self.shared_hidden_units


class SharedEmbeddingModule(nn.Module):
    def __init__(self, user_num, item_num, feat_statistics, feat_types, args, interaction_vocab=None):
        super().__init__()
        if any(len(v) for v in feat_types.values()) or interaction_vocab:
            raise NotImplementedError("only the ids-only configuration (empty feature groups) is implemented")
        if not hasattr(args, 'shared_hidden_units'):
            SharedEmbeddingConfig(args)
        self.user_num, self.item_num, self.args, self.dev = user_num, item_num, args, args.device
        d = args.shared_hidden_units
        self.item_emb = HipEmbedding(item_num + 1, d, padding_idx=0)
        self.user_emb = HipEmbedding(user_num + 1, d, padding_idx=0)
        self.pos_emb = HipEmbedding(2 * args.maxlen + 1, d, padding_idx=0)
        self.action_emb = HipEmbedding(args.num_actions + 1, d, padding_idx=0)
        self.time_span_emb = HipEmbedding(getattr(args, 'time_span', 512), d, padding_idx=0)
        self.dwell_time_emb = HipEmbedding(7, d, padding_idx=0)
        # embedding.py:56,181-185 registers a LayerNorm even with no modality: kept for state_dict parity
        self.cross_modal_fusion = nn.Module()
        self.cross_modal_fusion.norm = nn.LayerNorm(d)

    def get_base_embeddings(self, seq, mask=None, include_user=False):
        """embedding.py:429-454."""
        if include_user and mask is not None:
            return ops.gather_routed(self.item_emb.weight, self.user_emb.weight, seq, mask)
        return self.item_emb(seq)

    def get_complete_embeddings(self, seq, feature_array, mask=None, include_user=False, fusion_style='late',
                                model_style='hstu', pos_seq=None):
        """embedding.py:613-652 with the ids-only zero feature block."""
        base = self.get_base_embeddings(seq, mask, include_user)
        if pos_seq is not None:
            base = base + self.pos_emb(pos_seq)
        if fusion_style == 'late':
            return torch.nn.functional.gelu(base)
        return torch.cat([base, torch.zeros_like(base)], dim=2)

    @torch.no_grad()
    def _late_item_rows(self, ids):
        """GELU(item_emb[ids]) fused into the row gather: what get_complete_embeddings(ids[:, None], ...,
        include_user=False, fusion_style='late').squeeze(1) returns in the ids-only configuration."""
        return ops.gather_gelu_rows(self.item_emb.weight, ids)

    @torch.no_grad()
    def get_item_embeddings(self, feat_dict=None, batch_size=1024):
        """embedding.py:686-735: (item_num + 1, d) candidate table, row 0 zeros, row i = late-fusion embedding of
        item i.  `batch_size` only bounded the reference's memory; one fused pass covers every row here."""
        ids = torch.arange(1, self.item_num + 1, device=self.item_emb.weight.device)
        out = torch.empty(self.item_num + 1, self.item_emb.embedding_dim, device=ids.device)
        out[0].zero_()
        out[1:] = self._late_item_rows(ids)
        return out

    @torch.no_grad()
    def save_item_embeddings(self, item_ids, retrieval_ids, feat_dict, save_path, batch_size=1024):
        """embedding.py:654-684: embedding.fbin (late-fusion rows of `item_ids`) and id.u64bin (`retrieval_ids`)."""
        import numpy as np
        from pathlib import Path
        from .ann_files import save_emb
        ids = torch.as_tensor(list(item_ids), dtype=torch.int64, device=self.item_emb.weight.device)
        embs = self._late_item_rows(ids).cpu().numpy().astype(np.float32)
        save_emb(embs, Path(save_path, 'embedding.fbin'))
        save_emb(np.array(list(retrieval_ids), dtype=np.uint64).reshape(-1, 1), Path(save_path, 'id.u64bin'))
