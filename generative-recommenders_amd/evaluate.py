"""Sampled-negative evaluation (main.py:170-239): rank of each user's last positive among 99 random items."""
import math

import numpy as np
import torch

from . import ops


@torch.no_grad()
def evaluate_model(model, data_loader, device, num_neg_samples=99, seed=42, neg_items_fn=None):
    """Drop-in for main.py:170-239 with the encoder, the row gathers and the projection on the HIP kernels.

    Per batch: user embedding = last position of log2feats; users whose last positive id is 0 are skipped; 99 negative
    item ids are drawn with torch.manual_seed(seed + batch_idx) on `device` (main.py:207-211); rank = number of
    negatives scoring >= the positive; HR@10 and NDCG@10 over all ranks.  `neg_items_fn(batch_idx, n_valid)` replaces
    the draw (the GPU generator's stream differs from the CPU's, so a test that compares with a CPU run passes the
    CPU run's ids)."""
    model.eval()
    torch.manual_seed(seed)
    np.random.seed(seed)
    ranks = []
    table = model.shared_embeddings.item_emb.weight
    for batch_idx, batch in enumerate(data_loader):
        (seq, targets, pos, neg, token_type, next_token_type, next_action_type, seq_feat, pos_feat, neg_feat, ts,
         action_type, dwell_bins) = batch
        seq, pos, token_type, ts = seq.to(device), pos.to(device), token_type.to(device), ts.to(device)
        action_type, dwell_bins = action_type.to(device), dwell_bins.to(device)
        user = model.log2feats(seq, seq_feat, token_type, ts, action_type, dwell_bins)[:, -1, :]
        valid = pos[:, -1] != 0
        if not bool(valid.any()):
            continue
        user = user[valid].contiguous()
        pos_items = pos[:, -1][valid]
        n = user.shape[0]
        total_items = table.shape[0] - 1
        torch.manual_seed(seed + batch_idx)
        if neg_items_fn is not None:
            neg_items = neg_items_fn(batch_idx, n).to(device)
        else:
            neg_items = torch.randint(1, total_items + 1, (n, num_neg_samples), device=device)
        cand = torch.cat([pos_items.unsqueeze(1), neg_items], dim=1)                 # (n, 1 + negatives)
        rows = model._project_small(ops.gather_rows(table, cand.reshape(-1)))      # embedding_projection
        scores = torch.bmm(rows.view(n, cand.shape[1], -1), user.unsqueeze(2)).squeeze(2)
        ranks.extend((scores[:, 1:] >= scores[:, :1]).sum(1).cpu().tolist())
    ranks = np.array(ranks)
    if ranks.size == 0:
        return {'hr@10': float('nan'), 'ndcg@10': float('nan'), 'num_samples': 0}
    return {'hr@10': float(np.mean(ranks < 10)),
            'ndcg@10': float(np.mean([1.0 / math.log2(r + 2.0) if r < 10 else 0.0 for r in ranks])),
            'num_samples': int(ranks.size)}
