"""Seeded synthetic inputs of the shapes the benchmark configurations name (no real data offline)."""
import torch


def item_table(n_items, d, device, seed=1234, std=1.0):
    """(n_items + 1, d) fp32 table, N(0, std^2), padding row 0 zeroed."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.randn(n_items + 1, d, device=device, generator=g) * std
    t[0] = 0
    return t


def queries(n_q, d, device, seed=4321):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randn(n_q, d, device=device, generator=g)


def train_batch(B, L, n_users, n_items, device, seed=99, pad_frac=0.0):
    """One batch in the layout of main.py:674-675 (ids-only): slot 0 of every sequence is the user token
    (token type 2), the rest are items (type 1); optional left padding; monotone timestamps."""
    g = torch.Generator(device=device).manual_seed(seed)
    ri = lambda lo, hi, shape: torch.randint(lo, hi, shape, device=device, generator=g)
    seq = ri(1, n_items + 1, (B, L))
    tok = torch.ones(B, L, dtype=torch.long, device=device)
    pads = (torch.rand(B, device=device, generator=g) * pad_frac * L).long()
    pos_idx = torch.arange(L, device=device).unsqueeze(0)
    is_pad = pos_idx < pads.unsqueeze(1)
    is_user = pos_idx == pads.unsqueeze(1)
    seq = torch.where(is_user, ri(1, n_users + 1, (B, L)), seq)
    tok = torch.where(is_user, torch.full_like(tok, 2), tok)
    seq = seq.masked_fill(is_pad, 0)
    tok = tok.masked_fill(is_pad, 0)
    ts = torch.cumsum(ri(1, 1000, (B, L)), dim=1).masked_fill(is_pad, 0)
    valid = seq != 0
    return dict(seq=seq, tok=tok, ts=ts,
                act=ri(0, 6, (B, L)) * valid, dwell=ri(0, 7, (B, L)) * valid,
                pos=ri(1, n_items + 1, (B, L)) * valid, neg=ri(1, n_items + 1, (B, L)) * valid,
                next_tok=valid.long())
