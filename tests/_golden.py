"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import functools
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def prefixes(store, head):
    return sorted({k.split("/")[0] for k in store if k.startswith(head)})


def t(store, key):
    a = store[key]
    return torch.from_numpy(np.array(a)) if a.shape else torch.tensor(a.item())


def js(store, key):
    """A JSON document stored as a 0-d string array (the reference's list-of-dicts feature format)."""
    import json
    return json.loads(str(store[key]))


def params(store, prefix):
    head = prefix + "/param/"
    return {k[len(head):]: torch.from_numpy(np.array(v)) for k, v in store.items() if k.startswith(head)}


def grads(store, prefix):
    head = prefix + "/grad/"
    return {k[len(head):]: torch.from_numpy(np.array(v)) for k, v in store.items() if k.startswith(head)}


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if not torch.allclose(a, b, rtol=rtol, atol=atol):
        err = (a - b).abs()
        raise AssertionError(f"{what}: max abs err {err.max().item():.3e} "
                             f"(ref max {b.abs().max().item():.3e}), rtol={rtol}, atol={atol}")


def assemble_train_draws(store, prefix, loss_mask, num_hard=153, num_inb=359, q_chunk=256):
    """(hard_ids (M, .), inb_idx (M, .)) of a G10 train-step fixture, from what the reference itself produced: its
    mined top-50 ids (`mined_ids`) and the outputs of its torch.randint calls in call order (`draw{i}`).  Mapping of
    the draws (hydra3.py:892-950, per chunk of q_chunk masked rows): 50 mined ids < num_hard, so the hard keys are
    the mined ids tiled and cut (no draw); the in-batch keys are either one draw r of (n, num_inb) values in
    [0, M-1) shifted past the row's own index, or -- when the batch has fewer than num_inb other rows -- all other
    rows plus one draw of random item rows that fills the shortfall (these are item-table keys: they join hard_ids)."""
    mask = loss_mask.bool()
    B, L = mask.shape
    mined = t(store, prefix + "/mined_ids")
    K = mined.shape[1]
    per_row = mined.unsqueeze(1).expand(-1, L, -1)[mask]                    # (M, 50)
    M = per_row.shape[0]
    tiled = per_row.repeat(1, (num_hard + K - 1) // K)[:, :num_hard]
    draws = [t(store, f"{prefix}/draw{i}") for i in range(int(store[prefix + "/n_draws"]))]
    hard, inb, di = [], [], 0
    rows = torch.arange(M)
    for s0 in range(0, M, q_chunk):
        e0 = min(s0 + q_chunk, M)
        if M - 1 >= num_inb:
            r = draws[di]
            di += 1
            assert tuple(r.shape) == (e0 - s0, num_inb)
            inb.append(r + (r >= rows[s0:e0].unsqueeze(1)).long())
            hard.append(tiled[s0:e0])
        else:
            base = rows.unsqueeze(0).repeat(M, 1)
            inb.append(base[base != rows.unsqueeze(1)].view(M, M - 1)[s0:e0])
            fill = draws[di]
            di += 1
            assert tuple(fill.shape) == (e0 - s0, num_inb - (M - 1))
            hard.append(torch.cat([tiled[s0:e0], fill], dim=1))
    assert di == len(draws)
    return torch.cat(hard), torch.cat(inb)


BLOCK_NAMES = ("hydra_first", "hydra.0", "hydra.1", "multi_scale_encoder.short_term_encoder",
               "multi_scale_encoder.mid_term_encoder", "multi_scale_encoder.long_term_encoder")


def block_drop_multipliers(store, prefix):
    """{block module name: (B, L', d) multiplier} of a train-step fixture recorded with the blocks' dropout on: the
    reference multiplied each block's input by mask / (1 - p)."""
    head = prefix + "/dropmask/"
    p = float(store[prefix + "/p_drop"])
    return {k[len(head):]: torch.from_numpy(np.array(v)).float() / (1.0 - p) for k, v in store.items() if k.startswith(head)}


class FixedDropout(torch.nn.Module):
    """Stands in for a block's nn.Dropout: multiplies by a given tensor (what nn.Dropout does with its own draw)."""

    def __init__(self, mult):
        super().__init__()
        self.mult = mult

    def forward(self, x):
        return x * self.mult
