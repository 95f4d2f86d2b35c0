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
