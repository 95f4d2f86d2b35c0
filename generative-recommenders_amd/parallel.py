"""Data-parallel gradient exchange for the Hydra train step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference trains in one process (main.py:672-715) and has no gradient exchange; this is the multi-GPU
form of the same step: every rank runs forward + backward on its own batch, `GradientSync` averages the
gradients in place, then clip + optimizer run identically on every rank.

Layout choices for a point-to-point xGMI fabric (ring collectives are bound by one link, so few large
messages beat many small ones):
  * every parameter of at least `big_bytes` (the item / user tables: 1 GB and 256 MB of dense gradient at
    4 M x 64) is reduced in place, one collective each, no staging copy;
  * all remaining gradients (the encoder's ~190 small tensors) are packed into ONE flat fp32 bucket, reduced
    with one collective, scaled and unpacked with foreach copies.
All collectives are launched asynchronously back to back and waited for together.  A parameter whose
gradient is None on this rank takes part as zeros, so ranks never disagree about the message sizes.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class GradientSync:
    def __init__(self, params, group=None, big_bytes: int = 32 << 20, average: bool = True):
        if isinstance(params, torch.nn.Module):
            params = params.parameters()
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.average = average
        self.big = [p for p in self.params if p.numel() * p.element_size() >= big_bytes]
        self.small = [p for p in self.params if p.numel() * p.element_size() < big_bytes]
        for p in self.small:
            if p.dtype != torch.float32:
                raise TypeError("GradientSync packs small gradients as fp32")
        self._flat: Optional[torch.Tensor] = None
        self._views: List[torch.Tensor] = []

    def _bucket(self):
        if self._flat is None and self.small:
            dev = self.small[0].device
            n = sum(p.numel() for p in self.small)
            self._flat = torch.empty(n, dtype=torch.float32, device=dev)
            off = 0
            for p in self.small:
                self._views.append(self._flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self._flat

    @torch.no_grad()
    def __call__(self) -> None:
        if self.world == 1:
            return
        scale = 1.0 / self.world if self.average else 1.0
        works = []
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        flat = self._bucket()
        if flat is not None:
            for p in self.small:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            torch._foreach_copy_(self._views, [p.grad for p in self.small])
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if flat is not None:
            if scale != 1.0:
                flat.mul_(scale)
            torch._foreach_copy_([p.grad for p in self.small], self._views)
        if scale != 1.0 and self.big:
            torch._foreach_mul_([p.grad for p in self.big], scale)

    def bytes_per_step(self) -> int:
        """Gradient bytes each rank contributes per step (the payload of the collectives)."""
        return sum(p.numel() * p.element_size() for p in self.params)
