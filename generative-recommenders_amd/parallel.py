"""Data-parallel gradient exchange for the Hydra train step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference trains in one process (main.py:672-715) and has no gradient exchange; this is the multi-GPU
form of the same step: every rank runs forward + backward on its own batch, `GradientSync` averages the
gradients in place, then clip + optimizer run identically on every rank.

Layout choices for a point-to-point xGMI fabric (ring collectives are bound by one link, so few large
messages beat many small ones):
  * every parameter of at least `big_bytes` (the item / user tables: 1 GB and 256 MB of dense gradient at
    4 M x 64) is exchanged SPARSELY by default: a step touches a few hundred thousand of the 4 M rows, so each
    rank sends only its touched (row id, gradient row) pairs -- one all_gather of ids and one of rows per table,
    padded to the largest count -- and every rank rebuilds the same dense gradient by adding all ranks' rows in
    rank order with the deterministic scatter-add.  (A ring all-reduce of the dense gradient moves
    2 * (N-1)/N * 1.28 GB per rank and step; the touched rows are ~64 MB per rank.)  `sparse_tables=False`
    reduces them in place instead, one collective each, no staging copy;
  * all remaining gradients (the encoder's ~190 small tensors) are packed into ONE flat fp32 bucket, reduced
    with one collective, scaled and unpacked with foreach copies.
All collectives are launched asynchronously back to back and waited for together.  A parameter whose
gradient is None on this rank takes part as zeros, so ranks never disagree about the message sizes.
"""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if _host_staged(group, t):
                host = t.data.cpu()
                dist.broadcast(host, src=src, group=group)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src=src, group=group)


# Diagnostics only: hand gloo the GPU tensors themselves (the form that was seen to lock up), e.g. to log it.
GLOO_ON_DEVICE_TENSORS = os.environ.get("HMX_GLOO_DEVICE_TENSORS", "0") == "1"


class _Done:
    def wait(self):
        return True


def _host_staged(group, t) -> bool:
    """gloo with GPU tensors (tests and rehearsals with several ranks on one GPU): collectives go through host copies
    and finish before returning; gloo's own handling of GPU tensors was seen to lock up now and then there."""
    return t.is_cuda and dist.get_backend(group) == "gloo" and not GLOO_ON_DEVICE_TENSORS


def _all_gather_flat(out, t, group, async_op=False):
    if _host_staged(group, t):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, t.cpu(), group=group)
        out.copy_(host)
        return _Done()
    return dist.all_gather_into_tensor(out, t, group=group, async_op=async_op)


def _all_reduce_sum(t, group, async_op=False):
    if _host_staged(group, t):
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _hip_row_ops():
    from . import ops
    return (ops.gather_rows_padded,
            lambda rows, ids, out: ops.scatter_add_rows(rows, ids, out.shape[0], out=out),
            ops.touched_rows, ops.touched_flags)


class GradientSync:
    """row_ops = (gather(table, ids) -> rows (zero rows for ids < 0), scatter_add(rows, ids, out) adding rows into out,
    ids < 0 ignored [, touched(grad) -> ascending ids of the non-zero rows [, touched_flags(grad) -> one byte per row,
    non-zero where the row was written, computed without a host sync]]): the HIP kernels by default; injectable so that
    the exchange logic can be exercised on CPU tensors with gloo."""

    def __init__(self, params, group=None, big_bytes: int = 32 << 20, average: bool = True,
                 sparse_tables: bool = True, row_ops=None):
        if isinstance(params, torch.nn.Module):
            params = params.parameters()
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.average = average
        self.sparse_tables = sparse_tables
        self.row_ops = row_ops
        self.last_stats = {}
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
        sparse = [p for p in self.big if self.sparse_tables and p.dim() == 2]
        dense_big = [p for p in self.big if not (self.sparse_tables and p.dim() == 2)]
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        for p in dense_big:
            works.append(_all_reduce_sum(p.grad, self.group, async_op=True))
            if p.grad.is_cuda:       # every row may change: a row-flagged table gradient is dense from here on
                from . import ops
                st = ops.table_grad_state(p)
                if st is not None:
                    st.mark_dense()
        pending_sparse = self._sparse_begin(sparse, scale)
        flat = self._bucket()
        if flat is not None:
            for p in self.small:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            torch._foreach_copy_(self._views, [p.grad for p in self.small])
            works.append(_all_reduce_sum(flat, self.group, async_op=True))
        for w in works:
            w.wait()
        if flat is not None:
            if scale != 1.0:
                flat.mul_(scale)
            torch._foreach_copy_([p.grad for p in self.small], self._views)
        for st in pending_sparse:
            self._sparse_finish(st)
        if scale != 1.0 and dense_big:
            torch._foreach_mul_([p.grad for p in dense_big], scale)

    # ---- sparse exchange of one table gradient ------------------------------------------------------------
    def _ops(self, t):
        if self.row_ops is not None:
            return self.row_ops
        if not t.is_cuda:
            raise RuntimeError("GradientSync: the sparse table exchange runs on the HIP kernels (CUDA tensors); "
                               "pass row_ops to exercise it elsewhere")
        return _hip_row_ops()

    def _sparse_begin(self, tables, scale=1.0):
        """Starts the row exchange of every table gradient in `tables`.  The touched-row counts of ALL tables travel in
        one collective and reach the host in one copy (the step's only host sync here): with row flags at hand
        (touched_flags) nothing before that copy waits for the device, and the padded id lists are then cut from the
        flags at exactly the largest count over ranks."""
        if not tables:
            return []
        prep, cnts = [], []
        for p in tables:
            g = p.grad
            row_ops = self._ops(g)
            if len(row_ops) > 3:
                flags = row_ops[3](g)
                prep.append((p, None, flags))
                cnts.append((flags != 0).sum(dtype=torch.int64).reshape(1))
            else:
                ids = row_ops[2](g) if len(row_ops) > 2 else (g != 0).any(dim=1).nonzero().flatten()
                prep.append((p, ids, None))
                cnts.append(torch.tensor([ids.numel()], dtype=torch.int64, device=g.device))
        dev = tables[0].grad.device
        counts = torch.cat([c.to(dev) for c in cnts])
        all_counts = torch.empty(self.world * len(tables), dtype=torch.int64, device=dev)
        _all_gather_flat(all_counts, counts, self.group)
        both = torch.stack([all_counts.view(self.world, len(tables)).max(dim=0).values, counts]).tolist()
        states = []
        for (p, ids, flags), n_max, n_own in zip(prep, both[0], both[1]):
            g = p.grad
            gather = self._ops(g)[0]
            n_pad = max(int(n_max), 1)
            if flags is not None:
                pad_ids = torch.nonzero_static(flags, size=n_pad, fill_value=-1).flatten()     # ascending, -1 after the last
            else:
                pad_ids = torch.full((n_pad,), -1, dtype=torch.int64, device=g.device)
                pad_ids[:ids.numel()] = ids
            pad_rows = gather(g, pad_ids)                # ids < 0 give zero rows
            if scale != 1.0:
                pad_rows.mul_(scale)                     # each rank scales what it sends: 1 / world of the gathered volume
            out_ids = torch.empty(self.world * n_pad, dtype=torch.int64, device=g.device)
            out_rows = torch.empty(self.world * n_pad, g.shape[1], dtype=g.dtype, device=g.device)
            w1 = _all_gather_flat(out_ids, pad_ids, self.group, async_op=True)
            w2 = _all_gather_flat(out_rows, pad_rows, self.group, async_op=True)
            self.last_stats[id(p)] = {"rows_sent": int(n_own), "rows_padded": n_pad}
            states.append((p, pad_ids, out_ids, out_rows, (w1, w2)))
        return states

    def _sparse_finish(self, state):
        p, pad_ids, out_ids, out_rows, works = state
        for w in works:
            w.wait()
        g = p.grad
        scatter_add = self._ops(g)[1]
        # This rank's rows come back inside the gathered set: zero them first.  The -1 padding is pointed at row 0, which
        # is either one of this rank's rows or was never written here (zero already), so no list has to be compacted.
        g.index_fill_(0, pad_ids.clamp_min(0), 0.0)
        scatter_add(out_rows, out_ids, g)                # every rank adds all rows in rank order: same bits everywhere

    def bytes_per_step(self) -> int:
        """Gradient bytes each rank contributes per step (the payload of the collectives)."""
        return sum(p.numel() * p.element_size() for p in self.params)
